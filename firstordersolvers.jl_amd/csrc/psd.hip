// Batched projection onto the PSD cone -- IndPSD(scaling=true) of ProximalOperators.jl as the reference uses it
// (conemap :SDP, src/cones.jl:11; called from ConeProduct.prox!/proxDual!, src/cones.jl:80-94).
//
// Input per cone: the packed lower triangle (column-major) of a k x k symmetric matrix with the off-diagonal
// entries pre-multiplied by sqrt(2) (MathProgBase svec).  ProximalOperators' vector method: scale the DIAGONAL by
// sqrt(2) (the packed matrix is then sqrt(2) x the true one), symmetric eigen-decomposition, clamp eigenvalues
// at 0, rebuild, repack the lower triangle, scale the diagonal back by 1/sqrt(2).  The dual copy of the cone is
// Moreau's  y = x + P(-x)  (cones.jl:80-85).
//
// One workgroup (256 threads) per (cone, part) matrix; ONE k x k array G lives in LDS (global scratch for orders
// that do not fit).  Eigen-solver: one-sided (Hestenes) Jacobi on the SHIFTED matrix M' = M + sigma I:
//   * G starts as M' and column pairs are rotated until all columns are mutually orthogonal; then G = V diag(l'),
//     |l'_j| = ||g_j||, v_j = g_j / ||g_j||: the eigenvectors are read off G itself, no accumulated V is kept
//     (halves LDS footprint and traffic).
//   * sigma = 0.505 ||M||_F >= |lambda_min| / 2 is enough: an eigenvalue kept by the projection (lambda > 0) has
//     ||g|| = lambda + sigma > sigma, every other column has ||g|| <= max(sigma, |lambda_min| - sigma) <= sigma, and two
//     columns whose l' = +-same magnitude (which plain one-sided Jacobi cannot separate) are both dropped ones.
//     So  P = sum_{||g_j|| > sigma} (||g_j|| - sigma) / ||g_j||^2  g_j g_j'.
//   * column pairs of a round-robin tournament step are disjoint, so a step needs one barrier; every pair is
//     handled by `tpp` lanes (a power of two <= 64) that reduce the three column dot products in-register (DPP
//     for the common 8-lane case).
// Accuracy: absolute error O(k eps ||M||), the class of LAPACK's dspev that the reference calls.
#include "fos_internal.hpp"
#include "dev_common.hpp"      // dpp_f64, swap_sum, wave_sum

namespace fos {

constexpr int PSD_THREADS = 256;
constexpr int PSD_MAX_SWEEPS = 40;
constexpr double SQRT2 = 1.4142135623730951;
constexpr double INV_SQRT2 = 0.7071067811865475;

__host__ __device__ inline int psd_ld(int k) {
    // leading dimension == 8 (mod 32) doubles: the 4 pair groups of a 32-lane half hit distinct bank quarters
    int ld = ((k + 23) / 32) * 32 + 8;
    if (ld < k) ld += 32;
    return ld;
}

constexpr int PSD_RED_SCRATCH = 512;      // doubles of LDS behind wgt / inv: the reducer of jacobi64_regs (order 64 only; [2][8][32])
__host__ inline size_t psd_lds_bytes(int k) { return (size_t)(32 + k * psd_ld(k) + 2 * k + 16 + (k == 64 ? PSD_RED_SCRATCH : 0)) * sizeof(double); }

__device__ __forceinline__ void idx_to_ij(int idx, int k, int& i, int& j) {
    // packed lower triangle, column-major: column j starts at S(j) = j k - j (j-1)/2
    const double b = 2.0 * k + 1.0;
    int jj = (int)floor((b - sqrt(b * b - 8.0 * (double)idx)) * 0.5);
    if (jj < 0) jj = 0;
    if (jj > k - 1) jj = k - 1;
    while (jj + 1 < k && ((jj + 1) * k - ((jj + 1) * jj) / 2) <= idx) ++jj;
    while (jj > 0 && (jj * k - (jj * (jj - 1)) / 2) > idx) --jj;
    j = jj;
    i = jj + (idx - (jj * k - (jj * (jj - 1)) / 2));
}

// ---- in-register reductions over a lane group (dpp_f64: dev_common.hpp)
// sum over aligned groups of 8 lanes: lane i + lane 7-i (row_half_mirror), then xor 2, xor 1 inside the quad
__device__ __forceinline__ double group8_sum(double v) {
    v += dpp_f64<0x141>(v);        // row_half_mirror
    v += dpp_f64<0x4E>(v);         // quad_perm [2,3,0,1]
    v += dpp_f64<0xB1>(v);         // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ double group16_sum(double v) {
    v = group8_sum(v);
    v += dpp_f64<0x140>(v);        // row_mirror: the other half of the 16-lane row
    return v;
}
__device__ __forceinline__ double group32_sum(double v) {
    v = group16_sum(v);
    v = swap_sum<16>(v);           // (gfx950 row exchange instead of a ds_bpermute round trip: dev_common.hpp)
    return v;
}
// (group_sum(v, tpp), any power of two: dev_common.hpp)

// 1/sqrt(x) to ~1 ulp: hardware estimate + two Newton steps (no IEEE division/sqrt sequences on the critical path)
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

template <int TPP>
__device__ __forceinline__ double groupc_sum(double v) {
    if constexpr (TPP == 8) return group8_sum(v);
    else if constexpr (TPP == 16) return group16_sum(v);
    else return group32_sum(v);
}

// ---- the two dense contractions of an order-64 projection on the matrix cores (v_mfma_f64_16x16x4_f64)
// Operand maps (cdna_hip_programming.md section 3, f64 form): lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// the 16 x 16 result has column l & 15 and rows (l >> 4) + 4 r in registers r = 0..3.  A 64 x 64 product is 4 x 4 tiles of
// 16 k-steps; the wavefronts of the workgroup share the tiles.
typedef double v4d __attribute__((ext_vector_type(4)));

// A sweep in which every rotation was TINY -- both the cosine between the two columns and the sine of the rotation angle
// below 1e-8 -- leaves, by the quadratic convergence of the cyclic method, every pair below the rotation threshold: the
// sweep that would only confirm it (no rotation, but all the dot products) is skipped.  The angle matters, not only the
// cosine: inside a cluster of (nearly) equal eigenvalues two almost orthogonal columns are still turned by a LARGE angle,
// which disturbs their products with all other columns to first order -- such a sweep is never the last one.
constexpr double JACOBI_SMALL2 = 1e-16;

// One-sided Jacobi on a 64 x 64 array: THREADS / TPP == 32 pair slots, one pair per slot and step; the 64 / TPP
// elements a lane owns of the two columns stay in registers between the dot products and the rotation, and the
// round-robin partner indices advance by one (mod 63) per step -- no division, no LDS re-read.
template <int THREADS, int TPP>
__device__ __forceinline__ int jacobi64(double* __restrict__ G, const int ld, const int tid, const double tol2) {
    static_assert(THREADS / TPP == 32, "one column pair per slot");
    constexpr int EPL = 64 / TPP;
    const int pr = tid / TPP, lig = tid % TPP;
    int p = pr, q = 63 - pr;                      // step 0: pr = 0 plays (0, 63); pr >= 1 plays (pr, 63 - pr)
    int sweep = 0;
    for (; sweep < PSD_MAX_SWEEPS; ++sweep) {
        int rotated = 0, big = 0;
        for (int step = 0; step < 63; ++step) {
            double* gp = G + (size_t)p * ld + lig;
            double* gq = G + (size_t)q * ld + lig;
            double u[EPL], v[EPL];
            double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                u[e] = gp[e * TPP]; v[e] = gq[e * TPP];
                a += u[e] * u[e]; b += v[e] * v[e]; g += u[e] * v[e];
            }
            a = groupc_sum<TPP>(a); b = groupc_sum<TPP>(b); g = groupc_sum<TPP>(g);
            const double gg = g * g, ab = a * b;
            if (gg > tol2 * ab) {
                rotated = 1;
                const double d = b - a;
                const double rh = fast_rsqrt(d * d + 4.0 * gg);
                const double c2 = 0.5 + 0.5 * fabs(d) * rh;
                const double rc = fast_rsqrt(c2);
                const double cs = c2 * rc;
                const double sn = copysign(g * rh * rc, d * g);
                if (gg > JACOBI_SMALL2 * ab || sn * sn > JACOBI_SMALL2) big = 1;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    gp[e * TPP] = cs * u[e] - sn * v[e];
                    gq[e * TPP] = sn * u[e] + cs * v[e];
                }
            }
            // next step of the tournament: player 63 stays, the others move one seat
            p = (p + 1 == 63) ? 0 : p + 1;
            if (pr != 0) q = (q + 1 == 63) ? 0 : q + 1;
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) break;
        if (!__syncthreads_or(big)) break;
    }
    return sweep + 1;
}

// WARM: start the Jacobi iteration from G0 = (M + sigma I) V_prev, V_prev = the eigenvector basis this (cone, copy)
// ended with at the previous outer iteration (read from vin, the new basis is written to vout).  The iterates of the
// solver change slowly, so V_prev nearly diagonalises the new matrix and 3-5 sweeps replace 9-10 (the convergence
// test -- a full sweep without a rotation -- is unchanged, so accuracy does not depend on the start).  Orders <= 64.
template <int NW>
__device__ __forceinline__ int jacobi64_regs(double* __restrict__ G, const int ld, const int tid, const double tol2, double* scratch);

template <bool USE_LDS, bool WARM, int THREADS>
__device__ __forceinline__ void psd_block(d2* __restrict__ out, const d2* __restrict__ in,
                                          const ConeDesc* __restrict__ cones,
                                          double* __restrict__ gscratch, size_t scratch_stride,
                                          const double* __restrict__ vin, double* __restrict__ vout,
                                          size_t vstride, int have_prev, int* __restrict__ stats, int phase_limit, double* __restrict__ smem) {
    const int tid = threadIdx.x;
    const int cone = blockIdx.x >> 1, part = blockIdx.x & 1;
    const ConeDesc cd = cones[cone];
    const int k = cd.k, len = cd.len, ld = psd_ld(k);
    const bool dual = (cd.dual_part == part);
    const double sgn = dual ? -1.0 : 1.0;
    const double* __restrict__ x = reinterpret_cast<const double*>(in + cd.start) + part;   // element idx at x[2 idx]
    double* __restrict__ y = reinterpret_cast<double*>(out + cd.start) + part;

    // all LDS comes from the one dynamic array (no static __shared__ in front of it: keeps its base 16-byte aligned)
    double* red = smem;                          // [0..15] wave partials, [16] sigma
    double* G;                                   // address space known at compile time: ds_* vs global_* accesses
    if constexpr (USE_LDS) G = smem + 32;
    else G = gscratch + (size_t)blockIdx.x * scratch_stride;
    double* wgt = G + (size_t)k * ld;           // k eigen-weights

    // ---- load: M = smat(sgn x) with the diagonal scaled by sqrt(2)
    double fro = 0.0;
    for (int idx = tid; idx < len; idx += THREADS) {
        int i, j;
        idx_to_ij(idx, k, i, j);
        double v = sgn * x[2 * (int64_t)idx];
        if (i == j) { v *= SQRT2; fro += v * v; }
        else fro += 2.0 * v * v;
        G[i + (size_t)j * ld] = v;
        G[j + (size_t)i * ld] = v;
    }
    fro = wave_sum(fro);
    if ((tid & 63) == 0) red[tid >> 6] = fro;
    __syncthreads();
    // WARM needs every shifted eigenvalue strictly positive (well defined column directions): sigma > |lambda_min|
    if (tid == 0) {
        double f2 = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) f2 += red[w];
        red[16] = (WARM ? 1.001 : 0.505) * sqrt(f2);
    }
    __syncthreads();
    const double sigma = red[16];
    for (int i = tid; i < k; i += THREADS) G[i + (size_t)i * ld] += sigma;
    __syncthreads();
    if (phase_limit == 1) return;               // diagnostic builds of the phase profile only (tools/psd_phases.py)

    if constexpr (WARM) {
        if (have_prev && sigma > 0.0) {
            const double* __restrict__ Vp = vin + (size_t)blockIdx.x * vstride;
            constexpr int NW = THREADS / 64;
            if (k == 64) {
                // G0 = M' V_prev on the matrix cores.  Tile tt = wave + NW q: column block jb = tt & 3 (the same for all tiles
                // of a wavefront, whose 16 x 64 slab of V_prev -- 8 KB, contiguous -- is fetched ONCE: 16 loads per lane, each
                // k-step's B operand), row block ib = tt >> 2; A operands = columns of the symmetric M' read from LDS.
                constexpr int NT = 16 / NW;                    // tiles per wavefront
                const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
                const int jb = w & 3, lr = lane & 15, lk = lane >> 4;
                double bv[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) bv[kk] = Vp[(size_t)(jb * 16 + lr) * 64 + kk * 4 + lk];
                v4d acc[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const int ib = (w + NW * q) >> 2;
                    acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
                    const double* __restrict__ ga = G + ib * 16 + lr + (size_t)lk * ld;
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk)
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[(size_t)kk * 4 * ld], bv[kk], acc[q], 0, 0, 0);
                }
                __syncthreads();                               // every wavefront has read M' before anyone overwrites it
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const int ib = (w + NW * q) >> 2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) G[ib * 16 + lk + 4 * r + (size_t)(jb * 16 + lr) * ld] = acc[q][r];
                }
                __syncthreads();
            } else {
                // G0 = M' V_prev : thread (row i = lane, columns j = wave + NW jj); M' rows from LDS, V_prev (wave-uniform) from L2
                constexpr int NJ = 64 / NW;
                const int i = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: V_prev by scalar loads
                double acc[NJ];
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) acc[jj] = 0.0;
                if (i < k) {
                    for (int t = 0; t < k; ++t) {
                        const double mv = G[i + (size_t)t * ld];
#pragma unroll
                        for (int jj = 0; jj < NJ; ++jj) {
                            const int j = w + NW * jj;
                            if (j < k) acc[jj] += mv * Vp[t + (size_t)j * k];
                        }
                    }
                }
                __syncthreads();
                if (i < k) {
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) {
                        const int j = w + NW * jj;
                        if (j < k) G[i + (size_t)j * ld] = acc[jj];
                    }
                }
                __syncthreads();
            }
        }
    }

    if (phase_limit == 2) return;

    // ---- one-sided Jacobi sweeps
    int nsweeps = 0;
    const int K = (k + 1) & ~1;               // even number of players (one bye when k is odd)
    const int npair = K >> 1;
    int tpp = 64;
    while (tpp > 1 && npair * tpp > THREADS) tpp >>= 1;
    const int sh = 31 - __clz(tpp);
    const int slot = tid >> sh, lig = tid & (tpp - 1), nslot = THREADS >> sh;
    const double tol = (double)k * 2.220446049250313e-16;
    const double tol2 = tol * tol;

    if (k == 64 && sigma > 0.0) {
        // 256 threads: the column halves live in registers (jacobi64_regs); more threads: G stays in LDS (jacobi64)
        if constexpr (USE_LDS && (THREADS == 256 || THREADS == 512)) nsweeps = jacobi64_regs<THREADS / 64>(G, ld, tid, tol2, wgt + 2 * k);
        else nsweeps = jacobi64<THREADS, THREADS / 32>(G, ld, tid, tol2);
    } else if (k > 1 && sigma > 0.0) {
        for (int sweep = 0; sweep < PSD_MAX_SWEEPS; ++sweep) {
            int rotated = 0, big = 0;
            nsweeps = sweep + 1;
            for (int step = 0; step < K - 1; ++step) {
                for (int pr = slot; pr < npair; pr += nslot) {
                    int p, q;
                    if (pr == 0) { p = step; q = K - 1; }           // step < K-1 ; the sums below are < 2 (K-1)
                    else { p = step + pr; if (p >= K - 1) p -= K - 1; q = step + (K - 1) - pr; if (q >= K - 1) q -= K - 1; }
                    if (p >= k || q >= k) continue;       // bye (uniform within the lane group)
                    double* gp = G + (size_t)p * ld;
                    double* gq = G + (size_t)q * ld;
                    double a = 0.0, b = 0.0, g = 0.0;
                    for (int i = lig; i < k; i += tpp) {
                        const double u = gp[i], v = gq[i];
                        a += u * u; b += v * v; g += u * v;
                    }
                    if (tpp == 8) { a = group8_sum(a); b = group8_sum(b); g = group8_sum(g); }
                    else if (tpp == 16) { a = group16_sum(a); b = group16_sum(b); g = group16_sum(g); }
                    else if (tpp == 32) { a = group32_sum(a); b = group32_sum(b); g = group32_sum(g); }
                    else { a = group_sum(a, tpp); b = group_sum(b, tpp); g = group_sum(g, tpp); }
                    if (g * g <= tol2 * (a * b)) continue;
                    rotated = 1;

                    // rotation that makes the two columns orthogonal: tan(2 theta) = 2g / (b - a), |theta| <= pi/4
                    //   h = sqrt(d^2 + 4 g^2), cos^2 = (1 + |d|/h)/2, sin = sign(d) g / (h cos)
                    const double d = b - a;
                    const double rh = fast_rsqrt(d * d + 4.0 * g * g);
                    const double c2 = 0.5 + 0.5 * fabs(d) * rh;
                    const double rc = fast_rsqrt(c2);
                    const double cs = c2 * rc;
                    const double sn = copysign(g * rh * rc, d * g);
                    if (g * g > JACOBI_SMALL2 * (a * b) || sn * sn > JACOBI_SMALL2) big = 1;
                    for (int i = lig; i < k; i += tpp) {
                        const double u = gp[i], v = gq[i];
                        gp[i] = cs * u - sn * v;
                        gq[i] = sn * u + cs * v;
                    }
                }
                __syncthreads();
            }
            if (!__syncthreads_or(rotated)) break;
            if (!__syncthreads_or(big)) break;
        }
    }

    if (stats && tid == 0) stats[blockIdx.x] = nsweeps;
    if (phase_limit == 3) return;

    // ---- weights: kept columns have ||g_j|| > sigma ; P = sum_j wgt_j g_j g_j', wgt_j = (||g_j|| - sigma) / ||g_j||^2
    double* inv = wgt + k;                      // 1/||g_j|| (WARM: to store the new basis)
    for (int j = tid; j < k; j += THREADS) {
        double s = 0.0;
        const double* gj = G + (size_t)j * ld;
        for (int i = 0; i < k; ++i) s += gj[i] * gj[i];
        const double nr = sqrt(s);
        wgt[j] = nr > sigma ? (nr - sigma) / s : 0.0;
        if constexpr (WARM) inv[j] = nr > 0.0 ? 1.0 / nr : 0.0;
    }
    __syncthreads();
    if constexpr (WARM) {
        // new basis: v_j = g_j / ||g_j|| (identity column if the matrix was all zero)
        double* __restrict__ Vn = vout + (size_t)blockIdx.x * vstride;
        for (int e = tid; e < k * k; e += THREADS) {
            const int i = e % k, j = e / k;
            const double s = inv[j];
            Vn[e] = (sigma > 0.0 && s > 0.0) ? G[i + (size_t)j * ld] * s : (i == j ? 1.0 : 0.0);
        }
    }

    if (phase_limit == 4) return;

    // ---- rebuild the lower triangle, repack, unscale the diagonal; dual: y = x + P(-x)
    if (USE_LDS && k == 64 && THREADS <= 1024) {
        // P = (G diag(wgt)) G' on the matrix cores: the 10 tiles on and below the diagonal, round-robin over the wavefronts;
        // the tiles replace G in LDS (after a barrier), the pack loop below then reads P[i][j]
        constexpr int NW = THREADS / 64;
        constexpr int NT = (10 + NW - 1) / NW;
        const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lr = lane & 15, lk = lane >> 4;
        v4d acc[NT];
        double wk[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wk[kk] = wgt[kk * 4 + lk];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int tt = w + NW * q;                      // lower-triangle tile list: (0,0) (1,0) (2,0) (3,0) (1,1) (2,1) (3,1) (2,2) (3,2) (3,3)
            acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
            if (tt < 10) {
                const int jb = tt < 4 ? 0 : (tt < 7 ? 1 : (tt < 9 ? 2 : 3));
                const int ib = tt < 4 ? tt : (tt < 7 ? tt - 3 : (tt < 9 ? tt - 5 : 3));
                const double* __restrict__ ga = G + ib * 16 + lr + (size_t)lk * ld;
                const double* __restrict__ gb = G + jb * 16 + lr + (size_t)lk * ld;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wk[kk] * ga[(size_t)kk * 4 * ld], gb[(size_t)kk * 4 * ld], acc[q], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int tt = w + NW * q;
            if (tt < 10) {
                const int jb = tt < 4 ? 0 : (tt < 7 ? 1 : (tt < 9 ? 2 : 3));
                const int ib = tt < 4 ? tt : (tt < 7 ? tt - 3 : (tt < 9 ? tt - 5 : 3));
#pragma unroll
                for (int r = 0; r < 4; ++r) G[ib * 16 + lk + 4 * r + (size_t)(jb * 16 + lr) * ld] = acc[q][r];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < len; idx += THREADS) {
            int i, j;
            idx_to_ij(idx, k, i, j);
            double s = G[i + (size_t)j * ld];
            if (i == j) s *= INV_SQRT2;
            if (dual) s = x[2 * (int64_t)idx] + s;
            y[2 * (int64_t)idx] = s;
        }
        return;
    }
    for (int idx = tid; idx < len; idx += THREADS) {
        int i, j;
        idx_to_ij(idx, k, i, j);
        double s = 0.0;
        for (int t = 0; t < k; ++t) s += wgt[t] * G[i + (size_t)t * ld] * G[j + (size_t)t * ld];
        if (i == j) s *= INV_SQRT2;
        if (dual) s = x[2 * (int64_t)idx] + s;
        y[2 * (int64_t)idx] = s;
    }
}

template <bool USE_LDS, bool WARM, int THREADS>
__global__ __launch_bounds__(THREADS) void psd_kernel(d2* __restrict__ out, const d2* __restrict__ in,
                                                          const ConeDesc* __restrict__ cones,
                                                          double* __restrict__ gscratch, size_t scratch_stride,
                                                          const double* __restrict__ vin, double* __restrict__ vout,
                                                          size_t vstride, int have_prev, int* __restrict__ stats, int phase_limit, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    extern __shared__ __attribute__((aligned(16))) double smem[];
    psd_block<USE_LDS, WARM, THREADS>(out, in, cones, gscratch, scratch_stride, vin, vout, vstride, have_prev, stats, phase_limit, smem);
}

// ------------------------------------------------------------------------------------------------------------------------
// Order 64, ONE WAVEFRONT per matrix, the Jacobi iteration in REGISTERS.
//
// The workgroup kernel above spends its sweeps on LDS traffic (every step re-reads and re-writes the 32 KB array) and on a
// barrier per step; its rotation angles and dot-product reductions are replicated over the 8 lanes of a pair.  Here lane
// (s, h) = (lane & 31, lane >> 5) owns rows 32 h .. 32 h + 31 of the two columns of seat pair s (64 doubles in registers), the
// three dot products of a pair are one add across the two halves (v_permlane32_swap), and the pairs follow the ODD-EVEN
// TRANSPOSITION ordering: 64 seats in a line, even steps rotate seats (2s, 2s+1) -- both columns in the same lane, no data
// movement at all -- odd steps rotate seats (2s+1, 2s+2): lane s fetches the partner column from lane s+1 and sends its own
// rotated column back (DPP wave shifts, 64 + 64 dword moves); after every rotation the two columns swap seats, so that in 64
// steps every pair of columns has met exactly once.  No LDS and no barrier inside a sweep; 1024 matrices are one wavefront
// on each of the 1024 SIMDs.  LDS (one 64 x 66 array per wavefront) only stages the packed input, the two matrix-core
// products (G0 = M' V_prev, P = G diag(w) G') and the packed output.  Same rotation formulas, thresholds, shift and weights as
// the workgroup kernel; the basis buffers are interchangeable.
constexpr int P64_LD = 66;
constexpr int P64_LEN = 64 * 65 / 2;
// (row, column) of every entry of the packed lower triangle of an order-64 matrix: i | j << 8
struct Psd64Index {
    unsigned short ij[P64_LEN + 32];
    constexpr Psd64Index() : ij() {
        int idx = 0;
        for (int j = 0; j < 64; ++j)
            for (int i = j; i < 64; ++i) ij[idx++] = (unsigned short)(i | (j << 8));
        for (; idx < P64_LEN + 32; ++idx) ij[idx] = 0;
    }
};
__device__ const Psd64Index psd64_index = Psd64Index();
__host__ inline size_t psd64w_lds_bytes() { return (size_t)(64 * P64_LD + 64 + 64 + 8) * sizeof(double); }

template <int CTRL>
__device__ __forceinline__ double dpp_shift_f64(double old, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_WAVE_SHL1 = 0x130;   // lane i reads lane i + 1 (lane 63 keeps `old`)
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i reads lane i - 1 (lane 0 keeps `old`)

// v[lane] + v[lane ^ 32] in every lane, the same bits in both
__device__ __forceinline__ double half_sum(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// the rotation that makes two columns with squared norms a, b and product g orthogonal (same formulas as jacobi64)
__device__ __forceinline__ void jacobi_angle(double a, double b, double g, double gg, double ab, double& cs, double& sn, int& big) {
    const double d = b - a;
    const double rh = fast_rsqrt(d * d + 4.0 * gg);
    const double c2 = 0.5 + 0.5 * fabs(d) * rh;
    const double rc = fast_rsqrt(c2);
    cs = c2 * rc;
    sn = copysign(g * rh * rc, d * g);
    if (gg > JACOBI_SMALL2 * ab || sn * sn > JACOBI_SMALL2) big = 1;
}

// The three step functions below are written for R rows per lane and a reducer `Red` whose sum(v) returns the total of v over
// ALL lanes that share a column, the same bits in each: the wave kernel (R = 32) needs only the add across the two halves of
// the wavefront; the four-wavefront form (R = 8, jacobi64_regs) adds an exchange through LDS.
struct RedHalf {
    __device__ __forceinline__ double sum(double v) { return half_sum(v); }
    __device__ __forceinline__ int any(int f) { return __ballot(f) != 0; }
};
// four wavefronts per matrix: lane (s, h), h = 2 w + (lane >> 5), owns rows 8h .. 8h+7; partial sums go through a double-buffered
// LDS array [2][4][32] with ONE workgroup barrier per reduction (a wavefront can reach its next write of a buffer only after the
// barrier in between, which every wavefront passes after its reads of that buffer)
template <int NW>
struct RedWaves {
    double* buf; int w, s; int phase;
    __device__ __forceinline__ double sum(double v) {
        v = half_sum(v);
        double* b = buf + (phase & 1) * (NW * 32);
        b[w * 32 + s] = v;                       // (both halves of the wavefront hold the same value)
        __syncthreads();
        double t;
        if constexpr (NW == 4) t = (b[s] + b[32 + s]) + (b[64 + s] + b[96 + s]);
        else t = ((b[s] + b[32 + s]) + (b[64 + s] + b[96 + s])) + ((b[128 + s] + b[160 + s]) + (b[192 + s] + b[224 + s]));
        ++phase;
        return t;
    }
    __device__ __forceinline__ int any(int f) { return __syncthreads_or(f); }
};

// g = U . V over the 64 rows (the squared norms a, b are NOT recomputed per step: they travel with the columns and are
// updated by the rotation -- a' = a - t g, b' = b + t g, t = tan(theta) -- and refreshed from the columns at every sweep start)
template <int R, class Red>
__device__ __forceinline__ double dotR(const double (&U)[R], const double (&V)[R], Red& red) {
    double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
#pragma unroll
    for (int r = 0; r < R; r += 4) {
        g0 += U[r] * V[r]; g1 += U[r + 1] * V[r + 1]; g2 += U[r + 2] * V[r + 2]; g3 += U[r + 3] * V[r + 3];
    }
    return red.sum((g0 + g1) + (g2 + g3));
}

// even step: the two columns of the lane's own seat pair
template <int R, class Red>
__device__ __forceinline__ void jstep_even(double (&U)[R], double (&V)[R], double& a, double& b, double tol2, int& rotated, int& big, Red& red) {
    const double g = dotR<R>(U, V, red);
    const double gg = g * g, ab = a * b;
    const bool need = gg > tol2 * ab;
    if (__ballot(need) == 0) return;
    double cs = 1.0, sn = 0.0;
    if (need) {
        rotated = 1;
        jacobi_angle(a, b, g, gg, ab, cs, sn, big);
        const double tg = sn / cs * g;
        a = fmax(a - tg, 0.0); b = b + tg;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double u = U[r], v = V[r];
        U[r] = cs * u - sn * v;
        V[r] = sn * u + cs * v;
    }
}

// odd step: the lane's column U (seat 2s+1, squared norm a) meets column W of lane s+1 (seat 2s+2, squared norm bw); afterwards
// the rotated partner takes the lane's seat and the rotated own column moves up into W of lane s+1.  Seat pair 31 has no upper
// neighbour: its U stays, and what it sends "up" -- into lane (0, h=1), which has no lower neighbour -- is that lane's own W,
// which it has just fetched.
template <int R, class Red>
__device__ __forceinline__ void jstep_odd(double (&U)[R], double (&W)[R], double& a, double& bw, bool last, double tol2, int& rotated, int& big, Red& red) {
    double X[R];
#pragma unroll
    for (int r = 0; r < R; ++r) X[r] = dpp_shift_f64<DPP_WAVE_SHL1>(0.0, W[r]);
    double b = dpp_shift_f64<DPP_WAVE_SHL1>(0.0, bw);
    const double g = dotR<R>(U, X, red);
    const double gg = g * g, ab = a * b;
    const bool need = !last && gg > tol2 * ab;
    double cs = 1.0, sn = 0.0, an = a;
    if (need) {
        rotated = 1;
        jacobi_angle(a, b, g, gg, ab, cs, sn, big);
        const double tg = sn / cs * g;
        an = fmax(a - tg, 0.0); b = b + tg;
    }
    // new own seat = c1 U + c2 X ; sent up = c3 U + c4 X      (rotation + seat swap; identity for the last seat pair)
    const double c1 = last ? 1.0 : sn, c2 = last ? 0.0 : cs, c3 = last ? 0.0 : cs, c4 = last ? 1.0 : -sn;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double u = U[r], x = X[r];
        const double up = c3 * u + c4 * x;
        U[r] = c1 * u + c2 * x;
        W[r] = dpp_shift_f64<DPP_WAVE_SHR1>(W[r], up);
    }
    // the squared norms travel with their columns
    a = last ? a : b;
    bw = dpp_shift_f64<DPP_WAVE_SHR1>(bw, last ? b : an);
}

// the sweeps of an order-64 matrix with the column halves (R rows per lane) in registers: odd-even ordering, 64 steps per sweep
template <int R, class Red>
__device__ __forceinline__ int jacobi64_sweeps(double (&A)[R], double (&B)[R], bool last, double tol2, Red& red) {
    int nsweeps = 0;
    for (int sweep = 0; sweep < PSD_MAX_SWEEPS; ++sweep) {
        int rotated = 0, big = 0;
        nsweeps = sweep + 1;
        double na = dotR<R>(A, A, red), nb = dotR<R>(B, B, red);
        for (int q = 0; q < 16; ++q) {
            jstep_even<R>(A, B, na, nb, tol2, rotated, big, red);          // seats (2s, 2s+1) = (A, B); afterwards seat 2s+1 holds A
            jstep_odd<R>(A, B, na, nb, last, tol2, rotated, big, red);     // seat 2s+1 (A) with seat 2s+2 (B of lane s+1)
            jstep_even<R>(A, B, na, nb, tol2, rotated, big, red);          // afterwards seat 2s+1 holds B
            jstep_odd<R>(B, A, nb, na, last, tol2, rotated, big, red);     // seat 2s+1 (B) with seat 2s+2 (A of lane s+1)
        }
        if (!red.any(rotated)) break;
        if (!red.any(big)) break;
    }
    return nsweeps;
}

// jacobi64 for a workgroup of FOUR wavefronts with the columns in registers (a batch too small to give every SIMD a matrix of
// its own -- a shard of a multi-GPU run): G (LDS, column-major) is read once, rotated in registers, written back.  `scratch`:
// 256 doubles of LDS for the reducer.
// NW = 8 (512 threads, 4 rows per lane): for batches so small that four wavefronts per matrix would leave half of the SIMDs idle
// (128 matrices on 256 CUs: a 1/8 shard of the 512-block SDP).
template <int NW>
__device__ __forceinline__ int jacobi64_regs(double* __restrict__ G, const int ld, const int tid, const double tol2, double* scratch) {
    constexpr int R = 32 / NW;
    const int w = tid >> 6, ln = tid & 63, s = ln & 31, h = 2 * w + (ln >> 5);
    double A[R], B[R];
    double* ga = G + R * h + (size_t)(2 * s) * ld;
#pragma unroll
    for (int r = 0; r < R; ++r) { A[r] = ga[r]; B[r] = ga[r + ld]; }
    RedWaves<NW> red{scratch, w, s, 0};
    const int nsweeps = jacobi64_sweeps<R>(A, B, s == 31, tol2, red);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) { ga[r] = A[r]; ga[r + ld] = B[r]; }
    __syncthreads();
    return nsweeps;
}

template <bool WARM>
__global__ __launch_bounds__(64) void psd64_wave_kernel(d2* __restrict__ out, const d2* __restrict__ in, const ConeDesc* __restrict__ cones,
                                                        const double* __restrict__ vin, double* __restrict__ vout, size_t vstride,
                                                        int have_prev, int* __restrict__ stats, int phase_limit, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int LD = P64_LD;
    const int lane = threadIdx.x;
    const int cone = blockIdx.x >> 1, part = blockIdx.x & 1;
    const ConeDesc cd = cones[cone];
    const bool dual = (cd.dual_part == part);
    const double sgn = dual ? -1.0 : 1.0;
    const double* __restrict__ x = reinterpret_cast<const double*>(in + cd.start) + part;
    double* __restrict__ y = reinterpret_cast<double*>(out + cd.start) + part;
    double* G = smem;                            // [64][LD], column-major
    double* wgt = G + 64 * LD;                   // [64]
    double* inv = wgt + 64;                      // [64]

    // ---- load: M = smat(sgn x), diagonal scaled by sqrt(2): 33 coalesced loads per lane, all in flight together
    double fro = 0.0;
    {
        double v[33];
        unsigned short ij[33];
#pragma unroll
        for (int q = 0; q < 33; ++q) {
            const int idx = q * 64 + lane;
            ij[q] = psd64_index.ij[idx];
            v[q] = idx < P64_LEN ? x[2 * (int64_t)idx] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 33; ++q) {
            const int i = ij[q] & 0xFF, j = ij[q] >> 8;
            double w = sgn * v[q];
            if (i == j) { w *= SQRT2; fro += w * w; }
            else fro += 2.0 * w * w;
            if (q * 64 + lane < P64_LEN) {
                G[i + j * LD] = w;
                G[j + i * LD] = w;
            }
        }
    }
    fro = wave_sum(fro);
    const double sigma = (WARM ? 1.001 : 0.505) * sqrt(fro);
    __syncthreads();
    G[lane + lane * LD] += sigma;
    __syncthreads();
    if (phase_limit == 1) return;

    const int lr = lane & 15, lk = lane >> 4;
    if constexpr (WARM) {
        if (have_prev && sigma > 0.0) {
            // G0 = M' V_prev : 16 tiles of 16 k-steps on the matrix cores, operands as in psd_kernel
            const double* __restrict__ Vp = vin + (size_t)blockIdx.x * vstride;
            double bv[4][16];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) bv[jb][kk] = Vp[(size_t)(jb * 16 + lr) * 64 + kk * 4 + lk];
            v4d acc[4][4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                double ga[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) ga[kk] = G[ib * 16 + lr + (kk * 4 + lk) * LD];
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    acc[ib][jb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) acc[ib][jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[kk], bv[jb][kk], acc[ib][jb], 0, 0, 0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) G[ib * 16 + lk + 4 * r + (jb * 16 + lr) * LD] = acc[ib][jb][r];
            __syncthreads();
        }
    }
    if (phase_limit == 2) return;

    // ---- one-sided Jacobi, odd-even ordering, in registers
    const int s = lane & 31, h = lane >> 5;
    double A[32], B[32];
    {
        const double* ga = G + 32 * h + (2 * s) * LD;
#pragma unroll
        for (int r = 0; r < 32; ++r) { A[r] = ga[r]; B[r] = ga[r + LD]; }
    }
    int nsweeps = 0;
    if (sigma > 0.0) {
        const double tol = 64.0 * 2.220446049250313e-16;
        RedHalf red;
        nsweeps = jacobi64_sweeps<32>(A, B, s == 31, tol * tol, red);
    }
    if (stats && lane == 0) stats[blockIdx.x] = nsweeps;
    if (phase_limit == 3) return;

    // ---- weights, new basis
    {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) { sa += A[r] * A[r]; sb += B[r] * B[r]; }
        sa = half_sum(sa); sb = half_sum(sb);
        __syncthreads();
        double* ga = G + 32 * h + (2 * s) * LD;
#pragma unroll
        for (int r = 0; r < 32; ++r) { ga[r] = A[r]; ga[r + LD] = B[r]; }
        if (h == 0) {
            const double na = sqrt(sa), nb = sqrt(sb);
            wgt[2 * s] = na > sigma ? (na - sigma) / sa : 0.0;
            wgt[2 * s + 1] = nb > sigma ? (nb - sigma) / sb : 0.0;
            inv[2 * s] = na > 0.0 ? 1.0 / na : 0.0;
            inv[2 * s + 1] = nb > 0.0 ? 1.0 / nb : 0.0;
        }
        __syncthreads();
    }
    if constexpr (WARM) {
        double* __restrict__ Vn = vout + (size_t)blockIdx.x * vstride;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const double sc = inv[j];
            Vn[j * 64 + lane] = (sigma > 0.0 && sc > 0.0) ? G[lane + j * LD] * sc : (lane == j ? 1.0 : 0.0);
        }
    }
    if (phase_limit == 4) return;

    // ---- P = (G diag(wgt)) G' : the 10 tiles on and below the diagonal, written over G, then packed
    {
        double fr[4][16], wk[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wk[kk] = wgt[kk * 4 + lk];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) fr[ib][kk] = G[ib * 16 + lr + (kk * 4 + lk) * LD];
        v4d acc[10];
        int t = 0;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int ib = jb; ib < 4; ++ib, ++t) {
                acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wk[kk] * fr[ib][kk], fr[jb][kk], acc[t], 0, 0, 0);
            }
        __syncthreads();
        t = 0;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int ib = jb; ib < 4; ++ib, ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) G[ib * 16 + lk + 4 * r + (jb * 16 + lr) * LD] = acc[t][r];
        __syncthreads();
    }
    {
        double xv[33];
        unsigned short ij[33];
#pragma unroll
        for (int q = 0; q < 33; ++q) {
            const int idx = q * 64 + lane;
            ij[q] = psd64_index.ij[idx];
            xv[q] = (dual && idx < P64_LEN) ? x[2 * (int64_t)idx] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 33; ++q) {
            const int idx = q * 64 + lane;
            const int i = ij[q] & 0xFF, j = ij[q] >> 8;
            double v = G[i + j * LD];
            if (i == j) v *= INV_SQRT2;
            if (idx < P64_LEN) y[2 * (int64_t)idx] = xv[q] + v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Order 64, WARM: the projection by eigenvector REFINEMENT -- matrix products only, no serial chain of Jacobi steps.
//
// The warm-started Jacobi iteration above is a chain of 64 barrier-separated steps per sweep (0.46 us each with four wavefronts
// per matrix) however few matrices there are: a shard of a multi-GPU run (at most one matrix per CU) spends 100 of its 118 us
// there.  But a basis V that nearly diagonalises M can be corrected by first-order perturbation theory, quadratically convergent
// and all of it 64 x 64 x 64 products on the matrix cores (the iteration of Ogita & Aishima, "Iterative refinement for symmetric
// eigenvalue decomposition", 2018, restated with ONE product for both correction terms):
//     G = M V;   d_j = (v_j . g_j) / (v_j . v_j);   N = V' (G - V diag d)      [ N_ij = (V'MV)_ij - (V'V)_ij d_j , N_jj = 0 ]
//     E_ij = N_ij / (d_j - d_i)  (i != j),   E_jj = (1 - v_j . v_j) / 2;       V <- V + V E
// (three products per iteration; the second term of N_ij is what pulls V back to orthonormal columns), until ||E||_F <= 1e-6; then
// ONE Newton-Schulz step V <- V (I + (I - V'V) / 2) -- the part of the orthogonality defect of a pair with a tiny gap that is below
// the rounding level of N (defect x gap < eps ||M||) is invisible to the iteration, and a defect of 1e-10 would be an error of 1e-10
// in the projection -- and P = V max(diag d, 0) V'.  3 it + 3 products.
// START: the solver's iterates move smoothly, so the basis is EXTRAPOLATED from the last two, V0 = 2 V_prev - V_pp (the two
// ping-pong buffers): in C4's steady state the first ||E||_F drops from 2e-2 to 1e-3 (outer iteration 200) ... 1e-4 (350), and
// two or three iterations do where the plain start needs four.  If the extrapolated start does not converge (a jump of the
// iterate), the kernel starts again from V_prev.
// Pairs whose coupling is NOT small against their gap (|N_ij| > RF_THETA |d_j - d_i|: eigenvalues that nearly coincide or have
// just crossed) are turned first by an exact two-sided Jacobi rotation of S = N + diag d and of the two columns of V -- a rare,
// slow path through LDS; couplings below the rounding level of the products (RF_NOISE ||M||_F) are left alone whatever the gap
// (a cluster is an invariant subspace: any basis of it gives the same projection).  A matrix that needs more than RF_MAX_ROT
// rotations or RF_MAX_IT iterations (a cold basis, a jump of the iterate, a cluster split by the change) is FLAGGED (record 1) and
// left to the Jacobi kernel, which the launcher runs behind this one for the flagged matrices only, from a cold start.  Record 2:
// accepted, but columns were rotated -- the next call must not extrapolate across that.
// One workgroup of four wavefronts per matrix; wavefront w owns the 16-column block w of every product (operand and result
// layouts of v_mfma_f64_16x16x4_f64 chain without data movement: a result tile IS the B operand of the next product); M and V
// live in LDS (leading dimension 66; the rotation path borrows M's array and unpacks M again afterwards).
constexpr int RF_LD = 66;
constexpr int RF_MAX_IT = 7;
constexpr int RF_MAX_ROT = 12;
constexpr double RF_THETA = 0.2;
constexpr double RF_ACCEPT2 = 1e-12;          // ||E||_F <= 1e-6 at the last update: error of the projection < 1e-13 ||M|| (measured on the oracle)
constexpr double RF_NOISE = 16.0 * 2.220446049250313e-16;
__host__ inline size_t psd64r_lds_bytes(int wps) {
    // one workgroup per CU: a third array for the rotation path (S = N + diag d); two per CU: that path borrows M's array and unpacks M again
    const size_t own = (size_t)((wps == 1 ? 3 : 2) * 64 * RF_LD + 64 + 64) * sizeof(double);
    return own > psd_lds_bytes(64) ? own : psd_lds_bytes(64);      // (the Jacobi path of a flagged matrix runs in the same workgroup)
}
// per-matrix record between calls: int32 code [nmat] (0 accepted, 1 left to Jacobi), then uint64 mask [nmat] of the columns that were
// rotated (the next call does not extrapolate those)
__host__ __device__ inline size_t psd64r_record_ints(int nmat) { return (size_t)nmat * 4; }

__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}
// sum over the four lanes l, l^16, l^32, l^48 (the four row groups of a tile column): the same bits in each
__device__ __forceinline__ double colgroup_sum(double v) {
    v = swap_sum<16>(v);
    v = swap_sum<32>(v);
    return v;
}
// the packed input of a thread (9 entries) -> M in LDS (both triangles, diagonal scaled by sqrt(2)); returns the thread's share of ||M||_F^2
// Fusion of the GAP / DR step around the projection (PsdFuse::on): the kernel's input is not read from a vector t1 but FORMED,
// t1 = a1 sol + (1 - a1) x (the relaxation behind S1, gap.jl:48), and its output is not written to t2 but taken on to the step's last pass,
// x = alpha (alpha2 t2 + (1 - alpha2) t1) + (1 - alpha) x (gap.jl:58,78), with the vector the next CG start applies M to (sol - [0; x2]) --
// for the entries of the PSD cones; the other entries (elementwise cones) are done by the extra workgroups of the same launch.
__device__ __forceinline__ void rf_fetch_m(double (&v)[9], const double* __restrict__ x, int tid, const PsdFuse& fz, const double* __restrict__ solc,
                                           const double* __restrict__ xc) {
    const double b1 = 1.0 - fz.a1;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int idx = q * 256 + tid;
        if (fz.on) v[q] = idx < P64_LEN ? fz.a1 * solc[2 * (int64_t)idx] + b1 * xc[2 * (int64_t)idx] : 0.0;
        else v[q] = idx < P64_LEN ? x[2 * (int64_t)idx] : 0.0;
    }
}
__device__ __forceinline__ double ew_apply_psd(int op, double v) {       // (= ew_apply of vecops.hip)
    switch (op) {
        case EW_COPY: return v;
        case EW_ZERO: return 0.0;
        case EW_MAX0: return v < 0.0 ? 0.0 : v;
        default:      return v > 0.0 ? 0.0 : v;
    }
}
__device__ __forceinline__ double rf_store_m(double* __restrict__ Ml, const double (&v)[9], double sgn, int tid) {
    double fro = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int idx = q * 256 + tid;
        const unsigned ij = psd64_index.ij[idx < P64_LEN ? idx : 0];
        const int i = ij & 0xFF, j = ij >> 8;
        double m = sgn * v[q];
        if (idx < P64_LEN) {
            if (i == j) { m *= SQRT2; fro += m * m; }
            else fro += 2.0 * m * m;
            Ml[i + j * RF_LD] = m;
            Ml[j + i * RF_LD] = m;
        }
    }
    return fro;
}
// acc[ib] += sum_kk A(ib, kk) B(kk) for the four row tiles ib of a column block: A(ib, kk) = base[ib SI + kk SK] (LDS), B(kk) = b[kk >> 2][kk & 3].
// The A fragments of four k-steps are requested while the matrix cores work on the previous four (the compiler, left alone, requests
// each step's fragments behind the previous step's last MFMA and exposes the LDS latency sixteen times per product).
template <int SI, int SK, int KC = 4>
__device__ __forceinline__ void rf_gemm(v4d (&acc)[4], const double* __restrict__ base, const v4d (&b)[4]) {
    // KC k-steps (4 KC fragments) per request group; two groups alternate (KC = 4: 64 registers of fragments, KC = 2: 32 -- the form for two
    // workgroups per CU, which has 256 registers per lane)
    constexpr int NQ = 4 * KC, NC = 16 / KC;
    double a0[NQ], a1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) a0[q] = base[(q & 3) * SI + (q >> 2) * SK];
#pragma unroll
    for (int c = 0; c < NC; c += 2) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) a1[q] = base[(q & 3) * SI + (KC * (c + 1) + (q >> 2)) * SK];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int kk = KC * c + (q >> 2); acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b[kk >> 2][kk & 3], acc[q & 3], 0, 0, 0); }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NC) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) a0[q] = base[(q & 3) * SI + (KC * (c + 2) + (q >> 2)) * SK];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int kk = KC * (c + 1) + (q >> 2); acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b[kk >> 2][kk & 3], acc[q & 3], 0, 0, 0); }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// In-kernel time stamps of the refinement kernel (timing experiments; -DFOS_PSD_STAMPS; tools/psd_stamps.py): workgroup FOS_PSD_STAMP_WG (default 0),
// lane 0 of every wavefront, consecutive slots: (ticks of the 100 MHz clock) * 64 + phase id
#ifdef FOS_PSD_STAMPS
#ifndef FOS_PSD_STAMP_WG
#define FOS_PSD_STAMP_WG 0
#endif
__device__ long long g_psd_stamps[4 * 128];
#define PSD_STAMP(id) do { if (stamp_on && stamp_slot < 128) { __builtin_amdgcn_sched_barrier(0); g_psd_stamps[w * 128 + stamp_slot++] = wall_clock64() * 64 + (id); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PSD_STAMP(id) do { } while (0)
#endif

template <int WPS>
__global__ __launch_bounds__(256, WPS) void psd64_refine_kernel(d2* __restrict__ out, const d2* __restrict__ in, const ConeDesc* __restrict__ cones,
                                                           const double* __restrict__ vin, double* __restrict__ vout, int have_prev,
                                                           int* __restrict__ stats, int32_t* __restrict__ rec, int phase_limit,
                                                           const int32_t* __restrict__ gate, const double theta, const PsdFuse fz) {
    if (gate && !*gate) return;
    if (fz.on && (int)blockIdx.x >= fz.nmat) {
        // the step's two passes on the entries of the elementwise cones (relax_ew_kernel + gap_final_kernel of vecops.hip, same arithmetic)
        const double b1 = 1.0 - fz.a1, b2 = 1.0 - fz.alpha2, b = 1.0 - fz.alpha;
        for (int64_t i = ((int64_t)blockIdx.x - fz.nmat) * 256 + threadIdx.x; i < fz.l; i += (int64_t)(gridDim.x - fz.nmat) * 256) {
            const uint8_t op = fz.ew_op[i];
            if (op == EW_SKIP) continue;
            const d2 si = fz.sol[i];
            d2 xi = fz.xv[i];
            const d2 v = make_double2(fz.a1 * si.x + b1 * xi.x, fz.a1 * si.y + b1 * xi.y);
            const d2 u = make_double2(ew_apply_psd(op & 3, v.x), ew_apply_psd((op >> 2) & 3, v.y));
            const double rx = fz.alpha2 * u.x + b2 * v.x, ry = fz.alpha2 * u.y + b2 * v.y;
            xi.x = fz.alpha * rx + b * xi.x;
            xi.y = fz.alpha * ry + b * xi.y;
            fz.xv[i] = xi;
            if (fz.shift) fz.shift[i] = make_double2(si.x, si.y - xi.y);
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int LD = RF_LD;
    double* Ml = smem;                      // M, column-major (symmetric); S = N + diag d in the rotation path; P at the end
    double* Vl = Ml + 64 * LD;              // V, column-major: V[k][j] at k + LD j
    double* dl = Vl + 64 * LD;              // [64] eigenvalue estimates
    double* red = dl + 64;                  // [64] reductions / broadcast
    int* ired = reinterpret_cast<int*>(red + 32);
    double* Sl = WPS == 1 ? red + 64 : Ml;  // the rotation path's S = N + diag d: an array of its own, or M's (M is unpacked again afterwards)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int jcol = 16 * w + lr;
#ifdef FOS_PSD_STAMPS
    const bool stamp_on = blockIdx.x == FOS_PSD_STAMP_WG && lane == 0;
    int stamp_slot = 0;
#endif
    PSD_STAMP(0);
    const int cone = blockIdx.x >> 1, part = blockIdx.x & 1;
    const ConeDesc cd = cones[cone];
    const bool dual = (cd.dual_part == part);
    const double sgn = dual ? -1.0 : 1.0;
    const double* __restrict__ x = reinterpret_cast<const double*>(in + cd.start) + part;
    double* __restrict__ y = reinterpret_cast<double*>(out + cd.start) + part;
    const d2* __restrict__ Vp = reinterpret_cast<const d2*>(vin + (size_t)blockIdx.x * 4096);
    d2* __restrict__ Vn = reinterpret_cast<d2*>(vout + (size_t)blockIdx.x * 4096);
    const int nmat = fz.on ? fz.nmat : (int)gridDim.x;
    int32_t* __restrict__ code = rec + blockIdx.x;
    unsigned long long* __restrict__ cmask = reinterpret_cast<unsigned long long*>(rec + 2 * nmat) + blockIdx.x;
    const double* __restrict__ solc = fz.on ? reinterpret_cast<const double*>(fz.sol + cd.start) + part : nullptr;      // this copy's component of sol, x
    double* __restrict__ xcomp = fz.on ? reinterpret_cast<double*>(fz.xv + cd.start) + part : nullptr;
    // extrapolation needs the bases of the last TWO projections, the second a continuation of the first (accepted by this kernel);
    // columns that were rotated then are not extrapolated
    int attempt = (have_prev >= 2 && *code == 0) ? 0 : 1;
    const unsigned long long skip = attempt == 0 ? *cmask : 0ull;

    // ---- everything the start needs is requested at once: the packed matrix and the two bases
    double fro;
    {
        double xv[9];
        rf_fetch_m(xv, x, tid, fz, solc, xcomp);
        d2 vv[8], vo[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) vv[q] = Vp[tid + 256 * q];
        if (attempt == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) vo[q] = Vn[tid + 256 * q];
        }
        fro = rf_store_m(Ml, xv, sgn, tid);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * (tid + 256 * q);           // rows k, k + 1 of column j = e >> 6
            if (attempt == 0 && !((skip >> (e >> 6)) & 1ull)) { vv[q].x = 2.0 * vv[q].x - vo[q].x; vv[q].y = 2.0 * vv[q].y - vo[q].y; }
            *reinterpret_cast<d2*>(Vl + (e & 63) + (e >> 6) * LD) = vv[q];
        }
    }
    PSD_STAMP(1);
    fro = wave_sum(fro);
    if (lane == 0) red[w] = fro;
    __syncthreads();
    PSD_STAMP(2);
    const double scale = sqrt((red[0] + red[1]) + (red[2] + red[3]));
    const double noise = RF_NOISE * scale;
    if (phase_limit == 11) return;

    v4d Vb[4];                              // this wavefront's column block: Vb[ib][r] = V[16 ib + 4 r + lk][16 w + lr] (B operand of k-step 4 ib + r)
    int it = 0, nrot = 0, fail = 0, total_it = 0;
    int dbg_c1 = 0, dbg_c2 = 0;
    unsigned long long rmask = 0ull;
    double dj = 0.0;
    for (;; ++attempt) {
        if (attempt == 1 && total_it > 0) {
            // the extrapolated start did not converge: again from the previous basis
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = 2 * (tid + 256 * q);
                *reinterpret_cast<d2*>(Vl + (e & 63) + (e >> 6) * LD) = Vp[tid + 256 * q];
            }
            __syncthreads();
        }
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) Vb[ib][r] = Vl[16 * ib + 4 * r + lk + jcol * LD];
        fail = 0; nrot = 0; rmask = 0ull;
        for (it = 0;; ++it) {
            if (it >= RF_MAX_IT) { fail = 1; break; }
            ++total_it;
            PSD_STAMP(3);
            // ---- G = M V (column block w)
            v4d acc[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[ib] = v4d{0.0, 0.0, 0.0, 0.0};
            rf_gemm<16, 4 * LD, WPS == 2 ? 2 : 4>(acc, Ml + lr + lk * LD, Vb);
            PSD_STAMP(4);
            // ---- Rayleigh quotients of the block's columns; G' = G - V diag d
            double vv = 0.0, vg = 0.0;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) { vv += Vb[ib][r] * Vb[ib][r]; vg += Vb[ib][r] * acc[ib][r]; }
            vv = colgroup_sum(vv); vg = colgroup_sum(vg);
            dj = vv > 0.0 ? vg / vv : 0.0;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ib][r] -= Vb[ib][r] * dj;
            if (lk == 0) dl[jcol] = dj;
            PSD_STAMP(5);
            // ONE barrier in front of the second product: d of all blocks is visible, and so is the V the previous iteration stored behind ITS last
            // barrier (the first product reads M and this wavefront's registers only, so that store needed no barrier of its own)
            __syncthreads();
            PSD_STAMP(6);
            // d_i of this lane's sixteen rows: requested here, in one go, so that their LDS latency passes beside the product (read one by one in
            // front of their use they cost sixteen exposed round trips: 1.3 us of a 10 us iteration, 5 us beside a second workgroup's products)
            // (two workgroups per CU: 256 registers per lane -- sixteen more live doubles spill; there the other workgroup's products cover the reads, so they stay in E)
            double di[WPS == 1 ? 16 : 1];
            if constexpr (WPS == 1) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) di[4 * ib + r] = dl[16 * ib + 4 * r + lk];
            }
            // ---- N = V' G' (column block w): A operand = columns of V read as rows
            v4d nac[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) nac[ib] = v4d{0.0, 0.0, 0.0, 0.0};
            rf_gemm<16 * LD, 4, WPS == 2 ? 2 : 4>(nac, Vl + lk + lr * LD, acc);
            PSD_STAMP(7);
            // ---- E from N (straight-line code: selects, no branch per entry); pairs that need a rotation first are flagged
            double conv2 = 0.0;
            int bad = 0;
            v4d ev[4];
            auto make_e = [&]() {
                conv2 = 0.0; bad = 0;
                const double ediag = 0.5 * (1.0 - vv);
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ib + 4 * r + lk;
                        const bool isdiag = i == jcol;
                        const double n = isdiag ? 0.0 : nac[ib][r];
                        const double den = dj - (WPS == 1 ? di[4 * ib + r] : dl[i]);
                        const double an = fabs(n), ad = fabs(den);
                        // rounding level (an <= noise): never chased, whatever the gap -- a quotient of two rounding errors would be an
                        // O(theta) "rotation" that is not even skew; small against the gap: first-order correction; else: a rotation first
                        const bool small = an <= noise, ok = an <= theta * ad;
                        double rc = __builtin_amdgcn_rcp(den);
                        rc = rc * (2.0 - den * rc);                  // (E needs a few digits only: the iteration corrects itself)
                        double e = (small || !ok) ? 0.0 : n * rc;
                        bad |= (!small && !ok) ? 1 : 0;
                        e = isdiag ? ediag : e;
                        conv2 += e * e;
                        ev[ib][r] = e;
                    }
            };
            v4d vn[4];
            double c2 = 0.0;
            for (int pass = 0;; ++pass) {
                make_e();
                PSD_STAMP(8);
                // ---- ||E||_F^2 and the rotation flag of this wavefront
                conv2 = group16_sum(conv2);                    // (DPP inside the 16-lane rows, then two cross-row exchanges)
                conv2 = swap_sum<16>(conv2);
                conv2 = swap_sum<32>(conv2);
                const int wbad = __builtin_amdgcn_ballot_w64(bad != 0) != 0ull ? 1 : 0;
                if (lane == 0) { red[8 + w] = conv2; ired[8 + w] = wbad; }
                PSD_STAMP(10);
                // ---- V + V E (column block w), not yet stored; A operand = rows of V
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) vn[ib] = Vb[ib];
                rf_gemm<16, 4 * LD, WPS == 2 ? 2 : 4>(vn, Vl + lr + lk * LD, ev);
                PSD_STAMP(11);
                __syncthreads();                               // every wavefront has read the old V; the four sums and flags are visible
                PSD_STAMP(12);
                c2 = (red[8] + red[9]) + (red[10] + red[11]);
                const int any_bad = (ired[8] | ired[9]) | (ired[10] | ired[11]);
                if (!any_bad || pass == 1) break;      // (pass 1: a pair can still be flagged only through the difference between (N_ij + N_ji) / 2 and N_ij: left to the next iteration)
                // ---- rare: exact rotations of the offending pairs on S = N + diag d (in M's array) and on the columns of V; the product above is discarded
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ib + 4 * r + lk;
                        Sl[i + jcol * LD] = (i == jcol) ? dj : nac[ib][r];
                    }
                __syncthreads();
                for (;;) {
                    // worst pair: the largest coupling among the pairs that are neither small against their gap nor below the noise
                    double best = 0.0; int bidx = -1;
                    {
                        const int j = tid & 63, i0 = (tid >> 6) * 16;
                        const double djj = Sl[j + j * LD];
#pragma unroll 4
                        for (int q = 0; q < 16; ++q) {
                            const int i = i0 + q;
                            if (i == j) continue;
                            const double an = fabs(Sl[i + j * LD]), ad = fabs(djj - Sl[i + i * LD]);
                            if (!(an <= theta * ad) && !(an <= noise) && !(an <= best)) { best = an; bidx = i | (j << 8); }
                        }
                    }
                    for (int off = 32; off > 0; off >>= 1) {
                        const double ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bidx, off, 64);
                        if (ob > best || (ob == best && oi > bidx)) { best = ob; bidx = oi; }
                    }
                    if (lane == 0) { red[w] = best; ired[w] = bidx; }
                    __syncthreads();
                    double gb = red[0]; int gi = ired[0];
#pragma unroll
                    for (int q = 1; q < 4; ++q) if (red[q] > gb || (red[q] == gb && ired[q] > gi)) { gb = red[q]; gi = ired[q]; }
                    __syncthreads();
                    if (gi < 0) break;
                    if (++nrot > RF_MAX_ROT) { fail = 1; break; }
                    const int pi = gi & 0xFF, pj = gi >> 8;
                    rmask |= (1ull << pi) | (1ull << pj);
                    const double a = Sl[pi + pi * LD], b = Sl[pj + pj * LD], g = 0.5 * (Sl[pi + pj * LD] + Sl[pj + pi * LD]);
                    double cs = 1.0, sn = 0.0;
                    if (g != 0.0) {
                        const double zeta = (b - a) / (2.0 * g);
                        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        cs = 1.0 / sqrt(1.0 + t * t); sn = t * cs;
                    }
                    __syncthreads();
                    if (tid < 64) {                         // columns pi, pj of S
                        const double ti = Sl[tid + pi * LD], tj = Sl[tid + pj * LD];
                        Sl[tid + pi * LD] = cs * ti - sn * tj; Sl[tid + pj * LD] = sn * ti + cs * tj;
                    } else if (tid < 128) {                 // columns pi, pj of V
                        const int rr = tid - 64;
                        const double ti = Vl[rr + pi * LD], tj = Vl[rr + pj * LD];
                        Vl[rr + pi * LD] = cs * ti - sn * tj; Vl[rr + pj * LD] = sn * ti + cs * tj;
                    }
                    __syncthreads();
                    if (tid < 64) {                         // rows pi, pj of S
                        const double ti = Sl[pi + tid * LD], tj = Sl[pj + tid * LD];
                        Sl[pi + tid * LD] = cs * ti - sn * tj; Sl[pj + tid * LD] = sn * ti + cs * tj;
                    }
                    __syncthreads();
                    if (tid == 0) { Sl[pi + pj * LD] = 0.0; Sl[pj + pi * LD] = 0.0; }
                    __syncthreads();
                }
                if (!fail) {
                    // back to registers: d, N, this block's columns of V
                    dj = Sl[jcol + jcol * LD];
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * ib + 4 * r + lk;
                            nac[ib][r] = Sl[i + jcol * LD];
                            Vb[ib][r] = Vl[i + jcol * LD];
                        }
                    if (lk == 0) dl[jcol] = dj;
                }
                __syncthreads();
                if constexpr (WPS != 1) {                      // M back into its array
                    double xv[9];
                    rf_fetch_m(xv, x, tid, fz, solc, xcomp);
                    rf_store_m(Ml, xv, sgn, tid);
                    __syncthreads();
                }
                if (fail) break;
                if constexpr (WPS == 1) {
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                        for (int r = 0; r < 4; ++r) di[4 * ib + r] = dl[16 * ib + 4 * r + lk];
                }
            }
            if (fail) break;
            // ---- the new V: registers and LDS (visible to the others behind the next barrier: the one in front of the next N product, or the one behind the loop)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                Vb[ib] = vn[ib];
#pragma unroll
                for (int r = 0; r < 4; ++r) Vl[16 * ib + 4 * r + lk + jcol * LD] = vn[ib][r];
            }
            if (it < 2) {       // (debug record: the decimal exponent of ||E||_F, from the binary exponent of its square -- no log10 on this path: its fp64 polynomial cost 4 us beside a second workgroup's products)
                const int ex = (int)((__double_as_longlong(c2) >> 52) & 0x7FF) - 1023;
                const int dg = c2 > 0.0 ? min(9, max(0, (int)(-0.150515f * (float)ex))) : 9;
                if (it == 0) dbg_c1 = dg; else dbg_c2 = dg;
            }
            PSD_STAMP(13);
            if (c2 <= RF_ACCEPT2) { ++it; break; }
            if (!(c2 < 1e300)) { fail = 1; break; }            // NaN / overflow: not ours
        }
        __syncthreads();                                       // the last V is in LDS for everybody (and nobody still reads what a restart overwrites)
        if (!fail || attempt >= 1) break;
    }
    if (fail) {
        // the basis does not fit this matrix: Jacobi from a cold start, in this workgroup
        if (tid == 0) { *code = 1; *cmask = 0ull; }
        if (fz.on) {                                       // (the Jacobi code reads t1 and writes t2: form t1 first ...)
            double t1v[9];
            rf_fetch_m(t1v, x, tid, fz, solc, xcomp);
            double* __restrict__ t1c = const_cast<double*>(x);
#pragma unroll
            for (int q = 0; q < 9; ++q) { const int idx = q * 256 + tid; if (idx < P64_LEN) t1c[2 * (int64_t)idx] = t1v[q]; }
        }
        __syncthreads();
        psd_block<true, true, 256>(out, in, cones, nullptr, 0, vin, vout, 4096, 0, stats, 0, smem);
        if (fz.on) {                                       // (... and take t2 on to the step's last pass)
            __syncthreads();
            const double b2 = 1.0 - fz.alpha2, b = 1.0 - fz.alpha;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int idx = q * 256 + tid;
                if (idx < P64_LEN) {
                    const double t1 = x[2 * (int64_t)idx], t2 = y[2 * (int64_t)idx];
                    const double xn = fz.alpha * (fz.alpha2 * t2 + b2 * t1) + b * xcomp[2 * (int64_t)idx];
                    xcomp[2 * (int64_t)idx] = xn;
                    if (fz.shift) (reinterpret_cast<double*>(fz.shift + cd.start) + part)[2 * (int64_t)idx] = part == 1 ? solc[2 * (int64_t)idx] - xn : solc[2 * (int64_t)idx];
                }
            }
        }
        return;
    }
    if (tid == 0) { *code = 0; *cmask = rmask; if (stats) stats[blockIdx.x] = 100 + 1000 * (attempt == 0) + 16 * nrot + total_it + (rec[4 * nmat] == 77 ? 10000 * dbg_c1 + 100000 * dbg_c2 + 1000000 * (skip != 0ull) : 0); }
    if (phase_limit == 16) return;
    PSD_STAMP(14);

    // ---- one Newton-Schulz step: V <- V (I + (I - V'V) / 2)
    {
        v4d g[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) g[ib] = v4d{0.0, 0.0, 0.0, 0.0};
        rf_gemm<16 * LD, 4, WPS == 2 ? 2 : 4>(g, Vl + lk + lr * LD, Vb);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) g[ib][r] = ((16 * ib + 4 * r + lk == jcol) ? 0.5 : 0.0) - 0.5 * g[ib][r];
        PSD_STAMP(15);
        v4d vn[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) vn[ib] = Vb[ib];
        rf_gemm<16, 4 * LD, WPS == 2 ? 2 : 4>(vn, Vl + lr + lk * LD, g);
        __syncthreads();
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) Vl[16 * ib + 4 * r + lk + jcol * LD] = vn[ib][r];
        __syncthreads();
    }
    if (phase_limit == 17) return;
    PSD_STAMP(16);
    // ---- P = V max(diag d, 0) V': the 10 tiles on and below the diagonal, round-robin over the wavefronts, into M's array
    {
        constexpr int NT = 3;
        v4d acc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int tt = w + 4 * q;
            acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
            if (tt < 10) {
                const int jb = tt < 4 ? 0 : (tt < 7 ? 1 : (tt < 9 ? 2 : 3));
                const int ib = tt < 4 ? tt : (tt < 7 ? tt - 3 : (tt < 9 ? tt - 5 : 3));
                const double* __restrict__ va = Vl + 16 * ib + lr + lk * LD;
                const double* __restrict__ vb = Vl + 16 * jb + lr + lk * LD;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const double f = fmax(dl[kk * 4 + lk], 0.0);
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kk * 4 * LD], f * vb[kk * 4 * LD], acc[q], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int tt = w + 4 * q;
            if (tt < 10) {
                const int jb = tt < 4 ? 0 : (tt < 7 ? 1 : (tt < 9 ? 2 : 3));
                const int ib = tt < 4 ? tt : (tt < 7 ? tt - 3 : (tt < 9 ? tt - 5 : 3));
#pragma unroll
                for (int r = 0; r < 4; ++r) Ml[ib * 16 + lk + 4 * r + (jb * 16 + lr) * LD] = acc[q][r];
            }
        }
    }
    PSD_STAMP(17);
    // the new basis (coalesced, from LDS)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = 2 * (tid + 256 * q);
        Vn[tid + 256 * q] = *reinterpret_cast<const d2*>(Vl + (e & 63) + (e >> 6) * LD);
    }
    __syncthreads();
    PSD_STAMP(18);
    if (fz.on) {
        double sv[9], xo[9]; unsigned short ij[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int idx = q * 256 + tid;
            ij[q] = psd64_index.ij[idx < P64_LEN ? idx : 0];
            sv[q] = idx < P64_LEN ? solc[2 * (int64_t)idx] : 0.0;
            xo[q] = idx < P64_LEN ? xcomp[2 * (int64_t)idx] : 0.0;
        }
        const double b1 = 1.0 - fz.a1, b2 = 1.0 - fz.alpha2, b = 1.0 - fz.alpha;
        double* __restrict__ shc = fz.shift ? reinterpret_cast<double*>(fz.shift + cd.start) + part : nullptr;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int idx = q * 256 + tid;
            const int i = ij[q] & 0xFF, j = ij[q] >> 8;
            double v = Ml[i + j * LD];
            if (i == j) v *= INV_SQRT2;
            const double t1 = fz.a1 * sv[q] + b1 * xo[q];              // (the same expression the input was formed with)
            const double t2 = (dual ? t1 : 0.0) + v;
            const double xn = fz.alpha * (fz.alpha2 * t2 + b2 * t1) + b * xo[q];
            if (idx < P64_LEN) {
                xcomp[2 * (int64_t)idx] = xn;
                if (shc) shc[2 * (int64_t)idx] = part == 1 ? sv[q] - xn : sv[q];
            }
        }
        PSD_STAMP(19);
        return;
    }
    {
        double xv[9]; unsigned short ij[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int idx = q * 256 + tid;
            ij[q] = psd64_index.ij[idx < P64_LEN ? idx : 0];
            xv[q] = (dual && idx < P64_LEN) ? x[2 * (int64_t)idx] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int idx = q * 256 + tid;
            const int i = ij[q] & 0xFF, j = ij[q] >> 8;
            double v = Ml[i + j * LD];
            if (i == j) v *= INV_SQRT2;
            if (idx < P64_LEN) y[2 * (int64_t)idx] = xv[q] + v;
        }
    }
    PSD_STAMP(19);
}

}  // namespace fos
// (not part of the ABI: timing experiments -- tools/psd_stamps.py; -1 unless compiled with -DFOS_PSD_STAMPS)
extern "C" int fos_debug_psd_stamps(long long* out, int n) {
#ifdef FOS_PSD_STAMPS
    if (n > 4 * 128) n = 4 * 128;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fos::g_psd_stamps), sizeof(long long) * (size_t)n);
#else
    (void)out; (void)n; return -1;
#endif
}
namespace fos {

size_t psd_scratch_bytes(int kmax, int ncones) {
    if (ncones <= 0) return 0;
    if (psd_lds_bytes(kmax) <= 160 * 1024 - 256) return 0;
    return (size_t)2 * ncones * (size_t)(kmax * psd_ld(kmax) + 2 * kmax + 16) * sizeof(double);
}

size_t psd_basis_doubles(int kmax, int ncones) {       // one warm-start basis buffer (two are kept, ping-pong)
    if (ncones <= 0 || kmax > 64) return 0;
    return (size_t)2 * ncones * (size_t)kmax * kmax;
}

// whether launch_cones_psd would take the refinement kernel (the only one the step fusion is built into)
// Whether a warm order-64 batch goes to the refinement kernel (FOS_PSD_REFINE=0 / 1 switches it off / on for every batch size; by default it
// runs at every size: 128 / 256 / 512 / 1024 matrices 55 / 58 / 89 / 146 us against 121 / 128 / 178 / 201 for Jacobi).  One exception: when peer
// ranks share THIS device (several ranks on one GPU: tests only) a workgroup of it -- a whole CU's registers and 100 KB of LDS -- finds no CU
// while a peer's CG kernel is resident there SPINNING for this rank's mailbox words, which this rank writes only after its projection:
// two ranks of the full C4 on one GPU dead-locked until the exchange timed out.  Then only batches small enough to fit beside everything.
static bool psd_refine_chosen(const LaunchCtx& c, int ncones) {
    if (c.psd_refine == 0 || (c.psd_refine < 0 && c.psd_wave == 1)) return false;
    if (c.psd_refine_max_mats > 0 && 2 * ncones > c.psd_refine_max_mats) return false;
    return true;
}
bool psd_fuse_possible(const LaunchCtx& c, int ncones, int kmin, int kmax, const double* vin, const double* vout, int have_prev, const int32_t* redo, int phase_limit) {
    const bool refine_ok = ncones > 0 && kmin == 64 && kmax == 64 && vin && vout && have_prev && redo && phase_limit == 0;
    return refine_ok && psd_refine_chosen(c, ncones);
}

int launch_cones_psd(const LaunchCtx& c, double2* out, const double2* in, const ConeDesc* cones, int ncones, int kmin, int kmax, double* gscratch,
                     const double* vin, double* vout, int have_prev, int* stats, int phase_limit, int32_t* redo, const PsdFuse* fuse) {
    if (ncones <= 0) return fuse ? FOS_EINVAL : FOS_OK;
    // per-handle configuration (fos_create reads the device's CU count and the FOS_PSD_* switches ONCE: two host threads driving
    // handles on different devices share nothing here, and the kernel choice cannot change between a speculative enqueue and its re-run)
    const int cus = c.cus > 0 ? c.cus : 256;
    // every cone of order 64: the sweeps run in registers, one WAVEFRONT per matrix when there are more than two matrices per CU
    // (every SIMD then has a wavefront of its own: 1024 matrices 195 us against 288), one WORKGROUP of four wavefronts per matrix
    // (psd_kernel<.., 256>, jacobi64_regs) below that (a shard of a multi-GPU run; 128 / 256 / 512 matrices: 118 / 127 / 179 us
    // against 173 / 182 / 190).  FOS_PSD_WAVE=0 / 1 forces the workgroup / the wavefront form.
    const int wave_env = c.psd_wave;
    // warm, every cone of order 64, a basis from the previous projection: refinement by matrix products (psd64_refine_kernel) with the
    // Jacobi code in the same workgroup for the matrices it flags (psd_refine_chosen above: when)
    const bool refine_ok = kmin == 64 && kmax == 64 && vin && vout && have_prev && redo && (phase_limit == 0 || phase_limit >= 11);
    if (refine_ok && psd_refine_chosen(c, ncones)) {
        const size_t rl1 = psd64r_lds_bytes(1), rl2 = psd64r_lds_bytes(2);
        if (!*c.psd_attr_set_r) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(psd64_refine_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl1);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(psd64_refine_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl2);
            if (e != hipSuccess) { set_error("hipFuncSetAttribute(psd64_refine_kernel): %s", hipGetErrorString(e)); return FOS_EHIP; }
            *c.psd_attr_set_r = true;
        }
        // more matrices than CUs: two workgroups per CU (256 registers per lane), so that one's vector phases run beside the other's products
        PsdFuse fz{};
        int grid = 2 * ncones;
        if (fuse) { fz = *fuse; fz.on = 1; fz.nmat = 2 * ncones; grid += std::max(16, std::min(256, (int)((fz.l + 16383) / 16384))); }
        const int have = c.psd_extrapolate ? have_prev : 1;
        const double theta = c.psd_theta > 0.0 ? c.psd_theta : RF_THETA;
        if (2 * ncones > cus)
            hipLaunchKernelGGL(psd64_refine_kernel<2>, dim3(grid), dim3(256), rl2, c.stream, out, in, cones, vin, vout, have, stats, redo, phase_limit, c.gate, theta, fz);
        else
            hipLaunchKernelGGL(psd64_refine_kernel<1>, dim3(grid), dim3(256), rl1, c.stream, out, in, cones, vin, vout, have, stats, redo, phase_limit, c.gate, theta, fz);
        return FOS_OK;
    }
    if (fuse) { set_error("launch_cones_psd: the fused step needs the refinement kernel"); return FOS_EINVAL; }
    if (kmin == 64 && kmax == 64 && wave_env != 0 && (wave_env == 1 || ncones > cus)) {
        const size_t wl = psd64w_lds_bytes();
        const size_t vs = (size_t)64 * 64;
        if (vin && vout)
            hipLaunchKernelGGL((psd64_wave_kernel<true>), dim3(2 * ncones), dim3(64), wl, c.stream, out, in, cones, vin, vout, vs, have_prev, stats, phase_limit, c.gate);
        else
            hipLaunchKernelGGL((psd64_wave_kernel<false>), dim3(2 * ncones), dim3(64), wl, c.stream, out, in, cones, nullptr, nullptr, vs, 0, stats, phase_limit, c.gate);
        return FOS_OK;
    }
    const size_t lds = psd_lds_bytes(kmax);
    const bool use_lds = lds <= 160 * 1024 - 256;
    const bool warm = vin && vout && kmax <= 64 && use_lds;
    // few matrices (a shard of a multi-GPU run, a small problem): 512 threads per matrix cut the latency of one
    // projection; many matrices: 256 threads (4 per CU) maximise throughput.  Per DEVICE: a process may hold handles on several.
    // (order 64 everywhere: the 256-thread kernel keeps the column halves in registers and beats 512 threads on a small batch)
    const bool all64 = kmin == 64 && kmax == 64;
    // (order 64 with EIGHT wavefronts per matrix, four rows of the two columns per lane -- psd_kernel<.., 512> through jacobi64_regs<8>,
    //  FOS_PSD_WIDE=1 -- was measured on 128 matrices: 145 us against 118 for four wavefronts; the barrier across eight wavefronts
    //  per step costs more than the halved arithmetic saves)
    const bool wide = (warm && !all64 && (2 * ncones <= 2 * cus) && !c.psd_narrow) || (warm && c.psd_wide);
    if (use_lds && !*c.psd_attr_set) {        // hipFuncSetAttribute acts on the CURRENT device; once per handle
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(psd_kernel<true, false, PSD_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(psd_kernel<true, true, PSD_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(psd_kernel<true, true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(psd_kernel<true, true, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(psd_kernel): %s", hipGetErrorString(e)); return FOS_EHIP; }
        *c.psd_attr_set = true;
    }
    const size_t stride = (size_t)(kmax * psd_ld(kmax) + 2 * kmax + 16);
    const size_t vstride = (size_t)kmax * kmax;
    const int wide_threads = c.psd_wide_threads;
    if (warm && wide && wide_threads == 512)
        hipLaunchKernelGGL((psd_kernel<true, true, 512>), dim3(2 * ncones), dim3(512), lds, c.stream, out, in, cones, gscratch, stride, vin, vout, vstride, have_prev, stats, phase_limit, c.gate);
    else if (warm && wide && wide_threads == 1024)
        hipLaunchKernelGGL((psd_kernel<true, true, 1024>), dim3(2 * ncones), dim3(1024), lds, c.stream, out, in, cones, gscratch, stride, vin, vout, vstride, have_prev, stats, phase_limit, c.gate);
    else if (warm)
        hipLaunchKernelGGL((psd_kernel<true, true, PSD_THREADS>), dim3(2 * ncones), dim3(PSD_THREADS), lds, c.stream, out, in, cones, gscratch, stride, vin, vout, vstride, have_prev, stats, phase_limit, c.gate);
    else if (use_lds)
        hipLaunchKernelGGL((psd_kernel<true, false, PSD_THREADS>), dim3(2 * ncones), dim3(PSD_THREADS), lds, c.stream, out, in, cones, gscratch, stride, nullptr, nullptr, vstride, 0, stats, phase_limit, c.gate);
    else
        hipLaunchKernelGGL((psd_kernel<false, false, PSD_THREADS>), dim3(2 * ncones), dim3(PSD_THREADS), 32 * sizeof(double), c.stream, out, in, cones, gscratch, stride, nullptr, nullptr, vstride, 0, stats, phase_limit, c.gate);
    return FOS_OK;
}

}  // namespace fos
