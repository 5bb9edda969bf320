// PSD projection of LARGE orders (k > 64: what a Convex.jl SDP with one big matrix variable hands over) by MATRIX PRODUCTS ONLY, on the fp64 matrix cores:
//
//     P_PSD(M) = (M + M sign(M)) / 2 ,     sign(M) = the limit of an inverse-free polynomial iteration  X <- X q(X^2),  X_0 = M / ||M||_F
//
// (cones.jl:11,89-94 -> ProximalOperators IndPSD(scaling=true) -> LAPACK's symmetric eigen-decomposition in the reference; the same unique point.)
// Why not the Jacobi kernel of psd.hip: beyond order 64 its array leaves the LDS, there is no warm start, and one workgroup walks k steps of k / 2 rotations per
// sweep -- measured (tools/psd_orders.py, profiles/r06_psd_orders.json) 2.7 ms at order 96, 62 ms at 200, 124 ms at 256 per projection, 30-140 x what the
// same flops take as matrix products.  A product of two order-K matrices, by contrast, is a grid of 64 x 64 tiles: ONE big matrix spreads over (K / 64)^2 CUs.
//
// The iteration.  For a symmetric X with eigenvalues in [-1, 1] the odd polynomial p(x) = x q(x^2) maps every eigenvalue by p and keeps the eigenvectors.
//   * GROWTH steps:  p(x) = a x + b x^3 + c x^5 with (a, b, c) = (3.4445, -4.7750, 2.0315) -- a quintic whose slope at 0 is 3.44 and which maps [0.68, 1.19]
//     into itself: an eigenvalue of relative size 1e-14 reaches that band in 26 steps (the classical Newton-Schulz cubic, slope 1.5, needs 72).  Three
//     products per step: A = X X;  T = c A A + b A + a I;  X <- X T.
//   * CONVERGENCE steps:  Newton-Schulz  p(x) = (3 x - x^3) / 2: quadratic from anywhere in (0, sqrt 3): six steps take 0.68 to 1 - 3e-23.  Two products.
//   * eigenvalues below ~1e-14 ||M||_F are not resolved: their sign stays undecided, and the error they leave in P is at most their own size.
//   Perturbations that do not commute with M (rounding) are damped between eigenvectors of equal sign and merely carried between opposite signs: the result
//   is LAPACK class (measured against numpy.linalg.eigh: tests/test_gpu_parity.py::test_psd_known_answer_and_sizes, orders up to 150, clustered spectra).
// Stateless (no basis from the previous call, nothing to fall back from), the same cost cold and warm, deterministic.
#include <algorithm>
#include <vector>

#include "fos_internal.hpp"
#include "dev_common.hpp"

namespace fos {

constexpr int PSG_THREADS = 256;
constexpr double PSG_SQRT2 = 1.4142135623730951, PSG_INV_SQRT2 = 0.7071067811865475;
constexpr int PSG_GROWTH_STEPS = 26, PSG_NEWTON_STEPS = 6;
constexpr double PSG_A = 3.4445, PSG_B = -4.7750, PSG_C = 2.0315;

typedef double psg_v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void psg_idx_to_ij(int idx, int k, int& i, int& j) {
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

// One workgroup per matrix (cone, copy): M = sqrt(2) smat(sgn x) (the scaling of psd.hip: off-diagonals as stored, diagonal x sqrt 2), zero padded to order K;
// X = M / ||M||_F.  The dual copy projects -x (Moreau: y = x + P(-x), cones.jl:80-85).
__global__ __launch_bounds__(PSG_THREADS) void psd_sign_unpack_kernel(const d2* __restrict__ in, const ConeDesc* __restrict__ cones, int K, size_t ms,
                                                                      double* __restrict__ M, double* __restrict__ X, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    __shared__ double red[8];
    const int tid = threadIdx.x, cone = blockIdx.x >> 1, part = blockIdx.x & 1;
    const ConeDesc cd = cones[cone];
    const int k = cd.k, len = cd.len;
    const double sgn = (cd.dual_part == part) ? -1.0 : 1.0;
    const double* __restrict__ x = reinterpret_cast<const double*>(in + cd.start) + part;
    double* __restrict__ Mb = M + (size_t)blockIdx.x * ms;
    double* __restrict__ Xb = X + (size_t)blockIdx.x * ms;
    for (size_t q = tid; q < (size_t)K * K; q += PSG_THREADS) Mb[q] = 0.0;
    __syncthreads();
    double fro = 0.0;
    for (int idx = tid; idx < len; idx += PSG_THREADS) {
        int i, j;
        psg_idx_to_ij(idx, k, i, j);
        double v = sgn * x[2 * (int64_t)idx];
        if (i == j) { v *= PSG_SQRT2; fro += v * v; }
        else fro += 2.0 * v * v;
        Mb[i + (size_t)j * K] = v;
        Mb[j + (size_t)i * K] = v;
    }
    fro = wave_sum(fro);
    if ((tid & 63) == 0) red[tid >> 6] = fro;
    __syncthreads();
    double f2 = 0.0;
    for (int w = 0; w < PSG_THREADS / 64; ++w) f2 += red[w];
    const double inv = f2 > 0.0 ? 1.0 / sqrt(f2) : 0.0;
    for (size_t q = tid; q < (size_t)K * K; q += PSG_THREADS) Xb[q] = Mb[q] * inv;
}

// Batched C = alpha A B + gamma D + diag I on v_mfma_f64_16x16x4_f64: order K (a multiple of 64), column-major, batch = blockIdx.z, 64 x 64 tile per workgroup
// of four wavefronts (32 x 32 each), operands staged through LDS 16 deep.  C may alias D (every element is read and written by the same lane).
__global__ __launch_bounds__(PSG_THREADS) void psd_sign_gemm_kernel(int K, size_t ms, double alpha, const double* __restrict__ A, const double* __restrict__ B,
                                                                    double gamma, const double* __restrict__ D, double diag, double* __restrict__ C,
                                                                    const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    __shared__ double As[16][64 + 2];            // As[k][i]
    __shared__ double Bs[16][64 + 2];            // Bs[k][j]
    const size_t off = (size_t)blockIdx.z * ms;
    A += off; B += off; C += off;
    if (D) D += off;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    psg_v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = psg_v4d{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid & 63, kk = (tid >> 6) + 4 * q;
            As[kk][i] = A[(size_t)(i0 + i) + (size_t)(k0 + kk) * K];
            const int k = tid & 15, j = (tid >> 4) + 16 * q;
            Bs[k][j] = B[(size_t)(k0 + k) + (size_t)(j0 + j) * K];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = As[ks + lk][wr + 16 * a + lr];
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[b] = Bs[ks + lk][wc + 16 * b + lr];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // result map of the f64 MFMA: column lane & 15, rows (lane >> 4) + 4 r
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + wr + 16 * a + lk + 4 * r, gj = j0 + wc + 16 * b + lr;
                const size_t idx = (size_t)gi + (size_t)gj * K;
                double v = alpha * acc[a][b][r];
                if (gamma != 0.0) v += gamma * D[idx];
                if (gi == gj) v += diag;
                C[idx] = v;
            }
}

// out = P (primal copy) or x + P (dual copy), P = the symmetric part of the product the last launch left in `Pm` (= (M + M S) / 2), packed, diagonal / sqrt 2
__global__ __launch_bounds__(PSG_THREADS) void psd_sign_pack_kernel(d2* __restrict__ out, const d2* __restrict__ in, const ConeDesc* __restrict__ cones, int K, size_t ms,
                                                                    const double* __restrict__ Pm, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    const int tid = threadIdx.x, cone = blockIdx.x >> 1, part = blockIdx.x & 1;
    const ConeDesc cd = cones[cone];
    const int k = cd.k, len = cd.len;
    const bool dual = cd.dual_part == part;
    const double* __restrict__ x = reinterpret_cast<const double*>(in + cd.start) + part;
    double* __restrict__ y = reinterpret_cast<double*>(out + cd.start) + part;
    const double* __restrict__ Pb = Pm + (size_t)blockIdx.x * ms;
    for (int idx = tid; idx < len; idx += PSG_THREADS) {
        int i, j;
        psg_idx_to_ij(idx, k, i, j);
        double s = 0.5 * (Pb[i + (size_t)j * K] + Pb[j + (size_t)i * K]);
        if (i == j) s *= PSG_INV_SQRT2;
        if (dual) s = x[2 * (int64_t)idx] + s;
        y[2 * (int64_t)idx] = s;
    }
}

// ---- host side: the big cones of a handle, grouped by padded order; workspace for the largest group
struct PsdSignGroup { int K = 0, ncones = 0; ConeDesc* cones = nullptr; };
struct PsdSign {
    std::vector<PsdSignGroup> groups;
    double* work = nullptr;            // five order-K arrays per matrix of the largest group: M, X, X', A, T
    size_t work_doubles = 0;
    std::vector<void*> owned;
};

void psd_sign_destroy(PsdSign* p) {
    if (!p) return;
    for (void* q : p->owned) (void)hipFree(q);
    delete p;
}

// takes the cones of order > 64 out of `psd` (the rest stays for the kernels of psd.hip); nullptr when there are none
int psd_sign_setup(std::vector<ConeDesc>& psd, PsdSign** out) {
    *out = nullptr;
    static const int min_order = getenv("FOS_PSD_SIGN_MIN") ? std::max(1, atoi(getenv("FOS_PSD_SIGN_MIN"))) : 65;      // (tests: smaller orders through this path too)
    std::vector<ConeDesc> small, big;
    for (const ConeDesc& cd : psd) (cd.k >= min_order ? big : small).push_back(cd);
    if (big.empty()) return FOS_OK;
    PsdSign* p = new PsdSign();
    std::stable_sort(big.begin(), big.end(), [](const ConeDesc& a, const ConeDesc& b) { return (a.k + 63) / 64 < (b.k + 63) / 64; });
    size_t need = 0;
    for (size_t q = 0; q < big.size();) {
        const int K = (big[q].k + 63) / 64 * 64;
        size_t e = q;
        while (e < big.size() && (big[e].k + 63) / 64 * 64 == K) ++e;
        PsdSignGroup g;
        g.K = K; g.ncones = (int)(e - q);
        void* d = nullptr;
        if (hipMalloc(&d, sizeof(ConeDesc) * (e - q)) != hipSuccess || hipMemcpy(d, big.data() + q, sizeof(ConeDesc) * (e - q), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("PSD (order > 64): uploading the cone table failed"); psd_sign_destroy(p); return FOS_ENOMEM;
        }
        p->owned.push_back(d);
        g.cones = static_cast<ConeDesc*>(d);
        p->groups.push_back(g);
        need = std::max(need, (size_t)5 * 2 * g.ncones * (size_t)K * K);
        q = e;
    }
    void* w = nullptr;
    if (hipMalloc(&w, need * sizeof(double)) != hipSuccess) { set_error("PSD (order > 64): %zu bytes of workspace", need * sizeof(double)); psd_sign_destroy(p); return FOS_ENOMEM; }
    p->owned.push_back(w);
    p->work = static_cast<double*>(w); p->work_doubles = need;
    psd.swap(small);
    *out = p;
    return FOS_OK;
}

int psd_sign_count(const PsdSign* p) { int n = 0; if (p) for (const PsdSignGroup& g : p->groups) n += g.ncones; return n; }

// prox!(y, IndPSD, x) for every big cone (both copies): 3 x 26 + 2 x 6 + 1 products per matrix, every launch gated like the other cone kernels
int launch_cones_psd_sign(const LaunchCtx& c, PsdSign* p, double2* out, const double2* in) {
    if (!p) return FOS_OK;
    for (const PsdSignGroup& g : p->groups) {
        const int K = g.K, nmat = 2 * g.ncones;
        const size_t ms = (size_t)K * K, bs = ms * nmat;
        double *M = p->work, *X = M + bs, *X2 = X + bs, *A = X2 + bs, *T = A + bs;
        hipLaunchKernelGGL(psd_sign_unpack_kernel, dim3(nmat), dim3(PSG_THREADS), 0, c.stream, in, g.cones, K, ms, M, X, c.gate);
        const dim3 grid(K / 64, K / 64, nmat), block(PSG_THREADS);
        auto gemm = [&](double alpha, const double* Aa, const double* Bb, double gamma, const double* Dd, double diag, double* Cc) {
            hipLaunchKernelGGL(psd_sign_gemm_kernel, grid, block, 0, c.stream, K, ms, alpha, Aa, Bb, gamma, Dd, diag, Cc, c.gate);
        };
        for (int s = 0; s < PSG_GROWTH_STEPS; ++s) {
            gemm(1.0, X, X, 0.0, nullptr, 0.0, A);                 // A = X X
            gemm(PSG_C, A, A, PSG_B, A, PSG_A, T);                 // T = c A A + b A + a I
            gemm(1.0, X, T, 0.0, nullptr, 0.0, X2);                // X <- X T
            std::swap(X, X2);
        }
        for (int s = 0; s < PSG_NEWTON_STEPS; ++s) {
            gemm(1.0, X, X, 0.0, nullptr, 0.0, A);
            gemm(-0.5, X, A, 1.5, X, 0.0, X2);                     // X <- (3 X - X A) / 2
            std::swap(X, X2);
        }
        gemm(0.5, M, X, 0.5, M, 0.0, T);                           // (M + M S) / 2
        hipLaunchKernelGGL(psd_sign_pack_kernel, dim3(nmat), dim3(PSG_THREADS), 0, c.stream, out, in, g.cones, K, ms, T, c.gate);
    }
    return FOS_OK;
}

}  // namespace fos
