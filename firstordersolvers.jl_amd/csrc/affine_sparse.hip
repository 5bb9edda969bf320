// IndAffine(A, b) with a SPARSE A on the device: the exact projection  y = x - A'(A A')^-1 (A x - b)  without forming anything dense.
// Replaces, for the Feasibility form, ProximalOperators.IndAffine over a sparse matrix (src/problemforms/Feasibility/Feasibility.jl:2-6 takes
// any two ProximableFunctions; the package factorises A -- source not in the checkout, the projection itself is unique).
//
//   * set-up (host): the rows of [A | b] are scaled to unit norm -- the set {x : A x = b} does not change, and the scaling is the Jacobi
//     preconditioner of A A' for free; A and A' are stored as two CSR matrices (fp64 values, int32 columns).
//   * a projection: lambda (kept from the previous projection: the iterates of a solve move slowly) is corrected by conjugate gradients on
//     A A' d = A y0 - b,  y0 = x - A' lambda, with p'A A'p taken as |A'p|^2; every launch is gated on a device flag, the host enqueues a batch
//     of iterations and looks at the flag once per batch.  Three launches per iteration: q = A'p (+ |q|^2), [alpha; lambda += alpha p;
//     r -= alpha A q (+ |r|^2)], [beta; p = r + beta p; stop test].
//   * exactness: the answer is always x - A'lambda (so y - x is in the range of A' by construction) and the TRUE residual A y - b is recomputed
//     from lambda after CG stops; CG continues from there while it is above the rounding level of the residual's own evaluation,
//     16 eps | |A| |y| + |b| |.  What is left is the error of an exact projection onto a set moved by that rounding level.
// Sums are formed in a fixed order (per-workgroup partials added in index order): bit-reproducible from run to run.
#include "fos_internal.hpp"
#include "dev_common.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace fos {

namespace {

constexpr int SA_THREADS = 256;
constexpr int SA_PARTS_MAX = 1024;            // workgroups of any launch (= length of a partial-sum array)

struct SaState {                              // device scalars of the solve (+ a pinned host copy)
    double rr[2];                             // |r|^2 of iteration k at rr[k & 1]
    double scale2;                            // | |A| |y0| + |b| |^2 of the last true-residual evaluation
    double tol;                               // CG stops at |r| <= tol
    int32_t iter, done, maxit, pad;
};

enum SaMode : int { SA_Q = 0, SA_UPD = 1, SA_Y = 2, SA_R = 3 };

__device__ __forceinline__ double sa_block_sum(double v, double* sh) {          // fixed order: lanes by butterfly, wavefronts in index order
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < SA_THREADS / 64; ++i) s += sh[i];
    return s;
}
__device__ __forceinline__ double sa_total(const double* __restrict__ part, int nparts, double* sh) {       // every workgroup adds the same partials in the same order
    double v = 0.0;
    for (int i = threadIdx.x; i < nparts; i += SA_THREADS) v += part[i];
    return sa_block_sum(v, sh);
}

// one row per group of LPR lanes; rows dealt to workgroups in contiguous chunks (grid-stride over chunks of SA_THREADS / LPR rows)
//   SA_Q   (matrix = A', n rows):  q_i = A'[i,:] p                                   part0[wg] = sum q_i^2
//   SA_UPD (matrix = A,  m rows):  t_j = A[j,:] q;  alpha = rr / sum(part0);  lam_j += alpha p_j;  r_j -= alpha t_j;  part1[wg] = sum r_j^2
//   SA_Y   (matrix = A', n rows):  y_i = x_i - A'[i,:] lam
//   SA_R   (matrix = A,  m rows):  r_j = A[j,:] y - b_j;  p_j = r_j;  part1[wg] = sum r_j^2;  part2[wg] = sum (|A|[j,:] |y| + |b_j|)^2
template <int LPR, int MODE>
__global__ __launch_bounds__(SA_THREADS) void sa_rows_kernel(int64_t nrows, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci, const double* __restrict__ va,
                                                             const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ aux,
                                                             double* __restrict__ lam, double* __restrict__ r, double* __restrict__ pvec,
                                                             double* __restrict__ part, int nparts_prev, SaState* __restrict__ st, int k, int gated) {
    __shared__ double sh[SA_THREADS / 64];
    if (gated && st->done) return;
    double alpha = 0.0;
    if (MODE == SA_UPD) {
        const double qq = sa_total(part, nparts_prev, sh);                 // part0 of the SA_Q launch in front
        alpha = qq > 0.0 ? st->rr[k & 1] / qq : 0.0;
    }
    constexpr int RPB = SA_THREADS / LPR;
    const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
    double acc1 = 0.0, acc2 = 0.0;
    for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < nrows; row0 += (int64_t)gridDim.x * RPB) {
        const int64_t row = row0 + rloc;
        double s = 0.0, sa = 0.0;
        if (row < nrows) {
            const int32_t e0 = rp[row], e1 = rp[row + 1];
            for (int32_t e = e0 + sub; e < e1; e += LPR) {
                const double a = va[e], v = in[ci[e]];
                s += a * v;
                if (MODE == SA_R) sa += fabs(a) * fabs(v);
            }
        }
#pragma unroll
        for (int d = LPR >> 1; d > 0; d >>= 1) {
            s += __shfl_xor(s, d, 64);
            if (MODE == SA_R) sa += __shfl_xor(sa, d, 64);
        }
        if (row < nrows && sub == 0) {
            if (MODE == SA_Q) { out[row] = s; acc1 += s * s; }
            else if (MODE == SA_UPD) {
                lam[row] += alpha * pvec[row];
                const double rn = r[row] - alpha * s;
                r[row] = rn; acc1 += rn * rn;
            } else if (MODE == SA_Y) { out[row] = aux[row] - s; }
            else {
                const double bj = aux[row], rn = s - bj, sc = sa + fabs(bj);
                r[row] = rn; pvec[row] = rn; acc1 += rn * rn; acc2 += sc * sc;
            }
        }
    }
    if (MODE == SA_Q) { const double t = sa_block_sum(acc1, sh); if (threadIdx.x == 0) part[blockIdx.x] = t; }
    if (MODE == SA_UPD || MODE == SA_R) {
        const double t = sa_block_sum(acc1, sh);
        if (threadIdx.x == 0) part[SA_PARTS_MAX + blockIdx.x] = t;
        if (MODE == SA_R) { const double t2 = sa_block_sum(acc2, sh); if (threadIdx.x == 0) part[2 * SA_PARTS_MAX + blockIdx.x] = t2; }
    }
}

// closes iteration k: beta = rr_new / rr_old, p = r + beta p, the stop test (workgroup 0 writes the scalars)
__global__ __launch_bounds__(SA_THREADS) void sa_close_kernel(int64_t m, const double* __restrict__ r, double* __restrict__ p, const double* __restrict__ part,
                                                              int nparts_prev, SaState* __restrict__ st, int k) {
    __shared__ double sh[SA_THREADS / 64];
    if (st->done) return;
    const double rr_new = sa_total(part + SA_PARTS_MAX, nparts_prev, sh);
    const double rr_old = st->rr[k & 1];
    const double beta = rr_old > 0.0 ? rr_new / rr_old : 0.0;
    for (int64_t j = (int64_t)blockIdx.x * SA_THREADS + threadIdx.x; j < m; j += (int64_t)gridDim.x * SA_THREADS) p[j] = r[j] + beta * p[j];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rr[(k + 1) & 1] = rr_new;
        const int it = st->iter + 1;
        st->iter = it;
        if (!(sqrt(rr_new) > st->tol) || it >= st->maxit) st->done = 1;          // (a NaN stops the solve too: the host reports it)
    }
}

// behind SA_R: the scalars of a (re)start -- rr[0], the rounding level of this residual, the tolerance from it
__global__ __launch_bounds__(SA_THREADS) void sa_start_kernel(const double* __restrict__ part, int nparts_prev, SaState* __restrict__ st, double tol_factor, int maxit) {
    __shared__ double sh[SA_THREADS / 64];
    const double rr = sa_total(part + SA_PARTS_MAX, nparts_prev, sh);
    const double sc2 = sa_total(part + 2 * SA_PARTS_MAX, nparts_prev, sh);
    if (threadIdx.x == 0) {
        const double tol = tol_factor * sqrt(sc2);
        st->rr[0] = rr; st->rr[1] = rr; st->scale2 = sc2; st->tol = tol; st->iter = 0; st->maxit = maxit;
        st->done = !(sqrt(rr) > tol) ? 1 : 0;
    }
}

int lanes_per_row(int64_t nnz, int64_t rows) {
    const double avg = rows > 0 ? (double)nnz / (double)rows : 1.0;
    int l = 1;
    while (l < 64 && 2 * l <= avg) l <<= 1;                      // the largest power of two not above the average row length
    return l;
}

}  // namespace

struct SparseAffine {
    int64_t m = 0, n = 0, nnz = 0;
    int32_t *rp = nullptr, *ci = nullptr, *trp = nullptr, *tci = nullptr;
    double *va = nullptr, *tva = nullptr;
    double *b = nullptr, *lam = nullptr, *r = nullptr, *p = nullptr, *q = nullptr, *part = nullptr;
    SaState *st = nullptr, *st_host = nullptr;
    int lpr_a = 1, lpr_t = 1, grid_a = 1, grid_t = 1, grid_m = 1;
    std::vector<void*> owned;
    // statistics (fos_feas_affine_stats)
    int64_t calls = 0, iters_total = 0;
    int last_iters = 0, last_rounds = 0;
    double last_resid = 0.0, last_level = 0.0;
};

namespace {

template <int MODE>
void sa_launch_rows(hipStream_t s, int lpr, int grid, int64_t nrows, const int32_t* rp, const int32_t* ci, const double* va, const double* in, double* out,
                    const double* aux, double* lam, double* r, double* p, double* part, int nparts_prev, SaState* st, int k, int gated) {
#define SA_CASE(L) case L: hipLaunchKernelGGL((sa_rows_kernel<L, MODE>), dim3(grid), dim3(SA_THREADS), 0, s, nrows, rp, ci, va, in, out, aux, lam, r, p, part, nparts_prev, st, k, gated); break;
    switch (lpr) { SA_CASE(1) SA_CASE(2) SA_CASE(4) SA_CASE(8) SA_CASE(16) SA_CASE(32) default: SA_CASE(64) }
#undef SA_CASE
}

template <class T>
int sa_upload(SparseAffine* a, T** out, const std::vector<T>& v) {
    void* q = nullptr;
    FOS_HIP(hipMalloc(&q, sizeof(T) * std::max<size_t>(v.size(), 1)));
    a->owned.push_back(q);
    if (!v.empty()) FOS_HIP(hipMemcpy(q, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    *out = static_cast<T*>(q);
    return FOS_OK;
}
int sa_zeros(SparseAffine* a, double** out, size_t count) {
    void* q = nullptr;
    FOS_HIP(hipMalloc(&q, sizeof(double) * std::max<size_t>(count, 1)));
    a->owned.push_back(q);
    FOS_HIP(hipMemset(q, 0, sizeof(double) * std::max<size_t>(count, 1)));
    *out = static_cast<double*>(q);
    return FOS_OK;
}

}  // namespace

void sparse_affine_destroy(SparseAffine* a) {
    if (!a) return;
    for (void* q : a->owned) (void)hipFree(q);
    if (a->st_host) (void)hipHostFree(a->st_host);
    delete a;
}

// A: m x n CSC, 1-based (Julia SparseMatrixCSC{Float64,Int64}); b[m].  Rows of [A | b] scaled to unit norm; A and A' to CSR; upload.
int sparse_affine_setup(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, int cus, SparseAffine** out) {
    if (m < 1 || n < 1 || !colptr || !rowval || !nzval || !b) { set_error("IndAffine (sparse): bad argument"); return FOS_EINVAL; }
    if (colptr[0] != 1) { set_error("IndAffine (sparse): colptr must be 1-based (colptr[0] = %lld)", (long long)colptr[0]); return FOS_EINVAL; }
    const int64_t nnz = colptr[n] - 1;
    if (nnz < 1 || nnz > 2000000000LL) { set_error("IndAffine (sparse): %lld stored entries (1 .. 2e9 supported)", (long long)nnz); return FOS_EINVAL; }
    std::vector<double> rn2((size_t)m, 0.0);
    std::vector<int32_t> rcount((size_t)m + 1, 0);
    for (int64_t j = 0; j < n; ++j) {
        if (colptr[j + 1] < colptr[j]) { set_error("IndAffine (sparse): colptr decreases at column %lld", (long long)j + 1); return FOS_EINVAL; }
        for (int64_t e = colptr[j] - 1; e < colptr[j + 1] - 1; ++e) {
            const int64_t i = rowval[e] - 1;
            const double v = nzval[e];
            if (i < 0 || i >= m) { set_error("IndAffine (sparse): row index %lld outside 1..%lld", (long long)rowval[e], (long long)m); return FOS_EINVAL; }
            if (!(v == v) || std::fabs(v) > 1e300) { set_error("IndAffine (sparse): A has non-finite entries"); return FOS_EINVAL; }
            rn2[(size_t)i] += v * v;
            rcount[(size_t)i + 1]++;
        }
    }
    std::vector<double> scale((size_t)m), bs((size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        if (!(rn2[(size_t)i] > 0.0)) { set_error("IndAffine (sparse): row %lld of A is zero (A must have full row rank)", (long long)i + 1); return FOS_EINVAL; }
        if (!(b[i] == b[i]) || std::fabs(b[i]) > 1e300) { set_error("IndAffine (sparse): b has non-finite entries"); return FOS_EINVAL; }
        scale[(size_t)i] = 1.0 / std::sqrt(rn2[(size_t)i]);
        bs[(size_t)i] = b[i] * scale[(size_t)i];
    }
    // A' in CSR = the CSC arrays as they are (0-based, scaled); A in CSR by a counting pass (columns ascending inside a row)
    std::vector<int32_t> trp((size_t)n + 1), tci((size_t)nnz), rp((size_t)m + 1, 0), ci((size_t)nnz);
    std::vector<double> tva((size_t)nnz), va((size_t)nnz);
    for (int64_t i = 0; i < m; ++i) rp[(size_t)i + 1] = rp[(size_t)i] + rcount[(size_t)i + 1];
    std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
    for (int64_t j = 0; j <= n; ++j) trp[(size_t)j] = (int32_t)(colptr[j] - 1);
    for (int64_t j = 0; j < n; ++j)
        for (int64_t e = colptr[j] - 1; e < colptr[j + 1] - 1; ++e) {
            const int64_t i = rowval[e] - 1;
            const double v = nzval[e] * scale[(size_t)i];
            tci[(size_t)e] = (int32_t)i; tva[(size_t)e] = v;
            const int32_t pos = fill[(size_t)i]++;
            ci[(size_t)pos] = (int32_t)j; va[(size_t)pos] = v;
        }
    SparseAffine* a = new SparseAffine();
    a->m = m; a->n = n; a->nnz = nnz;
    int rc = FOS_OK;
    auto fail = [&](int code) { sparse_affine_destroy(a); return code; };
    if ((rc = sa_upload(a, &a->rp, rp)) != FOS_OK || (rc = sa_upload(a, &a->ci, ci)) != FOS_OK || (rc = sa_upload(a, &a->va, va)) != FOS_OK ||
        (rc = sa_upload(a, &a->trp, trp)) != FOS_OK || (rc = sa_upload(a, &a->tci, tci)) != FOS_OK || (rc = sa_upload(a, &a->tva, tva)) != FOS_OK ||
        (rc = sa_upload(a, &a->b, bs)) != FOS_OK || (rc = sa_zeros(a, &a->lam, (size_t)m)) != FOS_OK || (rc = sa_zeros(a, &a->r, (size_t)m)) != FOS_OK ||
        (rc = sa_zeros(a, &a->p, (size_t)m)) != FOS_OK || (rc = sa_zeros(a, &a->q, (size_t)n)) != FOS_OK || (rc = sa_zeros(a, &a->part, 3 * SA_PARTS_MAX)) != FOS_OK)
        return fail(rc);
    void* q = nullptr;
    if (hipMalloc(&q, sizeof(SaState)) != hipSuccess) { set_error("IndAffine (sparse): hipMalloc failed"); return fail(FOS_ENOMEM); }
    a->owned.push_back(q); a->st = static_cast<SaState*>(q);
    if (hipMemset(q, 0, sizeof(SaState)) != hipSuccess) { set_error("IndAffine (sparse): hipMemset failed"); return fail(FOS_EHIP); }
    if (hipHostMalloc((void**)&a->st_host, sizeof(SaState), hipHostMallocDefault) != hipSuccess) { set_error("IndAffine (sparse): hipHostMalloc failed"); return fail(FOS_ENOMEM); }
    a->lpr_a = lanes_per_row(nnz, m); a->lpr_t = lanes_per_row(nnz, n);
    const int wg_max = std::min(SA_PARTS_MAX, 4 * std::max(1, cus));
    auto grid_for = [&](int64_t rows, int lpr) { return (int)std::max<int64_t>(1, std::min<int64_t>(wg_max, (rows + SA_THREADS / lpr - 1) / (SA_THREADS / lpr))); };
    a->grid_a = grid_for(m, a->lpr_a); a->grid_t = grid_for(n, a->lpr_t);
    a->grid_m = (int)std::max<int64_t>(1, std::min<int64_t>(wg_max, (m + SA_THREADS - 1) / SA_THREADS));
    *out = a;
    return FOS_OK;
}

// y = the projection of x onto {A x = b} (device vectors of length n, y must not alias x), on `stream`
int sparse_affine_project(SparseAffine* a, hipStream_t stream, double* y, const double* x) {
    constexpr double TOL_FACTOR = 16.0 * 2.220446049250313e-16;
    constexpr int BATCH = 24, ROUNDS_MAX = 6;
    const int maxit = (int)std::min<int64_t>(20000, 2 * a->m + 100);
    int total = 0, round = 0;
    a->calls++;
    for (;; ++round) {
        // y = x - A' lambda; r = p = A y - b; the scalars of the (re)start
        sa_launch_rows<SA_Y>(stream, a->lpr_t, a->grid_t, a->n, a->trp, a->tci, a->tva, a->lam, y, x, nullptr, nullptr, nullptr, a->part, 0, a->st, 0, 0);
        sa_launch_rows<SA_R>(stream, a->lpr_a, a->grid_a, a->m, a->rp, a->ci, a->va, y, nullptr, a->b, nullptr, a->r, a->p, a->part, 0, a->st, 0, 0);
        hipLaunchKernelGGL(sa_start_kernel, dim3(1), dim3(SA_THREADS), 0, stream, (const double*)a->part, a->grid_a, a->st, TOL_FACTOR, maxit);
        FOS_HIP(hipMemcpyAsync(a->st_host, a->st, sizeof(SaState), hipMemcpyDeviceToHost, stream));
        FOS_HIP(hipStreamSynchronize(stream));
        const double resid = std::sqrt(a->st_host->rr[0]), level = std::sqrt(a->st_host->scale2);
        a->last_resid = resid; a->last_level = level;
        if (!(resid == resid)) { set_error("IndAffine (sparse): the residual is NaN (non-finite input?)"); return FOS_EINVAL; }
        if (a->st_host->done) break;                       // the true residual is at its rounding level: y is the projection
        if (round >= ROUNDS_MAX) {
            set_error("IndAffine (sparse): |A y - b| = %.3e stays above its rounding level %.3e after %d restarts and %d CG iterations "
                      "(A without full row rank?)", resid, TOL_FACTOR * level, round, total);
            return FOS_EINVAL;
        }
        int k = 0;
        for (;;) {
            for (int q = 0; q < BATCH; ++q, ++k) {
                sa_launch_rows<SA_Q>(stream, a->lpr_t, a->grid_t, a->n, a->trp, a->tci, a->tva, a->p, a->q, nullptr, nullptr, nullptr, nullptr, a->part, 0, a->st, k, 1);
                sa_launch_rows<SA_UPD>(stream, a->lpr_a, a->grid_a, a->m, a->rp, a->ci, a->va, a->q, nullptr, nullptr, a->lam, a->r, a->p, a->part, a->grid_t, a->st, k, 1);
                hipLaunchKernelGGL(sa_close_kernel, dim3(a->grid_m), dim3(SA_THREADS), 0, stream, a->m, (const double*)a->r, a->p, (const double*)a->part, a->grid_a, a->st, k);
            }
            FOS_HIP(hipMemcpyAsync(a->st_host, a->st, sizeof(SaState), hipMemcpyDeviceToHost, stream));
            FOS_HIP(hipStreamSynchronize(stream));
            if (a->st_host->done) break;
        }
        total += a->st_host->iter;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("IndAffine (sparse): a kernel launch failed: %s", hipGetErrorString(e)); return FOS_EHIP; }
    a->last_iters = total; a->last_rounds = round; a->iters_total += total;
    return FOS_OK;
}

void sparse_affine_reset(SparseAffine* a, hipStream_t stream) {           // a new solve starts from lambda = 0
    if (a) (void)hipMemsetAsync(a->lam, 0, sizeof(double) * (size_t)a->m, stream);
}

void sparse_affine_stats(const SparseAffine* a, double* out8) {
    out8[0] = (double)a->calls; out8[1] = (double)a->iters_total; out8[2] = (double)a->last_iters; out8[3] = (double)a->last_rounds;
    out8[4] = a->last_resid; out8[5] = a->last_level; out8[6] = (double)a->nnz; out8[7] = (double)(a->lpr_a * 1000 + a->lpr_t);
}

}  // namespace fos
