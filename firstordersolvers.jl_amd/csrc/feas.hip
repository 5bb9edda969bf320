// Feasibility form on the device: find a point of S1 n S2 with the solvers' own steps.
//   reference: src/problemforms/Feasibility/Feasibility.jl (model, zeros(n) start, populate_solution),
//              src/problemforms/Feasibility/FeasibilityStatus.jl:32-72 (err = norm(prev - z) every checki-th iteration),
//              src/solvers/{gap,gapa,fista,dykstra}.jl (the steps: the same files the HSDE path follows).
// The reference takes ANY two ProximalOperators objects (host callbacks); here the two sets are device objects, the two its own
// test uses (test/testfeasibility.jl:9-10): IndAffine(A, b) -- dense A with full row rank, an exact projection through a one-time
// inverse of A A' (Newton-Schulz on the fp64 MFMA GEMM of vecops.hip, as fos_enable_direct does for the HSDE) -- and
// IndBox(lo, hi).  Vectors are plain n-vectors in HBM; every iteration is a handful of streaming kernels plus, for IndAffine,
// one dense symmetric matrix-vector product (n^2 x 8 bytes: the kernel that bounds the path, HBM).  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "dev_common.hpp"
#include "fos_internal.hpp"

namespace fos {

constexpr int FEAS_THREADS = 256;
constexpr int FEAS_PARTS = 256;               // partial sums per reduction (fixed: results do not depend on the device)

#define FEAS_STRIDE(i, n) for (int64_t i = blockIdx.x * (int64_t)FEAS_THREADS + threadIdx.x; i < (n); i += (int64_t)gridDim.x * FEAS_THREADS)

__global__ __launch_bounds__(FEAS_THREADS) void feas_fill_kernel(int64_t n, double* __restrict__ y, double v) { FEAS_STRIDE(i, n) y[i] = v; }
__global__ __launch_bounds__(FEAS_THREADS) void feas_copy_kernel(int64_t n, double* __restrict__ y, const double* __restrict__ x) { FEAS_STRIDE(i, n) y[i] = x[i]; }
// y = x - P x + q      IndAffine: x - A'(A A')^-1 (A x - b) with P = A'(A A')^-1 A, q = A'(A A')^-1 b
__global__ __launch_bounds__(FEAS_THREADS) void feas_affine_finish_kernel(int64_t n, double* __restrict__ y, const double* __restrict__ x,
                                                                          const double* __restrict__ Px, const double* __restrict__ q) {
    FEAS_STRIDE(i, n) y[i] = (x[i] - Px[i]) + q[i];
}
// y = clamp(x, lo, hi)      IndBox
__global__ __launch_bounds__(FEAS_THREADS) void feas_box_kernel(int64_t n, double* __restrict__ y, const double* __restrict__ x, double lo, double hi) {
    FEAS_STRIDE(i, n) y[i] = fmin(fmax(x[i], lo), hi);
}
// y = clamp(x, lo[i], hi[i])      IndBox with array bounds
__global__ __launch_bounds__(FEAS_THREADS) void feas_boxv_kernel(int64_t n, double* __restrict__ y, const double* __restrict__ x,
                                                                 const double* __restrict__ lo, const double* __restrict__ hi) {
    FEAS_STRIDE(i, n) y[i] = fmin(fmax(x[i], lo[i]), hi[i]);
}
// y = a y + (1 - a) x       gap.jl:48,58 ; gapa.jl:67,77 (a = alpha12, a device scalar) ; fista.jl:37
__global__ __launch_bounds__(FEAS_THREADS) void feas_relax_kernel(int64_t n, double* __restrict__ y, const double* __restrict__ x, double a,
                                                                  const double* __restrict__ a_dev) {
    const double aa = a_dev ? *a_dev : a;
    FEAS_STRIDE(i, n) y[i] = aa * y[i] + (1.0 - aa) * x[i];
}
// out = a + b ; p = a - b
__global__ __launch_bounds__(FEAS_THREADS) void feas_add_kernel(int64_t n, double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b) {
    FEAS_STRIDE(i, n) out[i] = a[i] + b[i];
}
__global__ __launch_bounds__(FEAS_THREADS) void feas_sub_kernel(int64_t n, double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b) {
    FEAS_STRIDE(i, n) out[i] = a[i] - b[i];
}
// y = x + coef (x - xold)      fista.jl:46
__global__ __launch_bounds__(FEAS_THREADS) void feas_extrap_kernel(int64_t n, double* __restrict__ y, const double* __restrict__ x,
                                                                   const double* __restrict__ xold, double coef) {
    FEAS_STRIDE(i, n) y[i] = x[i] + coef * (x[i] - xold[i]);
}

// out = base + a dir      linesearch.jl:58,70
__global__ __launch_bounds__(FEAS_THREADS) void feas_axpy_kernel(int64_t n, double* __restrict__ out, const double* __restrict__ base, double a,
                                                                 const double* __restrict__ dir) {
    FEAS_STRIDE(i, n) out[i] = base[i] + a * dir[i];
}

template <int NACC>
__device__ __forceinline__ void feas_block_store(const double (&acc)[NACC], double* __restrict__ partials) {
    __shared__ double sm[4 * NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const double v = wave_sum(acc[a]);
        if (lane == 0) sm[wave * NACC + a] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double s = 0.0;
        for (int w = 0; w < FEAS_THREADS / 64; ++w) s += sm[w * NACC + threadIdx.x];
        partials[(size_t)blockIdx.x * NACC + threadIdx.x] = s;
    }
}
// partial sums of |a - b|^2      FeasibilityStatus.jl:40
__global__ __launch_bounds__(FEAS_THREADS) void feas_normdiff_kernel(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                                     double* __restrict__ partials) {
    double acc[1] = {0.0};
    FEAS_STRIDE(i, n) { const double d = a[i] - b[i]; acc[0] += d * d; }
    feas_block_store<1>(acc, partials);
}
// alpha12 = (1 - beta) 2 / (1 + sqrt(1 - scl^2)) + 2 beta,  scl = clamp(|s| / sqrt(n1 n2), 0, 1), NaN -> 0      gapa.jl:96-101
__global__ void feas_alpha12_kernel(const double* __restrict__ partials, int nparts, double beta, double* __restrict__ a12) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0, n1 = 0.0, n2 = 0.0;
    for (int b = 0; b < nparts; ++b) { s += partials[3 * b]; n1 += partials[3 * b + 1]; n2 += partials[3 * b + 2]; }
    double scl = fabs(s) / sqrt(n1 * n2);
    if (scl != scl) scl = 0.0;
    scl = fmin(fmax(scl, 0.0), 1.0);
    const double aopt = 2.0 / (1.0 + sqrt(1.0 - scl * scl));
    *a12 = (1.0 - beta) * aopt + beta * 2.0;
}
// checkstatus in ONE pass (FeasibilityStatus.jl:40,61,69): partial sums of |prev - z|^2 when this iteration checks, and prev = z -- every call
__global__ __launch_bounds__(FEAS_THREADS) void feas_check_kernel(int64_t n, double* __restrict__ prev, const double* __restrict__ z, int do_norm,
                                                                  double* __restrict__ partials) {
    double acc[1] = {0.0};
    if (do_norm) {
        FEAS_STRIDE(i, n) { const double zi = z[i], d = prev[i] - zi; acc[0] += d * d; prev[i] = zi; }
        feas_block_store<1>(acc, partials);
    } else {
        FEAS_STRIDE(i, n) prev[i] = z[i];
    }
}
// the tail of a GAP / GAPA step in ONE pass: t2 = a2 t2 + (1 - a2) t1 (gap.jl:58, gapa.jl:77; a2 = alpha12 from the device for GAPA), for GAPA the
// partial sums of normedScalar(t2, t1, t1, x) on the relaxed t2 and the OLD x (gapa.jl:96), then x = alpha t2 + (1 - alpha) x (gap.jl:78, gapa.jl:103)
template <bool GAPA>
__global__ __launch_bounds__(FEAS_THREADS) void feas_gap_tail_kernel(int64_t n, double* __restrict__ x, double* __restrict__ t2, const double* __restrict__ t1,
                                                                     double a2, const double* __restrict__ a_dev, double alpha, double* __restrict__ partials) {
    const double aa = a_dev ? *a_dev : a2;
    double acc[3] = {0.0, 0.0, 0.0};
    FEAS_STRIDE(i, n) {
        const double t1i = t1[i], xi = x[i];
        const double t2i = aa * t2[i] + (1.0 - aa) * t1i;
        t2[i] = t2i;
        if constexpr (GAPA) {
            const double d1 = t2i - t1i, d2 = t1i - xi;
            acc[0] += d1 * d2; acc[1] += d1 * d1; acc[2] += d2 * d2;
        }
        x[i] = alpha * t2i + (1.0 - alpha) * xi;
    }
    if constexpr (GAPA) feas_block_store<3>(acc, partials);
}
// x = alpha t2 + (1 - alpha) x      gap.jl:78, gapa.jl:103
__global__ __launch_bounds__(FEAS_THREADS) void feas_combine_kernel(int64_t n, double* __restrict__ x, const double* __restrict__ t2, double alpha) {
    FEAS_STRIDE(i, n) x[i] = alpha * t2[i] + (1.0 - alpha) * x[i];
}
// plain vector <-> the two-part layout of the cone kernels (part 1 = the vector, part 2 = zeros: its dual projections are of zero)
__global__ __launch_bounds__(FEAS_THREADS) void feas_to_parts_kernel(int64_t n, double2* __restrict__ z, const double* __restrict__ x) {
    FEAS_STRIDE(i, n) z[i] = make_double2(x[i], 0.0);
}
__global__ __launch_bounds__(FEAS_THREADS) void feas_from_parts_kernel(int64_t n, double* __restrict__ y, const double2* __restrict__ z) {
    FEAS_STRIDE(i, n) y[i] = z[i].x;
}
// set-up: the row-major m x n matrix A as a column-major L x L array (rows 0..m-1, zero padded) and its transpose
__global__ __launch_bounds__(FEAS_THREADS) void feas_spread_kernel(int64_t m, int64_t n, int64_t L, const double* __restrict__ A,
                                                                   double* __restrict__ Ah, double* __restrict__ At) {
    FEAS_STRIDE(e, m * n) {
        const int64_t i = e / n, j = e % n;
        const double v = A[e];
        Ah[i + j * L] = v;
        At[j + i * L] = v;
    }
}
// set-up: E[i + i L] = 1 for i >= m (the padding of G = A A' + E keeps it positive definite)
__global__ __launch_bounds__(FEAS_THREADS) void feas_pad_identity_kernel(int64_t L, int64_t m, double* __restrict__ E) {
    FEAS_STRIDE(i, L) if (i >= m) E[i + i * L] = 1.0;
}

struct FeasSet {
    int kind = 0;                   // 0 unset, 1 IndAffine (dense), 2 IndBox, 3 ConeProduct, 4 host callback, 5 IndAffine (sparse A)
    SparseAffine* sa = nullptr;     // kind 5 (affine_sparse.hip)
    fos_prox_fn cb = nullptr;       // kind 4: prox!(y, S, x) evaluated by the caller on pinned host vectors
    void* cb_ctx = nullptr;
    double *cb_x = nullptr, *cb_y = nullptr;
    double* P = nullptr;            // [L x L] A'(A A')^-1 A, column-major
    double* q = nullptr;            // [L]     A'(A A')^-1 b
    double lo = 0.0, hi = 0.0;
    double *lov = nullptr, *hiv = nullptr;      // [n] array bounds (IndBox with vectors); nullptr: the scalars
    int ns_iters = 0;               // Newton-Schulz steps of the set-up
    double ns_resid = 0.0;          // max |G X - I| it ended with
    // kind 3: ConeProduct (cones.jl:31-94) on the batched cone kernels of the HSDE path
    uint8_t* ew_op = nullptr;       // [n] elementwise op per index (EW_SKIP inside SOC / Exp / PSD cones)
    ConeDesc *soc = nullptr, *expc = nullptr, *psd = nullptr;
    int nsoc = 0, nexp = 0, npsd = 0, psd_kmin = 0, psd_kmax = 0;
    double* psd_scratch = nullptr;
    double* psd_V[2] = {nullptr, nullptr};
    int psd_cur = 0, psd_have_prev = 0;
    bool psd_attr_set = false;
    PsdSign* psd_big = nullptr;     // PSD cones of order > 64 (psd_sign.hip)
};

}  // namespace fos

using namespace fos;

struct fos_feas {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0, L = 0;
    FeasSet S[2];
    int alg = FOS_ALG_GAP;
    double alpha = 0.8, alpha1 = 1.8, alpha2 = 1.8, beta = 0.0;
    double fista_t = 1.0;
    // vectors of length L (zero padded)
    double *x = nullptr, *t1 = nullptr, *t2 = nullptr, *y = nullptr, *xold = nullptr, *p = nullptr, *q = nullptr, *prev = nullptr, *tmp = nullptr, *px = nullptr;
    double2 *zin = nullptr, *zout = nullptr;      // [n] two-part scratch of the cone kernels (allocated with the first ConeProduct set)
    int cus = 256;
    double* partials = nullptr;     // [3 x FEAS_PARTS]
    double* a12 = nullptr;          // device scalar alpha12 (GAPA)
    int grid = 1;
    int status = FOS_STATUS_CONTINUE;
    int checked = 0;
    double err = NAN;
    int64_t ls_interval = 0;        // LineSearchWrapper (wrappers/linesearch.jl): every ls_interval-th iteration is a step-length search
    double ls_log[34] = {0};        // normres, 31 test residuals, alpha_best, iteration (the layout of fos_linesearch_log)
    LongPlanes lp;                  // LongstepWrapper (wrappers/longstep.jl): planes saved in the last nsave + 1 iterations of every interval (vectors viewed as L/2 double2)
    int64_t gapp_iproj = 100;       // GAPP (solvers/gapproj.jl): every iproj-th iteration is a projected search
    double gapp_log[23] = {0};      // 21 test norms, alpha_best, iteration
    std::vector<void*> owned;
};

namespace {

int feas_check_launch(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: a kernel launch failed: %s", where, hipGetErrorString(e)); return FOS_EHIP; }
    return FOS_OK;
}

int feas_alloc(fos_feas* h, double** out, size_t count) {
    void* q = nullptr;
    const hipError_t e = hipMalloc(&q, sizeof(double) * std::max<size_t>(count, 1));
    if (e != hipSuccess) { set_error("feasibility form: hipMalloc of %zu doubles failed: %s", count, hipGetErrorString(e)); return FOS_ENOMEM; }
    if (hipMemsetAsync(q, 0, sizeof(double) * std::max<size_t>(count, 1), h->stream) != hipSuccess) { (void)hipFree(q); set_error("hipMemsetAsync failed"); return FOS_EHIP; }
    h->owned.push_back(q);
    *out = static_cast<double*>(q);
    return FOS_OK;
}

#define FEAS_K(kernel, ...) hipLaunchKernelGGL(kernel, dim3(h->grid), dim3(FEAS_THREADS), 0, h->stream, __VA_ARGS__)

LaunchCtx feas_ctx(const fos_feas* h, int64_t l) {
    LaunchCtx c{};
    c.stream = h->stream;
    c.l = l;
    return c;
}

// the LongstepWrapper's kernels see the zero-padded vectors of L doubles as L / 2 double2
LaunchCtx feas_long_ctx(const fos_feas* h) {
    LaunchCtx c = feas_ctx(h, h->L / 2);
    c.vec_blocks = h->grid;
    return c;
}
inline void feas_long_save(fos_feas* h, int which, const double* y, const double* x) {
    if (h->lp.now) long_save_plane(feas_long_ctx(h), h->lp, which, reinterpret_cast<const double2*>(y), reinterpret_cast<const double2*>(x));
}

// prox!(y, S, x): the projection onto set `which`; y must not alias x
int feas_prox(fos_feas* h, int which, double* y, const double* x) {
    const FeasSet& s = h->S[which];
    if (s.kind == 1) {
        launch_dense_symv(feas_ctx(h, h->n), h->L, s.P, x, h->px);        // P is symmetric: column dots = P x
        FEAS_K(feas_affine_finish_kernel, h->n, y, x, h->px, s.q);
    } else if (s.kind == 2) {
        if (s.lov) FEAS_K(feas_boxv_kernel, h->n, y, x, (const double*)s.lov, (const double*)s.hiv);
        else FEAS_K(feas_box_kernel, h->n, y, x, s.lo, s.hi);
    } else if (s.kind == 3) {                                             // prox!(y, ::ConeProduct, x)   cones.jl:89-94
        FeasSet& ms = h->S[which];
        LaunchCtx c = feas_ctx(h, h->n);
        c.vec_blocks = h->grid; c.cus = h->cus; c.psd_attr_set = &ms.psd_attr_set;
        FEAS_K(feas_to_parts_kernel, h->n, h->zin, x);
        launch_cones_elementwise(c, h->zout, h->zin, s.ew_op);
        launch_cones_soc(c, h->zout, h->zin, s.soc, s.nsoc);
        launch_cones_exp(c, h->zout, h->zin, s.expc, s.nexp);
        FOS_TRY(launch_cones_psd(c, h->zout, h->zin, s.psd, s.npsd, s.psd_kmin, s.psd_kmax, s.psd_scratch, s.psd_V[s.psd_cur], s.psd_V[1 - s.psd_cur],
                                 s.psd_have_prev, nullptr, 0));
        FOS_TRY(launch_cones_psd_sign(c, s.psd_big, h->zout, h->zin));          // cones of order > 64 (psd_sign.hip)
        if (s.npsd > 0 && s.psd_V[0]) { ms.psd_cur = 1 - ms.psd_cur; ms.psd_have_prev = 1; }      // warm start of the next projection
        FEAS_K(feas_from_parts_kernel, h->n, y, (const double2*)h->zout);
    } else if (s.kind == 5) {                                             // IndAffine over a sparse A: CG on the normal equations, exact to the residual's rounding level
        FOS_TRY(sparse_affine_project(s.sa, h->stream, y, x));
    } else if (s.kind == 4) {                                             // any other ProximableFunction: the caller's prox! on host vectors
        FOS_HIP(hipMemcpyAsync(s.cb_x, x, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
        FOS_HIP(hipStreamSynchronize(h->stream));
        const int32_t rc = s.cb(s.cb_ctx, h->n, s.cb_x, s.cb_y);
        if (rc != 0) { set_error("feasibility form: the prox callback of set %d returned %d", which + 1, (int)rc); return FOS_EINVAL; }
        FOS_HIP(hipMemcpyAsync(y, s.cb_y, sizeof(double) * h->n, hipMemcpyHostToDevice, h->stream));
    } else { set_error("feasibility form: set %d has not been defined", which + 1); return FOS_EINVAL; }
    return FOS_OK;
}

// checkstatus(stat, z) at iteration i      FeasibilityStatus.jl:32-72
int feas_check(fos_feas* h, const double* z, int64_t i, int64_t checki, double eps, bool override_) {
    if (override_ || (checki > 0 && i % checki == 0)) {
        FEAS_K(feas_check_kernel, h->n, h->prev, z, 1, h->partials);                   // |prev - z|^2 and prev = z in one pass
        std::vector<double> part((size_t)h->grid);
        FOS_HIP(hipMemcpyAsync(part.data(), h->partials, sizeof(double) * h->grid, hipMemcpyDeviceToHost, h->stream));
        FOS_HIP(hipStreamSynchronize(h->stream));
        double s = 0.0;
        for (double v : part) s += v;
        h->err = std::sqrt(s);
        h->status = (h->err <= eps) ? FOS_STATUS_OPTIMAL : FOS_STATUS_CONTINUE;       // :57 (NaN <= eps is false)
        h->checked = 1;
    } else {
        h->checked = 0;
        FEAS_K(feas_check_kernel, h->n, h->prev, z, 0, h->partials);                   // :61,69: prev = z at every call
    }
    return FOS_OK;
}

int feas_norm_of_diff(fos_feas* h, const double* a, const double* b, double* out) {
    FEAS_K(feas_normdiff_kernel, h->n, a, b, h->partials);
    std::vector<double> part((size_t)h->grid);
    FOS_HIP(hipMemcpyAsync(part.data(), h->partials, sizeof(double) * h->grid, hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (double v : part) s += v;
    *out = std::sqrt(s);
    return FOS_OK;
}

// LineSearchWrapper(GAP / GAPA), one search iteration      wrappers/linesearch.jl:36-75
// scratch: y = tmp1 (the iterate the search starts from), xold = res, p = tmp3 -- unused by these two algorithms
int feas_linesearch_iteration(fos_feas* h, int64_t i, int64_t checki, double eps) {
    const int64_t n = h->n;
    const double* a12 = h->alg == FOS_ALG_GAPA ? h->a12 : nullptr;
    FEAS_K(feas_copy_kernel, n, h->y, (const double*)h->x);                             // tmp1 .= x                 :41
    FOS_TRY(feas_prox(h, 0, h->t1, h->x));                                             // S1!(tmp2, x)              :45
    FEAS_K(feas_relax_kernel, n, h->t1, (const double*)h->x, h->alpha1, a12);
    FOS_TRY(feas_prox(h, 1, h->x, h->t1));                                             // S2!(x, tmp2, status)      :46
    FOS_TRY(feas_check(h, h->x, i, checki, eps, false));
    FEAS_K(feas_relax_kernel, n, h->x, (const double*)h->t1, h->alpha2, a12);
    FEAS_K(feas_sub_kernel, n, h->xold, (const double*)h->x, (const double*)h->y);     // res .= x .- tmp1          :49
    FOS_TRY(feas_norm_of_diff(h, h->x, h->y, &h->ls_log[0]));                          // normres = norm(res)       :50
    double best = INFINITY, abest = 1.0, a = 0.1;                                      // :53-55
    for (int k = 0; k <= 30; ++k) {                                                    // :56
        a = a * 1.8;                                                                   // :57
        FEAS_K(feas_axpy_kernel, n, h->x, (const double*)h->y, a, (const double*)h->xold);      // x .= tmp1 .+ a.*res   :58
        FOS_TRY(feas_prox(h, 0, h->t1, h->x));                                         // S1!(tmp2, x, nostatus)    :60
        FEAS_K(feas_relax_kernel, n, h->t1, (const double*)h->x, h->alpha1, a12);
        FOS_TRY(feas_prox(h, 1, h->p, h->t1));                                         // S2!(tmp3, tmp2, nostatus) :61
        FEAS_K(feas_relax_kernel, n, h->p, (const double*)h->t1, h->alpha2, a12);
        double tr = 0.0;
        FOS_TRY(feas_norm_of_diff(h, h->x, h->p, &tr));                                // testres = normdiff(x, tmp3) :62
        h->ls_log[1 + k] = tr;
        if (tr < best) { best = tr; abest = a; }                                       // :64-67
    }
    FEAS_K(feas_axpy_kernel, n, h->x, (const double*)h->y, abest, (const double*)h->xold);      // x .= tmp1 .+ abest.*res :70
    h->ls_log[32] = abest;
    h->ls_log[33] = (double)i;
    return FOS_OK;
}

// GAPP, one iteration      solvers/gapproj.jl:29-72      scratch: xold = res, p = tmp3, q = tmp4
int feas_gapp_iteration(fos_feas* h, int64_t i, int64_t checki, double eps) {
    const int64_t n = h->n;
    FOS_TRY(feas_prox(h, 0, h->t1, h->x));                                             // prox!(tmp1, S1, x)        :33
    if (i % h->gapp_iproj != 0) {                                                      // the GAP step              :63-70
        FEAS_K(feas_relax_kernel, n, h->t1, (const double*)h->x, h->alpha1, (const double*)nullptr);
        FOS_TRY(feas_prox(h, 1, h->t2, h->t1));
        FOS_TRY(feas_check(h, h->t2, i, checki, eps, false));
        FEAS_K(feas_relax_kernel, n, h->t2, (const double*)h->t1, h->alpha2, (const double*)nullptr);
        FEAS_K(feas_combine_kernel, n, h->x, (const double*)h->t2, h->alpha);
        return FOS_OK;
    }
    FOS_TRY(feas_prox(h, 1, h->t2, h->t1));                                            // prox!(tmp2, S2, tmp1)     :39
    FOS_TRY(feas_prox(h, 0, h->xold, h->t2));                                          // prox!(res, S1, tmp2)      :40
    FEAS_K(feas_sub_kernel, n, h->xold, (const double*)h->xold, (const double*)h->t1); // res .= res - tmp1         :41
    double normbest = INFINITY, abest = -1.0;                                          // :44-45
    for (int k = 0; k <= 20; ++k) {                                                    // :46
        const double at = std::ldexp(1.0, k);                                          // 2.0^k
        FEAS_K(feas_axpy_kernel, n, h->p, (const double*)h->t1, at, (const double*)h->xold);       // tmp3 .= tmp1 .+ at.*res
        FOS_TRY(feas_prox(h, 1, h->q, h->p));                                          // prox!(tmp4, S2, tmp3)
        double nt = 0.0;
        FOS_TRY(feas_norm_of_diff(h, h->q, h->p, &nt));                                // norm(tmp4 - tmp3)
        h->gapp_log[k] = nt;
        if (nt < normbest) { abest = at; normbest = nt; }
    }
    h->gapp_log[21] = abest; h->gapp_log[22] = (double)i;
    FEAS_K(feas_axpy_kernel, n, h->p, (const double*)h->t1, abest, (const double*)h->xold);        // tmp1 .= tmp1 .+ abest.*res   :58
    FOS_TRY(feas_prox(h, 1, h->t2, h->p));                                             // prox!(tmp2, S2, tmp1)     :59
    FOS_TRY(feas_check(h, h->t2, i, checki, eps, false));                              // :60
    FEAS_K(feas_relax_kernel, n, h->t2, (const double*)h->p, h->alpha2, (const double*)nullptr);   // :61
    FEAS_K(feas_copy_kernel, n, h->x, (const double*)h->t2);                           // x .= tmp2                 :62
    return FOS_OK;
}

int feas_step_once(fos_feas* h, int64_t i, int64_t checki, double eps) {
    const int64_t n = h->n;
    if (h->alg == FOS_ALG_GAPP) return feas_gapp_iteration(h, i, checki, eps);
    if (h->ls_interval > 0 && i % h->ls_interval == 0) return feas_linesearch_iteration(h, i, checki, eps);      // linesearch.jl:39
    h->lp.now = false;
    if (h->lp.interval > 0) {                                                           // longstep.jl:44-49
        const int64_t savepos = (i - 1) % h->lp.interval - h->lp.interval + h->lp.nsave + 2;
        if (savepos > 0) h->lp.savepos = savepos;
        h->lp.now = h->lp.savepos > 0;
    }
    switch (h->alg) {
    case FOS_ALG_GAP:
    case FOS_ALG_GAPA: {
        const bool ad = h->alg == FOS_ALG_GAPA;
        const double* a12 = ad ? h->a12 : nullptr;
        FOS_TRY(feas_prox(h, 0, h->t1, h->x));                                         // S1!  gap.jl:42-51, gapa.jl:61-70
        feas_long_save(h, 0, h->t1, h->x);                                             // addprojeq(longstep, y, x)
        FEAS_K(feas_relax_kernel, n, h->t1, (const double*)h->x, h->alpha1, a12);
        FOS_TRY(feas_prox(h, 1, h->t2, h->t1));                                        // S2!  gap.jl:53-59, gapa.jl:72-78
        FOS_TRY(feas_check(h, h->t2, i, checki, eps, false));
        feas_long_save(h, 1, h->t2, h->t1);                                            // addprojineq(longstep, y, x)
        if (ad) {                                                                      // relaxation, the sums of gapa.jl:96 (relaxed t1, t2 and the old x) and x's update: one pass
            FEAS_K(feas_gap_tail_kernel<true>, n, h->x, h->t2, (const double*)h->t1, h->alpha2, a12, h->alpha, h->partials);
            hipLaunchKernelGGL(feas_alpha12_kernel, dim3(1), dim3(64), 0, h->stream, (const double*)h->partials, h->grid, h->beta, h->a12);    // gapa.jl:96-101
        } else {
            FEAS_K(feas_gap_tail_kernel<false>, n, h->x, h->t2, (const double*)h->t1, h->alpha2, a12, h->alpha, h->partials);                  // gap.jl:58,78
        }
        break;
    }
    case FOS_ALG_FISTA: {                                                              // fista.jl:28-48
        if (i == 1) FEAS_K(feas_copy_kernel, n, h->y, (const double*)h->x);
        FOS_TRY(feas_prox(h, 0, h->t1, h->y));
        feas_long_save(h, 0, h->t1, h->y);                                             // addprojeq(longstep, tmp1, y)      fista.jl:36
        FEAS_K(feas_relax_kernel, n, h->t1, (const double*)h->y, h->alpha, (const double*)nullptr);
        FEAS_K(feas_copy_kernel, n, h->xold, (const double*)h->x);
        FOS_TRY(feas_prox(h, 1, h->x, h->t1));
        FOS_TRY(feas_check(h, h->x, i, checki, eps, false));
        feas_long_save(h, 1, h->x, h->t1);                                             // addprojineq(longstep, x, tmp1)    fista.jl:42
        const double told = h->fista_t;
        h->fista_t = (1.0 + std::sqrt(1.0 + 4.0 * told * told)) / 2.0;
        FEAS_K(feas_extrap_kernel, n, h->y, (const double*)h->x, (const double*)h->xold, (told - 1.0) / h->fista_t);
        break;
    }
    case FOS_ALG_DYKSTRA: {                                                            // dykstra.jl:25-36
        FEAS_K(feas_add_kernel, n, h->tmp, (const double*)h->x, (const double*)h->p);
        FOS_TRY(feas_prox(h, 0, h->y, h->tmp));
        feas_long_save(h, 0, h->y, h->tmp);                                            // addprojeq(longstep, y, x .+ p)    dykstra.jl:30
        FEAS_K(feas_sub_kernel, n, h->p, (const double*)h->tmp, (const double*)h->y);
        FEAS_K(feas_add_kernel, n, h->tmp, (const double*)h->y, (const double*)h->q);
        FOS_TRY(feas_prox(h, 1, h->x, h->tmp));
        FOS_TRY(feas_check(h, h->x, i, checki, eps, false));
        feas_long_save(h, 1, h->x, h->tmp);                                            // addprojineq(longstep, x, y .+ q)  dykstra.jl:34
        FEAS_K(feas_sub_kernel, n, h->q, (const double*)h->tmp, (const double*)h->x);
        break;
    }
    default: set_error("feasibility form: unknown algorithm %d", h->alg); return FOS_EINVAL;
    }
    if (h->lp.now && h->lp.savepos == h->lp.nsave + 1) {                                // longstep.jl:53-58: projectonnormals!, x .= tmp
        FOS_TRY(long_project_planes(feas_long_ctx(h), h->lp, reinterpret_cast<double2*>(h->x), i));
        h->lp.savepos = -1;
    }
    h->lp.now = false;
    return FOS_OK;
}

}  // namespace

extern "C" {

int fos_feas_create(int64_t n, int32_t device, fos_feas_handle* out) {
    if (!out || n < 1) { set_error("fos_feas_create: n >= 1 and a handle pointer are required"); return FOS_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device is visible: the HIP path has no CPU fallback"); return FOS_ENODEVICE; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (0..%d)", device, ndev - 1); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(device));
    fos_feas* h = new fos_feas();
    h->device = device; h->n = n; h->L = (n + 63) / 64 * 64;
    h->grid = (int)std::min<int64_t>(FEAS_PARTS, (n + FEAS_THREADS - 1) / FEAS_THREADS);
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; set_error("hipStreamCreate failed"); return FOS_EHIP; }
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->cus = prop.multiProcessorCount; }
    int rc = FOS_OK;
    double** vecs[] = {&h->x, &h->t1, &h->t2, &h->y, &h->xold, &h->p, &h->q, &h->prev, &h->tmp, &h->px};
    for (double** v : vecs) { rc = feas_alloc(h, v, (size_t)h->L); if (rc != FOS_OK) break; }
    if (rc == FOS_OK) rc = feas_alloc(h, &h->partials, 3 * (size_t)FEAS_PARTS);
    if (rc == FOS_OK) rc = feas_alloc(h, &h->a12, 8);
    if (rc != FOS_OK) { fos_feas_destroy(h); return rc; }
    *out = h;
    return fos_feas_set_iterate(h, nullptr);
}

int fos_feas_destroy(fos_feas_handle h) {
    if (!h) return FOS_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* q : h->owned) (void)hipFree(q);
    for (FeasSet& s : h->S) { if (s.cb_x) (void)hipHostFree(s.cb_x); if (s.cb_y) (void)hipHostFree(s.cb_y); psd_sign_destroy(s.psd_big); sparse_affine_destroy(s.sa); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return FOS_OK;
}

// IndAffine(A, b): A is m x n, ROW-major (C order), full row rank.  One-time set-up on the device:
//   G = A A' (+ identity on the padding), X = G^-1 by Newton-Schulz  X <- 2 X - X (G X)  from X_0 = I / (1.25 |A|_F^2),
//   P = A' X A,  q = A' X b.
int fos_feas_set_affine(fos_feas_handle h, int32_t which, int64_t m, const double* A, const double* b) {
    if (!h || !A || !b || which < 1 || which > 2 || m < 1 || m > h->n) { set_error("fos_feas_set_affine: bad argument (1 <= m <= n, which = 1 | 2)"); return FOS_EINVAL; }
    if (h->n > 46000) { set_error("IndAffine: the dense projector holds %lld x %lld doubles: supported up to n = 46000", (long long)h->n, (long long)h->n); return FOS_EUNSUPPORTED; }
    FOS_HIP(hipSetDevice(h->device));
    const int64_t n = h->n, L = h->L;
    const size_t L2 = (size_t)L * (size_t)L;
    std::vector<double> bp((size_t)L, 0.0);
    double fro2 = 0.0;
    for (int64_t e = 0; e < m * n; ++e) {
        const double v = A[e];
        if (!(v == v) || std::fabs(v) > 1e300) { set_error("IndAffine: A has non-finite entries"); return FOS_EINVAL; }
        fro2 += v * v;
    }
    for (int64_t i = 0; i < m; ++i) bp[(size_t)i] = b[i];
    if (!(fro2 > 0.0)) { set_error("IndAffine: A is zero"); return FOS_EINVAL; }
    double *dA = nullptr, *dAt = nullptr, *G = nullptr, *B0 = nullptr, *B1 = nullptr, *B2 = nullptr, *db = nullptr, *dt = nullptr, *draw = nullptr;
    auto cleanup = [&]() { for (double* q : {dA, dAt, G, B0, B1, B2, db, dt, draw}) (void)hipFree(q); };
    hipError_t e = hipSuccess;
    for (double** q : {&dA, &dAt, &G, &B0, &B1, &B2}) if (e == hipSuccess) e = hipMalloc((void**)q, sizeof(double) * L2);
    for (double** q : {&db, &dt}) if (e == hipSuccess) e = hipMalloc((void**)q, sizeof(double) * (size_t)L);
    if (e == hipSuccess) e = hipMalloc((void**)&draw, sizeof(double) * (size_t)(m * n));
    if (e != hipSuccess) { cleanup(); set_error("IndAffine set-up: hipMalloc of six %lld x %lld buffers failed: %s", (long long)L, (long long)L, hipGetErrorString(e)); return FOS_ENOMEM; }
    auto fail = [&](int code) { cleanup(); return code; };
#define FEAS_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("IndAffine set-up: %s -> %s", #expr, hipGetErrorString(_e)); return fail(FOS_EHIP); } } while (0)
    FEAS_HIP(hipMemcpyAsync(draw, A, sizeof(double) * (size_t)(m * n), hipMemcpyHostToDevice, h->stream));
    FEAS_HIP(hipMemsetAsync(dA, 0, sizeof(double) * L2, h->stream));
    FEAS_HIP(hipMemsetAsync(dAt, 0, sizeof(double) * L2, h->stream));
    hipLaunchKernelGGL(feas_spread_kernel, dim3(1024), dim3(FEAS_THREADS), 0, h->stream, m, n, L, (const double*)draw, dA, dAt);
    FEAS_HIP(hipMemcpyAsync(db, bp.data(), sizeof(double) * (size_t)L, hipMemcpyHostToDevice, h->stream));
    FEAS_HIP(hipMemsetAsync(B0, 0, sizeof(double) * L2, h->stream));
    FEAS_HIP(hipMemsetAsync(B1, 0, sizeof(double) * L2, h->stream));
    const LaunchCtx c = feas_ctx(h, L);
    FEAS_K(feas_pad_identity_kernel, L, m, B0);                                        // E
    launch_dense_gemm(c, (int)L, 1.0, dA, dAt, 1.0, B0, G);                            // G = A A' + E   (block diagonal, positive definite)
    launch_dense_scale_identity(c, L, B1, 1.0 / (1.25 * std::max(fro2, 1.0)));         // X_0: lambda_max(G) <= max(|A|_F^2, 1)
    double *X = B1, *Xn = B2;
    std::vector<double> part(256);
    double resid = 1.0, prev_resid = 2.0;
    int it = 0;
    for (; it < 120; ++it) {
        launch_dense_gemm(c, (int)L, 1.0, G, X, 0.0, nullptr, B0);                     // Y = G X
        launch_dense_resid(c, L, B0, h->partials, 256);
        FEAS_HIP(hipMemcpyAsync(part.data(), h->partials, sizeof(double) * 256, hipMemcpyDeviceToHost, h->stream));
        FEAS_HIP(hipStreamSynchronize(h->stream));
        resid = 0.0;
        for (double r : part) resid = (r > resid || r != r) ? r : resid;
        if (resid != resid) { set_error("IndAffine set-up: the inverse of A A' produced NaN"); return fail(FOS_EINVAL); }
        if (resid <= 2e-14 || (resid <= 1e-10 && resid >= 0.5 * prev_resid)) break;   // rounding level: the residual no longer squares
        prev_resid = resid;
        launch_dense_gemm(c, (int)L, -1.0, X, B0, 2.0, X, Xn);                         // X <- 2 X - X Y
        std::swap(X, Xn);
    }
    if (!(resid <= 1e-10)) {
        set_error("IndAffine set-up: the inverse of A A' did not converge (max |G X - I| = %.3e after %d Newton-Schulz steps): A needs full row rank and a moderate condition number", resid, it);
        return fail(FOS_EINVAL);
    }
    FeasSet& s = h->S[which - 1];
    int rc = FOS_OK;
    if (!s.P) rc = feas_alloc(h, &s.P, L2);
    if (rc == FOS_OK && !s.q) rc = feas_alloc(h, &s.q, (size_t)L);
    if (rc != FOS_OK) return fail(rc);
    launch_dense_gemm(c, (int)L, 1.0, X, dA, 0.0, nullptr, B0);                        // B0 = X A
    launch_dense_gemm(c, (int)L, 1.0, dAt, B0, 0.0, nullptr, s.P);                     // P = A' X A
    launch_dense_symv(c, L, X, db, dt);                                                // t = X b   (X symmetric)
    launch_dense_symv(c, L, dA, dt, s.q);                                              // q[col] = A[:, col] . t = (A' t)[col]
    FEAS_HIP(hipStreamSynchronize(h->stream));
#undef FEAS_HIP
    rc = feas_check_launch("IndAffine set-up");
    cleanup();
    if (rc != FOS_OK) return rc;
    s.kind = 1; s.ns_iters = it; s.ns_resid = resid;
    return FOS_OK;
}

// IndAffine(A, b), A a sparse m x n matrix (CSC, 1-based: Julia's SparseMatrixCSC{Float64,Int64}), full row rank: nothing dense is formed, any n
int fos_feas_set_affine_sparse(fos_feas_handle h, int32_t which, int64_t m, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b) {
    if (!h || which < 1 || which > 2 || m < 1 || m > h->n) { set_error("fos_feas_set_affine_sparse: bad argument (1 <= m <= n, which = 1 | 2)"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    SparseAffine* sa = nullptr;
    FOS_TRY(sparse_affine_setup(m, h->n, colptr, rowval, nzval, b, h->cus, &sa));
    FeasSet& s = h->S[which - 1];
    sparse_affine_destroy(s.sa);
    s.sa = sa; s.kind = 5;
    return FOS_OK;
}

// out8: projections so far, CG iterations so far, CG iterations / restarts of the last projection, its |A y - b| and the rounding level | |A||y| + |b| |
// (rows scaled to unit norm), stored entries, lanes per row of A x 1000 + of A'
int fos_feas_affine_stats(fos_feas_handle h, int32_t which, double* out8) {
    if (!h || !out8 || which < 1 || which > 2 || h->S[which - 1].kind != 5) { set_error("fos_feas_affine_stats: set %d is not a sparse IndAffine", (int)which); return FOS_EINVAL; }
    sparse_affine_stats(h->S[which - 1].sa, out8);
    return FOS_OK;
}

int fos_feas_set_box(fos_feas_handle h, int32_t which, double lo, double hi) {
    if (!h || which < 1 || which > 2 || !(lo <= hi)) { set_error("fos_feas_set_box: which = 1 | 2 and lo <= hi are required"); return FOS_EINVAL; }
    FeasSet& s = h->S[which - 1];
    s.kind = 2; s.lo = lo; s.hi = hi; s.lov = s.hiv = nullptr;
    return FOS_OK;
}

// IndBox(lo, hi) with array bounds (n entries each; +-INFINITY allowed)
int fos_feas_set_box_arrays(fos_feas_handle h, int32_t which, const double* lo, const double* hi) {
    if (!h || which < 1 || which > 2 || !lo || !hi) { set_error("fos_feas_set_box_arrays: bad argument"); return FOS_EINVAL; }
    for (int64_t i = 0; i < h->n; ++i) if (!(lo[i] <= hi[i])) { set_error("IndBox: lo[%lld] <= hi[%lld] is required", (long long)i + 1, (long long)i + 1); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FeasSet& s = h->S[which - 1];
    if (!s.lov) { FOS_TRY(feas_alloc(h, &s.lov, (size_t)h->n)); FOS_TRY(feas_alloc(h, &s.hiv, (size_t)h->n)); }
    FOS_HIP(hipMemcpyAsync(s.lov, lo, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    FOS_HIP(hipMemcpyAsync(s.hiv, hi, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    s.kind = 2;
    return FOS_OK;
}

// ConeProduct (cones.jl:31-77): ncones cones in order, contiguous from index 1, together covering all n entries
int fos_feas_set_cones(fos_feas_handle h, int32_t which, int64_t ncones, const int32_t* type, const int64_t* len) {
    if (!h || which < 1 || which > 2 || ncones < 1 || !type || !len) { set_error("fos_feas_set_cones: bad argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    const int64_t n = h->n;
    std::vector<uint8_t> ew((size_t)n, 0);
    std::vector<ConeDesc> soc, psd, expc;
    int64_t pos = 0;
    for (int64_t i = 0; i < ncones; ++i) {
        if (len[i] < 1 || pos + len[i] > n) { set_error("ConeProduct: cone %lld (length %lld) does not fit the %lld entries (cones.jl:66-72: contiguous, gap-free ranges)", (long long)i + 1, (long long)len[i], (long long)n); return FOS_EINVAL; }
        uint8_t prim = 0;
        bool ewise = true;
        switch (type[i]) {
            case FOS_CONE_FREE: prim = EW_COPY; break;
            case FOS_CONE_ZERO: prim = EW_ZERO; break;
            case FOS_CONE_NONNEG: prim = EW_MAX0; break;
            case FOS_CONE_NONPOS: prim = EW_MIN0; break;
            case FOS_CONE_SOC: case FOS_CONE_SOCROT: case FOS_CONE_SDP: case FOS_CONE_EXPPRIMAL: case FOS_CONE_EXPDUAL: ewise = false; break;
            default: set_error("ConeProduct: unknown cone code %d", type[i]); return FOS_EINVAL;
        }
        if (ewise) { for (int64_t k = 0; k < len[i]; ++k) ew[(size_t)(pos + k)] = (uint8_t)(prim | (EW_COPY << 2)); }
        else {
            ConeDesc cd;
            cd.start = pos; cd.len = (int32_t)len[i]; cd.type = type[i]; cd.dual_part = 1; cd.k = 0;      // part 1 (the vector): primal
            if (type[i] == FOS_CONE_SDP) {
                const int64_t k = (int64_t)std::floor((std::sqrt(8.0 * (double)len[i] + 1.0) - 1.0) / 2.0 + 0.5);
                if (k * (k + 1) / 2 != len[i]) { set_error("ConeProduct cone %lld: SDP length %lld is not k(k+1)/2", (long long)i + 1, (long long)len[i]); return FOS_EINVAL; }
                cd.k = (int32_t)k;
                psd.push_back(cd);
            } else if (type[i] == FOS_CONE_EXPPRIMAL || type[i] == FOS_CONE_EXPDUAL) {
                if (len[i] != 3) { set_error("ConeProduct cone %lld: an exponential cone has exactly 3 entries (got %lld)", (long long)i + 1, (long long)len[i]); return FOS_EINVAL; }
                expc.push_back(cd);
            } else {
                if (type[i] == FOS_CONE_SOCROT && len[i] < 2) { set_error("ConeProduct cone %lld: rotated SOC needs >= 2 entries", (long long)i + 1); return FOS_EINVAL; }
                soc.push_back(cd);
            }
            for (int64_t k = 0; k < len[i]; ++k) ew[(size_t)(pos + k)] = EW_SKIP;
        }
        pos += len[i];
    }
    if (pos != n) { set_error("ConeProduct: the cones cover %lld of the %lld entries", (long long)pos, (long long)n); return FOS_EINVAL; }
    FeasSet& s = h->S[which - 1];
    auto upload = [&](auto** dst, const auto& v) -> int {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (v.empty()) { *dst = nullptr; return FOS_OK; }
        void* q = nullptr;
        if (hipMalloc(&q, sizeof(T) * v.size()) != hipSuccess) { set_error("ConeProduct: hipMalloc failed"); return FOS_ENOMEM; }
        h->owned.push_back(q);
        if (hipMemcpy(q, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) { set_error("ConeProduct: upload failed"); return FOS_EHIP; }
        *dst = static_cast<T*>(q);
        return FOS_OK;
    };
    FOS_TRY(upload(&s.ew_op, ew));
    FOS_TRY(upload(&s.soc, soc));
    FOS_TRY(upload(&s.expc, expc));
    psd_sign_destroy(s.psd_big); s.psd_big = nullptr;
    FOS_TRY(psd_sign_setup(psd, &s.psd_big));
    FOS_TRY(upload(&s.psd, psd));
    s.nsoc = (int)soc.size(); s.nexp = (int)expc.size(); s.npsd = (int)psd.size();
    s.psd_kmax = 0;
    for (auto& cd : psd) s.psd_kmax = std::max(s.psd_kmax, cd.k);
    s.psd_kmin = s.psd_kmax;
    for (auto& cd : psd) s.psd_kmin = std::min(s.psd_kmin, cd.k);
    const size_t sb = psd_scratch_bytes(s.psd_kmax, s.npsd);
    if (sb) FOS_TRY(feas_alloc(h, &s.psd_scratch, sb / sizeof(double)));
    const size_t vb = psd_basis_doubles(s.psd_kmax, s.npsd);
    if (vb) { FOS_TRY(feas_alloc(h, &s.psd_V[0], vb)); FOS_TRY(feas_alloc(h, &s.psd_V[1], vb)); }
    s.psd_cur = 0; s.psd_have_prev = 0;
    if (!h->zin) {
        double* q = nullptr;
        FOS_TRY(feas_alloc(h, &q, 2 * (size_t)n)); h->zin = reinterpret_cast<double2*>(q);
        FOS_TRY(feas_alloc(h, &q, 2 * (size_t)n)); h->zout = reinterpret_cast<double2*>(q);
    }
    FOS_HIP(hipStreamSynchronize(h->stream));
    s.kind = 3;
    return FOS_OK;
}

int fos_feas_set_callback(fos_feas_handle h, int32_t which, fos_prox_fn fn, void* ctx) {
    if (!h || (which != 1 && which != 2) || !fn) { set_error("fos_feas_set_callback: NULL handle / callback or which not in {1, 2}"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FeasSet& s = h->S[which - 1];
    if (!s.cb_x) {
        void *px = nullptr, *py = nullptr;
        if (hipHostMalloc(&px, sizeof(double) * std::max<int64_t>(h->n, 1)) != hipSuccess || hipHostMalloc(&py, sizeof(double) * std::max<int64_t>(h->n, 1)) != hipSuccess) {
            if (px) (void)hipHostFree(px);
            set_error("fos_feas_set_callback: pinned host buffers of %lld doubles could not be allocated", (long long)h->n);
            return FOS_ENOMEM;
        }
        s.cb_x = static_cast<double*>(px); s.cb_y = static_cast<double*>(py);
    }
    s.cb = fn; s.cb_ctx = ctx;
    s.kind = 4;
    return FOS_OK;
}

int fos_feas_set_alg(fos_feas_handle h, int32_t alg, double alpha, double alpha1, double alpha2, double beta) {
    if (!h || alg < FOS_ALG_GAP || alg > FOS_ALG_DYKSTRA) { set_error("fos_feas_set_alg: unknown algorithm"); return FOS_EINVAL; }
    h->alg = alg; h->alpha = alpha; h->alpha1 = alpha1; h->alpha2 = alpha2; h->beta = beta;
    if (alg == FOS_ALG_GAPA) h->alpha1 = h->alpha2 = 2.0;              // (unused: GAPA relaxes with the device scalar alpha12)
    h->ls_interval = 0;                                                // a fresh algorithm is unwrapped (fos_feas_set_linesearch follows)
    h->lp.interval = 0; h->lp.savepos = 0; h->lp.now = false;          // ... (fos_feas_set_longstep follows)
    return FOS_OK;
}

// GAPP(alpha, alpha1, alpha2; iproj)      solvers/gapproj.jl:5-13 (Feasibility form only)
int fos_feas_set_gapp(fos_feas_handle h, double alpha, double alpha1, double alpha2, int64_t iproj) {
    if (!h || iproj < 1) { set_error("fos_feas_set_gapp: iproj >= 1 is required"); return FOS_EINVAL; }
    h->alg = FOS_ALG_GAPP; h->alpha = alpha; h->alpha1 = alpha1; h->alpha2 = alpha2; h->beta = 0.0; h->gapp_iproj = iproj;
    h->ls_interval = 0;
    h->lp.interval = 0; h->lp.savepos = 0; h->lp.now = false;          // a fresh algorithm is unwrapped (support_longstep(::GAPP) = false, gapproj.jl:83)
    return FOS_OK;
}
int fos_feas_gapp_log(fos_feas_handle h, double* out23) {
    if (!h || !out23) { set_error("NULL argument"); return FOS_EINVAL; }
    memcpy(out23, h->gapp_log, sizeof(h->gapp_log));
    return FOS_OK;
}

int fos_feas_set_linesearch(fos_feas_handle h, int64_t lsinterval) {
    if (!h || lsinterval < 0) { set_error("bad argument"); return FOS_EINVAL; }
    if (lsinterval > 0 && h->alg != FOS_ALG_GAP && h->alg != FOS_ALG_GAPA) {
        set_error("this algorithm does not support line search (support_linesearch: GAP and GAPA only, solvers/defaults.jl:22)");
        return FOS_EUNSUPPORTED;
    }
    // (a search iteration returns before the planes of a saving window are written: the two wrappers exclude each other, both ways)
    if (lsinterval > 0 && h->lp.interval > 0) { set_error("LineSearchWrapper inside a LongstepWrapper is not supported (fos_feas_set_longstep(h, 0, 0) first)"); return FOS_EUNSUPPORTED; }
    h->ls_interval = lsinterval;
    return FOS_OK;
}
// LongstepWrapper on the Feasibility form (as fos_set_longstep)
int fos_feas_set_longstep(fos_feas_handle h, int64_t longinterval, int64_t nsave) {
    if (!h || longinterval < 0 || nsave < 0) { set_error("bad argument"); return FOS_EINVAL; }
    if (longinterval == 0) { h->lp.interval = 0; h->lp.savepos = 0; h->lp.now = false; return FOS_OK; }
    if (h->alg == FOS_ALG_GAPP) { set_error("Algorithm alg does not support longstep (support_longstep(::GAPP) = false, gapproj.jl:83)"); return FOS_EUNSUPPORTED; }
    if (2 * (nsave + 1) > LONG_KMAX_ROWS) { set_error("LongstepWrapper: nsave <= %d", LONG_KMAX_ROWS / 2 - 1); return FOS_EUNSUPPORTED; }
    if (longinterval < nsave + 1) { set_error("LongstepWrapper: longinterval must be at least nsave + 1"); return FOS_EINVAL; }
    if (h->ls_interval > 0) { set_error("LongstepWrapper around a LineSearchWrapper is not supported"); return FOS_EUNSUPPORTED; }
    FOS_HIP(hipSetDevice(h->device));
    const int64_t K = 2 * (nsave + 1);
    if (!h->lp.P || h->lp.nsave != nsave) {
        FOS_HIP(hipStreamSynchronize(h->stream));
        for (void* old : {(void*)h->lp.P, (void*)h->lp.bpart, (void*)h->lp.dots, (void*)h->lp.nu}) {       // replaced, not piled up (feas_alloc zeroes the new ones)
            if (!old) continue;
            auto it = std::find(h->owned.begin(), h->owned.end(), old);
            if (it != h->owned.end()) h->owned.erase(it);
            (void)hipFree(old);
        }
        h->lp.P = nullptr; h->lp.bpart = nullptr; h->lp.dots = nullptr; h->lp.nu = nullptr;
        double* q = nullptr;
        FOS_TRY(feas_alloc(h, &q, (size_t)K * h->L)); h->lp.P = reinterpret_cast<double2*>(q);
        FOS_TRY(feas_alloc(h, &h->lp.bpart, (size_t)K * h->grid));
        FOS_TRY(feas_alloc(h, &h->lp.dots, (size_t)h->grid * 2 * (LONG_KMAX_ROWS + 1)));
        FOS_TRY(feas_alloc(h, &h->lp.nu, (size_t)2 * K));
    }
    h->lp.interval = longinterval; h->lp.nsave = nsave; h->lp.savepos = 0; h->lp.now = false;
    h->lp.max_supports = getenv("FOS_LONG_MAX_SUPPORTS") ? atoll(getenv("FOS_LONG_MAX_SUPPORTS")) : 4096;
    h->lp.log[7] = 0.0;
    return FOS_OK;
}
int fos_feas_longstep_log(fos_feas_handle h, double* out8) {
    if (!h || !out8) { set_error("NULL argument"); return FOS_EINVAL; }
    memcpy(out8, h->lp.log, sizeof(h->lp.log));
    return FOS_OK;
}
int fos_feas_linesearch_log(fos_feas_handle h, double* out34) {
    if (!h || !out34) { set_error("NULL argument"); return FOS_EINVAL; }
    memcpy(out34, h->ls_log, sizeof(h->ls_log));
    return FOS_OK;
}

// x = x0 (NULL: zeros(n), Feasibility.jl:58) and the state of a fresh init_algorithm!: alpha12 = 2 (gapa.jl:29), t = 1, y = xold = 0
// (fista.jl:22-24), p = q = 0 (dykstra.jl:21-22), prev = NaN (Feasibility.jl:78)
int fos_feas_set_iterate(fos_feas_handle h, const double* x0) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    const size_t bytes = sizeof(double) * (size_t)h->L;
    for (double* v : {h->x, h->t1, h->t2, h->y, h->xold, h->p, h->q, h->tmp, h->px}) FOS_HIP(hipMemsetAsync(v, 0, bytes, h->stream));
    if (x0) FOS_HIP(hipMemcpyAsync(h->x, x0, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    FEAS_K(feas_fill_kernel, h->n, h->prev, (double)NAN);
    const double two = 2.0;
    FOS_HIP(hipMemcpyAsync(h->a12, &two, sizeof(double), hipMemcpyHostToDevice, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    h->fista_t = 1.0; h->status = FOS_STATUS_CONTINUE; h->checked = 0; h->err = NAN;
    for (FeasSet& s : h->S) { s.psd_cur = 0; s.psd_have_prev = 0; sparse_affine_reset(s.sa, h->stream); }        // a new solve starts its PSD projections cold, the sparse IndAffine's multipliers at zero
    return feas_check_launch("fos_feas_set_iterate");
}

// iterations first_iter .. first_iter + niter - 1 of `iterate` (solverwrapper.jl:23-29): stops behind the iteration whose check
// found err <= eps.  done = iterations run; status / err / checked describe the LAST iteration (FeasibilityStatus fields).
int fos_feas_step(fos_feas_handle h, int64_t first_iter, int64_t niter, int64_t checki, double eps, int64_t* done, int32_t* status, double* err,
                  int32_t* checked) {
    if (!h || first_iter < 1 || niter < 0) { set_error("fos_feas_step: bad argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    int64_t k = 0;
    for (; k < niter; ++k) {
        FOS_TRY(feas_step_once(h, first_iter + k, checki, eps));
        if (h->status != FOS_STATUS_CONTINUE) { ++k; break; }
    }
    FOS_HIP(hipStreamSynchronize(h->stream));
    if (done) *done = k;
    if (status) *status = h->status;
    if (err) *err = h->err;
    if (checked) *checked = h->checked;
    return feas_check_launch("fos_feas_step");
}

// getsol (gap.jl:82-87, gapa.jl:107-112, fista.jl:50-56, dykstra.jl:38-44): P_S2(P_S1(x)); force_check = the `checkstatus(...,
// override = true)` of solverwrapper.jl:31-33 on that point
int fos_feas_getsol(fos_feas_handle h, double* sol, int32_t force_check, double eps, int32_t* status, double* err) {
    if (!h || !sol) { set_error("fos_feas_getsol: NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(feas_prox(h, 0, h->t1, h->x));
    FOS_TRY(feas_prox(h, 1, h->tmp, h->t1));
    if (force_check) FOS_TRY(feas_check(h, h->tmp, 0, 0, eps, true));
    FOS_HIP(hipMemcpyAsync(sol, h->tmp, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    if (status) *status = h->status;
    if (err) *err = h->err;
    return feas_check_launch("fos_feas_getsol");
}

int fos_feas_get_iterate(fos_feas_handle h, double* x) {
    if (!h || !x) { set_error("fos_feas_get_iterate: NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_HIP(hipMemcpyAsync(x, h->x, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

// prox!(y, S_which, x) on host vectors (tests; the reference's prox! protocol)
int fos_feas_prox(fos_feas_handle h, int32_t which, const double* x, double* y) {
    if (!h || !x || !y || which < 1 || which > 2) { set_error("fos_feas_prox: bad argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_HIP(hipMemcpyAsync(h->tmp, x, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    FOS_TRY(feas_prox(h, which - 1, h->t2, h->tmp));
    FOS_HIP(hipMemcpyAsync(y, h->t2, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return feas_check_launch("fos_feas_prox");
}

// diagnostics: alpha12 (GAPA), Newton-Schulz steps / final residual of an IndAffine set-up (0 when the set is not affine)
int fos_feas_info(fos_feas_handle h, double* alpha12, int32_t* ns_iters, double* ns_resid) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (alpha12) { FOS_HIP(hipMemcpyAsync(alpha12, h->a12, sizeof(double), hipMemcpyDeviceToHost, h->stream)); FOS_HIP(hipStreamSynchronize(h->stream)); }
    for (int k = 0; k < 2; ++k) {
        if (ns_iters) ns_iters[k] = h->S[k].ns_iters;
        if (ns_resid) ns_resid[k] = h->S[k].ns_resid;
    }
    return FOS_OK;
}

}  // extern "C"
