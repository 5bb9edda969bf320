/*
 * fos_cport.c -- plain-C restatement of the FirstOrderSolvers.jl hot path, used as the timed CPU baseline.
 *
 * *** TEST / MEASUREMENT INFRASTRUCTURE ONLY. ***  Like oracle/fos_oracle.py (the parity checker this file is itself
 * checked against in tests/test_cport.py), it may be used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg only.  The product (firstordersolvers.jl_amd) never links or calls it.
 *
 * It mirrors the reference's execution model (SURVEY.md section 8(d) "CPU reference timing"):
 *   - Julia SparseMatrixCSC, Int64 indices: A*x is a column SCATTER, A'*x a column GATHER   (SparseArrays.mul!,
 *     call sites src/problemforms/HSDE/HSDEAffine.jl:51-52)
 *   - four sweeps of A per KKT apply, unfused vector passes   (src/utilities/affinepluslinear.jl:37-52)
 *   - textbook CG with the reference's stopping rule          (src/utilities/conjugategradients.jl:31-55)
 *   - serial loop over the cones, one symmetric eigen-decomposition per PSD cone   (src/cones.jl:80-142)
 * threads = 1 reproduces that single-threaded model.  threads > 1 is the "all-core" variant: rows of A (a CSR copy) and
 * columns of A, vector passes and cones are divided among OpenMP threads -- what a tuned multi-core CPU code would do; the
 * arithmetic per entry is the same.
 *
 * The symmetric eigen-solver (Householder tridiagonalisation + implicit QL, the classical EISPACK tred2/tql2 scheme)
 * stands in for LAPACK's dsyev that ProximalOperators' IndPSD reaches through Julia's `eigen`.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { C_FREE = 0, C_ZERO = 1, C_NONNEG = 2, C_NONPOS = 3, C_SOC = 4, C_SOCROT = 5, C_SDP = 6 };

typedef struct {
    int64_t m, n, l, N, nnz;
    /* CSC (0-based) as given; CSR copy for the multi-threaded A*x */
    int64_t *colptr, *rowval; double *nzval;
    int64_t *rowptr, *colidx; double *rval;
    double *b, *c;
    int64_t nK1, nK2; int32_t *K1type, *K2type; int64_t *K1start, *K1len, *K2start, *K2len;
    /* AffinePlusLinear / CGdata  (affinepluslinear.jl:58-69, conjugategradients.jl:1-11) */
    double *rhs, *r, *p, *z, *xinit; int firstrun; int64_t i, cgiter;
    double *tmp1, *tmp2, *qtmp, *negx; /* GAPData.tmp1/tmp2, scratch for Q applies, -x of proxDual! (cones.jl:81) */
    double *eigA, *eigd, *eige;        /* per-thread eigen workspaces, kmax*kmax + 2 kmax each */
    int kmax, threads;
} fosc;

static int nthreads(const fosc* s) { return s->threads > 0 ? s->threads : 1; }

/* ------------------------------------------------------------------ SpMV   HSDEAffine.jl:51-52 */
static void spmv_A(const fosc* s, double* y, const double* x) {           /* y = A x */
    if (nthreads(s) == 1) {                                             /* CSC scatter, as SparseArrays.mul! */
        memset(y, 0, sizeof(double) * (size_t)s->m);
        for (int64_t j = 0; j < s->n; ++j) {
            const double xj = x[j];
            for (int64_t k = s->colptr[j]; k < s->colptr[j + 1]; ++k) y[s->rowval[k]] += s->nzval[k] * xj;
        }
    } else {
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
        for (int64_t i = 0; i < s->m; ++i) {
            double acc = 0.0;
            for (int64_t k = s->rowptr[i]; k < s->rowptr[i + 1]; ++k) acc += s->rval[k] * x[s->colidx[k]];
            y[i] = acc;
        }
    }
}
static void spmv_At(const fosc* s, double* y, const double* x) {          /* y = A' x : column gather */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t j = 0; j < s->n; ++j) {
        double acc = 0.0;
        for (int64_t k = s->colptr[j]; k < s->colptr[j + 1]; ++k) acc += s->nzval[k] * x[s->rowval[k]];
        y[j] = acc;
    }
}
static double dotp(const fosc* s, const double* a, const double* b, int64_t len) {
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc) num_threads(nthreads(s))
    for (int64_t i = 0; i < len; ++i) acc += a[i] * b[i];
    return acc;
}

/* mul!(Y, Q, B)   HSDEAffine.jl:41-59 ; sign = -1 gives the transpose (Q' = -Q, :61-65) */
static void q_mul(const fosc* s, double* Y, const double* B, double sign) {
    const int64_t n = s->n, m = s->m;
    const double b3 = B[n + m];
    spmv_At(s, Y, B + n);                                                /* :51 */
    spmv_A(s, Y + n, B);                                                 /* :52 */
    const double last = -dotp(s, s->c, B, n) - dotp(s, s->b, B + n, m);  /* :57 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t j = 0; j < n; ++j) Y[j] = sign * (Y[j] + b3 * s->c[j]);            /* :54 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < m; ++i) Y[n + i] = sign * (-(Y[n + i] - b3 * s->b[i]));  /* :55-56 */
    Y[n + m] = sign * last;
}
/* mul!(y, KKTMatrix(Q), x)   affinepluslinear.jl:37-49 : y1 = Q'x2 + x1 ; y2 = Q x1 - x2 */
void fosc_kkt_mul(fosc* s, double* y, const double* x) {
    const int64_t l = s->l;
    q_mul(s, y, x + l, -1.0);                                            /* :45 */
    q_mul(s, y + l, x, 1.0);                                             /* :47 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < l; ++i) { y[i] += x[i]; y[l + i] -= x[l + i]; }   /* :46,:48 */
}

/* conjugategradient!   conjugategradients.jl:31-55 */
static int64_t cg(fosc* s, double* x, const double* b, double tol, int64_t max_iters) {
    const int64_t N = s->N;
    double *r = s->r, *p = s->p, *Ap = s->z;
    fosc_kkt_mul(s, Ap, x);                                              /* :32 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < N; ++i) { r[i] = b[i] - Ap[i]; p[i] = r[i]; }    /* :33-34 */
    double rn = dotp(s, r, r, N);                                        /* :35 */
    int64_t it = 1;
    for (;;) {
        fosc_kkt_mul(s, Ap, p);                                          /* :38 */
        const double alpha = rn / dotp(s, Ap, p, N);                     /* :39 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
        for (int64_t i = 0; i < N; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; }   /* :40-41 */
        const double rr = dotp(s, r, r, N);
        if (sqrt(rr) <= tol || it >= max_iters) break;                   /* :42 */
        const double beta = rr / rn;                                     /* :45-47 */
        rn = rr;
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
        for (int64_t i = 0; i < N; ++i) p[i] = p[i] * beta + r[i];       /* :49-50 */
        it += 1;
    }
    return it;
}

/* prox!(y, S1::AffinePlusLinear, x) with q = 0, b = 0, beta = 1 (HSDE.jl:22)   affinepluslinear.jl:83-126 */
int fosc_prox_affine(fosc* s, double* y, const double* x) {
    const int64_t l = s->l, N = s->N;
    q_mul(s, s->rhs, x + l, -1.0);                                       /* :94  rhs1 = Q' x2 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < l; ++i) { s->rhs[i] += x[i]; s->rhs[l + i] = 0.0; }   /* :95 ; rhs2 = b = 0 */
    if (s->firstrun) { memcpy(s->xinit, x, sizeof(double) * (size_t)N); s->firstrun = 0; }   /* :101-104 */
    memcpy(y, s->xinit, sizeof(double) * (size_t)N);                     /* :106 */
    const double eps = 2.220446049250313e-16;
    double tol = pow(0.2, sqrt((double)s->i));                           /* :108-112 */
    if (tol < (double)l * eps) tol = (double)l * eps;
    s->i += 1;                                                           /* :114 */
    s->cgiter = cg(s, y, s->rhs, tol, 1000);                             /* :115-121 */
    memcpy(s->xinit, y, sizeof(double) * (size_t)N);                     /* :122 */
    return 0;
}

/* ------------------------------------------------------------------ symmetric eigen-decomposition (tred2 + tql2) */
static void tred2(int k, double* V, double* d, double* e) {             /* V: k x k row-major symmetric in, Q out */
    for (int j = 0; j < k; ++j) d[j] = V[(k - 1) * k + j];
    for (int i = k - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int q = 0; q < i; ++q) scale += fabs(d[q]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = V[(i - 1) * k + j]; V[i * k + j] = 0.0; V[j * k + i] = 0.0; }
        } else {
            for (int q = 0; q < i; ++q) { d[q] /= scale; h += d[q] * d[q]; }
            double f = d[i - 1], g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                V[j * k + i] = f;
                g = e[j] + V[j * k + j] * f;
                for (int q = j + 1; q <= i - 1; ++q) { g += V[q * k + j] * d[q]; e[q] += V[q * k + j] * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int q = j; q <= i - 1; ++q) V[q * k + j] -= (f * e[q] + g * d[q]);
                d[j] = V[(i - 1) * k + j];
                V[i * k + j] = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < k - 1; ++i) {
        V[(k - 1) * k + i] = V[i * k + i];
        V[i * k + i] = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int q = 0; q <= i; ++q) d[q] = V[q * k + i + 1] / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int q = 0; q <= i; ++q) g += V[q * k + i + 1] * V[q * k + j];
                for (int q = 0; q <= i; ++q) V[q * k + j] -= g * d[q];
            }
        }
        for (int q = 0; q <= i; ++q) V[q * k + i + 1] = 0.0;
    }
    for (int j = 0; j < k; ++j) { d[j] = V[(k - 1) * k + j]; V[(k - 1) * k + j] = 0.0; }
    V[(k - 1) * k + k - 1] = 1.0;
    e[0] = 0.0;
}
static void tql2(int k, double* V, double* d, double* e) {
    for (int i = 1; i < k; ++i) e[i - 1] = e[i];
    e[k - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < k; ++l) {
        const double t = fabs(d[l]) + fabs(e[l]);
        if (t > tst1) tst1 = t;
        int m = l;
        while (m < k) { if (fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m > l) {
            int iter = 0;
            do {
                ++iter;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < k; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                const double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int q = 0; q < k; ++q) {
                        h = V[q * k + i + 1];
                        V[q * k + i + 1] = s * V[q * k + i] + c * h;
                        V[q * k + i] = c * V[q * k + i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1 && iter < 200);
        }
        d[l] += f;
        e[l] = 0.0;
    }
}

/* IndPSD(scaling=true), vector method (see oracle/fos_oracle.py prox_psd_scaled): packed lower triangle, column-major */
static void prox_psd(double* y, const double* x, int64_t len, double* A, double* d, double* e) {
    const int k = (int)llround(sqrt(0.25 + 2.0 * (double)len) - 0.5);
    const double r2 = 1.4142135623730951, ir2 = 0.7071067811865475;
    int64_t idx = 0;
    for (int j = 0; j < k; ++j)
        for (int i = j; i < k; ++i, ++idx) {
            const double v = (i == j) ? x[idx] * r2 : x[idx];
            A[i * k + j] = v; A[j * k + i] = v;
        }
    tred2(k, A, d, e);
    tql2(k, A, d, e);
    idx = 0;
    for (int j = 0; j < k; ++j)
        for (int i = j; i < k; ++i, ++idx) {
            double acc = 0.0;
            for (int t = 0; t < k; ++t) if (d[t] > 0.0) acc += d[t] * A[i * k + t] * A[j * k + t];
            y[idx] = (i == j) ? acc * ir2 : acc;
        }
}
static void prox_soc(double* y, const double* x, int64_t len) {          /* IndSOC: t = first entry */
    double nx2 = 0.0;
    for (int64_t i = 1; i < len; ++i) nx2 += x[i] * x[i];
    const double nx = sqrt(nx2), t = x[0];
    if (t <= -nx) memset(y, 0, sizeof(double) * (size_t)len);
    else if (t >= nx) memcpy(y, x, sizeof(double) * (size_t)len);
    else {
        const double r = 0.5 * (1.0 + t / nx);
        y[0] = r * nx;
        for (int64_t i = 1; i < len; ++i) y[i] = r * x[i];
    }
}
static void prox_socrot(double* y, const double* x, int64_t len) {       /* IndRotatedSOC through a pi/4 rotation */
    const double s45 = 0.7071067811865475;
    const double x1 = s45 * x[0] + s45 * x[1], x2 = s45 * x[0] - s45 * x[1];
    double nx2 = x2 * x2;
    for (int64_t i = 2; i < len; ++i) nx2 += x[i] * x[i];
    const double nx = sqrt(nx2);
    double y1, y2, r = 1.0;
    if (x1 <= -nx) { y1 = 0; y2 = 0; r = 0.0; }
    else if (x1 >= nx) { y1 = x1; y2 = x2; }
    else { r = 0.5 * (1.0 + x1 / nx); y1 = r * nx; y2 = r * x2; }
    y[0] = s45 * y1 + s45 * y2;
    y[1] = s45 * y1 - s45 * y2;
    for (int64_t i = 2; i < len; ++i) y[i] = r * x[i];
}
static int cone_prox(int type, double* y, const double* x, int64_t len, double* A, double* d, double* e) {
    switch (type) {
        case C_FREE: memcpy(y, x, sizeof(double) * (size_t)len); return 0;
        case C_ZERO: memset(y, 0, sizeof(double) * (size_t)len); return 0;
        case C_NONNEG: for (int64_t i = 0; i < len; ++i) y[i] = x[i] > 0.0 ? x[i] : 0.0; return 0;
        case C_NONPOS: for (int64_t i = 0; i < len; ++i) y[i] = x[i] < 0.0 ? x[i] : 0.0; return 0;
        case C_SOC: prox_soc(y, x, len); return 0;
        case C_SOCROT: prox_socrot(y, x, len); return 0;
        case C_SDP: prox_psd(y, x, len, A, d, e); return 0;
        default: return -1;
    }
}
/* proxDual!   cones.jl:80-85 with the shortcuts :97-102 */
static int cone_prox_dual(int type, double* y, const double* x, int64_t len, double* negx, double* A, double* d, double* e) {
    switch (type) {
        case C_ZERO: return cone_prox(C_FREE, y, x, len, A, d, e);
        case C_FREE: return cone_prox(C_ZERO, y, x, len, A, d, e);
        case C_NONNEG: case C_NONPOS: return cone_prox(type, y, x, len, A, d, e);
        default: {
            for (int64_t i = 0; i < len; ++i) negx[i] = -x[i];          /* :81 */
            const int rc = cone_prox(type, y, negx, len, A, d, e);       /* :82 */
            for (int64_t i = 0; i < len; ++i) y[i] += x[i];              /* :83 */
            return rc;
        }
    }
}
static int product_prox(fosc* s, int dual, int64_t nK, const int32_t* type, const int64_t* start, const int64_t* len,
                        double* y, const double* x, double* negx) {
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads(s))
    for (int64_t q = 0; q < nK; ++q) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const size_t ws = (size_t)s->kmax * s->kmax + 2 * (size_t)s->kmax;
        double* A = s->eigA + (size_t)tid * ws;
        double* d = A + (size_t)s->kmax * s->kmax;
        double* e = d + s->kmax;
        const int rc = dual ? cone_prox_dual(type[q], y + start[q], x + start[q], len[q], negx + start[q], A, d, e)
                            : cone_prox(type[q], y + start[q], x + start[q], len[q], A, d, e);
        if (rc) bad = 1;
    }
    return bad ? -1 : 0;
}
/* prox!(y, S2::DualConeProduct, x)   cones.jl:122-142 */
int fosc_prox_cones(fosc* s, double* y, const double* x) {
    const int64_t n = s->n, m = s->m, nu = s->l;
    int rc = 0;
    rc |= product_prox(s, 0, s->nK2, s->K2type, s->K2start, s->K2len, y, x, s->negx);                      /* :136 */
    rc |= product_prox(s, 1, s->nK1, s->K1type, s->K1start, s->K1len, y + n, x + n, s->negx + n);          /* :137 */
    y[nu - 1] = x[nu - 1] > 0.0 ? x[nu - 1] : 0.0;                                                           /* :138 */
    rc |= product_prox(s, 1, s->nK2, s->K2type, s->K2start, s->K2len, y + nu, x + nu, s->negx + nu);       /* :139 */
    rc |= product_prox(s, 0, s->nK1, s->K1type, s->K1start, s->K1len, y + nu + n, x + nu + n, s->negx + nu + n);   /* :140 */
    y[2 * nu - 1] = x[2 * nu - 1] > 0.0 ? x[2 * nu - 1] : 0.0;                                               /* :141 */
    (void)m;
    return rc;
}

/* one outer iteration of GAP(alpha, alpha1, alpha2) without the status check   gap.jl:61-80 */
int fosc_gap_step(fosc* s, double* x, double alpha, double alpha1, double alpha2) {
    const int64_t N = s->N;
    double *t1 = s->tmp1, *t2 = s->tmp2;
    fosc_prox_affine(s, t1, x);                                          /* :45 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < N; ++i) t1[i] = alpha1 * t1[i] + (1 - alpha1) * x[i];      /* :48 */
    if (fosc_prox_cones(s, t2, t1)) return -1;                           /* :55 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < N; ++i) t2[i] = alpha2 * t2[i] + (1 - alpha2) * t1[i];     /* :58 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < N; ++i) x[i] = alpha * t2[i] + (1 - alpha) * x[i];         /* :78 */
    return 0;
}
/* one outer iteration of GAPA(alpha, beta); alpha12 in/out   gapa.jl:80-105 */
int fosc_gapa_step(fosc* s, double* x, double alpha, double beta, double* alpha12) {
    const int64_t N = s->N;
    const double a12 = *alpha12;
    double *t1 = s->tmp1, *t2 = s->tmp2;
    fosc_prox_affine(s, t1, x);
    for (int64_t i = 0; i < N; ++i) t1[i] = a12 * t1[i] + (1 - a12) * x[i];            /* :67 */
    if (fosc_prox_cones(s, t2, t1)) return -1;
    for (int64_t i = 0; i < N; ++i) t2[i] = a12 * t2[i] + (1 - a12) * t1[i];           /* :77 */
    double sum = 0.0, n1 = 0.0, n2 = 0.0;                                /* normedScalar  :36-47 (scalar loop, like the reference) */
    for (int64_t i = 0; i < N; ++i) {
        const double d1 = t2[i] - t1[i], d2 = t1[i] - x[i];
        sum += d1 * d2; n1 += d1 * d1; n2 += d2 * d2;
    }
    double scl = fabs(sum) / sqrt(n1 * n2);
    if (scl != scl) scl = 0.0; else if (scl > 1.0) scl = 1.0; else if (scl < 0.0) scl = 0.0;   /* :96-97 */
    const double sq = sqrt(1 - scl * scl);                                /* :98 */
    *alpha12 = (1 - beta) * (2 / (1 + sq)) + beta * 2.0;                  /* :100-101 */
    for (int64_t i = 0; i < N; ++i) x[i] = alpha * t2[i] + (1 - alpha) * x[i];          /* :103 */
    return 0;
}

/* one outer iteration of FISTA(alpha) without the status check; y, xold (N doubles each) and t are FISTAData's, in/out   fista.jl:28-48 */
int fosc_fista_step(fosc* s, double* x, double alpha, double* y, double* xold, double* t) {
    const int64_t N = s->N;
    double* t1 = s->tmp1;
    fosc_prox_affine(s, t1, y);                                          /* :35 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < N; ++i) t1[i] = alpha * t1[i] + (1 - alpha) * y[i];        /* :37 */
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < N; ++i) xold[i] = x[i];                                      /* :39 */
    if (fosc_prox_cones(s, x, t1)) return -1;                            /* :40 */
    const double told = *t;
    *t = (1 + sqrt(1 + 4 * told * told)) / 2;                            /* :45 */
    const double cf = (told - 1) / *t;
#pragma omp parallel for schedule(static) num_threads(nthreads(s))
    for (int64_t i = 0; i < N; ++i) y[i] = x[i] + cf * (x[i] - xold[i]);               /* :46 */
    return 0;
}

/* ------------------------------------------------------------------ set-up */
static void* xcalloc(size_t n, size_t sz) { void* p = calloc(n ? n : 1, sz); return p; }

void fosc_free(fosc* s) {
    if (!s) return;
    free(s->colptr); free(s->rowval); free(s->nzval); free(s->rowptr); free(s->colidx); free(s->rval);
    free(s->b); free(s->c); free(s->K1type); free(s->K2type); free(s->K1start); free(s->K1len); free(s->K2start); free(s->K2len);
    free(s->rhs); free(s->r); free(s->p); free(s->z); free(s->xinit); free(s->tmp1); free(s->tmp2); free(s->negx); free(s->eigA);
    free(s);
}

/* colptr/rowval: Julia CSC, 1-based.  Cone arrays: type codes as in include/foship.h, lengths; ranges are contiguous. */
fosc* fosc_create(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                  const double* b, const double* c, int64_t nK1, const int32_t* K1type, const int64_t* K1len,
                  int64_t nK2, const int32_t* K2type, const int64_t* K2len, int threads) {
    fosc* s = (fosc*)xcalloc(1, sizeof(fosc));
    if (!s) return NULL;
    s->m = m; s->n = n; s->l = n + m + 1; s->N = 2 * s->l; s->nnz = colptr[n] - 1;
    s->threads = threads > 0 ? threads : 1;
    s->colptr = (int64_t*)xcalloc((size_t)n + 1, 8); s->rowval = (int64_t*)xcalloc((size_t)s->nnz, 8);
    s->nzval = (double*)xcalloc((size_t)s->nnz, 8);
    for (int64_t j = 0; j <= n; ++j) s->colptr[j] = colptr[j] - 1;
    for (int64_t k = 0; k < s->nnz; ++k) { s->rowval[k] = rowval[k] - 1; s->nzval[k] = nzval[k]; }
    /* CSR copy (counting transpose) */
    s->rowptr = (int64_t*)xcalloc((size_t)m + 1, 8); s->colidx = (int64_t*)xcalloc((size_t)s->nnz, 8);
    s->rval = (double*)xcalloc((size_t)s->nnz, 8);
    for (int64_t k = 0; k < s->nnz; ++k) s->rowptr[s->rowval[k] + 1] += 1;
    for (int64_t i = 0; i < m; ++i) s->rowptr[i + 1] += s->rowptr[i];
    {
        int64_t* fill = (int64_t*)xcalloc((size_t)m, 8);
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = s->colptr[j]; k < s->colptr[j + 1]; ++k) {
                const int64_t i = s->rowval[k], pos = s->rowptr[i] + fill[i]++;
                s->colidx[pos] = j; s->rval[pos] = s->nzval[k];
            }
        free(fill);
    }
    s->b = (double*)xcalloc((size_t)m, 8); s->c = (double*)xcalloc((size_t)n, 8);
    memcpy(s->b, b, sizeof(double) * (size_t)m); memcpy(s->c, c, sizeof(double) * (size_t)n);
    s->nK1 = nK1; s->nK2 = nK2;
    s->K1type = (int32_t*)xcalloc((size_t)nK1, 4); s->K2type = (int32_t*)xcalloc((size_t)nK2, 4);
    s->K1start = (int64_t*)xcalloc((size_t)nK1, 8); s->K1len = (int64_t*)xcalloc((size_t)nK1, 8);
    s->K2start = (int64_t*)xcalloc((size_t)nK2, 8); s->K2len = (int64_t*)xcalloc((size_t)nK2, 8);
    int64_t pos = 0, kmax = 1;
    for (int64_t q = 0; q < nK1; ++q) {
        s->K1type[q] = K1type[q]; s->K1start[q] = pos; s->K1len[q] = K1len[q]; pos += K1len[q];
        if (K1type[q] == C_SDP) { const int64_t k = llround(sqrt(0.25 + 2.0 * (double)K1len[q]) - 0.5); if (k > kmax) kmax = k; }
    }
    if (pos != m) { fosc_free(s); return NULL; }
    pos = 0;
    for (int64_t q = 0; q < nK2; ++q) {
        s->K2type[q] = K2type[q]; s->K2start[q] = pos; s->K2len[q] = K2len[q]; pos += K2len[q];
        if (K2type[q] == C_SDP) { const int64_t k = llround(sqrt(0.25 + 2.0 * (double)K2len[q]) - 0.5); if (k > kmax) kmax = k; }
    }
    if (pos != n) { fosc_free(s); return NULL; }
    s->kmax = (int)kmax;
    const size_t N = (size_t)s->N;
    s->rhs = (double*)xcalloc(N, 8); s->r = (double*)xcalloc(N, 8); s->p = (double*)xcalloc(N, 8); s->z = (double*)xcalloc(N, 8);
    s->xinit = (double*)xcalloc(N, 8); s->tmp1 = (double*)xcalloc(N, 8); s->tmp2 = (double*)xcalloc(N, 8); s->negx = (double*)xcalloc(N, 8);
    s->eigA = (double*)xcalloc((size_t)s->threads * ((size_t)kmax * kmax + 2 * (size_t)kmax), 8);
    s->firstrun = 1; s->i = 1; s->cgiter = 0;
    return s;
}
void fosc_set_affine_state(fosc* s, const double* xinit, int64_t i) {    /* CGdata.xinit, AffinePlusLinear.i */
    memcpy(s->xinit, xinit, sizeof(double) * (size_t)s->N);
    s->firstrun = 0; s->i = i;
}
int64_t fosc_get_cgiter(const fosc* s) { return s->cgiter; }
int fosc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
