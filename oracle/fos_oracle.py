"""
fos_oracle.py -- CPU restatement (numpy/scipy, fp64) of the FirstOrderSolvers.jl hot path.

*** TEST INFRASTRUCTURE ONLY. ***  This module is the parity checker for the HIP path.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it.  The product (`firstordersolvers.jl_amd`) never imports or falls back to it.

What it restates (paths under /root/reference; every function cites the lines it follows):

  src/utilities/conjugategradients.jl:31-55      conjugategradient!
  src/utilities/affinepluslinear.jl:4-15,37-52   KKTMatrix, mul!
  src/utilities/affinepluslinear.jl:58-126       AffinePlusLinear, prox!
  src/problemforms/HSDE/HSDEAffine.jl:2-65       HSDEMatrixQ, mul!, transpose mul!
  src/problemforms/HSDE/HSDEAffine.jl:68-147     HSDEMatrix, prox!, mul!
  src/cones.jl:4-14,31-142                       conemap, ConeProduct, proxDual!, DualConeProduct
  src/solvers/gap.jl, gapa.jl, fista.jl, dykstra.jl, solvers.jl   step / getsol / S1! / S2!
  src/solverwrapper.jl:2-41                      solve!, iterate
  src/problemforms/HSDE/HSDE.jl:7-61             HSDE assembly, initial value, solution
  src/problemforms/HSDE/HSDEStatus.jl:27-102     checkstatus, print formats, getvalues
  src/FOSSolverInterface.jl:8-64                 loadproblem!, optimize!

Third-party arithmetic NOT under /root/reference: the per-cone projections are calls into
ProximalOperators.jl (Project.toml:10, version UNPINNED, no Manifest).  They are restated
here from that package's published algorithm (IndFree/IndZero/IndPoint/IndNonnegative/
IndNonpositive/IndSOC/IndRotatedSOC/IndPSD(scaling=true)/IndExpPrimal/IndExpDual).  Pinning status:
  * IndPSD: pinned by the reference's RNG-free known answer test/testPSD.jl:3-4,14-25.
  * IndNonnegative / Zero / Free / IndSOC (the two second-order cones of the NNLS epigraph form): pinned through the
    DR/GAPA solves of test/testDRandGAPA.jl ON THE REFERENCE'S OWN DATA -- oracle/julia_random.py restates Julia's
    MersenneTwister + randn, the regenerated `Random.seed!(2)` inputs reproduce the file's literal optima
    (12.38418747141913 to 3e-15), and the solves meet every assertion and threshold of that file
    (tests/test_reference_known_answers.py).  IndAffine / IndBox and the Feasibility form's stopping rule: pinned the
    same way by the seven data-dependent outcomes of test/testfeasibility.jl.
  * IndRotatedSOC / IndExpPrimal / IndExpDual: PARITY UNPINNED upstream -- no reference test touches them
    (the exponential cone follows the SCS-style bisection + Newton projection).  What is checked instead, both
    independent of this file: the projections' optimality conditions (tests/cone_certificates.py) and CLOSED-FORM optima
    of textbook programs over these cones in MathProgBase's conventions (entry order, 2pq, the e of the dual cone:
    tests/analytic_cases.py, reached by this oracle and by the HIP path to 1e-8).
The Julia reference itself cannot be executed in the build container (no julia binary), so
the oracle is pinned against the reference's own tests restated in tests/test_oracle_*.py
(identities vs dense linear algebra, CG property test, PSD known answer, print formats) and, for whole solves,
against the literals and outcomes those tests hold for seeded data (tests/test_reference_known_answers.py).

Indices are 0-based here; the z layout is the reference's  [x(n); y(m); tau; r(n); s(m); kappa]
(src/cones.jl:126-141, src/problemforms/HSDE/HSDEStatus.jl:93-101).
"""
from __future__ import annotations

import math
import time
import warnings

import numpy as np
import scipy.sparse as sp

EPS = float(np.finfo(np.float64).eps)   # Julia eps()


# ----------------------------------------------------------------------------------------
# Reduction "space": where inner products live.  The reference is one process (LocalSpace).  For the
# cone-sharded multi-GPU design (no reference equivalent) a rank holds a SHARD
# z_g = [x_g; y_g; tau; r_g; s_g; kappa] with tau/kappa replicated; every inner product is then
# all-reduced over ranks with the replicated entries counted once.  tests/test_sharding_gloo.py plugs
# a torch.distributed(gloo) all-reduce in here to check the reduction points the HIP path uses.
# ----------------------------------------------------------------------------------------

class LocalSpace:
    def allsum(self, v):
        return v

    def global_l(self, l_local):
        return l_local

    def dotN(self, a, b):
        """dot of two N-vectors (N = 2l, layout [part1(l); part2(l)], tau-like entries at l-1 and 2l-1)."""
        return np.float64(np.dot(a, b))

    def dotL(self, a, b):
        """dot of two vectors without replicated entries (x-, y-, r-, s-like)."""
        return np.float64(np.dot(a, b))

    # part-aware forms (the row-sharded space below treats column-space and row-space vectors differently)
    def dotX(self, a, b):
        """dot of two column-space (x-, r-, c-like) vectors."""
        return self.dotL(a, b)

    def dotY(self, a, b):
        """dot of two row-space (y-, s-, b-like) vectors."""
        return self.dotL(a, b)

    def sumX(self, v):
        """a column-space vector that every rank holds a partial sum of (A'y) -> its sum over ranks."""
        return v


class ShardedSpace(LocalSpace):
    """allreduce: callable mapping a float64 ndarray to its sum over ranks."""

    def __init__(self, allreduce, l_global):
        self._ar = allreduce
        self.l_global = l_global

    def allsum(self, v):
        return np.float64(self._ar(np.array([v], dtype=np.float64))[0])

    def global_l(self, l_local):
        return self.l_global

    def dotN(self, a, b):
        l = a.shape[0] // 2
        rep = a[l - 1] * b[l - 1] + a[2 * l - 1] * b[2 * l - 1]
        return self.allsum(np.dot(a, b) - rep) + rep

    def dotL(self, a, b):
        return self.allsum(np.dot(a, b))


class RowShardedSpace(LocalSpace):
    """Row sharding of a NON block-diagonal A: rank g owns the rows of A of its K1 cones (y, s, b local), every
    rank holds all n columns (x, r, c and tau, kappa replicated).  What crosses ranks: the n-vector A'y = sum_g A_g'y_g (one
    all-reduce per Q apply, HSDEAffine.jl:51) and the scalar sums, in which the replicated entries are counted once.
    allreduce: callable mapping a float64 ndarray to its sum over ranks; n = number of (replicated) columns."""

    def __init__(self, allreduce, l_global, n):
        self._ar = allreduce
        self.l_global = l_global
        self.n = n

    def allsum(self, v):
        return np.float64(self._ar(np.array([v], dtype=np.float64))[0])

    def global_l(self, l_local):
        return self.l_global

    def dotN(self, a, b):
        l, n = a.shape[0] // 2, self.n
        rep = (np.dot(a[:n], b[:n]) + a[l - 1] * b[l - 1] + np.dot(a[l:l + n], b[l:l + n]) + a[2 * l - 1] * b[2 * l - 1])
        loc = np.dot(a[n:l - 1], b[n:l - 1]) + np.dot(a[l + n:2 * l - 1], b[l + n:2 * l - 1])
        return self.allsum(loc) + np.float64(rep)

    def dotL(self, a, b):
        raise TypeError("RowShardedSpace: use dotX (replicated column space) or dotY (sharded row space)")

    def dotX(self, a, b):
        return np.float64(np.dot(a, b))

    def dotY(self, a, b):
        return self.allsum(np.dot(a, b))

    def sumX(self, v):
        return np.asarray(self._ar(np.array(v, dtype=np.float64)), dtype=np.float64)


LOCAL = LocalSpace()

# ----------------------------------------------------------------------------------------
# HSDEMatrixQ            src/problemforms/HSDE/HSDEAffine.jl:2-65
# ----------------------------------------------------------------------------------------


class HSDEMatrixQ:
    """Matrix-free  Q = [0 A' c; -A 0 b; -c' -b' 0]   (HSDEAffine.jl:2-20)."""

    def __init__(self, A, b, c, space=LOCAL):
        A = sp.csc_matrix(A) if not sp.issparse(A) else A.tocsc()
        self.space = space
        self.A = A
        self.At = A.T.tocsr()        # transpose(A) of a CSC matrix: a CSR view, row gather
        self.b = np.asarray(b, dtype=np.float64).reshape(-1)
        self.c = np.asarray(c, dtype=np.float64).reshape(-1)
        self.am, self.an = A.shape
        assert self.b.shape[0] == self.am      # HSDEAffine.jl:15
        assert self.c.shape[0] == self.an      # HSDEAffine.jl:16

    @property
    def shape(self):                            # HSDEAffine.jl:20
        l = self.am + self.an + 1
        return (l, l)

    @property
    def global_size(self):
        return self.space.global_l(self.am + self.an + 1)

    def mul(self, Y, B):
        """mul!(Y, Q, B)   HSDEAffine.jl:41-59."""
        n, m = self.an, self.am
        assert self.shape == (Y.shape[0], B.shape[0])
        b1 = B[:n]
        b2 = B[n:n + m]
        b3 = B[n + m]
        y1 = self.space.sumX(self.At @ b2)      # :51  mul!(y1, transpose(A), b2)   (row-sharded A: summed over the ranks)
        y2 = self.A @ b1                        # :52  mul!(y2, A, b1)
        y1 = y1 + b3 * self.c                   # :54  y1 .+= b3.*c
        y2 = y2 - b3 * self.b                   # :55  y2 .-= b3.*b
        y2 = -y2                                # :56  y2 .= .-y2
        last = -self.space.dotX(self.c, b1) - self.space.dotY(self.b, b2)   # :57 (computed before Y is written: B may not alias Y)
        Y[:n] = y1
        Y[n:n + m] = y2
        Y[n + m] = last
        return Y

    def mul_t(self, Y, B):
        """mul!(Y, transpose(Q), B) = -(Q*B)   HSDEAffine.jl:61-65."""
        self.mul(Y, B)
        np.negative(Y, out=Y)
        return Y

    def todense(self):
        n, m = self.an, self.am
        l = n + m + 1
        Q = np.zeros((l, l))
        Ad = self.A.toarray()
        Q[:n, n:n + m] = Ad.T
        Q[:n, l - 1] = self.c
        Q[n:n + m, :n] = -Ad
        Q[n:n + m, l - 1] = self.b
        Q[l - 1, :n] = -self.c
        Q[l - 1, n:n + m] = -self.b
        return Q


class _PlainMatrix:
    """Adapter giving a dense/sparse matrix the mul/mul_t protocol (used by KKTMatrix tests,
    test/affinepluslinear.jl:7-19 where the inner operator is a plain randn(10,20))."""

    def __init__(self, A):
        self.A = A
        self.shape = A.shape

    def mul(self, Y, B):
        Y[:] = self.A @ B
        return Y

    def mul_t(self, Y, B):
        Y[:] = self.A.T @ B
        return Y


def _as_operator(A):
    return A if hasattr(A, "mul") else _PlainMatrix(A)


# ----------------------------------------------------------------------------------------
# KKTMatrix              src/utilities/affinepluslinear.jl:4-15,37-52
# ----------------------------------------------------------------------------------------


class KKTMatrix:
    """Matrix-free [I A'; A -I]."""

    def __init__(self, A):
        self.A = _as_operator(A)
        self.am, self.an = self.A.shape

    @property
    def shape(self):
        return (self.am + self.an, self.am + self.an)

    def mul(self, y, x):
        """mul!(y, M, x)   affinepluslinear.jl:37-49."""
        an, am = self.an, self.am
        x1, x2 = x[:an], x[an:an + am]
        y1, y2 = y[:an], y[an:an + am]
        self.A.mul_t(y1, x2)        # :45
        y1 += x1                    # :46
        self.A.mul(y2, x1)          # :47
        y2 -= x2                    # :48
        return y

    mul_t = mul                     # :52  transpose == self


# ----------------------------------------------------------------------------------------
# conjugategradient!     src/utilities/conjugategradients.jl:31-55
# ----------------------------------------------------------------------------------------


def conjugategradient(x, A, b, r, p, Ap, tol=None, max_iters=10000, space=LOCAL):
    """Golub/Van Loan CG exactly as the reference runs it (also on the indefinite KKT system).
    Returns the iteration count (>= 1).  x is the warm start and receives the solution."""
    if tol is None:
        tol = A.shape[1] * EPS                   # :31 default
    A.mul(Ap, x)                                 # :32
    np.subtract(b, Ap, out=r)                    # :33
    p[:] = r                                     # :34
    rn = space.dotN(r, r)                        # :35   (np.float64: x/0 -> inf/nan silently, as in Julia)
    it = 1                                       # :36
    while True:
        A.mul(Ap, p)                             # :38
        with np.errstate(divide="ignore", invalid="ignore"):
            alpha = rn / space.dotN(Ap, p)           # :39
        x += alpha * p                           # :40
        r -= alpha * Ap                          # :41
        if np.sqrt(space.dotN(r, r)) <= tol or it >= max_iters:   # :42  norm(r)
            break
        rnold = rn                               # :45
        rn = space.dotN(r, r)                    # :46
        with np.errstate(divide="ignore", invalid="ignore"):
            beta = rn / rnold                    # :47
        p *= beta                                # :49
        p += r                                   # :50
        it += 1                                  # :51
    if it == max_iters:                          # :53
        warnings.warn("CG reached max iterations, result may be inaccurate")
    return it


def conjugategradient_merged(x, A, b, r, p, s, w, tol=None, max_iters=10000, space=LOCAL):
    """NOT in the reference: the merged-reduction (Chronopoulos & Gear) rearrangement of the recurrence above that the HIP path
    offers as a CG variant (include/foship.h FOS_CG_MERGED_*; default on sharded handles).  Same Krylov iterates in exact
    arithmetic, the same iteration count convention and the same stop test on the same recursively updated residual
    (conjugategradients.jl:42); what changes is where the inner products are formed: w = A r is swept instead of A p, s = A p
    follows by recurrence, and ONE reduction point per iteration carries r.r and w.r:
        beta_i = g_i / g_{i-1},   alpha_i = g_i / (d_i - beta_i g_i / alpha_{i-1}),   g = r.r, d = w.r   (i = it - 1).
    Exists so that the parity tests of that variant compare against the same arithmetic; nothing else uses it."""
    if tol is None:
        tol = A.shape[1] * EPS
    A.mul(w, x)                                  # :32
    np.subtract(b, w, out=r)                     # :33
    gam = space.dotN(r, r)                       # :35
    A.mul(w, r)
    delta = space.dotN(w, r)
    gam_prev = alpha_prev = None
    it = 1                                       # :36
    while True:
        with np.errstate(divide="ignore", invalid="ignore"):
            if it == 1:
                alpha = gam / delta
                p[:] = r
                s[:] = w
            else:
                beta = gam / gam_prev
                alpha = gam / (delta - beta * gam / alpha_prev)
                p *= beta                        # :49
                p += r                           # :50
                s *= beta
                s += w
        x += alpha * p                           # :40
        r -= alpha * s                           # :41   (s = A p)
        gam_prev, alpha_prev = gam, alpha
        gam = space.dotN(r, r)
        if np.sqrt(gam) <= tol or it >= max_iters:   # :42
            break
        A.mul(w, r)
        delta = space.dotN(w, r)
        it += 1                                  # :51
    if it == max_iters:                          # :53
        warnings.warn("CG reached max iterations, result may be inaccurate")
    return it


class CGdata:
    """conjugategradients.jl:1-11."""

    def __init__(self, size):
        self.r = np.empty(size)
        self.p = np.empty(size)
        self.z = np.empty(size)
        self.xinit = np.empty(size)
        self.firstrun = True


# ----------------------------------------------------------------------------------------
# AffinePlusLinear       src/utilities/affinepluslinear.jl:58-126
# ----------------------------------------------------------------------------------------


class AffinePlusLinear:
    """f([x;z]) = q'x + indicator(Ax - beta z == b), beta = +-1."""

    def __init__(self, A, b, q, beta, decreasing_accuracy=False):
        self.A = _as_operator(A)
        am, an = self.A.shape
        assert beta == 1 or beta == -1                       # :73
        self.M = KKTMatrix(self.A)
        self.beta = beta
        self.b = np.asarray(b, dtype=np.float64)
        self.q = np.asarray(q, dtype=np.float64)
        self.rhs = np.empty(am + an)
        self.rhs[an:] = self.b                               # :76-77
        self.decreasing_accuracy = decreasing_accuracy
        self.i = 1                                           # :78 call counter (decides tol)
        self.cgiter = 0
        self.cgdata = CGdata(am + an)
        self.cg_variant = "reference"                        # "merged": conjugategradient_merged (the HIP path's variant; tests only)

    def getcgiter(self):                                     # :81
        return self.cgiter

    def tolerance(self):
        """:108-112 -- tolerance that the *next* prox! call will use."""
        size2 = getattr(self.A, "global_size", self.A.shape[1])      # size(S.A, 2)
        if self.decreasing_accuracy:
            return max(0.2 ** math.sqrt(self.i), size2 * EPS)
        return size2 * EPS

    def prox(self, y, x):
        """prox!(y, S, x)   affinepluslinear.jl:83-126."""
        an, am = self.M.an, self.M.am
        rhs1 = self.rhs[:an]
        x1 = x[:an]
        x2 = x[an:an + am]
        beta = self.beta
        self.A.mul_t(rhs1, x2)                               # :94
        rhs1[:] = beta * rhs1 + x1 - self.q                  # :95
        cg = self.cgdata
        if cg.firstrun:                                      # :101-104
            cg.xinit[:] = x
            cg.firstrun = False
        y[:] = cg.xinit                                      # :106
        tol = self.tolerance()                               # :108-112
        self.i += 1                                          # :114
        max_iters = 1000                                     # :115
        if self.cg_variant == "merged":
            if not hasattr(cg, "s"):
                cg.s = np.empty_like(cg.r)
            it = conjugategradient_merged(y, self.M, self.rhs, cg.r, cg.p, cg.s, cg.z, tol=tol, max_iters=max_iters,
                                          space=getattr(self.A, "space", LOCAL))
        else:
            it = conjugategradient(y, self.M, self.rhs, cg.r, cg.p, cg.z, tol=tol, max_iters=max_iters,
                                   space=getattr(self.A, "space", LOCAL))                             # :117
        self.cgiter = it                                     # :121
        cg.xinit[:] = y                                      # :122
        y[an:an + am] *= beta                                # :124
        return 0.0


# ----------------------------------------------------------------------------------------
# HSDEMatrix             src/problemforms/HSDE/HSDEAffine.jl:68-147
# ----------------------------------------------------------------------------------------


class HSDEMatrix:
    """Matrix-free [I Q'; Q -I] with its own CG state (HSDEAffine.jl:68-89)."""

    def __init__(self, Q):
        self.Q = Q
        self.cgdata = CGdata(2 * Q.shape[0])

    @property
    def shape(self):
        m = self.Q.shape[0]
        return (2 * m, 2 * m)

    def mul(self, Y, B):
        """HSDEAffine.jl:131-144."""
        mQ = self.Q.shape[0]
        y1, y2 = Y[:mQ], Y[mQ:2 * mQ]
        b1, b2 = B[:mQ], B[mQ:2 * mQ]
        self.Q.mul_t(y1, b2)     # :139
        self.Q.mul(y2, b1)       # :140
        y1 += b1                 # :141
        y2 -= b2                 # :142
        return Y

    mul_t = mul                  # :146-147

    def prox(self, y, x):
        """argmin ||x-y|| s.t. Q u == v, [u;v] = y   HSDEAffine.jl:105-126."""
        tol = self.shape[1] * EPS                # :106
        max_iters = 1000                         # :107
        cg = self.cgdata
        if cg.firstrun:                          # :109-112
            cg.xinit[:] = x
            cg.firstrun = False
        y[:] = cg.xinit                          # :114
        it = conjugategradient(y, self, x, cg.r, cg.p, cg.z, tol=tol, max_iters=max_iters)  # :116
        cg.xinit[:] = y                          # :119
        m = self.Q.shape[0]
        u = y[:m].copy()
        self.Q.mul(y[m:2 * m], u)                # :122-124   v = Q*u
        self.cgiter = it
        return 0.0


# ----------------------------------------------------------------------------------------
# Cone projections (ProximalOperators.jl -- third-party, restated from its published algorithm)
# ----------------------------------------------------------------------------------------

CONE_FREE, CONE_ZERO, CONE_NONNEG, CONE_NONPOS, CONE_SOC, CONE_SOCROT, CONE_SDP, CONE_EXPP, CONE_EXPD = range(9)

# Symbol -> code, the keys of conemap  (src/cones.jl:4-14)
CONE_CODES = {
    "Free": CONE_FREE, "Zero": CONE_ZERO, "NonNeg": CONE_NONNEG, "NonPos": CONE_NONPOS,
    "SOC": CONE_SOC, "SOCRotated": CONE_SOCROT, "SDP": CONE_SDP,
    "ExpPrimal": CONE_EXPP, "ExpDual": CONE_EXPD,
}
CONE_NAMES = {v: k for k, v in CONE_CODES.items()}


def prox_free(y, x):            # IndFree: identity
    y[:] = x


def prox_zero(y, x):            # IndZero / IndPoint(): the origin
    y[:] = 0.0


def prox_nonneg(y, x):          # IndNonnegative
    np.maximum(x, 0.0, out=y)


def prox_nonpos(y, x):          # IndNonpositive
    np.minimum(x, 0.0, out=y)


def prox_soc(y, x):
    """IndSOC: {(t, v): ||v|| <= t}, t first.  (ProximalOperators indSOC.jl)"""
    nx = float(np.linalg.norm(x[1:]))
    t = float(x[0])
    if t <= -nx:
        y[:] = 0.0
    elif t >= nx:
        y[:] = x
    else:
        r = 0.5 * (1.0 + t / nx)
        y[0] = r * nx
        y[1:] = r * x[1:]


_S45 = 0.7071067811865475


def prox_socrot(y, x):
    """IndRotatedSOC: {(p, q, v): ||v||^2 <= 2pq, p,q >= 0} via a pi/4 rotation onto IndSOC."""
    x1 = _S45 * x[0] + _S45 * x[1]
    x2 = _S45 * x[0] - _S45 * x[1]
    nx = math.sqrt(x2 * x2 + float(np.dot(x[2:], x[2:])))
    t = x1
    if t <= -nx:
        y[:] = 0.0
    elif t >= nx:
        y[0] = x1
        y[1] = x2
        y[2:] = x[2:]
    else:
        r = 0.5 * (1.0 + t / nx)
        y[0] = r * nx
        y[1] = r * x2
        y[2:] = r * x[2:]
    y1 = _S45 * y[0] + _S45 * y[1]
    y2 = _S45 * y[0] - _S45 * y[1]
    y[0] = y1
    y[1] = y2


# Exponential cone  K = cl{(r,s,t): s > 0, s exp(r/s) <= t}.  ProximalOperators' IndExpPrimal is a port of the SCS
# projection (bisection on the dual variable rho with a 1-D Newton solve inside); IndExpDual is
# PrecomposeDiagonal(Conjugate(IndExpPrimal()), -1), i.e. Moreau's  P_K*(x) = x + P_K(-x).  PARITY UNPINNED: no
# reference test touches these cones and the package source is not in the checkout.
EXP_PROJ_TOL = 1e-15
EXP_PROJ_MAXIT = 100


def _exp_newton_onz(rho, y_hat, z_hat, w):
    t = max(max(w - z_hat, -z_hat), EXP_PROJ_TOL)
    for _ in range(EXP_PROJ_MAXIT):
        f = (1.0 / rho ** 2) * t * (t + z_hat) - y_hat / rho + math.log(t / rho) + 1.0
        fp = (1.0 / rho ** 2) * (2.0 * t + z_hat) + 1.0 / t
        t = t - f / fp
        if t <= -z_hat:
            t = -z_hat
            break
        elif t <= 0:
            t = 0.0
            break
        elif abs(f) < EXP_PROJ_TOL:
            break
    return t + z_hat


def _exp_solve_with_rho(v, rho, w):
    x3 = _exp_newton_onz(rho, v[1], v[2], w)
    x2 = (1.0 / rho) * (x3 - v[2]) * x3
    x1 = v[0] - rho
    return (x1, x2, x3)


def _exp_calc_grad(v, rho, warm):
    x = _exp_solve_with_rho(v, rho, warm[1])
    if x[1] == 0:
        g = x[0]
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            g = x[0] + x[1] * float(np.log(np.float64(x[1]) / np.float64(x[2])))
    return g, x


def _exp_rho_ub(v):
    lb = 0.0
    rho = 2.0 ** -3
    g, z = _exp_calc_grad(v, rho, v)
    while g > 0:
        lb = rho
        rho = rho * 2
        g, z = _exp_calc_grad(v, rho, z)
    return rho, lb


def prox_exp_primal(y, x):
    r, s, t = float(x[0]), float(x[1]), float(x[2])
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        in_cone = (s > 0 and np.float64(s) * np.exp(np.float64(r) / np.float64(s)) <= t) or (r <= 0 and s == 0 and t >= 0)
        in_polar = (-r < 0 and np.float64(r) * np.exp(np.float64(s) / np.float64(r)) <= -math.e * t) or (-r == 0 and -s >= 0 and -t >= 0)
    if in_cone:
        y[:] = x
    elif in_polar:
        y[:] = 0.0
    elif r < 0 and s < 0:
        y[0] = r
        y[1] = max(s, 0.0)
        y[2] = max(t, 0.0)
    else:
        v = (r, s, t)
        ub, lb = _exp_rho_ub(v)
        z = v
        for _ in range(EXP_PROJ_MAXIT):
            rho = (ub + lb) / 2
            g, z = _exp_calc_grad(v, rho, z)
            if g > 0:
                lb = rho
            else:
                ub = rho
            if ub - lb < EXP_PROJ_TOL:
                break
        y[:] = z


def prox_exp_dual(y, x):
    tmp = np.empty(3)
    prox_exp_primal(tmp, -np.asarray(x, dtype=np.float64))
    y[:] = x + tmp


def psd_dim(length):
    """Matrix order k with k(k+1)/2 == length."""
    k = int(round(math.sqrt(0.25 + 2.0 * length) - 0.5))
    assert k * (k + 1) // 2 == length, "PSD cone length is not triangular"
    return k


def svec_to_mat(x, k):
    """Lower triangle, column-major packed vector -> full symmetric k x k matrix."""
    M = np.zeros((k, k))
    idx = 0
    for j in range(k):
        cnt = k - j
        M[j:, j] = x[idx:idx + cnt]
        M[j, j:] = x[idx:idx + cnt]
        idx += cnt
    return M


def mat_to_svec(M, y):
    k = M.shape[0]
    idx = 0
    for j in range(k):
        cnt = k - j
        y[idx:idx + cnt] = M[j:, j]
        idx += cnt


def prox_psd_scaled(y, x):
    """IndPSD(scaling=true) on the packed lower triangle with off-diagonals pre-multiplied by
    sqrt(2) (the MathProgBase/SCS svec convention; conemap :SDP, src/cones.jl:11).
    ProximalOperators' vector method: scale the DIAGONAL by sqrt(2) (so the whole packed
    matrix is sqrt(2) x the true one), symmetric eigendecomposition of the packed matrix,
    clamp eigenvalues at 0, rebuild, repack the lower triangle, scale the diagonal back."""
    k = psd_dim(x.shape[0])
    w = np.array(x, dtype=np.float64, copy=True)
    dpos = np.cumsum([0] + [k - j for j in range(k - 1)])      # positions of the diagonal entries
    w[dpos] *= math.sqrt(2.0)
    M = svec_to_mat(w, k)
    lam, V = np.linalg.eigh(M)
    lam = np.maximum(lam, 0.0)
    P = (V * lam) @ V.T
    mat_to_svec(P, y)
    y[dpos] *= 1.0 / math.sqrt(2.0)


def prox_psd_matrix(Y):
    """IndPSD() matrix method (test/testPSD.jl:14-16): projection of a symmetric matrix."""
    lam, V = np.linalg.eigh(Y)
    lam = np.maximum(lam, 0.0)
    return (V * lam) @ V.T


_PROX = {
    CONE_FREE: prox_free, CONE_ZERO: prox_zero, CONE_NONNEG: prox_nonneg, CONE_NONPOS: prox_nonpos,
    CONE_SOC: prox_soc, CONE_SOCROT: prox_socrot, CONE_SDP: prox_psd_scaled,
    CONE_EXPP: prox_exp_primal, CONE_EXPD: prox_exp_dual,
}


def cone_prox(code, y, x):
    try:
        f = _PROX[code]
    except KeyError:
        raise NotImplementedError("cone %s not restated" % CONE_NAMES.get(code, code))
    f(y, x)


def cone_prox_dual(code, y, x):
    """proxDual!   src/cones.jl:80-85 (Moreau: y = x + P_K(-x)) with the shortcuts :97-102."""
    if code == CONE_ZERO:            # :98   dual of Zero is Free
        prox_free(y, x)
    elif code == CONE_FREE:          # :100  dual of Free is the origin (IndPoint())
        prox_zero(y, x)
    elif code == CONE_NONNEG:        # :101  self dual
        prox_nonneg(y, x)
    elif code == CONE_NONPOS:        # :102
        prox_nonpos(y, x)
    else:                            # :80-85 generic
        cone_prox(code, y, -x)
        y += x


class ConeProduct:
    """src/cones.jl:31-77.  `cones` = list of (code, start0, length); contiguous, ordered, gap-free."""

    def __init__(self, cones=()):
        self.cones = []
        prev_end = 0
        for (code, start, length) in cones:
            if isinstance(code, str):
                code = CONE_CODES[code]
            assert start == prev_end, "ranges must be contiguous and ordered (cones.jl:66-72)"
            prev_end = start + length
            self.cones.append((code, start, length))
        self.total = prev_end

    @staticmethod
    def from_lengths(pairs):
        cones, start = [], 0
        for code, length in pairs:
            cones.append((code, start, length))
            start += length
        return ConeProduct(cones)

    def prox(self, y, x):            # :89-94
        for code, s, ln in self.cones:
            cone_prox(code, y[s:s + ln], x[s:s + ln])

    def prox_dual(self, y, x):       # :106-111
        for code, s, ln in self.cones:
            cone_prox_dual(code, y[s:s + ln], x[s:s + ln])


class DualConeProduct:
    """K2 x K1* x R+ x K2* x K1 x R+   src/cones.jl:113-142."""

    def __init__(self, K1, K2):
        self.K1, self.K2 = K1, K2
        self.m, self.n = K1.total, K2.total      # :121

    def prox(self, y, x):
        m, n = self.m, self.n
        nu = n + m + 1
        self.K2.prox(y[0:n], x[0:n])                                  # :136
        self.K1.prox_dual(y[n:n + m], x[n:n + m])                     # :137
        y[nu - 1] = max(x[nu - 1], 0.0)                               # :138
        self.K2.prox_dual(y[nu:nu + n], x[nu:nu + n])                 # :139
        self.K1.prox(y[nu + n:nu + n + m], x[nu + n:nu + n + m])      # :140
        y[2 * nu - 1] = max(x[2 * nu - 1], 0.0)                       # :141


# ----------------------------------------------------------------------------------------
# Model / HSDE assembly    src/types.jl:30-60, src/FOSSolverInterface.jl:27-64, HSDE.jl:7-61
# ----------------------------------------------------------------------------------------

class Model:
    """The fields of FOSMathProgModel the hot path reads (types.jl:30-52)."""

    def __init__(self, A, b, c, K1, K2, space=LOCAL):
        self.space = space
        self.A = sp.csc_matrix(A) if not sp.issparse(A) else A.tocsc()     # loadproblem! sparsifies, :27-29
        self.b = np.asarray(b, dtype=np.float64).reshape(-1)
        self.c = np.asarray(c, dtype=np.float64).reshape(-1)
        self.K1 = K1 if isinstance(K1, ConeProduct) else ConeProduct.from_lengths(K1)
        self.K2 = K2 if isinstance(K2, ConeProduct) else ConeProduct.from_lengths(K2)
        m, n = self.A.shape
        assert self.K1.total == m and self.K2.total == n
        self.history = {}
        self.init_duration = 0


class IndAffineDirect:
    """IndAffine([sparse(Q) -I], zeros(l)) of HSDE.jl:12-15 (direct = true): the exact projection onto {[u; v]: Q u - v = 0}.
    ProximalOperators factorises the sparse matrix (source not in the checkout); the projection itself is unique:
        w = (I + Q Q')^-1 (Q u - v),  u+ = u - Q'w,  v+ = v + w.
    Restated with a dense Cholesky factor (scipy) -- small problems only, as the reference's own direct tests are."""

    def __init__(self, Q):
        import scipy.linalg
        self.A = Q
        self.Qd = Q.todense()
        self.l = self.Qd.shape[0]
        self.chol = scipy.linalg.cho_factor(np.eye(self.l) + self.Qd @ self.Qd.T)

    def prox(self, y, x):
        import scipy.linalg
        l = self.l
        u, v = x[:l], x[l:]
        w = scipy.linalg.cho_solve(self.chol, self.Qd @ u - v)
        y[:l] = u - self.Qd.T @ w
        y[l:] = v + w
        return 0.0


def hsde_sets(model, direct=False):
    """get_sets_and_status: HSDE(model; direct=false)   HSDE.jl:7-29  ->  (S1, S2, N); a FeasibilityModel hands over its own two
    sets (Feasibility.jl:73-80 -- the method of the same generic function for that model type)."""
    if isinstance(model, FeasibilityModel):
        return model.S1, model.S2, model.n
    Q = HSDEMatrixQ(model.A, model.b, model.c, space=model.space)           # :17
    l = Q.shape[0]
    if direct:
        S1 = IndAffineDirect(Q)                                             # :12-15
    else:
        S1 = AffinePlusLinear(Q, np.zeros(l), np.zeros(l), 1, decreasing_accuracy=True)   # :22
    S2 = DualConeProduct(model.K1, model.K2)                                # :24
    return S1, S2, 2 * l                                                    # :28


def hsde_initialvalue(model):
    """HSDE_getinitialvalue   HSDE.jl:40-47."""
    m, n = model.A.shape
    l = m + n + 1
    x = np.zeros(2 * l)
    x[l - 1] = 1.0
    x[2 * l - 1] = 1.0
    return x


def hsde_populatesolution(model, z, status):
    """HSDE_populatesolution   HSDE.jl:49-61  ->  (x, y, s, status)."""
    m, n = model.A.shape
    l = m + n + 1
    assert z.shape[0] == 2 * l
    tau = z[l - 1]
    endstatus = status.status
    if endstatus == "Continue":
        endstatus = "Indeterminate"
    with np.errstate(divide="ignore", invalid="ignore"):
        return z[0:n] / tau, z[n:n + m] / tau, z[l + n:l + n + m] / tau, endstatus


# ----------------------------------------------------------------------------------------
# HSDEStatus / checkstatus    src/problemforms/HSDE/HSDEStatus.jl:2-102
# ----------------------------------------------------------------------------------------

HEADER_CG = " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | cg  | time"     # :79-81
HEADER_DIRECT = " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | time"


def _jl_e(v, prec=2, width=9):
    """C printf '% {width}.{prec}e' (Julia @printf follows C)."""
    return "% *.*e" % (width, prec, v)


def format_status_iter(i, p, d, g, ctx, bty, kt, cgiter, t_ns):
    """printstatusiter   HSDEStatus.jl:85-91 (prints -bty as the dual objective)."""
    if cgiter is None:
        s = "%6d|% 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % .1es" % (i, p, d, g, ctx, -bty, kt, t_ns / 1e9)
    else:
        s = "%6d|% 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % 4d % .1es" % (i, p, d, g, ctx, -bty, kt, cgiter, t_ns / 1e9)
    return s.replace("inf", "Inf").replace("nan", "NaN")           # @printf's spelling


def residuals(model, z):
    """The scalars checkstatus derives from z   HSDEStatus.jl:33-38,58-61.
    Returns dict(p, d, g, ctx, bty, kappa, tau, nAxs, nATy, nb, nc)."""
    m, n = model.A.shape
    nu = n + m + 1
    x = z[0:n]
    y = z[n:n + m]
    r = z[nu:nu + n]
    s = z[nu + n:nu + n + m]
    tau = z[nu - 1]
    kappa = z[2 * nu - 1]
    A, b, c = model.A, model.b, model.c
    sp_ = getattr(model, "space", LOCAL)
    normx = lambda v: float(np.sqrt(sp_.dotX(v, v)))         # norms over the (possibly sharded) column / row space
    normy = lambda v: float(np.sqrt(sp_.dotY(v, v)))
    nb = normy(b)
    nc = normx(c)
    Ax = A @ x
    ATy = sp_.sumX(A.T @ y)
    with np.errstate(divide="ignore", invalid="ignore"):      # Julia: x/0.0 -> Inf/NaN silently
        p = normy(Ax / tau + s / tau - b) / abs(1 + nb)                           # :34
        d = normx(ATy / tau + c - r / tau) / abs(1 + nc)                          # :35
        ctx = float(sp_.dotX(c, x))                                               # :36
        bty = float(sp_.dotY(b, y))                                               # :37
        g = abs(ctx / tau + bty / tau) / (1 + abs(ctx / tau) + abs(bty / tau))    # :38
    return dict(p=p, d=d, g=g, ctx=ctx, bty=bty, kappa=float(kappa), tau=float(tau),
                nAxs=normy(Ax + s), nATy=normx(ATy), nb=nb, nc=nc)


def decide_status(res, eps):
    """Status decision   HSDEStatus.jl:53-63 (doubly normalised, reproduced as is)."""
    p, d, g, ctx, bty, tau = res["p"], res["d"], res["g"], res["ctx"], res["bty"], res["tau"]
    nb, nc = res["nb"], res["nc"]
    if p <= eps * (1 + nb) and d <= eps * (1 + nc) and g <= eps * (1 + abs(ctx / tau) + abs(bty / tau)):
        return "Optimal"
    # Julia: x/0.0 = +-Inf / NaN silently; comparisons with NaN are false
    with np.errstate(divide="ignore", invalid="ignore"):
        ub = np.float64(-ctx) / np.float64(nc)
        if res["nAxs"] <= eps * ub:
            return "Unbounded"
        ib = np.float64(-bty) / np.float64(nb)
        if res["nATy"] <= eps * ib:
            return "Infeasible"
    return "Continue"


class HSDEStatus:
    """HSDEStatus.jl:2-16 + checkstatus :27-71."""

    def __init__(self, model, checki, eps, verbose, debug, S1=None, out=None):
        self.model = model
        self.i = 0
        self.status = "Continue"
        self.checki = checki
        self.eps = eps
        self.verbose = verbose
        self.checked = False
        self.direct = False
        self.debug = debug
        self.init_time = time.perf_counter_ns()
        self.S1 = S1
        self.lines = [] if out is None else out

    def _println(self, s):
        self.lines.append(s)

    def printstatusheader(self):                           # :73-83
        if self.verbose > 0:
            self._println("Time to initialize: %ss" % (self.model.init_duration / 1e9))
            width = 76 + (0 if self.direct else 5)
            self._println("-" * width)
            self._println(HEADER_DIRECT if self.direct else HEADER_CG)
            self._println("-" * width)

    def checkstatus(self, z, override=False):              # :27-71
        if self.i % self.checki == 0 or override:
            t = time.perf_counter_ns() - self.init_time
            model, i = self.model, self.i
            res = residuals(model, z)
            if self.debug > 0:                             # :39-41, savedata :125-139
                h = model.history
                for key in ("p", "d", "g", "ctx", "bty", "kappa", "tau"):
                    h.setdefault(key, []).append((i, res[key]))
                h.setdefault("t", []).append((i, t))
                if self.debug > 1:
                    m, n = model.A.shape
                    nu = n + m + 1
                    h.setdefault("x", []).append((i, z[0:n].copy()))
                    h.setdefault("y", []).append((i, z[n:n + m].copy()))
                    h.setdefault("s", []).append((i, z[nu + n:nu + n + m].copy()))
            if self.verbose > 0:                           # :43-51
                if not self.direct:
                    cgiter = self.S1.getcgiter() if self.S1 is not None else 0
                    model.history.setdefault("cgiter", []).append((i, cgiter))
                else:
                    cgiter = None                          # :48-50: no cg column, no :cgiter history
                self._println(format_status_iter(i, res["p"], res["d"], res["g"], res["ctx"], res["bty"],
                                                 _ieee_div(res["kappa"], res["tau"]), cgiter, t))
            status = decide_status(res, self.eps)
            if status == "Optimal" and self.verbose > 0:   # :55-57
                self._println("Found solution i=%d" % i)
            self.status = status
            self.checked = True
            self.last = res
            return True
        self.checked = False
        return False


# ----------------------------------------------------------------------------------------
# Algorithms     src/solvers/{gap,gapa,fista,dykstra,solvers}.jl
# ----------------------------------------------------------------------------------------

class GAP:
    """GAP(alpha=0.8, alpha1=1.8, alpha2=1.8)   gap.jl:6-13."""

    def __init__(self, alpha=0.8, alpha1=1.8, alpha2=1.8, **options):
        self.alpha, self.alpha1, self.alpha2 = alpha, alpha1, alpha2
        self.direct = bool(options.pop("direct", False))    # the positional `direct` field of gap.jl:10
        self.options = options

    def init(self, model):                                  # gap.jl:23-28
        self.S1, self.S2, n = hsde_sets(model, self.direct)
        self.tmp1 = np.empty(n)
        self.tmp2 = np.empty(n)

    def S1_(self, y, x, longstep=None):                     # gap.jl:42-51
        self.S1.prox(y, x)
        addprojeq(longstep, y, x)                           # :47
        y[:] = self.alpha1 * y + (1 - self.alpha1) * x

    def S2_(self, y, x, status, longstep=None):             # gap.jl:53-59
        self.S2.prox(y, x)
        status.checkstatus(y)
        addprojineq(longstep, y, x)                         # :57
        y[:] = self.alpha2 * y + (1 - self.alpha2) * x

    def step(self, x, i, status, longstep=None):            # gap.jl:61-80
        self.S1_(self.tmp1, x, longstep)
        self.S2_(self.tmp2, self.tmp1, status, longstep)
        x[:] = self.alpha * self.tmp2 + (1 - self.alpha) * x

    def getsol(self, x):                                    # gap.jl:82-87
        self.S1.prox(self.tmp1, x)
        self.S2.prox(self.tmp2, self.tmp1)
        return self.tmp2


def DR(alpha=0.5, **kw):                                    # solvers.jl:10
    return GAP(alpha, 2.0, 2.0, **kw)


def AP(alpha=1, **kw):                                      # solvers.jl:11
    return GAP(alpha, 1.0, 1.0, **kw)


def normed_scalar(x1, x2, y1, y2, space=LOCAL):
    """normedScalar   gapa.jl:36-47."""
    d1 = x1 - x2
    d2 = y1 - y2
    s = float(space.dotN(d1, d2))
    n1 = float(space.dotN(d1, d1))
    n2 = float(space.dotN(d2, d2))
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(abs(s)) / np.sqrt(np.float64(n1 * n2)))


class GAPA:
    """GAPA(alpha=1.0, beta=0.0)   gapa.jl:9-15."""

    def __init__(self, alpha=1.0, beta=0.0, **options):
        self.alpha, self.beta = alpha, beta
        self.direct = bool(options.pop("direct", False))
        self.options = options

    def init(self, model):                                  # gapa.jl:27-32
        self.S1, self.S2, n = hsde_sets(model, self.direct)
        self.alpha12 = 2.0
        self.tmp1 = np.empty(n)
        self.tmp2 = np.empty(n)

    def S1_(self, y, x, longstep=None):                     # gapa.jl:61-70
        a12 = self.alpha12
        self.S1.prox(y, x)
        addprojeq(longstep, y, x)                           # :66
        y[:] = a12 * y + (1 - a12) * x

    def S2_(self, y, x, status, longstep=None):             # gapa.jl:72-78
        a12 = self.alpha12
        self.S2.prox(y, x)
        status.checkstatus(y)
        addprojineq(longstep, y, x)                         # :76
        y[:] = a12 * y + (1 - a12) * x

    def step(self, x, i, status, longstep=None):            # gapa.jl:80-105
        self.S1_(self.tmp1, x, longstep)
        self.S2_(self.tmp2, self.tmp1, status, longstep)
        scl = normed_scalar(self.tmp2, self.tmp1, self.tmp1, x, getattr(getattr(self.S1, "A", None), "space", LOCAL))     # :96
        scl = 0.0 if math.isnan(scl) else min(max(scl, 0.0), 1.0)   # :96-97 (clamp then NaN -> 0)
        s = math.sqrt(1 - scl ** 2)                         # :98
        aopt = 2 / (1 + s)                                  # :100
        self.alpha12 = (1 - self.beta) * aopt + self.beta * 2.0     # :101
        x[:] = self.alpha * self.tmp2 + (1 - self.alpha) * x        # :103

    def getsol(self, x):                                    # gapa.jl:107-112
        self.S1.prox(self.tmp1, x)
        self.S2.prox(self.tmp2, self.tmp1)
        return self.tmp2


class FISTA:
    """FISTA(alpha=1.0)   fista.jl:6-11."""

    def __init__(self, alpha=1.0, **options):
        self.alpha = alpha
        self.direct = bool(options.pop("direct", False))
        self.options = options

    def init(self, model):                                  # fista.jl:20-25
        self.S1, self.S2, n = hsde_sets(model, self.direct)
        self.t = 1.0
        self.y = np.zeros(n)
        self.xold = np.zeros(n)
        self.tmp1 = np.empty(n)

    def step(self, x, i, status, longstep=None):            # fista.jl:28-48
        if i == 1:
            self.y[:] = x                                   # :31-33
        self.S1.prox(self.tmp1, self.y)                     # :35
        addprojeq(longstep, self.tmp1, self.y)              # :36
        self.tmp1[:] = self.alpha * self.tmp1 + (1 - self.alpha) * self.y   # :37
        self.xold[:] = x                                    # :39
        self.S2.prox(x, self.tmp1)                          # :40
        status.checkstatus(x)                               # :41
        addprojineq(longstep, x, self.tmp1)                 # :42
        told = self.t
        self.t = (1 + math.sqrt(1 + 4 * self.t ** 2)) / 2   # :45
        self.y[:] = x + (told - 1) / self.t * (x - self.xold)   # :46

    def getsol(self, x):                                    # fista.jl:50-56
        tmp2 = np.empty_like(x)
        self.S1.prox(self.tmp1, x)
        self.S2.prox(tmp2, self.tmp1)
        return tmp2


class GAPP:
    """GAPP(alpha=0.8, alpha1=1.8, alpha2=1.8; direct=true, iproj=100)   gapproj.jl:5-13 -- "projected GAP": GAP whose every iproj-th
    iteration searches 21 step lengths 2^k along P_S1(P_S2(P_S1 x)) - P_S1 x (gapproj.jl:29-79).  Experimental in the reference
    (it prints inside the loop); used by test/testfeasibility.jl:36-44, restated for the Feasibility form."""

    def __init__(self, alpha=0.8, alpha1=1.8, alpha2=1.8, iproj=100, direct=True, out=None, **options):
        self.alpha, self.alpha1, self.alpha2, self.iproj, self.direct = alpha, alpha1, alpha2, iproj, direct
        self.options, self.out = options, out
        self.log = []

    def _println(self, s):
        if self.out is None:
            print(s)
        else:
            self.out.append(s)

    def init(self, model):                                  # :22-26
        self.S1, self.S2, n = hsde_sets(model, self.direct)
        self.tmp1, self.tmp2 = np.empty(n), np.empty(n)

    def step(self, x, i, status):                           # :29-72
        tmp1, tmp2 = self.tmp1, self.tmp2
        self.S1.prox(tmp1, x)                               # :33
        if i % self.iproj == 0:                             # :34
            tmp3, tmp4, res = np.empty_like(tmp1), np.empty_like(tmp1), np.empty_like(tmp1)
            self.S2.prox(tmp2, tmp1)                        # :39
            self.S1.prox(res, tmp2)                         # :40
            res[:] = res - tmp1                             # :41
            normbest, abest, tests = math.inf, -1.0, []
            for k in range(21):                             # :46
                atest = 2.0 ** k
                tmp3[:] = tmp1 + atest * res
                self.S2.prox(tmp4, tmp3)
                normtest = float(np.linalg.norm(tmp4 - tmp3))
                tests.append(normtest)
                self._println("normtest: %s" % julia_float(normtest))
                if normtest < normbest:
                    abest, normbest = atest, normtest
            self._println("\u03b1best: %s" % julia_float(abest))
            tmp1[:] = tmp1 + abest * res                    # :58
            self.S2.prox(tmp2, tmp1)                        # :59
            status.checkstatus(tmp2)
            tmp2[:] = self.alpha2 * tmp2 + (1 - self.alpha2) * tmp1
            x[:] = tmp2                                     # :62
            self.log.append((i, tests, abest))
        else:
            tmp1[:] = self.alpha1 * tmp1 + (1 - self.alpha1) * x        # :64
            self.S2.prox(tmp2, tmp1)
            status.checkstatus(tmp2)
            tmp2[:] = self.alpha2 * tmp2 + (1 - self.alpha2) * tmp1
            x[:] = self.alpha * tmp2 + (1 - self.alpha) * x             # :70

    def getsol(self, x):                                    # :76-81
        self.S1.prox(self.tmp1, x)
        self.S2.prox(self.tmp2, self.tmp1)
        return self.tmp2


class Dykstra:
    """Dykstra()   dykstra.jl:5-9."""

    def __init__(self, **options):
        self.direct = bool(options.pop("direct", False))
        self.options = options

    def init(self, model):                                  # dykstra.jl:19-23
        self.S1, self.S2, n = hsde_sets(model, self.direct)
        self.p = np.zeros(n)
        self.q = np.zeros(n)
        self.y = np.empty(n)

    def step(self, x, i, status, longstep=None):            # dykstra.jl:25-36
        self.S1.prox(self.y, x + self.p)
        addprojeq(longstep, self.y, x + self.p)             # :30
        self.p[:] = x + self.p - self.y
        self.S2.prox(x, self.y + self.q)
        status.checkstatus(x)
        addprojineq(longstep, x, self.y + self.q)           # :34
        self.q[:] = self.y + self.q - x

    def getsol(self, x):                                    # dykstra.jl:38-44
        tmp1 = np.empty_like(x)
        self.S1.prox(tmp1, x)
        self.S2.prox(self.y, tmp1)
        return self.y


# ----------------------------------------------------------------------------------------
# LineSearchWrapper      src/wrappers/linesearch.jl ; NoStatus  src/status.jl:3-11
# ----------------------------------------------------------------------------------------


def _ieee_div(a, b):
    """a / b as Julia evaluates it (Inf / NaN instead of an exception when tau is still 0 at an early check)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(a) / np.float64(b))


class NoStatus:
    """checkstatus / logextra / printstatusheader are no-ops (status.jl:3-11)."""
    status = "Continue"

    def checkstatus(self, z, override=False):
        return False


def julia_float(x):
    """How Julia's println shows a Float64 (shortest round-trip digits; exponent form outside 1e-4 <= |x| < 1e6)."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    sci = np.format_float_scientific(x, unique=True, trim="0")       # e.g. 1.234567e+06 / 1.e-05 -> mantissa with .0
    mant, ex = sci.split("e")
    ex = int(ex)
    if -4 <= ex < 6:
        return np.format_float_positional(x, unique=True, trim="0")
    if mant.endswith("."):
        mant += "0"
    return "%se%d" % (mant, ex)


class LineSearchWrapper:
    """LineSearchWrapper(alg; lsinterval=100, kwargs...)   linesearch.jl:3-24.  Every lsinterval-th iteration: one S2!(S1!(x))
    evaluation with the real status, then 31 trial step lengths 0.1*1.8^(k+1) along res = S2(S1(x)) - x scored by
    ||x_trial - S2(S1(x_trial))|| with NoStatus; the best one is taken (linesearch.jl:36-75).  The reference's println calls
    are reproduced through `out` (default: stdout)."""

    def __init__(self, alg, lsinterval=100, out=None, **options):
        if not isinstance(alg, (GAP, GAPA)):                 # support_linesearch: gap.jl:89, gapa.jl:117, defaults.jl:22
            raise ValueError("Algorithm %s does not support line search" % type(alg).__name__)
        self.alg, self.lsinterval, self.out = alg, lsinterval, out
        self.options = {**alg.options, **options}           # merge(alg.options, kwargs)   :23
        self.direct = alg.direct
        self.log = []                                        # (i, normres, [testres...], alpha_best) per search

    def _println(self, s):                                  # (same sink convention as HSDEStatus: a list, or stdout)
        if self.out is None:
            print(s)
        else:
            self.out.append(s)

    def init(self, model):                                  # :26-32
        self.alg.init(model)
        self.S1, self.S2 = self.alg.S1, self.alg.S2
        n = self.alg.tmp1.size
        self.tmp1, self.tmp2, self.tmp3, self.res = np.empty(n), np.empty(n), np.empty(n), np.empty(n)

    def step(self, x, i, status):                           # :36-75
        if i % self.lsinterval != 0:                        # :39
            self.alg.step(x, i, status)                     # :73
            return
        tmp1, tmp2, tmp3, res = self.tmp1, self.tmp2, self.tmp3, self.res
        tmp1[:] = x                                         # :41
        self.alg.S1_(tmp2, x)                               # :45
        self.alg.S2_(x, tmp2, status)                       # :46
        nostatus = NoStatus()                               # :48
        res[:] = x - tmp1                                   # :49
        normres = float(np.linalg.norm(res))                # :50
        self._println("test, %s" % julia_float(normres))    # :51
        best, abest, a = math.inf, 1.0, 0.1                 # :53-55
        tests = []
        for k in range(31):                                 # :56
            a = a * 1.8                                     # :57
            x[:] = tmp1 + a * res                           # :58
            self.alg.S1_(tmp2, x)                           # :60
            self.alg.S2_(tmp3, tmp2, nostatus)              # :61
            d = x - tmp3                                    # normdiff  :62,77-85
            testres = math.sqrt(float(np.dot(d, d)))
            tests.append(testres)
            self._println("\u03b1: %s, %s" % (julia_float(a), julia_float(testres)))   # :63
            if testres < best:                              # :64-67
                best, abest = testres, a
        self._println("\u03b1: %s" % julia_float(abest))    # :69
        x[:] = tmp1 + abest * res                           # :70
        self.log.append((i, normres, tests, abest))

    def getsol(self, x):                                    # :93-95
        return self.alg.getsol(x)


# ----------------------------------------------------------------------------------------
# LongstepWrapper        src/wrappers/longstep.jl, src/wrappers/saveplanes.jl
# ----------------------------------------------------------------------------------------


class SavedPlanes:
    """mutable struct SavedPlanes (saveplanes.jl:5-11) and its constructor (:30-33): (n + 1) neq + (n + 1) nineq rows (uninitialised in the
    reference; every row is written before the first projection reads it when longinterval >= nsave + 1)."""

    def __init__(self, x, n, neq, nineq):
        total = (n + 1) * neq + (n + 1) * nineq
        self.A = np.full((total, x.size), np.nan)
        self.b = np.full(total, np.nan)
        self.n, self.neq, self.nineq = n, neq, nineq


def project_onto_planes(A, b, C, d, x, tol=1e-12):
    """argmin 1/2 |v|^2 - x'v  s.t.  A v = b, C v >= d  -- what saveplanes.jl:24-26 hands to QPDAS (a package outside the checkout) as
    QuadraticProgram(BigFloat.(A), BigFloat.(b), BigFloat.(-C), BigFloat.(-d), BigFloat.(-x), I, eps = 1e-12): the projection of x onto a polyhedron,
    solved in EXTENDED precision because the saved normals of successive iterations are nearly dependent (measured on the README NNLS with the
    default nsave = 10: cond(P) 1e8 .. 1e10; a float64 solve through G = P P' is then off by 1e-4, ~1 % of the step).  The solution is unique,
    whatever solves it.  Here, in 60-digit arithmetic (mpmath) on the float64 data, as the reference: v = x + A'lam + C'mu, mu >= 0; the support
    of mu is found by enumeration (smallest KKT violation), each candidate by the pseudo-inverse of its Gram block (eigenvalues below 1e-26 of
    the trace dropped: dependent planes).  Test infrastructure: tests/test_longstep_oracle.py checks it against scipy's SLSQP on the primal."""
    import itertools
    import mpmath as mp
    with mp.workdps(60):
        P = np.vstack([A, C])
        neq, nin = A.shape[0], C.shape[0]
        K = neq + nin
        Pm = mp.matrix(P.tolist()) if K else mp.matrix(0, len(x))
        xm = mp.matrix([float(t) for t in x])
        beta = mp.matrix([float(t) for t in np.concatenate([b, d])]) if K else mp.matrix(0, 1)
        G = Pm * Pm.T
        c = beta - Pm * xm
        scale = max([abs(c[a]) + mp.sqrt(G[a, a]) for a in range(K)] + [mp.mpf("1e-300")])
        best = (mp.inf, mp.zeros(K, 1))
        for r in range(nin + 1):
            for S in itertools.combinations(range(nin), r):
                F = list(range(neq)) + [neq + j for j in S]
                nu = mp.zeros(K, 1)
                if F:
                    nf = len(F)
                    GF = mp.matrix(nf, nf)
                    cF = mp.matrix(nf, 1)
                    for a_, i_ in enumerate(F):
                        cF[a_] = c[i_]
                        for b_, j_ in enumerate(F):
                            GF[a_, b_] = G[i_, j_]
                    E, Q = mp.eigsy(GF)
                    tr = sum(abs(e) for e in E)
                    sol = mp.zeros(nf, 1)
                    for k in range(nf):
                        if E[k] > tr * mp.mpf("1e-26"):
                            q = Q[:, k]
                            sol += q * ((q.T * cF)[0] / E[k])
                    for a_, i_ in enumerate(F):
                        nu[i_] = sol[a_]
                g = G * nu - c                                  # P v - beta
                viol = max([abs(g[a]) for a in range(neq)] + [mp.mpf(0)])
                for j_ in range(nin):
                    if j_ in S:
                        viol = max(viol, abs(g[neq + j_]), -nu[neq + j_] * mp.sqrt(max(G[neq + j_, neq + j_], mp.mpf("1e-300"))))
                    else:
                        viol = max(viol, -g[neq + j_])
                if viol < best[0]:
                    best = (viol, nu)
                if best[0] <= tol * scale:
                    break
            if best[0] <= tol * scale:
                break
        v = xm + Pm.T * best[1] if K else xm
        return np.array([float(t) for t in v]), float(best[0])


def projectonnormals(s, x, y):
    """projectonnormals!(s, x, y)   saveplanes.jl:13-28 -- the FIRST neq (n + 1) rows are taken as equalities and the rest as inequalities,
    although the rows were saved equality, inequality, equality, ... (longstep.jl:69,88): the reference's behaviour, kept."""
    ne = s.neq * (s.n + 1)
    A, b = s.A[:ne], s.b[:ne]
    C, d = s.A[ne:(s.neq + s.nineq) * (s.n + 1)], s.b[ne:(s.neq + s.nineq) * (s.n + 1)]
    y[:], viol = project_onto_planes(A, b, C, d, x)
    return False, viol


def addprojeq(long, y, x):                                  # longstep.jl:62,65-79
    if long is None:
        return
    if long.savepos > 0:
        s = long.saved
        i = (long.savepos - 1) * (s.neq + s.nineq * long.saveineq) + long.eqi       # (0-based; the reference's i is this + 1)
        s.A[i, :] = x - y
        s.b[i] = float(np.dot(x - y, y))
        long.eqi += 1


def addprojineq(long, y, x):                                # longstep.jl:63,81-97
    if long is None:
        return
    if long.savepos > 0 and (long.saveineq or long.savepos == long.nsave + 1):
        s = long.saved
        i = (long.savepos - 1) * (s.neq + s.nineq * long.saveineq) + long.eqi + long.uneqi
        s.A[i, :] = x - y
        s.b[i] = float(np.dot(x - y, y))
        long.uneqi += 1


class LongstepWrapper:
    """LongstepWrapper(alg; longinterval=100, nsave=10, kwargs...)   longstep.jl:5-24 (+ LongstepWrapperData :12-22 in the same object)."""

    def __init__(self, alg, longinterval=100, nsave=10, **options):
        if not isinstance(alg, (GAP, GAPA, FISTA, Dykstra)) or isinstance(alg, GAPP):      # support_longstep: gap.jl:91, gapa.jl:114, fista.jl:58, dykstra.jl:47
            raise ValueError("Algorithm alg does not support longstep")                    # longstep.jl:28 (@error)
        self.alg, self.longinterval, self.nsave = alg, longinterval, nsave
        self.options = {**options, **alg.options}           # [kwargs..., alg.options...]   :23
        self.direct = alg.direct
        self.log = []                                        # (i, KKT violation of the projection) per projection

    def init(self, model):                                  # longstep.jl:26-38
        self.alg.init(model)
        self.S1, self.S2 = self.alg.S1, self.alg.S2          # (what solve() hands to the status object)
        neq, nineq = 1, 1                                   # projections_per_step: (1, 1) for the four algorithms
        x = hsde_initialvalue(model) if not isinstance(model, FeasibilityModel) else np.zeros(model.n)
        self.saved = SavedPlanes(x, self.nsave, neq, nineq)
        self.saveineq, self.savepos, self.eqi, self.uneqi = True, 0, 1, 1
        self.tmp = np.empty_like(x)

    def step(self, x, i, status):                           # longstep.jl:41-59
        savepos = (i - 1) % self.longinterval - self.longinterval + self.nsave + 2     # :45
        if 0 < savepos:                                     # :47-50
            self.savepos = savepos
            self.eqi, self.uneqi = 0, 0
        self.alg.step(x, i, status, self)                   # :51
        if self.savepos == self.nsave + 1:                  # :53-58
            _, viol = projectonnormals(self.saved, x, self.tmp)
            self.savepos = -1
            x[:] = self.tmp
            self.log.append((i, viol))

    def getsol(self, x):                                    # :61-63
        return self.alg.getsol(x)


# ----------------------------------------------------------------------------------------
# solve! / iterate       src/solverwrapper.jl:2-41, src/FOSSolverInterface.jl:8-21
# ----------------------------------------------------------------------------------------


def iterate(alg, status, x, max_iters):
    """iterate   solverwrapper.jl:20-41."""
    t1 = time.time()
    status.printstatusheader()
    for i in range(1, max_iters + 1):
        status.i = i
        alg.step(x, i, status)
        if status.status != "Continue":
            break
    guess = alg.getsol(x)
    if not status.checked:
        status.checkstatus(guess, override=True)
    if status.verbose > 0:
        status._println("Time for iterations: ")
        status._println("%s s" % (time.time() - t1))
    return guess


class Solution:
    def __init__(self, x, y, s, status):
        self.x, self.y, self.s, self.status = x, y, s, status


def solve(model, alg, out=None):
    """loadproblem! + optimize! for an already conic-form model.
    Option defaults: solverwrapper.jl:5-10 (max_iters=10000, verbose=1, debug=1, eps=1e-5, checki=100)."""
    opts = dict(alg.options)
    max_iters = opts.get("max_iters", 10000)
    verbose = opts.get("verbose", 1)
    debug = opts.get("debug", 1)
    eps = opts.get("eps", 1e-5)
    checki = opts.get("checki", 100)
    alg.init(model)                                          # init_algorithm!  (loadproblem! :58)
    x = opts["initx"] if "initx" in opts else hsde_initialvalue(model)
    model.history = {}                                       # optimize! :10
    status = HSDEStatus(model, checki, eps, verbose, debug, S1=alg.S1, out=out)
    status.direct = bool(getattr(alg, "direct", False))      # HSDE.jl:27
    guess = iterate(alg, status, x, max_iters)
    xs, ys, ss, st = hsde_populatesolution(model, guess, status)
    sol = Solution(xs, ys, ss, st)
    sol.obj_val = float(np.dot(model.c, xs))                 # optimize! :20
    sol.iterations = status.i
    sol.status_obj = status
    sol.z = guess
    return sol


# ----------------------------------------------------------------------------------------
# Feasibility form      src/problemforms/Feasibility/Feasibility.jl, FeasibilityStatus.jl
# (SURVEY 8(f) rank 4).  The two sets are ProximalOperators objects in the reference; restated here: the two its own test uses
# (test/testfeasibility.jl:9-10), IndAffine(A, b) and IndBox(lo, hi).  Their projections are unique, so "what the package computes"
# is pinned by the definition: x - A'(A A')^-1 (A x - b) and clamp(x, lo, hi).
# ----------------------------------------------------------------------------------------
class IndAffine:
    """ProximalOperators.IndAffine(A, b): indicator of {x : A x = b}, A with full row rank (dense: QR in the package)."""

    def __init__(self, A, b):
        import scipy.linalg
        self.A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        self.b = np.asarray(b, dtype=np.float64).copy()
        assert self.A.ndim == 2 and self.A.shape[0] == self.b.shape[0]
        self.chol = scipy.linalg.cho_factor(self.A @ self.A.T)

    def prox(self, y, x):
        import scipy.linalg
        y[:] = x - self.A.T @ scipy.linalg.cho_solve(self.chol, self.A @ x - self.b)
        return 0.0


class IndBox:
    """ProximalOperators.IndBox(lo, hi): indicator of {x : lo <= x <= hi} (scalars or vectors; +-inf allowed)."""

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def prox(self, y, x):
        y[:] = np.minimum(np.maximum(x, self.lo), self.hi)
        return 0.0


class Feasibility:
    """struct Feasibility   Feasibility.jl:2-6."""

    def __init__(self, S1, S2, n):
        self.S1, self.S2, self.n = S1, S2, int(n)


class FeasibilitySolution:
    """FeasibilitySolution   Feasibility.jl:8-11."""

    def __init__(self, x, status):
        self.x, self.status = x, status


class FeasibilityModel:
    """FeasibilityModel   Feasibility.jl:15-50 (the fields the path reads)."""

    def __init__(self, problem, alg, **kwargs):
        self.S1, self.S2, self.n = problem.S1, problem.S2, problem.n
        self.alg = alg
        self.options = dict(alg.options)
        self.options.update(kwargs)                           # kwargs of solve! override the algorithm's   :37-41
        self.solve_stat = "NotSolved"
        self.history = {}
        alg.init(self)                                        # init_algorithm!   :46


class FeasibilityStatus:
    """FeasibilityStatus + checkstatus   FeasibilityStatus.jl:1-72."""

    def __init__(self, model, checki, eps, verbose, debug, out=None):
        self.n, self.i, self.model = model.n, 0, model
        self.prev = np.full(model.n, np.nan)                  # Feasibility.jl:78  fill(NaN, n)
        self.status = "Continue"
        self.checki, self.eps, self.verbose, self.debug = checki, eps, verbose, debug
        self.checked = False
        self.direct = True                                    # Feasibility.jl:74
        self.out = out
        self.err = float("nan")

    def _println(self, s):
        if self.out is not None:
            self.out.append(s)
        else:
            print(s)

    def printstatusheader(self):                              # :74-84
        if self.verbose > 0:
            self._println("Time to initialize: s")
            width = 22 + (0 if self.direct else 5)
            self._println("-" * width)
            self._println(" Iter | res" + ("" if self.direct else " | cg ") + " | time")
            self._println("-" * width)

    def checkstatus(self, z, override=False):                 # :32-72
        if self.i % self.checki == 0 or override:
            err = float(np.linalg.norm(self.prev - z))        # :40
            self.err = err
            if self.debug > 0:                                # savedata :95-103
                self.model.history.setdefault("err", []).append((self.i, err))
            if self.verbose > 0:
                self._println("%6d|%s" % (self.i, _jl_e(err)))
            status = "Continue"
            if err <= self.eps:                               # :57 (NaN <= eps is false: the first check can never stop the solve)
                if self.verbose > 0:
                    self._println("Found solution i=%d" % self.i)
                status = "Optimal"
            self.status = status
            self.checked = True
            self.prev[:] = z
            return True
        self.checked = False
        self.prev[:] = z                                      # :69 (every call)
        return False


def feasibility_solve(problem, alg, out=None, **kwargs):
    """solve!(problem::Feasibility, alg; kwargs...)   Feasibility.jl:52-56 -> (solution, model); the loop is solve!(model) of
    solverwrapper.jl:2-17 with getinitialvalue = zeros(n) (Feasibility.jl:58) and populate_solution of :61-68."""
    model = FeasibilityModel(problem, alg, **kwargs)
    opts = model.options
    max_iters = opts.get("max_iters", 10000)
    verbose = opts.get("verbose", 1)
    debug = opts.get("debug", 1)
    eps = opts.get("eps", 1e-5)
    checki = opts.get("checki", 100)
    x = np.zeros(model.n)
    status = FeasibilityStatus(model, checki, eps, verbose, debug, out=out)
    guess = iterate(alg, status, x, max_iters)
    endstatus = "Indeterminate" if status.status == "Continue" else status.status
    model.solve_stat = endstatus
    sol = FeasibilitySolution(np.array(guess, copy=True), endstatus)
    sol.iterations = status.i
    sol.err = status.err
    return sol, model
