"""
Host-side mirror of the reference's solver interface for the HSDE hot path.

Same names, argument meaning and behaviour as the Julia reference (paths under the reference checkout):

  GAP / DR / AP / GAPA / FISTA / Dykstra constructors   src/solvers/gap.jl:13, solvers.jl:10-11, gapa.jl:15,
                                                        fista.jl:11, dykstra.jl:9
  FOSMathProgModel + loadproblem! / optimize! / status / getobjval / getsolution / numvar / numconstr
                                                        src/types.jl:30-60, src/FOSSolverInterface.jl:5-69
  solve! option handling, iterate                       src/solverwrapper.jl:2-41
  HSDEStatus printing, history keys                     src/problemforms/HSDE/HSDEStatus.jl:73-91,125-139
  HSDE_populatesolution                                 src/problemforms/HSDE/HSDE.jl:49-61

Every numerical operation is a call into libfoship.so (HIP, gfx950) through `_lib`; nothing here computes on
the CPU beyond scalar bookkeeping, and there is no fallback when the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import time

import math

import numpy as np
import scipy.sparse as sp

from . import _lib

HEADER_CG = " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | cg  | time"       # HSDEStatus.jl:79-81
HEADER_DIRECT = " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | time"


# ---------------------------------------------------------------------------------------------- algorithms

class FOSAlgorithm:
    """abstract type FOSAlgorithm (src/types.jl:13)."""
    direct = False
    options: dict

    def _alg_args(self):
        raise NotImplementedError


class GAP(FOSAlgorithm):
    """GAP(alpha=0.8, alpha1=1.8, alpha2=1.8; direct=false, kwargs...)   gap.jl:6-13"""

    def __init__(self, alpha=0.8, alpha1=1.8, alpha2=1.8, direct=False, **kwargs):
        self.alpha, self.alpha1, self.alpha2, self.direct, self.options = alpha, alpha1, alpha2, direct, kwargs

    def _alg_args(self):
        return (_lib.ALG_GAP, self.alpha, self.alpha1, self.alpha2, 0.0)


def DR(alpha=0.5, **kwargs):
    """DR(alpha=0.5) = GAP(alpha, 2.0, 2.0)   solvers.jl:10"""
    return GAP(alpha, 2.0, 2.0, **kwargs)


def AP(alpha=1, **kwargs):
    """AP(alpha=1) = GAP(alpha, 1.0, 1.0)   solvers.jl:11"""
    return GAP(alpha, 1.0, 1.0, **kwargs)


class GAPA(FOSAlgorithm):
    """GAPA(alpha=1.0, beta=0.0; direct=false, kwargs...)   gapa.jl:9-15"""

    def __init__(self, alpha=1.0, beta=0.0, direct=False, **kwargs):
        self.alpha, self.beta, self.direct, self.options = alpha, beta, direct, kwargs

    def _alg_args(self):
        return (_lib.ALG_GAPA, self.alpha, 0.0, 0.0, self.beta)


class FISTA(FOSAlgorithm):
    """FISTA(alpha=1.0; direct=false, kwargs...)   fista.jl:6-11"""

    def __init__(self, alpha=1.0, direct=False, **kwargs):
        self.alpha, self.direct, self.options = alpha, direct, kwargs

    def _alg_args(self):
        return (_lib.ALG_FISTA, self.alpha, 0.0, 0.0, 0.0)


class GAPP(FOSAlgorithm):
    """GAPP(alpha=0.8, alpha1=1.8, alpha2=1.8; direct=true, iproj=100, kwargs...)   gapproj.jl:5-13, the last row of the reference's
    solver table (README.md:32-40): GAP whose every iproj-th iteration is a 21-point projected search.  On the HSDE path
    (fos_set_gapp) and on the Feasibility form (fos_feas_set_gapp); `direct` defaults to true as in the reference (dense, l <= 46 000)."""

    def __init__(self, alpha=0.8, alpha1=1.8, alpha2=1.8, direct=True, iproj=100, **kwargs):
        self.alpha, self.alpha1, self.alpha2, self.direct, self.iproj, self.options = alpha, alpha1, alpha2, direct, int(iproj), kwargs

    def _alg_args(self):
        return (_lib.ALG_GAP, self.alpha, self.alpha1, self.alpha2, 0.0)         # + fos_set_gapp(iproj)


class Dykstra(FOSAlgorithm):
    """Dykstra(; direct=false, kwargs...)   dykstra.jl:5-9"""

    def __init__(self, direct=False, **kwargs):
        self.direct, self.options = direct, kwargs

    def _alg_args(self):
        return (_lib.ALG_DYKSTRA, 0.0, 0.0, 0.0, 0.0)


# ---------------------------------------------------------------------------------------------- device handle

def _ieee_div(a, b):
    """a / b as Julia evaluates it (Inf / NaN instead of an exception when tau is still 0 at an early check)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(a) / np.float64(b))


def _julia_specials(s):
    """@printf shows Inf / NaN where Python's % formatting writes inf / nan."""
    return s.replace("inf", "Inf").replace("nan", "NaN")


def julia_float(x):
    """How Julia's println shows a Float64 (shortest round-trip digits; exponent form outside 1e-4 <= |x| < 1e6)."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    mant, ex = np.format_float_scientific(x, unique=True, trim="0").split("e")
    ex = int(ex)
    if -4 <= ex < 6:
        return np.format_float_positional(x, unique=True, trim="0")
    return "%se%d" % (mant + "0" if mant.endswith(".") else mant, ex)


class LineSearchWrapper(FOSAlgorithm):
    """LineSearchWrapper(alg; lsinterval=100, kwargs...)   wrappers/linesearch.jl:3-24 -- around GAP (AP, DR) or GAPA, the
    algorithms with support_linesearch == Val{:Fast}.  Every lsinterval-th iteration is a 31-point search of the step length
    along S2!(S1!(x)) - x, run on the device (fos_set_linesearch); the reference's println output of a search is reproduced
    from fos_linesearch_log."""

    def __init__(self, alg, lsinterval=100, **kwargs):
        if not isinstance(alg, (GAP, GAPA)):                 # linesearch.jl:20-22 (an @error in the reference)
            raise ValueError("Algorithm %s does not support line search" % type(alg).__name__)
        self.alg, self.lsinterval = alg, int(lsinterval)
        self.options = {**alg.options, **kwargs}            # merge(alg.options, kwargs)
        self.direct = alg.direct

    def _alg_args(self):
        return self.alg._alg_args()


class LongstepWrapper(FOSAlgorithm):
    """LongstepWrapper(alg; longinterval=100, nsave=10, kwargs...)   wrappers/longstep.jl:5-24 -- around GAP (AP, DR), GAPA, FISTA or Dykstra
    (support_longstep).  The last nsave + 1 iterations of every longinterval save the half-planes of their two projections (addprojeq /
    addprojineq); the iterate is then projected onto the saved planes (projectonnormals!, saveplanes.jl:13-35).  All on the device
    (fos_set_longstep): the planes never leave it, the projection's small dual QP is solved by the library."""

    def __init__(self, alg, longinterval=100, nsave=10, **kwargs):
        if not isinstance(alg, (GAP, GAPA, FISTA, Dykstra)) or isinstance(alg, GAPP):      # longstep.jl:28 (an @error in the reference)
            raise ValueError("Algorithm %s does not support longstep" % type(alg).__name__)
        self.alg, self.longinterval, self.nsave = alg, int(longinterval), int(nsave)
        self.options = {**kwargs, **alg.options}            # [kwargs..., alg.options...]: the wrapped algorithm's options win  longstep.jl:23
        self.direct = alg.direct

    def _alg_args(self):
        return self.alg._alg_args()


def _normalize_cones(cones, total, what):
    """(name, length) or (name, 1-based index range/list) tuples -> (types int32, starts int64 1-based, lens int64).
    Index lists must be contiguous: toRanges, src/cones.jl:44-56."""
    types, starts, lens = [], [], []
    run = 1
    for name, spec in cones:
        if name not in _lib.CONE_CODES:
            raise KeyError("Cone type %s not supported" % name)          # FOSSolverInterface.jl:37-42
        if isinstance(spec, (int, np.integer)):
            s, ln = run, int(spec)
        else:
            idx = np.asarray(list(spec), dtype=np.int64)
            if idx.size == 0 or not np.array_equal(idx, np.arange(idx[0], idx[-1] + 1)):
                raise ValueError("Invalid range in input")               # cones.jl:50
            s, ln = int(idx[0]), int(idx.size)
        types.append(_lib.CONE_CODES[name])
        starts.append(s)
        lens.append(ln)
        run = s + ln
    return (np.asarray(types, dtype=np.int32), np.asarray(starts, dtype=np.int64), np.asarray(lens, dtype=np.int64))


class HipHSDE:
    """Owns one fos_handle: the device-resident S1 (AffinePlusLinear over HSDEMatrixQ), S2 (DualConeProduct),
    iterate and algorithm data.  == what init_algorithm!/get_sets_and_status build (FOSSolverInterface.jl:76-79)."""

    def __init__(self, A, b, c, K1, K2, device=0, row_sharded=False):
        """row_sharded: this rank holds the rows of A of its K1 cones and all columns (sharding.shard_rows);
        follow with comm_init on every rank."""
        self._lib = _lib.load()
        A = sp.csc_matrix(A)                       # loadproblem! sparsifies dense input, FOSSolverInterface.jl:27-29
        A.sort_indices()
        m, n = A.shape
        self.m, self.n, self.l, self.N = m, n, m + n + 1, 2 * (m + n + 1)
        self.nnz = int(A.nnz)
        b = _lib.as_f64(b, m)
        c = _lib.as_f64(c, n)
        colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1          # Julia 1-based
        rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
        nzval = np.ascontiguousarray(A.data, dtype=np.float64)
        k1t, k1s, k1l = _normalize_cones(K1, m, "K1")
        k2t, k2s, k2l = _normalize_cones(K2, n, "K2")
        h = C.c_void_p()
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        _lib.check(self._lib.fos_create2(m, n, i64(colptr), i64(rowval), _lib.dptr(nzval), _lib.dptr(b), _lib.dptr(c),
                                         len(k1t), i32(k1t), i64(k1s), i64(k1l),
                                         len(k2t), i32(k2t), i64(k2s), i64(k2l), device, 1 if row_sharded else 0, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fos_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- algorithm / state
    def set_alg(self, alg: FOSAlgorithm):
        code, a, a1, a2, beta = alg._alg_args()
        _lib.check(self._lib.fos_set_alg(self._h, code, a, a1, a2, beta))
        if isinstance(alg, LineSearchWrapper):
            self.set_linesearch(alg.lsinterval)
        if isinstance(alg, GAPP):
            _lib.check(self._lib.fos_set_gapp(self._h, alg.iproj))
        if isinstance(alg, LongstepWrapper):
            self.set_longstep(alg.longinterval, alg.nsave)

    def set_longstep(self, longinterval, nsave):
        """LongstepWrapper around the current algorithm (longinterval 0: off); fos_set_longstep."""
        _lib.check(self._lib.fos_set_longstep(self._h, int(longinterval), int(nsave)))

    def longstep_log(self):
        """last projection onto the saved planes: dict(iteration, active inequalities, KKT violation of the small dual, step length, rows, supports tried)"""
        out = np.zeros(8)
        _lib.check(self._lib.fos_longstep_log(self._h, _lib.dptr(out)))
        return dict(iteration=int(out[0]), active=int(out[1]), violation=float(out[2]), step=float(out[3]), rows=int(out[4]), tried=int(out[5]), failed=bool(out[6]),
                    given_up=int(out[7]))

    def gapp_log(self):
        """(iteration, [21 test norms], alpha_best) of GAPP's last search."""
        out = np.zeros(23)
        _lib.check(self._lib.fos_gapp_log(self._h, _lib.dptr(out)))
        return int(out[22]), out[0:21].copy(), float(out[21])

    def set_linesearch(self, lsinterval):
        """LineSearchWrapper around the current algorithm (0: off); fos_set_linesearch."""
        _lib.check(self._lib.fos_set_linesearch(self._h, int(lsinterval)))

    def linesearch_log(self):
        """(iteration, ||res||, [31 test residuals], alpha_best) of the last search."""
        out = np.zeros(34)
        _lib.check(self._lib.fos_linesearch_log(self._h, _lib.dptr(out)))
        return int(out[33]), float(out[0]), out[1:32].copy(), float(out[32])

    def reset_affine(self):
        _lib.check(self._lib.fos_reset_affine(self._h))

    def set_iterate(self, z=None):
        if z is None:
            _lib.check(self._lib.fos_set_iterate(self._h, None))
        else:
            z = _lib.as_f64(z, self.N)
            _lib.check(self._lib.fos_set_iterate(self._h, _lib.dptr(z)))

    def get_iterate(self):
        z = np.empty(self.N)
        _lib.check(self._lib.fos_get_iterate(self._h, _lib.dptr(z)))
        return z

    def get_checked(self):
        z = np.empty(self.N)
        _lib.check(self._lib.fos_get_checked(self._h, _lib.dptr(z)))
        return z

    def step(self, i_first, count, checki, eps):
        done = C.c_int64(0)
        checked = C.c_int32(0)
        res = _lib.CheckResult()
        _lib.check(self._lib.fos_step(self._h, i_first, count, checki, eps, C.byref(done), C.byref(checked), C.byref(res)))
        return done.value, bool(checked.value), res

    def getsol(self, force_check=False, eps=1e-5):
        z = np.empty(self.N)
        res = _lib.CheckResult()
        _lib.check(self._lib.fos_getsol(self._h, _lib.dptr(z), 1 if force_check else 0, eps, C.byref(res)))
        return z, (res if force_check else None)

    def get_affine_state(self):
        """(xinit, i, firstrun): CGdata.xinit, AffinePlusLinear.i, CGdata.firstrun."""
        z = np.empty(self.N)
        i = C.c_int64(0)
        fr = C.c_int32(0)
        _lib.check(self._lib.fos_get_affine_state(self._h, _lib.dptr(z), C.byref(i), C.byref(fr)))
        return z, i.value, bool(fr.value)

    def set_affine_state(self, xinit, i):
        xinit = _lib.as_f64(xinit, self.N)
        _lib.check(self._lib.fos_set_affine_state(self._h, _lib.dptr(xinit), int(i)))

    def get_alg_state(self):
        """(a, b, t, alpha12): FISTAData.y / .xold / .t (fista.jl:15-25), DykstraData.p / .q (dykstra.jl:12-23), GAPAData.alpha12."""
        a, b, sc = np.empty(self.N), np.empty(self.N), np.zeros(2)
        _lib.check(self._lib.fos_get_alg_state(self._h, _lib.dptr(a), _lib.dptr(b), _lib.dptr(sc)))
        return a, b, float(sc[0]), float(sc[1])

    def set_alg_state(self, a=None, b=None, t=None, alpha12=None):
        """installs what is given (fos_set_alg_state); t / alpha12 default to the handle's current values"""
        sc = None
        if t is not None or alpha12 is not None:
            cur = np.zeros(2)
            _lib.check(self._lib.fos_get_alg_state(self._h, None, None, _lib.dptr(cur)))
            sc = np.array([cur[0] if t is None else float(t), cur[1] if alpha12 is None else float(alpha12)])
        a = None if a is None else _lib.as_f64(a, self.N)
        b = None if b is None else _lib.as_f64(b, self.N)
        _lib.check(self._lib.fos_set_alg_state(self._h, None if a is None else _lib.dptr(a), None if b is None else _lib.dptr(b),
                                               None if sc is None else _lib.dptr(sc)))

    def cgiter(self):
        v = C.c_int64(0)
        _lib.check(self._lib.fos_get_cgiter(self._h, C.byref(v)))
        return v.value

    def alpha12(self):
        v = C.c_double(0)
        _lib.check(self._lib.fos_get_alpha12(self._h, C.byref(v)))
        return v.value

    def prox_count(self):
        v = C.c_int64(0)
        _lib.check(self._lib.fos_get_prox_count(self._h, C.byref(v)))
        return v.value

    # -- fine grained operators (each: host -> device -> host)
    def q_apply(self, x, transpose=False):
        x = _lib.as_f64(x, self.l)
        y = np.empty(self.l)
        _lib.check(self._lib.fos_q_apply(self._h, _lib.dptr(y), _lib.dptr(x), 1 if transpose else 0))
        return y

    def kkt_apply(self, x):
        x = _lib.as_f64(x, self.N)
        y = np.empty(self.N)
        _lib.check(self._lib.fos_kkt_apply(self._h, _lib.dptr(y), _lib.dptr(x)))
        return y

    def cg_kkt(self, x0, rhs, tol, max_iters=10000):
        x = _lib.as_f64(x0, self.N).copy()
        rhs = _lib.as_f64(rhs, self.N)
        it = C.c_int64(0)
        _lib.check(self._lib.fos_cg_kkt(self._h, _lib.dptr(x), _lib.dptr(rhs), tol, max_iters, C.byref(it)))
        return x, it.value

    def prox_affine(self, x):
        x = _lib.as_f64(x, self.N)
        y = np.empty(self.N)
        _lib.check(self._lib.fos_prox_affine(self._h, _lib.dptr(y), _lib.dptr(x)))
        return y

    def hsdematrix_prox(self, x):
        x = _lib.as_f64(x, self.N)
        y = np.empty(self.N)
        _lib.check(self._lib.fos_hsdematrix_prox(self._h, _lib.dptr(y), _lib.dptr(x)))
        return y

    def prox_cones(self, x):
        x = _lib.as_f64(x, self.N)
        y = np.empty(self.N)
        _lib.check(self._lib.fos_prox_cones(self._h, _lib.dptr(y), _lib.dptr(x)))
        return y

    def check(self, z, eps=1e-5):
        z = _lib.as_f64(z, self.N)
        res = _lib.CheckResult()
        _lib.check(self._lib.fos_check(self._h, _lib.dptr(z), eps, C.byref(res)))
        return res

    # -- measurement
    def profile(self, enable):
        """False/0: off; True/1: events around every KKT launch; N > 1: around every N-th (sampling)."""
        _lib.check(self._lib.fos_profile(self._h, int(enable)))

    def profile_read(self):
        n = C.c_int64(0)
        ms = C.c_double(0)
        by = C.c_double(0)
        _lib.check(self._lib.fos_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(by)))
        return n.value, ms.value, by.value

    def bench_kkt(self, reps):
        ms = C.c_double(0)
        _lib.check(self._lib.fos_bench_kkt(self._h, reps, C.byref(ms)))
        return ms.value

    def cg_total(self):
        v = C.c_int64(0)
        _lib.check(self._lib.fos_get_cg_total(self._h, C.byref(v)))
        return v.value

    def operator_stats(self):
        """Format statistics of the device operator (fos_operator_stats)."""
        st = (C.c_int64 * 12)()
        _lib.check(self._lib.fos_operator_stats(self._h, st))
        keys = ("blocks", "ell", "lds", "long", "run", "vals", "cols", "waves", "tiles", "slots", "deferred", "tile_vals")
        out = dict(zip(keys, list(st)))
        ws = (C.c_int64 * 4)()
        _lib.check(self._lib.fos_window_stats(self._h, ws))
        out.update(zip(("win_panels", "win_segments", "win_slices", "win_vals"), list(ws)))
        return out

    def sync(self):
        _lib.check(self._lib.fos_sync(self._h))

    def enable_direct(self, A):
        """direct = true (HSDE.jl:12-15): exact affine projection through a one-time dense factorisation (fos_enable_direct)."""
        A = sp.csc_matrix(A)
        A.sort_indices()
        colptr = (A.indptr.astype(np.int64) + 1)
        rowval = (A.indices.astype(np.int64) + 1)
        nz = np.ascontiguousarray(A.data, dtype=np.float64)
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        _lib.check(self._lib.fos_enable_direct(self._h, i64(colptr), i64(rowval), _lib.dptr(nz)))

    def disable_direct(self):
        _lib.check(self._lib.fos_disable_direct(self._h))

    def direct_mode(self):
        """'off' | 'dense' | 'block' | 'cg': the form S1 = IndAffine([Q -I], 0) runs in (fos_get_direct_mode)"""
        v = C.c_int32(0)
        _lib.check(self._lib.fos_get_direct_mode(self._h, C.byref(v)))
        return ("off", "dense", "block", "cg")[v.value]

    def set_tuning(self, spmv_workgroups=0, cg_chunk=0, fuse_p=-1):
        """fuse_p: -1 keeps the library's choice, 0 / 1 force the three- / two-launch CG iteration."""
        _lib.check(self._lib.fos_set_tuning(self._h, spmv_workgroups, cg_chunk, fuse_p))

    def set_cg_variant(self, variant):
        """which CG recurrence the affine projection runs: 'reference' | 'fused_p' | 'merged_sweep' | 'merged_update' | 'resident' | None (default).
        'resident' (one launch per solve, operator and vectors in registers) raises FosError(FOS_EUNSUPPORTED) when the operator does not qualify."""
        codes = {None: -1, "default": -1, "reference": _lib.CG_REFERENCE, "fused_p": _lib.CG_FUSED_P,
                 "merged_sweep": _lib.CG_MERGED_SWEEP, "merged_update": _lib.CG_MERGED_UPDATE, "resident": _lib.CG_RESIDENT}
        _lib.check(self._lib.fos_set_cg_variant(self._h, codes[variant] if not isinstance(variant, int) else variant))

    def cg_variant_name(self):
        v = C.c_int32(0)
        _lib.check(self._lib.fos_get_cg_variant(self._h, C.byref(v)))
        return ("reference", "fused_p", "merged_sweep", "merged_update", "resident")[v.value]

    def resident_stats(self):
        """The plan of the resident CG solve (foship.h fos_resident_stats)."""
        st = (C.c_int64 * 8)()
        _lib.check(self._lib.fos_resident_stats(self._h, st))
        keys = ("qualifies", "workgroups", "waves_per_workgroup", "tiles_per_wave", "units", "max_tiles_per_workgroup", "steps_per_tile", "all_ranks_qualify")
        out = {k: int(st[i]) for i, k in enumerate(keys)}
        out["form"] = "streamed" if out["tiles_per_wave"] < 0 else "registers"        # streamed: tiles re-read every iteration (resident.hip, cg_stream_kernel)
        out["tiles_per_wave"] = abs(out["tiles_per_wave"])
        return out

    def debug_set(self, what, value):
        _lib.check(self._lib.fos_debug_set(self._h, int(what), int(value)))

    def profile_read_classes(self):
        """{'kkt' | 'psd' | 'cgvec': (launch groups, summed ms), 'other': (sampled outer iterations, summed ms of every other launch
        group in them)} of the bracketed launches since the last read."""
        n = (C.c_int64 * 5)()
        ms = (C.c_double * 5)()
        _lib.check(self._lib.fos_profile_read_classes(self._h, n, ms))
        return {k: (n[i], ms[i]) for i, k in enumerate(("kkt", "psd", "cgvec", "other", "resident"))}

    def psd_debug(self, collect_stats=True, phase_limit=0):
        _lib.check(self._lib.fos_psd_debug(self._h, 1 if collect_stats else 0, int(phase_limit)))

    def psd_sweeps(self):
        """Jacobi sweeps of the last PSD projection per (cone, copy) matrix (after psd_debug(True))."""
        n = C.c_int64(0)
        _lib.check(self._lib.fos_psd_stats(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), dtype=np.int32)
        _lib.check(self._lib.fos_psd_stats(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return out[:n.value]

    def bench_cg_chain(self, iters, reps=5, use_graph=False):
        """ms per CG iteration of a chain that never converges (fos_bench_cg_chain)."""
        ms = C.c_double(0)
        _lib.check(self._lib.fos_bench_cg_chain(self._h, iters, reps, 1 if use_graph else 0, C.byref(ms)))
        return ms.value

    # -- sharding; the host side hands over an ncclUniqueId obtained on rank 0
    @staticmethod
    def comm_unique_id():
        buf = (C.c_ubyte * 128)()
        _lib.check(_lib.load().fos_comm_get_unique_id(buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id: bytes):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _lib.check(self._lib.fos_comm_init(self._h, nranks, rank, buf))

    def comm_init_host(self, nranks, rank, allreduce_sum):
        """Sharding over the caller's own collective (fos_comm_init_host): `allreduce_sum(a)` must replace the float64 numpy
        array `a` by its sum over the ranks, in place, blocking -- e.g. torch.distributed.all_reduce(torch.from_numpy(a)) on
        any backend, or MPI's Allreduce."""
        def _cb(user, buf, count):
            try:
                allreduce_sum(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:           # noqa: BLE001 -- reported through the ABI's error code
                import traceback
                traceback.print_exc()
                return 1
        self._host_cb = _lib.ALLREDUCE_FN(_cb)          # keep the trampoline alive as long as the handle
        _lib.check(self._lib.fos_comm_init_host(self._h, nranks, rank, C.cast(self._host_cb, C.c_void_p), None))

    # -- peer mailboxes: the sharded sums without a collective call (include/foship.h, fos_peer_*)
    def peer_export(self) -> bytes:
        buf = (C.c_ubyte * 64)()
        _lib.check(self._lib.fos_peer_export(self._h, buf))
        return bytes(buf)

    def peer_open(self, nranks, rank, handles, timeout_s=0.0):
        """handles: the peer_export() bytes of every rank, in rank order."""
        blob = b"".join(handles)
        assert len(blob) == 64 * nranks
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        _lib.check(self._lib.fos_peer_open(self._h, nranks, rank, buf, float(timeout_s)))

    def peer_open_host(self, nranks, rank, shm_name, timeout_s=0.0):
        """host-pinned mailboxes (fos_peer_open_host): `shm_name` = "/something-unique-to-the-job", the same on every rank of the node;
        replaces peer_export + peer_open, then peer_selftest / peer_enable as usual."""
        _lib.check(self._lib.fos_peer_open_host(self._h, nranks, rank, shm_name.encode(), float(timeout_s)))

    def peer_close(self):
        """drop the open mailboxes (device or host) so that another transport can be opened on this handle"""
        _lib.check(self._lib.fos_peer_close(self._h))

    def peer_vec_export(self) -> bytes:
        """row-sharded handles: the exchange buffer of the n-vector A'y (after peer_open)"""
        buf = (C.c_ubyte * 64)()
        _lib.check(self._lib.fos_peer_vec_export(self._h, buf))
        return bytes(buf)

    def peer_vec_open(self, handles):
        blob = b"".join(handles)
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        _lib.check(self._lib.fos_peer_vec_open(self._h, buf))

    def peer_selftest(self, rounds=32) -> bool:
        ok = C.c_int32(0)
        _lib.check(self._lib.fos_peer_selftest(self._h, rounds, C.byref(ok)))
        return bool(ok.value)

    def exchange_bench(self, rounds=200) -> float:
        """microseconds per exchange of four doubles on the handle's transport (collective; foship.h fos_exchange_bench)."""
        us = C.c_double(0.0)
        _lib.check(self._lib.fos_exchange_bench(self._h, int(rounds), C.byref(us)))
        return float(us.value)

    def peer_enable(self, on=True):
        _lib.check(self._lib.fos_peer_enable(self._h, 1 if on else 0))


# ---------------------------------------------------------------------------------------------- status

class HSDEStatus:
    """HSDEStatus (HSDEStatus.jl:2-16): fields the outer loop reads/writes + printing + history."""

    def __init__(self, model, checki, eps, verbose, debug, out=None):
        self.model = model
        self.i = 0
        self.status = "Continue"
        self.checki, self.eps, self.verbose, self.debug = checki, eps, verbose, debug
        self.checked = False
        self.direct = False
        self.init_time = time.perf_counter_ns()
        self._out = out

    def _println(self, s):
        if self._out is None:
            print(s)
        else:
            self._out.append(s)

    def printstatusheader(self):                                   # HSDEStatus.jl:73-83
        if self.verbose > 0:
            self._println("Time to initialize: %ss" % (self.model.init_duration / 1e9))
            width = 76 + (0 if self.direct else 5)
            self._println("-" * width)
            self._println(HEADER_DIRECT if self.direct else HEADER_CG)
            self._println("-" * width)

    def record(self, res, z=None):
        """What checkstatus does once the residual scalars are known (HSDEStatus.jl:39-65)."""
        t = time.perf_counter_ns() - self.init_time
        i, h = self.i, self.model.history
        if self.debug > 0:                                         # savedata :125-139
            for key, val in (("p", res.p), ("d", res.d), ("g", res.g), ("ctx", res.ctx), ("bty", res.bty),
                             ("κ", res.kappa), ("τ", res.tau), ("t", t)):
                h.setdefault(key, []).append((i, val))
            if self.debug > 1 and z is not None:
                m, n = self.model.m, self.model.n
                nu = n + m + 1
                h.setdefault("x", []).append((i, z[0:n].copy()))
                h.setdefault("y", []).append((i, z[n:n + m].copy()))
                h.setdefault("s", []).append((i, z[nu + n:nu + n + m].copy()))
        if self.verbose > 0 and not self.direct:                   # :43-47
            h.setdefault("cgiter", []).append((i, int(res.cgiter)))
            self._println(_julia_specials("%6d|% 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % 4d % .1es" %
                          (i, res.p, res.d, res.g, res.ctx, -res.bty, _ieee_div(res.kappa, res.tau), res.cgiter, t / 1e9)))
        elif self.verbose > 0:                                     # :48-50 (direct: no cg column, no :cgiter history)
            self._println(_julia_specials("%6d|% 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % 9.2e % .1es" %
                          (i, res.p, res.d, res.g, res.ctx, -res.bty, _ieee_div(res.kappa, res.tau), t / 1e9)))
        if res.cg_maxiter_hit:
            import warnings
            warnings.warn("CG reached max iterations, result may be inaccurate")     # conjugategradients.jl:53
        if isinstance(self.model.alg, LongstepWrapper):            # (the reference's QP solver throws where no projection exists; here the step stands and the host is told)
            gu = self.model.data.longstep_log()["given_up"]
            if gu > getattr(self, "_long_given_up", 0):
                import warnings
                warnings.warn("LongstepWrapper: %d projection(s) onto the saved planes found no KKT point of the small dual within its budget "
                              "(inconsistent or dependent planes); those iterations kept the wrapped algorithm's iterate" % (gu - getattr(self, "_long_given_up", 0)))
            self._long_given_up = gu
        self.status = _lib.STATUS_NAMES[res.status]
        if self.status == "Optimal" and self.verbose > 0:
            self._println("Found solution i=%d" % i)               # :55-57
        self.checked = True
        self.last = res


class Solution:                                                    # src/types.jl:6-11
    def __init__(self, x, y, s, status):
        self.x, self.y, self.s, self.status = x, y, s, status


# ---------------------------------------------------------------------------------------------- model

class FOSMathProgModel:
    """FOSMathProgModel (src/types.jl:30-60) with the MathProgBase methods of src/FOSSolverInterface.jl."""

    def __init__(self, alg: FOSAlgorithm, device=0, **kwargs):
        self.alg = alg
        self.options = dict(alg.options)
        self.options.update(kwargs)
        self.device = device
        self.input_numconstr = 0
        self.input_numvar = 0
        self.solve_stat = "NotSolved"
        self.obj_val = 0.0
        self.primal_sol = np.zeros(0)
        self.dual_sol = np.zeros(0)
        self.slack = np.zeros(0)
        self.history = {}
        self.init_duration = 1
        self.data = None
        self.out = None            # list collecting printed lines (None -> stdout)

    # loadproblem!(model, c, A, b, constr_cones, var_cones)        FOSSolverInterface.jl:27-64
    def loadproblem(self, c, A, b, constr_cones, var_cones):
        t1 = time.perf_counter_ns()
        A = sp.csc_matrix(A)
        self.input_numconstr, self.input_numvar = A.shape
        self.m, self.n = A.shape
        self.A, self.b, self.c = A, _lib.as_f64(b, A.shape[0]), _lib.as_f64(c, A.shape[1])
        self.K1, self.K2 = list(constr_cones), list(var_cones)
        if self.data is not None:
            self.data.close()
        self.data = HipHSDE(A, self.b, self.c, self.K1, self.K2, device=self.device)     # init_algorithm!  :58
        self.data.set_alg(self.alg)
        if "cg_variant" in self.options:                           # device-side key (the reference ignores unknown option keys):
            self.data.set_cg_variant(self.options["cg_variant"])   # which CG recurrence the affine projection runs (foship.h FOS_CG_*)
        if self.alg.direct:                                        # HSDE(model, direct=alg.direct)   HSDE.jl:12-15
            self.data.enable_direct(A)
        self.init_duration = time.perf_counter_ns() - t1
        return self

    def numvar(self):
        return self.input_numvar

    def numconstr(self):
        return self.input_numconstr

    @staticmethod
    def supportedcones():                                          # FOSSolverInterface.jl:69
        return ["Free", "Zero", "NonNeg", "NonPos", "SOC", "SDP", "ExpPrimal", "ExpDual"]

    # optimize!(m)                                                 FOSSolverInterface.jl:8-21
    def optimize(self):
        self.history = {}
        sol = self._solve()
        self.solve_stat = sol.status
        self.primal_sol, self.dual_sol, self.slack = sol.x, sol.y, sol.s
        self.obj_val = float(np.dot(self.c, self.primal_sol))
        return self

    def status(self):
        return self.solve_stat

    def getobjval(self):
        return self.obj_val

    def getsolution(self):
        return self.primal_sol.copy()

    # solve!(model) + iterate                                      solverwrapper.jl:2-41
    def _solve(self):
        opts = self.options
        max_iters = opts.get("max_iters", 10000)
        verbose = opts.get("verbose", 1)
        debug = opts.get("debug", 1)
        eps = opts.get("eps", 1e-5)
        checki = opts.get("checki", 100)
        dev = self.data                                            # model.data persists across optimize! calls, as in the
        dev.set_iterate(opts.get("initx", None))                   # reference (alpha12 / t / y / S1 counters carry over); :10
        status = HSDEStatus(self, checki, eps, verbose, debug, out=self.out)
        # HSDE.jl:27 -- the table drops its cg column when S1 runs no CG; beyond the sizes the exact forms cover, direct = true is the same
        # projection by CG at its tolerance floor (fos_enable_direct): the column stays, the iterations are real
        status.direct = bool(self.alg.direct) and self.data.direct_mode() in ("dense", "block")
        self.status_obj = status
        t1 = time.time()
        status.printstatusheader()
        i = 0
        ls = self.alg.lsinterval if isinstance(self.alg, LineSearchWrapper) else 0
        gp = self.alg.iproj if isinstance(self.alg, GAPP) else 0
        while i < max_iters:                                       # for i = 1:max_iters   :23
            count = min(max_iters - i, checki - (i % checki))
            if ls > 0:
                count = min(count, ls - (i % ls))                  # stop at every line-search iteration: its output is printed
            if gp > 0:
                count = min(count, gp - (i % gp))
            done, checked, res = dev.step(i + 1, count, checki, eps)
            i += done
            status.i = i
            if gp > 0 and i % gp == 0:                             # what gapproj.jl:51,57 print (unconditionally)
                _, tests, abest = dev.gapp_log()
                for nt in tests:
                    status._println("normtest: %s" % julia_float(nt))
                status._println("\u03b1best: %s" % julia_float(abest))
            if ls > 0 and i % ls == 0:                             # what linesearch.jl:51,63,69 print
                _, normres, tests, abest = dev.linesearch_log()
                status._println("test, %s" % julia_float(normres))
                a = 0.1
                for tr in tests:
                    a = a * 1.8
                    status._println("\u03b1: %s, %s" % (julia_float(a), julia_float(tr)))
                status._println("\u03b1: %s" % julia_float(abest))
            if checked:
                z = dev.get_checked() if debug > 1 else None      # debug=2 stores x,y,s of the checked point
                status.record(res, z)
                if status.status != "Continue":                    # :26
                    break
            else:
                status.checked = False
        guess, res = dev.getsol(force_check=not status.checked, eps=eps)       # :31-34
        if not status.checked:
            status.record(res, guess)
        if verbose > 0:                                            # :35-39
            status._println("Time for iterations: ")
            status._println("%s s" % (time.time() - t1))
        self.iterations = i
        self.guess = guess
        # HSDE_populatesolution                                    HSDE.jl:49-61
        m, n = self.m, self.n
        l = m + n + 1
        tau = guess[l - 1]
        endstatus = status.status if status.status != "Continue" else "Indeterminate"
        with np.errstate(divide="ignore", invalid="ignore"):
            return Solution(guess[0:n] / tau, guess[n:n + m] / tau, guess[l + n:l + n + m] / tau, endstatus)


def solve(problem, alg, device=0, out=None):
    """Convenience: ConicModel(alg) -> loadproblem! -> optimize!  for a workloads.ConicProblem."""
    model = FOSMathProgModel(alg, device=device)
    model.out = out
    model.loadproblem(problem.c, problem.A, problem.b, problem.K1, problem.K2)
    model.optimize()
    return model


# ---------------------------------------------------------------------------------------------- Feasibility form
# src/problemforms/Feasibility/Feasibility.jl, FeasibilityStatus.jl.  The reference takes any two
# ProximalOperators objects; the device path takes the two set types below (the ones test/testfeasibility.jl uses).

class IndAffine:
    """ProximalOperators.IndAffine(A, b): {x : A x = b}, A m x n of full row rank.  A dense A (n <= 46 000) becomes a dense projector on the
    device (fos_feas_set_affine); a scipy.sparse A stays sparse at any n (fos_feas_set_affine_sparse: warm-started CG on the row-scaled normal
    equations, ended by the recomputed residual) -- `sparse=True / False` forces either form."""

    DENSE_MAX = 46000

    def __init__(self, A, b, sparse=None):
        self.sparse = bool(sp.issparse(A) or np.shape(A)[1] > self.DENSE_MAX) if sparse is None else bool(sparse)
        if self.sparse:
            self.A = sp.csc_matrix(A, dtype=np.float64)
            self.A.sum_duplicates(); self.A.sort_indices()
        else:
            self.A = np.ascontiguousarray(A.toarray() if sp.issparse(A) else np.asarray(A, dtype=np.float64))
        self.b = np.ascontiguousarray(np.asarray(b, dtype=np.float64))
        if self.A.ndim != 2 or self.b.shape != (self.A.shape[0],):
            raise ValueError("IndAffine(A, b): A must be m x n and b of length m")


class IndBox:
    """ProximalOperators.IndBox(lo, hi): {x : lo <= x <= hi}, scalar or array bounds (+-inf allowed)."""

    def __init__(self, lo, hi):
        self.arrays = np.ndim(lo) > 0 or np.ndim(hi) > 0
        if self.arrays:
            self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
        else:
            self.lo, self.hi = float(lo), float(hi)


class ConeProduct:
    """FirstOrderSolvers.ConeProduct (src/cones.jl:31-94) as a set of the Feasibility form: cones = [(name, length), ...] with the
    names of `conemap` (cones.jl:4-14: Free, Zero, NonNeg, NonPos, SOC, SOCRotated, SDP, ExpPrimal, ExpDual), in order, contiguous."""

    def __init__(self, cones):
        self.cones = [(str(k), int(l)) for k, l in cones]
        for k, _ in self.cones:
            if k not in _lib.CONE_CODES:
                raise ValueError("unknown cone %r (supportedcones: %s)" % (k, ", ".join(_lib.CONE_CODES)))


PROX_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double))     # fos_prox_fn


class Feasibility:
    """struct Feasibility{T1,T2}(S1, S2, n)   Feasibility.jl:2-6.  S1, S2: IndAffine, IndBox, ConeProduct (device resident) or any
    object with a `prox(y, x)` method filling y = prox_S(x) in place (the ProximableFunction protocol; evaluated on the host)."""

    def __init__(self, S1, S2, n):
        self.S1, self.S2, self.n = S1, S2, int(n)


class FeasibilitySolution:
    """mutable struct FeasibilitySolution   Feasibility.jl:8-11"""

    def __init__(self, x, status):
        self.x, self.status = x, status


class HipFeasibility:
    """The device handle of a Feasibility problem (fos_feas_*): both sets, the algorithm's vectors and the status state."""

    def __init__(self, problem: Feasibility, device=0):
        self._lib = _lib.load()
        self.n = problem.n
        h = C.c_void_p()
        _lib.check(self._lib.fos_feas_create(self.n, device, C.byref(h)))
        self._h = h
        self._callbacks = []
        self.callback_error = None
        for which, S in ((1, problem.S1), (2, problem.S2)):
            if isinstance(S, IndAffine):
                if S.A.shape[1] != self.n:
                    raise ValueError("IndAffine: A has %d columns, the problem has n = %d" % (S.A.shape[1], self.n))
                if S.sparse:
                    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
                    colptr = np.ascontiguousarray(S.A.indptr, dtype=np.int64) + 1          # Julia 1-based
                    rowval = np.ascontiguousarray(S.A.indices, dtype=np.int64) + 1
                    nzval = np.ascontiguousarray(S.A.data, dtype=np.float64)
                    _lib.check(self._lib.fos_feas_set_affine_sparse(self._h, which, S.A.shape[0], i64(colptr), i64(rowval), _lib.dptr(nzval), _lib.dptr(S.b)))
                else:
                    _lib.check(self._lib.fos_feas_set_affine(self._h, which, S.A.shape[0], _lib.dptr(S.A), _lib.dptr(S.b)))
            elif isinstance(S, IndBox) and S.arrays:
                lo = np.ascontiguousarray(np.broadcast_to(S.lo, (self.n,)), dtype=np.float64)
                hi = np.ascontiguousarray(np.broadcast_to(S.hi, (self.n,)), dtype=np.float64)
                _lib.check(self._lib.fos_feas_set_box_arrays(self._h, which, _lib.dptr(lo), _lib.dptr(hi)))
            elif isinstance(S, IndBox):
                _lib.check(self._lib.fos_feas_set_box(self._h, which, S.lo, S.hi))
            elif isinstance(S, ConeProduct):
                types = np.ascontiguousarray([_lib.CONE_CODES[k] for k, _ in S.cones], dtype=np.int32)
                lens = np.ascontiguousarray([l for _, l in S.cones], dtype=np.int64)
                _lib.check(self._lib.fos_feas_set_cones(self._h, which, len(S.cones), types.ctypes.data_as(C.POINTER(C.c_int32)),
                                                        lens.ctypes.data_as(C.POINTER(C.c_int64))))
            elif callable(getattr(S, "prox", None)):                    # any other ProximableFunction: prox!(y, S, x) as a host callback
                cb = PROX_FN(self._make_prox_callback(S))
                self._callbacks.append(cb)                              # (the library keeps the pointer: it must outlive the handle)
                _lib.check(self._lib.fos_feas_set_callback(self._h, which, C.cast(cb, C.c_void_p), None))
            else:
                raise _lib.FosError(-4, "Feasibility: set %d must be IndAffine, IndBox, ConeProduct or an object with a prox(y, x) method "
                                        "(evaluated on the host through fos_feas_set_callback), got %s" % (which, type(S).__name__))

    def _make_prox_callback(self, S):
        n = self.n

        def call(ctx, nn, xp, yp):
            try:
                x = np.ctypeslib.as_array(xp, shape=(n,))
                y = np.ctypeslib.as_array(yp, shape=(n,))
                S.prox(y, x)                                            # prox!(y, S, x): fills y in place
                return 0
            except Exception as exc:  # noqa: BLE001  (an exception must not unwind through the C frames)
                self.callback_error = exc
                return 1
        return call

    def _check(self, rc):
        """_lib.check, with the exception a prox callback raised (if that is what stopped the call) as the cause"""
        try:
            _lib.check(rc)
        except _lib.FosError as err:
            exc, self.callback_error = self.callback_error, None
            if exc is not None:
                raise err from exc
            raise

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fos_feas_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_alg(self, alg: FOSAlgorithm):
        if isinstance(alg, GAPP):
            _lib.check(self._lib.fos_feas_set_gapp(self._h, alg.alpha, alg.alpha1, alg.alpha2, alg.iproj))
            return
        _lib.check(self._lib.fos_feas_set_alg(self._h, *alg._alg_args()))
        if isinstance(alg, LineSearchWrapper):                          # the wrapped algorithm's arguments, then the search
            _lib.check(self._lib.fos_feas_set_linesearch(self._h, alg.lsinterval))
        if isinstance(alg, LongstepWrapper):
            _lib.check(self._lib.fos_feas_set_longstep(self._h, alg.longinterval, alg.nsave))

    def longstep_log(self):
        out = np.zeros(8)
        _lib.check(self._lib.fos_feas_longstep_log(self._h, _lib.dptr(out)))
        return dict(iteration=int(out[0]), active=int(out[1]), violation=float(out[2]), step=float(out[3]), rows=int(out[4]), tried=int(out[5]), failed=bool(out[6]),
                    given_up=int(out[7]))

    def gapp_log(self):
        """(iteration, [21 test norms], alpha_best) of GAPP's last search."""
        out = np.zeros(23)
        _lib.check(self._lib.fos_feas_gapp_log(self._h, _lib.dptr(out)))
        return int(out[22]), out[0:21].copy(), float(out[21])

    def linesearch_log(self):
        """(iteration, ||res||, [31 test residuals], alpha_best) of the last search."""
        out = np.zeros(34)
        _lib.check(self._lib.fos_feas_linesearch_log(self._h, _lib.dptr(out)))
        return int(out[33]), float(out[0]), out[1:32].copy(), float(out[32])

    def set_iterate(self, x0=None):
        if x0 is None:
            _lib.check(self._lib.fos_feas_set_iterate(self._h, None))
        else:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            assert x0.shape == (self.n,)
            _lib.check(self._lib.fos_feas_set_iterate(self._h, _lib.dptr(x0)))

    def step(self, i_first, count, checki, eps):
        """-> (iterations run, status name, err of the last check or nan, checked flag of the last iteration)"""
        done, st, err, chk = C.c_int64(0), C.c_int32(0), C.c_double(float("nan")), C.c_int32(0)
        self._check(self._lib.fos_feas_step(self._h, i_first, count, checki, eps, C.byref(done), C.byref(st), C.byref(err), C.byref(chk)))
        return done.value, _lib.STATUS_NAMES[st.value], err.value, bool(chk.value)

    def getsol(self, force_check=False, eps=1e-5):
        sol = np.empty(self.n)
        st, err = C.c_int32(0), C.c_double(float("nan"))
        self._check(self._lib.fos_feas_getsol(self._h, _lib.dptr(sol), 1 if force_check else 0, eps, C.byref(st), C.byref(err)))
        return sol, _lib.STATUS_NAMES[st.value], err.value

    def get_iterate(self):
        x = np.empty(self.n)
        _lib.check(self._lib.fos_feas_get_iterate(self._h, _lib.dptr(x)))
        return x

    def prox(self, which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n)
        self._check(self._lib.fos_feas_prox(self._h, which, _lib.dptr(x), _lib.dptr(y)))
        return y

    def info(self):
        a12 = C.c_double(0.0)
        its = (C.c_int32 * 2)()
        res = (C.c_double * 2)()
        _lib.check(self._lib.fos_feas_info(self._h, C.byref(a12), its, res))
        return {"alpha12": a12.value, "ns_iters": list(its), "ns_resid": list(res)}

    def affine_stats(self, which):
        """the sparse IndAffine's counters (fos_feas_affine_stats)"""
        o = np.zeros(8)
        _lib.check(self._lib.fos_feas_affine_stats(self._h, which, _lib.dptr(o)))
        return {"projections": int(o[0]), "cg_iterations": int(o[1]), "last_cg_iterations": int(o[2]), "last_restarts": int(o[3]), "last_residual": o[4],
                "last_rounding_level": o[5], "nnz": int(o[6]), "lanes_per_row": (int(o[7]) // 1000, int(o[7]) % 1000)}


class FeasibilityModel:
    """mutable struct FeasibilityModel   Feasibility.jl:15-50: problem + algorithm + merged options, solve_stat, history."""

    def __init__(self, problem: Feasibility, alg: FOSAlgorithm, device=0, **kwargs):
        self.S1, self.S2, self.n, self.alg = problem.S1, problem.S2, problem.n, alg
        self.options = dict(alg.options)
        self.options.update(kwargs)                                    # kwargs of solve! override the algorithm's   :37-41
        self.solve_stat = "NotSolved"
        self.obj_val = 0.0
        self.enditr = -1
        self.history = {}
        self.out = None
        t1 = time.perf_counter_ns()
        self.dev = HipFeasibility(problem, device=device)              # init_algorithm!   :46
        self.dev.set_alg(alg)
        self.init_duration = time.perf_counter_ns() - t1

    def _println(self, s):
        if self.out is not None:
            self.out.append(s)
        else:
            print(s)

    def solve(self):
        """solve!(model)   solverwrapper.jl:2-41 with FeasibilityStatus (FeasibilityStatus.jl:32-91)."""
        o = self.options
        max_iters, verbose, debug = o.get("max_iters", 10000), o.get("verbose", 1), o.get("debug", 1)
        eps, checki = o.get("eps", 1e-5), o.get("checki", 100)
        dev = self.dev
        dev.set_iterate(o.get("initx"))
        t0 = time.perf_counter_ns()
        if verbose > 0:                                                # printstatusheader :74-84 (direct = true: no cg column)
            self._println("Time to initialize: %ss" % julia_float(self.init_duration / 1e9))
            self._println("-" * 22)
            self._println(" Iter | res | time")
            self._println("-" * 22)
        i, status, checked, err = 0, "Continue", False, float("nan")
        ls = self.alg.lsinterval if isinstance(self.alg, LineSearchWrapper) else 0
        gp = self.alg.iproj if isinstance(self.alg, GAPP) else 0
        while i < max_iters and status == "Continue":
            nxt = min(max_iters, (i // checki + 1) * checki) if checki > 0 else max_iters
            if ls > 0:
                nxt = min(nxt, (i // ls + 1) * ls)                     # stop at every line-search iteration: its output is printed
            if gp > 0:
                nxt = min(nxt, (i // gp + 1) * gp)
            done, status, err, checked = dev.step(i + 1, nxt - i, checki, eps)
            i += done
            if gp > 0 and i % gp == 0:                                 # what gapproj.jl:51,57 print (unconditionally)
                _, tests, abest = dev.gapp_log()
                for nt in tests:
                    self._println("normtest: %s" % julia_float(nt))
                self._println("\u03b1best: %s" % julia_float(abest))
            if ls > 0 and i % ls == 0:                                 # what linesearch.jl:51,63,69 print
                _, normres, tests, abest = dev.linesearch_log()
                self._println("test, %s" % julia_float(normres))
                a = 0.1
                for tr in tests:
                    a = a * 1.8
                    self._println("\u03b1: %s, %s" % (julia_float(a), julia_float(tr)))
                self._println("\u03b1: %s" % julia_float(abest))
            if checked:
                t = time.perf_counter_ns() - t0
                if debug > 0:                                          # savedata :95-103
                    self.history.setdefault("err", []).append((i, err))
                    self.history.setdefault("t", []).append((i, t))
                if verbose > 0:                                        # printstatusiter :86-88
                    self._println("%6d|%s % .1es" % (i, _jl_e9(err), t / 1e9))
                    if status == "Optimal":
                        self._println("Found solution i=%d" % i)
        guess, st2, err2 = dev.getsol(force_check=not checked, eps=eps)            # solverwrapper.jl:30-34
        if not checked:
            status, err = st2, err2
            if debug > 0:
                self.history.setdefault("err", []).append((i, err))
            if verbose > 0:
                self._println("%6d|%s % .1es" % (i, _jl_e9(err), (time.perf_counter_ns() - t0) / 1e9))
                if status == "Optimal":
                    self._println("Found solution i=%d" % i)
        if verbose > 0:                                                # solverwrapper.jl:35-39
            self._println("Time for iterations: ")
            self._println("%s s" % julia_float((time.perf_counter_ns() - t0) / 1e9))
        self.enditr = i
        endstatus = "Indeterminate" if status == "Continue" else status            # populate_solution :61-68
        self.solve_stat = endstatus
        sol = FeasibilitySolution(guess, endstatus)
        sol.iterations, sol.err = i, err
        return sol


def _jl_e9(v):
    """@printf("% 9.2e", v)"""
    return "% 9.2e" % v


def solve_feasibility(problem: Feasibility, alg: FOSAlgorithm, device=0, out=None, **kwargs):
    """solve!(problem::Feasibility, alg; kwargs...) -> (solution, model)   Feasibility.jl:52-56"""
    model = FeasibilityModel(problem, alg, device=device, **kwargs)
    model.out = out
    return model.solve(), model
