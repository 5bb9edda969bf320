"""ctypes binding of libfoship.so (include/foship.h).  No fallback: a missing library is an ImportError-class failure."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["FOSHIP_LIB"]) if os.environ.get("FOSHIP_LIB") else _HERE / "csrc" / "libfoship.so"      # (FOSHIP_LIB: as in julia/FOSHip.jl)

# error codes / enums (mirror of include/foship.h)
FOS_OK = 0
CONE_CODES = {"Free": 0, "Zero": 1, "NonNeg": 2, "NonPos": 3, "SOC": 4, "SOCRotated": 5, "SDP": 6,
              "ExpPrimal": 7, "ExpDual": 8}
ALG_GAP, ALG_GAPA, ALG_FISTA, ALG_DYKSTRA = 0, 1, 2, 3
CG_REFERENCE, CG_FUSED_P, CG_MERGED_SWEEP, CG_MERGED_UPDATE, CG_RESIDENT = 0, 1, 2, 3, 4
DEBUG_PUPDATE_DELAY = 1
STATUS_NAMES = {0: "Continue", 1: "Optimal", 2: "Unbounded", 3: "Infeasible"}


class FosError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfoship error %d: %s" % (code, msg))
        self.code = code


class CheckResult(C.Structure):
    _fields_ = [("p", C.c_double), ("d", C.c_double), ("g", C.c_double), ("ctx", C.c_double), ("bty", C.c_double),
                ("kappa", C.c_double), ("tau", C.c_double), ("norm_axs", C.c_double), ("norm_aty", C.c_double),
                ("norm_b", C.c_double), ("norm_c", C.c_double), ("cgiter", C.c_int64), ("status", C.c_int32),
                ("cg_maxiter_hit", C.c_int32)]


_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_h = C.c_void_p

# name -> (restype, argtypes): every symbol include/foship.h declares
PROTOTYPES = {
    "fos_abi_version": (C.c_int, []),
    "fos_last_error": (C.c_char_p, []),
    "fos_device_count": (C.c_int, [_i32p]),
    "fos_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "fos_create": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i64p, _dp, _dp, _dp,
                             C.c_int64, _i32p, _i64p, _i64p, C.c_int64, _i32p, _i64p, _i64p, C.c_int, C.POINTER(_h)]),
    "fos_create2": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i64p, _dp, _dp, _dp,
                              C.c_int64, _i32p, _i64p, _i64p, C.c_int64, _i32p, _i64p, _i64p, C.c_int, C.c_int32, C.POINTER(_h)]),
    "fos_destroy": (C.c_int, [_h]),
    "fos_sizes": (C.c_int, [_h, _i64p, _i64p, _i64p, _i64p]),
    "fos_get_cg_total": (C.c_int, [_h, _i64p]),
    "fos_operator_stats": (C.c_int, [_h, _i64p]),
    "fos_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "fos_set_linesearch": (C.c_int, [_h, C.c_int64]),
    "fos_linesearch_log": (C.c_int, [_h, _dp]),
    "fos_set_longstep": (C.c_int, [_h, C.c_int64, C.c_int64]),
    "fos_longstep_log": (C.c_int, [_h, _dp]),
    "fos_comm_init": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    "fos_comm_init_host": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fos_peer_export": (C.c_int, [_h, C.c_void_p]),
    "fos_peer_open": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p, C.c_double]),
    "fos_peer_open_host": (C.c_int, [_h, C.c_int, C.c_int, C.c_char_p, C.c_double]),
    "fos_peer_close": (C.c_int, [_h]),
    "fos_peer_selftest": (C.c_int, [_h, C.c_int, C.POINTER(C.c_int32)]),
    "fos_exchange_bench": (C.c_int, [_h, C.c_int, _dp]),
    "fos_peer_vec_export": (C.c_int, [_h, C.c_void_p]),
    "fos_peer_vec_open": (C.c_int, [_h, C.c_void_p]),
    "fos_peer_enable": (C.c_int, [_h, C.c_int32]),
    "fos_set_alg": (C.c_int, [_h, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "fos_reset_affine": (C.c_int, [_h]),
    "fos_enable_direct": (C.c_int, [_h, _i64p, _i64p, _dp]),
    "fos_disable_direct": (C.c_int, [_h]),
    "fos_get_direct_mode": (C.c_int, [_h, _i32p]),
    "fos_set_iterate": (C.c_int, [_h, _dp]),
    "fos_get_iterate": (C.c_int, [_h, _dp]),
    "fos_get_checked": (C.c_int, [_h, _dp]),
    "fos_step": (C.c_int, [_h, C.c_int64, C.c_int64, C.c_int64, C.c_double, _i64p, _i32p, C.POINTER(CheckResult)]),
    "fos_getsol": (C.c_int, [_h, _dp, C.c_int32, C.c_double, C.POINTER(CheckResult)]),
    "fos_get_affine_state": (C.c_int, [_h, _dp, _i64p, _i32p]),
    "fos_set_affine_state": (C.c_int, [_h, _dp, C.c_int64]),
    "fos_get_alg_state": (C.c_int, [_h, _dp, _dp, _dp]),
    "fos_set_alg_state": (C.c_int, [_h, _dp, _dp, _dp]),
    "fos_get_cgiter": (C.c_int, [_h, _i64p]),
    "fos_get_alpha12": (C.c_int, [_h, _dp]),
    "fos_get_prox_count": (C.c_int, [_h, _i64p]),
    "fos_q_apply": (C.c_int, [_h, _dp, _dp, C.c_int32]),
    "fos_kkt_apply": (C.c_int, [_h, _dp, _dp]),
    "fos_cg_kkt": (C.c_int, [_h, _dp, _dp, C.c_double, C.c_int64, _i64p]),
    "fos_prox_affine": (C.c_int, [_h, _dp, _dp]),
    "fos_hsdematrix_prox": (C.c_int, [_h, _dp, _dp]),
    "fos_prox_cones": (C.c_int, [_h, _dp, _dp]),
    "fos_check": (C.c_int, [_h, _dp, C.c_double, C.POINTER(CheckResult)]),
    "fos_profile": (C.c_int, [_h, C.c_int32]),
    "fos_profile_read": (C.c_int, [_h, _i64p, _dp, _dp]),
    "fos_profile_read_classes": (C.c_int, [_h, _i64p, _dp]),
    "fos_bench_cg_chain": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "fos_bench_kkt": (C.c_int, [_h, C.c_int32, _dp]),
    "fos_psd_debug": (C.c_int, [_h, C.c_int32, C.c_int32]),
    "fos_psd_stats": (C.c_int, [_h, _i32p, C.c_int64, _i64p]),
    "fos_sync": (C.c_int, [_h]),
    "fos_host_stacked_spmv": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, C.c_int32, C.c_int32, _i64p]),
    "fos_host_resident_cg": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, C.c_int32, _dp, _dp, C.c_double, C.c_int64, _i64p, _i64p]),
    "fos_host_stacked_spmv_mode": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, C.c_int32, _i64p]),
    "fos_window_stats": (C.c_int, [_h, _i64p]),
    "fos_set_tuning": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int32]),
    "fos_set_cg_variant": (C.c_int, [_h, C.c_int32]),
    "fos_get_cg_variant": (C.c_int, [_h, _i32p]),
    "fos_resident_stats": (C.c_int, [_h, _i64p]),
    "fos_debug_set": (C.c_int, [_h, C.c_int32, C.c_int64]),
    "fos_set_gapp": (C.c_int, [_h, C.c_int64]),
    "fos_gapp_log": (C.c_int, [_h, _dp]),
    # Feasibility form (src/problemforms/Feasibility/*.jl)
    "fos_feas_create": (C.c_int, [C.c_int64, C.c_int32, C.POINTER(_h)]),
    "fos_feas_destroy": (C.c_int, [_h]),
    "fos_feas_set_affine": (C.c_int, [_h, C.c_int32, C.c_int64, _dp, _dp]),
    "fos_feas_set_affine_sparse": (C.c_int, [_h, C.c_int32, C.c_int64, _i64p, _i64p, _dp, _dp]),
    "fos_feas_affine_stats": (C.c_int, [_h, C.c_int32, _dp]),
    "fos_feas_set_box": (C.c_int, [_h, C.c_int32, C.c_double, C.c_double]),
    "fos_feas_set_box_arrays": (C.c_int, [_h, C.c_int32, _dp, _dp]),
    "fos_feas_set_cones": (C.c_int, [_h, C.c_int32, C.c_int64, _i32p, _i64p]),
    "fos_feas_set_callback": (C.c_int, [_h, C.c_int32, C.c_void_p, C.c_void_p]),
    "fos_feas_set_alg": (C.c_int, [_h, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]),
    "fos_feas_set_gapp": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, C.c_int64]),
    "fos_feas_gapp_log": (C.c_int, [_h, _dp]),
    "fos_feas_set_linesearch": (C.c_int, [_h, C.c_int64]),
    "fos_feas_linesearch_log": (C.c_int, [_h, _dp]),
    "fos_feas_set_longstep": (C.c_int, [_h, C.c_int64, C.c_int64]),
    "fos_feas_longstep_log": (C.c_int, [_h, _dp]),
    "fos_feas_set_iterate": (C.c_int, [_h, _dp]),
    "fos_feas_step": (C.c_int, [_h, C.c_int64, C.c_int64, C.c_int64, C.c_double, _i64p, _i32p, _dp, _i32p]),
    "fos_feas_getsol": (C.c_int, [_h, _dp, C.c_int32, C.c_double, _i32p, _dp]),
    "fos_feas_get_iterate": (C.c_int, [_h, _dp]),
    "fos_feas_prox": (C.c_int, [_h, C.c_int32, _dp, _dp]),
    "fos_feas_info": (C.c_int, [_h, _dp, _i32p, _dp]),
}

_lib = None


def load(check_symbols=False):
    """dlopen csrc/libfoship.so and bind every prototype.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and not check_symbols:
        return _lib
    if not LIB_PATH.exists():
        raise FosError(-100, "%s not found: build it with `make -C %s` (hipcc, gfx950); there is no CPU fallback"
                       % (LIB_PATH, LIB_PATH.parent))
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_GLOBAL", 0x100))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.fos_abi_version() != 1:
        raise FosError(-101, "ABI version mismatch")
    _lib = lib
    return lib


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)     # fos_allreduce_fn


def header_symbols():
    """The function names declared in include/foship.h (parsed, so the test notices a header/binding drift)."""
    import re
    text = (_HERE.parent / "include" / "foship.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fos_[a-z0-9_]+)\s*\(", text)))


def check(code):
    if code != FOS_OK:
        msg = load().fos_last_error()
        raise FosError(code, msg.decode("utf-8", "replace") if msg else "")


def dptr(a):
    return a.ctypes.data_as(_dp)


def as_f64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    if n is not None and a.shape[0] != n:
        raise ValueError("expected %d entries, got %d" % (n, a.shape[0]))
    return a


def device_count():
    n = C.c_int32(0)
    check(load().fos_device_count(C.byref(n)))
    return n.value
