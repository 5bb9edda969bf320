"""
ctypes wrapper of oracle/fos_cport.c -- the plain-C restatement used as the timed CPU baseline.

*** TEST / MEASUREMENT INFRASTRUCTURE ONLY *** (same rule as fos_oracle.py: tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg).  It is checked against fos_oracle.py in tests/test_cport.py, which in turn is pinned
against the reference's own tests.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = HERE / "libfoscport.so"
    if not so.exists() or so.stat().st_mtime < (HERE / "fos_cport.c").stat().st_mtime:
        subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    lib = C.CDLL(str(so))
    i64p, f64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.fosc_create.restype = C.c_void_p
    lib.fosc_create.argtypes = [C.c_int64, C.c_int64, i64p, i64p, f64p, f64p, f64p, C.c_int64, i32p, i64p, C.c_int64, i32p, i64p, C.c_int]
    lib.fosc_free.argtypes = [C.c_void_p]
    lib.fosc_kkt_mul.argtypes = [C.c_void_p, f64p, f64p]
    lib.fosc_prox_affine.argtypes = [C.c_void_p, f64p, f64p]
    lib.fosc_prox_cones.argtypes = [C.c_void_p, f64p, f64p]
    lib.fosc_gap_step.argtypes = [C.c_void_p, f64p, C.c_double, C.c_double, C.c_double]
    lib.fosc_gapa_step.argtypes = [C.c_void_p, f64p, C.c_double, C.c_double, f64p]
    lib.fosc_fista_step.argtypes = [C.c_void_p, f64p, C.c_double, f64p, f64p, f64p]
    lib.fosc_set_affine_state.argtypes = [C.c_void_p, f64p, C.c_int64]
    lib.fosc_get_cgiter.restype = C.c_int64
    lib.fosc_get_cgiter.argtypes = [C.c_void_p]
    lib.fosc_max_threads.restype = C.c_int
    _LIB = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class CPort:
    """One HSDE problem in the C port.  K1/K2: lists of (cone code, length) with the codes of fos_oracle.CONE_CODES."""

    def __init__(self, A, b, c, K1, K2, threads=1):
        lib = load()
        A = sp.csc_matrix(A)
        A.sort_indices()
        m, n = A.shape
        self.m, self.n, self.N = m, n, 2 * (m + n + 1)
        colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
        rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
        nz = np.ascontiguousarray(A.data, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        t1 = np.array([k for k, _ in K1], dtype=np.int32)
        l1 = np.array([ln for _, ln in K1], dtype=np.int64)
        t2 = np.array([k for k, _ in K2], dtype=np.int32)
        l2 = np.array([ln for _, ln in K2], dtype=np.int64)
        self._lib = lib
        self._h = lib.fosc_create(m, n, _p(colptr, C.c_int64), _p(rowval, C.c_int64), _p(nz, C.c_double), _p(b, C.c_double),
                                  _p(c, C.c_double), len(t1), _p(t1, C.c_int32), _p(l1, C.c_int64), len(t2), _p(t2, C.c_int32),
                                  _p(l2, C.c_int64), int(threads))
        if not self._h:
            raise ValueError("fosc_create failed (cone lengths do not add up to m / n?)")

    def close(self):
        if self._h:
            self._lib.fosc_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def kkt_mul(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.N)
        self._lib.fosc_kkt_mul(self._h, _p(y, C.c_double), _p(x, C.c_double))
        return y

    def prox_affine(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.N)
        self._lib.fosc_prox_affine(self._h, _p(y, C.c_double), _p(x, C.c_double))
        return y

    def prox_cones(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.N)
        if self._lib.fosc_prox_cones(self._h, _p(y, C.c_double), _p(x, C.c_double)):
            raise NotImplementedError("cone kind not in the C port")
        return y

    def gap_step(self, x, alpha, alpha1, alpha2):
        """In place on x (contiguous float64)."""
        if self._lib.fosc_gap_step(self._h, _p(x, C.c_double), alpha, alpha1, alpha2):
            raise NotImplementedError("cone kind not in the C port")

    def gapa_step(self, x, alpha, beta, alpha12):
        a = C.c_double(alpha12)
        if self._lib.fosc_gapa_step(self._h, _p(x, C.c_double), alpha, beta, C.byref(a)):
            raise NotImplementedError("cone kind not in the C port")
        return a.value

    def fista_step(self, x, alpha, y, xold, t):
        """In place on x, y, xold (contiguous float64); returns the new t."""
        tt = C.c_double(t)
        if self._lib.fosc_fista_step(self._h, _p(x, C.c_double), alpha, _p(y, C.c_double), _p(xold, C.c_double), C.byref(tt)):
            raise NotImplementedError("cone kind not in the C port")
        return tt.value

    def set_affine_state(self, xinit, i):
        xinit = np.ascontiguousarray(xinit, dtype=np.float64)
        self._lib.fosc_set_affine_state(self._h, _p(xinit, C.c_double), int(i))

    def cgiter(self):
        return int(self._lib.fosc_get_cgiter(self._h))


def max_threads():
    return int(load().fosc_max_threads())
