"""Seeded instances of the reference's Feasibility test (test/testfeasibility.jl:4-12: IndAffine(A, b) n IndBox(0, Inf) with
b = A xsol) with numpy's numbers; the reference's literal draw and its outcomes are in tests/test_reference_known_answers.py
and tests/test_gpu_known_answers.py)."""
import numpy as np


def affine_box_instance(seed=2, m=50, n=100, boundary=True):
    rng = np.random.default_rng(seed)
    xs = rng.standard_normal(n)
    xs = np.maximum(xs, 0.0) if boundary else np.abs(xs)       # a point of the intersection (on its boundary / inside)
    A = rng.standard_normal((m, n))
    return A, A @ xs


ALGS = {
    "DR": lambda M, **kw: M.DR(**kw),
    "AP": lambda M, **kw: M.AP(**kw),
    "GAP": lambda M, **kw: M.GAP(0.8, 1.5, 1.6, **kw),
    "GAPA": lambda M, **kw: M.GAPA(0.9, 0.3, **kw),
    "FISTA": lambda M, **kw: M.FISTA(**kw),
    "Dykstra": lambda M, **kw: M.Dykstra(**kw),
}
GAPP = lambda M, **kw: M.GAPP(0.8, 1.5, 1.6, **kw)


MIXED_CONES = [("NonNeg", 7), ("SOC", 6), ("SDP", 21), ("Free", 3), ("SOCRotated", 5), ("SDP", 2080), ("NonPos", 4), ("SOC", 1), ("SDP", 3),
               ("ExpPrimal", 3), ("ExpDual", 3), ("Zero", 2)]


def cone_instance(orc, cones=MIXED_CONES, m_frac=0.3, seed=11):
    """A x = b with b = A x0, x0 a point of the cone product (the projection of a random vector): IndAffine n ConeProduct is non-empty."""
    rng = np.random.default_rng(seed)
    n = sum(l for _, l in cones)
    K = orc.ConeProduct.from_lengths([(orc.CONE_CODES[k], l) for k, l in cones])
    x0 = np.empty(n)
    K.prox(x0, rng.standard_normal(n))
    m = max(1, int(m_frac * n))
    A = rng.standard_normal((m, n)) / np.sqrt(n)
    return A, A @ x0, K, n
