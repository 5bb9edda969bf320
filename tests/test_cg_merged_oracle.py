"""The merged-reduction CG restatement in the oracle (`conjugategradient_merged`: NOT a reference function -- the arithmetic of
the HIP path's FOS_CG_MERGED_* variants) against the reference recurrence restated in `conjugategradient`
(src/utilities/conjugategradients.jl:31-55): same iterates while rounding has not been amplified, same iteration-count
convention, same stop rule; and the reference's own CG property test (test/conjugateGradient.jl) on it."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import fos_oracle as orc


def _kkt(rng, m=60, n=90, density=0.1):
    A = sp.random(m, n, density=density, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    Q = orc.HSDEMatrixQ(A, rng.standard_normal(m), rng.standard_normal(n))
    return orc.KKTMatrix(Q), 2 * (m + n + 1)


def _run(fn, M, x0, rhs, tol, maxit):
    N = x0.shape[0]
    x = x0.copy()
    bufs = [np.empty(N) for _ in range(4 if fn is orc.conjugategradient_merged else 3)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        it = fn(x, M, rhs, *bufs, tol=tol, max_iters=maxit)
    return x, it


def test_merged_matches_reference_recurrence_early_and_at_convergence():
    rng = np.random.default_rng(11)
    M, N = _kkt(rng)
    rhs, x0 = rng.standard_normal(N), rng.standard_normal(N)
    for k in (1, 2, 3, 5, 8):
        xr, itr = _run(orc.conjugategradient, M, x0, rhs, 1e-300, k)
        xm, itm = _run(orc.conjugategradient_merged, M, x0, rhs, 1e-300, k)
        assert itr == itm == k                                    # max_iters cap returns max_iters (:42)
        assert np.linalg.norm(xr - xm) <= 1e-12 * np.linalg.norm(xr), k
    tol = N * orc.EPS
    xr, itr = _run(orc.conjugategradient, M, x0, rhs, tol, 10000)
    xm, itm = _run(orc.conjugategradient_merged, M, x0, rhs, tol, 10000)
    assert abs(itr - itm) <= 6, (itr, itm)
    y = np.empty(N)
    M.mul(y, xm)
    assert np.linalg.norm(y - rhs) <= 50 * tol                    # the recursive residual met ||r|| <= tol; the true one is close
    assert np.linalg.norm(xr - xm) <= 1e-9 * np.linalg.norm(xr)
    # loose tolerance: both stop by the same rule within a few iterations of each other
    xr, itr = _run(orc.conjugategradient, M, x0, rhs, 1e-3, 10000)
    xm, itm = _run(orc.conjugategradient_merged, M, x0, rhs, 1e-3, 10000)
    assert abs(itr - itm) <= 4, (itr, itm)
    M.mul(y, xm)
    assert np.linalg.norm(y - rhs) <= 1e-3 * (1 + 1e-6)


def test_merged_iteration_count_semantics():
    """count starts at 1, at least one iteration always runs (conjugategradients.jl:36-52)."""
    A = orc._PlainMatrix(np.eye(5) * 2.0)
    b = np.ones(5)
    x = np.zeros(5)
    it = orc.conjugategradient_merged(x, A, b, *[np.empty(5) for _ in range(4)], tol=1e-12, max_iters=10)
    assert it == 1 and np.allclose(x, 0.5)
    x = np.full(5, 0.5)        # exact warm start: 0/0 -> NaN, never stops before max_iters -- as the reference recurrence
    with pytest.warns(UserWarning):
        it = orc.conjugategradient_merged(x, A, b, *[np.empty(5) for _ in range(4)], tol=1e-12, max_iters=10)
    assert it == 10 and np.all(np.isnan(x))


def test_merged_on_the_reference_cg_property_test():
    """test/conjugateGradient.jl:3-33 (dense SPD 1000 x 1000) on the merged recurrence."""
    rng = np.random.default_rng(2)
    A0 = rng.random((1000, 1000))
    A = A0.T @ A0
    b = rng.standard_normal(1000)
    x = rng.standard_normal(1000)
    op = orc._PlainMatrix(A)
    bufs = [np.empty(1000) for _ in range(4)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orc.conjugategradient_merged(x, op, b, *bufs, max_iters=100)
        orc.conjugategradient_merged(x, op, b, *bufs, max_iters=5000)
    n1 = np.linalg.norm(A @ x - b)
    assert n1 < 1e-5                                                          # :26
    xcopy = x + 1e-5 * rng.standard_normal(1000)
    n2 = np.linalg.norm(A @ xcopy - b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orc.conjugategradient_merged(xcopy, op, b, *bufs, max_iters=100)
    assert np.linalg.norm(A @ xcopy - b) < 10 * n2                            # :33


def test_affine_prox_with_merged_variant_meets_the_stop_rule():
    """AffinePlusLinear.prox (affinepluslinear.jl:83-126) with cg_variant = "merged": same counter / tolerance schedule, result
    within 2 tol of the exact projection."""
    rng = np.random.default_rng(6)
    A = sp.random(60, 90, density=0.1, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    l = 151
    Q = orc.HSDEMatrixQ(A, rng.standard_normal(60), rng.standard_normal(90))
    S = orc.AffinePlusLinear(Q, np.zeros(l), np.zeros(l), 1, decreasing_accuracy=True)
    S.cg_variant = "merged"
    R = orc.AffinePlusLinear(Q, np.zeros(l), np.zeros(l), 1, decreasing_accuracy=True)
    Qd = Q.todense()
    Md = np.block([[np.eye(l), Qd.T], [Qd, -np.eye(l)]])
    for call in range(1, 7):
        x = rng.standard_normal(2 * l)
        tol = S.tolerance()
        y, yr = np.empty(2 * l), np.empty(2 * l)
        S.prox(y, x)
        R.prox(yr, x)
        assert S.i == R.i == call + 1
        assert abs(S.getcgiter() - R.getcgiter()) <= 4
        exact = np.linalg.solve(Md, np.concatenate([x[:l] + Qd.T @ x[l:], np.zeros(l)]))
        assert np.linalg.norm(y - exact) <= 2 * tol + 1e-12
