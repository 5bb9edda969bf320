"""
Pins the oracle (oracle/fos_oracle.py) against the reference's OWN tests, restated.
Every test cites the reference test file it follows (paths under /root/reference).
Where the reference test is an identity or a property (any seed) it is run on numpy data; RNG-free literals are
used verbatim.  The literals that depend on Julia's seeded draws (the NNLS optima, the feasibility test's outcomes)
are in tests/test_reference_known_answers.py, on the regenerated draws.
CPU only (no gpu marker).
"""
import math

import numpy as np
import pytest
import scipy.optimize
import scipy.sparse as sp

import fos_oracle as orc


def isapprox(a, b, rtol=math.sqrt(np.finfo(float).eps)):
    """Julia's `≈` for vectors: norm(a-b) <= rtol*max(norm(a),norm(b))."""
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) <= rtol * max(np.linalg.norm(a), np.linalg.norm(b))


# ------------------------------------------------------------------ test/conjugateGradient.jl


def test_conjugate_gradient_dense_spd():
    """test/conjugateGradient.jl:3-33 (A = rand(1000,1000); A = A'A)."""
    rng = np.random.default_rng(2)
    A0 = rng.random((1000, 1000))
    A = A0.T @ A0
    b = rng.standard_normal(1000)
    x = rng.standard_normal(1000)
    op = orc._PlainMatrix(A)
    r, p, z = np.empty(1000), np.empty(1000), np.empty(1000)
    import warnings
    with pytest.warns(UserWarning):
        it = orc.conjugategradient(x, op, b, r, p, z, max_iters=100)          # :21
    assert it == 100
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        it = orc.conjugategradient(x, op, b, r, p, z, max_iters=5000)         # :23
    n1 = np.linalg.norm(A @ x - b)
    assert n1 < 1e-5                                                          # :26
    xcopy = x + 1e-5 * rng.standard_normal(1000)                              # :28
    n2 = np.linalg.norm(A @ xcopy - b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orc.conjugategradient(xcopy, op, b, r, p, z, max_iters=100)           # :30
    n3 = np.linalg.norm(A @ xcopy - b)
    assert n3 < 10 * n2                                                       # :33


def test_cg_iteration_count_semantics():
    """conjugategradients.jl:36-52: count starts at 1, at least one iteration always runs."""
    A = orc._PlainMatrix(np.eye(5) * 2.0)
    b = np.ones(5)
    x = np.full(5, 0.5)        # exact solution as warm start
    with pytest.warns(UserWarning):
        it = orc.conjugategradient(x, A, b, np.empty(5), np.empty(5), np.empty(5), tol=1e-12, max_iters=10)
    # r == 0 -> alpha = 0/0 = NaN in the reference too (Julia does not trap); it never guards this:
    # norm(NaN) <= tol is false, so it runs to max_iters with a NaN iterate.
    assert it == 10 and np.all(np.isnan(x))
    x = np.zeros(5)
    it = orc.conjugategradient(x, A, b, np.empty(5), np.empty(5), np.empty(5), tol=1e-12, max_iters=10)
    assert it == 1 and np.allclose(x, 0.5)


# ------------------------------------------------------------------ test/HSDEAffine.jl


def _getQ1Q2(A, rng):
    m, n = A.shape
    b = rng.standard_normal(m)
    c = rng.standard_normal(n)
    Ad = A.toarray() if sp.issparse(A) else A
    Q1 = np.block([[np.zeros((n, n)), Ad.T, c[:, None]],
                   [-Ad, np.zeros((m, m)), b[:, None]],
                   [-c[None, :], -b[None, :], np.zeros((1, 1))]])            # :7-9
    Q2 = orc.HSDEMatrixQ(A, b, c)                                            # :11
    return Q1, Q2


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_hsde_affine(kind):
    """test/HSDEAffine.jl:26-90 -- Q mul!, transpose mul!, HSDEMatrix mul!, HSDEMatrix prox!."""
    rng = np.random.default_rng(1)
    if kind == "dense":
        A = rng.standard_normal((100, 200))                                   # :84-86
    else:
        A = sp.random(1000, 2000, density=0.001, format="csc", random_state=rng,
                      data_rvs=rng.standard_normal)                           # :88-90
        if A.shape[0] > 400:   # dense solve below is O(l^3): keep the sparse case at the reference's size/4
            A = sp.random(250, 500, density=0.004, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    m, n = A.shape
    Q1, Q2 = _getQ1Q2(A, rng)
    l = m + n + 1
    assert Q2.shape == (l, l)
    # testHSDEQ_A_mul_B  :26-43
    rhs1 = rng.standard_normal(l)
    rhs2 = rhs1.copy()
    y2 = rng.standard_normal(l)
    y1 = Q1 @ rhs1
    Q2.mul(y2, rhs2)
    assert np.array_equal(rhs1, rhs2)                                         # inputs not mutated :35
    assert isapprox(y1, y2)
    y1 = Q1.T @ rhs1
    Q2.mul_t(y2, rhs2)
    assert np.array_equal(rhs1, rhs2)                                         # :41
    assert isapprox(y1, y2)
    # testHSDEMatrix_A_mul_B  :45-62
    M1 = np.block([[np.eye(l), Q1.T], [Q1, -np.eye(l)]])                     # :20-21
    M2 = orc.HSDEMatrix(Q2)
    assert M2.shape == (2 * l, 2 * l)
    rhs1 = rng.standard_normal(2 * l)
    rhs2 = rhs1.copy()
    y2 = rng.standard_normal(2 * l)
    M2.mul(y2, rhs2)
    assert np.array_equal(rhs1, rhs2)
    assert isapprox(M1 @ rhs1, y2)
    M2.mul_t(y2, rhs2)
    assert isapprox(M1.T @ rhs1, y2)
    # prox  :71-81  (y3 = M1\b with v = Q u;  IndAffine([Q -I],0) is the same projection)
    b = rng.standard_normal(2 * l)
    y2 = np.empty(2 * l)
    M2.prox(y2, b)
    y3 = np.linalg.solve(M1, b)
    y3[l:] = Q1 @ y3[:l]
    assert isapprox(y2, y3)
    # projection onto {Qu = v} computed independently: u = (I + Q'Q)^-1 (b1 + Q'b2)
    u = np.linalg.solve(np.eye(l) + Q1.T @ Q1, b[:l] + Q1.T @ b[l:])
    assert isapprox(y2, np.concatenate([u, Q1 @ u]))


# ------------------------------------------------------------------ test/affinepluslinear.jl


def test_kkt_matrix_and_affine_plus_linear():
    """test/affinepluslinear.jl:4-68."""
    rng = np.random.default_rng(10)
    A = rng.standard_normal((10, 20))
    M1 = np.block([[np.eye(20), A.T], [A, -np.eye(10)]])                     # :8
    M2 = orc.KKTMatrix(A)
    x = rng.standard_normal(30)
    y2 = rng.standard_normal(30)
    M2.mul(y2, x)
    assert isapprox(M1 @ x, y2)                                               # :15
    M2.mul_t(y2, x)
    assert isapprox(M1.T @ x, y2)                                             # :19
    x0 = rng.standard_normal(20)
    z0 = rng.standard_normal(10)
    q = rng.standard_normal(20)
    b = rng.standard_normal(10)
    # beta = 1   :28-47
    S2 = orc.AffinePlusLinear(A, b, q, 1)
    y2 = np.empty(30)
    S2.prox(y2, np.concatenate([x0, z0]))
    y3 = np.linalg.solve(M1, np.concatenate([x0 - q + A.T @ z0, b]))         # :46
    assert isapprox(y3, y2)
    assert S2.i == 2 and S2.getcgiter() >= 1
    # beta = -1  :50-68
    S2 = orc.AffinePlusLinear(A, b, q, -1)
    y2 = np.empty(30)
    S2.prox(y2, np.concatenate([x0, z0]))
    Mm = np.block([[np.eye(20), -A.T], [A, np.eye(10)]])
    y3 = np.linalg.solve(Mm, np.concatenate([x0 - q - A.T @ z0, b]))         # :67
    assert isapprox(y3, y2)


def test_affine_plus_linear_tolerance_schedule():
    """affinepluslinear.jl:108-114: tol = max(0.2^sqrt(i), l*eps), i counted from 1, +1 per call."""
    A = orc.HSDEMatrixQ(sp.identity(3, format="csc"), np.ones(3), np.ones(3))
    S = orc.AffinePlusLinear(A, np.zeros(7), np.zeros(7), 1, decreasing_accuracy=True)
    assert S.tolerance() == 0.2
    S.i = 4
    assert S.tolerance() == pytest.approx(0.04)
    S.i = 10 ** 6
    assert S.tolerance() == 7 * np.finfo(float).eps
    S2 = orc.AffinePlusLinear(A, np.zeros(7), np.zeros(7), 1)
    assert S2.tolerance() == 7 * np.finfo(float).eps


# ------------------------------------------------------------------ test/testPSD.jl

YS = np.array([[-0.0064709, -0.22443], [-0.22443, -1.02411]])                 # testPSD.jl:3-4
# SURVEY.md section 4 / BASELINE.md: P_PSD(ys) computed independently (eigs -1.0714..., 0.04082...)
PSD_KNOWN = np.array([[0.03909044662082823, -0.00823811392936668],
                      [-0.00823811392936668, 0.00173614084718757]])


def test_psd_known_answer_matrix_and_scaled_vector():
    """test/testPSD.jl:14-19: IndPSD() projection of the fixed 2x2 matrix (known answer, atol 1e-8)."""
    X = orc.prox_psd_matrix(YS)
    assert np.allclose(X, PSD_KNOWN, atol=1e-12, rtol=0)
    # IndPSD(scaling=true) on svec(ys): the path the solvers use (cones.jl:11)
    r2 = math.sqrt(2)
    xs = np.array([YS[0, 0], r2 * YS[1, 0], YS[1, 1]])
    y = np.empty(3)
    orc.prox_psd_scaled(y, xs)
    assert np.allclose(y, [PSD_KNOWN[0, 0], r2 * PSD_KNOWN[1, 0], PSD_KNOWN[1, 1]], atol=1e-12, rtol=0)
    # closed form for 2x2: eigen-decomposition by hand
    tr, det = np.trace(YS), np.linalg.det(YS)
    lam_max = tr / 2 + math.sqrt(tr * tr / 4 - det)
    assert lam_max == pytest.approx(0.0408265874680158, abs=1e-12)


def test_psd_dr_solve_matches_projection(pkg):
    """test/testPSD.jl:22-25: DR(eps=1e-8) on  min ||vec(Y-ys)|| s.t. Y PSD  == projection (atol 1e-8)."""
    prob = pkg.workloads.psd2x2_reference_problem()
    model = orc.Model(prob.A, prob.b, prob.c, [(orc.CONE_CODES[k], l) for k, l in prob.K1],
                      [(orc.CONE_CODES[k], l) for k, l in prob.K2])
    sol = orc.solve(model, orc.DR(eps=1e-8, verbose=0))
    assert sol.status == "Optimal"
    v = sol.x[1:]
    Y = np.array([[v[0], v[1] / math.sqrt(2)], [v[1] / math.sqrt(2), v[2]]])
    assert np.allclose(Y, PSD_KNOWN, atol=1e-8, rtol=0)
    assert sol.x[0] == pytest.approx(np.linalg.norm(PSD_KNOWN - YS), abs=1e-7)


# ------------------------------------------------------------------ test/testDRandGAPA.jl


@pytest.fixture(scope="module")
def nnls(pkg):
    prob = pkg.workloads.c1_readme_nnls(seed=2)
    xs, rnorm = scipy.optimize.nnls(prob.meta["Ad"], prob.meta["bd"])
    model = orc.Model(prob.A, prob.b, prob.c, [(orc.CONE_CODES[k], l) for k, l in prob.K1],
                      [(orc.CONE_CODES[k], l) for k, l in prob.K2])
    return prob, model, xs, rnorm ** 2


def test_readme_nnls_dr(nnls):
    """test/testDRandGAPA.jl:9-23: DR(eps=1e-8) -> :Optimal, optval, min(x) > -10 eps.
    (numpy data, scipy.optimize.nnls as the optimum; the reference's literal optimum on the reference's own draw is
    checked in tests/test_reference_known_answers.py.)"""
    prob, model, xs, opt = nnls
    eps = 1e-8
    lines = []
    sol = orc.solve(model, orc.DR(eps=eps, verbose=1), out=lines)
    assert sol.status == "Optimal"
    assert sol.obj_val == pytest.approx(opt, rel=1e-6)
    n = prob.meta["n"]
    assert abs(min(sol.x[:n].min(), 0.0)) < 10 * eps                           # :23
    assert np.max(np.abs(sol.x[:n] - xs)) < 1e-5
    # printed table: header + rows + "Found solution"  (HSDEStatus.jl:73-91, testprint.jl:15-19)
    assert lines[2] == orc.HEADER_CG
    assert lines[1] == "-" * 81
    assert lines[4].startswith("   100|")
    assert any(s.startswith("Found solution i=") for s in lines)
    # history keys  (HSDEStatus.jl:127-131, :44-47)
    for key in ("p", "d", "g", "ctx", "bty", "kappa", "tau", "t", "cgiter"):
        assert key in model.history and len(model.history[key]) >= 1


def test_readme_nnls_gapa_and_beta(nnls):
    """test/testDRandGAPA.jl:28-33,44-49."""
    prob, model, xs, opt = nnls
    n = prob.meta["n"]
    sol = orc.solve(model, orc.GAPA(eps=1e-4, verbose=0))
    assert sol.status == "Optimal"
    assert abs((sol.obj_val - opt) / opt) < 2e-3                              # :32
    assert np.max(np.abs(sol.x[:n] - xs)) < 1e-2
    sol = orc.solve(model, orc.GAPA(0.5, 0.9, eps=1e-9, verbose=0))
    assert sol.status == "Optimal"
    assert abs((sol.obj_val - opt) / opt) < 1e-6                              # :48 (1e-8 vs DR's own optimum there)
    assert np.max(np.abs(sol.x[:n] - xs)) < 1e-5


def test_readme_gap_max_iters_forced_check(nnls):
    """README.md:26 GAP(0.5,2.0,2.0,max_iters=2000); solverwrapper.jl:31-34 forced check when the
    last iteration was not a check iteration."""
    prob, model, xs, opt = nnls
    sol = orc.solve(model, orc.GAP(0.5, 2.0, 2.0, max_iters=150, verbose=0))
    assert sol.status in ("Indeterminate", "Optimal")
    st = sol.status_obj
    assert st.checked and st.i == 150
    assert [i for i, _ in model.history["p"]] == [100, 150]


def test_fista_and_dykstra_run(nnls):
    prob, model, xs, opt = nnls
    for alg in (orc.FISTA(eps=1e-3, verbose=0, max_iters=300), orc.Dykstra(eps=1e-3, verbose=0, max_iters=300),
                orc.AP(eps=1e-3, verbose=0, max_iters=300)):
        sol = orc.solve(model, alg)
        assert sol.status in ("Optimal", "Indeterminate")
        assert np.isfinite(sol.obj_val)


# ------------------------------------------------------------------ test/testprint.jl


def test_print_formats():
    """test/testprint.jl:15-19 header strings and row prefix; HSDEStatus.jl:85-91 formats."""
    assert orc.HEADER_CG == " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | cg  | time"
    assert orc.HEADER_DIRECT == " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | time"
    row = orc.format_status_iter(100, 1.234e-3, 5e-5, 0.5, -3.25, 7.5, 1e-9, 12, 2.5e9)
    assert row == "   100| 1.23e-03  5.00e-05  5.00e-01 -3.25e+00 -7.50e+00  1.00e-09   12  2.5e+00s"
    row = orc.format_status_iter(200, 1.0, 2.0, 3.0, 4.0, -5.0, 6.0, None, 1e9)
    assert row[:7] == "   200|"
    assert row.endswith(" 1.0e+00s")


def test_print_problem_gapa(pkg):
    """test/testprint.jl:21-46: GAPA(0.8,0.9,eps=1e-8,checki=100) on min ||Ax-b|| s.t. sum(x)==sum(xbar),
    A = sprandn(n,2n,0.1).  Conic form: vars (x, t): (t, Ax-b) in SOC(n+1), sum(x) = sum(xbar) (Zero)."""
    rng = np.random.default_rng(10)
    n = 60
    A = sp.random(n, 2 * n, density=0.1, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    xbar = rng.standard_normal(2 * n)
    b = A @ xbar
    nv = 2 * n + 1
    top = sp.hstack([sp.csc_matrix((1, 2 * n)), sp.csc_matrix([[-1.0]])])
    mid = sp.hstack([-A, sp.csc_matrix((n, 1))])
    bot = sp.hstack([sp.csc_matrix(np.ones((1, 2 * n))), sp.csc_matrix((1, 1))])
    Ac = sp.vstack([top, mid, bot]).tocsc()
    bc = np.concatenate([[0.0], -b, [xbar.sum()]])
    c = np.zeros(nv)
    c[-1] = 1.0
    model = orc.Model(Ac, bc, c, [(orc.CONE_SOC, n + 1), (orc.CONE_ZERO, 1)], [(orc.CONE_FREE, nv)])
    lines = []
    sol = orc.solve(model, orc.GAPA(0.8, 0.9, verbose=2, debug=0, eps=1e-8, checki=100, max_iters=20000), out=lines)
    assert lines[2] == orc.HEADER_CG
    assert lines[4][:7] == "   100|"
    assert lines[5][:7] == "   200|"
    assert sol.status == "Optimal"
    x = sol.x[:2 * n]
    assert abs(x.sum() - xbar.sum()) < 1e-6
    assert np.max(np.abs(A @ x - b)) < 1e-6
    assert "p" not in model.history            # debug=0: no savedata  (HSDEStatus.jl:39-41)
    assert "cgiter" in model.history


# ------------------------------------------------------------------ cones: properties (SOC is parity-unpinned upstream)


@pytest.mark.parametrize("code", [orc.CONE_SOC, orc.CONE_SOCROT, orc.CONE_SDP, orc.CONE_NONNEG, orc.CONE_NONPOS])
def test_cone_projection_properties(code):
    """Idempotence, Moreau decomposition x = P_K(x) - P_K*(-x) with orthogonality (cones.jl:80-85),
    membership."""
    rng = np.random.default_rng(5)
    for trial in range(30):
        d = 10 if code == orc.CONE_SDP else 7
        x = rng.standard_normal(d) * (3 if trial % 2 else 0.3)
        y = np.empty(d)
        orc.cone_prox(code, y, x)
        yy = np.empty(d)
        orc.cone_prox(code, yy, y)
        assert np.allclose(yy, y, atol=1e-12)
        yd = np.empty(d)
        orc.cone_prox_dual(code, yd, -x)       # P_K*(-x)
        assert np.allclose(y - yd, x, atol=1e-12)
        assert abs(np.dot(y, yd)) < 1e-10
        if code == orc.CONE_SOC:
            assert y[0] >= np.linalg.norm(y[1:]) - 1e-12
        if code == orc.CONE_SOCROT:
            assert y[0] >= -1e-12 and y[1] >= -1e-12 and 2 * y[0] * y[1] >= np.dot(y[2:], y[2:]) - 1e-10
        if code == orc.CONE_SDP:
            M = orc.svec_to_mat(y, 4)
            M[np.tril_indices(4, -1)] /= math.sqrt(2)
            M = np.tril(M) + np.tril(M, -1).T
            assert np.linalg.eigvalsh(M).min() > -1e-12


def test_soc_edge_cases():
    y = np.empty(4)
    orc.prox_soc(y, np.array([0.0, 0, 0, 0]))
    assert np.array_equal(y, np.zeros(4))
    orc.prox_soc(y, np.array([-5.0, 1, 2, 2]))          # t <= -||v||: origin
    assert np.array_equal(y, np.zeros(4))
    x = np.array([5.0, 1, 2, 2])                        # inside: identity
    orc.prox_soc(y, x)
    assert np.array_equal(y, x)
    orc.prox_soc(y, np.array([0.0, 3, 0, 4]))           # boundary case r = 1/2
    assert np.allclose(y, [2.5, 1.5, 0, 2.0])
    y1 = np.empty(1)
    orc.prox_soc(y1, np.array([-2.0]))                  # SOC(1) = R+
    assert y1[0] == 0.0
    orc.prox_soc(y1, np.array([2.0]))
    assert y1[0] == 2.0


def test_dual_cone_product_layout():
    """cones.jl:122-142 on the z layout [x; y; tau; r; s; kappa]."""
    K1 = orc.ConeProduct.from_lengths([(orc.CONE_ZERO, 2), (orc.CONE_NONNEG, 2)])
    K2 = orc.ConeProduct.from_lengths([(orc.CONE_FREE, 1), (orc.CONE_NONNEG, 1)])
    S2 = orc.DualConeProduct(K1, K2)
    assert (S2.m, S2.n) == (4, 2)
    z = -np.arange(1.0, 15.0)
    z[::2] *= -1                       # 1,-2,3,-4,...
    out = np.empty(14)
    S2.prox(out, z)
    #        x: Free, NonNeg | y: Zero*->Free x2, NonNeg x2 | tau | r: Free*->0, NonNeg | s: Zero x2, NonNeg x2 | kappa
    expect = [1, 0, 3, -4, 5, 0, 7, 0, 9, 0, 0, 0, 13, 0]
    assert np.array_equal(out, np.array(expect, dtype=float))
    with pytest.raises(AssertionError):
        orc.ConeProduct([(orc.CONE_ZERO, 0, 2), (orc.CONE_ZERO, 3, 2)])      # gap -> cones.jl:69 assertion


def test_exponential_cone_properties():
    """IndExpPrimal/IndExpDual restatement (parity unpinned upstream): Moreau decomposition x = P_K(x) - P_K*(-x)
    with P_K(x) orthogonal to P_K*(-x), membership, closed-form cases."""
    rng = np.random.default_rng(3)
    for trial in range(200):
        x = rng.standard_normal(3) * 10.0 ** rng.uniform(-2, 2)
        y, yd = np.empty(3), np.empty(3)
        orc.prox_exp_primal(y, x)
        orc.prox_exp_dual(yd, -x)
        assert np.linalg.norm(y - yd - x) <= 1e-12 * max(1.0, np.linalg.norm(x))
        assert abs(y @ yd) <= 1e-6 * max(1.0, np.linalg.norm(x) ** 2)      # accuracy of the SCS-style bisection in extreme cases
        r, s, t = y
        assert (s > 0 and s * math.exp(r / s) <= t * (1 + 1e-7) + 1e-9) or (r <= 1e-9 and abs(s) <= 1e-9 and t >= -1e-9)
    y = np.empty(3)
    orc.prox_exp_primal(y, np.array([0.5, 1.0, 3.0]))
    assert np.array_equal(y, [0.5, 1.0, 3.0])
    orc.prox_exp_primal(y, np.array([1.0, -1.0, -5.0]))          # in the polar cone
    assert np.array_equal(y, np.zeros(3))
    orc.prox_exp_primal(y, np.array([-1.0, -2.0, 3.0]))          # r < 0, s < 0: analytical
    assert np.array_equal(y, [-1.0, 0.0, 3.0])


# ------------------------------------------------------------------------------------------------ direct = true
def test_direct_affine_projection_is_the_exact_projection():
    """HSDE.jl:12-15 (direct=true): S1 = IndAffine([Q -I], 0).  Its prox is the orthogonal projection onto {Q u = v}: the result
    is feasible to rounding, the displacement is orthogonal to the subspace, and AffinePlusLinear's CG (the direct=false S1)
    run to its floor reaches the same point -- the two S1 of the reference are the same operator (test/testDRandGAPA.jl:28-43
    solves the same problem both ways)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(23)
    m, n = 17, 11
    A = sp.random(m, n, density=0.4, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    Q = orc.HSDEMatrixQ(A, b, c)
    l = n + m + 1
    S1d = orc.IndAffineDirect(Q)
    x = rng.standard_normal(2 * l)
    y = np.empty(2 * l)
    S1d.prox(y, x)
    Qd = Q.todense()
    assert np.linalg.norm(Qd @ y[:l] - y[l:]) < 1e-12 * np.linalg.norm(y)
    # x - y is orthogonal to every [u; Q u]
    for _ in range(5):
        u = rng.standard_normal(l)
        t = np.concatenate([u, Qd @ u])
        assert abs((x - y) @ t) < 1e-11 * np.linalg.norm(t) * np.linalg.norm(x)
    S1 = orc.AffinePlusLinear(Q, np.zeros(l), np.zeros(l), 1, decreasing_accuracy=False)
    y2 = np.empty(2 * l)
    S1.prox(y2, x)
    assert np.linalg.norm(y2 - y) < 1e-9 * np.linalg.norm(y)


def test_direct_solve_prints_without_cg_column_and_matches_indirect():
    """test/testprint.jl:15-16,49-61 and testDRandGAPA.jl:28-43: with direct=true the header has no cg column, rows carry no cg
    count, history has no :cgiter, and the solve reaches the same optimum as the CG path."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    m, n = 30, 20
    A = sp.random(m, n, density=0.3, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    x0 = np.maximum(rng.standard_normal(n), 0.0)
    r0 = np.where(x0 > 0, 0.0, rng.random(n))
    y0 = rng.standard_normal(m)
    b, c = A @ x0, r0 - A.T @ y0                    # equality-constrained LP: K1 = Zero, K2 = NonNeg
    om = orc.Model(A, b, c, [(orc.CONE_CODES["Zero"], m)], [(orc.CONE_CODES["NonNeg"], n)])
    out_d, out_i = [], []
    sd = orc.solve(om, orc.GAPA(0.8, 0.9, direct=True, verbose=2, debug=1, eps=1e-8, checki=100, max_iters=20000), out=out_d)
    si = orc.solve(om, orc.GAPA(0.8, 0.9, verbose=2, debug=1, eps=1e-8, checki=100, max_iters=20000), out=out_i)
    assert out_d[2] == " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | time"      # o12
    assert out_i[2] == " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | cg  | time"  # o11
    assert out_d[4].startswith("   100|") and len(out_d[4].split()) == len(out_i[4].split()) - 1
    assert sd.status == "Optimal" == si.status
    assert sd.obj_val == pytest.approx(si.obj_val, abs=1e-6)
    assert float(c @ sd.x) == pytest.approx(float(c @ x0), abs=1e-6)


# ------------------------------------------------------------------ src/wrappers/linesearch.jl (test/linesearch.jl configuration)


def test_linesearch_wrapper_oracle(nnls):
    """LineSearchWrapper(GAP(0.5, 1.0, 1.0), lsinterval) as test/linesearch.jl:11 builds it (that script is not part of the
    reference's runtests and its literal optimum depends on Julia's RNG, so this pins the restatement's own semantics,
    linesearch.jl:36-75): plain steps between searches; a search evaluates the 31 step lengths 0.1*1.8^(k+1) and takes the
    argmin (first one on ties) of ||x - S2(S1(x))||; the CG tolerance counter advances by one per S1 evaluation; the printed
    lines have the reference's form; only GAP and GAPA are accepted."""
    prob, model, xs, opt = nnls
    ls = 7
    lines = []
    wrap = orc.LineSearchWrapper(orc.GAP(0.5, 1.0, 1.0, eps=1e-8, verbose=0, checki=1, max_iters=3 * ls), lsinterval=ls, out=lines)
    assert wrap.options == dict(eps=1e-8, verbose=0, checki=1, max_iters=3 * ls)
    plain = orc.GAP(0.5, 1.0, 1.0)
    wrap.init(model)
    plain.init(model)
    x = orc.hsde_initialvalue(model)
    xp = x.copy()
    st = orc.NoStatus()
    for i in range(1, ls):                                   # iterations 1 .. ls-1: the wrapped algorithm's own steps
        wrap.step(x, i, st)
        plain.step(xp, i, st)
    assert np.array_equal(x, xp)
    calls_before = wrap.S1.i
    x0 = x.copy()
    wrap.step(x, ls, st)                                     # the search
    assert wrap.S1.i == calls_before + 32                    # one S1 for the base point + 31 trials
    i_log, normres, tests, abest = wrap.log[-1]
    assert i_log == ls and len(tests) == 31
    alphas = [0.1 * 1.8 ** (k + 1) for k in range(31)]
    assert abest == pytest.approx(alphas[int(np.argmin(tests))], rel=1e-12)
    assert lines[0] == "test, %s" % orc.julia_float(normres)
    assert lines[1].startswith("α: 0.18000000000000002, ") and lines[32] == "α: %s" % orc.julia_float(abest) and len(lines) == 33
    # the new iterate lies on the search line through x0, at the chosen step length (linesearch.jl:41,49,70)
    assert np.array_equal(wrap.tmp1, x0) and np.array_equal(x, x0 + abest * wrap.res)
    assert normres == pytest.approx(np.linalg.norm(wrap.res), rel=1e-15)
    # whole solve through solve(): still converges on the README problem
    sol = orc.solve(model, orc.LineSearchWrapper(orc.DR(eps=1e-6, verbose=0, checki=10), lsinterval=50, out=[]))
    assert sol.status == "Optimal" and sol.obj_val == pytest.approx(opt, rel=1e-4)
    with pytest.raises(ValueError):
        orc.LineSearchWrapper(orc.FISTA())
    assert [orc.julia_float(v) for v in (0.18, 1e-5, 0.00012, 1234567.0, 1e6, 100000.0)] == ["0.18", "1.0e-5", "0.00012", "1.234567e6", "1.0e6", "100000.0"]
