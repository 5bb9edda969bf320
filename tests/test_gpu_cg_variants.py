"""
GPU tests of the CG variants the affine projection can run (include/foship.h FOS_CG_*): the merged-reduction recurrence
(two launches, ONE reduction point per iteration; default on sharded handles) against the oracle's restatement of the same
arithmetic (`conjugategradient_merged`) and against the reference recurrence (conjugategradients.jl:31-55), plus the
late-workgroup stress test of the kernel that closes an iteration of the reference recurrence.
"""
import math
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import fos_oracle as orc

pytestmark = pytest.mark.gpu

VARIANTS = ["merged_sweep", "merged_update"]


def _codes(cones):
    return [(orc.CONE_CODES[k], l) for k, l in cones]


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(1e-300, np.linalg.norm(b)))


def _operators():
    rng = np.random.default_rng(5)
    return [("sparse", sp.random(100, 200, density=0.05, format="csc", random_state=rng, data_rvs=rng.standard_normal)),
            ("dual-tiles", sp.csc_matrix(rng.standard_normal((100, 40)))),                                   # every row slot-spread or in a tile
            ("tile-mixed", sp.vstack([sp.csc_matrix(rng.standard_normal((48, 150))),
                                      sp.random(120, 150, density=0.05, format="csc", random_state=rng, data_rvs=rng.standard_normal),
                                      sp.csc_matrix(rng.standard_normal((20, 150)))]).tocsc())]


def _ocg(fn, M, x0, rhs, tol, maxit):
    N = x0.shape[0]
    x = x0.copy()
    nb = 4 if fn is orc.conjugategradient_merged else 3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        it = fn(x, M, rhs, *[np.empty(N) for _ in range(nb)], tol=tol, max_iters=maxit)
    return x, it


@pytest.mark.parametrize("variant", VARIANTS)
def test_merged_cg_matches_dense_solve_and_oracle(pkg, variant):
    """conjugategradient! in the merged-reduction form: first iterations to rounding against the oracle's restatement of the same
    arithmetic, the reference's stop rule and iteration count convention, the dense solution at the tolerance floor."""
    for name, A in _operators():
        m, n = A.shape
        rng = np.random.default_rng(m * 1000 + n)
        b, c = rng.standard_normal(m), rng.standard_normal(n)
        d = pkg.HipHSDE(A, b, c, [("Free", m)], [("Free", n)])
        d.set_cg_variant(variant)
        Q = orc.HSDEMatrixQ(A, b, c)
        M = orc.KKTMatrix(Q)
        rhs, x0 = rng.standard_normal(d.N), rng.standard_normal(d.N)
        Qd = Q.todense()
        Md = np.block([[np.eye(d.l), Qd.T], [Qd, -np.eye(d.l)]])
        xs = np.linalg.solve(Md, rhs)
        # the first iterations agree to rounding with the same recurrence on the host
        for k in (1, 2, 5):
            xk, it = d.cg_kkt(x0, rhs, 1e-300, k)
            xo, ito = _ocg(orc.conjugategradient_merged, M, x0, rhs, 1e-300, k)
            assert it == ito == k, (name, k, it)                  # max_iters cap (conjugategradients.jl:42)
            assert relerr(xk, xo) < 1e-12, (name, k)
        # tolerance floor: the dense solution, iteration count near both host recurrences
        tol = d.N * np.finfo(float).eps
        x, it = d.cg_kkt(x0, rhs, tol, 10000)
        assert relerr(x, xs) < 1e-12, name
        _, it_m = _ocg(orc.conjugategradient_merged, M, x0, rhs, tol, 10000)
        _, it_r = _ocg(orc.conjugategradient, M, x0, rhs, tol, 10000)
        # (hundreds of iterations of a chaotic recurrence on the indefinite system: counts agree to a few per cent)
        assert abs(it - it_m) <= 8 + it_m // 20 and abs(it - it_r) <= 8 + it_r // 20, (name, it, it_m, it_r)
        # loose tolerance: the stop rule ||r|| <= tol on a result as close to the solution as the reference recurrence's
        x, it = d.cg_kkt(x0, rhs, 1e-3, 10000)
        xr, it_r = _ocg(orc.conjugategradient, M, x0, rhs, 1e-3, 10000)
        assert abs(it - it_r) <= 6, (name, it, it_r)
        assert np.linalg.norm(Md @ x - rhs) <= 1e-3 * (1 + 1e-6), name
        assert np.linalg.norm(x - xs) <= 3 * max(np.linalg.norm(xr - xs), 1e-3), name
        # the same solve twice: bit-reproducible (fixed summation orders, no atomics)
        x2, it2 = d.cg_kkt(x0, rhs, 1e-3, 10000)
        assert it2 == it and np.array_equal(x, x2), name
        d.close()


@pytest.mark.parametrize("variant,geom", [(v, "1") for v in ["reference"] + VARIANTS] + [("reference", "2"), (VARIANTS[-1], "2")])
def test_cg_variants_on_window_panel_storage(pkg, variant, geom, monkeypatch):
    """The same solves with the operator forced into window-panel storage (FOS_WINDOWS=1: 2016-row panels, two workgroups per CU;
    2: the tall geometry of C5, 4032-row panels, one 1024-thread workgroup per CU; the sweep kernel carries its own closing prologue
    and tau stash): every recurrence against the dense solution and its host twin."""
    monkeypatch.setenv("FOS_WINDOWS", geom)
    rng = np.random.default_rng(15)
    shape = (700, 900) if geom == "1" else (5200, 4700)            # (tall: more than one panel per half, slices for every wavefront)
    A = sp.random(*shape, density=0.02 if geom == "1" else 0.004, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    m, n = A.shape
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    d = pkg.HipHSDE(A, b, c, [("Free", m)], [("Free", n)])
    assert d.operator_stats()["win_panels"] > 0
    d.set_cg_variant(variant)
    Q = orc.HSDEMatrixQ(A, b, c)
    M = orc.KKTMatrix(Q)
    rhs, x0 = rng.standard_normal(d.N), rng.standard_normal(d.N)
    ofn = orc.conjugategradient if variant == "reference" else orc.conjugategradient_merged
    for k in (1, 3):
        xk, it = d.cg_kkt(x0, rhs, 1e-300, k)
        xo, ito = _ocg(ofn, M, x0, rhs, 1e-300, k)
        assert it == ito == k and relerr(xk, xo) < 1e-12, (variant, k)
    x, it = d.cg_kkt(x0, rhs, 1e-6, 10000)
    y = np.empty(d.N)
    M.mul(y, x)
    assert np.linalg.norm(y - rhs) <= 1e-6 * (1 + 1e-3)
    _, it_o = _ocg(ofn, M, x0, rhs, 1e-6, 10000)
    assert abs(it - it_o) <= 6 + it_o // 20
    d.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_prox_affine_sequence_merged(pkg, variant):
    """prox!(y, S1::AffinePlusLinear, x) with the merged recurrence: call counter, tolerance schedule, warm start
    (affinepluslinear.jl:83-126); every result satisfies the reference's stopping rule and lies within 2 tol of the projection."""
    rng = np.random.default_rng(6)
    A = sp.random(60, 90, density=0.1, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    b, c = rng.standard_normal(60), rng.standard_normal(90)
    d = pkg.HipHSDE(A, b, c, [("Free", 60)], [("Free", 90)])
    d.set_cg_variant(variant)
    Q = orc.HSDEMatrixQ(A, b, c)
    S = orc.AffinePlusLinear(Q, np.zeros(d.l), np.zeros(d.l), 1, decreasing_accuracy=True)
    S.cg_variant = "merged"
    Qd = Q.todense()
    Md = np.block([[np.eye(d.l), Qd.T], [Qd, -np.eye(d.l)]])
    for call in range(1, 7):
        x = rng.standard_normal(d.N)
        tol = S.tolerance()
        y_ref = np.empty(d.N)
        S.prox(y_ref, x)
        y = d.prox_affine(x)
        assert d.prox_count() == S.i == call + 1
        rhs = np.concatenate([x[:d.l] + Qd.T @ x[d.l:], np.zeros(d.l)])
        exact = np.linalg.solve(Md, rhs)
        assert abs(d.cgiter() - S.getcgiter()) <= 4
        assert np.linalg.norm(y - y_ref) <= 2 * tol + 1e-12
        assert np.linalg.norm(Md @ y - rhs) <= tol * (1 + 1e-6) + 1e-13
        assert np.linalg.norm(y - exact) <= 2 * tol + 1e-12
    d.close()


def _oracle_run(prob, mk, iters, perturb, cg_variant):
    alg = mk(orc)
    mo = orc.Model(prob.A, prob.b, prob.c, _codes(prob.K1), _codes(prob.K2))
    alg.init(mo)
    alg.S1.cg_variant = cg_variant
    x = orc.hsde_initialvalue(mo)
    if perturb:
        x[np.abs(x) > 0] *= (1 + 2.220446049250313e-16)       # one ulp on tau and kappa
    st = orc.HSDEStatus(mo, 10 ** 9, 1e-5, 0, 0)
    out = []
    for i in range(1, iters + 1):
        st.i = i
        alg.step(x, i, st)
        out.append((x.copy(), alg.S1.getcgiter(), getattr(alg, "alpha12", None)))
    return out


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("algname", ["DR", "GAPA", "FISTA"])
def test_first_iterations_match_merged_oracle(pkg, algname, variant):
    """Iterate-level parity of whole outer iterations run on the merged recurrence, in the one-ulp chaos envelope of
    tests/test_gpu_parity.py::test_first_iterations_match_oracle (the oracle runs the same recurrence)."""
    prob = pkg.workloads.small_mixed()
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_cg_variant(variant)
    mk = {"DR": lambda M: M.DR(), "GAPA": lambda M: M.GAPA(0.8, 0.5), "FISTA": lambda M: M.FISTA()}[algname]
    iters = 25
    ref = _oracle_run(prob, mk, iters, False, "merged")
    per = _oracle_run(prob, mk, iters, True, "merged")
    d.set_alg(mk(pkg))
    d.set_iterate(None)
    envelope = kicks = 0.0
    for i in range(1, iters + 1):
        xo, cg_o, a12 = ref[i - 1]
        done, checked, _ = d.step(i, 1, 10 ** 9, 1e-5)
        assert done == 1 and not checked
        z = d.get_iterate()
        dev = np.linalg.norm(z - xo) / max(1.0, np.linalg.norm(xo))
        envelope = max(envelope, np.linalg.norm(per[i - 1][0] - xo) / max(1.0, np.linalg.norm(xo)))
        if d.cgiter() != cg_o:
            kicks += 4 * max(0.2 ** math.sqrt(i), d.l * 2.2e-16) / max(1.0, np.linalg.norm(xo))
        assert dev <= 50 * envelope + kicks + 1e-12, (i, dev, envelope, kicks)
        if i == 1:
            assert dev < 1e-7
        if dev < 1e-6 and envelope < 1e-6:
            assert d.cgiter() == cg_o, (i, d.cgiter(), cg_o)
    d.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("algname", ["DR", "GAPA"])
def test_whole_solves_on_merged_cg_match_reference_oracle(pkg, algname, variant):
    """Whole solves with the merged recurrence against the oracle running the REFERENCE recurrence: same status, iteration count
    within one check interval, same solution and residuals (the variants differ by rounding only)."""
    prob = pkg.workloads.small_mixed()
    mk = {"DR": lambda M, **o: M.DR(**o), "GAPA": lambda M, **o: M.GAPA(0.8, 0.5, **o)}[algname]
    opts = dict(eps=1e-6, verbose=0, max_iters=3000, checki=50)
    model = pkg.solve(prob, mk(pkg, cg_variant=variant, **opts))
    sol = orc.solve(orc.Model(prob.A, prob.b, prob.c, _codes(prob.K1), _codes(prob.K2)), mk(orc, **opts), out=[])
    assert model.status() == sol.status == "Optimal"
    assert abs(model.iterations - sol.iterations) <= 50
    ref_err = np.max(np.abs(sol.x - prob.x0))
    assert np.max(np.abs(model.getsolution() - prob.x0)) <= 3 * ref_err + 1e-9
    last, olast = model.status_obj.last, sol.status_obj.last
    for key in ("p", "d", "g"):
        assert abs(getattr(last, key) - olast[key]) < 1e-6
    # ... and on an operator with dual tiles (slot-spread rows finished inside the update kernel)
    prob = pkg.workloads.small_lp(seed=21, m=96, n=180)
    model = pkg.solve(prob, mk(pkg, cg_variant=variant, eps=1e-6, verbose=0, max_iters=600, checki=50))
    sol = orc.solve(orc.Model(prob.A, prob.b, prob.c, _codes(prob.K1), _codes(prob.K2)),
                    mk(orc, eps=1e-6, verbose=0, max_iters=600, checki=50), out=[])
    assert model.status() == sol.status and model.iterations == sol.iterations
    assert np.max(np.abs(model.getsolution() - sol.x)) <= 1e-4 * max(1.0, np.max(np.abs(sol.x)))


def test_late_workgroup_still_applies_the_last_x_update(pkg):
    """Reference recurrence, three launches: the kernel that closes iteration j (cg_pupdate_kernel) also carries x += alpha p_j.
    Workgroup 0 of that launch raises DevState.done when CG stops at j; a workgroup that starts AFTER that store must still
    apply the update to its slice (round-2 advisor finding: it used to return at the live flag, leaving a mix of x_j and
    x_{j-1}).  The test hook delays every workgroup but the first by 20 us; the result must be bit-identical to the undelayed
    run (all summation orders are fixed)."""
    rng = np.random.default_rng(9)
    A = sp.random(3000, 5000, density=0.002, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    m, n = A.shape
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    d = pkg.HipHSDE(A, b, c, [("Free", m)], [("Free", n)])
    d.set_cg_variant("reference")
    rhs, x0 = rng.standard_normal(d.N), rng.standard_normal(d.N)
    for tol in (1e-2, 1e-6):
        d.debug_set(pkg.lib.DEBUG_PUPDATE_DELAY, 0)
        x_ref, it_ref = d.cg_kkt(x0, rhs, tol, 10000)
        d.debug_set(pkg.lib.DEBUG_PUPDATE_DELAY, 2000)            # 20 us of the 100 MHz clock
        x_del, it_del = d.cg_kkt(x0, rhs, tol, 10000)
        d.debug_set(pkg.lib.DEBUG_PUPDATE_DELAY, 0)
        assert it_del == it_ref and it_ref > 3
        assert np.array_equal(x_del, x_ref), float(np.max(np.abs(x_del - x_ref)))
        # the stopped iterate satisfies the stop rule on the TRUE residual up to the recursion's drift
        y = np.empty(d.N)
        orc.KKTMatrix(orc.HSDEMatrixQ(A, b, c)).mul(y, x_del)
        assert np.linalg.norm(y - rhs) <= tol * (1 + 1e-3) + 1e-10
    d.close()


def test_cg_chain_bench_runs_for_every_variant(pkg):
    """fos_bench_cg_chain (measurement entry) on every variant, eager and as a replayed graph."""
    prob = pkg.workloads.small_lp(seed=21, m=96, n=180)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_iterate(None)
    for variant in ("reference", "fused_p", "merged_sweep", "merged_update"):
        d.set_cg_variant(variant)
        for graph in (0, 1):
            assert d.bench_cg_chain(5, 2, graph) > 0.0
    d.close()
