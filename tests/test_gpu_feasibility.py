"""
GPU: the Feasibility form on the device (fos_feas_* through the Python mirror; src/problemforms/Feasibility/*.jl) against the
oracle's restatement: the two set projections, the first iterations of every algorithm, whole solves (status, iteration count,
solution), the assertions of the reference's own test (test/testfeasibility.jl), the printed table, and the error paths.
"""
import numpy as np
import pytest

from feasibility_cases import ALGS, GAPP, MIXED_CONES, affine_box_instance, cone_instance

pytestmark = pytest.mark.gpu


def _problems(pkg, orc, **kw):
    A, b = affine_box_instance(**kw)
    n = A.shape[1]
    return A, b, pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), n), orc.Feasibility(orc.IndAffine(A, b), orc.IndBox(0.0, np.inf), n)


@pytest.mark.parametrize("m,n", [(50, 100), (80, 100), (3, 7), (120, 257)])
def test_set_projections_match_oracle(pkg, oracle, m, n):
    orc = oracle
    A, b, hp, op = _problems(pkg, orc, seed=m + n, m=m, n=n)
    d = pkg.HipFeasibility(hp)
    info = d.info()
    assert info["ns_iters"][0] > 0 and info["ns_resid"][0] <= 1e-10 and info["ns_iters"][1] == 0
    rng = np.random.default_rng(1)
    for scale in (1.0, 1e3):
        x = scale * rng.standard_normal(n)
        y = np.empty(n)
        op.S1.prox(y, x)
        yd = d.prox(1, x)
        assert np.abs(yd - y).max() <= 1e-11 * max(1.0, np.abs(x).max())
        assert np.abs(A @ yd - b).max() <= 1e-10 * max(1.0, np.abs(x).max())
        op.S2.prox(y, x)
        assert np.array_equal(d.prox(2, x), y)                 # elementwise: bit exact
    # a two-sided box; array bounds
    lo, hi = -np.abs(rng.standard_normal(n)), np.abs(rng.standard_normal(n))
    hi[::7] = np.inf
    d2 = pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(-0.5, 0.25), pkg.IndBox(lo, hi), n))
    x = 2 * rng.standard_normal(n)
    assert np.array_equal(d2.prox(1, x), np.clip(x, -0.5, 0.25)) and np.array_equal(d2.prox(2, x), np.minimum(np.maximum(x, lo), hi))
    d3 = pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(-np.inf, 1.0), pkg.IndBox(lo, 3.0), n))
    assert np.array_equal(d3.prox(1, x), np.minimum(x, 1.0)) and np.array_equal(d3.prox(2, x), np.minimum(np.maximum(x, lo), 3.0))


@pytest.mark.parametrize("algname", sorted(ALGS))
def test_first_iterations_match_oracle(pkg, oracle, algname):
    orc = oracle
    A, b, hp, op = _problems(pkg, orc, m=80, n=100)
    oalg = ALGS[algname](orc, verbose=0)
    omodel = orc.FeasibilityModel(op, oalg)
    ost = orc.FeasibilityStatus(omodel, 4, 1e-30, 0, 1)
    d = pkg.HipFeasibility(hp)
    d.set_alg(ALGS[algname](pkg))
    d.set_iterate(None)
    xo = np.zeros(op.n)
    for i in range(1, 41):
        ost.i = i
        oalg.step(xo, i, ost)
        done, status, err, checked = d.step(i, 1, 4, 1e-30)
        assert done == 1 and status == "Continue" and checked == (i % 4 == 0)
        z = d.get_iterate()
        assert np.abs(z - xo).max() <= 1e-11 * max(1.0, np.abs(xo).max()), (algname, i)
        if checked and i > 4:
            assert err == pytest.approx(ost.err, rel=1e-6, abs=1e-13)
        if checked and i == 4:
            assert np.isfinite(err) == np.isfinite(ost.err)    # prev refreshed at every iteration: only i = 1 would see the NaN start
    if algname == "GAPA":
        assert d.info()["alpha12"] == pytest.approx(oalg.alpha12, rel=1e-9)
    g, _, _ = d.getsol()
    assert np.abs(g - oalg.getsol(xo)).max() <= 1e-11 * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("algname,kw", [("DR", dict(checki=10)), ("AP", dict(checki=1)), ("GAP", {}), ("GAPA", {}), ("FISTA", {}), ("Dykstra", {})])
def test_whole_solves_match_oracle(pkg, oracle, algname, kw):
    """test/testfeasibility.jl:15-44 on the seeded instance: status and iteration count of the oracle, the same point, and the
    reference's own assertions wherever the oracle ends :Optimal."""
    orc = oracle
    A, b, hp, op = _problems(pkg, orc)
    osol, _ = orc.feasibility_solve(op, ALGS[algname](orc, eps=1e-8, verbose=0), **kw)
    sol, model = pkg.solve_feasibility(hp, ALGS[algname](pkg, eps=1e-8, verbose=0), **kw)
    assert sol.status == osol.status == model.solve_stat
    assert abs(sol.iterations - osol.iterations) <= max(kw.get("checki", 100), osol.iterations // 20)     # err crosses eps at a slightly different check
    assert np.abs(sol.x - osol.x).max() <= 1e-6
    if sol.status == "Optimal":
        assert sol.x.min() > -1e-12 or algname in ("AP", "FISTA", "Dykstra")
        assert np.abs(A @ sol.x - b).max() < (1e-12 if algname == "DR" else 1e-6)
    assert model.history["err"][-1][0] == sol.iterations


def test_max_iters_forced_check_and_table(pkg, oracle):
    orc = oracle
    A, b, hp, op = _problems(pkg, orc)
    lines = []
    sol, model = pkg.solve_feasibility(hp, pkg.AP(eps=1e-14), out=lines, max_iters=7, checki=5)
    osol, omodel = orc.feasibility_solve(op, orc.AP(eps=1e-14, verbose=0), max_iters=7, checki=5)
    assert sol.status == "Indeterminate" and sol.iterations == 7
    assert [i for i, _ in model.history["err"]] == [5, 7]
    for (i1, e1), (i2, e2) in zip(model.history["err"], omodel.history["err"]):
        assert i1 == i2 and e1 == pytest.approx(e2, rel=1e-8)
    assert lines[0].startswith("Time to initialize: ") and lines[1] == "-" * 22 and lines[2] == " Iter | res | time" and lines[3] == "-" * 22
    assert lines[4].startswith("     5|") and lines[4].endswith("s")
    # initx (solverwrapper.jl:12-16) and a second solve on the same model: a fresh init_algorithm! state
    x0 = np.full(hp.n, 0.5)
    sol2, _ = pkg.solve_feasibility(hp, pkg.DR(eps=1e-8, verbose=0), checki=10, initx=x0)
    assert sol2.status == "Optimal" and np.abs(A @ sol2.x - b).max() < 1e-12


def test_error_paths(pkg):
    A, b = affine_box_instance()
    n = A.shape[1]
    with pytest.raises(pkg.lib.FosError):                      # host callbacks cannot be sets of the device path
        pkg.HipFeasibility(pkg.Feasibility(object(), pkg.IndBox(0.0, 1.0), n))
    with pytest.raises(pkg.lib.FosError):                      # rank-deficient A: A A' is singular
        A2 = np.vstack([A[:5], A[:5]])
        pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A2, np.ones(10)), pkg.IndBox(0.0, 1.0), n))
    with pytest.raises(pkg.lib.FosError):                      # the dense projector is bounded (two boxes of that length are fine)
        pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(np.ones((1, 50000)), np.ones(1), sparse=False), pkg.IndBox(0.0, 1.0), 50000))
    wide = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(np.ones((1, 50000)), np.ones(1)), pkg.IndBox(0.0, 1.0), 50000))      # (by default a matrix that wide takes the sparse form)
    assert abs(wide.prox(1, np.zeros(50000)).sum() - 1.0) < 1e-12
    big = pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(0.0, 1.0), pkg.IndBox(0.5, 2.0), 3_000_001))
    big.set_alg(pkg.AP())
    big.set_iterate(np.full(3_000_001, -1.0))
    done, status, err, checked = big.step(1, 3, 1, 1e-12)
    assert status == "Optimal" and done == 2 and err == 0.0 and np.array_equal(big.get_iterate(), np.full(3_000_001, 0.5))
    with pytest.raises(pkg.lib.FosError):
        pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(1.0, 0.0), pkg.IndBox(0.0, 1.0), n))
    d = pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(0.0, 1.0), pkg.IndBox(0.0, 1.0), n))
    d.set_alg(pkg.FISTA())
    with pytest.raises(pkg.lib.FosError):                      # support_linesearch: GAP and GAPA only
        d._lib.fos_feas_set_linesearch.restype = int
        pkg.lib.check(d._lib.fos_feas_set_linesearch(d._h, 5))


@pytest.mark.parametrize("algname", ["DR", "GAP", "GAPA"])
def test_linesearch_wrapper_matches_oracle(pkg, oracle, algname):
    """LineSearchWrapper(alg; lsinterval) on the Feasibility form (test/testfeasibility.jl:36-44 runs LineSearchWrapper(GAP)):
    iterates through two searches, the 31 test residuals and the chosen step length against the oracle; then the whole solve."""
    orc = oracle
    A, b, hp, op = _problems(pkg, orc, m=80, n=100)
    ls = 5
    lines = []
    owrap = orc.LineSearchWrapper(ALGS[algname](orc, verbose=0), lsinterval=ls, out=lines)
    omodel = orc.FeasibilityModel(op, owrap)
    ost = orc.FeasibilityStatus(omodel, 10 ** 9, 1e-30, 0, 1)
    d = pkg.HipFeasibility(hp)
    d.set_alg(pkg.LineSearchWrapper(ALGS[algname](pkg), lsinterval=ls))
    d.set_iterate(None)
    xo = np.zeros(op.n)
    for i in range(1, 2 * ls + 3):
        ost.i = i
        owrap.step(xo, i, ost)
        d.step(i, 1, 10 ** 9, 1e-30)
        z = d.get_iterate()
        assert np.abs(z - xo).max() <= 1e-10 * max(1.0, np.abs(xo).max()), (algname, i)
        if i % ls == 0:
            it, normres, tests, abest = d.linesearch_log()
            oi, onormres, otests, oabest = owrap.log[-1]
            assert it == oi == i and abest == oabest
            assert normres == pytest.approx(onormres, rel=1e-9)
            assert np.allclose(tests, otests, rtol=1e-7, atol=1e-12)
    # whole solve: the reference's test expects :Optimal with |A x - b| < 1e-6
    out = []
    sol, model = pkg.solve_feasibility(hp, pkg.LineSearchWrapper(ALGS[algname](pkg, eps=1e-8), lsinterval=20), out=out, checki=10)
    osol, _ = orc.feasibility_solve(op, orc.LineSearchWrapper(ALGS[algname](orc, eps=1e-8, verbose=0), lsinterval=20, out=[]), checki=10)
    assert sol.status == osol.status == "Optimal"
    assert abs(sol.iterations - osol.iterations) <= 10
    assert sol.x.min() > -1e-9 and np.abs(A @ sol.x - b).max() < 1e-6
    assert any(l.startswith("test, ") for l in out) and sum(l.startswith("\u03b1: ") for l in out) % 32 == 0


def test_cone_product_set_matches_oracle(pkg, oracle):
    """ConeProduct (src/cones.jl:31-94) as a set of the Feasibility form: every cone type, PSD orders 2, 6 and 64, projected by the
    batched cone kernels of the HSDE path; cold and warm-started PSD projections."""
    orc = oracle
    A, b, K, n = cone_instance(orc)
    d = pkg.HipFeasibility(pkg.Feasibility(pkg.ConeProduct(MIXED_CONES), pkg.IndBox(-np.inf, np.inf), n))
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n)
    y = np.empty(n)
    for rep in range(3):                                       # rep > 0: warm-started from the previous basis, slowly moving input
        K.prox(y, x)
        yd = d.prox(1, x)
        assert np.abs(yd - y).max() <= 5e-12 * max(1.0, np.abs(x).max()), rep
        x = x + 1e-3 * rng.standard_normal(n)
    d.set_iterate(None)                                        # a new solve starts cold again
    K.prox(y, x)
    assert np.abs(d.prox(1, x) - y).max() <= 5e-12
    for bad in ([("SDP", 4), ("Free", n - 4)], [("NonNeg", n - 1)], [("ExpPrimal", 6), ("Free", n - 6)], [("NonNeg", n + 1)]):
        with pytest.raises(pkg.lib.FosError):
            pkg.HipFeasibility(pkg.Feasibility(pkg.ConeProduct(bad), pkg.IndBox(0.0, 1.0), n))


@pytest.mark.parametrize("algname", ["DR", "GAPA", "FISTA", "Dykstra"])
def test_affine_cone_feasibility_matches_oracle(pkg, oracle, algname):
    """find x in {A x = b} n K, K a product of all cone types (the conic feasibility problem the reference's form is made for):
    first iterations against the oracle, then the whole solve."""
    orc = oracle
    A, b, K, n = cone_instance(orc)
    hp = pkg.Feasibility(pkg.IndAffine(A, b), pkg.ConeProduct(MIXED_CONES), n)
    op = orc.Feasibility(orc.IndAffine(A, b), K, n)
    oalg = ALGS[algname](orc, verbose=0)
    omodel = orc.FeasibilityModel(op, oalg)
    ost = orc.FeasibilityStatus(omodel, 10 ** 9, 1e-30, 0, 1)
    d = pkg.HipFeasibility(hp)
    d.set_alg(ALGS[algname](pkg))
    d.set_iterate(None)
    xo = np.zeros(n)
    for i in range(1, 26):
        ost.i = i
        oalg.step(xo, i, ost)
        d.step(i, 1, 10 ** 9, 1e-30)
        assert np.abs(d.get_iterate() - xo).max() <= 1e-10 * max(1.0, np.abs(xo).max()), (algname, i)
    sol, model = pkg.solve_feasibility(hp, ALGS[algname](pkg, eps=1e-7, verbose=0, max_iters=4000), checki=10)
    osol, _ = orc.feasibility_solve(op, ALGS[algname](orc, eps=1e-7, verbose=0, max_iters=4000), checki=10)
    assert sol.status == osol.status
    assert abs(sol.iterations - osol.iterations) <= max(10, osol.iterations // 20)
    assert np.abs(sol.x - osol.x).max() <= 1e-5
    if sol.status == "Optimal":
        proj = np.empty(n)
        K.prox(proj, sol.x)
        assert np.abs(proj - sol.x).max() <= 1e-6 and np.abs(A @ sol.x - b).max() <= 1e-5      # in the cone product and on the affine set


def test_gapp_matches_oracle(pkg, oracle):
    """GAPP (solvers/gapproj.jl) on the Feasibility form: GAP steps and two projected searches against the oracle (iterates, the 21 test
    norms, the chosen step), then the whole solve of test/testfeasibility.jl:36-44."""
    orc = oracle
    A, b, hp, op = _problems(pkg, orc, m=80, n=100)
    oalg = GAPP(orc, iproj=4, out=[])
    omodel = orc.FeasibilityModel(op, oalg)
    ost = orc.FeasibilityStatus(omodel, 10 ** 9, 1e-30, 0, 1)
    d = pkg.HipFeasibility(hp)
    d.set_alg(GAPP(pkg, iproj=4))
    d.set_iterate(None)
    xo = np.zeros(op.n)
    for i in range(1, 11):
        ost.i = i
        oalg.step(xo, i, ost)
        d.step(i, 1, 10 ** 9, 1e-30)
        assert np.abs(d.get_iterate() - xo).max() <= 1e-10 * max(1.0, np.abs(xo).max()), i
        if i % 4 == 0:
            it, tests, abest = d.gapp_log()
            oi, otests, oabest = oalg.log[-1]
            assert it == oi == i and abest == oabest and np.allclose(tests, otests, rtol=1e-7, atol=1e-11)
    out = []
    sol, model = pkg.solve_feasibility(hp, pkg.GAPP(eps=1e-8, verbose=0), out=out)
    osol, _ = orc.feasibility_solve(op, orc.GAPP(eps=1e-8, verbose=0, out=[]))
    assert sol.status == osol.status == "Optimal" and abs(sol.iterations - osol.iterations) <= 100
    assert sol.x.min() > -1e-9 and np.abs(A @ sol.x - b).max() < 1e-6
    assert sum(l.startswith("normtest: ") for l in out) == 21 * (sol.iterations // 100)


class IndBallL2:
    """A ProximableFunction the device has no kernel for (ProximalOperators.IndBallL2): projection onto the Euclidean ball of radius r
    -- the protocol object of both the oracle and, through fos_feas_set_callback, the device path."""

    def __init__(self, r, center):
        self.r, self.c = float(r), np.array(center, dtype=np.float64)
        self.calls = 0

    def prox(self, y, x):
        self.calls += 1
        d = x - self.c
        nd = float(np.linalg.norm(d))
        y[:] = x if nd <= self.r else self.c + d * (self.r / nd)


@pytest.mark.parametrize("algname", ["DR", "GAPA", "FISTA", "Dykstra"])
def test_host_callback_set_matches_oracle(pkg, oracle, algname):
    """Feasibility.jl:2-6 takes ANY two ProximableFunctions: a set without a device kernel is evaluated by the caller (fos_feas_set_callback:
    the iterate crosses the host link per projection, the other set / relaxations / status stay on the device).  Same iterates as the
    oracle running the same object; the callback is really called; whole solve reaches the oracle's status and point."""
    orc = oracle
    A, b = affine_box_instance(m=30, n=100)
    n = A.shape[1]
    x_ls = np.linalg.lstsq(A, b, rcond=None)[0]                 # a point of the affine set; the ball around a point 2 away from it, radius 2.2,
    u = np.random.default_rng(9).standard_normal(n)             # meets the set without containing the start: the projection is active
    center = x_ls + 2.0 * u / np.linalg.norm(u)
    ball_d, ball_o = IndBallL2(2.2, center), IndBallL2(2.2, center)
    hp = pkg.Feasibility(pkg.IndAffine(A, b), ball_d, n)
    op = orc.Feasibility(orc.IndAffine(A, b), ball_o, n)
    oalg = ALGS[algname](orc, verbose=0)
    omodel = orc.FeasibilityModel(op, oalg)
    ost = orc.FeasibilityStatus(omodel, 4, 1e-30, 0, 1)
    d = pkg.HipFeasibility(hp)
    d.set_alg(ALGS[algname](pkg))
    d.set_iterate(None)
    xo = np.zeros(n)
    # (GAPA's step-length estimate is ill-conditioned against a curved set: a relative perturbation of 1e-15 after the first iteration
    #  moves the ORACLE's own alpha12 of the second by 1e-3 -- its iterates are compared for two iterations, its solve for the result)
    nsame = 2 if algname == "GAPA" else 30
    for i in range(1, nsame + 1):
        ost.i = i
        oalg.step(xo, i, ost)
        done, status, err, checked = d.step(i, 1, 4, 1e-30)
        assert done == 1                                       # (Dykstra's iterate becomes exactly stationary here: err = 0 is Optimal even at eps = 1e-30)
        assert np.abs(d.get_iterate() - xo).max() <= 1e-11 * max(1.0, np.abs(xo).max()), (algname, i)
    assert ball_d.calls == ball_o.calls > 0
    if algname in ("DR", "GAPA"):                              # the whole solve ends in the intersection
        done, status, err, checked = d.step(nsame + 1, 3000, 10, 1e-9)
        assert status == "Optimal", (status, err)
        g, _, _ = d.getsol()
        assert np.abs(A @ g - b).max() <= 1e-6 and np.linalg.norm(g - center) <= 2.2 * (1 + 1e-6)
    x = np.random.default_rng(3).standard_normal(n) * 50
    y = np.empty(n)
    ball_o.prox(y, x)
    assert np.array_equal(d.prox(2, x), y)                     # the callback's own arithmetic, untouched by the two copies


def test_host_callback_on_both_sets_and_errors(pkg, oracle):
    """Both sets as callbacks (the oracle's IndBox / IndAffine objects handed to the device path): bit-identical to the oracle for the
    elementwise set; an exception inside a callback stops the step with FosError (cause = the exception) instead of unwinding through C."""
    orc = oracle
    A, b = affine_box_instance(m=20, n=60)
    n = A.shape[1]
    d = pkg.HipFeasibility(pkg.Feasibility(orc.IndAffine(A, b), orc.IndBox(0.0, np.inf), n))
    dd = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), n))
    x = np.random.default_rng(0).standard_normal(n)
    assert np.array_equal(d.prox(2, x), dd.prox(2, x))
    assert np.abs(d.prox(1, x) - dd.prox(1, x)).max() <= 1e-11
    for h in (d, dd):
        h.set_alg(pkg.DR())
        h.set_iterate(None)
        h.step(1, 50, 10, 1e-30)
    assert np.abs(d.get_iterate() - dd.get_iterate()).max() <= 1e-10

    class Broken:
        def prox(self, y, x):
            raise ZeroDivisionError("boom")
    e = pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(0.0, 1.0), Broken(), n))
    e.set_alg(pkg.DR())
    e.set_iterate(None)
    with pytest.raises(pkg.lib.FosError) as ei:
        e.step(1, 1, 1, 1e-6)
    assert isinstance(ei.value.__cause__, ZeroDivisionError) and "callback" in str(ei.value)
    with pytest.raises(pkg.lib.FosError):
        pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(0.0, 1.0), object(), n))


@pytest.mark.parametrize("algname", ["DR", "FISTA", "Dykstra"])
def test_longstep_on_the_feasibility_form_matches_oracle(pkg, oracle, algname):
    """LongstepWrapper (wrappers/longstep.jl) around an algorithm on the Feasibility form (fos_feas_set_longstep): saving iterations,
    the projection onto the saved planes and the steps around them against the oracle's restatement."""
    orc = oracle
    A, b, hp, op = _problems(pkg, orc, m=60, n=100)
    oalg = orc.LongstepWrapper(ALGS[algname](orc, verbose=0), longinterval=8, nsave=3)
    omodel = orc.FeasibilityModel(op, oalg)
    ost = orc.FeasibilityStatus(omodel, 4, 1e-30, 0, 1)
    d = pkg.HipFeasibility(hp)
    d.set_alg(pkg.LongstepWrapper(ALGS[algname](pkg), longinterval=8, nsave=3))
    d.set_iterate(None)
    xo = np.zeros(op.n)
    for i in range(1, 27):
        ost.i = i
        oalg.step(xo, i, ost)
        d.step(i, 1, 4, 1e-30)
        z = d.get_iterate()
        assert np.abs(z - xo).max() <= 1e-8 * max(1.0, np.abs(xo).max()), (algname, i)
        if i % 8 == 0:
            log = d.longstep_log()
            assert log["iteration"] == i == oalg.log[-1][0] and log["rows"] == 8
    assert len(oalg.log) == 3
