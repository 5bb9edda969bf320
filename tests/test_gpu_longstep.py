"""
GPU test of the LongstepWrapper (src/wrappers/longstep.jl, saveplanes.jl) on the device -- fos_set_longstep / fos_longstep_log through the
Python mirror -- against the oracle's restatement on the same problem: the iterates of the saving iterations, of the projection onto the
saved planes and of the steps around them, for the four algorithms with support_longstep; the error paths.
(direct = true, as in the reference's own use of the wrapper, test/testspecific.jl:8,23: with the exact S1 projection device and oracle
agree to rounding; with CG the first iterations are solved to 0.2^sqrt(i) only.)
"""
import numpy as np
import pytest

from feasibility_cases import affine_box_instance

pytestmark = pytest.mark.gpu

ALGS = {"DR": lambda M: M.DR(direct=True), "GAP": lambda M: M.GAP(0.8, 1.5, 1.6, direct=True), "GAPA": lambda M: M.GAPA(0.8, 0.5, direct=True),
        "FISTA": lambda M: M.FISTA(direct=True), "Dykstra": lambda M: M.Dykstra(direct=True)}


def _omodel(orc, prob):
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    return orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))


@pytest.mark.parametrize("algname", sorted(ALGS))
@pytest.mark.parametrize("longinterval,nsave", [(7, 2), (9, 4)])
def test_longstep_iterates_match_oracle(pkg, oracle, algname, longinterval, nsave):
    orc = oracle
    prob = pkg.workloads.small_mixed()
    owrap = orc.LongstepWrapper(ALGS[algname](orc), longinterval=longinterval, nsave=nsave)
    mo = _omodel(orc, prob)
    owrap.init(mo)
    xo = orc.hsde_initialvalue(mo)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.enable_direct(prob.A)
    d.set_alg(pkg.LongstepWrapper(ALGS[algname](pkg), longinterval=longinterval, nsave=nsave))
    d.set_iterate(None)
    st = orc.HSDEStatus(mo, 10 ** 9, 1e-9, 0, 1, S1=owrap.alg.S1)
    worst = 0.0
    for i in range(1, 3 * longinterval + 2):                      # three projections and the steps around them
        st.i = i
        owrap.step(xo, i, st)
        d.step(i, 1, 10 ** 9, 1e-9)
        z = d.get_iterate()
        err = np.linalg.norm(z - xo) / max(1.0, np.linalg.norm(xo))
        worst = max(worst, err)
        # GAPA's step-length estimate amplifies rounding (tests/test_gpu_feasibility.py): its iterates are followed more loosely
        assert err <= (1e-6 if algname == "GAPA" else 2e-8), (algname, i, err)       # (projections onto nearly parallel planes amplify rounding: 1e-9 after three of them)
        if i % longinterval == 0:
            log = d.longstep_log()
            assert log["iteration"] == i == owrap.log[-1][0]
            assert log["rows"] == 2 * (nsave + 1) and log["violation"] <= 1e-9 * max(1.0, log["step"])
    assert len(owrap.log) == 3
    d.close()


def test_longstep_solve_and_error_paths(pkg, oracle):
    """Whole solve through FOSMathProgModel with the wrapper (CG projections): same status as the oracle's wrapped solve and an objective
    within the solve's own tolerance; what the reference refuses is refused; so is what this build does not support."""
    orc = oracle
    prob = pkg.workloads.c1_readme_nnls(seed=2)
    opts = dict(eps=1e-6, verbose=0, checki=10, max_iters=600, direct=True)
    model = pkg.FOSMathProgModel(pkg.LongstepWrapper(pkg.DR(**opts), longinterval=50, nsave=3))
    model.loadproblem(prob.c, prob.A, prob.b, prob.K1, prob.K2)
    model.optimize()
    ow = orc.LongstepWrapper(orc.DR(**opts), longinterval=50, nsave=3)
    sol = orc.solve(_omodel(orc, prob), ow)
    assert model.status() == sol.status
    assert model.getobjval() == pytest.approx(sol.obj_val, rel=1e-5)
    with pytest.raises(ValueError):
        pkg.LongstepWrapper(pkg.GAPP())                           # support_longstep(::GAPP) = false   gapproj.jl:83
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.DR())
    with pytest.raises(pkg.lib.FosError):
        d.set_longstep(100, 16)                                   # 2 (nsave + 1) > 32 planes
    with pytest.raises(pkg.lib.FosError):
        d.set_longstep(3, 5)                                      # longinterval < nsave + 1: a plane would be read before it is written
    d.set_longstep(20, 2)
    d.set_longstep(0, 0)                                          # off again
    d.set_linesearch(10)
    with pytest.raises(pkg.lib.FosError):
        d.set_longstep(20, 2)                                     # not around a LineSearchWrapper
    d.close()


@pytest.mark.parametrize("algname", ["AP", "GAP"])
def test_default_nsave_projection_is_extended_precision(pkg, oracle, algname):
    """The reference's default nsave = 10 saves 22 planes whose normals are nearly dependent (here sigma_max / sigma_min of the saved rows
    reaches 1e10 .. 1e11), which is why it solves the projection in BigFloat (saveplanes.jl:24).  The device forms the Gram products in
    double-double, solves the small dual in 113-bit arithmetic and applies the multipliers in double-double; the oracle solves it in 60 digits.
    On the Feasibility form (no CG noise in the iterates) the two agree through two projections to 1e-7 of the iterate (measured 1e-9 .. 2e-8) -- a float64 solve
    through G = P P' loses cond^2 eps = everything here (it was off by 1e-4 at cond 1e8 on the README NNLS)."""
    orc = oracle
    mk = {"AP": lambda M: M.AP(), "GAP": lambda M: M.GAP(0.8, 1.5, 1.6)}[algname]
    A, b = affine_box_instance(m=50, n=100)
    hp = pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), 100)
    op = orc.Feasibility(orc.IndAffine(A, b), orc.IndBox(0.0, np.inf), 100)
    owrap = orc.LongstepWrapper(mk(orc), longinterval=25, nsave=10)
    omodel = orc.FeasibilityModel(op, owrap)
    ost = orc.FeasibilityStatus(omodel, 10 ** 9, 1e-30, 0, 1)
    conds = []
    orig = orc.project_onto_planes

    def hook(A_, b_, C_, d_, x_, tol=1e-12):
        sv = np.linalg.svd(np.vstack([A_, C_]), compute_uv=False)
        nz = sv[sv > 1e-13 * sv[0]]                        # (GAP's planes of an inactive bound are exactly zero rows: rank deficient on top)
        conds.append(float(nz[0] / nz[-1]))
        return orig(A_, b_, C_, d_, x_, tol)
    d = pkg.HipFeasibility(hp)
    try:
        orc.project_onto_planes = hook
        d.set_alg(pkg.LongstepWrapper(mk(pkg), longinterval=25, nsave=10))
        d.set_iterate(None)
        xo = np.zeros(100)
        worst = 0.0
        for i in range(1, 52):
            ost.i = i
            owrap.step(xo, i, ost)
            d.step(i, 1, 10 ** 9, 1e-30)
            worst = max(worst, np.abs(d.get_iterate() - xo).max() / max(1.0, np.abs(xo).max()))
            if i % 25 == 0:
                log = d.longstep_log()
                assert log["rows"] == 22 and log["iteration"] == i and log["step"] > 0
    finally:
        orc.project_onto_planes = orig
        d.close()
    assert len(conds) == 2 and max(conds) > 1e8, conds           # float64 through the Gram matrix would be lost here
    assert worst <= 1e-7, (worst, conds)                         # measured 1.4e-9 (AP), 1.7e-8 (GAP: zero rows, pseudo-inverse)


def test_longstep_without_a_kkt_point_leaves_the_iterate(pkg, monkeypatch):
    """The small dual of the plane projection found no KKT point within its budget (forced: a budget of zero candidate supports --
    in production: inconsistent or dependent planes): the step must not apply a non-projection -- the iterate stays what the wrapped
    algorithm's step made it, the log says `failed`, the sweep of an nsave change does not leak, and excluded wrapper pairs are refused both ways."""
    prob = pkg.workloads.small_mixed()
    runs = {}
    for budget in ("0", None):
        if budget is None:
            monkeypatch.delenv("FOS_LONG_MAX_SUPPORTS", raising=False)
        else:
            monkeypatch.setenv("FOS_LONG_MAX_SUPPORTS", budget)
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        d.enable_direct(prob.A)
        d.set_alg(pkg.LongstepWrapper(pkg.DR(direct=True), longinterval=7, nsave=2))
        d.set_iterate(None)
        d.step(1, 7, 10 ** 9, 1e-9)
        runs[budget] = (d.get_iterate(), d.longstep_log())
        d.close()
    # the same seven DR steps without any wrapper
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.enable_direct(prob.A)
    d.set_alg(pkg.DR(direct=True))
    d.set_iterate(None)
    d.step(1, 7, 10 ** 9, 1e-9)
    plain = d.get_iterate()
    assert runs["0"][1]["failed"] and runs["0"][1]["tried"] == 0 and runs["0"][1]["step"] == 0.0 and runs["0"][1]["given_up"] == 1
    assert runs[None][1]["given_up"] == 0
    assert np.linalg.norm(runs["0"][0] - plain) <= 1e-12 * np.linalg.norm(plain)           # unfused vs fused step: rounding only
    assert not runs[None][1]["failed"] and runs[None][1]["step"] > 0
    assert np.linalg.norm(runs[None][0] - plain) > 1e-6 * np.linalg.norm(plain)           # (the projection does move the iterate)
    # nsave swept on one handle, and the exclusions in both directions
    for nsave in (1, 3, 2, 5):
        d.set_longstep(20, nsave)
    with pytest.raises(pkg.lib.FosError):
        d.set_linesearch(10)
    d.set_longstep(0, 0)
    d.set_alg(pkg.GAP(0.8, 1.5, 1.6, direct=True))
    d.set_longstep(20, 2)
    with pytest.raises(pkg.lib.FosError):
        pkg.lib.check(d._lib.fos_set_gapp(d._h, 10))
    d.close()
