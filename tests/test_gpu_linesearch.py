"""
GPU test of the LineSearchWrapper (src/wrappers/linesearch.jl) on the device -- fos_set_linesearch / fos_linesearch_log through the
Python mirror -- against the oracle's restatement on the same problem: iterates before, at and after a search, the 31 test
residuals and the chosen step length, the printed lines, the CG tolerance counter, and whole solves.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _omodel(orc, prob):
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    return orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))


@pytest.mark.parametrize("algname", ["DR", "GAP", "GAPA"])
def test_linesearch_iterates_match_oracle(pkg, oracle, algname):
    orc = oracle
    prob = pkg.workloads.small_mixed()
    mk = {"DR": lambda M: M.DR(), "GAP": lambda M: M.GAP(0.8, 1.5, 1.6), "GAPA": lambda M: M.GAPA(0.8, 0.5)}[algname]
    ls = 5
    # oracle
    lines = []
    owrap = orc.LineSearchWrapper(mk(orc), lsinterval=ls, out=lines)
    mo = _omodel(orc, prob)
    owrap.init(mo)
    xo = orc.hsde_initialvalue(mo)
    # device
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.LineSearchWrapper(mk(pkg), lsinterval=ls))
    d.set_iterate(None)
    st = orc.HSDEStatus(mo, 10 ** 9, 1e-9, 0, 1, S1=owrap.S1)
    for i in range(1, 2 * ls + 2):                              # two searches (i = 5, 10) and the steps around them
        st.i = i
        owrap.step(xo, i, st)
        d.step(i, 1, 10 ** 9, 1e-9)
        z = d.get_iterate()
        scale = max(1.0, np.linalg.norm(xo))
        # (the first iterations solve S1 to the loose tolerance 0.2^sqrt(i) of affinepluslinear.jl:108-112: the inexact
        #  solutions depend on rounding at the 1e-5 level, on the device and in the oracle alike)
        assert np.linalg.norm(z - xo) <= 1e-4 * scale, (algname, i)
        assert d.prox_count() == owrap.S1.i, (algname, i)       # S.i of affinepluslinear.jl:114: +1 per step, +32 per search
        if i % ls == 0:
            it, normres, tests, abest = d.linesearch_log()
            oi, onormres, otests, oabest = owrap.log[-1]
            assert it == oi == i
            assert normres == pytest.approx(onormres, rel=1e-4)
            assert np.allclose(tests, otests, rtol=2e-3, atol=1e-9), (algname, i)
            # same step length unless two trial residuals tie to within the CG tolerance
            k, ko = int(np.argmin(tests)), int(np.argmin(otests))
            assert abest == oabest or abs(otests[k] - otests[ko]) <= 2e-3 * abs(otests[ko]), (algname, i, abest, oabest)
    d.close()
    assert len(lines) == 2 * 33


def test_linesearch_solve_prints_and_converges(pkg, oracle):
    """Whole solve through FOSMathProgModel: the search output is printed in the reference's form after every lsinterval-th
    iteration, the solve still ends Optimal at the oracle's objective; algorithms without support_linesearch are refused."""
    orc = oracle
    prob = pkg.workloads.c1_readme_nnls(seed=2)
    out = []
    model = pkg.FOSMathProgModel(pkg.LineSearchWrapper(pkg.DR(eps=1e-6, verbose=1, checki=10), lsinterval=50))
    model.out = out
    model.loadproblem(prob.c, prob.A, prob.b, prob.K1, prob.K2)
    model.optimize()
    assert model.status() == "Optimal"
    sol = orc.solve(_omodel(orc, prob), orc.LineSearchWrapper(orc.DR(eps=1e-6, verbose=0, checki=10), lsinterval=50, out=[]))
    assert sol.status == "Optimal" and model.getobjval() == pytest.approx(sol.obj_val, rel=1e-4)
    tests = [s for s in out if s.startswith("test, ")]
    alphas = [s for s in out if s.startswith("α: ")]
    assert len(tests) == model.iterations // 50 and len(alphas) == 32 * len(tests)
    assert alphas[0].startswith("α: 0.18000000000000002, ")
    with pytest.raises(ValueError):
        pkg.LineSearchWrapper(pkg.FISTA())
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.FISTA())
    with pytest.raises(pkg.lib.FosError):
        d.set_linesearch(10)
    d.close()


def test_linesearch_is_refused_on_sharded_handles(pkg):
    """normres / normdiff of the search (linesearch.jl:50,62) are GLOBAL norms; the device search adds one rank's partial sums only,
    so a sharded handle must refuse the wrapper -- in both orders -- instead of letting the ranks pick different step lengths
    (round-2 advisor finding)."""
    prob = pkg.workloads.c1_readme_nnls(seed=2)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.DR())
    d.comm_init_host(1, 0, lambda a: None)                     # a one-rank "sharded" handle
    with pytest.raises(pkg.lib.FosError) as ei:
        d.set_linesearch(10)
    assert ei.value.code == -4                                 # FOS_EUNSUPPORTED
    d.close()
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.DR())
    d.set_linesearch(10)
    with pytest.raises(pkg.lib.FosError) as ei:
        d.comm_init_host(1, 0, lambda a: None)
    assert ei.value.code == -4
    d.set_linesearch(0)
    d.comm_init_host(1, 0, lambda a: None)                     # fine once the wrapper is off
    d.close()
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2, row_sharded=True)
    d.set_alg(pkg.DR())
    with pytest.raises(pkg.lib.FosError):
        d.set_linesearch(10)
    d.close()


@pytest.mark.parametrize("direct", [False, True])
def test_gapp_on_the_hsde_path_matches_oracle(pkg, oracle, direct):
    """GAPP ("projected GAP", src/solvers/gapproj.jl -- the last row of the reference's solver table) on the device: GAP steps and two
    projected searches against the oracle's restatement -- iterates, the 21 test norms, the chosen step length, the CG tolerance
    counter (S1 is applied twice in a search iteration) -- with the CG projection and with direct=true (GAPP's default)."""
    orc = oracle
    prob = pkg.workloads.small_mixed()
    ip = 4
    lines = []
    oalg = orc.GAPP(0.8, 1.5, 1.6, iproj=ip, direct=direct, out=lines)
    mo = _omodel(orc, prob)
    oalg.init(mo)
    xo = orc.hsde_initialvalue(mo)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    if direct:
        d.enable_direct(prob.A)
    d.set_alg(pkg.GAPP(0.8, 1.5, 1.6, iproj=ip, direct=direct))
    d.set_iterate(None)
    st = orc.HSDEStatus(mo, 10 ** 9, 1e-9, 0, 1, S1=oalg.S1)
    for i in range(1, 2 * ip + 2):
        st.i = i
        oalg.step(xo, i, st)
        d.step(i, 1, 10 ** 9, 1e-9)
        z = d.get_iterate()
        scale = max(1.0, np.linalg.norm(xo))
        tol = 1e-9 if direct else 5e-2                      # (CG: the first iterations solve S1 to the loose tolerance 0.2^sqrt(i))
        assert np.linalg.norm(z - xo) <= tol * scale, (i, np.linalg.norm(z - xo))
        if i % ip == 0 and direct:
            it, tests, abest = d.gapp_log()
            oi, otests, oabest = oalg.log[-1]
            assert it == oi == i and abest == oabest and np.allclose(tests, otests, rtol=1e-6, atol=1e-10)
    if not direct:
        _, pi, _ = d.get_affine_state()
        assert pi == oalg.S1.i                               # two S1 applications per search iteration


def test_gapp_whole_solve_and_refusals(pkg, oracle):
    orc = oracle
    prob = pkg.workloads.small_mixed()
    out = []
    # (direct = true, GAPP's default: both sides project exactly, so the whole trajectory is comparable; 450 iterations = four searches)
    model = pkg.FOSMathProgModel(pkg.GAPP(eps=1e-7, verbose=0, max_iters=450, checki=50, iproj=100))
    model.out = out
    model.loadproblem(prob.c, prob.A, prob.b, prob.K1, prob.K2)
    model.optimize()
    osol = orc.solve(_omodel(orc, prob), orc.GAPP(eps=1e-7, verbose=0, max_iters=450, checki=50, iproj=100, out=[]))
    assert model.status() == osol.status
    assert model.iterations == osol.iterations
    assert np.max(np.abs(model.getsolution() - osol.x)) <= 1e-6 * max(1.0, np.max(np.abs(osol.x)))
    assert sum(l.startswith("normtest: ") for l in out) == 21 * (model.iterations // 100)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.FISTA())
    with pytest.raises(pkg.lib.FosError):                     # GAPP is GAP with a search
        pkg.lib.check(d._lib.fos_set_gapp(d._h, 10))
