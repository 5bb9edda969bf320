"""
The reference's whole-solve known answers on the HIP path: test/testDRandGAPA.jl:9-53 with the reference's own data
(tests/golden/reference_test_inputs.npz: `Random.seed!(2); randn(40, 50); randn(40, 1)` as Julia < 1.5 and 1.5/1.6 drew
them -- see tests/test_reference_known_answers.py for how the fixture is pinned), the literal optima that file holds and
its thresholds.  No oracle in these assertions: the numbers on the right-hand sides are the reference's.
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_test_inputs.npz")
RTOL_APPROX = math.sqrt(np.finfo(float).eps)        # Julia's `≈`


@pytest.mark.parametrize("tag", ["pre15", "v15"])
def test_testDRandGAPA_literals_on_the_hip_path(pkg, tag):
    d = np.load(GOLD)
    prob = pkg.workloads.c1_readme_nnls(data=(d["A_" + tag], d["b_" + tag]))
    opt = float(d["opt_" + tag])
    n = prob.meta["n"]
    eps = 1e-8
    model = pkg.solve(prob, pkg.DR(eps=eps, verbose=0))                                  # :19-21
    assert model.status() == "Optimal"                                                  # :23
    assert abs(model.getobjval() - opt) <= RTOL_APPROX * max(abs(model.getobjval()), opt)   # :24
    assert abs(model.getsolution()[:n].min()) < 10 * eps                                # :25
    xsave = model.getsolution()[:n].copy()
    for alg in (pkg.GAPA(eps=1e-4, verbose=0), pkg.GAPA(direct=True, eps=1e-4, verbose=0)):    # :29-35, :38-44
        model = pkg.solve(prob, alg)
        assert model.status() == "Optimal"
        assert abs((model.getobjval() - opt) / opt) < 2e-3
        assert np.max(np.abs(model.getsolution()[:n] - xsave)) < 1e-3
    model = pkg.solve(prob, pkg.GAPA(0.5, 0.9, eps=1e-9, verbose=0))                     # :47-53
    assert model.status() == "Optimal"
    assert abs((model.getobjval() - opt) / opt) < 1e-8
    assert np.max(np.abs(model.getsolution()[:n] - xsave)) < 1e-7


def test_readme_gap_call_on_the_references_data(pkg):
    """README.md:21-26: `GAP(0.5, 2.0, 2.0, max_iters=2000)` on the same problem; the README shows no number, the
    literal of the test file bounds what 2000 iterations reach."""
    d = np.load(GOLD)
    prob = pkg.workloads.c1_readme_nnls(data=(d["A_pre15"], d["b_pre15"]))
    opt = float(d["opt_pre15"])
    model = pkg.solve(prob, pkg.GAP(0.5, 2.0, 2.0, max_iters=2000, verbose=0))
    assert model.status() in ("Optimal", "Indeterminate")
    assert abs((model.getobjval() - opt) / opt) < 1e-3


def test_testfeasibility_outcomes_on_the_hip_path(pkg):
    """test/testfeasibility.jl:15-44 on the reference's data (`Random.seed!(2); xsol1 = randn(100); A = randn(50, 100)`, Julia >= 1.5
    draw): the statuses of its seven solves -- three of them :Indeterminate -- and its accuracy bounds, through fos_feas_*."""
    d = np.load(GOLD)
    A = d["feas_A"]
    b = A @ d["feas_xsol"]                                                              # :7
    prob = pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), 100)           # :9-12
    sol, _ = pkg.solve_feasibility(prob, pkg.DR(eps=1e-8, verbose=0), checki=10)         # :15
    assert sol.status == "Optimal"                                                      # :17
    assert sol.x.min() > -1e-12                                                         # :18
    assert np.abs(A @ sol.x - b).max() < 1e-12                                          # :19
    sol, _ = pkg.solve_feasibility(prob, pkg.AP(eps=1e-8, verbose=0), checki=1)          # :21-23
    assert sol.status == "Indeterminate"
    sol, _ = pkg.solve_feasibility(prob, pkg.GAP(eps=1e-8, verbose=0))                   # :25-27
    assert sol.status == "Indeterminate"
    sol, _ = pkg.solve_feasibility(prob, pkg.FISTA(eps=1e-8, verbose=0))                 # :29-31
    assert sol.status == "Indeterminate"
    for alg in (pkg.GAPP(eps=1e-8, verbose=0, proji=50), pkg.GAPA(eps=1e-8, verbose=0),
                pkg.LineSearchWrapper(pkg.GAP(eps=1e-8, verbose=0))):                    # :33-36
        sol, _ = pkg.solve_feasibility(prob, alg, out=[])
        assert sol.status == "Optimal", type(alg).__name__                              # :41
        assert sol.x.min() > -1e-12                                                     # :42
        assert np.abs(A @ sol.x - b).max() < 1e-6                                       # :43


def _analytic_cases():
    from analytic_cases import cases
    return cases()


@pytest.mark.parametrize("cs", _analytic_cases(), ids=[c["name"] for c in _analytic_cases()])
def test_closed_form_optima_on_the_hip_path(pkg, cs):
    """Textbook conic programs with closed-form optima (tests/analytic_cases.py): second-order, ROTATED second-order, primal and dual
    EXPONENTIAL cones and svec'd PSD, as row cones and as variable cones, through fos_create / fos_iterate -- the cone conventions of
    `conemap` (src/cones.jl:4-14) pinned at whole-solve level without the oracle."""
    prob = pkg.workloads.ConicProblem(cs["name"], cs["A"], cs["b"], cs["c"], cs["K1"], cs["K2"])
    model = pkg.solve(prob, pkg.DR(eps=1e-8, verbose=0, max_iters=20000))
    assert model.status() == "Optimal"
    assert abs(model.getobjval() - cs["opt"]) < 1e-8 * max(1.0, abs(cs["opt"]))
    assert np.abs(model.getsolution() - cs["x"]).max() < 1e-7
    for w in cs["wrong"]:
        assert abs(model.getobjval() - w) > 1e-3
