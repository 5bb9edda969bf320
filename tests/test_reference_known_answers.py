"""
Reference-held known answers for WHOLE SOLVES: test/testDRandGAPA.jl (and README.md:21-26) hold the literal optimum of
`Random.seed!(2); A = randn(40, 50); b = randn(40, 1); minimize(sumsquares(A x - b), x >= 0)` for Julia < 1.5
(12.38418747141913) and Julia >= 1.5 (10.945929126466417).  oracle/julia_random.py restates Julia's generator (dSFMT-19937 +
the stdlib ziggurat), which regenerates those inputs (tests/golden/reference_test_inputs.npz, script beside it); the two
literals pin the restatement (first test) and then the oracle's solves with the reference's own assertions and
thresholds, quoted line by line.  test/testfeasibility.jl's data comes from the same stream: its seven solves' statuses
(three of them :Indeterminate) and accuracies are data-dependent known answers too.  CPU only.
"""
import math
import os

import numpy as np
import pytest
import scipy.optimize

import fos_oracle as orc
import julia_random as jr

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_test_inputs.npz")
RTOL_APPROX = math.sqrt(np.finfo(float).eps)        # Julia's `≈` on scalars


def exact_nnls_optimum(A, b):
    x, _ = scipy.optimize.nnls(A, b)
    S = x > 0
    xs = np.linalg.lstsq(A[:, S], b, rcond=None)[0]    # the active set's least squares, to rounding
    assert xs.min() > 0
    return float(np.sum((A[:, S] @ xs - b) ** 2))


@pytest.mark.parametrize("gen,tag", [("pre1.5", "pre15"), ("1.5", "v15")])
def test_julia_stream_reproduces_the_references_literal_optima(gen, tag):
    """The restated generator gives the committed fixture bit for bit, and the exact optimum of the regenerated problem is
    the literal of test/testDRandGAPA.jl:12/:15 -- to 1e-13 for the pre-1.5 draw (an exact value), within `≈` for the
    1.5 draw (that literal was taken from a solve at eps = 1e-8).  One wrong draw moves it in the second digit."""
    d = np.load(GOLD)
    A, b, opt = jr.readme_nnls_data(gen)
    assert np.array_equal(A, d["A_" + tag]) and np.array_equal(b, d["b_" + tag]) and opt == float(d["opt_" + tag])
    e = exact_nnls_optimum(A, b)
    assert abs(e - opt) <= (1e-13 if tag == "pre15" else RTOL_APPROX) * opt
    # first draw of `Random.seed!(2); randn()` on Julia 0.7 .. 1.6
    assert A[0, 0] == pytest.approx(0.7396206598864331, rel=0, abs=2e-15)
    # perturbation: the optimum is nowhere near if a single entry is redrawn
    b2 = b.copy()
    b2[3] = -b2[3]
    assert abs(exact_nnls_optimum(A, b2) - opt) > 1e-3 * opt
    assert abs(exact_nnls_optimum(np.roll(A.reshape(-1, order="F"), 1).reshape(A.shape, order="F"), b) - opt) > 1e-3 * opt


def test_julia_stream_building_blocks():
    """Structure of the restated generator: the cache refill and the straight fill draw the same sequence (dSFMT's
    fill_array is the recursion itself), every word is a double in [1, 2), and the ziggurat tables are monotone with
    the published end points (r = 3.6541528853610088)."""
    a, b = jr.JuliaMersenneTwister(7), jr.JuliaMersenneTwister(7)
    seq = [a.raw() for _ in range(1002)]
    assert seq == b.fill_raw(1002)
    assert all((w >> 52) == 0x3FF for w in seq)
    assert [a.raw() for _ in range(1002)] == b.fill_raw(1002)          # state carried across refills identically
    u = np.array([jr.JuliaMersenneTwister(5).rand() for _ in range(1)] + [a.rand() for _ in range(5000)])
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.02
    assert jr._WI[255] * jr._NMANT == pytest.approx(jr._NOR_R, rel=1e-15) and jr._FI[0] == 1.0 and jr._KI[1] == 0
    assert all(x < y for x, y in zip(jr._WI[1:], jr._WI[2:])) and all(x > y for x, y in zip(jr._FI, jr._FI[1:]))
    z = jr.randn_scalar_fill(jr.JuliaMersenneTwister(11), 20000)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03 and abs(np.mean(z ** 4) - 3) < 0.2
    z2 = jr.randn_array_fill(jr.JuliaMersenneTwister(11), 20000)
    assert abs(z2.mean()) < 0.03 and abs(z2.std() - 1) < 0.03
    assert z[0] == z2[0]                                                # same first word, consumed differently later


def reference_case(pkg, tag):
    d = np.load(GOLD)
    prob = pkg.workloads.c1_readme_nnls(data=(d["A_" + tag], d["b_" + tag]))
    model = orc.Model(prob.A, prob.b, prob.c, [(orc.CONE_CODES[k], l) for k, l in prob.K1],
                      [(orc.CONE_CODES[k], l) for k, l in prob.K2])
    return prob, model, float(d["opt_" + tag])


@pytest.mark.parametrize("tag", ["pre15", "v15"])
def test_oracle_meets_testDRandGAPA_on_the_references_own_data(pkg, tag):
    """test/testDRandGAPA.jl:9-49 with the reference's data, literals and thresholds (the conic form handed over is ours --
    Convex.jl is not here -- the optimum and minimiser do not depend on it)."""
    prob, model, opt = reference_case(pkg, tag)
    n = prob.meta["n"]
    eps = 1e-8
    sol = orc.solve(model, orc.DR(eps=eps, verbose=0))                                   # :19-21
    assert sol.status == "Optimal"                                                      # :23
    assert abs(sol.obj_val - opt) <= RTOL_APPROX * max(abs(sol.obj_val), opt)           # :24  optval ≈ opt
    assert abs(sol.x[:n].min()) < 10 * eps                                              # :25
    xsave = sol.x[:n].copy()
    sol = orc.solve(model, orc.GAPA(direct=True, eps=1e-4, verbose=0))                   # :38-44
    assert sol.status == "Optimal"
    assert abs((sol.obj_val - opt) / opt) < 2e-3
    assert np.max(np.abs(sol.x[:n] - xsave)) < 1e-3
    if tag == "pre15":                     # the two CG solves once (the oracle takes ~10 s for each)
        sol = orc.solve(model, orc.GAPA(eps=1e-4, verbose=0))                            # :29-35
        assert sol.status == "Optimal"
        assert abs((sol.obj_val - opt) / opt) < 2e-3
        assert np.max(np.abs(sol.x[:n] - xsave)) < 1e-3
        sol = orc.solve(model, orc.GAPA(0.5, 0.9, eps=1e-9, verbose=0))                  # :47-53
        assert sol.status == "Optimal"
        assert abs((sol.obj_val - opt) / opt) < 1e-8
        assert np.max(np.abs(sol.x[:n] - xsave)) < 1e-7


def feasibility_case(orc):
    d = np.load(GOLD)
    A, xsol = d["feas_A"], d["feas_xsol"]
    b = A @ xsol                                                                        # :7
    return A, b, orc.Feasibility(orc.IndAffine(A, b), orc.IndBox(0.0, np.inf), 100)     # :9-12


def test_feasibility_fixture_is_the_julia_draw():
    d = np.load(GOLD)
    xsol, A = jr.feasibility_test_data()
    assert np.array_equal(xsol, d["feas_xsol"]) and np.array_equal(A, d["feas_A"])
    assert xsol[0] == d["A_v15"][0, 0]                  # same seed, same first word


def test_oracle_meets_testfeasibility_on_the_references_own_data():
    """test/testfeasibility.jl:15-44, every assertion, on the reference's data.  Which solves end :Optimal and which
    :Indeterminate depends on the numbers (with the pre-1.5 draw of the same seed the intersection is empty and nearly
    every status differs), so seven matching outcomes pin the set projections, the iterations and the stopping rule."""
    A, b, prob = feasibility_case(orc)
    sol, _ = orc.feasibility_solve(prob, orc.DR(eps=1e-8, verbose=0), checki=10)         # :15
    assert sol.status == "Optimal"                                                      # :17
    assert sol.x.min() > -1e-12                                                         # :18
    assert np.abs(A @ sol.x - b).max() < 1e-12                                          # :19
    sol, _ = orc.feasibility_solve(prob, orc.AP(eps=1e-8, verbose=0), checki=1)          # :21
    assert sol.status == "Indeterminate"                                                # :23
    sol, _ = orc.feasibility_solve(prob, orc.GAP(eps=1e-8, verbose=0))                   # :25
    assert sol.status == "Indeterminate"                                                # :27
    sol, _ = orc.feasibility_solve(prob, orc.FISTA(eps=1e-8, verbose=0))                 # :29
    assert sol.status == "Indeterminate"                                                # :31
    for alg in (orc.GAPP(eps=1e-8, verbose=0, proji=50, out=[]), orc.GAPA(eps=1e-8, verbose=0),
                orc.LineSearchWrapper(orc.GAP(eps=1e-8, verbose=0), out=[])):            # :33-36
        sol, _ = orc.feasibility_solve(prob, alg)
        assert sol.status == "Optimal"                                                  # :41
        assert sol.x.min() > -1e-12                                                     # :42
        assert np.abs(A @ sol.x - b).max() < 1e-6                                       # :43
    # the pre-1.5 draw of the same lines is a different problem altogether (empty intersection)
    rng = jr.JuliaMersenneTwister(2)
    xs = jr.randn_scalar_fill(rng, 100)
    A0 = jr.randn_scalar_fill(rng, 50, 100)
    assert scipy.optimize.nnls(A0, A0 @ xs)[1] > 1.0
