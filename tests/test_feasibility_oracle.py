"""
CPU: the oracle's restatement of the Feasibility form (src/problemforms/Feasibility/Feasibility.jl, FeasibilityStatus.jl) on the
reference's own test (test/testfeasibility.jl): the assertions that test makes, and the status semantics line by line.
"""
import numpy as np
import pytest

from feasibility_cases import ALGS, GAPP, affine_box_instance


def _problem(orc, **kw):
    A, b = affine_box_instance(**kw)
    return A, b, orc.Feasibility(orc.IndAffine(A, b), orc.IndBox(0.0, np.inf), A.shape[1])


def test_reference_test_assertions(oracle):
    """testfeasibility.jl:15-19: DR(eps=1e-8), checki=10 -> :Optimal, minimum(x) > -1e-12, |A x - b| < 1e-12; :36-44 GAPA likewise."""
    orc = oracle
    A, b, prob = _problem(orc)
    sol, model = orc.feasibility_solve(prob, orc.DR(eps=1e-8, verbose=0), checki=10)
    assert sol.status == "Optimal" and model.solve_stat == "Optimal"
    assert sol.x.min() > -1e-12 and np.abs(A @ sol.x - b).max() < 1e-12
    assert sol.iterations % 10 == 0                        # stops at a check iteration
    sol, _ = orc.feasibility_solve(prob, orc.GAPA(eps=1e-8, verbose=0))
    assert sol.status == "Optimal" and sol.x.min() > -1e-12 and np.abs(A @ sol.x - b).max() < 1e-6


def test_status_semantics(oracle):
    orc = oracle
    A, b, prob = _problem(orc)
    # Continue -> :Indeterminate when max_iters ends the loop (Feasibility.jl:62-65); the forced check on the guess runs because the
    # last iteration was not a check iteration (solverwrapper.jl:31-33)
    sol, model = orc.feasibility_solve(prob, orc.AP(eps=1e-12, verbose=0), max_iters=7, checki=5)
    assert sol.status == "Indeterminate" and sol.iterations == 7
    assert [i for i, _ in model.history["err"]] == [5, 7]
    # kwargs of solve! override the algorithm's options (Feasibility.jl:37-41)
    sol, model = orc.feasibility_solve(prob, orc.DR(eps=1e-8, verbose=0, checki=1000), checki=10)
    assert sol.iterations < 1000
    # prev starts as NaN: the first check can never stop the solve (err = NaN), and prev is refreshed at EVERY iteration, so err at
    # a check is the distance between two consecutive iterates
    alg = orc.DR(eps=1e30, verbose=0)
    model = orc.FeasibilityModel(prob, alg)
    st = orc.FeasibilityStatus(model, 1, 1e30, 0, 1)
    x = np.zeros(prob.n)
    zs = []
    for i in (1, 2, 3):
        st.i = i
        alg.step(x, i, st)
        zs.append(st.prev.copy())
        if i == 1:
            assert st.status == "Continue" and np.isnan(st.err)
        else:
            assert st.status == "Optimal" and st.err == pytest.approx(np.linalg.norm(zs[-1] - zs[-2]), rel=1e-14)
    # the header of printstatusheader (direct = true: no cg column, FeasibilityStatus.jl:74-84)
    lines = []
    orc.feasibility_solve(prob, orc.DR(eps=1e-8), out=lines, checki=10)
    assert lines[1] == "-" * 22 and lines[2] == " Iter | res | time"
    assert lines[-3].startswith("Found solution i=") and lines[-2] == "Time for iterations: "


def test_set_projections_are_projections(oracle):
    orc = oracle
    A, b, _ = _problem(orc)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(A.shape[1])
    y = np.empty_like(x)
    orc.IndAffine(A, b).prox(y, x)
    assert np.abs(A @ y - b).max() < 1e-12
    N = np.linalg.svd(A)[2][A.shape[0]:]                   # a basis of null(A): x - y must be orthogonal to it
    assert np.abs(N @ (x - y)).max() < 1e-12
    orc.IndBox(-0.5, 0.25).prox(y, x)
    assert np.array_equal(y, np.clip(x, -0.5, 0.25))


@pytest.mark.parametrize("algname", sorted(ALGS))
def test_every_algorithm_reaches_the_intersection(oracle, algname):
    orc = oracle
    A, b, prob = _problem(orc, m=50, n=100, boundary=False)
    sol, _ = orc.feasibility_solve(prob, ALGS[algname](orc, eps=1e-9, verbose=0, max_iters=20000), checki=10)
    assert sol.status == "Optimal"
    assert sol.x.min() > -1e-7 and np.abs(A @ sol.x - b).max() < 1e-6


def test_gapp_on_the_reference_test(oracle):
    """testfeasibility.jl:36-44 runs GAPP(eps=1e-8, verbose=0, proji=50) -- `proji` is not the keyword (`iproj`), so the search interval stays
    100 -- and expects :Optimal with |A x - b| < 1e-6; the search prints 21 test norms and the chosen step (gapproj.jl:51,57)."""
    orc = oracle
    A, b, prob = _problem(orc, m=80, n=100)
    lines = []
    sol, _ = orc.feasibility_solve(prob, orc.GAPP(eps=1e-8, verbose=0, proji=50, out=lines))
    assert sol.status == "Optimal" and sol.x.min() > -1e-9 and np.abs(A @ sol.x - b).max() < 1e-6
    assert sum(l.startswith("normtest: ") for l in lines) == 21 * (sol.iterations // 100)
    # a search step never leaves the affine set: x_new = P_S2-relaxed point built from tmp1 + a res with tmp1, res in S1 - S1
    alg = GAPP(orc, iproj=3, out=[])
    model = orc.FeasibilityModel(prob, alg)
    st = orc.FeasibilityStatus(model, 10 ** 9, 0.0, 0, 0)
    x = np.zeros(prob.n)
    for i in (1, 2, 3):
        st.i = i
        alg.step(x, i, st)
    assert alg.log and alg.log[0][0] == 3 and len(alg.log[0][1]) == 21 and alg.log[0][2] in [2.0 ** k for k in range(21)]
