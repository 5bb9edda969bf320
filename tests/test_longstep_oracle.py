"""
CPU: the oracle's restatement of the LongstepWrapper (src/wrappers/longstep.jl, saveplanes.jl).  The reference hands the projection onto
the saved planes to QPDAS (a package outside the checkout, BigFloat): the problem is a projection onto a polyhedron, its solution is unique,
so the restated solver is checked against an independent one (scipy SLSQP on the primal); the wrapper's bookkeeping -- which iterations
save, which rows they write, which rows the projection reads as equalities -- is checked against the reference's index arithmetic.
"""
import math

import numpy as np
import pytest
from scipy.optimize import minimize


def _slsqp(A, b, C, d, x, v0):
    cons = [{"type": "eq", "fun": lambda v: A @ v - b, "jac": lambda v: A}, {"type": "ineq", "fun": lambda v: C @ v - d, "jac": lambda v: C}]
    r = minimize(lambda v: 0.5 * v @ v - x @ v, v0, jac=lambda v: v - x, constraints=cons, method="SLSQP", options={"ftol": 1e-15, "maxiter": 1000})
    return r.x


@pytest.mark.parametrize("seed", range(8))
def test_projection_onto_planes_matches_slsqp(oracle, seed):
    orc = oracle
    rng = np.random.default_rng(seed)
    n, ne, ni = int(rng.integers(8, 40)), int(rng.integers(1, 5)), int(rng.integers(1, 5))
    A, C, x, v0 = rng.standard_normal((ne, n)), rng.standard_normal((ni, n)), rng.standard_normal(n), rng.standard_normal(n)
    b = A @ v0
    d = C @ v0 - rng.random(ni) * (rng.random(ni) < 0.6)          # feasible at v0; some inequalities tight there
    A0, b0 = A, b
    if seed % 3 == 0:                                            # a duplicated plane: singular Gram matrix (SLSQP gets the problem without it)
        A = np.vstack([A, A[:1]])
        b = np.concatenate([b, b[:1]])
    v, viol = orc.project_onto_planes(A, b, C, d, x)
    assert viol <= 1e-10
    assert np.abs(A @ v - b).max() <= 1e-9 and (C @ v - d).min() >= -1e-9
    assert np.abs(v - _slsqp(A0, b0, C, d, x, v0)).max() <= 1e-7
    # optimality directly: v - x = A'lam + C'mu with mu >= 0 and complementary slackness, checked through a second projection
    v2, _ = orc.project_onto_planes(A, b, C, d, v)
    assert np.abs(v2 - v).max() <= 1e-9


def test_wrapper_bookkeeping_follows_the_reference(pkg, oracle):
    """longstep.jl:45-58 with longinterval = 10, nsave = 2: iterations 8, 9, 10 (18, 19, 20, ...) save, rows are written equality,
    inequality, equality, ... (0-based 0..5), the projection follows iteration 10 and reads rows 0..2 as equalities, 3..5 as inequalities."""
    orc = oracle
    prob = pkg.workloads.small_mixed()
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    mo = orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))
    w = orc.LongstepWrapper(orc.DR(direct=True), longinterval=10, nsave=2)
    w.init(mo)
    x = orc.hsde_initialvalue(mo)
    st = orc.HSDEStatus(mo, 10 ** 9, 1e-9, 0, 1, S1=w.alg.S1)
    written = []
    for i in range(1, 24):
        before = np.isnan(w.saved.b).copy()
        xin = x.copy()
        st.i = i
        w.step(x, i, st)
        newly = np.nonzero(before & ~np.isnan(w.saved.b))[0].tolist()
        written.append((i, newly))
        if i in (8, 9, 10):
            assert newly == [2 * (i - 8), 2 * (i - 8) + 1], (i, newly)
        if i == 10:
            assert len(w.log) == 1 and w.log[0][0] == 10 and w.savepos == -1
            s = w.saved
            v = x
            assert np.abs(s.A[:3] @ v - s.b[:3]).max() <= 1e-8 and (s.A[3:] @ v - s.b[3:]).min() >= -1e-8     # on the planes it was projected onto
        if i < 8:
            assert newly == [] and w.savepos == 0
        if 10 < i < 18:
            assert w.savepos == -1
    assert [i for i, _ in w.log] == [10, 20]
    with pytest.raises(ValueError):
        orc.LongstepWrapper(orc.GAPP())


def test_wrapped_solve_runs_the_projection_every_interval(pkg, oracle):
    """A whole solve through solve(): one projection per longinterval, each a feasible point of its planes.  (The wrapper is experimental in
    the reference -- it reads interleaved rows as equalities / inequalities and keeps the half-space the sets are NOT in -- and, restated
    faithfully, slows DR down on this problem: 940 iterations unwrapped, the cap with it.  Nothing here asserts that it helps.)"""
    orc = oracle
    prob = pkg.workloads.c1_readme_nnls(seed=2)
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    mo = orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))
    w = orc.LongstepWrapper(orc.DR(eps=1e-6, verbose=0, checki=10, direct=True, max_iters=600), longinterval=50, nsave=3)
    sol = orc.solve(mo, w)
    assert sol.status in ("Optimal", "Indeterminate")
    assert [i for i, _ in w.log] == list(range(50, 601, 50)) and max(v for _, v in w.log) <= 1e-10


def test_python_mirror_wrapper_class(pkg):
    """The host mirror's LongstepWrapper (interface.py): what the reference's constructor does (longstep.jl:22-24, 28) -- options merged with the
    wrapped algorithm's winning, the algorithm's code and relaxation parameters handed on, GAPP refused."""
    w = pkg.LongstepWrapper(pkg.GAPA(0.8, 0.5, eps=1e-7, checki=10), longinterval=50, nsave=3, eps=1e-3, verbose=0)
    assert (w.longinterval, w.nsave) == (50, 3)
    assert w.options["eps"] == 1e-7 and w.options["checki"] == 10 and w.options["verbose"] == 0      # [kwargs..., alg.options...]: the algorithm's win
    assert w._alg_args() == w.alg._alg_args() and w.direct == w.alg.direct
    for bad in (pkg.GAPP(),):
        with pytest.raises(ValueError):
            pkg.LongstepWrapper(bad)
    for ok in (pkg.DR(), pkg.AP(), pkg.GAP(), pkg.FISTA(), pkg.Dykstra()):
        pkg.LongstepWrapper(ok)
