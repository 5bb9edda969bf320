"""CPU: the plain-C baseline (oracle/fos_cport.c) against the numpy oracle it restates -- operator, CG-based affine
projection, every cone kind it implements (incl. its own tred2/tql2 eigen-solver vs LAPACK), whole GAP / GAPA steps,
and the multi-threaded variant against the single-threaded one."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import fos_cport as cp  # noqa: E402


def _codes(orc, cones):
    return [(orc.CONE_CODES[k], l) for k, l in cones]


@pytest.fixture(scope="module")
def problems(pkg):
    w = pkg.workloads
    return [w.small_lp(), w.small_mixed(), w.c4_block_sdp(nblocks=3, k=9, p=4), w.c1_readme_nnls(),
            w.c5_mixed(nblocks=2, nb_cols=30, nonneg=8, nsoc=2, socdim=5, npsd=2, k=7, density=0.3)]


def test_operator_and_projections(problems, oracle):
    orc = oracle
    rng = np.random.default_rng(1)
    for prob in problems:
        if any(k.startswith("Exp") for k, _ in prob.K1 + prob.K2):
            continue
        om = orc.Model(prob.A, prob.b, prob.c, _codes(orc, prob.K1), _codes(orc, prob.K2))
        c = cp.CPort(prob.A, prob.b, prob.c, _codes(orc, prob.K1), _codes(orc, prob.K2))
        N = 2 * (prob.m + prob.n + 1)
        x = rng.standard_normal(N)
        ref = np.empty(N)
        orc.KKTMatrix(orc.HSDEMatrixQ(prob.A, prob.b, prob.c)).mul(ref, x)
        assert np.allclose(c.kkt_mul(x), ref, rtol=1e-13, atol=1e-13)
        orc.DualConeProduct(om.K1, om.K2).prox(ref, x)
        got = c.prox_cones(x)
        assert np.linalg.norm(got - ref) <= 1e-12 * max(1.0, np.linalg.norm(ref)), prob.name
        # the affine projection: same CG stop iteration; both results satisfy the stopping rule, hence (the KKT matrix has
        # singular values >= 1) lie within tol of the exact projection and within 2 tol of each other.  Closer than that is
        # not guaranteed: plain CG on the indefinite system amplifies summation-order differences.
        S1, _, _ = orc.hsde_sets(om)
        M = orc.KKTMatrix(orc.HSDEMatrixQ(prob.A, prob.b, prob.c))
        y = np.empty(N)
        for call in range(3):
            tol = S1.tolerance()
            if call > 0:
                c.set_affine_state(S1.cgdata.xinit, S1.i)      # same warm start and call counter as the oracle
            S1.prox(y, x)
            got = c.prox_affine(x)
            assert c.cgiter() == S1.getcgiter(), (prob.name, call)
            res = np.empty(N)
            M.mul(res, got)
            assert np.linalg.norm(res - S1.rhs) <= 1.01 * tol + 1e-9, (prob.name, call)
            assert np.linalg.norm(got - y) <= 2.0 * tol, (prob.name, call)
            if True:
                assert np.linalg.norm(got - y) <= 1e-5 * max(1.0, np.linalg.norm(y)), (prob.name, call)
            x = 0.5 * x + 0.5 * y
        c.close()


@pytest.mark.parametrize("algname", ["DR", "GAP", "GAPA", "FISTA"])
def test_steps_match_oracle(problems, oracle, algname):
    orc = oracle
    for prob in problems[:3]:
        om = orc.Model(prob.A, prob.b, prob.c, _codes(orc, prob.K1), _codes(orc, prob.K2))
        alg = {"DR": orc.DR, "GAP": orc.GAP, "GAPA": lambda: orc.GAPA(0.9, 0.3), "FISTA": lambda: orc.FISTA(0.9)}[algname]()
        alg.init(om)
        x = orc.hsde_initialvalue(om)
        xc = x.copy()
        st = orc.HSDEStatus(om, 10 ** 9, 1e-8, 0, 0)
        c = cp.CPort(prob.A, prob.b, prob.c, _codes(orc, prob.K1), _codes(orc, prob.K2))
        a12 = 2.0
        for i in range(1, 6):
            st.i = i
            # every step starts from the oracle's state (iterate, warm start, call counter, alpha12): plain CG on the
            # indefinite KKT system amplifies rounding differences from step to step
            xc[:] = x
            if i > 1:
                c.set_affine_state(alg.S1.cgdata.xinit, alg.S1.i)
            a12 = getattr(alg, "alpha12", 2.0)
            if algname == "FISTA":                          # FISTAData handed over like the rest (fista.jl:15-25; y = x at i == 1, :31-33)
                yc, xoldc, tc = (x.copy() if i == 1 else alg.y.copy()), alg.xold.copy(), alg.t
            alg.step(x, i, st)
            if algname == "FISTA":
                tc = c.fista_step(xc, alg.alpha, yc, xoldc, tc)
                assert tc == alg.t
                assert np.linalg.norm(yc - alg.y) <= 1e-5 * max(1.0, np.linalg.norm(alg.y)), (prob.name, i)
                assert np.array_equal(xoldc, alg.xold)
            elif algname == "GAPA":
                a12 = c.gapa_step(xc, alg.alpha, alg.beta, a12)
                assert a12 == pytest.approx(alg.alpha12, rel=1e-5)
            else:
                c.gap_step(xc, alg.alpha, alg.alpha1, alg.alpha2)
            assert c.cgiter() == alg.S1.getcgiter(), (prob.name, i)
            assert np.linalg.norm(xc - x) <= 1e-5 * max(1.0, np.linalg.norm(x)), (prob.name, i)
        c.close()


def test_threads_agree(problems, oracle):
    orc = oracle
    prob = problems[2]
    rng = np.random.default_rng(3)
    x = rng.standard_normal(2 * (prob.m + prob.n + 1))
    outs = []
    for threads in (1, 4):
        c = cp.CPort(prob.A, prob.b, prob.c, _codes(orc, prob.K1), _codes(orc, prob.K2), threads=threads)
        outs.append((c.kkt_mul(x), c.prox_cones(x)))
        c.close()
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-13, atol=1e-13)      # CSC scatter vs CSR gather: order only
    assert np.array_equal(outs[0][1], outs[1][1])
    assert cp.max_threads() >= 1
