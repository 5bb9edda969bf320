"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them (guards against silent edits of the checker).
GPU: the HIP path reproduces them through the C ABI, WITHOUT importing the oracle.

What these fixtures can and cannot show: they come from THIS repository's numpy oracle (Julia is not available to run the
reference itself), so they catch drift of either side from the oracle -- not a mistake the oracle shares with nobody.  The
oracle's agreement with the reference rests on tests/test_oracle_reference.py (the reference's own known answers: the README
NNLS problem, test/testPSD.jl's 2x2 projection, test/testprint.jl's formats, the CG / AffinePlusLinear / HSDEMatrix identities
against dense linear algebra) and, for the cones whose arithmetic lives in ProximalOperators.jl, on the oracle-free projection
certificates (tests/cone_certificates.py)."""
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

GOLD = Path(__file__).resolve().parent / "golden"
NAMES = {0: "Free", 1: "Zero", 2: "NonNeg", 3: "NonPos", 4: "SOC", 5: "SOCRotated", 6: "SDP"}


def load_problem(z):
    m, n = int(z["m"]), int(z["n"])
    A = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(m, n))
    K1 = [(NAMES[int(t)], int(l)) for t, l in z["K1"]]
    K2 = [(NAMES[int(t)], int(l)) for t, l in z["K2"]]
    return A, z["b"], z["c"], K1, K2


OPS = ["small_mixed", "c4_tiny", "c1_nnls"]
SOLVES = ["c1_nnls_dr", "psd2x2_dr", "small_mixed_dr"]


@pytest.mark.parametrize("name", OPS)
def test_oracle_reproduces_operator_goldens(name, oracle):
    orc = oracle
    z = np.load(GOLD / ("%s_operators.npz" % name))
    A, b, c, K1, K2 = load_problem(z)
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    mo = orc.Model(A, b, c, codes(K1), codes(K2))
    Q = orc.HSDEMatrixQ(A, b, c)
    l = A.shape[0] + A.shape[1] + 1
    out = np.empty(l)
    assert np.allclose(Q.mul(out, z["xl"]), z["q_out"], rtol=1e-14, atol=1e-15)
    outN = np.empty(2 * l)
    orc.KKTMatrix(Q).mul(outN, z["xN"])
    assert np.allclose(outN, z["kkt_out"], rtol=1e-14, atol=1e-15)
    orc.DualConeProduct(mo.K1, mo.K2).prox(outN, z["xN"])
    assert np.allclose(outN, z["cone_out"], rtol=1e-13, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("name", OPS)
def test_hip_reproduces_operator_goldens(name, pkg):
    z = np.load(GOLD / ("%s_operators.npz" % name))
    A, b, c, K1, K2 = load_problem(z)
    d = pkg.HipHSDE(A, b, c, K1, K2)
    rel = lambda a, r: np.linalg.norm(a - r) / max(1e-300, np.linalg.norm(r))
    assert rel(d.q_apply(z["xl"]), z["q_out"]) < 1e-13
    assert rel(d.q_apply(z["xl"], transpose=True), z["qt_out"]) < 1e-13
    assert rel(d.kkt_apply(z["xN"]), z["kkt_out"]) < 1e-13
    assert np.linalg.norm(d.prox_cones(z["xN"]) - z["cone_out"]) <= 1e-12 * np.linalg.norm(z["xN"])
    res = d.check(z["zc"], 1e-5)
    got = np.array([res.p, res.d, res.g, res.ctx, res.bty, res.kappa, res.tau, res.norm_axs, res.norm_aty, res.norm_b, res.norm_c])
    assert np.allclose(got, z["res"], rtol=1e-12, atol=1e-14)
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SOLVES)
def test_hip_reproduces_solve_goldens(name, pkg):
    """Whole DR solves: same status, iteration count within one check interval, solution and final residuals
    within the accuracy eps gives (1e-8-class problems: x within 1e-6, p/d/g within 1e-8)."""
    z = np.load(GOLD / ("%s_solve.npz" % name))
    A, b, c, K1, K2 = load_problem(z)
    eps, checki, max_iters = float(z["opts"][0]), int(z["opts"][1]), int(z["opts"][2])
    prob = pkg.workloads.ConicProblem(name, A, b, c, K1, K2)
    model = pkg.solve(prob, pkg.DR(eps=eps, checki=checki, max_iters=max_iters, verbose=0))
    assert model.status() == str(z["status"][0])
    assert abs(model.iterations - int(z["iterations"])) <= checki
    tol_x = 1e-6 if eps <= 1e-8 else 2e-3
    assert np.max(np.abs(model.getsolution() - z["x"])) < tol_x
    assert model.getobjval() == pytest.approx(float(z["obj"]), abs=tol_x)
    last = model.status_obj.last
    for key in ("p", "d", "g"):
        assert abs(getattr(last, key) - z["hist_" + key][-1]) < max(1e-8, eps)
    # the first recorded check (iteration `checki`) agrees closely: before the two trajectories drift apart
    assert model.history["p"][0][0] == int(z["hist_iter"][0])
    assert model.history["p"][0][1] == pytest.approx(float(z["hist_p"][0]), rel=0.05)


def test_mid_size_solve_fixture_is_the_oracles(pkg, oracle):
    """tests/golden/mid_mixed_solves.npz (the three whole solves the GPU suite compares the device with at l ~ 1e4) against the oracle,
    without repeating the solves: the stored end points evaluated with the oracle's residual formulas and status decision
    (HSDEStatus.jl:27-63) give the stored p, d, g and status."""
    orc = oracle
    gold = np.load(GOLD / "mid_mixed_solves.npz")
    prob = pkg.workloads.mid_mixed()
    mo = orc.Model(prob.A, prob.b, prob.c, [(orc.CONE_CODES[k], l) for k, l in prob.K1], [(orc.CONE_CODES[k], l) for k, l in prob.K2])
    for name in ("DR", "GAPA", "FISTA"):
        z = gold[name + "_zchecked"]                    # the point of the last check of the solve
        eps = float(gold[name + "_opts"][0])
        res = orc.residuals(mo, z)
        assert np.allclose([res["p"], res["d"], res["g"]], gold[name + "_pdg"], rtol=1e-9, atol=1e-15), name
        status = orc.decide_status(res, eps)
        assert ("Indeterminate" if status == "Continue" else status) == str(gold[name + "_status"][0])
        assert gold[name + "_x"].shape == (prob.n,) and np.isfinite(gold[name + "_x"]).all()
