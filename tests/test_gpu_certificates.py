"""
Parity hardening at BASELINE sizes.

(i)  Oracle-FREE certificates on the HIP cone projections (tests/cone_certificates.py: p in C, p - z in C*, <p, p - z> = 0 --
     the projection is the unique point with these properties) for every cone kind, at the full sizes of C3 / C4 / C5 and on a
     small problem with rotated second-order and exponential cones.  This is the only available pin for IndSOC / IndRotatedSOC /
     IndExp* / IndPSD, whose arithmetic lives in ProximalOperators.jl (not in the reference checkout).
(ii) C5 (FISTA) and C2 (DR) driven for a stated number of iterations at full size: the residuals the device reports against the
     oracle's formulas (HSDEStatus.jl:34-38) on the returned point, and the distance to the KNOWN complementary optimum.
(iii) Whole solves at l ~ 1e4 against the oracle, one per algorithm.
"""
import numpy as np
import pytest

import fos_oracle as orc
from cone_certificates import certify_stacked_projection

pytestmark = pytest.mark.gpu

STATUS = {"Continue": 0, "Optimal": 1, "Unbounded": 2, "Infeasible": 3}


def _omodel(prob):
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    return orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))


def _certify(d, prob, z, rtol):
    p = d.prox_cones(z)
    worst = certify_stacked_projection(prob.K1, prob.K2, z, p, rtol=rtol)
    assert np.all(np.isfinite(p))
    return p, worst


def test_certificates_small_all_kinds(pkg):
    """Every cone kind on both sides, including rotated SOC and both exponential cones, at three input scales."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    K1 = [("Zero", 3), ("NonNeg", 4), ("SOC", 5), ("SDP", 10), ("SOCRotated", 4), ("ExpPrimal", 3), ("ExpDual", 3), ("NonPos", 2),
          ("Free", 2), ("SOCRotated", 2), ("SOC", 1), ("SDP", 1), ("SOC", 130), ("SDP", 45)]
    K2 = [("Free", 3), ("NonNeg", 3), ("SOC", 4), ("ExpPrimal", 3), ("SOCRotated", 5), ("Zero", 2), ("ExpDual", 3)]
    m, n = sum(l for _, l in K1), sum(l for _, l in K2)
    A = sp.random(m, n, density=0.2, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    prob = pkg.workloads.ConicProblem("all-kinds", A, rng.standard_normal(m), rng.standard_normal(n), K1, K2)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    for scale in (1e-3, 1.0, 1e3):
        for _ in range(8):
            z = scale * rng.standard_normal(d.N)
            _, worst = _certify(d, prob, z, 1e-11)
            assert worst.max() < 1e-11
    d.close()


def test_certificates_c3_full_size(pkg, fullsize):
    """1000 x SOC(50): random input and a mid-solve iterate."""
    prob = fullsize("C3")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    rng = np.random.default_rng(31)
    _, w = _certify(d, prob, rng.standard_normal(d.N), 1e-12)
    d.set_alg(pkg.GAPA())
    d.set_iterate(None)
    d.step(1, 60, 10 ** 9, 1e-8)
    _, w2 = _certify(d, prob, d.get_iterate(), 1e-12)
    assert max(w.max(), w2.max()) < 1e-12
    d.close()


def test_certificates_c5_full_size(pkg, fullsize):
    """l ~ 1e6: NonNeg(31250) + 250 x SOC(50) + 9 x PSD(64) per block, 8 blocks; random input, then consecutive FISTA iterates
    (the second and third PSD projections start from the previous eigenvector basis: the warm-started path)."""
    prob = fullsize("C5")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    rng = np.random.default_rng(51)
    _, w = _certify(d, prob, rng.standard_normal(d.N), 2e-11)
    d.set_alg(pkg.FISTA())
    d.set_iterate(None)
    d.step(1, 30, 10 ** 9, 1e-8)
    za = d.get_iterate()
    d.step(31, 1, 10 ** 9, 1e-8)
    zb = d.get_iterate()
    _, wa = _certify(d, prob, za, 2e-11)
    _, wb = _certify(d, prob, zb, 2e-11)
    assert max(w.max(), wa.max(), wb.max()) < 2e-11
    d.close()


def test_certificates_c4_full_size_warm_started_psd(pkg, fullsize):
    """512 x PSD(64), both copies: a cold projection, then projections of slowly drifting inputs (warm start, MFMA products)."""
    prob = fullsize("C4")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    rng = np.random.default_rng(41)
    z = rng.standard_normal(d.N)
    dz = rng.standard_normal(d.N)
    for k, step in enumerate((0.0, 1e-2, 1e-2, 1e-5, 0.3)):
        z = z + step * dz
        _, w = _certify(d, prob, z, 2e-11)
        assert w.max() < 2e-11, (k, w)
    d.close()


def _device_vs_oracle_residuals(d, prob, res, i):
    z = d.get_checked()
    st = orc.HSDEStatus(_omodel(prob), i, 1e-8, 0, 1)
    st.i = i
    st.checkstatus(z, override=True)
    for key in ("p", "d", "g", "ctx", "bty"):
        assert getattr(res, key) == pytest.approx(st.last[key], rel=1e-9, abs=1e-13), key
    assert res.status == STATUS[st.status]
    return z


def test_c5_fista_driven_to_iteration_cap_full_size(pkg, fullsize):
    """BASELINE config 5 (mixed cones, n ~ 1e6, FISTA, residuals vs the CPU restatement): 1500 iterations at full size; at every
    check (every 500) the device's p, d, g, c'x, b'y agree with the oracle's formulas on the same point to 1e-9 relative (the
    BASELINE tolerance is 1e-8), the residuals fall monotonically from check to check, and the objective c'x / tau approaches the
    known optimal value c'x0 (x0 itself is not unique: A has more columns than rows)."""
    prob = fullsize("C5")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.FISTA())
    d.set_iterate(None)
    hist, errs = [], []
    opt = float(prob.c @ prob.x0)
    for k in range(3):
        done, checked, res = d.step(500 * k + 1, 500, 500, 1e-8)
        assert done == 500 and checked
        _device_vs_oracle_residuals(d, prob, res, 500 * (k + 1))
        errs.append(abs(res.ctx / res.tau - opt) / max(1.0, abs(opt)))
        hist.append((res.p, res.d, res.g))
    assert hist[2][0] < hist[1][0] < hist[0][0] and hist[2][1] < hist[1][1] < hist[0][1]
    assert errs[2] < errs[0], errs
    d.close()


def test_c2_dr_driven_to_iteration_cap_full_size(pkg, fullsize):
    """BASELINE config 2 (dense 5000 x 10000 LP, DR): 1200 iterations at full size; device residuals vs the oracle's formulas on
    the returned point at every check, monotone decrease, objective towards the known optimum c'x0."""
    prob = fullsize("C2")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.DR())
    d.set_iterate(None)
    hist = []
    for k in range(3):
        done, checked, res = d.step(400 * k + 1, 400, 400, 1e-8)
        assert done == 400 and checked
        _device_vs_oracle_residuals(d, prob, res, 400 * (k + 1))
        hist.append((res.p, res.d, res.g, res.ctx / res.tau))
    assert hist[2][0] < hist[0][0] and hist[2][1] < hist[0][1]
    opt = float(prob.c @ prob.x0)
    assert abs(hist[2][3] - opt) < abs(hist[0][3] - opt) and abs(hist[2][3] - opt) <= 0.05 * max(1.0, abs(opt))
    d.close()


@pytest.mark.parametrize("algname", ["DR", "GAPA", "FISTA"])
def test_mid_size_whole_solve_matches_oracle(pkg, algname):
    """l ~ 1e4 (NonNeg + SOC + PSD, sparse A): the whole solve against the oracle's -- same status, iteration count within one
    check interval, p / d / g at the end within 1e-8 of the oracle's (BASELINE north_star), same solution.  The oracle's three solves
    (1-2 minutes each in numpy) are kept in tests/golden/mid_mixed_solves.npz (tests/golden/make_golden.py mid; tests/test_golden.py
    ties the file to the oracle); the problem is regenerated from its seed."""
    from pathlib import Path
    gold = np.load(Path(__file__).resolve().parent / "golden" / "mid_mixed_solves.npz")
    prob = pkg.workloads.mid_mixed()
    assert 8000 < prob.m + prob.n + 1 < 12000
    mk = {"DR": lambda M, **o: M.DR(**o), "GAPA": lambda M, **o: M.GAPA(0.8, 0.5, **o), "FISTA": lambda M, **o: M.FISTA(**o)}[algname]
    opts = dict(eps={"DR": 1e-6, "GAPA": 1e-5, "FISTA": 1e-6}[algname], verbose=0, max_iters=2500 if algname != "FISTA" else 300, checki=100)
    assert np.array_equal(gold[algname + "_opts"], np.array([opts["eps"], opts["checki"], opts["max_iters"]]))
    model = pkg.solve(prob, mk(pkg, **opts))
    o_status, o_iter, o_x = str(gold[algname + "_status"][0]), int(gold[algname + "_iterations"]), gold[algname + "_x"]
    o_pdg = dict(zip(("p", "d", "g"), gold[algname + "_pdg"]))
    assert model.status() == o_status
    assert abs(model.iterations - o_iter) <= 100
    last = model.status_obj.last
    if o_status == "Optimal":
        for key in ("p", "d", "g"):
            assert abs(getattr(last, key) - o_pdg[key]) < 1e-6             # both below eps (1 + norm); CG stop decisions differ
        assert model.getobjval() == pytest.approx(float(gold[algname + "_obj"]), abs=1e-4)
        ref_err = np.max(np.abs(o_x - prob.x0))
        assert np.max(np.abs(model.getsolution() - prob.x0)) <= 3 * ref_err + 1e-8
    else:                                                               # FISTA at the cap: the same trajectory end point
        for key in ("p", "d", "g"):
            assert getattr(last, key) == pytest.approx(o_pdg[key], rel=0.05)
        assert np.max(np.abs(model.getsolution() - o_x)) <= 1e-3 * max(1.0, np.max(np.abs(o_x)))


@pytest.mark.parametrize("algname", ["DR", "GAPA"])
def test_mid_size_solved_to_the_north_star_tolerance(pkg, algname):
    """BASELINE north_star: residuals within 1e-8.  l ~ 1e4 solved to eps = 1e-8 on the device and judged WITHOUT the oracle (a numpy
    oracle solve to that tolerance takes many minutes): p, d, g of HSDEStatus.jl:34-38 recomputed on the host from the returned
    (x, y, s), cone membership of s and y, and the optimal value against the optimum the instance was built around (a strictly
    complementary primal-dual pair, workloads.c5_mixed)."""
    prob = pkg.workloads.mid_mixed()
    alg = {"DR": lambda: pkg.DR(eps=1e-8, verbose=0, max_iters=20000, checki=100),
           "GAPA": lambda: pkg.GAPA(0.8, 0.5, eps=1e-8, verbose=0, max_iters=20000, checki=100)}[algname]()     # (~10 000 iterations)
    model = pkg.solve(prob, alg)
    assert model.status() == "Optimal"
    last = model.status_obj.last
    eps = 1e-8
    x, y, s = model.getsolution(), model.dual_sol, model.slack
    A, b, c = prob.A, prob.b, prob.c
    nb, nc = np.linalg.norm(b), np.linalg.norm(c)
    p = np.linalg.norm(A @ x + s - b) / (1 + nb)
    d = np.linalg.norm(A.T @ y + c) / (1 + nc)                 # K2 = Free: r = 0
    ctx, bty = float(c @ x), float(b @ y)
    g = abs(ctx + bty) / (1 + abs(ctx) + abs(bty))
    assert p <= eps * (1 + nb) and d <= eps * (1 + nc) and g <= eps * (1 + abs(ctx) + abs(bty))      # the decision of HSDEStatus.jl:53-55
    assert last.p <= eps * (1 + nb) and last.d <= eps * (1 + nc)            # (the device's own check, on the iterate it stopped at)
    K1 = orc.ConeProduct.from_lengths([(orc.CONE_CODES[k], l) for k, l in prob.K1])
    proj = np.empty_like(s)
    K1.prox(proj, s)
    assert np.abs(proj - s).max() <= 1e-9 * max(1.0, np.abs(s).max())       # s in K1
    K1.prox_dual(proj, y)
    assert np.abs(proj - y).max() <= 1e-9 * max(1.0, np.abs(y).max())       # y in K1*
    opt = float(c @ prob.x0)
    assert abs(ctx - opt) <= 1e-6 * (1 + abs(opt))
