"""Edge cases through the C ABI: tiny and degenerate shapes, many small cones, one huge cone, zero operator,
re-use of a handle across algorithms, checkpoint/resume of the affine state."""
import numpy as np
import pytest
import scipy.sparse as sp

import fos_oracle as orc

pytestmark = pytest.mark.gpu


def _codes(cones):
    return [(orc.CONE_CODES[k], l) for k, l in cones]


def _cone_ref(K1, K2, z):
    S2 = orc.DualConeProduct(orc.ConeProduct.from_lengths(_codes(K1)), orc.ConeProduct.from_lengths(_codes(K2)))
    ref = np.empty_like(z)
    S2.prox(ref, z)
    return ref


def test_one_by_one_and_zero_operator(pkg):
    A = sp.csc_matrix(np.array([[2.0]]))
    d = pkg.HipHSDE(A, np.array([1.0]), np.array([-1.0]), [("NonNeg", 1)], [("NonNeg", 1)])
    z = np.arange(1.0, 7.0)
    Q = orc.HSDEMatrixQ(A, np.array([1.0]), np.array([-1.0]))
    ref = np.empty(6)
    orc.KKTMatrix(Q).mul(ref, z)
    assert np.allclose(d.kkt_apply(z), ref, rtol=1e-14)
    d.close()
    Z = sp.csc_matrix((5, 7))                       # no stored entries at all
    d = pkg.HipHSDE(Z, np.ones(5), np.ones(7), [("Zero", 5)], [("Free", 7)])
    x = np.random.default_rng(0).standard_normal(d.l)
    Q = orc.HSDEMatrixQ(Z, np.ones(5), np.ones(7))
    ref = np.empty(d.l)
    Q.mul(ref, x)
    assert np.allclose(d.q_apply(x), ref, rtol=1e-14, atol=1e-15)
    d.close()


def test_many_tiny_cones_and_one_huge_cone(pkg):
    rng = np.random.default_rng(1)
    # 3000 cones of 1..3 entries of every kind on the row side
    kinds = ["Zero", "Free", "NonNeg", "NonPos", "SOC", "SDP", "SOCRotated", "ExpPrimal"]
    K1 = []
    for i in range(3000):
        k = kinds[i % len(kinds)]
        ln = {"SDP": [1, 3][i % 2], "SOCRotated": 2 + i % 2, "ExpPrimal": 3}.get(k, 1 + i % 3)
        K1.append((k, ln))
    m = sum(l for _, l in K1)
    n = 40
    A = sp.random(m, n, density=0.02, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    d = pkg.HipHSDE(A, rng.standard_normal(m), rng.standard_normal(n), K1, [("Free", n)])
    z = rng.standard_normal(d.N)
    assert np.linalg.norm(d.prox_cones(z) - _cone_ref(K1, [("Free", n)], z)) <= 1e-9 * np.linalg.norm(z)
    d.close()
    # one SOC with 300 000 entries (a single wavefront walks it) + a PSD cone of order 1
    K1 = [("SOC", 300000), ("SDP", 1)]
    m = 300001
    A = sp.random(m, 3, density=1e-4, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    d = pkg.HipHSDE(A, np.zeros(m), np.zeros(3), K1, [("NonNeg", 3)])
    z = rng.standard_normal(d.N)
    assert np.linalg.norm(d.prox_cones(z) - _cone_ref(K1, [("NonNeg", 3)], z)) <= 1e-12 * np.linalg.norm(z)
    d.close()


def test_handle_reuse_across_algorithms_and_resume(pkg):
    """A handle keeps S1's state (call counter, CG warm start) across solves like the reference's model does
    (affinepluslinear.jl:66,114); fos_get/set_affine_state + fos_set_iterate resume a run bit for bit."""
    prob = pkg.workloads.small_mixed()
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.DR())
    d.set_iterate(None)
    d.step(1, 40, 10 ** 9, 1e-8)
    assert d.prox_count() == 41
    z40 = d.get_iterate()
    xinit, i, fr = d.get_affine_state()
    assert i == 41 and not fr
    d.step(41, 20, 10 ** 9, 1e-8)
    z60 = d.get_iterate()
    # resume from the checkpoint on a FRESH handle
    e = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    e.set_alg(pkg.DR())
    e.set_iterate(z40)
    e.set_affine_state(xinit, i)
    e.step(41, 20, 10 ** 9, 1e-8)
    # identical up to the PSD warm-start basis (a fresh handle starts its Jacobi from the identity: ~1e-14 different
    # projections, amplified by 20 iterations of the chaotic inexact-CG recurrence -- see test_gpu_parity.py)
    assert np.linalg.norm(e.get_iterate() - z60) <= 1e-7 * np.linalg.norm(z60)
    # switching the algorithm on a live handle keeps S1's counter
    d.set_alg(pkg.GAPA(0.8, 0.5))
    d.step(61, 5, 10 ** 9, 1e-8)
    assert d.prox_count() == 66 and d.alpha12() != 2.0
    d.set_alg(pkg.FISTA())
    d.set_iterate(None)
    done, checked, res = d.step(1, 10, 10, 1e-8)
    assert done == 10 and checked and np.isfinite(res.p)
    d.close()
    e.close()


def test_nan_input_propagates_like_the_reference(pkg):
    """The reference never guards NaN: a NaN iterate stays NaN, CG runs to its 1000-iteration cap
    (conjugategradients.jl:42: norm(NaN) <= tol is false) and the shim is told to warn."""
    prob = pkg.workloads.small_lp()
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.DR())
    z = np.zeros(d.N)
    z[0] = np.nan
    z[d.l - 1] = z[-1] = 1.0
    d.set_iterate(z)
    done, checked, res = d.step(1, 1, 1, 1e-8)
    assert done == 1 and checked
    assert d.cgiter() == 1000 and res.cg_maxiter_hit == 1
    assert np.isnan(d.get_iterate()).any()
    assert pkg.lib.STATUS_NAMES[res.status] == "Continue"
    d.close()


def test_profile_event_sampling_counts(pkg):
    """fos_profile(1) brackets every KKT launch of the CG iterations with events, fos_profile(N) those whose iteration number
    COUNTED OVER ALL SOLVES is a multiple of N (so every position inside a solve gets sampled); launches enqueued past
    convergence (gated no-ops) never count; fos_get_cg_total sums getcgiter."""
    prob = pkg.workloads.small_mixed()
    dev = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    dev.set_alg(pkg.DR())
    dev.set_cg_variant("reference")          # (a small cache-resident operator takes the merged recurrence by default: one sweep more per solve)
    dev.set_iterate(None)
    dev.step(1, 30, 10 ** 9, 1e-9)
    for period in (1, 3):
        dev.profile(period)
        dev.profile_read()
        c0 = dev.cg_total()
        total = 0
        for i in range(31, 41):
            dev.step(i, 1, 10 ** 9, 1e-9)
            total += dev.cgiter()
        expect = sum(1 for g in range(c0, c0 + total) if g % period == 0)
        launches, ms, nbytes = dev.profile_read()
        assert dev.cg_total() - c0 == total
        assert launches == expect and ms > 0.0 and nbytes > 0.0
        dev.profile(0)
    dev.close()


def test_nan_and_inf_through_dual_tiles_and_repartition(pkg, oracle):
    """A NaN / Inf entry of the input vector must reach exactly the outputs it reaches in the reference's four SpMV sweeps,
    also when the operator is stored as dual tiles (padding lanes and padded steps must not leak or swallow it); and
    re-partitioning the sweep (fos_set_tuning) must not change a single bit of any row sum."""
    import scipy.sparse as sp
    orc = oracle
    rng = np.random.default_rng(9)
    A = sp.vstack([sp.csc_matrix(rng.standard_normal((70, 45))),                                   # tiles: 64 + (6 rows left over)
                   sp.random(40, 45, density=0.1, format="csc", random_state=rng, data_rvs=rng.standard_normal)]).tocsc()
    m, n = A.shape
    prob = pkg.workloads.from_complementary_pair("tiles-nan", A, [("Zero", m)], [("NonNeg", n)], rng)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    assert d.operator_stats()["tiles"] > 0
    M = orc.KKTMatrix(orc.HSDEMatrixQ(prob.A, prob.b, prob.c))
    base = rng.standard_normal(d.N)
    for idx, bad in ((3, np.nan), (n + 10, np.nan), (n + 69, np.inf), (d.l + 7, np.nan), (d.l + n + 64, -np.inf), (n + 100, np.nan)):
        z = base.copy()
        z[idx] = bad
        ref = np.empty(d.N)
        with np.errstate(invalid="ignore"):
            M.mul(ref, z)
        out = d.kkt_apply(z)
        assert np.array_equal(np.isnan(out), np.isnan(ref)), idx
        assert np.array_equal(np.isinf(out), np.isinf(ref)), idx
        ok = np.isfinite(ref)
        assert np.allclose(out[ok], ref[ok], rtol=1e-12, atol=1e-12), idx
    ref_bits = d.kkt_apply(base)
    for wg in (8, 64, 1000):
        d.set_tuning(spmv_workgroups=wg)
        out = d.kkt_apply(base)
        rows = np.ones(d.N, dtype=bool)
        rows[[d.l - 1, d.N - 1]] = False            # the tau rows are sums of per-workgroup partial sums: grouping may change the last bit
        assert np.array_equal(out[rows], ref_bits[rows]), wg
        assert np.allclose(out[~rows], ref_bits[~rows], rtol=1e-13, atol=0), wg
    d.close()


@pytest.mark.parametrize("algname", ["DR", "GAPA"])
def test_speculation_past_cg_is_bitwise_neutral(pkg, monkeypatch, algname):
    """The kernels behind a CG solve are enqueued, gated, before the host knows the iteration count (speculation past
    the CG solve); on a misprediction -- the first solves of every run, whose counts grow -- they are no-ops and are enqueued
    again.  Arithmetic and order are unchanged, so the iterates, the CG counts and the status sums must be BITWISE those of
    the synchronous path (FOS_SPECULATE=0), on problems with and without PSD cones (the warm-start basis ping-pong is
    host-side state that a misprediction has to roll back)."""
    for prob in (pkg.workloads.small_mixed(), pkg.workloads.c4_block_sdp(nblocks=6, k=8, p=8)):
        runs = []
        for spec in ("1", "0"):
            monkeypatch.setenv("FOS_SPECULATE", spec)
            d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
            d.set_alg(pkg.DR() if algname == "DR" else pkg.GAPA(0.8, 0.5))
            d.set_iterate(None)
            cg = []
            for i in range(1, 41):
                d.step(i, 1, 10 ** 9, 1e-9)
                cg.append(d.cgiter())
            done, checked, res = d.step(41, 4, 4, 1e-9)          # a batch of four ending in a status check (i = 44)
            runs.append((d.get_iterate(), cg, (res.p, res.d, res.g, res.ctx, res.bty)))
            d.close()
        assert runs[0][1] == runs[1][1]
        assert np.array_equal(runs[0][0], runs[1][0])
        assert np.array_equal(np.array(runs[0][2]), np.array(runs[1][2]), equal_nan=True)
