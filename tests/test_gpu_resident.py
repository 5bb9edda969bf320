"""
GPU tests of FOS_CG_RESIDENT (include/foship.h; csrc/resident.hip): conjugategradient! (conjugategradients.jl:31-55) as ONE launch per
solve, tiles and CG vectors in registers / LDS, the four sums of an iteration crossing the workgroups as self-validating words.

Checked against the oracle's restatement of the same recurrence (`conjugategradient_merged`), against the launch-per-iteration kernels of
the same recurrence (FOS_CG_MERGED_UPDATE) and against dense linear algebra: first iterations, stop rule and iteration counting, the
tolerance floor, bit-reproducibility, the affine projection's call sequence, whole outer iterations and whole solves -- with one tile
per workgroup, several wavefronts per workgroup, units split over workgroups, two tiles per wavefront and 64-step tiles.
"""
import math
import zlib
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import fos_oracle as orc

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(1e-300, np.linalg.norm(b)))


def block_op(rng, shapes):
    return sp.block_diag([sp.csc_matrix(rng.standard_normal((r, c)) / math.sqrt(r)) for r, c in shapes], format="csc")


def _ocg(fn, M, x0, rhs, tol, maxit):
    N = x0.shape[0]
    x = x0.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        it = fn(x, M, rhs, *[np.empty(N) for _ in range(4 if fn is orc.conjugategradient_merged else 3)], tol=tol, max_iters=maxit)
    return x, it


# (name, block shapes, FOS_RESIDENT_GMAX or None, what the plan must look like)
CASES = [
    ("one-tile-units", [(48, 12)] * 6, None, dict(workgroups=6, waves_per_workgroup=1, tiles_per_wave=1)),
    ("split-units", [(300, 20)] * 3, None, dict(workgroups=15, waves_per_workgroup=1)),                       # 5 tiles per unit, one per workgroup
    ("waves", [(300, 20)] * 3, "3", dict(workgroups=3, waves_per_workgroup=5, tiles_per_wave=1)),            # a unit per workgroup, five wavefronts
    ("split-and-waves", [(600, 24)] * 2 + [(200, 9)], "7", dict(workgroups=7)),                              # 10 + 10 + 4 tiles over 7 workgroups
    ("two-tiles-per-wave", [(64 * 13 + 20, 16)] * 2, "2", dict(workgroups=2, waves_per_workgroup=7, tiles_per_wave=2, max_tiles_per_workgroup=14)),
    ("three-tiles-per-wave", [(64 * 20 + 20, 16)] * 2, "2", dict(workgroups=2, waves_per_workgroup=7, tiles_per_wave=3, max_tiles_per_workgroup=21, form="registers")),
    ("wide-tiles", [(130, 50), (200, 40)], None, dict(steps_per_tile=64)),
    ("psd16-blocks", [(136, 12)] * 5, None, dict(workgroups=15)),                                             # 64 + 64 + 8 rows: the ragged last tile
    # the STREAMED form (FOS_RESIDENT_STREAM=2 asks for it where the register form would do): tiles re-read every iteration, whole units per workgroup
    ("stream-units-split-four-ways", [(700, 24)] * 4, None, dict(form="streamed", workgroups=16, max_tiles_per_workgroup=3, tiles_per_wave=3)),
    ("stream-units-split-two-ways", [(64 * 17, 30)] * 2, "4", dict(form="streamed", workgroups=4, max_tiles_per_workgroup=9)),
    ("stream-one-unit-per-workgroup", [(700, 24)] * 4, "4", dict(form="streamed", workgroups=4, max_tiles_per_workgroup=11, tiles_per_wave=3)),
    ("stream-two-units-per-workgroup", [(300, 20)] * 6, "3", dict(form="streamed", workgroups=3, max_tiles_per_workgroup=10)),
    ("stream-many-tiles", [(64 * 30 + 10, 32)] * 2, "1", dict(form="streamed", workgroups=1, max_tiles_per_workgroup=62, tiles_per_wave=9)),
    ("stream-ragged-units", [(136, 12)] * 5, "2", dict(form="streamed", workgroups=2, max_tiles_per_workgroup=9)),
    ("stream-wide-tiles", [(64 * 9 + 30, 50), (64 * 4, 40), (200, 64)], "3", dict(form="streamed", workgroups=3, steps_per_tile=64)),
    ("stream-five-tiles-per-wave", [(64 * 17, 30)] * 4, "2", dict(form="streamed", workgroups=2, max_tiles_per_workgroup=34, tiles_per_wave=5)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_resident_cg_matches_oracle_merged_update_and_dense_solve(pkg, case, monkeypatch):
    name, shapes, gmax, want = case
    if gmax:
        monkeypatch.setenv("FOS_RESIDENT_GMAX", gmax)
    if name.startswith("stream"):
        monkeypatch.setenv("FOS_RESIDENT_STREAM", "2")
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 7)
    A = block_op(rng, shapes)
    m, n = A.shape
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    d = pkg.HipHSDE(A, b, c, [("Free", m)], [("Free", n)])
    st = d.resident_stats()
    assert st["qualifies"] == 1, st
    for k, v in want.items():
        assert st[k] == v, (name, k, st)
    Q = orc.HSDEMatrixQ(A, b, c)
    M = orc.KKTMatrix(Q)
    rhs, x0 = rng.standard_normal(d.N), rng.standard_normal(d.N)
    Qd = Q.todense()
    Md = np.block([[np.eye(d.l), Qd.T], [Qd, -np.eye(d.l)]])
    xs = np.linalg.solve(Md, rhs)
    # first iterations: the oracle's merged recurrence (in the envelope of its distance to the reference recurrence: CG on the indefinite
    # system amplifies rounding) and the launch-per-iteration kernels of the same recurrence
    for k in (1, 2, 5):
        d.set_cg_variant("resident")
        assert d.cg_variant_name() == "resident"
        xk, it = d.cg_kkt(x0, rhs, 1e-300, k)
        xo, ito = _ocg(orc.conjugategradient_merged, M, x0, rhs, 1e-300, k)
        xr, _ = _ocg(orc.conjugategradient, M, x0, rhs, 1e-300, k)
        assert it == ito == k, (name, k, it)
        env = max(1e-14, relerr(xr, xo))
        assert relerr(xk, xo) <= 50 * env, (name, k, relerr(xk, xo), env)
        d.set_cg_variant("merged_update")
        xm, itm = d.cg_kkt(x0, rhs, 1e-300, k)
        assert itm == k and relerr(xk, xm) <= 50 * env, (name, k, relerr(xk, xm), env)
    d.set_cg_variant("resident")
    # tolerance floor: the dense solution, the iteration count near both host recurrences
    tol = d.N * np.finfo(float).eps
    x, it = d.cg_kkt(x0, rhs, tol, 10000)
    assert relerr(x, xs) < 1e-11, (name, relerr(x, xs))
    _, it_m = _ocg(orc.conjugategradient_merged, M, x0, rhs, tol, 10000)
    _, it_r = _ocg(orc.conjugategradient, M, x0, rhs, tol, 10000)
    assert abs(it - it_m) <= 8 + it_m // 20 and abs(it - it_r) <= 8 + it_r // 20, (name, it, it_m, it_r)
    # loose tolerance: the stop rule ||r|| <= tol (conjugategradients.jl:42)
    x, it = d.cg_kkt(x0, rhs, 1e-3, 10000)
    xr, it_r = _ocg(orc.conjugategradient, M, x0, rhs, 1e-3, 10000)
    assert abs(it - it_r) <= 6, (name, it, it_r)
    assert np.linalg.norm(Md @ x - rhs) <= 1e-3 * (1 + 1e-6), name
    assert np.linalg.norm(x - xs) <= 3 * max(np.linalg.norm(xr - xs), 1e-3), name
    # the same solve twice: bit-reproducible (fixed summation orders; workgroup order, not arrival order)
    x2, it2 = d.cg_kkt(x0, rhs, 1e-3, 10000)
    assert it2 == it and np.array_equal(x, x2), name
    # max_iters cap and the warning flag
    x3, it3 = d.cg_kkt(x0, rhs, 1e-300, 3)
    assert it3 == 3
    d.close()


def test_resident_is_refused_with_a_reason_where_the_operator_does_not_qualify(pkg):
    rng = np.random.default_rng(1)
    A = sp.random(100, 200, density=0.05, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    d = pkg.HipHSDE(A, rng.standard_normal(100), rng.standard_normal(200), [("Free", 100)], [("Free", 200)])
    assert d.resident_stats()["qualifies"] == 0
    with pytest.raises(pkg.lib.FosError) as e:
        d.set_cg_variant("resident")
    assert "qualify" in str(e.value)
    assert d.cg_variant_name() == "reference"
    d.close()


def test_prox_affine_sequence_resident(pkg):
    """prox!(y, S1::AffinePlusLinear, x) on the resident solve: call counter, tolerance schedule, warm start (affinepluslinear.jl:83-126);
    every result satisfies the reference's stopping rule and lies within 2 tol of the projection."""
    rng = np.random.default_rng(6)
    A = block_op(rng, [(70, 10), (150, 24), (64, 8)])
    m, n = A.shape
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    d = pkg.HipHSDE(A, b, c, [("Free", m)], [("Free", n)])
    d.set_cg_variant("resident")
    Q = orc.HSDEMatrixQ(A, b, c)
    S = orc.AffinePlusLinear(Q, np.zeros(d.l), np.zeros(d.l), 1, decreasing_accuracy=True)
    S.cg_variant = "merged"
    Qd = Q.todense()
    Md = np.block([[np.eye(d.l), Qd.T], [Qd, -np.eye(d.l)]])
    for call in range(1, 8):
        x = rng.standard_normal(d.N)
        tol = S.tolerance()
        y_ref = np.empty(d.N)
        S.prox(y_ref, x)
        y = d.prox_affine(x)
        assert d.prox_count() == S.i == call + 1
        rhs = np.concatenate([x[:d.l] + Qd.T @ x[d.l:], np.zeros(d.l)])
        exact = np.linalg.solve(Md, rhs)
        assert abs(d.cgiter() - S.getcgiter()) <= 4
        assert np.linalg.norm(y - y_ref) <= 2 * tol + 1e-12
        assert np.linalg.norm(Md @ y - rhs) <= tol * (1 + 1e-6) + 1e-13
        assert np.linalg.norm(y - exact) <= 2 * tol + 1e-12
    d.close()


def _block_sdp(pkg, nblocks, k, p):
    return pkg.workloads.c4_block_sdp(nblocks=nblocks, k=k, p=p)


@pytest.mark.parametrize("algname", ["DR", "GAPA", "FISTA", "Dykstra"])
def test_outer_iterations_resident_vs_merged_update(pkg, algname):
    """Whole outer iterations (CG projection + PSD cones + relaxations, speculation past the solve included) with the affine projection on the
    resident solve against the same handle on the launch-per-iteration kernels of the same recurrence: same CG counts, iterates to rounding."""
    prob = _block_sdp(pkg, 6, 16, 12)                           # 136 x 12 blocks: three tiles per unit
    mk = {"DR": pkg.DR, "GAPA": lambda: pkg.GAPA(0.8, 0.5), "FISTA": pkg.FISTA, "Dykstra": pkg.Dykstra}[algname]
    out = {}
    for variant in ("merged_update", "resident"):
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        assert d.resident_stats()["qualifies"] == 1
        d.set_cg_variant(variant)
        d.set_alg(mk())
        d.set_iterate(None)
        zs, cg = [], []
        for i in range(1, 31):
            d.step(i, 1, 10 ** 9, 1e-9)
            zs.append(d.get_iterate())
            cg.append(d.cgiter())
        out[variant] = (zs, cg)
        d.close()
    (za, ca), (zb, cb) = out["merged_update"], out["resident"]
    for i in range(30):
        dev = relerr(zb[i], za[i])
        if i == 0:
            assert dev < 1e-9, dev
        # (a CG count that differs by one moves the iterate by the tolerance of that call: affinepluslinear.jl:108-112)
        kick = 0.0 if ca[i] == cb[i] else 4 * 0.2 ** math.sqrt(i + 1)
        assert dev <= 1e-6 + kick, (algname, i, dev, ca[i], cb[i])
    assert sum(abs(a - b) for a, b in zip(ca, cb)) <= 6, (ca, cb)


def test_whole_solve_resident_matches_default(pkg):
    """solve!(model) to :Optimal with cg_variant = "resident": status, iteration count (one check interval) and solution of the default path."""
    prob = _block_sdp(pkg, 8, 16, 12)
    opts = dict(eps=1e-6, verbose=0, max_iters=6000, checki=50)
    ma = pkg.solve(prob, pkg.DR(**opts))
    mb = pkg.solve(prob, pkg.DR(cg_variant="resident", **opts))
    assert ma.status() == mb.status() == "Optimal", (ma.status(), mb.status())
    assert abs(ma.iterations - mb.iterations) <= opts["checki"], (ma.iterations, mb.iterations)
    assert np.max(np.abs(ma.getsolution() - mb.getsolution())) < 1e-4
    assert abs(ma.getobjval() - mb.getobjval()) < 1e-5 * (1 + abs(ma.getobjval()))


def test_c4_shard_one_steady_state_iteration_vs_oracle_resident(pkg, oracle, fullsize):
    """What one of eight ranks holds of C4 (64 blocks = 2112 tiles on 256 workgroups of nine wavefronts: the configuration the resident solve was
    built for), outer iteration 201 from the device's state at iteration 200 against the oracle's step (solverwrapper.jl:23-29, gap.jl:61-80), 1e-9."""
    from test_gpu_fullsize import _same_step_vs_oracle
    prob = fullsize("C4shard64")
    _same_step_vs_oracle(pkg, prob, pkg.DR(), oracle.DR(), 200, 1e-9, cg_variant="resident")


@pytest.mark.parametrize("seed", range(36))
def test_resident_forms_on_random_block_structures(pkg, seed, monkeypatch):
    """Random block-diagonal operators (1 .. 12 blocks of 20 .. 1400 rows and 8 .. 64 columns, random workgroup caps, the streamed form asked for in
    half of the cases): wherever the planner says the operator qualifies -- tiles in registers with one to three tiles per wavefront, units split
    over workgroups, the streamed form with whole or split units, 32- and 64-step tiles -- the resident solve and the launch-per-iteration kernels
    of the same recurrence give the same iterates (fixed iteration counts) and the same solves (tolerance-limited), and the result solves M x = rhs."""
    rng = np.random.default_rng(9000 + seed)
    nb = int(rng.integers(1, 13))
    wide = rng.random() < 0.3
    shapes = [(int(rng.integers(20, 1400)), int(rng.integers(8, 65 if wide else 33))) for _ in range(nb)]
    gmax = rng.choice(["", "1", "2", "3", "5", "8", "40"])
    if gmax:
        monkeypatch.setenv("FOS_RESIDENT_GMAX", str(gmax))
    if rng.random() < 0.5:
        monkeypatch.setenv("FOS_RESIDENT_STREAM", "2")
    A = block_op(rng, shapes)
    m, n = A.shape
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    d = pkg.HipHSDE(A, b, c, [("Free", m)], [("Free", n)])
    st = d.resident_stats()
    if not st["qualifies"]:                                # (e.g. more than 64 columns or 73 tiles would fall to one workgroup): refused with the reason, never run
        with pytest.raises(pkg.lib.FosError):
            d.set_cg_variant("resident")
        assert d.cg_variant_name() != "resident"
        d.close()
        return
    M = orc.KKTMatrix(orc.HSDEMatrixQ(A, b, c))
    rhs, x0 = rng.standard_normal(d.N), rng.standard_normal(d.N)
    for k in (1, 4):
        d.set_cg_variant("resident")
        xr, itr = d.cg_kkt(x0, rhs, 1e-300, k)
        d.set_cg_variant("merged_update")
        xm, itm = d.cg_kkt(x0, rhs, 1e-300, k)
        xo, _ = _ocg(orc.conjugategradient_merged, M, x0, rhs, 1e-300, k)
        xref, _ = _ocg(orc.conjugategradient, M, x0, rhs, 1e-300, k)
        env = max(1e-14, relerr(xref, xo))
        assert itr == itm == k and relerr(xr, xm) <= 50 * env and relerr(xr, xo) <= 50 * env, (seed, st, k, relerr(xr, xm), relerr(xr, xo), env)
    tol = 1e-9
    d.set_cg_variant("resident")
    xr, itr = d.cg_kkt(x0, rhs, tol, 5000)
    d.set_cg_variant("merged_update")
    xm, itm = d.cg_kkt(x0, rhs, tol, 5000)
    assert abs(itr - itm) <= 2 + itm // 20, (seed, st, itr, itm)
    y = np.empty(d.N)
    M.mul(y, xr)
    assert np.linalg.norm(y - rhs) <= 3 * tol, (seed, st)
    d.close()
