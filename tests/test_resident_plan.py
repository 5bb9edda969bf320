"""CPU tests of FOS_CG_RESIDENT's host side (csrc/csr_build.cpp: build_resident_plan, host_resident_cg): which operators qualify, how the
tiles of a unit are dealt to workgroups, and the arithmetic of the resident solve -- the plan walked on the host the way the kernel
(csrc/resident.hip) walks it -- against the oracle's restatement of conjugategradient! (conjugategradients.jl:31-55) in the
merged-reduction form.  No GPU needed."""
import ctypes as C
import warnings

import numpy as np
import scipy.sparse as sp

import fos_oracle as orc

KEYS = ("qualifies", "workgroups", "waves_per_workgroup", "tiles_per_wave", "units", "max_tiles_per_workgroup", "steps_per_tile", "all")


def host_resident(pkg, A, b, c, gmax, x=None, rhs=None, tol=0.0, max_iters=1):
    lib = pkg.lib.load()
    A = sp.csc_matrix(A)
    A.sort_indices()
    m, n = A.shape
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    nz = np.ascontiguousarray(A.data, dtype=np.float64)
    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    st = np.zeros(8, dtype=np.int64)
    it = np.zeros(1, dtype=np.int64)
    xo = None if x is None else np.ascontiguousarray(x, dtype=np.float64).copy()
    rc = lib.fos_host_resident_cg(m, n, i64(colptr), i64(rowval), pkg.lib.dptr(nz), pkg.lib.dptr(np.ascontiguousarray(b, dtype=np.float64)),
                                  pkg.lib.dptr(np.ascontiguousarray(c, dtype=np.float64)), gmax,
                                  None if xo is None else pkg.lib.dptr(xo), None if rhs is None else pkg.lib.dptr(np.ascontiguousarray(rhs, dtype=np.float64)),
                                  float(tol), int(max_iters), i64(it), i64(st))
    return rc, dict(zip(KEYS, st.tolist())), xo, int(it[0])


def block_sdp(rng, nblocks, rows, cols):
    return sp.block_diag([sp.csc_matrix(rng.standard_normal((rows, cols)) / np.sqrt(rows)) for _ in range(nblocks)], format="csc")


def test_which_operators_qualify(pkg):
    rng = np.random.default_rng(0)
    b = lambda A: rng.standard_normal(A.shape[0])
    c = lambda A: rng.standard_normal(A.shape[1])
    # block-diagonal dense blocks = units; tiles of 64 rows
    A = block_sdp(rng, 8, 136, 12)                     # PSD(16): 64 + 64 + 8 rows -- the ragged last group of a stack is a tile too
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 256)
    assert rc == 0 and st["qualifies"] == 1 and st["units"] == 8 and st["workgroups"] == 24
    A = block_sdp(rng, 8, 12, 12)                      # blocks too small for a tile: rows of A outside the tiles
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 256)
    assert rc == 0 and st["qualifies"] == 0
    A = block_sdp(rng, 8, 150, 12)                     # 64 + 64 + 22 rows
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 256)
    assert rc == 0 and st["qualifies"] == 1 and st["units"] == 8 and st["steps_per_tile"] == 32
    assert st["workgroups"] == 24 and st["max_tiles_per_workgroup"] == 1 and st["waves_per_workgroup"] == 1 and st["tiles_per_wave"] == 1
    # fewer workgroups than tiles: several tiles per workgroup; than units: does not qualify
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 8)
    assert st["qualifies"] == 1 and st["workgroups"] == 8 and st["max_tiles_per_workgroup"] == 3 and st["waves_per_workgroup"] == 3
    # fewer workgroups than units: the STREAMED form -- whole consecutive units per workgroup, tiles re-read every iteration (tiles_per_wave < 0)
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 7)
    assert st["qualifies"] == 1 and st["tiles_per_wave"] < 0 and st["workgroups"] == 4 and st["max_tiles_per_workgroup"] == 6
    # wide blocks (33..64 columns): 64-step tiles
    A = block_sdp(rng, 4, 200, 40)
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 256)
    assert st["qualifies"] == 1 and st["steps_per_tile"] == 64 and st["units"] == 4
    # more than 64 columns per block: rows spread over column chunks; sparse operators; a dense LP: no
    for A in (block_sdp(rng, 2, 200, 80), sp.random(300, 200, density=0.05, format="csc", random_state=rng), sp.csc_matrix(rng.standard_normal((100, 300)))):
        rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 256)
        assert rc == 0 and st["qualifies"] == 0
    # an empty column: columns of A outside the units
    A = sp.hstack([block_sdp(rng, 3, 100, 10), sp.csc_matrix((300, 1))]).tocsc()
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 256)
    assert st["qualifies"] == 0
    # asking for the solve on an operator that does not qualify is an error with a reason
    x = rng.standard_normal(2 * (A.shape[0] + A.shape[1] + 1))
    rc, st, _, _ = host_resident(pkg, A, b(A), c(A), 256, x, x)
    assert rc != 0 and b"qualify" in pkg.lib.load().fos_last_error()


def test_c4_shard_plans(pkg):
    """What the ranks of a multi-GPU C4 run would hold (structure only: 2080 x 32 blocks of ones are enough for the planner)."""
    blk = sp.csc_matrix(np.ones((2080, 32)))
    for nblocks, want in ((64, dict(qualifies=1, workgroups=256, max_tiles_per_workgroup=9, waves_per_workgroup=9, tiles_per_wave=1, units=64)),
                          (32, dict(qualifies=1, workgroups=224, max_tiles_per_workgroup=5, waves_per_workgroup=5, units=32)),
                          (128, dict(qualifies=1, workgroups=256, max_tiles_per_workgroup=17, waves_per_workgroup=7, tiles_per_wave=3, units=128)),   # three tiles per wavefront; beyond 21 per workgroup the
                          (256, dict(qualifies=1, workgroups=256, max_tiles_per_workgroup=33, tiles_per_wave=-5, units=256)),   # registers are full: the streamed form
                          (512, dict(qualifies=1, workgroups=256, max_tiles_per_workgroup=66, tiles_per_wave=-9, units=512))):  # the whole of C4: two units per CU
        A = sp.block_diag([blk] * nblocks, format="csc")
        rc, st, _, _ = host_resident(pkg, A, np.zeros(A.shape[0]), np.zeros(A.shape[1]), 256)
        assert rc == 0
        for k, v in want.items():
            assert st[k] == v, (nblocks, k, st)


def _ocg(M, x0, rhs, tol, maxit, fn=None):
    fn = fn or orc.conjugategradient_merged
    N = x0.shape[0]
    x = x0.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        it = fn(x, M, rhs, *[np.empty(N) for _ in range(4 if fn is orc.conjugategradient_merged else 3)], tol=tol, max_iters=maxit)
    return x, it


def test_host_walk_of_the_plan_matches_the_oracle(pkg, monkeypatch):
    """conjugategradient! over KKTMatrix(HSDEMatrixQ(A, b, c)) by the plan's walk: the first iterations to rounding against the oracle's merged
    recurrence, the stop rule and the iteration count at a loose and at the floor tolerance -- with one, several and split units per workgroup,
    and in the streamed form (several whole units per workgroup: gmax below the number of units, or FOS_RESIDENT_STREAM=2)."""
    rng = np.random.default_rng(3)
    for name, A, gmaxes in (("one-tile-units", block_sdp(rng, 6, 48, 12), (256, 6, 2)),
                            ("stream-asked-for", block_sdp(rng, 5, 136, 12), (256, 2)),        # 256: every unit split over its three tiles
                            ("stream-split", block_sdp(rng, 2, 600, 24), (4, 5)),
                            ("stream-wide", block_sdp(rng, 2, 300, 50), (256, 2)),
                            ("split-units", block_sdp(rng, 3, 300, 20), (256, 7, 3)),
                            ("wide", block_sdp(rng, 2, 130, 50), (256, 2)),
                            ("uneven", sp.block_diag([sp.csc_matrix(rng.standard_normal((r, cc)) / 8) for r, cc in ((70, 8), (200, 31), (40, 16), (64, 12))], format="csc"), (256, 5))):
        m, n = A.shape
        b, c = rng.standard_normal(m), rng.standard_normal(n)
        Q = orc.HSDEMatrixQ(sp.csc_matrix(A), b, c)
        M = orc.KKTMatrix(Q)
        N = 2 * (m + n + 1)
        rhs, x0 = rng.standard_normal(N), rng.standard_normal(N)
        if name.startswith("stream"):
            monkeypatch.setenv("FOS_RESIDENT_STREAM", "2")
        else:
            monkeypatch.delenv("FOS_RESIDENT_STREAM", raising=False)
        for gmax in gmaxes:
            for k in (1, 2, 7):
                rc, st, x, it = host_resident(pkg, A, b, c, gmax, x0, rhs, 1e-300, k)
                assert rc == 0 and st["qualifies"] == 1, (name, gmax, pkg.lib.load().fos_last_error())
                assert (st["tiles_per_wave"] < 0) == (name.startswith("stream") or (name == "one-tile-units" and gmax == 2)), (name, gmax, st)
                xo, ito = _ocg(M, x0, rhs, 1e-300, k)
                assert it == ito == k, (name, gmax, k, it, ito)
                # (CG on the indefinite KKT system from a random start amplifies rounding: the yardstick is how far the oracle's OWN two
                #  recurrences -- the reference's and the merged one -- are from each other after k iterations)
                xr, _ = _ocg(M, x0, rhs, 1e-300, k, orc.conjugategradient)
                env = np.linalg.norm(xr - xo) / np.linalg.norm(xo)
                assert np.linalg.norm(x - xo) <= 50 * max(1e-14, env) * np.linalg.norm(xo), (name, gmax, k, env)
            for tol in (1e-3, N * np.finfo(float).eps):
                rc, st, x, it = host_resident(pkg, A, b, c, gmax, x0, rhs, tol, 10000)
                xo, ito = _ocg(M, x0, rhs, tol, 10000)
                assert abs(it - ito) <= 4 + ito // 20, (name, gmax, tol, it, ito)
                y = np.empty(N)
                M.mul(y, x)
                assert np.linalg.norm(y - rhs) <= max(tol, 1e-11) * 3, (name, gmax, tol)
