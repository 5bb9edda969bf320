"""CPU test of the host logic that builds the device operator format (csr_build.cpp): the library's host emulation
walks the row blocks exactly like the kernel (ELL lane-major / LDS / LONG, run-compressed or indexed) and must
reproduce [A'vy; A vx]; every row must be covered exactly once.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

WIN_ROWS = 2016          # rows per window panel (csrc/fos_internal.hpp, WinStd)
WIN_ROWS_TALL = 4032     # ... of the tall geometry (WinTall)


def host_spmv(pkg, A, v, wg=0, waves=0):
    lib = pkg.lib.load()
    A = sp.csc_matrix(A)
    A.sort_indices()
    m, n = A.shape
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    nz = np.ascontiguousarray(A.data, dtype=np.float64)
    out = np.empty(n + m)
    stats = np.zeros(12, dtype=np.int64)
    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    pkg.lib.check(lib.fos_host_stacked_spmv(m, n, i64(colptr), i64(rowval), pkg.lib.dptr(nz), pkg.lib.dptr(v), pkg.lib.dptr(out),
                                            wg, waves, i64(stats)))
    return out, dict(zip(("blocks", "ell", "lds", "long", "run", "vals", "cols", "waves", "tiles", "slots", "deferred", "tile_vals"), stats.tolist()))


def reference(A, v):
    A = sp.csc_matrix(A)
    m, n = A.shape
    return np.concatenate([A.T @ v[n:], A @ v[:n]])


CASES = {
    "dense-long-rows": lambda rng: sp.csc_matrix(rng.standard_normal((7, 5000))),
    "dense-tall": lambda rng: sp.csc_matrix(rng.standard_normal((3000, 9))),
    "block-diag-dense": lambda rng: sp.block_diag([rng.standard_normal((40, 6)) for _ in range(30)], format="csc"),
    "sparse-random": lambda rng: sp.random(700, 900, density=0.01, format="csc", random_state=rng, data_rvs=rng.standard_normal),
    "power-law-rows": lambda rng: sp.vstack([sp.random(1, 3000, density=d, format="csr", random_state=rng, data_rvs=rng.standard_normal)
                                             for d in np.minimum(1.0, 1.0 / np.arange(1, 120))]).tocsc(),
    "identity": lambda rng: sp.identity(777, format="csc"),
    "banded": lambda rng: sp.diags([rng.standard_normal(500 - abs(k)) for k in (-2, -1, 0, 1, 2)], (-2, -1, 0, 1, 2), format="csc"),
    "empty": lambda rng: sp.csc_matrix((13, 17)),
    "one-entry": lambda rng: sp.csc_matrix(([3.5], ([4], [2])), shape=(9, 6)),
    "mixed-run-and-indexed": lambda rng: sp.vstack([sp.csc_matrix(rng.standard_normal((20, 300))),
                                                    sp.random(200, 300, density=0.05, format="csc", random_state=rng,
                                                              data_rvs=rng.standard_normal)]).tocsc(),
    # dual tiles (dense rectangles stored once): one chunk, several chunks (C > 128), ragged tails, sparse neighbours that
    # leave columns partially covered (deferred rows with an own partial), rectangles next to each other
    "tile-single-chunk": lambda rng: sp.csc_matrix(rng.standard_normal((100, 40))),
    "tile-multi-chunk": lambda rng: sp.csc_matrix(rng.standard_normal((70, 300))),
    "tile-block-diag": lambda rng: sp.block_diag([rng.standard_normal((80, 12)) for _ in range(6)], format="csc"),
    "tile-plus-sparse-rows": lambda rng: sp.vstack([sp.csc_matrix(rng.standard_normal((48, 150))),
                                                    sp.random(120, 150, density=0.05, format="csc", random_state=rng,
                                                              data_rvs=rng.standard_normal),
                                                    sp.csc_matrix(rng.standard_normal((20, 150)))]).tocsc(),
    "tile-offset-runs": lambda rng: sp.bmat([[sp.csc_matrix(rng.standard_normal((33, 17))), None],
                                             [None, sp.csc_matrix(rng.standard_normal((64, 9)))],
                                             [sp.csc_matrix(rng.standard_normal((16, 17))), sp.csc_matrix(rng.standard_normal((16, 9)))]],
                                            format="csc"),
    "row-of-2048+": lambda rng: sp.csc_matrix(rng.standard_normal((2, 2049))),
    "row-of-exactly-2048": lambda rng: sp.csc_matrix(rng.standard_normal((3, 2048))),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("waves", [0, 7168])
def test_block_format_reproduces_spmv(pkg, name, waves):
    rng = np.random.default_rng(sum(map(ord, name)))
    A = CASES[name](rng)
    m, n = A.shape
    v = rng.standard_normal(n + m)
    out, st = host_spmv(pkg, A, v, waves=waves)
    ref = reference(A, v)
    assert np.allclose(out, ref, rtol=1e-13, atol=1e-13), (name, st)
    assert st["ell"] + st["lds"] + st["long"] + st["tiles"] == st["blocks"]
    if name.startswith("tile-"):
        assert st["tiles"] > 0 and st["deferred"] > 0 and st["tile_vals"] > 0, st
        if name in ("tile-single-chunk", "tile-multi-chunk"):
            # (nearly) every entry lives in a tile, once: the stored values are about nnz, not 2 nnz
            assert st["tile_vals"] >= 0.6 * A.nnz and st["vals"] < 1.7 * A.nnz, st
    if name in ("sparse-random", "identity", "banded", "empty", "one-entry", "power-law-rows", "block-diag-dense", "dense-long-rows"):
        assert st["tiles"] == 0 and st["deferred"] == 0 and st["slots"] == 0, st     # nothing rectangular enough
    if name in ("dense-long-rows", "dense-tall", "block-diag-dense", "identity", "banded", "row-of-2048+"):
        assert st["run"] == st["blocks"] - st["lds"], st   # consecutive columns everywhere -> index-compressed (LDS blocks never are)
    if name in ("dense-long-rows", "dense-tall", "block-diag-dense"):
        assert st["cols"] < max(64, st["vals"] // 4), st   # one first-column per row instead of one index per entry
    if name == "sparse-random":
        assert st["run"] < st["blocks"]


def test_partition_sizes(pkg):
    rng = np.random.default_rng(0)
    A = sp.random(5000, 3000, density=0.004, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    v = rng.standard_normal(8000)
    for wg in (1, 8, 64, 1024, 4096):
        out, st = host_spmv(pkg, A, v, wg=wg)
        assert np.allclose(out, reference(A, v), rtol=1e-13, atol=1e-13)
        assert st["waves"] % 4 == 0 and st["waves"] >= 4


def test_malformed_csc_is_rejected(pkg):
    lib = pkg.lib.load()
    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    v = np.zeros(5)
    out = np.zeros(5)
    nz = np.array([1.0, 2.0])
    bad = [
        (np.array([0, 1, 2], dtype=np.int64), np.array([1, 2], dtype=np.int64)),      # 0-based colptr
        (np.array([1, 3, 2], dtype=np.int64), np.array([1, 2], dtype=np.int64)),      # not monotone
        (np.array([1, 2, 3], dtype=np.int64), np.array([1, 9], dtype=np.int64)),      # row index out of range
    ]
    for colptr, rowval in bad:
        rc = lib.fos_host_stacked_spmv(3, 2, i64(colptr), i64(rowval), pkg.lib.dptr(nz), pkg.lib.dptr(v), pkg.lib.dptr(out), 0, 0, None)
        assert rc == -1


def _random_structured(rng):
    """Random mix of what the builder distinguishes: dense rectangles (tile candidates: >= 16 equal rows; and smaller ones that
    must NOT become tiles), long dense rows, sparse noise, empty rows and columns."""
    m, n = int(rng.integers(20, 400)), int(rng.integers(10, 300))
    A = sp.lil_matrix((m, n))
    for _ in range(int(rng.integers(0, 5))):                        # rectangles
        r0, c0 = int(rng.integers(0, m)), int(rng.integers(0, n))
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 200))
        r1, c1 = min(m, r0 + h), min(n, c0 + w)
        A[r0:r1, c0:c1] = rng.standard_normal((r1 - r0, c1 - c0))
    if rng.random() < 0.6:                                          # sparse noise (breaks some runs, leaves partial columns)
        A = A + sp.random(m, n, density=float(rng.uniform(0.001, 0.03)), format="lil", random_state=rng, data_rvs=rng.standard_normal)
        A = A.tolil()
    if rng.random() < 0.5:
        A[int(rng.integers(0, m)), :] = 0
    if rng.random() < 0.5:
        A[:, int(rng.integers(0, n))] = 0
    A = A.tocsc()
    A.eliminate_zeros()
    return A


@pytest.mark.parametrize("seed", range(40))
def test_random_structures(pkg, seed):
    rng = np.random.default_rng(1000 + seed)
    A = _random_structured(rng)
    m, n = A.shape
    v = rng.standard_normal(n + m)
    for waves in (0, 7168):
        out, st = host_spmv(pkg, A, v, waves=waves)
        ref = reference(A, v)
        assert np.allclose(out, ref, rtol=1e-12, atol=1e-12), (seed, st)
        assert st["ell"] + st["lds"] + st["long"] + st["tiles"] == st["blocks"]
        assert st["tile_vals"] <= A.nnz


# ------------------------------------------------------------------------------------------------ window panels
def _host_spmv_mode(pkg, A, v, mode):
    import ctypes as C
    lib = pkg.lib.load()
    A = sp.csc_matrix(A)
    A.sort_indices()
    m, n = A.shape
    colptr = (A.indptr.astype(np.int64) + 1)
    rowval = (A.indices.astype(np.int64) + 1)
    nz = np.ascontiguousarray(A.data, dtype=np.float64)
    out = np.zeros(n + m)
    stats = (C.c_int64 * 16)()
    i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    pkg.lib.check(lib.fos_host_stacked_spmv_mode(m, n, i64(colptr), i64(rowval), pkg.lib.dptr(nz), pkg.lib.dptr(np.ascontiguousarray(v)),
                                                 pkg.lib.dptr(out), mode, stats))
    return out, list(stats)


@pytest.mark.parametrize("shape,density,seed", [((300, 260), 0.05, 1), ((5000, 4100), 0.004, 2), ((9000, 200), 0.03, 3), ((70, 9000), 0.02, 4),
                                                 ((4097, 4096), 0.002, 5), ((1, 1), 1.0, 6), ((3000, 2500), 0.0, 7)])
@pytest.mark.parametrize("mode", [1, 2])
def test_window_panels_host_emulation(pkg, shape, density, seed, mode):
    """Forced window-panel storage (fos_internal.hpp, WinPanel): the host walk of the panels / windows / slices -- the traversal
    the kernel performs -- reproduces S v = [A'vy; A vx]; rows without entries, empty operators, windows cut at the end of the
    vector, panels with fewer than 64 rows and rows longer than a window are all covered."""
    rng = np.random.default_rng(seed)
    m, n = shape
    A = sp.random(m, n, density=density, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    if density > 0 and m > 100:                            # a few rows much longer than the rest, one dense column
        A = A.tolil()
        A[3, :] = rng.standard_normal(n) * (rng.random(n) < 0.5)
        A[:, 1] = rng.standard_normal((m, 1))
        A = A.tocsc()
    v = rng.standard_normal(n + m)
    out, st = _host_spmv_mode(pkg, A, v, mode)                      # 1: standard geometry, 2: tall panels (one workgroup per CU)
    ref = np.concatenate([A.T @ v[n:], A @ v[:n]])
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(ref).max()))
    R = WIN_ROWS if mode == 1 else WIN_ROWS_TALL
    assert st[12] == -(-n // R) + -(-m // R) and st[0] == 0        # panels of <= R rows (the rows of A' and of A apart), no row blocks
    assert st[15] >= 2 * A.nnz                                      # stored entries (padding included)
    out0, st0 = _host_spmv_mode(pkg, A, v, 0)                       # the same operator in row blocks / tiles
    assert np.allclose(out0, ref, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(ref).max())) and st0[12] == 0


def test_window_panels_chosen_for_large_random_sparse_only(pkg):
    """fos_create's choice (window_mode -1): a C5-like random-sparse operator (10^6 stacked rows in 8 diagonal blocks) goes to window
    panels -- the tall geometry, one panel per CU, none straddling two blocks, little padding; a random-sparse operator too small for
    its panels' walks to beat the row-block form (by the builder's time model), a small one, a dense LP (dual tiles) and a banded
    (run-compressed) one keep their row-block formats."""
    rng = np.random.default_rng(11)
    m = n = 70000                                                   # 140 000 stacked rows = 70 panels of ~23 windows each: row blocks are faster
    A = sp.random(m, n, density=20.0 / n, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    v = rng.standard_normal(n + m)
    out, st = _host_spmv_mode(pkg, A, v, -1)
    ref = np.concatenate([A.T @ v[n:], A @ v[:n]])
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-11)
    assert st[12] == 0 and st[0] > 0
    out, st = _host_spmv_mode(pkg, A, v, 1)                         # forced: SELL-sigma slices, padding well under the 40-50 % of panel-wide ELL
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-11)
    assert st[12] == 2 * -(-n // WIN_ROWS) and st[15] <= 1.25 * 2 * A.nnz, (st[15], 2 * A.nnz)
    m = n = 500000                                                  # 10^6 stacked rows: 256 tall panels fill the 256 CUs in one round
    A = sp.block_diag([sp.random(62500, 62500, density=20.0 / 62500, format="csc", random_state=rng, data_rvs=rng.standard_normal)
                       for _ in range(8)], format="csc")            # (C5's structure: 8 blocks; a panel's windows span its block only)
    v = rng.standard_normal(n + m)
    out, st = _host_spmv_mode(pkg, A, v, -1)
    assert np.allclose(out, np.concatenate([A.T @ v[n:], A @ v[:n]]), rtol=1e-12, atol=1e-11)
    assert st[12] == 256 and st[0] == 0                             # 16 column ranges (8 blocks of A', 8 of A) x 16 equal panels: none straddles two blocks
    assert st[15] <= 1.25 * 2 * A.nnz, (st[15], 2 * A.nnz)
    small = sp.random(3000, 2500, density=0.01, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    assert _host_spmv_mode(pkg, small, rng.standard_normal(5500), -1)[1][12] == 0
    dense = sp.csc_matrix(rng.standard_normal((128, 96)))
    st = _host_spmv_mode(pkg, dense, rng.standard_normal(224), -1)[1]
    assert st[12] == 0 and st[8] > 0


def test_tall_dual_tiles_stack_vertically_adjacent_groups(pkg, monkeypatch):
    """Tall dual tiles (fos_internal.hpp): K groups of 64 rows below each other over the same columns become ONE block whose
    column sums leave it added up -- K times fewer partial-sum slots.  The host walk (same traversal as the kernel) must give
    S v for every K, the stored values never change, and the slot count shrinks as the stacks grow."""
    import ctypes as C
    lib = pkg.lib.load()
    rng = np.random.default_rng(0)
    A = sp.block_diag([rng.standard_normal((300, 20)) for _ in range(3)] + [rng.standard_normal((70, 150))], format="csc")
    m, n = A.shape
    S = sp.bmat([[None, A.T], [A, None]]).tocsr()
    v = rng.standard_normal(m + n)
    ref = S @ v
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    colptr, rowval, nz = (A.indptr + 1).astype(np.int64), (A.indices + 1).astype(np.int64), A.data.astype(np.float64)
    seen = {}
    for tall in (1, 2, 3, 4, 8, 16):
        monkeypatch.setenv("FOS_TILE_TALL", str(tall))
        out = np.empty(m + n)
        st = (C.c_int64 * 12)()
        rc = lib.fos_host_stacked_spmv(m, n, p(colptr, C.c_int64), p(rowval, C.c_int64), p(nz, C.c_double), p(v, C.c_double),
                                       p(out, C.c_double), 0, 0, st)
        assert rc == 0, lib.fos_last_error()
        assert np.linalg.norm(out - ref) <= 1e-13 * np.linalg.norm(ref), tall
        seen[tall] = dict(blocks=st[0], vals=st[5], tiles=st[8], slots=st[9], deferred=st[10], tile_vals=st[11])
    assert len({s["vals"] for s in seen.values()}) == 1 and len({s["tile_vals"] for s in seen.values()}) == 1
    assert seen[1]["tiles"] == 3 * 5 + 3              # 300 rows = 5 groups per block; 70 x 150: one 64-row group x three 64-column chunks (6 rows left over)
    assert seen[4]["tiles"] == 3 * 2 + 3 and seen[8]["tiles"] == 3 * 1 + 3
    slots = [seen[t]["slots"] for t in (1, 2, 3, 4, 8, 16)]
    assert all(a >= b for a, b in zip(slots, slots[1:])) and slots[0] > slots[3] > 0


# ------------------------------------------------------------------------------------------------ row-sharded builder with dual tiles
@pytest.mark.parametrize("seed", range(12))
def test_row_sharded_builder_with_dual_tiles(pkg, monkeypatch, seed):
    """a row-sharded rank stores the dense rectangles of its rows of A ONCE (dual tiles); every row of A' is then a
    deferred row whose list -- own partial + tile column sums, possibly empty -- is what the device adds up before the n-vector
    crosses the ranks.  The host emulation walks the same blocks and lists: out = S v, every row of A' deferred."""
    monkeypatch.setenv("FOS_HOST_SPMV_ROW_SHARDED", "1")
    rng = np.random.default_rng(2000 + seed)
    A = _random_structured(rng)
    m, n = A.shape
    v = rng.standard_normal(n + m)
    out, st = _host_spmv_mode(pkg, A, v, 0)
    assert np.allclose(out, reference(A, v), rtol=1e-12, atol=1e-12), seed
    monkeypatch.setenv("FOS_ROW_SHARDED_TILES", "0")            # the one-slot-per-row form of round 2
    out0, st0 = _host_spmv_mode(pkg, A, v, 0)
    assert np.allclose(out0, reference(A, v), rtol=1e-12, atol=1e-12) and st0[8] == 0 and st0[9] == n
    if st[8] > 0:                                               # tiles: more slots than rows of A', fewer stored values
        assert st[9] >= st0[9] - n and st[5] < st0[5]


def test_row_sharded_dense_operator_uses_tiles(pkg, monkeypatch):
    monkeypatch.setenv("FOS_HOST_SPMV_ROW_SHARDED", "1")
    rng = np.random.default_rng(7)
    A = sp.csc_matrix(rng.standard_normal((200, 300)))
    v = rng.standard_normal(500)
    out, st = _host_spmv_mode(pkg, A, v, 0)
    assert np.allclose(out, reference(A, v), rtol=1e-12, atol=1e-12)
    assert st[8] > 0 and st[5] <= 1.1 * A.nnz                   # A stored once (plus padding), not twice


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("seed", range(4))
def test_row_sharded_builder_with_window_panels(pkg, monkeypatch, seed, mode):
    """SURVEY 8(f2), random-sparse rows: a row-sharded rank may store its rows as window panels (both geometries).  Every row of A'
    is then a deferred row -- the panel's walk parks its share in slot j, rows without local entries are skipped -- and the host
    emulation (same traversal, same lists) gives S v."""
    monkeypatch.setenv("FOS_HOST_SPMV_ROW_SHARDED", "1")
    rng = np.random.default_rng(3000 + seed)
    m, n = int(rng.integers(50, 6000)), int(rng.integers(50, 5000))
    A = sp.random(m, n, density=float(rng.choice([0.002, 0.01, 0.05])), format="csc", random_state=rng, data_rvs=rng.standard_normal)
    v = rng.standard_normal(n + m)
    out, st = _host_spmv_mode(pkg, A, v, mode)
    assert np.allclose(out, reference(A, v), rtol=1e-12, atol=1e-12), seed
    assert st[12] > 0 and st[9] == n                               # window panels, one slot per row of A'
