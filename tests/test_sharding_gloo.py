"""
N > 1 path on CPU (gloo, world_size 2): the cone-sharding planner and the reduction points the HIP path uses
(SURVEY.md 8(e)): tau row of Q, CG inner products, GAPA's angle sums, status sums -- each all-reduced over ranks
with the replicated tau/kappa entries counted once.  Two processes each run the ORACLE on their shard with a
gloo all-reduce plugged into its reduction space; the gathered iterate must match the unsharded oracle.
(The GPU kernels themselves are covered by tests/test_gpu_parity.py; here the checker checks the protocol.)
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_rows(rank, world, port, algname, iters, q):
    """Row sharding of a NON block-diagonal problem: x replicated, A'y all-reduced as an n-vector."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import fos_oracle as orc
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    prob = pkg.workloads.small_mixed()
    shard = pkg.sharding.shard_rows(prob, world, rank)

    def allreduce(v):
        t = torch.from_numpy(np.array(v, dtype=np.float64))
        dist.all_reduce(t)
        return t.numpy()
    space = orc.RowShardedSpace(allreduce, prob.m + prob.n + 1, prob.n)
    lp = shard.problem
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    mo = orc.Model(lp.A, lp.b, lp.c, codes(lp.K1), codes(lp.K2), space=space)
    alg = {"DR": orc.DR, "GAPA": lambda: orc.GAPA(0.8, 0.5), "FISTA": orc.FISTA}[algname]()
    alg.init(mo)
    x = orc.hsde_initialvalue(mo)
    st = orc.HSDEStatus(mo, iters, 1e-6, 0, 1, S1=alg.S1)
    cg = []
    for i in range(1, iters + 1):
        st.i = i
        alg.step(x, i, st)
        cg.append(alg.S1.getcgiter())
    q.put((rank, x.copy(), cg, dict(st.last), getattr(alg, "alpha12", None)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algname", ["DR", "GAPA"])
def test_row_sharded_oracle_matches_unsharded(pkg, oracle, algname):
    """Two gloo ranks, each with the rows of half of the K1 cones of a problem whose A couples everything: the replicated
    parts (x, r, tau, kappa) stay bitwise identical on both ranks, the gathered iterate follows the unsharded oracle, the
    status sums agree."""
    import torch.multiprocessing as mp
    orc = oracle
    iters, world = 10, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rows, args=(r, world, port, algname, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, x, cg, last, a12 = q.get(timeout=240)
        got[r] = (x, cg, last, a12)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prob = pkg.workloads.small_mixed()
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    mo = orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))
    alg = {"DR": orc.DR, "GAPA": lambda: orc.GAPA(0.8, 0.5)}[algname]()
    alg.init(mo)
    x = orc.hsde_initialvalue(mo)
    st = orc.HSDEStatus(mo, iters, 1e-6, 0, 1, S1=alg.S1)
    cg = []
    for i in range(1, iters + 1):
        st.i = i
        alg.step(x, i, st)
        cg.append(alg.S1.getcgiter())
    shards = [pkg.sharding.shard_rows(prob, world, r) for r in range(world)]
    n = prob.n
    l0, l1 = shards[0].problem.m + n + 1, shards[1].problem.m + n + 1
    z0, z1 = got[0][0], got[1][0]
    # replicated parts: bitwise the same on both ranks
    assert np.array_equal(z0[:n], z1[:n]) and np.array_equal(z0[l0:l0 + n], z1[l1:l1 + n])
    assert z0[l0 - 1] == z1[l1 - 1] and z0[-1] == z1[-1]
    assert got[0][1] == got[1][1] and got[0][1][:3] == cg[:3]
    z = pkg.sharding.rows_local_to_global([z0, z1], shards)
    if got[0][1] == cg:          # (same CG stop iterations: what is left is rounding amplified by plain CG on the indefinite system)
        assert np.linalg.norm(z - x) <= 1e-3 * max(1.0, np.linalg.norm(x))
    else:
        assert np.linalg.norm(z - x) <= 0.2 * max(1.0, np.linalg.norm(x))
    for key in ("p", "d", "g", "ctx", "bty", "nb", "nc"):
        assert got[0][2][key] == pytest.approx(got[1][2][key], rel=1e-12, abs=1e-300)
    assert got[0][2]["nb"] == pytest.approx(st.last["nb"], rel=1e-12) and got[0][2]["nc"] == pytest.approx(st.last["nc"], rel=1e-12)


def _worker(rank, world, port, algname, iters, q, cg_variant="reference"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import fos_oracle as orc
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    prob = pkg.workloads.c5_mixed(nblocks=4, nb_cols=12, nonneg=5, nsoc=2, socdim=4, npsd=1, k=3, density=0.4)
    shard = pkg.sharding.shard_problem(prob, world, rank)

    def allreduce(v):
        t = torch.from_numpy(np.array(v, dtype=np.float64))
        dist.all_reduce(t)
        return t.numpy()
    space = orc.ShardedSpace(allreduce, prob.m + prob.n + 1)
    lp = shard.problem
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    mo = orc.Model(lp.A, lp.b, lp.c, codes(lp.K1), codes(lp.K2), space=space)
    alg = {"DR": orc.DR, "GAPA": lambda: orc.GAPA(0.8, 0.5), "FISTA": orc.FISTA}[algname]()
    alg.init(mo)
    alg.S1.cg_variant = cg_variant
    x = orc.hsde_initialvalue(mo)
    st = orc.HSDEStatus(mo, iters, 1e-6, 0, 1, S1=alg.S1)
    cg = []
    for i in range(1, iters + 1):
        st.i = i
        alg.step(x, i, st)
        cg.append(alg.S1.getcgiter())
    q.put((rank, x.copy(), cg, dict(st.last), getattr(alg, "alpha12", None)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algname,cg_variant", [("DR", "reference"), ("GAPA", "reference"), ("FISTA", "reference"),
                                                ("DR", "merged"), ("GAPA", "merged")])
def test_sharded_oracle_matches_unsharded(pkg, oracle, algname, cg_variant):
    """cg_variant = "merged": the merged-reduction recurrence -- what sharded HIP handles run by default -- with its inner products
    all-reduced over two gloo ranks (replicated entries counted once) against the unsharded run of the same recurrence."""
    import torch.multiprocessing as mp
    orc = oracle
    iters, world = 12, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, algname, iters, q, cg_variant)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, x, cg, last, a12 = q.get(timeout=240)
        got[r] = (x, cg, last, a12)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded reference run
    prob = pkg.workloads.c5_mixed(nblocks=4, nb_cols=12, nonneg=5, nsoc=2, socdim=4, npsd=1, k=3, density=0.4)
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    mo = orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))
    alg = {"DR": orc.DR, "GAPA": lambda: orc.GAPA(0.8, 0.5), "FISTA": orc.FISTA}[algname]()
    alg.init(mo)
    alg.S1.cg_variant = cg_variant
    x = orc.hsde_initialvalue(mo)
    st = orc.HSDEStatus(mo, iters, 1e-6, 0, 1, S1=alg.S1)
    cg = []
    for i in range(1, iters + 1):
        st.i = i
        alg.step(x, i, st)
        cg.append(alg.S1.getcgiter())
    shards = [pkg.sharding.shard_problem(prob, world, r) for r in range(world)]
    z = pkg.sharding.local_to_global([got[r][0] for r in range(world)], shards)
    # same CG stop iterations on both ranks and as the unsharded run for the first iterations (before the
    # chaotic amplification of plain CG on the indefinite system sets in -- see tests/test_gpu_parity.py)
    assert got[0][1] == got[1][1]
    assert got[0][1][:3] == cg[:3]
    # tau / kappa replicated identically
    l0 = shards[0].problem.m + shards[0].problem.n + 1
    l1 = shards[1].problem.m + shards[1].problem.n + 1
    assert got[0][0][l0 - 1] == got[1][0][l1 - 1] and got[0][0][-1] == got[1][0][-1]
    if got[0][1] == cg:
        # (12 outer iterations of the chaos envelope of tests/test_gpu_parity.py: summation order differs between the sharded and
        #  the unsharded inner products; measured 1.4e-6 for the merged recurrence, below 1e-6 for the reference one)
        assert np.linalg.norm(z - x) <= (1e-6 if cg_variant == "reference" else 1e-5) * max(1.0, np.linalg.norm(x))
    else:
        assert np.linalg.norm(z - x) <= 0.2 * max(1.0, np.linalg.norm(x))
    # status sums agree between ranks (all-reduced) and with the unsharded check on the same kind of point
    for key in ("p", "d", "g", "ctx", "bty", "nb", "nc"):
        assert got[0][2][key] == pytest.approx(got[1][2][key], rel=1e-12, abs=1e-300)
    assert got[0][2]["nb"] == pytest.approx(st.last["nb"], rel=1e-12)
    assert got[0][2]["nc"] == pytest.approx(st.last["nc"], rel=1e-12)


def test_sharded_operator_identities(pkg, oracle):
    """Single process, simulated ranks: Q apply and KKT dot products assembled from shards equal the global ones."""
    orc = oracle
    prob = pkg.workloads.c4_block_sdp(nblocks=6, k=4, p=3)
    world = 3
    shards = [pkg.sharding.shard_problem(prob, world, r) for r in range(world)]
    rng = np.random.default_rng(0)
    lg = prob.m + prob.n + 1
    v = rng.standard_normal(lg)
    Q = orc.HSDEMatrixQ(prob.A, prob.b, prob.c)
    ref = np.empty(lg)
    Q.mul(ref, v)
    # local pieces: rows of the result live where the shard lives; the tau row is the all-reduced sum
    tau_partials = []
    pieces = []
    for sh in shards:
        lp = sh.problem
        vl = np.concatenate([v[sh.cols[0]:sh.cols[1]], v[prob.n + sh.rows[0]:prob.n + sh.rows[1]], [v[-1]]])
        cap = {}
        space = orc.ShardedSpace(lambda a, cap=cap: cap.setdefault("v", []).append(a[0]) or a, lg)
        Ql = orc.HSDEMatrixQ(lp.A, lp.b, lp.c, space=space)
        out = np.empty(lp.m + lp.n + 1)
        Ql.mul(out, vl)
        tau_partials.append(sum(cap["v"]))
        pieces.append(out)
    assert -sum(tau_partials) == pytest.approx(ref[-1], rel=1e-12)
    for sh, out in zip(shards, pieces):
        nl = sh.cols[1] - sh.cols[0]
        assert np.allclose(out[:nl], ref[sh.cols[0]:sh.cols[1]], rtol=1e-12, atol=1e-14)
        assert np.allclose(out[nl:-1], ref[prob.n + sh.rows[0]:prob.n + sh.rows[1]], rtol=1e-12, atol=1e-14)


def test_plan_rejects_non_separable(pkg):
    import scipy.sparse as sp
    w = pkg.workloads
    A = sp.csc_matrix(np.ones((4, 4)))
    prob = w.ConicProblem("dense", A, np.zeros(4), np.zeros(4), [("NonNeg", 2), ("NonNeg", 2)], [("Free", 4)])
    with pytest.raises(ValueError):
        pkg.sharding.shard_problem(prob, 2, 0)
    # a column cut inside a SOC variable cone is refused
    A = sp.block_diag([np.ones((2, 2)), np.ones((2, 2))], format="csc")
    prob = w.ConicProblem("soc-col", A, np.zeros(4), np.zeros(4), [("NonNeg", 2), ("NonNeg", 2)], [("SOC", 4)])
    with pytest.raises(ValueError):
        pkg.sharding.shard_problem(prob, 2, 0)
    prob = w.ConicProblem("ok", A, np.zeros(4), np.zeros(4), [("NonNeg", 2), ("NonNeg", 2)], [("NonNeg", 4)])
    s0, s1 = (pkg.sharding.shard_problem(prob, 2, r) for r in range(2))
    assert s0.problem.K2 == [("NonNeg", 2)] and s1.cols == (2, 4)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_plan_cuts_uneven_multi_cone_blocks_at_block_boundaries(pkg, seed):
    """Blocks of several K1 / K2 cones each (every cone kind) and uneven sizes: the balanced cone split falls inside a block, so the plan moves its
    cuts to the block boundaries nearest to the balanced targets; the shards partition A exactly (no entry lost, none outside a shard's columns),
    no second-order / PSD / exponential cone of K2 is cut, and asking for more ranks than there are blocks is refused."""
    import scipy.sparse as sp
    import fuzz_parity as fz
    rng = np.random.default_rng([seed, 8081])
    blocks, K1, K2, nb = [], [], [], int(rng.integers(4, 8))
    for _ in range(nb):
        m, n = int(rng.integers(8, 70)), int(rng.integers(6, 50))
        blocks.append(sp.csc_matrix(rng.standard_normal((m, n))))            # dense blocks: exactly nb diagonal blocks
        K1 += fz.random_cones(rng, m, 1)
        K2 += fz.random_cones(rng, n, 2)
    A = sp.block_diag(blocks, format="csc")
    prob = pkg.workloads.ConicProblem("blocks", A, rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[1]), K1, K2)
    for world in range(2, nb + 1):
        shards = [pkg.sharding.shard_problem(prob, world, r) for r in range(world)]
        assert sum(s.problem.A.nnz for s in shards) == A.nnz
        assert [s.rows[0] for s in shards] + [shards[-1].rows[1]] == sorted({s.rows[0] for s in shards} | {A.shape[0]})
        assert sum(s.problem.m for s in shards) == A.shape[0] and sum(s.problem.n for s in shards) == A.shape[1]
        assert sum(len(s.problem.K1) for s in shards) == len(K1)
        assert sum(l for s in shards for _, l in s.problem.K2) == A.shape[1]
        for s in shards:
            assert all(l == dict(Free=l, Zero=l, NonNeg=l, NonPos=l).get(k, l) for k, l in s.problem.K2)
        # K2 cones that are not elementwise arrive whole
        whole = sorted((k, l) for k, l in K2 if k not in ("Free", "Zero", "NonNeg", "NonPos"))
        assert sorted((k, l) for s in shards for k, l in s.problem.K2 if k not in ("Free", "Zero", "NonNeg", "NonPos")) == whole
    with pytest.raises(ValueError):
        pkg.sharding.plan(prob, nb + 1)
