"""
The complete `bench.py --gpus N` flow on ONE GPU (the driver launches it on 2/4/8 GPUs at round end; it must not rot):
two ranks started by torch.distributed.run share the device, host-side coordination over gloo (`FOS_BENCH_BACKEND=gloo`:
RCCL refuses several ranks per device), scalar sums through the peer mailboxes -- per-rank shard generation, mailbox export /
open / self test, barrier + max-over-ranks timing, the strong-scaling headline plus the weak-scaling extra, ONE JSON line.
The same problem run by one rank must report the same residuals.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


def _check_diagnostics(out, nranks):
    """What a failure analysis of an N-rank run needs is IN the line (bench.py `config.diagnostics`): the transport, why the ones in front of it were passed
    over, one exchange's cost on it per rank, who can reach whose memory, every rank's CG variant and resident-solve plan."""
    dg = out["config"]["diagnostics"]
    assert dg["transport"] == out["config"]["transport"] and isinstance(dg["passed_over"], dict)
    assert len(dg["per_rank"]) == nranks and sorted(r["rank"] for r in dg["per_rank"]) == list(range(nranks))
    for r in dg["per_rank"]:
        assert isinstance(r["exchange_us"], float) and 0.0 < r["exchange_us"] < 5e4, r
        assert r["cg_variant"] == out["config"]["cg_variant"] and "qualifies" in r["resident"] and "all_ranks_qualify" in r["resident"]
    assert "hipDeviceCanAccessPeer" in dg


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_bench_two_ranks_on_one_gpu_matches_single_rank():
    env = dict(os.environ, FOS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--small"]
    r2 = subprocess.run(cmd2, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    out2 = _last_json(r2.stdout)
    cmd1 = [sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "3", "--small", "--no-cpu-baseline"]
    r1 = subprocess.run(cmd1, cwd=str(ROOT), env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-3000:]
    out1 = _last_json(r1.stdout)

    # contract fields of the N = 2 line
    assert out2["n_gpus"] == 2 and out2["steps"] == 6 and out2["warmup"] == 3 and out2["scaling"] == "strong"
    assert out2["unit"] == "iterations/s" and out2["value"] > 0 and out2["higher_is_better"] is True and out2["dtype"] == "f64"
    assert out2["warmup_effective"] >= out2["warmup"] and out2["warmup_effective"] == out1["warmup_effective"]   # same global l
    assert "peer mailboxes" in out2["config"]["parallelism"] and out2["config"]["peer_fallback_reason"] is None
    per_rank = out2["config"]["all_ranks_ms_per_step"]                   # a straggler would show here, not only in the maximum
    assert len(per_rank) == 2 and max(per_rank) == pytest.approx(out2["ms_per_step"], rel=1e-3) and min(per_rank) > 0
    assert out1["config"]["peer_fallback_reason"] is None
    assert out2["roofline"]["all_ranks"]["achieved_all_ranks"] >= out2["roofline"]["achieved"]
    assert "cpu_baseline" not in out2                       # N = 1 only
    _check_diagnostics(out2, 2)
    # each rank holds half of the blocks
    assert 2 * out2["config"]["local_m"] == out1["config"]["local_m"] and 2 * out2["config"]["local_nnz"] == out1["config"]["local_nnz"]
    # the weak-scaling extra: twice the blocks in total, the single-rank shard size per rank
    w = out2["weak_scaling"]
    assert w["scaling"] == "weak" and w["value"] > 0 and w["config"]["local_m"] == out1["config"]["local_m"]
    # same problem, same iteration count, same residuals (sums cross the ranks in a different order: rounding only, then the
    # CG-chaos envelope of DESIGN.md 4 over a few hundred iterations)
    ra, rb = out1["config"]["residuals_after_run"], out2["config"]["residuals_after_run"]
    assert ra["iteration"] == rb["iteration"]
    for k in ("p", "d", "g"):
        assert rb[k] == pytest.approx(ra[k], rel=1e-4), k
    assert out2["config"]["cg_iters_per_step"] == pytest.approx(out1["config"]["cg_iters_per_step"], abs=1.5)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how a user -- or a driver without torchrun -- calls it): the
    script starts torch.distributed.run itself, as a child process and before it touches the GPU, passes the ranks' single JSON
    line through and exits with the launcher's code.  (gloo coordination: two ranks share the test box's one GPU.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FOS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--small", "--no-weak-extra"]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0
    # (sharded handles on the mailboxes: the resident solve -- one launch per CG solve -- when every rank's shard qualifies, as the reduced block SDP's
    #  do; otherwise the merged-reduction recurrence, two launches per iteration)
    assert (out["config"]["cg_variant"], out["config"]["cg_launches_per_iteration"], out["config"]["cg_launches_per_solve"]) in (("resident", 0, 1), ("merged_update", 2, None))
    # a failing child must fail the call: an unknown workload makes every rank exit non-zero
    bad = subprocess.run(cmd + ["--workload", "nope"], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


@pytest.mark.parametrize("inject,expect", [("peer_warmup_fail", "host-pinned"), ("mailbox_warmup_fail", "RCCL"), ("peer_open_fail", "host-pinned")])
def test_bench_falls_back_through_the_transports(inject, expect):
    """The order of transports for the scalar sums is device mailboxes (HIP IPC) -> host-pinned mailboxes -> in-stream RCCL all-reduce, each
    decided collectively.  A mailbox self test can pass and the first real exchanges still fail (a transport that has never run between
    two devices): every rank votes after the warm-up, and the job runs again, in the same processes, over the next transport.  One rank in
    a RCCL group of its own (`FOS_FORCE_DIST=1`), the failures injected: in the warm-up on the device mailboxes, on both kinds of
    mailboxes, and at the opening of the device mailboxes (no peer access)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FOS_BENCH_BACKEND")}
    env.update(FOS_FORCE_DIST="1", FOS_BENCH_INJECT=inject, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--small", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert expect in out["config"]["parallelism"] and out["value"] > 0
    why = out["config"]["peer_fallback_reason"]
    if inject == "peer_open_fail":
        assert "peer:" in why and "injected: no peer access" in why
    else:
        assert "timed out on the peer mailboxes" in why and "warm-up failed on the peer mailboxes" in r.stderr
    if inject == "mailbox_warmup_fail":
        assert "timed out on the host mailboxes" in why and "warm-up failed on the host mailboxes" in r.stderr
    # with only the device mailboxes allowed the same failure must end the run
    bad = subprocess.run(cmd, cwd=str(ROOT), env=dict(env, FOS_REDUCTION="peer"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_bench_retries_a_transport_without_the_resident_solve_before_leaving_it():
    """A warm-up that fails while the CG solves run resident (one launch per solve, the exchange inside the kernel) is retried on the SAME mailboxes
    with a launch group per CG iteration before the transport is given up: every rank takes that branch after the collective vote, and the line
    says so (config.diagnostics.resident_fallback)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FOS_BENCH_BACKEND")}
    env.update(FOS_FORCE_DIST="1", FOS_BENCH_INJECT="resident_warmup_fail", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--small", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["value"] > 0 and out["config"]["cg_variant"] == "merged_update" and out["config"]["transport"] == "peer"
    assert out["config"]["peer_fallback_reason"] is None
    assert "resident CG solve" in out["config"]["diagnostics"]["resident_fallback"]
    assert "warm-up failed on the peer mailboxes" in r.stderr


@pytest.mark.parametrize("transport", ["peer", "host"])
@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_bench_ranks_on_one_gpu(nranks, transport):
    """The rank counts the driver launches (2, 4, 8), here sharing the one GPU: cone shards of 1/N of the blocks, every CG iteration's
    sums through N mailboxes -- in device memory mapped through HIP IPC (`peer`) or in ONE pinned host segment every rank registers
    (`host`: fos_peer_open_host; workgroup 0 polls over PCIe and republishes in a local relay).  Same iteration count and residuals
    as one rank on the same problem (timing is meaningless here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FOS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", FOS_REDUCTION=transport)
    base = [sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "3", "--small"]
    rn = subprocess.run(base + ["--gpus", str(nranks), "--no-weak-extra"], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert rn.returncode == 0, rn.stdout[-3000:] + rn.stderr[-3000:]
    outn = _last_json(rn.stdout)
    r1 = subprocess.run(base + ["--no-cpu-baseline"], cwd=str(ROOT), env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-3000:]
    out1 = _last_json(r1.stdout)
    assert outn["n_gpus"] == nranks and ("peer mailboxes" if transport == "peer" else "host-pinned mailboxes") in outn["config"]["parallelism"]
    assert outn["config"]["peer_fallback_reason"] is None and len(outn["config"]["all_ranks_ms_per_step"]) == nranks
    _check_diagnostics(outn, nranks)
    assert nranks * outn["config"]["local_m"] == out1["config"]["local_m"]
    ra, rb = out1["config"]["residuals_after_run"], outn["config"]["residuals_after_run"]
    assert ra["iteration"] == rb["iteration"]
    for k in ("p", "d", "g"):
        assert rb[k] == pytest.approx(ra[k], rel=1e-4), k
    assert outn["config"]["cg_iters_per_step"] == pytest.approx(out1["config"]["cg_iters_per_step"], abs=1.5)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("foship-")]            # rank 0 unlinked the segment
    # the extra line of the same job under DR(direct = true): the block form on every rank, its scalar sums through the same mailboxes, no CG
    dt = outn["direct_true"]
    assert "failed" not in dt, dt
    assert dt["form"] == "block" and dt["value"] > 0 and dt["sweeps_per_step"] == 3 and transport in dt["parallelism"].replace("host-pinned", "host")
    if nranks == 2:
        rd = subprocess.run(base + ["--no-cpu-baseline", "--direct"], cwd=str(ROOT), env=dict(os.environ), capture_output=True, text=True, timeout=600)
        assert rd.returncode == 0, rd.stdout[-3000:] + rd.stderr[-3000:]
        da, db = _last_json(rd.stdout)["config"]["residuals_after_run"], dt["residuals_after_run"]
        assert da["iteration"] == db["iteration"]
        for k in ("p", "d", "g"):                                                        # (no inexact CG in the loop: the order of the sums is all that differs)
            assert db[k] == pytest.approx(da[k], rel=1e-7), k


@pytest.mark.parametrize("inject,reason", [("peer_open_fail_rank1", "injected: no peer access"), ("peer_map_fail_rank1", "injected: IPC mapping failed")])
def test_one_rank_without_peer_access_takes_every_rank_to_the_host_segment(inject, reason):
    """The transports are chosen COLLECTIVELY: when ONE rank cannot use the device mailboxes (injected on rank 1 of two: at the peer-access query, or at the
    IPC mapping -- where rank 0's own mapping SUCCEEDS), both ranks pass them over and meet on the host-pinned ones -- a rank whose own steps succeeded does not
    go on alone (it would wait for its peer's words until the time-out)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FOS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", FOS_BENCH_INJECT=inject)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--small", "--no-weak-extra"]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert "host-pinned mailboxes" in out["config"]["parallelism"] and out["value"] > 0
    why = out["config"]["peer_fallback_reason"]
    assert ("rank 1: open: " + reason if "map" in inject else "rank 1: " + reason) in why and "rank 0:" not in why
