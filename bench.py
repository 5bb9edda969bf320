#!/usr/bin/env python3
"""
bench.py -- GAP/DR outer iterations per second + achieved HBM GB/s of the CG SpMV (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4|C2|C3|C5|...]

A "step" is ONE outer iteration of the solver (solverwrapper.jl:23-29: affine projection by warm-started CG over
the KKT operator -> cone projection -> relaxations) on a synthetic problem already resident in HBM.  The metric is
defined on the STEADY STATE: past the outer iteration at which the CG tolerance schedule
max(0.2^sqrt(i), l*eps) (affinepluslinear.jl:108-112) reaches its floor -- i = 190 for C4.  Whatever --warmup says, the
untimed warm-up therefore runs at least to that iteration (`warmup_effective` in the output; `warmup` echoes the flag);
exactly --steps iterations are then timed.  The timed region excludes set-up and the every-`checki` status.

Default workload (all N): C4, BASELINE.json configs[3] "Block-diagonal SDP, 512 PSD blocks of size 64x64, DR,
cone-sharded across 1/2/4/8 MI355X via RCCL" -- the configuration the metric ("... at 1/2/4/8 GPUs") and the
north-star targets (1e6-variable problem, >=3.5x at 8 GPUs on the block-PSD workload) are quoted on; it fits one
GPU.  N > 1 is STRONG scaling of that one problem by default: rank g owns blocks [512 g/N, 512 (g+1)/N) and only scalars
cross GPUs (peer mailboxes over xGMI, or an in-stream RCCL all-reduce); `--scaling weak` gives every rank 512 blocks of a
512 N-block problem instead, and an N > 1 run in the default mode also times that weak-scaling problem and reports it under
`weak_scaling` in the same line.  `--workload C2` runs configs[1] (dense 5000x10000 LP) on one GPU.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` for the kernel with the largest measured share of the
step (HIP events on the solver's stream around sampled launches of the KKT sweep, the batched PSD projection and the CG
vector updates), `roofline_psd`, `time_shares` and, at N = 1, `cpu_baseline` (the C port of the oracle restatement timed on
one host core and on all of them, for a bounded sample of steady-state outer iterations from the GPU's state).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured float4 copy
FP64_PEAK_TFLOPS = 78.6        # MI355X fp64 vector = matrix peak (AMD data sheet; the micro-architecture guide lists no fp64 MFMA rate)
EPS = 2.220446049250313e-16


def tolerance_floor_iteration(l_global):
    """First outer iteration whose CG tolerance max(0.2^sqrt(i), l*eps) is the floor l*eps (affinepluslinear.jl:108-112)."""
    import math
    i = max(1, int(math.ceil((math.log(l_global * EPS) / math.log(0.2)) ** 2)))
    while 0.2 ** math.sqrt(i) > l_global * EPS:
        i += 1
    return i



def decode_psd_records(rec):
    """Per-matrix records of the last batched PSD(64) projection (fos_psd_stats): >= 100 = accepted by the refinement kernel,
    100 + 1000 [the extrapolated start was the one accepted] + 16 x rotations + iterations (both starts counted; further decimal digits
    only with FOS_PSD_DEBUG_REC); below 100 = sweeps of the Jacobi kernel.  Returns (refined mask, iterations, rotations, extrapolated mask)."""
    import numpy as np
    rec = np.asarray(rec, dtype=np.int64) % 10000
    refined = rec >= 100
    body = np.where(refined, (rec % 1000) - 100, 0)
    return refined, body % 16, body // 16, refined & (rec >= 1100)


def build_problem(pkg, workload, nranks, rank, small, weak=False, c4_scale=None):
    w = pkg.workloads
    if workload == "C4":
        nb = 64 if small else 512
        if weak:
            nb *= nranks                                  # every rank owns 512 blocks of a 512 N-block problem
        lo, hi = (nb * rank) // nranks, (nb * (rank + 1)) // nranks
        prob = w.c4_block_sdp(nblocks=nb, block_range=(lo, hi), scale=c4_scale)
        desc = "C4 block-diagonal SDP, %d PSD(64) blocks, 32 free vars/block, DR" % nb
        if c4_scale is not None:
            desc += " [A_j / %g instead of A_j / 32]" % c4_scale
        alg = pkg.DR()
        glob = dict(m=nb * 2080, n=nb * 32, nnz=nb * 2080 * 32)
    elif workload == "C5":
        nb = 8
        if nranks > nb or nb % nranks:
            raise SystemExit("C5 has 8 blocks: --gpus must divide 8")
        lo, hi = (nb * rank) // nranks, (nb * (rank + 1)) // nranks
        kw = dict(nb_cols=6250, nonneg=3125, nsoc=25, npsd=1) if small else {}
        prob = w.c5_mixed(nblocks=nb, block_range=(lo, hi), **kw)
        desc = "C5 mixed cones (NonNeg+SOC+PSD), 8 blocks, FISTA"
        alg = pkg.FISTA()
        glob = None
    elif workload == "C2":
        if nranks != 1:
            raise SystemExit("C2 (dense LP) is not block separable: single GPU only")
        prob = w.c2_lp(m=500, n=1000) if small else w.c2_lp()
        desc = "C2 random LP, dense A %dx%d stored sparse, Zero/NonNeg cones, DR" % prob.A.shape
        alg = pkg.DR()
        glob = None
    elif workload == "C3":
        if nranks != 1:
            raise SystemExit("C3 is generated unsharded: single GPU only")
        prob = w.c3_socp(n=2000, ncones=100) if small else w.c3_socp()
        desc = "C3 sparse SOCP, %d x SOC(50), GAPA" % len(prob.K1)
        alg = pkg.GAPA()
        glob = None
    else:
        raise SystemExit("unknown workload %s" % workload)
    return prob, alg, desc, glob


class PeerTransportFailed(RuntimeError):
    """raised by every rank together (after a collective vote) when the warm-up over the peer mailboxes ended in an error"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--small", action="store_true", help="reduced sizes (smoke / CI); not a valid benchmark number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-raw-instance", action="store_true",
                    help="N = 1, C4: skip the additional timing of the instance as SURVEY 8(d) writes it (A_j not divided by 32)")
    ap.add_argument("--spmv-wg", type=int, default=0)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1, C4: strong = the 512-block problem split over the ranks (default, the metric's definition); "
                         "weak = 512 blocks per rank")
    ap.add_argument("--no-weak-extra", action="store_true", help="N > 1: skip the additional weak-scaling measurement")
    ap.add_argument("--direct", action="store_true",
                    help="direct = true (HSDE.jl:12-15): S1 = IndAffine([Q -I], 0), the exact affine projection without CG -- on C4 the block form, "
                         "three KKT sweeps per projection (fos_enable_direct); a secondary line, the headline stays the CG path")
    ap.add_argument("--no-recurrence-extras", action="store_true",
                    help="N = 1: skip the additional timings of the same job on the launch-per-iteration recurrences (reference_recurrence_n1, merged_recurrence_n1)")
    ap.add_argument("--no-direct-extra", action="store_true", help="N = 1, C4: skip the additional timing of DR(direct=true)")
    ap.add_argument("--c4-scale", type=float, default=None,
                    help="C4: divide the random symmetric constraint matrices by this instead of 32 (1 = the raw, badly conditioned "
                         "instance: ~4x the CG iterations per outer iteration; not the headline configuration)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as CHILD processes of torch.distributed.run,
        # BEFORE this process touches the GPU (it never does), pass their output through and exit with the launcher's code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        proc = subprocess.run(cmd, env=env)
        raise SystemExit(proc.returncode)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # FOS_BENCH_BACKEND=gloo (testing only): host-side coordination over gloo, scalar sums through the peer mailboxes only, ranks
    # mapped round-robin onto the visible GPUs -- lets the whole N > 1 path run with several ranks on ONE GPU, which RCCL refuses
    host_gloo = os.environ.get("FOS_BENCH_BACKEND") == "gloo"
    if host_gloo:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    tdev = "cpu" if host_gloo else "cuda"
    dist = None
    reduction = "in-stream RCCL all-reduce"
    force_dist = os.environ.get("FOS_FORCE_DIST") == "1"      # exercise the distributed path with one rank (testing)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "RANK" not in os.environ:          # FOS_FORCE_DIST=1 without a launcher: a one-rank group of our own
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
        if host_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    pkg = ge.load_package()
    BIG = 10 ** 12

    def agree(ok):
        """every collective here is executed by every rank in the same order, whatever fails locally"""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def run_case(weak):
        """Build the (shard of the) problem, warm up to the steady state, time exactly --steps outer iterations."""
        reduction = "in-stream RCCL all-reduce"
        transport = "rccl"
        peer_reason = None             # why the transports in front of the chosen one were passed over (None: the first was taken, or there is one rank)
        t0 = time.time()
        # FOS_BENCH_SHARD="k/N" (with FOS_FORCE_DIST=1, one process): rank k's shard of an N-rank run on this GPU, in the sharded code path -- what an N-GPU
        # run's ranks would each step (no hop between devices is paid: a projection, labelled as such in config.workload)
        emu = os.environ.get("FOS_BENCH_SHARD") if world == 1 else None
        if emu:
            ek, en = (int(v) for v in emu.split("/"))
            prob, alg, desc, glob = build_problem(pkg, args.workload, en, ek, args.small, weak=weak, c4_scale=args.c4_scale)
            desc += " [shard %d of %d, alone on this GPU]" % (ek, en)
        else:
            prob, alg, desc, glob = build_problem(pkg, args.workload, world, rank, args.small, weak=weak, c4_scale=args.c4_scale)
        t_gen = time.time() - t0
        t0 = time.time()
        dev = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2, device=local_rank)
        if dist is not None:
            if not host_gloo:
                idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(pkg.HipHSDE.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                dev.comm_init(world, rank, bytes(idt.cpu().numpy().tobytes()))
            # scalar sums, in this order (FOS_REDUCTION=auto, the default): device mailboxes mapped through HIP IPC ("peer": one xGMI write latency
            # per exchange), mailboxes in pinned host memory ("host": needs no peer access between the devices -- two PCIe latencies), the in-stream
            # RCCL all-reduce set up above ("rccl").  Each is decided COLLECTIVELY by its self test; the reason a transport was passed over is
            # recorded.  FOS_REDUCTION=peer|host|rccl forces one; FOS_REDUCTION_SKIP (set by the warm-up fallback below) lists transports that
            # passed their self test but failed in the warm-up.
            want = os.environ.get("FOS_REDUCTION", "auto")
            order = {"auto": ["peer", "host", "rccl"], "peer": ["peer"], "host": ["host"], "rccl": ["rccl"]}[want]
            if host_gloo:
                order = [t for t in order if t != "rccl"]          # (no RCCL communicator in this mode)
            skipped = {t: os.environ.get("FOS_REDUCTION_SKIP_NOTE_" + t.upper(), "failed in the warm-up")
                       for t in os.environ.get("FOS_REDUCTION_SKIP", "").split(",") if t}
            reasons = {}

            def gather_reason(why):
                """every rank learns every rank's reason (the votes only say that SOME rank failed)"""
                whys = [None] * world
                dist.all_gather_object(whys, why)
                return "; ".join("rank %d: %s" % (g, w) for g, w in enumerate(whys) if w) or "a peer failed"

            def selftest_and_enable():
                why = None
                dist.barrier()
                try:
                    ok = dev.peer_selftest(64)
                    if not ok:
                        why = "self test: wrong sums"
                except Exception as exc:
                    print("rank %d: mailbox self test raised (%s)" % (rank, exc), file=sys.stderr, flush=True)
                    ok, why = False, "self test: %s" % exc
                if agree(ok):
                    dev.peer_enable(True)
                    return True, None
                return False, why

            def try_peer():
                # first contact between DIFFERENT devices: can this device map its peers' memory at all?  (the IPC mapping below can
                # succeed where loads and stores over the link do not)  Asked of the runtime, not assumed; a "no" names the pair.
                why = None
                inj_open = os.environ.get("FOS_BENCH_INJECT", "")
                if inj_open == "peer_open_fail" or (inj_open == "peer_open_fail_rank1" and rank == 1):          # tests only
                    why = "injected: no peer access"
                try:
                    if not host_gloo and why is None:
                        for g in range(torch.cuda.device_count()):
                            if g != local_rank and g < world and not torch.cuda.can_device_access_peer(local_rank, g):
                                why = "hipDeviceCanAccessPeer(%d, %d) = 0" % (local_rank, g)
                                break
                except Exception as exc:
                    why = "peer-access query failed: %s" % exc
                try:
                    mine = dev.peer_export() if why is None else None
                except Exception as exc:                      # no uncached allocation / IPC export on this device
                    print("rank %d: peer mailbox export failed (%s)" % (rank, exc), file=sys.stderr, flush=True)
                    mine, why = None, "export: %s" % exc
                handles = [None] * world
                dist.all_gather_object(handles, mine)
                ok = all(h is not None for h in handles)
                if ok:
                    try:
                        if inj_open == "peer_map_fail_rank1" and rank == 1:          # tests only: the mapping fails on ONE rank
                            raise RuntimeError("injected: IPC mapping failed")
                        dev.peer_open(world, rank, handles, timeout_s=60.0)
                        dev.sync()
                    except Exception as exc:                  # IPC mapping not available between these devices
                        print("rank %d: peer mailboxes unavailable (%s)" % (rank, exc), file=sys.stderr, flush=True)
                        ok, why = False, "open: %s" % exc
                if agree(ok):
                    ok, why2 = selftest_and_enable()
                    why = why or why2
                else:
                    ok = False                                # (a rank whose own steps succeeded must not go on alone)
                return ok, why

            def try_host():
                name = [None]
                if rank == 0:
                    name[0] = "/foship-%d-%d" % (os.getpid(), time.time_ns() & 0xFFFFFFFF)
                dist.broadcast_object_list(name, src=0)
                why = None
                try:
                    dev.peer_open_host(world, rank, name[0], timeout_s=60.0)
                    dev.sync()
                    ok = True
                except Exception as exc:                      # no shm / the runtime cannot register it
                    print("rank %d: host-pinned mailboxes unavailable (%s)" % (rank, exc), file=sys.stderr, flush=True)
                    ok, why = False, "open: %s" % exc
                if agree(ok):
                    ok, why2 = selftest_and_enable()
                    why = why or why2
                else:
                    ok = False
                return ok, why

            chosen = None
            for tr in order:
                if tr in skipped:
                    reasons[tr] = skipped[tr]
                    continue
                if tr == "rccl":
                    chosen = tr
                    break
                ok, why = (try_peer if tr == "peer" else try_host)()
                if ok:
                    chosen = tr
                    break
                reasons[tr] = gather_reason(why)
                dev.peer_close()                              # the next transport opens its own mailboxes on the same handle
            if chosen is None:
                raise SystemExit("no usable transport for the scalar sums (FOS_REDUCTION=%s): %s" % (want, reasons))
            transport = chosen
            reduction = {"peer": "peer mailboxes over xGMI (HIP IPC)", "host": "host-pinned mailboxes (POSIX shm + hipHostRegister, PCIe)",
                         "rccl": "in-stream RCCL all-reduce"}[chosen]
            peer_reason = "; ".join("%s: %s" % (t, w) for t, w in reasons.items()) or None
            if want == "rccl":
                peer_reason = "FOS_REDUCTION=rccl"
        # ---- what a failure analysis of an N-rank run needs, measured here so that the line carries it (no reference equivalent: SURVEY 2.2 C1):
        # the cost of one exchange on the chosen transport per rank, who can reach whose memory, the resident solve's plan per rank
        diagnostics = None
        if dist is not None:
            try:
                ex_us = dev.exchange_bench(200)
            except Exception as exc:  # noqa: BLE001
                ex_us = "failed: %s" % exc
            try:
                nd = torch.cuda.device_count()
                acc = [[1 if (a == b or torch.cuda.can_device_access_peer(a, b)) else 0 for b in range(nd)] for a in range(nd)] if not host_gloo else None
            except Exception as exc:  # noqa: BLE001
                acc = "query failed: %s" % exc
            try:
                rst = dev.resident_stats()
            except Exception as exc:  # noqa: BLE001
                rst = {"failed": repr(exc)}
            mine_diag = {"rank": rank, "device": local_rank, "exchange_us": ex_us, "resident": rst, "cg_variant": dev.cg_variant_name()}
            every_diag = [None] * world
            dist.all_gather_object(every_diag, mine_diag)
            diagnostics = {"transport": transport, "per_rank": every_diag, "hipDeviceCanAccessPeer": acc,
                           "exchange_us_note": "one exchange of four doubles on the chosen transport, 200 back to back in stream (mailboxes: inside one launch)",
                           "passed_over": reasons, "resident_fallback": os.environ.get("FOS_BENCH_RESIDENT_NOTE")}
        if args.spmv_wg:
            dev.set_tuning(spmv_workgroups=args.spmv_wg)
        direct_form = None
        if args.direct:
            dev.enable_direct(prob.A)                         # (on sharded handles: the block form on every rank or an error -- its three scalar sums per projection ride the chosen transport)
            direct_form = dev.direct_mode()
        dev.set_alg(alg)
        dev.set_iterate(None)
        t_setup = time.time() - t0

        def barrier():
            dev.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        # ---- warm-up (untimed): at least to the iteration where the CG tolerance reaches its floor l*eps
        nm = torch.tensor([float(prob.m + prob.n)], dtype=torch.float64, device=tdev)
        if dist is not None:
            dist.all_reduce(nm, op=dist.ReduceOp.SUM)
        l_global = int(nm.item()) + 1
        i_floor = tolerance_floor_iteration(l_global)
        warm = max(args.warmup, i_floor)
        it = 0
        if warm > 0:
            try:
                inj = os.environ.get("FOS_BENCH_INJECT", "")
                if dist is not None and inj in ("peer_warmup_fail", "host_warmup_fail", "mailbox_warmup_fail") and \
                        transport in {"peer_warmup_fail": ("peer",), "host_warmup_fail": ("host",), "mailbox_warmup_fail": ("peer", "host")}[inj]:   # tests only
                    raise pkg.lib.FosError(-5, "injected: mailbox exchange timed out")
                if dist is not None and inj == "resident_warmup_fail" and transport != "rccl" and dev.cg_variant_name() == "resident":      # tests only
                    raise pkg.lib.FosError(-5, "injected: a record of the resident solve never arrived")
                done, _, _ = dev.step(1, warm, BIG, 1e-8)
                dev.sync()
                ok = True
            except pkg.lib.FosError as exc:                   # e.g. FOS_ECOMM: a mailbox word that never arrived
                if dist is None or transport == "rccl":
                    raise
                print("rank %d: warm-up failed on the %s mailboxes (%s)" % (rank, transport, exc), file=sys.stderr, flush=True)
                done, ok = 0, False
            if dist is not None and transport != "rccl" and not agree(ok):
                # a rank that waits for a silent peer runs into the mailbox time-out too, so every rank arrives here
                was_resident = dev.cg_variant_name() == "resident"
                dev.close()
                raise PeerTransportFailed(transport, was_resident)
            it += done
        # ---- timed: exactly K outer iterations
        # HIP events around every PROF_PERIOD-th launch group of each class: an event pair per launch costs ~5 % of a C4 step, every
        # 4th 2 %, every 16th 1 % (536 / 542 / 547 it/s at periods 4 / 16 / 64); short runs keep the dense sampling for the sample count
        PROF_PERIOD = int(os.environ.get("FOS_BENCH_PROF_PERIOD", "16" if args.steps >= 20 else "4"))
        dev.psd_debug(True, 0)
        dev.profile(PROF_PERIOD)
        dev.profile_read_classes()
        cg0 = dev.cg_total()
        barrier()
        t1 = time.perf_counter()
        done, _, _ = dev.step(it + 1, args.steps, BIG, 1e-8)
        dev.sync()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        it += done
        elapsed = t2 - t1
        per_rank_ms = None
        if dist is not None:
            mine_t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            every = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(every, mine_t)                    # a straggler shows in the line, not only in the maximum
            per_rank_ms = [round(1e3 * float(t.item()) / max(1, args.steps), 4) for t in every]
            tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        cls = dev.profile_read_classes()
        cg_timed = dev.cg_total() - cg0
        if direct_form in ("block", "dense"):
            cg_timed = (3 if direct_form == "block" else 2) * args.steps          # no CG: the sweeps of the exact projection, per outer iteration
        sweeps = dev.psd_sweeps()
        dev.profile(False)
        barrier()

        # one check on the current point (not timed): residuals for the record
        _, _, chk = dev.step(it + 1, 1, 1, 1e-8)
        it += 1

        ms_per_step = 1e3 * elapsed / max(1, args.steps)
        value = args.steps / elapsed
        launches, kms = cls["kkt"]
        avg_kernel_ms = kms / max(1, launches)
        # ---- bytes of ONE KKT sweep launch
        # (1) what the device format really streams (model, checked against the PMC counters within 1 %): stored
        #     values + stored column indices + block descriptors + partial-sum slots written + the vector's rows read + [c;b]
        #     read + the result written; the gathered columns of the vector are served from L2.
        ost = dev.operator_stats()
        nmr = prob.m + prob.n
        stored_bytes = (8.0 * ost["vals"] + 4.0 * ost["cols"] + 48.0 * ost["blocks"] + 16.0 * ost["slots"]
                        + 16.0 * nmr + 8.0 * nmr + 16.0 * nmr)
        # (2) the same from rocprofv3 PMC passes of the latest round's kernels (tools/gpu_profile_r03.sh -> profiles/), when committed
        traffic, traffic_src, trace_us = None, None, None
        try:
            cands = sorted(Path(ROOT / "profiles").glob("r0*_kkt_traffic.json"))          # the latest round's PMC passes
            # (a FOS_BENCH_SHARD run sweeps a SHARD: the full-size counter bytes do not describe its launches -- the stored-format model does)
            if cands and not args.small and world == 1 and not emu:
                ent = json.load(open(cands[-1])).get(args.workload, {})
                traffic, traffic_src = ent.get("traffic_bytes"), ent.get("source")
                trace_us = ent.get("kernel_trace_us_in_solve") or ent.get("kernel_trace_us_sweep_alone")
        except Exception:
            traffic = None
        # (3) SURVEY 8(d)'s model of a fused dual-RHS apply on a CSR operator (A and A' each streamed once)
        survey_bytes = 24.0 * prob.nnz + 4.0 * (nmr + 2) + 32.0 * nmr
        # (4) the least any storage of this operator could move: every non-zero of A once, 8 B, no indices (what dual tiles reach for dense
        #     blocks), the vector in and out (16 B each per row: two right-hand sides), [c; b]
        minimal_bytes = 8.0 * prob.nnz + 40.0 * nmr
        moved = traffic if traffic else stored_bytes
        achieved = moved / (avg_kernel_ms * 1e-3) / 1e9 if launches else 0.0
        survey_gbs = survey_bytes / (avg_kernel_ms * 1e-3) / 1e9 if launches else 0.0
        # ---- PSD projection: flops of the algorithm as run (sweeps counted by the kernel)
        npsd_mats = int(sweeps.size)
        psd_n, psd_ms = cls["psd"]
        avg_psd_ms = psd_ms / max(1, psd_n)
        # records of the last launch: >= 100 = refinement by matrix products (100 + 1000 [extrapolated start accepted] + 16 rotations +
        # iterations), otherwise the sweeps of the Jacobi kernel
        rec = np.asarray(sweeps, dtype=np.int64)
        refined, rf_it, rf_rot, rf_extrap = decode_psd_records(rec)
        mean_sweeps = float(rec[~refined].mean()) if (~refined).any() else 0.0
        kk = 64
        # per sweep: 64 steps x 32 pairs x (one dot product + the rotation of 2 columns); the wave kernel carries the column norms
        # along instead of recomputing them, so this is fewer flops per sweep than the 3-dot-product workgroup kernel did
        flop_sweep = kk * (kk // 2) * (2 * kk + 2 * kk * 3)
        flop_fixed = 2 * kk ** 3 + 10 * 16 * 16 * kk * 2 + 2 * kk * kk         # warm-start product, rebuild (10 tiles), weights
        # refinement: three 64^3 products per iteration, two for the Newton-Schulz step, 10 of 16 tiles for P
        flop_refine = float(np.sum((3.0 * rf_it + 2.0 + 10.0 / 16.0) * 2.0 * kk ** 3))
        psd_flops = int((~refined).sum()) * (mean_sweeps * flop_sweep + flop_fixed) + flop_refine
        psd_tflops = psd_flops / (avg_psd_ms * 1e-3) / 1e12 if psd_n and avg_psd_ms > 0 else 0.0
        vec_n, vec_ms = cls["cgvec"]
        avg_vec_ms = vec_ms / max(1, vec_n)
        oth_n, oth_ms = cls["other"]                          # (sampled outer iterations, ms of every other launch group in them)
        res_n, res_ms = cls.get("resident", (0, 0.0))         # FOS_CG_RESIDENT: whole CG solves run as one launch each (sampled solves, ms)
        avg_res_ms = res_ms / max(1, res_n)
        RES_KEY = "cg_resident_solve (ONE launch per CG solve: tiles and CG vectors in registers / LDS)"
        raw = {
            "kkt_sweep": avg_kernel_ms * cg_timed / (1e3 * elapsed) if elapsed > 0 else 0.0,
            "cg_vector_updates": avg_vec_ms * cg_timed / (1e3 * elapsed) if elapsed > 0 else 0.0,
            "psd_projection": avg_psd_ms * args.steps / (1e3 * elapsed) if elapsed > 0 and psd_n else 0.0,
            RES_KEY: avg_res_ms * args.steps / (1e3 * elapsed) if elapsed > 0 and res_n else 0.0,      # (one solve per outer iteration)
            "other_launches (CG start sweep + r0 kernel, relaxations, elementwise/SOC/Exp cones, last pass)":
                (oth_ms / oth_n) * args.steps / (1e3 * elapsed) if elapsed > 0 and oth_n else 0.0,
        }
        # every class is measured between two events on the solver's stream, which also see part of the gap in front of the group: the
        # classes can add up to slightly more than the wall time.  Shares are therefore normalised by max(1, their sum); what is left of
        # the wall time is gaps between launch groups and the host's polls.
        raw_sum = sum(raw.values())
        norm = max(1.0, raw_sum)
        shares = {k: round(v / norm, 4) for k, v in raw.items()}
        shares["gaps_and_host_poll"] = round(max(0.0, 1.0 - raw_sum / norm), 4)
        shares["event_brackets_sum_before_normalisation"] = round(raw_sum, 4)
        dominant = max((k for k in ("kkt_sweep", "cg_vector_updates", "psd_projection", RES_KEY)), key=lambda k: shares[k] or 0.0)
        if not res_n:
            shares.pop(RES_KEY, None)
        roof_res = None
        if res_n:
            # the resident solve reads the operator ONCE per solve (tiles -> registers) and the vectors x, rhs, v once, writes x once; between
            # that nothing moves through HBM: its time is iterations x (sweep out of registers + one exchange of four doubles), i.e. latency
            rplan = dev.resident_stats()
            its = cg_timed / max(1, args.steps)
            if rplan.get("form") == "streamed":
                # STREAMED form: every sweep of a solve (its + 1 are needed: the start residual and one per iteration; the kernel walks one more when its
                # early exchange of r.r did not catch the stop) reads the stored tiles once and [c; b] of its rows (8 B per row of A); r, w AND x stay in registers,
                # p, s in LDS, the columns in the communication wavefront; per solve v, rhs, x come in and x goes out once -- so the launch is HBM
                # bound, and it is priced on the bound no stored format beats: 8 B per non-zero, per sweep
                sweep_bytes = 8.0 * ost["vals"] + 48.0 * ost["blocks"] + 8.0 * prob.m
                res_bytes = sweep_bytes * (its + 1) + 4 * 16.0 * nmr
                stream_traffic, stream_traffic_src = None, None
                try:
                    cands = sorted(Path(ROOT / "profiles").glob("r0*_stream_traffic.json"))
                    if cands and not os.environ.get("FOS_BENCH_SHARD") and dist is None:
                        tj = json.load(open(cands[-1])).get(args.workload)
                        if tj:
                            stream_traffic = tj["bytes_per_sweep"] * (its + 1)
                            stream_traffic_src = "HBM-side bytes per sweep from separate --pmc passes of the same kernel (%s) x (iterations + 1) sweeps" % tj["source"]
                except Exception:  # noqa: BLE001
                    pass
                roof_res = {
                    "bound": "hbm",
                    "kernel": "cg_stream_kernel (FOS_CG_RESIDENT, streamed form): a whole CG solve (conjugategradients.jl:31-55) as one persistent launch -- per "
                              "iteration ONE pass over the dual tiles does the KKT sweep, its reductions AND the vector updates of the iteration (r, w in "
                              "registers, p, s in LDS, x in HBM), the four sums cross the workgroups as self-validating words",
                    "achieved": round(res_bytes / (avg_res_ms * 1e-3) / 1e9, 1) if avg_res_ms > 0 else 0.0,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(res_bytes / (avg_res_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_res_ms > 0 else 0.0,
                    "frac_time_base": "HIP events on the solver's stream around every %d-th solve of the timed region; the launch's whole duration -- entry, the "
                                      "start sweep, every iteration's exchange and row update, exit -- not only its streaming phases" % PROF_PERIOD,
                    "bytes_per_launch": res_bytes, "bytes_per_sweep": sweep_bytes,
                    "bytes_basis": "ALGORITHMIC: (8 B x stored tile values + 48 B x tiles + 8 B x rows of A) x (CG iterations + 1) sweeps per solve + v, rhs, x "
                                   "in and x out once -- the launch-per-iteration form moved the tiles + 160 B per row and element of x in three launches "
                                   "per iteration, so a lower fraction here sits beside FEWER bytes and a shorter iteration (`us_per_cg_iteration`)",
                    "algorithmic_bytes_per_launch": res_bytes, "traffic": stream_traffic,
                    "traffic_basis": stream_traffic_src,
                    "traffic_over_algorithmic": (round(stream_traffic / res_bytes, 4) if stream_traffic else None),
                    "avg_kernel_ms": round(avg_res_ms, 5), "launches_timed": res_n,
                    "cg_iterations_per_solve": round(its, 2),
                    "us_per_cg_iteration": round(1e3 * avg_res_ms / max(1e-9, its + 1), 3),
                    "us_per_cg_iteration_note": "launch duration / (iterations + 1 sweeps); in-kernel stamps (profiles/r06_stream_stamps.txt): sweep 43 us "
                                                "(the tiles at 6.4 TB/s), exchange 7, row update 1",
                    "resident_plan": rplan,
                    "kernel_share_of_step": shares.get(RES_KEY),
                }
            else:
                res_bytes = 8.0 * ost["vals"] + 48.0 * ost["blocks"] + 16.0 * nmr * 4 + 8.0 * nmr
                roof_res = {
                    "bound": "hbm",
                    "kernel": "cg_resident_kernel: a whole CG solve (conjugategradients.jl:31-55) as one persistent launch -- dual tiles and CG vectors "
                              "held in registers / LDS, four sums per iteration exchanged between workgroups as self-validating words",
                    "achieved": round(res_bytes / (avg_res_ms * 1e-3) / 1e9, 1) if avg_res_ms > 0 else 0.0,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(res_bytes / (avg_res_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_res_ms > 0 else 0.0,
                    "bytes_per_launch": res_bytes,
                    "bytes_basis": "the operator's stored tiles ONCE per solve + x, rhs, v in, x out; nothing else crosses HBM between the first and the last iteration",
                    "note": "NOT bandwidth bound by construction: the launch's time is iterations x (register-resident sweep + one exchange); the fraction of the "
                            "HBM peak says how little HBM matters, the figure of merit is `us_per_cg_iteration` against the launch-per-iteration form's",
                    "avg_kernel_ms": round(avg_res_ms, 5), "launches_timed": res_n,
                    "cg_iterations_per_solve": round(its, 2),
                    "us_per_cg_iteration": round(1e3 * avg_res_ms / max(1e-9, its + 1), 3),
                    "resident_plan": rplan,
                    "kernel_share_of_step": shares.get(RES_KEY),
                    "traffic": None,
                }
        # N > 1: every rank sweeps its own shard at the same time; the job's SpMV rate is the sum over ranks
        agg = None
        if dist is not None:
            tt = torch.tensor([roof_res["achieved"] if roof_res else achieved], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            agg = {"achieved_all_ranks": round(float(tt[0]), 1),
                   "frac_of_n_gpus_peak": round(float(tt[0]) / (HBM_PEAK_GBS * world), 4),
                   "note": "this rank's KKT sweep (or resident solve) on its shard is what `achieved` prices; the sum over the ranks is the job's rate"}
        if roof_res is not None:
            roof_res["all_ranks"] = agg
        kname = "kkt2_kernel"
        if ost.get("win_panels", 0):                        # window panels; more than 2016 rows per panel = the tall geometry (fos_internal.hpp, WinTall)
            tall = nmr / ost["win_panels"] > 2016
            kname = "kkt2_win_kernel (window panels, %s)" % ("4032-row panels, one 1024-thread workgroup per CU" if tall else "2016-row panels, two 512-thread workgroups per CU")
        roof_kkt = {
            "bound": "hbm",
            "kernel": kname + ": fused dual-RHS KKT sweep of a CG iteration (4 reference SpMV sweeps + epilogue + 3 reductions)",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_time_base": "HIP events on the solver's stream around every %d-th sweep of the timed region: the bracket includes the dependent-launch gap "
                              "in front of the kernel (~5 us on C4); `frac_kernel_trace` prices the same bytes on the kernel's own duration" % PROF_PERIOD,
            "frac_kernel_trace": (round(traffic / (trace_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if traffic and trace_us else None),
            "kernel_trace_us": trace_us,
            "kernel_trace_source": ("rocprofv3 --kernel-trace of the sweep, committed with the PMC passes (%s)" % traffic_src) if trace_us else None,
            "frac_of_measured_copy_6290": round(achieved / 6290.0, 4),
            "bytes_per_launch": moved,
            "bytes_basis": ("HBM-side bytes per launch from rocprofv3 PMC passes of this round's kernels (%s)" % traffic_src) if traffic
                           else ("bytes the stored format streams per launch (model; agrees with the PMC counters to 1 % on C4/C2, DESIGN.md 5)"
                                 + ("; a shard of the workload: no counter pass exists for it" if emu else "")),
            "minimal_bytes_per_launch": minimal_bytes,
            "frac_on_minimal_bytes": round(minimal_bytes / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if launches else None,
            "minimal_bytes_basis": "8 B x nnz(A) (every non-zero once, no indices) + 40 B x (m + n) (vector in, result out, [c; b]): a bound no stored format beats",
            "traffic": traffic,
            "stored_bytes_per_launch_model": stored_bytes,
            "survey_model_bytes_per_launch": survey_bytes,
            "algorithmic_bytes_per_launch": survey_bytes if not ost["tiles"] else stored_bytes,
            "algorithmic_bytes_basis": ("SURVEY 8(d): B_kkt,min = 24 nnz + 4 (m+n+2) + 32 (m+n)" if not ost["tiles"] else
                                        "dense rectangles stored once (dual tiles, 8 B per non-zero for both products): the stored-format model; "
                                        "SURVEY 8(d)'s two-stream CSR formula is `survey_model_bytes_per_launch`"),
            "traffic_over_algorithmic": (round(traffic / (survey_bytes if not ost["tiles"] else stored_bytes), 4) if traffic else None),
            "frac_on_algorithmic_bytes": round((survey_bytes if not ost["tiles"] else stored_bytes) / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if launches else None,
            "frac_vs_survey_model": round(survey_gbs / HBM_PEAK_GBS, 4),
            "survey_model_note": "SURVEY 8(d) counts A and A' as two CSR streams (24 B per non-zero of A per dual-RHS apply); dual tiles "
                                 "store a dense rectangle of A once for both products (8 B), so this ratio can exceed 1 -- it is NOT a "
                                 "roofline fraction, `frac` is" if ost["tiles"] else None,
            "operator_format": ost,
            "all_ranks": agg,
            "avg_kernel_ms": round(avg_kernel_ms, 5),
            "launches_timed": launches,
            "launches_in_region": cg_timed,
            "event_sampling": "KKT sweeps whose iteration number (counted over all CG solves) is a multiple of %d" % PROF_PERIOD,
            "kernel_share_of_step": shares["kkt_sweep"],
        }
        roof_psd = None
        if psd_n:
            n_ref = int(refined.sum())
            if n_ref * 2 >= npsd_mats:
                roof_psd = {
                    "bound": "mfma",
                    "kernel": "psd64_refine_kernel: batched order-64 PSD projection by eigenvector refinement -- three 64^3 products per "
                              "iteration on v_mfma_f64_16x16x4 from a basis extrapolated over the last two projections, one Newton-Schulz "
                              "step, P = V max(D, 0) V'; one workgroup of four wavefronts per matrix; Jacobi kernel for flagged matrices",
                    "achieved": round(psd_tflops, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(psd_tflops / FP64_PEAK_TFLOPS, 4),
                    "flops_per_launch": psd_flops,
                    "flops_model": "refined matrices x (3 iterations + 2 + 10/16) x 2 x 64^3 (iterations counted by the kernel) + Jacobi matrices "
                                   "x (sweeps x 64 steps x 32 pairs x 512 + 2 x 64^3 + ...)",
                    "matrices_per_launch": npsd_mats, "refined": n_ref, "left_to_jacobi": npsd_mats - n_ref,
                    "iterations_histogram_last_launch": {int(k): int(v) for k, v in zip(*np.unique(rf_it[refined], return_counts=True))},
                    "matrices_with_rotations": int((rf_rot > 0).sum()), "extrapolated_starts_accepted": int(rf_extrap.sum()),
                    "avg_kernel_ms": round(avg_psd_ms, 5), "launches_timed": psd_n,
                    "note": "the launch lasts as long as its slowest matrix (the largest iteration count); a 64^3 product is 64 MFMAs of "
                            "64 cycles per wavefront, measured 2.5-3.2 us of a 10 us iteration (tools/psd_time.py, profiles/r04_psd_time.json)",
                    "kernel_share_of_step": shares["psd_projection"],
                }
            else:
                roof_psd = {
                    "bound": "valu-issue",
                    "kernel": "psd64_wave_kernel: batched order-64 PSD projection, one wavefront per matrix, one-sided Jacobi (odd-even ordering) in registers with warm start; warm-start product and rebuild on v_mfma_f64_16x16x4 (12 % of its time)",
                    "achieved": round(psd_tflops, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(psd_tflops / FP64_PEAK_TFLOPS, 4),
                    "flops_per_launch": psd_flops,
                    "flops_model": "matrices x (sweeps x 64 steps x 32 pairs x 512 + 2 x 64^3 + 10 x 16 x 16 x 64 x 2 + 2 x 64^2); sweeps counted by the kernel",
                    "matrices_per_launch": npsd_mats, "mean_jacobi_sweeps": round(mean_sweeps, 3),
                    "sweeps_histogram_last_launch": {int(k): int(v) for k, v in zip(*np.unique(rec, return_counts=True))},
                    "avg_kernel_ms": round(avg_psd_ms, 5), "launches_timed": psd_n,
                    "note": "fp64 VALU issue bound, not MFMA bound: 88 % of the kernel is the Jacobi sweeps on the vector ALUs (one wavefront per "
                            "SIMD issues an fp64 op every 7.5 cycles and a DPP move every 9, measured); the two dense contractions on the "
                            "matrix cores are 12 % of its time (profiles/r02_psd_phases_wave.json)",
                    "kernel_share_of_step": shares["psd_projection"],
                }
        out = {
            "metric": "GAP/DR outer iterations/sec (+ achieved HBM GB/s of the CG SpMV in `roofline`)",
            "value": round(value, 4),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_effective": warm,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (numpy default_rng, seeds in firstordersolvers.jl_amd/workloads.py)",
            "config": {
                "workload": desc + (" [SMALL]" if args.small else ""),
                "solver": type(alg).__name__,
                "regime": "steady state: CG tolerance at its floor l*eps = %.3g from outer iteration %d on (affinepluslinear.jl:108-112); "
                          "timed iterations %d..%d" % (l_global * EPS, i_floor, warm + 1, warm + args.steps),
                "local_m": int(prob.m), "local_n": int(prob.n), "local_nnz": int(prob.nnz),
                "cg_iters_per_step": round(cg_timed / max(1, args.steps), 2) if direct_form not in ("block", "dense") else 0,
                "direct": direct_form,
                "sweeps_per_step": round(cg_timed / max(1, args.steps), 2),
                "cg_variant": dev.cg_variant_name(),
                "cg_launches_per_iteration": {"reference": 3, "resident": 0}.get(dev.cg_variant_name(), 2),
                "cg_launches_per_solve": 1 if dev.cg_variant_name() == "resident" else None,
                "parallelism": "cone-sharded x%d (scalar sums: %s)" % (world, reduction) if dist is not None else "single GPU",
                "peer_fallback_reason": peer_reason, "transport": transport if dist is not None else None,
                "all_ranks_ms_per_step": per_rank_ms,
                "diagnostics": diagnostics,
                "residuals_after_run": {"p": chk.p, "d": chk.d, "g": chk.g, "iteration": it},
                "setup_s": round(t_setup, 2), "generate_s": round(t_gen, 2),
                "instance_note": ("data scaled as BASELINE.md 3 records (||b|| = ||c|| = 10, A_j / 32): a well-conditioned instance, 17 CG "
                                  "iterations per outer iteration at the floor; unscaled data needs ~75" if args.c4_scale is None else
                                  "NOT the headline instance: A_j / %g" % args.c4_scale) if args.workload == "C4" else None,
            },
            "dominant_kernel": dominant,
            "time_shares": shares,
            "roofline": roof_res if (dominant == RES_KEY and roof_res is not None) else (roof_kkt if dominant != "psd_projection" or roof_psd is None else roof_psd),
            "roofline_kkt": (roof_kkt if dominant == "psd_projection" and roof_psd is not None else "= roofline") if not res_n else
                            "no KKT sweep launches: the CG solves ran resident (roofline_resident)",
            "roofline_psd": roof_psd,
            "roofline_resident": roof_res,
        }
        return out, dev, prob, alg, it

    weak_main = args.scaling == "weak"
    for _attempt in range(6):
        try:
            out, dev, prob, alg, it = run_case(weak_main)
            break
        except PeerTransportFailed as exc:
            # the self test passed but the first real exchanges did not: the same job, in this process, over the next transport in the order
            failed = str(exc.args[0])
            if len(exc.args) > 1 and exc.args[1] and os.environ.get("FOS_RESIDENT_DEFAULT") != "0":
                # ... but first the SAME transport with a launch group per CG iteration: the resident solve's in-kernel exchange (every workgroup polls
                # every record, the peers' words read from inside a persistent launch) is a harder test of the mailboxes than the folded exchange of
                # the launch-per-iteration kernels -- every rank takes this branch (the vote above), and the line says so (`resident_fallback`)
                os.environ["FOS_RESIDENT_DEFAULT"] = "0"
                os.environ["FOS_BENCH_RESIDENT_NOTE"] = ("the resident CG solve's exchanges timed out in the warm-up on the %s mailboxes (FOS_ECOMM); "
                                                        "the same transport was retried with a launch group per CG iteration" % failed)
                continue
            if os.environ.get("FOS_REDUCTION", "auto") != "auto":
                raise SystemExit("the %s mailboxes failed during the warm-up and FOS_REDUCTION allows no other transport" % failed)
            os.environ["FOS_REDUCTION_SKIP"] = ",".join(filter(None, [os.environ.get("FOS_REDUCTION_SKIP", ""), failed]))
            os.environ["FOS_REDUCTION_SKIP_NOTE_" + failed.upper()] = \
                "the self test passed, but an exchange of the warm-up timed out on the %s mailboxes (FOS_ECOMM)" % failed
    else:
        raise SystemExit("every transport failed during the warm-up")
    if world == 1 and dist is None and not args.direct and not args.no_recurrence_extras and out["config"]["cg_variant"] in ("reference", "resident") \
            and not os.environ.get("FOS_BENCH_SHARD"):
        # the same job on the other recurrences -- same handle, same state, `--steps` more outer iterations each: the merged-reduction recurrence
        # with a launch group per iteration (what sharded handles without mailboxes run: a scaling ratio can divide like by like) and, where the
        # headline ran the resident solve, the reference's recurrence in its three launches per iteration (what `value` was until round 5)
        extras = [("merged_recurrence_n1", "merged_update",
                   "N = 1 on the merged-reduction recurrence with a launch group per CG iteration (two launches, one reduction point)")]
        if out["config"]["cg_variant"] == "resident":
            extras.insert(0, ("reference_recurrence_n1", "reference",
                              "N = 1 on the reference's recurrence, three launches and two reduction points per CG iteration (the headline of rounds 1-5); "
                              "`value` runs the merged recurrence resident: one launch per CG SOLVE"))
        for key, variant, note in extras:
            # (these run BEHIND the headline: the solve has moved on, a step needs fewer CG iterations than in the timed region -- `cg_iters_per_step`
            #  says how many; a like-for-like rate needs a run of its own: FOS_RESIDENT_DEFAULT=0 / FOS_CG_VARIANT)
            try:
                dev.set_cg_variant(variant)
                dev.step(it + 1, 5, BIG, 1e-8)
                dev.sync()
                cg0m = dev.cg_total()
                tm = time.perf_counter()
                done_m, _, _ = dev.step(it + 6, args.steps, BIG, 1e-8)
                dev.sync()
                torch.cuda.synchronize()
                dtm = time.perf_counter() - tm
                it += 5 + done_m
                out[key] = {"value": round(done_m / dtm, 4), "unit": "iterations/s", "ms_per_step": round(1e3 * dtm / max(1, done_m), 4),
                            "cg_variant": dev.cg_variant_name(), "cg_iters_per_step": round((dev.cg_total() - cg0m) / max(1, done_m), 2),
                            "note": note + " -- measured behind the headline's timed region (compare `cg_iters_per_step` with the headline's)"}
            except Exception as exc:  # noqa: BLE001
                out[key] = {"failed": repr(exc)}
            finally:
                dev.set_cg_variant(None)
    if world > 1 and not weak_main and not args.no_weak_extra and args.workload == "C4":
        # the same job once more with 512 blocks PER RANK: weak scaling, reported beside the strong-scaling headline
        dev.close()
        # (an extra line: whatever goes wrong here -- the warm-up's collective vote, a time-out of the mailboxes, an out-of-memory -- costs this
        #  line, never the headline; every rank leaves run_case the same way: the vote, or the peers' time-outs behind a rank that failed alone)
        w_ok, w_exc = True, None
        try:
            wout, dev, prob, alg, it = run_case(True)
        except (Exception, SystemExit) as exc:  # noqa: BLE001
            w_ok, w_exc = False, exc
        if dist is not None and not agree(w_ok):
            if w_ok:
                dev.close()
            out["weak_scaling"] = {"failed": repr(w_exc) if w_exc is not None else "another rank's weak-scaling run failed"}
        else:
            out["weak_scaling"] = {k: wout[k] for k in ("value", "unit", "ms_per_step", "scaling", "warmup_effective")}
            out["weak_scaling"]["config"] = {k: wout["config"][k] for k in ("workload", "local_m", "local_n", "local_nnz", "cg_iters_per_step", "parallelism")}
            rk = wout["roofline_kkt"] if isinstance(wout["roofline_kkt"], dict) else wout["roofline"]
            out["weak_scaling"]["roofline_kkt"] = {k: rk.get(k) for k in ("achieved", "frac", "avg_kernel_ms", "all_ranks")}
    if world > 1 and dist is not None and args.workload == "C4" and not args.direct and not args.no_direct_extra and not weak_main \
            and out["config"].get("transport") in ("peer", "host"):
        # the same sharded job under DR(direct = true): the block form's diagonal blocks are local to a rank, its three scalar sums per projection ride
        # the same mailboxes (which time out instead of hanging: a failure here costs the extra line, not the headline)
        args.direct = True
        dev.close()                                   # (the headline's handle: its mailboxes and device memory are not needed any more at N > 1)
        d_ok, d_exc = True, None
        try:
            dout, ddev, _, _, _ = run_case(False)
            ddev.close()
        except (Exception, SystemExit) as exc:  # noqa: BLE001
            d_ok, d_exc = False, exc
        # one outcome for the job: a rank whose run failed (its peers then ran into the mailbox time-out, or did not) must not leave the others
        # reporting a number -- every rank takes part in this vote, whatever happened locally
        all_ok = agree(d_ok)
        try:
            if not all_ok:
                raise RuntimeError("rank %d: %s" % (rank, repr(d_exc) if d_exc is not None else "another rank's direct = true run failed"))
            out["direct_true"] = {
                "workload": dout["config"]["workload"] + ", DR(direct=true)", "value": dout["value"], "unit": dout["unit"], "ms_per_step": dout["ms_per_step"],
                "steps": dout["steps"], "warmup_effective": dout["warmup_effective"], "form": dout["config"]["direct"], "scaling": dout["scaling"],
                "sweeps_per_step": dout["config"]["sweeps_per_step"], "parallelism": dout["config"]["parallelism"],
                "time_shares": dout["time_shares"], "residuals_after_run": dout["config"]["residuals_after_run"],
                "note": "S1 = IndAffine([Q -I], 0) (the reference's `direct = true`): block form, three KKT sweeps and three exchanges of <= 3 doubles per "
                        "projection, no CG.  A different algorithm configuration than the headline (direct = false)",
            }
        except (Exception, SystemExit) as exc:  # noqa: BLE001
            out["direct_true"] = {"failed": repr(exc)}
        finally:
            args.direct = False

    # ---- CPU baseline (rank 0, N = 1): the C port of the oracle restatement on one core (+ all cores), bounded sample
    def cpu_baseline(dev, prob, alg, it, budget_one=10.0, budget_all=5.0):
        """Times oracle/fos_cport.c from the device's steady-state point and cross-checks ONE outer iteration against the numpy oracle."""
        sys.path.insert(0, str(ROOT / "oracle"))
        import fos_oracle as orc
        z = dev.get_iterate()
        xinit, pi, _ = dev.get_affine_state()
        om = orc.Model(prob.A, prob.b, prob.c, [(orc.CONE_CODES[k], l) for k, l in prob.K1],
                       [(orc.CONE_CODES[k], l) for k, l in prob.K2])
        oalg = {"GAP": lambda: orc.GAP(alg.alpha, alg.alpha1, alg.alpha2), "GAPA": lambda: orc.GAPA(alg.alpha, alg.beta),
                "FISTA": lambda: orc.FISTA(alg.alpha)}[type(alg).__name__]()
        oalg.init(om)
        oalg.S1.cgdata.xinit[:] = xinit
        oalg.S1.cgdata.firstrun = False
        oalg.S1.i = pi
        fa, fb, ft, fa12 = dev.get_alg_state()           # the algorithm's own *Data struct (fos_get_alg_state)
        if isinstance(oalg, orc.GAPA):
            oalg.alpha12 = fa12
        elif isinstance(oalg, orc.FISTA):
            oalg.y[:], oalg.xold[:], oalg.t = fa, fb, ft   # fista.jl:15-25
        try:                                    # one host core, like the single-threaded reference
            from threadpoolctl import threadpool_limits
            threadpool_limits(1)
        except Exception:
            pass
        ost = orc.HSDEStatus(om, BIG, 1e-8, 0, 0)
        ost.i = it + 1
        xo = z.copy()
        if True:
            # (1) parity cross-check: ONE outer iteration of the numpy oracle and of the GPU from the same state
            tc = time.perf_counter()
            oalg.step(xo, it + 1, ost)
            t_np = time.perf_counter() - tc
            cg_cpu = oalg.S1.getcgiter()
            # (2) the timed baseline: the plain-C port (oracle/fos_cport.c, checked against the oracle in tests/test_cport.py)
            #     from the same state, a bounded sample of >= ~10 s (at most 8 iterations); one thread like the
            #     single-threaded reference, then all host cores (OpenMP: rows/columns of A, vector passes, cones)
            import fos_cport as cport
            codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]

            def time_cport(threads, budget_s):
                cp = cport.CPort(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2), threads=threads)
                cp.set_affine_state(xinit, pi)
                xc = np.ascontiguousarray(z, dtype=np.float64).copy()
                a12 = fa12 if isinstance(oalg, orc.GAPA) else 2.0
                yc, xoldc, tc = (fa.copy(), fb.copy(), ft) if isinstance(oalg, orc.FISTA) else (None, None, 1.0)
                n_it, t0c, cgs = 0, time.perf_counter(), []
                while n_it < 8 and (n_it == 0 or time.perf_counter() - t0c < budget_s):
                    if isinstance(oalg, orc.GAPA):
                        a12 = cp.gapa_step(xc, alg.alpha, alg.beta, a12)
                    elif isinstance(oalg, orc.FISTA):
                        tc = cp.fista_step(xc, alg.alpha, yc, xoldc, tc)
                    else:
                        cp.gap_step(xc, alg.alpha, alg.alpha1, alg.alpha2)
                    cgs.append(cp.cgiter())
                    n_it += 1
                dt = time.perf_counter() - t0c
                cp.close()
                return n_it / dt, n_it, dt, cgs

            v1, n1, dt1, cgs1 = time_cport(1, budget_one)
            try:                                  # (threadpool_limits(1) above also caps OpenMP's default team size; the C
                ncores = len(os.sched_getaffinity(0))   # port passes num_threads explicitly, so ask the OS for the core count)
            except Exception:
                ncores = os.cpu_count() or 1
            try:                                  # a container's CPU quota, when there is one
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if quota != "max":
                    ncores = max(1, min(ncores, int(float(quota) / float(period) + 0.5)))
            except Exception:
                pass
            ncores = min(ncores, 32)              # beyond ~32 threads the fork/join of every vector pass dominates (measured:
                                                  # 256 threads are 20x slower than one)
            vall = time_cport(ncores, budget_all) if ncores > 1 and budget_all > 0 else None
            dev.step(it + 1, 1, BIG, 1e-8)
            zg = dev.get_iterate()
            cpu = {
                "value": round(v1, 5), "unit": "iterations/s", "cores": 1, "kind": "port",
                "sample": "%d steady-state outer iterations (from i=%d, %s CG iterations each) of oracle/fos_cport.c -- plain C, "
                          "CSC scatter/gather SpMV and 4 sweeps per KKT apply as the reference, one thread -- from the GPU's state, "
                          "%.1f s" % (n1, it + 1, "/".join(map(str, cgs1[:4])), dt1),
                "seconds": round(dt1, 3),
                "multi_core": ({"value": round(vall[0], 5), "cores": ncores, "iterations": vall[1], "seconds": round(vall[2], 3),
                               "how": "same C port, OpenMP over rows/columns of A, vector passes and cones"} if vall else None),
                "numpy_oracle_same_step": {"value": round(1.0 / t_np, 5), "seconds": round(t_np, 3), "cg_iters": cg_cpu},
                "gpu_vs_cpu_same_step_rel_dev": float(np.linalg.norm(zg - xo) / max(1.0, np.linalg.norm(xo))),
                "gpu_cg_iters_same_step": dev.cgiter(),
            }
            return cpu
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(dev, prob, alg, it)
    # ---- the C4 instance as SURVEY 8(d) writes it (A_j as drawn, not divided by 32): same size, same code path, twice the CG
    #      iterations per outer iteration -- timed here so that the driver's line carries it beside the headline
    if world == 1 and args.workload == "C4" and args.c4_scale is None and not args.small and not args.no_raw_instance:
        args.c4_scale = 1.0
        try:
            rout, rdev, rprob, ralg, rit = run_case(False)
            rr = rout["roofline_kkt"] if isinstance(rout["roofline_kkt"], dict) else rout["roofline"]
            rcpu = None
            if not args.no_cpu_baseline:
                try:
                    rcpu = cpu_baseline(rdev, rprob, ralg, rit, budget_one=8.0, budget_all=0.0)      # (a shorter, single-core sample: the default run stays within minutes)
                except Exception as exc:  # noqa: BLE001
                    rcpu = {"failed": repr(exc)}
            rdev.close()
            out["raw_instance"] = {
                "workload": rout["config"]["workload"], "value": rout["value"], "unit": rout["unit"], "ms_per_step": rout["ms_per_step"],
                "steps": rout["steps"], "warmup_effective": rout["warmup_effective"],
                "cg_iters_per_step": rout["config"]["cg_iters_per_step"],
                "roofline": {k: rr[k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_kernel_ms", "bytes_per_launch")},
                "time_shares": rout["time_shares"], "residuals_after_run": rout["config"]["residuals_after_run"],
                "cpu_baseline": rcpu,
                "note": "SURVEY 8(d)'s C4 without the data scaling of the headline instance (BASELINE.md 3): the same operator format, "
                        "kernels and sizes; the KKT matrix is worse conditioned, so CG needs about twice the iterations per outer iteration",
            }
            # the same numbers at the top level, under the name the survey's recipe deserves: whoever quotes this line against SURVEY 8(d)'s
            # instance quotes THIS value; `value` is the instance BASELINE.md 3 records (data scaled, ||b|| = ||c|| = 10, A_j / 32)
            out["value_as_specified"] = {
                "value": rout["value"], "unit": rout["unit"], "ms_per_step": rout["ms_per_step"], "cg_iters_per_step": rout["config"]["cg_iters_per_step"],
                "workload": rout["config"]["workload"],
                "roofline_frac": rr["frac"],
                "cpu_baseline": ({k: rcpu[k] for k in ("value", "unit", "cores", "kind", "sample", "gpu_vs_cpu_same_step_rel_dev")} if rcpu and "value" in rcpu else rcpu),
                "speedup_vs_cpu_port_one_core": (round(rout["value"] / rcpu["value"], 1) if rcpu and rcpu.get("value") else None),
                "what": "the C4 instance exactly as SURVEY 8(d) writes it (A_j as drawn); details under `raw_instance`",
            }
        finally:
            args.c4_scale = None
    # ---- the same C4 with DR(direct = true) (HSDE.jl:12-15): the exact projection in the block form, three sweeps instead of ~17 CG iterations
    if world == 1 and dist is None and args.workload == "C4" and not args.direct and not args.small and not args.no_direct_extra and args.c4_scale is None \
            and not os.environ.get("FOS_BENCH_SHARD"):
        args.direct = True
        try:
            dout, ddev, _, _, _ = run_case(False)
            ddev.close()
            dr = dout["roofline_kkt"] if isinstance(dout["roofline_kkt"], dict) else dout["roofline"]
            out["direct_true"] = {
                "workload": dout["config"]["workload"] + ", DR(direct=true)", "value": dout["value"], "unit": dout["unit"], "ms_per_step": dout["ms_per_step"],
                "steps": dout["steps"], "warmup_effective": dout["warmup_effective"], "form": dout["config"]["direct"],
                "sweeps_per_step": dout["config"]["sweeps_per_step"],
                "roofline": {k: dr[k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_kernel_ms", "bytes_per_launch")},
                "time_shares": dout["time_shares"], "residuals_after_run": dout["config"]["residuals_after_run"],
                "note": "S1 = IndAffine([Q -I], 0) as the reference's `direct = true` option defines it: the EXACT projection, here through the block form "
                        "(I + A'A block diagonal with 512 blocks of 32 columns, inverted once; three KKT sweeps + one block-diagonal product per projection, "
                        "no CG).  A different algorithm configuration than the headline (which keeps the reference's default direct = false)",
            }
        finally:
            args.direct = False
    if dist is not None:
        # librccl prints its version banner through C stdio, which a pipe buffers until exit: flush it now so the
        # JSON line below is the last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
