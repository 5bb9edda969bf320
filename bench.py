#!/usr/bin/env python3
"""
bench.py -- GAP/DR outer iterations per second + achieved HBM GB/s of the CG SpMV (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4|C2|C3|C5|...]

A "step" is ONE outer iteration of the solver (solverwrapper.jl:23-29: affine projection by warm-started CG over
the KKT operator -> cone projection -> relaxations) on a synthetic problem already resident in HBM.  Warm-up
defaults to 200 iterations so that the CG tolerance schedule max(0.2^sqrt(i), l*eps) has reached its floor
(steady state, ~all of the per-iteration work).  The timed region excludes set-up and the every-`checki` status.

Default workload (all N): C4, BASELINE.json configs[3] "Block-diagonal SDP, 512 PSD blocks of size 64x64, DR,
cone-sharded across 1/2/4/8 MI355X via RCCL" -- the configuration the metric ("... at 1/2/4/8 GPUs") and the
north-star targets (1e6-variable problem, >=3.5x at 8 GPUs on the block-PSD workload) are quoted on; it fits one
GPU.  N > 1 is STRONG scaling of that one problem: rank g owns blocks [512 g/N, 512 (g+1)/N) and only scalars
cross GPUs (RCCL all-reduce, in stream).  `--workload C2` runs configs[1] (dense 5000x10000 LP) on one GPU.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = fused dual-RHS KKT SpMV,
HIP events on the solver's stream around every launch in the timed region) and, at N = 1, `cpu_baseline`
(the oracle restatement timed on one host core for ONE steady-state outer iteration from the GPU's state).
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


def build_problem(pkg, workload, nranks, rank, small):
    w = pkg.workloads
    if workload == "C4":
        nb = 64 if small else 512
        lo, hi = (nb * rank) // nranks, (nb * (rank + 1)) // nranks
        prob = w.c4_block_sdp(nblocks=nb, block_range=(lo, hi))
        desc = "C4 block-diagonal SDP, %d PSD(64) blocks, 32 free vars/block, DR" % nb
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--small", action="store_true", help="reduced sizes (smoke / CI); not a valid benchmark number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spmv-wg", type=int, default=0)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
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
        if host_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    pkg = ge.load_package()

    t0 = time.time()
    prob, alg, desc, glob = build_problem(pkg, args.workload, world, rank, args.small)
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
        # scalar sums: peer mailboxes (one xGMI write latency per exchange) when every rank's self test passes,
        # otherwise the in-stream RCCL all-reduce set up above.  FOS_REDUCTION=rccl|peer|auto (default auto).
        want = "peer" if host_gloo else os.environ.get("FOS_REDUCTION", "auto")
        if want != "rccl":
            # every collective below is executed by every rank in the same order, whatever fails locally
            def agree(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=tdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1
            try:
                mine = dev.peer_export()
            except Exception as exc:                      # no uncached allocation / IPC export on this device
                print("rank %d: peer mailbox export failed (%s)" % (rank, exc), file=sys.stderr, flush=True)
                mine = None
            handles = [None] * world
            dist.all_gather_object(handles, mine)
            ok = all(h is not None for h in handles)
            if ok:
                try:
                    dev.peer_open(world, rank, handles, timeout_s=20.0)
                    dev.sync()
                except Exception as exc:                  # IPC mapping not available between these devices
                    print("rank %d: peer mailboxes unavailable (%s)" % (rank, exc), file=sys.stderr, flush=True)
                    ok = False
            if agree(ok):
                dist.barrier()
                try:
                    ok = dev.peer_selftest(64)
                except Exception as exc:
                    print("rank %d: peer mailbox self test raised (%s)" % (rank, exc), file=sys.stderr, flush=True)
                    ok = False
                if agree(ok):
                    dev.peer_enable(True)
                    reduction = "peer mailboxes over xGMI (HIP IPC)"
            if want == "peer" and not reduction.startswith("peer"):
                raise SystemExit("FOS_REDUCTION=peer but the peer mailboxes are not usable")
    if args.spmv_wg:
        dev.set_tuning(spmv_workgroups=args.spmv_wg)
    dev.set_alg(alg)
    dev.set_iterate(None)
    t_setup = time.time() - t0

    BIG = 10 ** 12

    def barrier():
        dev.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- warm-up (untimed): W outer iterations
    it = 0
    cg_hist = []
    if args.warmup > 0:
        done, _, _ = dev.step(1, args.warmup, BIG, 1e-8)
        it += done
    # ---- timed: exactly K outer iterations
    PROF_PERIOD = 4            # HIP events around every 4th KKT launch of the timed region (an event pair per launch costs ~5 % of a step)
    dev.profile(PROF_PERIOD)
    dev.profile_read()
    cg0 = dev.cg_total()
    barrier()
    t1 = time.perf_counter()
    done, _, _ = dev.step(it + 1, args.steps, BIG, 1e-8)
    dev.sync()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    it += done
    elapsed = t2 - t1
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    launches, kms, kbytes = dev.profile_read()
    cg_timed = dev.cg_total() - cg0
    dev.profile(False)
    barrier()

    # one check on the current point (not timed): residuals for the record
    _, _, chk = dev.step(it + 1, 1, 1, 1e-8)
    it += 1

    # HBM traffic of the dominant kernel measured offline with PMC counters on this workload (profiles/, per launch)
    traffic = None
    try:
        cands = sorted(Path(ROOT / "profiles").glob("r*_kkt_traffic.json"))
        if cands and not args.small and world == 1:
            traffic = json.load(open(cands[-1])).get(args.workload, {}).get("traffic_bytes")
    except Exception:
        traffic = None
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    value = args.steps / elapsed
    avg_kernel_ms = kms / max(1, launches)
    achieved = kbytes / (avg_kernel_ms * 1e-3) / 1e9 if launches else 0.0
    # what the device format actually streams per KKT apply (model, not a counter): stored values + stored column indices +
    # block descriptors + partial-sum slots (written by the sweep, read by the deferred-row kernel) + vectors in/out + [c;b]
    ost = dev.operator_stats()
    stored_bytes = (8.0 * ost["vals"] + 4.0 * ost["cols"] + 32.0 * ost["blocks"] + 2 * 16.0 * ost["slots"]
                    + 2 * 16.0 * (prob.m + prob.n) + 8.0 * (prob.m + prob.n))
    stored_gbs = stored_bytes / (avg_kernel_ms * 1e-3) / 1e9 if launches else 0.0
    # N > 1: every rank sweeps its own shard at the same time; the job's SpMV rate is the sum over ranks
    agg = None
    if dist is not None:
        tt = torch.tensor([achieved, stored_gbs], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        agg = {"achieved_all_ranks": round(float(tt[0]), 1), "stored_gbs_all_ranks": round(float(tt[1]), 1),
               "frac_of_n_gpus_peak": round(float(tt[0]) / (HBM_PEAK_GBS * world), 4),
               "note": "this rank's KKT apply on its shard is what `achieved` prices; the sum over the ranks is the job's rate"}
    out = {
        "metric": "GAP/DR outer iterations/sec (+ achieved HBM GB/s of the CG SpMV in `roofline`)",
        "value": round(value, 4),
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (numpy default_rng, seeds in firstordersolvers.jl_amd/workloads.py)",
        "config": {
            "workload": desc + (" [SMALL]" if args.small else ""),
            "solver": type(alg).__name__,
            "local_m": int(prob.m), "local_n": int(prob.n), "local_nnz": int(prob.nnz),
            "cg_iters_per_step": round(cg_timed / max(1, args.steps), 2),
            "parallelism": "cone-sharded x%d (scalar sums: %s)" % (world, reduction) if dist is not None else "single GPU",
            "residuals_after_run": {"p": chk.p, "d": chk.d, "g": chk.g, "iteration": it},
            "setup_s": round(t_setup, 2), "generate_s": round(t_gen, 2),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "kkt2_kernel (fused dual-RHS KKT SpMV)" + (" + kkt2_deferred_kernel (rows spread over dual tiles)" if ost["tiles"] else ""),
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_of_measured_copy_6290": round(achieved / 6290.0, 4),
            "traffic": traffic,
            "traffic_note": "bytes per launch from rocprofv3 PMC passes committed under profiles/ (not collected in this run)" if traffic else None,
            "algorithmic_bytes_per_launch": kbytes,
            "stored_bytes_per_launch_model": stored_bytes,
            "stored_gbs": round(stored_gbs, 1),
            "frac_stored": round(stored_gbs / HBM_PEAK_GBS, 4),
            "note": ("algorithmic bytes follow SURVEY 8(d) (A and A' each streamed once per dual-RHS apply); dual tiles store dense "
                     "rectangles of A once for both products, so the sweep moves fewer bytes than that and `frac` can exceed 1; "
                     "`frac_stored` prices the bytes the format really streams") if ost["tiles"] else None,
            "operator_format": ost,
            "all_ranks": agg,
            "avg_kernel_ms": round(avg_kernel_ms, 5),
            "launches_timed": launches,
            "launches_in_region": cg_timed,
            "event_sampling": "every %d-th KKT launch of the timed region" % PROF_PERIOD,
            "kernel_share_of_step": round(avg_kernel_ms * cg_timed / (1e3 * elapsed), 4) if elapsed > 0 else None,
        },
    }

    # ---- CPU baseline (rank 0, N = 1): the C port of the oracle restatement on one core (+ all cores), bounded sample
    if world == 1 and not args.no_cpu_baseline:
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
        if isinstance(oalg, orc.GAPA):
            oalg.alpha12 = dev.alpha12()
        try:                                    # one host core, like the single-threaded reference
            from threadpoolctl import threadpool_limits
            threadpool_limits(1)
        except Exception:
            pass
        ost = orc.HSDEStatus(om, BIG, 1e-8, 0, 0)
        ost.i = it + 1
        xo = z.copy()
        if isinstance(oalg, orc.FISTA):
            cpu = None          # FISTA's y/xold/t live on the device only; skip the hand-off
        else:
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
                a12 = dev.alpha12() if isinstance(oalg, orc.GAPA) else 2.0
                n_it, t0c, cgs = 0, time.perf_counter(), []
                while n_it < 8 and (n_it == 0 or time.perf_counter() - t0c < budget_s):
                    if isinstance(oalg, orc.GAPA):
                        a12 = cp.gapa_step(xc, alg.alpha, alg.beta, a12)
                    else:
                        cp.gap_step(xc, alg.alpha, alg.alpha1, alg.alpha2)
                    cgs.append(cp.cgiter())
                    n_it += 1
                dt = time.perf_counter() - t0c
                cp.close()
                return n_it / dt, n_it, dt, cgs

            v1, n1, dt1, cgs1 = time_cport(1, 10.0)
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
            vall = time_cport(ncores, 5.0) if ncores > 1 else None
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
            out["cpu_baseline"] = cpu
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
