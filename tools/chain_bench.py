"""CG chain timing (fos_bench_cg_chain): `python tools/chain_bench.py C4 [iters] [--wg N ...]` -- ms per CG iteration for the
two- and three-launch iteration, eager and hipGraph replay, optionally at several sweep grid sizes."""
import sys; sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
import numpy as np
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 20
wgs = [int(a) for a in sys.argv[3:] if not a.startswith("-")] or [0]
gen = {"C4": pkg.workloads.c4_block_sdp, "C2": pkg.workloads.c2_lp, "C3": pkg.workloads.c3_socp, "C5": pkg.workloads.c5_mixed,
       "C4s": lambda: pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, 64))}[wl]
prob = gen()
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
d.set_iterate(np.random.default_rng(0).standard_normal(d.N))
for wg in wgs:
    if wg: d.set_tuning(spmv_workgroups=wg)
    import os
    for fuse in ([int(os.environ["FUSE"])] if "FUSE" in os.environ else (0, 1)):
        d.set_tuning(fuse_p=fuse)
        for graph in ([bool(int(os.environ["GRAPH"]))] if "GRAPH" in os.environ else (False, True)):
            ms = min(d.bench_cg_chain(iters, 5, graph) for _ in range(3))
            print("%s wg=%d fuse_p=%d graph=%d : %.2f us per CG iteration" % (wl, wg, fuse, int(graph), 1e3 * ms), flush=True)
print("kkt apply alone: %.2f us" % (1e3 * d.bench_kkt(50) / 50))
