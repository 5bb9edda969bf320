"""KKT apply alone (fos_bench_kkt): `python tools/kkt_only.py C4|C2|C3|C5 [spmv workgroups] [reps]` -- used by tools/pmc_kkt.sh."""
import sys; sys.path.insert(0,'.')
import __graft_entry__ as ge
pkg = ge.load_package()
wl = sys.argv[1] if len(sys.argv)>1 else "C4"
wg = int(sys.argv[2]) if len(sys.argv)>2 else 0
reps = int(sys.argv[3]) if len(sys.argv)>3 else 50
prob = {"C4": pkg.workloads.c4_block_sdp, "C2": pkg.workloads.c2_lp, "C3": pkg.workloads.c3_socp, "C5": pkg.workloads.c5_mixed}[wl]()
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
if wg: d.set_tuning(spmv_workgroups=wg)
import numpy as np
d.set_iterate(np.random.default_rng(0).standard_normal(d.N))
ms = d.bench_kkt(reps)
by = 24.0*prob.nnz + 4*(prob.m+prob.n+2) + 32*(prob.m+prob.n)
print(wl, "wg", wg, "kkt avg us %.2f"%(1e3*ms/reps), "alg GB/s %.1f"%(by/(ms/reps*1e-3)/1e9))
