#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import sys, os; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
prob = pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, 64))
for flags in ("0", "1"):
    os.environ["FOS_DBG_FLAGS"] = flags
    for wg in (0, 512, 264, 1056):
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        d.set_iterate(np.random.default_rng(0).standard_normal(d.N))
        if wg: d.set_tuning(spmv_workgroups=wg)
        d.set_cg_variant("merged_update")
        ms = min(d.bench_cg_chain(20, 5, False) for _ in range(3))
        k = min(d.bench_kkt(50) / 50 for _ in range(3))
        print("dbg_flags=%s wg=%d : chain %.2f us per CG iteration, stand-alone apply %.2f us" % (flags, wg, 1e3 * ms, 1e3 * k), flush=True)
        d.close()
PY
