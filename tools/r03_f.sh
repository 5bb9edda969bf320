#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import sys, os; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
cases = {"C4 shard": (lambda: pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, 64)), "merged_update"),
         "C3": (lambda: pkg.workloads.c3_socp(), "reference")}
for name, (gen, variant) in cases.items():
    prob = gen()
    for res in ("0", "1"):
        os.environ["FOS_RESIDENT"] = res
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        d.set_iterate(np.random.default_rng(0).standard_normal(d.N))
        for v in (variant, "merged_sweep"):
            d.set_cg_variant(v)
            ms = min(d.bench_cg_chain(20, 5, False) for _ in range(3))
            print("%s resident=%s %s: chain %.2f us per CG iteration" % (name, res, v, 1e3 * ms), flush=True)
        k = min(d.bench_kkt(50) / 50 for _ in range(3))
        print("%s resident=%s: stand-alone apply %.2f us" % (name, res, 1e3 * k), flush=True)
        d.close()
PY
for R in 0 1; do
FOS_RESIDENT=$R FOS_CG_VARIANT=3 python3 bench.py --small --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('s64 resident=$R', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
FOS_RESIDENT=$R python3 bench.py --workload C3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 resident=$R', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['cg_iters_per_step'])"
done
