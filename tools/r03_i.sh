#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03i
mkdir -p $OUT
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cg_variants.py tests/test_gpu_peer_mailbox.py tests/test_gpu_bench_flow.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -6 $OUT/t.log
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py --no-cpu-baseline "$@" 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json; }
run c4 A=1 --
run c4_expl FOS_DEF_EXPLICIT=1 --
run c2 A=1 -- --workload C2
run c2_expl FOS_DEF_EXPLICIT=1 -- --workload C2
run s64_v3 FOS_CG_VARIANT=3 -- --small
run s64_v3_expl FOS_CG_VARIANT=3 FOS_DEF_EXPLICIT=1 -- --small
run c4_raw A=1 -- --c4-scale 1 --steps 20
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03i/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"]["cg_iters_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], "shares", d["time_shares"]["kkt_sweep"], d["time_shares"]["cg_vector_updates"], d["config"]["residuals_after_run"])
    except Exception as e:
        print(f, "ERR", e)
PY
