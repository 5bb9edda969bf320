#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03l
mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_cg_variants.py -q -m gpu -x -k "pipelined or chain" > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -30 $OUT/t.log
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json; }
run s64_v4 FOS_CG_VARIANT=4 -- --small
run s64_v3 FOS_CG_VARIANT=3 -- --small
run s64_dist_v4 FOS_CG_VARIANT=4 FOS_FORCE_DIST=1 -- --small
run c3_v4 FOS_CG_VARIANT=4 -- --workload C3
run c3_v0 A=1 -- --workload C3
run c4_v4 FOS_CG_VARIANT=4 --
run c2_v4 FOS_CG_VARIANT=4 -- --workload C2
bash tools/r03_trace.sh s64_v4 "FOS_CG_VARIANT=4" --small > /dev/null 2>&1
cat gpurun_out/r03/trace_s64_v4.md
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03l/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"].get("cg_variant"), d["config"]["cg_iters_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], d["config"]["residuals_after_run"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
