#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03g
mkdir -p $OUT
timeout 1800 python3 -m pytest tests -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -6 $OUT/t.log
for W in C5 C2 C3; do
  python3 bench.py --workload $W --no-cpu-baseline 2> $OUT/$W.err | tail -1 > $OUT/$W.json
done
bash tools/r03_trace.sh c5 "" --workload C5 > /dev/null 2>&1
cat gpurun_out/r03/trace_c5.md
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03g/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"].get("cg_variant"), d["config"]["cg_iters_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["time_shares"])
    except Exception as e:
        print(f, "ERR", e)
PY
