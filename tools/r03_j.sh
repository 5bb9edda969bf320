#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03j
mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cg_variants.py -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/t.log
bash tools/r03_trace.sh c4 "" > /dev/null 2>&1
cat gpurun_out/r03/trace_c4.md
bash tools/r03_trace.sh c2 "" --workload C2 > /dev/null 2>&1
head -8 gpurun_out/r03/trace_c2.md
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py --no-cpu-baseline "$@" 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json; }
run c4 A=1 --
run c2 A=1 -- --workload C2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03j/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"]["cg_iters_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], "shares", d["time_shares"]["kkt_sweep"], d["time_shares"]["cg_vector_updates"])
    except Exception as e:
        print(f, "ERR", e)
PY
