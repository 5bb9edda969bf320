#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03n
mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c5" > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/t.log
python3 tools/kkt_only.py C5 0 50
FOS_DBG_FLAGS=2 python3 tools/kkt_only.py C5 0 50
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json; }
run c5_pre A=1 -- --workload C5
run c5_old FOS_DBG_FLAGS=2 -- --workload C5
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03n/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"]["cg_iters_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
