#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03m
mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cg_variants.py tests/test_gpu_linesearch.py -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/t.log
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json; }
run c4_split A=1 --
run c4_nosplit FOS_UPD_SPLIT=0 --
run c4_split2 A=1 --
run c4_nosplit2 FOS_UPD_SPLIT=0 --
run s64_v0_split FOS_CG_VARIANT=0 -- --small
run s64_v0_nosplit FOS_CG_VARIANT=0 FOS_UPD_SPLIT=0 -- --small
bash tools/r03_trace.sh c4 "" > /dev/null 2>&1
head -8 gpurun_out/r03/trace_c4.md
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03m/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"].get("cg_variant"), d["config"]["cg_iters_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], d["time_shares"]["cg_vector_updates"])
    except Exception as e:
        print(f, "ERR", e)
PY
