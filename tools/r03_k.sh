#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03k
mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -6 $OUT/t.log
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py --no-cpu-baseline "$@" 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json; }
run c4 A=1 --
run c4_oldstart FOS_CG_FUSED_START=0 --
run c3 A=1 -- --workload C3
run c3_oldstart FOS_CG_FUSED_START=0 -- --workload C3
run c5 A=1 -- --workload C5
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03k/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"]["cg_iters_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], "shares", d["time_shares"]["kkt_sweep"], d["time_shares"]["cg_vector_updates"])
    except Exception as e:
        print(f, "ERR", e)
PY
