#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03h
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py::test_c4_operators_and_projections_full_size tests/test_gpu_fullsize.py::test_c2_full_size_operators_and_progress -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/t.log
run() { # tag env... -- flags
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py --no-cpu-baseline --steps 30 "$@" 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json
}
for K in 1 2 3 4 5 6 8; do run c4_k$K FOS_TILE_TALL=$K -- ; done
run c4_auto A=1 --
for K in 1 2 4 8 16; do run c2_k$K FOS_TILE_TALL=$K -- --workload C2; done
run c3 A=1 -- --workload C3
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03h/*.json")):
    try:
        d=json.load(open(f))
        o=d["roofline"]["operator_format"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"]["cg_iters_per_step"], "sweep", d["roofline"]["avg_kernel_ms"], "shares", d["time_shares"]["kkt_sweep"], d["time_shares"]["cg_vector_updates"], "blocks", o["blocks"], "slots", o["slots"])
    except Exception as e:
        print(f, "ERR", e)
PY
