#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 1800 python3 -m pytest tests -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -8 $OUT/t.log
for V in 0 3; do
  bash tools/r03_trace.sh s64_v$V "FOS_CG_VARIANT=$V" --small > /dev/null 2>&1
  cat gpurun_out/r03/trace_s64_v$V.md
done
bash tools/r03_trace.sh s64_v3_nodeep "FOS_CG_VARIANT=3 FOS_TILE_DEEP=0" --small > /dev/null 2>&1
head -6 gpurun_out/r03/trace_s64_v3_nodeep.md
bash tools/r03_trace.sh c4_v0 "FOS_CG_VARIANT=0" > /dev/null 2>&1
cat gpurun_out/r03/trace_c4_v0.md
for V in 0 2 3; do
  FOS_CG_VARIANT=$V python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_v$V.err | tail -1 > $OUT/s64_v$V.json
done
FOS_CG_VARIANT=3 FOS_TILE_DEEP=0 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_v3nd.err | tail -1 > $OUT/s64_v3nd.json
FOS_CG_VARIANT=3 FOS_SHIFT_FUSE=0 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_v3ns.err | tail -1 > $OUT/s64_v3ns.json
FOS_FORCE_DIST=1 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_dist.err | tail -1 > $OUT/s64_dist.json
FOS_FORCE_DIST=1 FOS_REDUCTION=rccl python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_dist_rccl.err | tail -1 > $OUT/s64_dist_rccl.json
python3 bench.py --no-cpu-baseline 2> $OUT/c4.err | tail -1 > $OUT/c4.json
FOS_CG_VARIANT=3 python3 bench.py --no-cpu-baseline 2> $OUT/c4_v3.err | tail -1 > $OUT/c4_v3.json
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"].get("cg_variant"), d["config"]["cg_iters_per_step"], d["roofline"]["avg_kernel_ms"], d["config"]["parallelism"])
    except Exception as e:
        print(f, "ERR", e)
PY
