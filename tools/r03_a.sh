#!/bin/bash
# round 3, first GPU pass: the new CG-variant tests, the whole GPU suite, then bench lines per CG variant (C4 and the 64-block shard)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_cg_variants.py -x -q -m gpu > $OUT/t_variants.log 2>&1; echo "variants rc=$?" | tee -a $OUT/rc.log
tail -15 $OUT/t_variants.log
timeout 1500 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_cg_variants.py > $OUT/t_all.log 2>&1; echo "all rc=$?" | tee -a $OUT/rc.log
tail -15 $OUT/t_all.log
for V in 0 2 3; do
  FOS_CG_VARIANT=$V python3 bench.py --no-cpu-baseline 2> $OUT/c4_v$V.err | tail -1 > $OUT/c4_v$V.json
  FOS_CG_VARIANT=$V python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_v$V.err | tail -1 > $OUT/s64_v$V.json
done
FOS_FORCE_DIST=1 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_dist.err | tail -1 > $OUT/s64_dist.json
FOS_FORCE_DIST=1 FOS_CG_VARIANT=0 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_dist_v0.err | tail -1 > $OUT/s64_dist_v0.json
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03a/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"].get("cg_variant"), d["config"]["cg_iters_per_step"], d["time_shares"], d["roofline"]["avg_kernel_ms"], d["config"]["parallelism"])
    except Exception as e:
        print(f, "ERR", e)
PY
