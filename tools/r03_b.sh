#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_cg_variants.py tests/test_gpu_peer_mailbox.py "tests/test_gpu_parity.py::test_rccl_reduction_path_single_rank" tests/test_gpu_parity.py::test_cg_kkt_matches_dense_solve_and_oracle_count tests/test_gpu_parity.py::test_first_iterations_match_oracle -q -m gpu > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -25 $OUT/t.log
for V in 0 2 3; do
  bash tools/r03_trace.sh s64_v$V "FOS_CG_VARIANT=$V" --small > /dev/null 2>&1
  cat gpurun_out/r03/trace_s64_v$V.md
  tail -1 gpurun_out/r03/trace_s64_v$V.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['config']['cg_variant'])"
done
for V in 0 2; do
  bash tools/r03_trace.sh c4_v$V "FOS_CG_VARIANT=$V" > /dev/null 2>&1
  cat gpurun_out/r03/trace_c4_v$V.md
done
for V in 0 2 3; do
  FOS_CG_VARIANT=$V python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_v$V.err | tail -1 > $OUT/s64_v$V.json
done
FOS_FORCE_DIST=1 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_dist.err | tail -1 > $OUT/s64_dist.json
FOS_CG_VARIANT=2 python3 bench.py --no-cpu-baseline 2> $OUT/c4_v2.err | tail -1 > $OUT/c4_v2.json
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03b/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"].get("cg_variant"), d["config"]["cg_iters_per_step"], d["roofline"]["avg_kernel_ms"], d["config"]["parallelism"])
    except Exception as e:
        print(f, "ERR", e)
PY
