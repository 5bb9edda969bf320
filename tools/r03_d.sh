#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_cg_variants.py tests/test_gpu_peer_mailbox.py tests/test_gpu_parity.py tests/test_gpu_certificates.py tests/test_gpu_edge_cases.py -q -m gpu -x > $OUT/t.log 2>&1; echo "tests rc=$?"
tail -6 $OUT/t.log
bash tools/r03_trace.sh s64_v3 "FOS_CG_VARIANT=3" --small > /dev/null 2>&1
cat gpurun_out/r03/trace_s64_v3.md
for V in 0 2 3; do
  FOS_CG_VARIANT=$V python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_v$V.err | tail -1 > $OUT/s64_v$V.json
done
FOS_CG_VARIANT=3 FOS_PSD_NARROW=1 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_v3_psd256.err | tail -1 > $OUT/s64_v3_psd256.json
FOS_FORCE_DIST=1 python3 bench.py --small --no-cpu-baseline 2> $OUT/s64_dist.err | tail -1 > $OUT/s64_dist.json
python3 bench.py --no-cpu-baseline 2> $OUT/c4.err | tail -1 > $OUT/c4.json
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03d/*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["config"].get("cg_variant"), d["config"]["cg_iters_per_step"], d["roofline"]["avg_kernel_ms"], (d.get("roofline_psd") or {}).get("avg_kernel_ms"), d["config"]["parallelism"])
    except Exception as e:
        print(f, "ERR", e)
PY
