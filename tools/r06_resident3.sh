#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0 FOS_RESIDENT_WAIT_S=3
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_bench_flow.py tests/test_gpu_longstep.py tests/test_gpu_peer_mailbox.py -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/r06_resident_tests.txt
cat gpurun_out/r06_resident_tests.txt
export FOS_FORCE_DIST=1 FOS_BENCH_SHARD=0/8
for tr in peer host; do
  for lb in 0 1; do
      tag=${tr}_res1_lb${lb}
      FOS_PEER_LOOPBACK=$lb FOS_REDUCTION=$tr timeout 600 python bench.py --steps 50 --no-cpu-baseline --no-raw-instance > gpurun_out/r06_shard64_$tag.json 2> gpurun_out/r06_shard64_$tag.err
      python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_shard64_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["config"]["cg_variant"], d["config"]["cg_iters_per_step"], {k[:12]: v for k, v in d["time_shares"].items()}, d["config"]["residuals_after_run"]["p"], d["roofline"].get("us_per_cg_iteration"))
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/r06_shard64_$tag.err").read()[-1500:])
PY
  done
done
