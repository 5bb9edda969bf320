#!/bin/bash
# Round 6, second pass on the resident CG solve: parity tests, in-kernel stamps, the shard A/B (resident on / off, peer / host, loopback on / off)
export HSA_ENABLE_IPC_MODE_LEGACY=0 FOS_RESIDENT_WAIT_S=3
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06_resident_tests.txt
cat gpurun_out/r06_resident_tests.txt
FOSHIP_LIB=firstordersolvers.jl_amd/csrc/libfoship_res_stamps.so timeout 300 python tools/res_stamps.py 64 > gpurun_out/r06_res_stamps_v3.txt 2>&1
grep "per compute\|mean over" gpurun_out/r06_res_stamps_v3.txt
export FOS_FORCE_DIST=1 FOS_BENCH_SHARD=0/8
for tr in peer host; do
  for res in 1 0; do
    for lb in 0 1; do
      tag=${tr}_res${res}_lb${lb}
      FOS_PEER_LOOPBACK=$lb FOS_RESIDENT_DEFAULT=$res FOS_REDUCTION=$tr timeout 600 python bench.py --steps 50 --no-cpu-baseline --no-raw-instance > gpurun_out/r06_shard64_$tag.json 2> gpurun_out/r06_shard64_$tag.err
      python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_shard64_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["config"]["cg_variant"], d["config"]["cg_iters_per_step"], {k[:12]: v for k, v in d["time_shares"].items()}, d["config"]["residuals_after_run"]["p"])
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/r06_shard64_$tag.err").read()[-1500:])
PY
    done
  done
done
