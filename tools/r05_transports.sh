#!/bin/bash
# The 64-block shard of C4 (what one of eight ranks holds) in the sharded code path on ONE GPU, under each transport of the scalar sums,
# with the rank's own words travelling through the mailbox too (FOS_PEER_LOOPBACK=1: a single rank then pays the transport's store -> poll latency
# in every folded exchange) and without (no hop paid).  Output: gpurun_out/r05_bench_c4_shard64_{peer,host,rccl}[_loopback].json
export FOS_FORCE_DIST=1 FOS_BENCH_SHARD=0/8 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for tr in peer host rccl; do
  FOS_REDUCTION=$tr python bench.py --steps 50 --no-cpu-baseline --no-raw-instance > gpurun_out/r05_bench_c4_shard64_$tr.json 2> gpurun_out/r05_bench_c4_shard64_$tr.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_c4_shard64_$tr.json").read().strip().splitlines()[-1])
print("$tr", d["ms_per_step"], d["config"]["parallelism"], d["config"]["cg_iters_per_step"])
PY
done
for tr in peer host; do
  FOS_PEER_LOOPBACK=1 FOS_REDUCTION=$tr python bench.py --steps 50 --no-cpu-baseline --no-raw-instance > gpurun_out/r05_bench_c4_shard64_${tr}_loopback.json 2> gpurun_out/r05_bench_c4_shard64_${tr}_loopback.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_c4_shard64_${tr}_loopback.json").read().strip().splitlines()[-1])
print("$tr loopback", d["ms_per_step"], d["config"]["parallelism"], d["config"]["cg_iters_per_step"])
PY
done
