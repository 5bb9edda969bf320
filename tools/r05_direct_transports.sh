#!/bin/bash
# DR(direct = true) in the block form on the 64-block shard of C4 (--small: what one of eight ranks holds), in the SHARDED code path on one GPU, under each
# transport of the three scalar sums a projection needs, with (FOS_PEER_LOOPBACK=1) and without the store -> poll latency paid.
# Output: gpurun_out/r05_bench_c4_shard64_direct_{peer,host,rccl}[_loopback].json
export FOS_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() {   # name, env...
  name=$1; shift
  env "$@" python bench.py --small --direct --steps 200 --no-cpu-baseline --no-raw-instance > gpurun_out/r05_bench_c4_shard64_direct_$name.json 2> gpurun_out/r05_bench_c4_shard64_direct_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05_bench_c4_shard64_direct_$name.json").read().strip().splitlines()[-1])
    print("$name", d["ms_per_step"], d["config"]["parallelism"], d["config"]["direct"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r05_bench_c4_shard64_direct_$name.err").read()[-1500:])
PY
}
for tr in peer host rccl; do run $tr FOS_REDUCTION=$tr; done
for tr in peer host; do run ${tr}_loopback FOS_REDUCTION=$tr FOS_PEER_LOOPBACK=1; done
