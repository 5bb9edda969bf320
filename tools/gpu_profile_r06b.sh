#!/bin/bash
# The lines of the round-6 pass that run the STREAMED resident solve, once more with the final bench.py (algorithmic bytes = iterations + 1 sweeps), and the
# C4 trace with the bench line of the traced command: into gpurun_out/r06final/ beside the rest of the pass (tools/gpu_profile_r06.sh).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06final
mkdir -p $OUT
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py "$@" 2> $OUT/bench_$tag.err | tail -1 > $OUT/bench_$tag.json; }
run c4 A=1 --
run c4_launch_per_iteration FOS_RESIDENT_DEFAULT=0 -- --no-cpu-baseline
run c4_shard1of2_dist1rank FOS_FORCE_DIST=1 FOS_REDUCTION=peer FOS_BENCH_SHARD=0/2 -- --no-cpu-baseline
run c4_shard1of4_dist1rank FOS_FORCE_DIST=1 FOS_REDUCTION=peer FOS_BENCH_SHARD=0/4 -- --no-cpu-baseline
run c4_shard64_peer FOS_FORCE_DIST=1 FOS_REDUCTION=peer -- --small --no-cpu-baseline
for tr in peer host; do
  FOS_REDUCTION=$tr FOS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --no-weak-extra 2> $OUT/bench_c4_two_ranks_one_gpu_$tr.err | tail -1 > $OUT/bench_c4_two_ranks_one_gpu_$tr.json
done
STEPS=300 bash tools/r04_trace.sh c4 "" --no-direct-extra > /dev/null 2>&1
cp gpurun_out/r04/trace_c4.md $OUT/trace_c4.md
grep -a '^{"metric"' gpurun_out/r04/trace_c4.log | tail -1 > $OUT/bench_c4_trace_run.json
PROG=tools/stream_only.py KPAT=cg_stream REPS=8 bash tools/pmc_sweep.sh r06final_stream C4 > $OUT/pmc_stream.log 2>&1
cp gpurun_out/r06final_stream/pmc_C4.md $OUT/pmc_c4_stream.md; cp gpurun_out/r06final_stream/pmc_C4.json $OUT/pmc_c4_stream.json
[ -f firstordersolvers.jl_amd/csrc/libfoship_res_stamps.so ] && FOSHIP_LIB=firstordersolvers.jl_amd/csrc/libfoship_res_stamps.so timeout 300 python3 tools/res_stamps.py 512 > $OUT/stream_stamps.txt 2>&1
ls $OUT | wc -l
