#!/bin/bash
# Round-4 evidence pass (GPU box, via gpurun): bench lines of every configuration, kernel traces of the same commands (steady
# state summaries by tools/trace_summary.py), HBM-traffic / cache / SQ counters of the KKT sweep alone (tools/pmc_sweep.sh:
# separate --pmc passes, no trace domains beside --pmc).  Everything lands in gpurun_out/r04final/ ; copy to profiles/r04_*.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04final
mkdir -p $OUT
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py "$@" 2> $OUT/bench_$tag.err | tail -1 > $OUT/bench_$tag.json; }
run c4 A=1 --
run c2 A=1 -- --workload C2
run c3 A=1 -- --workload C3
run c5 A=1 -- --workload C5 --no-cpu-baseline
run c4_shard64 A=1 -- --small --no-cpu-baseline
run c4_shard64_reference FOS_CG_VARIANT=0 -- --small --no-cpu-baseline
run c4_shard64_dist1rank FOS_FORCE_DIST=1 -- --small --no-cpu-baseline
run c4_shard64_dist1rank_rccl FOS_FORCE_DIST=1 FOS_REDUCTION=rccl -- --small --no-cpu-baseline
run c4_shard64_jacobi FOS_FORCE_DIST=1 FOS_PSD_REFINE=0 -- --small --no-cpu-baseline
run c3_reference FOS_CG_VARIANT=0 -- --workload C3 --no-cpu-baseline
# one rank's shard of an N-rank run of C5 alone on the GPU, in the sharded code path (a projection: no hop between devices is paid)
run c5_shard1of2_dist1rank FOS_FORCE_DIST=1 FOS_BENCH_SHARD=0/2 -- --workload C5 --no-cpu-baseline
run c5_shard1of8_dist1rank FOS_FORCE_DIST=1 FOS_BENCH_SHARD=0/8 -- --workload C5 --no-cpu-baseline
run c4_shard64_dist1rank_unfused FOS_FORCE_DIST=1 FOS_PSD_FUSE=0 -- --small --no-cpu-baseline
# two ranks of the full C4 on the ONE GPU of the box (host coordination over gloo, sums through the peer mailboxes): residuals of the single-rank run.
# (Eight ranks of the full problem cannot share one GPU: a rank's update kernel spins for its peers' mailbox words, and eight such grids do not fit the
# device together -- the exchange times out, as it must; eight ranks run on the reduced problem: tests/test_gpu_bench_flow.py.)
FOS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --no-weak-extra 2> $OUT/bench_c4_two_ranks_one_gpu.err | tail -1 > $OUT/bench_c4_two_ranks_one_gpu.json
for W in c4 c2 c3 c5; do
  extra=""; [ $W != c4 ] && extra="--workload ${W^^}"
  bash tools/r04_trace.sh $W "" $extra > /dev/null 2>&1
  cp gpurun_out/r04/trace_$W.md $OUT/trace_$W.md
done
bash tools/r04_trace.sh c4_shard64_dist1rank "FOS_FORCE_DIST=1" --small > /dev/null 2>&1
cp gpurun_out/r04/trace_c4_shard64_dist1rank.md $OUT/
REPS=20 bash tools/pmc_sweep.sh r04final_pmc C4 C2 C3 C5 > $OUT/pmc.log 2>&1
cp gpurun_out/r04final_pmc/pmc_*.md gpurun_out/r04final_pmc/pmc_*.json $OUT/ 2>/dev/null
timeout 900 python3 tools/psd_time.py 250 64 128 256 512 > $OUT/psd_time.json 2> /dev/null
./scratch/ub_mfma > $OUT/ub_mfma_f64.txt 2>&1 || true
ls $OUT
