#!/bin/bash
# Round-6 evidence pass (GPU box, via gpurun): bench lines of every configuration (C4 with the raw instance and DR(direct=true) beside the headline),
# the 64-block problem one of eight ranks holds under each transport of the scalar sums (device mailboxes, host-pinned mailboxes, in-stream RCCL;
# with and without the rank's own words travelling through the mailbox: FOS_PEER_LOOPBACK), kernel traces of the same commands, HBM-traffic /
# cache / SQ counters of the KKT sweep alone (tools/pmc_sweep.sh: separate --pmc passes).  Everything lands in gpurun_out/r06final/ ;
# `python tools/collect_profiles_r06.py` copies it to profiles/r06_*.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06final
mkdir -p $OUT
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py "$@" 2> $OUT/bench_$tag.err | tail -1 > $OUT/bench_$tag.json; }
run c4 A=1 --
run c2 A=1 -- --workload C2
run c3 A=1 -- --workload C3
run c5 A=1 -- --workload C5
run c4_shard64 A=1 -- --small --no-cpu-baseline
for tr in peer host rccl; do
  run c4_shard64_$tr FOS_FORCE_DIST=1 FOS_REDUCTION=$tr -- --small --no-cpu-baseline
done
for tr in peer host; do
  run c4_shard64_${tr}_loopback FOS_FORCE_DIST=1 FOS_REDUCTION=$tr FOS_PEER_LOOPBACK=1 -- --small --no-cpu-baseline
done
# the same shard with one launch group per CG iteration (the merged recurrence of rounds 3-5) instead of the resident solve: same box, same build
for tr in peer host; do
  run c4_shard64_${tr}_launch_per_iteration FOS_FORCE_DIST=1 FOS_REDUCTION=$tr FOS_RESIDENT_DEFAULT=0 -- --small --no-cpu-baseline
done
run c4_shard64_direct A=1 -- --small --no-cpu-baseline --direct
# what one of TWO / FOUR ranks holds (256 / 128 blocks: the streamed form of the resident solve), sharded code path, one rank
run c4_shard1of2_dist1rank FOS_FORCE_DIST=1 FOS_REDUCTION=peer FOS_BENCH_SHARD=0/2 -- --no-cpu-baseline
run c4_shard1of4_dist1rank FOS_FORCE_DIST=1 FOS_REDUCTION=peer FOS_BENCH_SHARD=0/4 -- --no-cpu-baseline
# C4 on one GPU with the launch-per-iteration recurrences as the measured configuration (FOS_RESIDENT_DEFAULT=0: the headline of rounds 1-5)
run c4_launch_per_iteration FOS_RESIDENT_DEFAULT=0 -- --no-cpu-baseline
# DR(direct = true), block form, in the SHARDED code path (three scalar exchanges per projection through the transport)
for tr in peer host rccl; do
  run c4_shard64_direct_sharded_$tr FOS_FORCE_DIST=1 FOS_REDUCTION=$tr -- --small --no-cpu-baseline --direct --steps 200
done
run c5_shard1of2_dist1rank FOS_FORCE_DIST=1 FOS_BENCH_SHARD=0/2 -- --workload C5 --no-cpu-baseline
run c5_shard1of8_dist1rank FOS_FORCE_DIST=1 FOS_BENCH_SHARD=0/8 -- --workload C5 --no-cpu-baseline
# two ranks of the full C4 on the ONE GPU of the box under both kinds of mailboxes (host coordination over gloo): residuals of the single-rank run
for tr in peer host; do
  FOS_REDUCTION=$tr FOS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --no-weak-extra 2> $OUT/bench_c4_two_ranks_one_gpu_$tr.err | tail -1 > $OUT/bench_c4_two_ranks_one_gpu_$tr.json
done
for W in c4 c2 c3 c5; do
  extra=""; [ $W != c4 ] && extra="--workload ${W^^}"
  STEPS=$([ $W = c4 ] && echo 300 || echo 20) bash tools/r04_trace.sh $W "" $extra --no-direct-extra > /dev/null 2>&1
  cp gpurun_out/r04/trace_$W.md $OUT/trace_$W.md
  grep -a '^{"metric"' gpurun_out/r04/trace_$W.log | tail -1 > $OUT/bench_${W}_trace_run.json      # the bench line of the traced command itself
done
bash tools/r04_trace.sh c4_direct "" --direct > /dev/null 2>&1
cp gpurun_out/r04/trace_c4_direct.md $OUT/
bash tools/r04_trace.sh c4_shard64_peer "FOS_FORCE_DIST=1 FOS_REDUCTION=peer" --small > /dev/null 2>&1
cp gpurun_out/r04/trace_c4_shard64_peer.md $OUT/
bash tools/r04_trace.sh c4_shard64_host "FOS_FORCE_DIST=1 FOS_REDUCTION=host" --small > /dev/null 2>&1
cp gpurun_out/r04/trace_c4_shard64_host.md $OUT/
REPS=20 bash tools/pmc_sweep.sh r06final_pmc C4 C2 C3 C5 > $OUT/pmc.log 2>&1
# the streamed resident solve alone: HBM-side bytes per launch of 22 sweeps, counters, kernel-trace duration
PROG=tools/stream_only.py KPAT=cg_stream REPS=8 bash tools/pmc_sweep.sh r06final_stream C4 > $OUT/pmc_stream.log 2>&1
cp gpurun_out/r06final_stream/pmc_C4.md $OUT/pmc_c4_stream.md; cp gpurun_out/r06final_stream/pmc_C4.json $OUT/pmc_c4_stream.json
# in-kernel stamps of the streamed solve on the whole of C4 (the stamps build of the library travels with the tree when it was built: make VARIANT=res_stamps EXTRA=-DFOS_RES_STAMPS)
[ -f firstordersolvers.jl_amd/csrc/libfoship_res_stamps.so ] && FOSHIP_LIB=firstordersolvers.jl_amd/csrc/libfoship_res_stamps.so timeout 300 python3 tools/res_stamps.py 512 > $OUT/stream_stamps.txt 2>&1
cp gpurun_out/r06final_pmc/pmc_*.md gpurun_out/r06final_pmc/pmc_*.json $OUT/ 2>/dev/null
timeout 900 python3 tools/psd_time.py 250 64 128 256 512 > $OUT/psd_time.json 2> /dev/null
timeout 900 python3 tools/psd_orders.py > $OUT/psd_orders_final.txt 2> /dev/null
# (in-kernel stamps need their own builds: make VARIANT=res_stamps EXTRA=-DFOS_RES_STAMPS; FOSHIP_LIB=.../libfoship_res_stamps.so python tools/res_stamps.py -> profiles/r06_res_stamps.txt)
ls $OUT
