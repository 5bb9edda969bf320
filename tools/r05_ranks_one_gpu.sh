#!/bin/bash
# The reduced C4 (64 blocks) as 2 and as 8 ranks sharing the ONE GPU of the box, under both kinds of mailboxes (host coordination over gloo: RCCL refuses
# several ranks per device).  Timing is of ranks that compete for one device; what the lines show is that the N-rank flow runs and reproduces the
# single-rank residuals under either transport.  -> gpurun_out/r05final/bench_small_<N>_ranks_one_gpu_<transport>.json
OUT=gpurun_out/r05final; mkdir -p $OUT
P=29600
for N in 2 8; do for tr in peer host; do
  P=$((P+1))
  FOS_REDUCTION=$tr FOS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $P \
    bench.py --gpus $N --small --steps 20 --no-weak-extra 2> $OUT/bench_small_${N}_ranks_one_gpu_$tr.err | tail -1 > $OUT/bench_small_${N}_ranks_one_gpu_$tr.json
  python3 -c "
import json
d=json.loads(open('$OUT/bench_small_${N}_ranks_one_gpu_$tr.json').read())
print($N, '$tr', d['value'], d['ms_per_step'], d['config']['parallelism'], d['config']['residuals_after_run'])"
done; done
python3 bench.py --small --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_small_1_rank.json
python3 -c "
import json
d=json.loads(open('$OUT/bench_small_1_rank.json').read()); print(1, d['value'], d['config']['residuals_after_run'])"
