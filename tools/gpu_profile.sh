#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines + rocprofv3 kernel stats + HBM traffic counters.
# Outputs land in gpurun_out/r01/ ; summaries are copied into profiles/ by tools/summarize_profiles.py.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r01
mkdir -p $OUT
python bench.py                      2>&1 | tail -1 > $OUT/bench_c4.json
python bench.py --workload C2        2>&1 | tail -1 > $OUT/bench_c2.json
python bench.py --workload C3 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_c3.json
python bench.py --workload C5 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_c5.json
# kernel trace + stats of the SAME command as the default bench (fewer steps, no CPU leg)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4 -- python bench.py --steps 20 --no-cpu-baseline > $OUT/trace_c4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- python bench.py --workload C2 --steps 20 --no-cpu-baseline > $OUT/trace_c2.log 2>&1
# HBM traffic of the dominant kernel: separate passes for FETCH_SIZE and WRITE_SIZE (TCC slots), counters only
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c4 -- python bench.py --steps 3 --warmup 20 --no-cpu-baseline > $OUT/pmc_fetch_c4.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c4 -- python bench.py --steps 3 --warmup 20 --no-cpu-baseline > $OUT/pmc_write_c4.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c2 -- python bench.py --workload C2 --steps 3 --warmup 20 --no-cpu-baseline > $OUT/pmc_fetch_c2.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c2 -- python bench.py --workload C2 --steps 3 --warmup 20 --no-cpu-baseline > $OUT/pmc_write_c2.log 2>&1
ls -R $OUT | head -50
