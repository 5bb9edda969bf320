#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines + rocprofv3 kernel stats of the same command + HBM traffic counters of the
# dominant kernel (separate --pmc passes for FETCH_SIZE and WRITE_SIZE; no trace domains beside --pmc).
# Outputs land in gpurun_out/r02/ ; tools/summarize_profiles_r02.py condenses them into profiles/r02_*.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02
mkdir -p $OUT
WLS=${WLS:-"C4 C2 C3 C5"}
for W in $WLS; do
  w=$(echo $W | tr A-Z a-z)
  extra=""; [ $W != C4 ] && extra="--workload $W"
  [ $W = C5 ] && extra="$extra --no-cpu-baseline"
  python3 bench.py $extra 2> $OUT/bench_$w.err | tail -1 > $OUT/bench_$w.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 bench.py --steps 20 --no-cpu-baseline $extra > $OUT/trace_$w.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > $OUT/pmc_fetch_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > $OUT/pmc_write_$w.log 2>&1
done
# condense on the box (the raw rocprofv3 output exceeds what gpurun copies back) and drop the raw files
python3 tools/summarize_profiles_r02.py r02 gpurun_out/r02_summary > $OUT/summary.log 2>&1
tail -30 $OUT/summary.log
for W in $WLS; do w=$(echo $W | tr A-Z a-z); rm -rf $OUT/trace_$w $OUT/pmc_fetch_$w $OUT/pmc_write_$w; done
ls gpurun_out/r02_summary
