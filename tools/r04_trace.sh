#!/bin/bash
# usage (GPU box): bash tools/r04_trace.sh <tag> "<ENV=.. ENV=..>" <bench.py flags...>   -> gpurun_out/r04/trace_<tag>.md
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; envs=$2; shift 2
OUT=gpurun_out/r04; mkdir -p $OUT
D=/tmp/tr_$tag; rm -rf $D
for kv in $envs; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps ${STEPS:-20} --no-cpu-baseline --no-raw-instance --no-recurrence-extras "$@" > $OUT/trace_$tag.log 2>&1
python3 tools/trace_summary.py $D "$tag ($envs $*)" > $OUT/trace_$tag.md 2>&1
cat $OUT/trace_$tag.md
for kv in $envs; do unset "${kv%%=*}"; done
rm -rf $D
