#!/bin/bash
# per-kernel durations of the CG chain bench: bash tools/trace_chain.sh <workload> <fuse 0|1> [wg]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
WL=${1:-C4}; export FUSE=${2:-1}; export GRAPH=0; WG=${3:-0}
OUT=gpurun_out/trc_${WL}_f${FUSE}_w${WG}
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 tools/chain_bench.py $WL 20 $WG > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v); med = v2[len(v2) // 2]
    print("%-72s n=%5d median %8.2f us  mean %8.2f" % (k, len(v), med, sum(v) / len(v)))
PY
