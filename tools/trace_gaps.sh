#!/bin/bash
# Kernel trace of one bench run with per-kernel averages of the REAL launches (gated no-ops dropped) and inter-kernel gaps.
# usage (on the GPU box): bash tools/trace_gaps.sh C4|C2|C3|C5 [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tr
rm -rf $OUT; mkdir -p $OUT
W=${1:-C4}
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python bench.py --steps 20 --no-cpu-baseline --workload $W $2 > $OUT/t.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("%-72s %7s %9s %9s"%("kernel","n_real","avg_us","tot_ms"))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:12]:
    mx=max(v); real=[x for x in v if x>=0.25*mx]
    print("%-72s %7d %9.2f %9.2f"%(k,len(real),sum(real)/len(real)/1e3,sum(v)/1e6))
# gap between kkt2 sweep end and deferred start
gaps=[]; gaps2=[]
for a,b in zip(rows,rows[1:]):
    if "kkt2_kernel" in a["Kernel_Name"] and "kkt2_deferred" in b["Kernel_Name"]:
        g=int(b["Start_Timestamp"])-int(a["End_Timestamp"]); da=int(a["End_Timestamp"])-int(a["Start_Timestamp"])
        if da>20000: gaps.append(g)
    if "kkt2_deferred" in a["Kernel_Name"] and "cg_alpha" in b["Kernel_Name"]:
        gaps2.append(int(b["Start_Timestamp"])-int(a["End_Timestamp"]))
if gaps: print("gap sweep->deferred avg us", sum(gaps)/len(gaps)/1e3, "deferred->alpha", sum(gaps2)/max(1,len(gaps2))/1e3)
# last 20 steps wall
PY
