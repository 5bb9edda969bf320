#!/bin/bash
# Counter evidence for the batched PSD(64) projection kernels: bash tools/pmc_psd.sh <tag> [blocks=64]  -> gpurun_out/<tag>/pmc_psd.md
# (tools/psd_time.py as the driver; one rocprofv3 run per counter group, no trace domains beside --pmc; a kernel-trace run for the durations)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-pmc_psd}; NB=${2:-64}
OUT=gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/psd_time.py 250 $NB > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/psd_time.py 250 $NB > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$NB" <<'PY'
import csv, glob, collections, sys
out, nb = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "psd" in k:
            acc[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "psd" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["# Batched PSD(64) projection, %d matrices per launch (`tools/psd_time.py 250 %s` under rocprofv3 --pmc, one pass per counter group)" % (2 * int(nb), nb), ""]
for k in sorted(acc):
    d = sorted(x for x in dur.get(k, []) if x > 20.0)
    d = d[len(d) // 4: max(len(d) // 4 + 1, 3 * len(d) // 4)] if d else []
    t_us = sum(d) / len(d) if d else float("nan")
    lines += ["## %s   (kernel-trace duration of the full launches, interquartile mean: %.2f us)" % (k, t_us), "", "| counter | launches | mean per launch (full launches: the upper half by SQ_WAVE_CYCLES) |", "|---|---|---|"]
    m = {}
    for c in sorted(acc[k]):
        v = sorted(acc[k][c]); v = v[len(v) // 2:]            # truncated diagnostic launches of the driver are the lower half
        m[c] = sum(v) / len(v)
        lines.append("| %s | %d | %.6g |" % (c, len(v), m[c]))
    lines.append("")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        lines.append("MFMA busy cycles / SQ busy cycles = %.3f ; MFMA instructions per launch %.4g" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, m["SQ_BUSY_CYCLES"]), m.get("SQ_INSTS_MFMA", float("nan"))))
    if "SQ_LDS_BANK_CONFLICT" in m:
        lines.append("LDS bank-conflict cycles / LDS active cycles = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1.0))))
    if "SQ_WAIT_ANY" in m:
        lines.append("wave cycles: waiting %.2f, issue stall %.2f, active %.2f of SQ_WAVE_CYCLES" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        lines.append("HBM-side traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE = %.2f MB" % ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e6))
    lines.append("")
open(out + "/pmc_psd.md", "w").write("\n".join(lines))
print("\n".join(lines))
PY
