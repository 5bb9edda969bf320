#!/bin/bash
# PMC passes on the KKT sweep alone (tools/kkt_only.py): wave-time breakdown and cache behaviour.
# (No TA_* counters: that pass aborted and hung a box of this pool; every pass runs under its own time-out.)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
WL=${1:-C4}; WG=${2:-3072}
OUT=gpurun_out/pmc_kkt_$WL
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/kkt_only.py $WL $WG 20 > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kkt2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(acc):
        v = acc[k]
        line = "%-40s n=%3d mean=%.6g" % (k, len(v), sum(v)/len(v))
        print(line); o.write(line + "\n")
PY
