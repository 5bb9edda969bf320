#!/bin/bash
# Counter evidence for the KKT sweep alone (tools/kkt_only.py = fos_bench_kkt: back-to-back KKT applies on a random vector).
# usage (on the GPU box):  bash tools/pmc_sweep.sh <tag> C3 C5 ...      -> gpurun_out/<tag>/pmc_<workload>.md
# One rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --pmc), each
# under its own time-out; a plain --kernel-trace run gives the duration the byte counts are divided by.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-pmc}; shift
REPS=${REPS:-20}
PROG=${PROG:-tools/kkt_only.py}     # the program profiled; KPAT: which kernels are tabulated (tools/stream_only.py + KPAT=cg_stream: the streamed resident solve)
export KPAT=${KPAT:-kkt2}
export PROG
for WL in "$@"; do
  OUT=gpurun_out/$TAG/$WL
  rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $PROG $WL 0 $REPS > $OUT/trace.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $PROG $WL 0 $REPS > $OUT/p$i.log 2>&1
  done
  python3 - "$OUT" "$WL" "$TAG" <<'PY'
import csv, glob, collections, sys, os
out, wl, tag = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ.get("KPAT", "kkt2") in k:
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ.get("KPAT", "kkt2") in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["# %s: kernels matching '%s' of `python3 %s %s`, rocprofv3 --pmc, one pass per counter group" % (wl, os.environ.get("KPAT", "kkt2"), os.environ.get("PROG", "tools/kkt_only.py"), wl), "",
         open(out + "/trace.log").read().strip().splitlines()[-1] if os.path.exists(out + "/trace.log") else "", ""]
for k in sorted(acc):
    d = dur.get(k, [])
    d = sorted(d)[len(d) // 4: max(len(d) // 4 + 1, 3 * len(d) // 4)] if d else []
    t_us = sum(d) / len(d) if d else float("nan")
    lines += ["## %s   (kernel-trace duration, interquartile mean: %.2f us)" % (k, t_us), "", "| counter | launches | mean per launch |", "|---|---|---|"]
    m = {}
    for c in sorted(acc[k]):
        v = acc[k][c]
        v = v[1:] if len(v) > 2 else v                        # first launch: cold caches
        m[c] = sum(v) / len(v)
        lines.append("| %s | %d | %.6g |" % (c, len(v), m[c]))
    lines.append("")
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rd, wr = 2 * m["FETCH_SIZE"] * 1024, m["WRITE_SIZE"] * 1024
        lines.append("HBM-side traffic per launch = 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE "
                     "= %.2f MB read + %.2f MB written = %.2f MB -> %.0f GB/s" % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6, (rd + wr) / t_us / 1e3))
    if "TCC_HIT_sum" in m:
        lines.append("L2 hit rate = %.3f ; L2 requests per launch %.4g ; TCP->TCC read requests %.4g (%.1f G/s)" %
                     (m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), m.get("TCC_REQ_sum", float("nan")),
                      m.get("TCP_TCC_READ_REQ_sum", float("nan")), m.get("TCP_TCC_READ_REQ_sum", float("nan")) / t_us / 1e3))
    lines.append("")
open("gpurun_out/%s/pmc_%s.md" % (tag, wl), "w").write("\n".join(lines) + "\n")
import json
js = {}
for k in sorted(acc):
    mm = {c: (lambda v: sum(v[1:]) / len(v[1:]) if len(v) > 2 else sum(v) / len(v))(acc[k][c]) for c in acc[k]}
    d = sorted(dur.get(k, []))
    d = d[len(d) // 4: max(len(d) // 4 + 1, 3 * len(d) // 4)] if d else []
    if "FETCH_SIZE" in mm and "WRITE_SIZE" in mm:
        js[k.replace("fos::", "").replace("void ", "").strip()] = {"fetch_kib": mm["FETCH_SIZE"], "write_kib": mm["WRITE_SIZE"], "bytes": (2 * mm["FETCH_SIZE"] + mm["WRITE_SIZE"]) * 1024,
                 "kernel_trace_us": (sum(d) / len(d)) if d else None}
json.dump(js, open("gpurun_out/%s/pmc_%s.json" % (tag, wl), "w"), indent=1)
print("\n".join(lines))
PY
  rm -rf $OUT
done
