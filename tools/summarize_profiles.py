#!/usr/bin/env python3
"""Condense gpurun_out/<round>/ (rocprofv3 csv output) into small tracked summaries under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", rnd)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("fos::", "")
    return name.split("(")[0][:60]


for tag in ("trace_c4", "trace_c2"):
    files = sorted(glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[-1])))
    with open(os.path.join(dst, "%s_%s_kernel_stats.md" % (rnd, tag)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats : `python bench.py --steps 20 --no-cpu-baseline%s`\n\n" % (" --workload C2" if tag.endswith("c2") else ""))
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s |\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                     float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    # steady-state average of the dominant kernel (last 20 x cg launches) from the trace
    tf = sorted(glob.glob(os.path.join(src, tag, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if tf:
        trows = list(csv.DictReader(open(tf[-1])))
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in trows if "kkt2_kernel" in r["Kernel_Name"]]
        ddef = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in trows if "kkt2_deferred_kernel" in r["Kernel_Name"]]
        med = lambda v: sorted(v)[len(v) // 2]
        real = [d for d in durs if d > 0.25 * med(durs)]      # launches enqueued past CG convergence exit at once (gated no-ops, ~3 us)
        tail = real[len(real) // 2:]
        dreal = [d for d in ddef if d > 0.4 * med(ddef)] if ddef else []   # (most launches are real, so the median is a real one)
        bj = os.path.join(src, "bench_%s.json" % tag.split("_")[1])
        ev = None
        if os.path.exists(bj):
            try:
                ev = json.load(open(bj))["roofline"]["avg_kernel_ms"] * 1e3
            except Exception:
                ev = None
        with open(os.path.join(dst, "%s_%s_kernel_stats.md" % (rnd, tag)), "a") as f:
            f.write("\nkkt2_kernel (the sweep): %d launches, %d of them gated no-ops (enqueued past CG convergence, ~3 us each); "
                    "average of the real launches = %.2f us, over the second half of the run (steady state) = %.2f us\n"
                    % (len(durs), len(durs) - len(real), sum(real) / len(real) / 1e3, sum(tail) / len(tail) / 1e3))
            if dreal:
                f.write("kkt2_deferred_kernel (rows spread over dual tiles; second kernel of every KKT apply): average of the real "
                        "launches = %.2f us; one KKT apply = sweep + deferred rows = %.2f us\n"
                        % (sum(dreal) / len(dreal) / 1e3, (sum(tail) / len(tail) + sum(dreal) / len(dreal)) / 1e3))
            if ev:
                f.write("bench.py on the same box, un-profiled, HIP events around the same launches (both kernels of an apply): "
                        "%.2f us\n" % ev)

for tag, counter in (("pmc_fetch_c4", "FETCH_SIZE"), ("pmc_write_c4", "WRITE_SIZE"), ("pmc_fetch_c2", "FETCH_SIZE"), ("pmc_write_c2", "WRITE_SIZE")):
    files = sorted(glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if r.get("Counter_Name") == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, "%s_%s.md" % (rnd, tag)), "w") as f:
        f.write("# rocprofv3 --pmc %s (own pass) : `python bench.py%s --steps 3 --warmup 20 --no-cpu-baseline`\n\n" % (counter, " --workload C2" if tag.endswith("c2") else ""))
        f.write("| kernel | dispatches | mean %s per dispatch (counter units: KB) | as MB |\n|---|---|---|---|\n" % counter)
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            if "kkt2" in k or "cg_" in k:      # drop gated no-op dispatches (counter ~0)
                v = [x for x in v if x > 0.25 * max(v)] or v
            f.write("| %s | %d | %.1f | %.2f |\n" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / 1024))
        f.write("\n(gated no-op dispatches of the CG kernels -- enqueued past convergence -- are excluded from the means)\n")

# measured HBM traffic of the dominant kernel per launch, for bench.py's roofline.traffic (2 x FETCH_SIZE: the gfx950
# under-count for coalesced streams, MI355X_MICROARCH.md "HBM"; + WRITE_SIZE), keyed by workload
traffic = {}
for wl in ("c4", "c2"):
    vals = {}
    for tag, counter in (("pmc_fetch_" + wl, "FETCH_SIZE"), ("pmc_write_" + wl, "WRITE_SIZE")):
        files = sorted(glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not files:
            continue
        tot = 0.0
        for kname in ("kkt2_kernel", "kkt2_deferred_kernel"):      # one KKT apply = sweep + deferred-row kernel
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[-1])) if r.get("Counter_Name") == counter and kname in r["Kernel_Name"]]
            v = [x for x in v if x > 0.25 * max(v)] if v else []
            if v:
                tot += sum(v) / len(v) * 1024.0
        vals[counter] = tot
    if len(vals) == 2:
        traffic[wl.upper()] = {"fetch_size_bytes_reported": vals["FETCH_SIZE"], "write_size_bytes_reported": vals["WRITE_SIZE"],
                               "traffic_bytes": 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"],
                               "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_profile.sh); "
                                      "traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 1/2 of coalesced stream reads)"}
if traffic:
    json.dump(traffic, open(os.path.join(dst, "%s_kkt_traffic.json" % rnd), "w"), indent=1)

for b in ("bench_c4", "bench_c2", "bench_c3", "bench_c5"):
    p = os.path.join(src, b + ".json")
    if os.path.exists(p):
        txt = open(p).read().strip()
        try:
            json.loads(txt)
            open(os.path.join(dst, "%s_%s.json" % (rnd, b)), "w").write(txt + "\n")
        except Exception:
            open(os.path.join(dst, "%s_%s.FAILED.txt" % (rnd, b)), "w").write(txt + "\n")
print("\n".join(sorted(os.listdir(dst))))
