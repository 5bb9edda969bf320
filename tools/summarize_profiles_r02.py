#!/usr/bin/env python3
"""Condense gpurun_out/r02/ (rocprofv3 csv output of tools/gpu_profile_r02.sh) into small tracked summaries under profiles/:
r02_bench_<w>.json, r02_trace_<w>_kernel_stats.md, r02_pmc_<w>.md and r02_kkt_traffic.json (HBM-side bytes per launch of the
CG sweep kernel: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md 'HBM')."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = os.path.join("gpurun_out", rnd), (sys.argv[2] if len(sys.argv) > 2 else "profiles")
os.makedirs(dst, exist_ok=True)
short = lambda name: name.replace("fos::", "").replace("void ", "").split("(")[0][:64]
traffic = {}
for w in ("c4", "c2", "c3", "c5"):
    bj = os.path.join(src, "bench_%s.json" % w)
    if os.path.exists(bj) and os.path.getsize(bj) > 10:
        shutil.copy(bj, os.path.join(dst, "%s_bench_%s.json" % (rnd, w)))
    files = sorted(glob.glob(os.path.join(src, "trace_" + w, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    tf = sorted(glob.glob(os.path.join(src, "trace_" + w, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    steady = {}
    if tf:
        rows = sorted(csv.DictReader(open(tf[-1])), key=lambda r: int(r["Start_Timestamp"]))
        rows = rows[len(rows) // 2:]                       # second half of the run: the timed, steady-state region and its neighbourhood
        d = defaultdict(list)
        for r in rows:
            d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in d.items():
            mx = sorted(v)[len(v) // 2]
            real = [x for x in v if x >= 0.3 * mx]        # launches enqueued past CG convergence are gated no-ops (~2 us)
            steady[k] = (len(real), sum(real) / len(real), len(v) - len(real))
        span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
    if files:
        rows = list(csv.DictReader(open(files[-1])))
        with open(os.path.join(dst, "%s_trace_%s_kernel_stats.md" % (rnd, w)), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats : `python3 bench.py --steps 20 --no-cpu-baseline%s`\n\n" % ("" if w == "c4" else " --workload " + w.upper()))
            f.write("| kernel | calls | total ms | avg us | min us | max us | % | steady-state real launches: n, avg us (gated no-ops) |\n|---|---|---|---|---|---|---|---|\n")
            for r in rows:
                k = short(r["Name"])
                st = steady.get(k)
                f.write("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s | %s |\n" % (k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"], ("%d, %.2f (%d)" % st) if st else ""))
            if tf:
                f.write("\nSecond half of the trace (steady state): span %.1f us, kernels busy %.1f us, idle share %.1f %%\n" % (span, busy, 100.0 * (1 - busy / span)))
    # PMC: per-kernel means of FETCH_SIZE / WRITE_SIZE (KiB per launch)
    pm = defaultdict(lambda: defaultdict(list))
    for c in ("fetch", "write"):
        for fcsv in glob.glob(os.path.join(src, "pmc_%s_%s" % (c, w), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fcsv)):
                pm[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if pm:
        with open(os.path.join(dst, "%s_pmc_%s.md" % (rnd, w)), "w") as f:
            f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) : `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline%s`\n\n"
                    "(the warm-up still runs to the tolerance floor; per-launch means over the REAL launches: gated no-ops move < 1 %% of the median)\n\n"
                    "| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM-side MB per launch = (2 x FETCH + WRITE) x 1024 |\n|---|---|---|---|---|\n" % ("" if w == "c4" else " --workload " + w.upper()))
            for k in sorted(pm, key=lambda k: -sum(pm[k].get("FETCH_SIZE", [0]))):
                fe, wr = pm[k].get("FETCH_SIZE", []), pm[k].get("WRITE_SIZE", [])
                if not fe or not wr:
                    continue
                med = sorted(fe)[len(fe) // 2]
                fe_r = [x for x in fe if x >= 0.3 * med] or fe
                medw = sorted(wr)[len(wr) // 2]
                wr_r = [x for x in wr if x >= 0.3 * medw] or wr
                fm, wm = sum(fe_r) / len(fe_r), sum(wr_r) / len(wr_r)
                mb = (2 * fm + wm) * 1024 / 1e6
                f.write("| %s | %d | %.1f | %.1f | %.2f |\n" % (k, len(fe_r), fm, wm, mb))
                if (k.startswith("kkt2_kernel") or k.startswith("kkt2_win_kernel")) and len(fe_r) > 20:
                    t = traffic.setdefault(w.upper(), {"traffic_bytes": 0.0, "kernels": {}, "source": "profiles/%s_pmc_%s.md" % (rnd, w)})
                    t["kernels"][k] = {"launches": len(fe_r), "fetch_kib": fm, "write_kib": wm, "bytes": mb * 1e6}
for w, t in traffic.items():
    # the CG sweep is the kkt2_kernel instantiation with the most launches
    k = max(t["kernels"], key=lambda k: t["kernels"][k]["launches"])
    t["traffic_bytes"] = t["kernels"][k]["bytes"]
    t["kernel"] = k
if traffic:
    json.dump(traffic, open(os.path.join(dst, "%s_kkt_traffic.json" % rnd), "w"), indent=1)
print(json.dumps(traffic, indent=1))
