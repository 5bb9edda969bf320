"""Per-outer-iteration timeline from a rocprofv3 kernel trace: `python tools/step_timeline.py <dir>` -- steps are cut at the PSD kernel;
prints, over the last 30 steps, the mean wall time of a step, the sum of its kernels' durations, and the gaps before / after the PSD kernel
and before the first sweep of the next solve."""
import csv, glob, os, sys
src = sys.argv[1]
tf = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(tf[-1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3, r["Kernel_Name"]) for r in rows]
psd = [i for i, e in enumerate(ev) if "psd64" in e[2] and e[1] - e[0] > 20.0]
psd = psd[-31:]
walls, busy, g_before, g_after, g_next = [], [], [], [], []
for a, b in zip(psd, psd[1:]):
    walls.append(ev[b][0] - ev[a][0])
    busy.append(sum(e[1] - e[0] for e in ev[a:b]))
    g_before.append(ev[b][0] - ev[b - 1][1])
    g_after.append(ev[a + 1][0] - ev[a][1])
    nxt = next((j for j in range(a + 1, b) if "kkt2" in ev[j][2]), None)
    if nxt: g_next.append(ev[nxt][0] - ev[nxt - 1][1])
n = len(walls)
print("steps %d: wall %.1f us, kernels %.1f us, idle %.1f us per step; gap before PSD %.2f, after PSD %.2f, before the next solve's first sweep %.2f" %
      (n, sum(walls) / n, sum(busy) / n, (sum(walls) - sum(busy)) / n, sum(g_before) / n, sum(g_after) / n, sum(g_next) / max(1, len(g_next))))
a, b = psd[-2], psd[-1]
for e in ev[a - 3:a + 8]:
    print("  %10.2f  +%7.2f  %s" % (e[0] - ev[a][0], e[1] - e[0], e[2][:60]))
