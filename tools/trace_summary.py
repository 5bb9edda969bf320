"""Steady-state summary of a rocprofv3 kernel trace: `python tools/trace_summary.py <dir with *kernel_trace.csv> [label]` prints, for
the second half of the trace, per kernel: real launches (gated no-ops dropped), their mean duration, the kernel's share of the span;
then the idle share (gaps between consecutive kernels).  Gated no-ops (launches enqueued past CG convergence: they return at their
first instruction) are told from real launches by the best two-way split of the log-durations, accepted only when the two groups' means
differ by 1.7x or more and the short group holds 1 % of the launches at least -- a fixed fraction of the LONGEST launch (the first version)
called every real 5-us launch of a kernel with one 27-us outlier a no-op."""
import math
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else src
tf = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(tf[-1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]


def short(n):
    n = re.sub(r"^void ", "", n)
    m = re.match(r"(?:fos::)?([A-Za-z0-9_]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


def split_noops(v):
    if len(v) < 8:
        return list(v), []
    lv = sorted(math.log(max(x, 1e-3)) for x in v)
    n = len(lv)
    pre = [0.0]
    pre2 = [0.0]
    for x in lv:
        pre.append(pre[-1] + x)
        pre2.append(pre2[-1] + x * x)
    best, cut = None, None
    for k in range(max(1, n // 100), n - max(1, n // 100)):          # both groups: 1 % of the launches at least
        sse = (pre2[k] - pre[k] ** 2 / k) + ((pre2[n] - pre2[k]) - (pre[n] - pre[k]) ** 2 / (n - k))
        if best is None or sse < best:
            best, cut = sse, k
    if cut is None:
        return list(v), []
    lo, hi = math.exp(pre[cut] / cut), math.exp((pre[n] - pre[cut]) / (n - cut))
    if hi < 1.7 * lo:
        return list(v), []
    thr = math.exp(0.5 * (lv[cut - 1] + lv[cut]))
    return [x for x in v if x >= thr], [x for x in v if x < thr]


dur = collections.defaultdict(list)
for r in rows:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
idle = sum((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:]))
print("## %s -- second half of the trace: span %.0f us, idle %.1f %%" % (label, span, 100 * idle / span))
print("| kernel | launches | real | mean us (real) | no-op mean us | share of span % |\n|---|---|---|---|---|---|")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    real, noop = split_noops(v)
    print("| %s | %d | %d | %.2f | %s | %.1f |" % (k, len(v), len(real), sum(real) / len(real), ("%.2f" % (sum(noop) / len(noop))) if noop else "-", 100 * sum(v) / span))
