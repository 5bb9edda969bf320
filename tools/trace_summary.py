"""Steady-state summary of a rocprofv3 kernel trace: `python tools/trace_summary.py <dir with *kernel_trace.csv> [label]` prints, for
the second half of the trace, per kernel: real launches (gated no-ops -- shorter than a quarter of the kernel's longest launch --
dropped), their mean duration, the kernel's share of the span; then the idle share (gaps between consecutive kernels)."""
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


dur = collections.defaultdict(list)
for r in rows:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
idle = sum((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:]))
print("## %s -- second half of the trace: span %.0f us, idle %.1f %%" % (label, span, 100 * idle / span))
print("| kernel | launches | real | mean us (real) | no-op mean us | share of span % |\n|---|---|---|---|---|---|")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    mx = max(v)
    real = [x for x in v if x >= 0.25 * mx]
    noop = [x for x in v if x < 0.25 * mx]
    print("| %s | %d | %d | %.2f | %s | %.1f |" % (k, len(v), len(real), sum(real) / len(real), ("%.2f" % (sum(noop) / len(noop))) if noop else "-", 100 * sum(v) / span))
