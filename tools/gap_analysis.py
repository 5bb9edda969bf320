"""Where the idle time between kernels sits: `python tools/gap_analysis.py <dir with a rocprofv3 *kernel_trace.csv>` prints, for the
second half of the trace (steady state), the mean gap between the end of one kernel and the start of the next, grouped by the
pair of kernel names, and each pair's share of the total idle time."""
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
tf = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(tf[-1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^.*fos::", "", n)).replace("void ", "")
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
total = sum(sum(v) for v in gaps.values())
print("span %.0f us, idle %.0f us (%.1f %%)" % (span, total, 100 * total / span))
print("| after -> before | n | mean gap us | share of idle % |\n|---|---|---|---|")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("| %s -> %s | %d | %.2f | %.1f |" % (k[0], k[1], len(v), sum(v) / len(v), 100 * sum(v) / total))
