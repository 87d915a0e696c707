"""Where a bench step's wall time goes on the device timeline: from a rocprofv3 kernel trace (csv) of `bench.py`,
the idle gaps between consecutive kernels of the steady-state steps, keyed by (previous kernel -> next kernel).
usage: python tools/step_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:32]


rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
# steady state: the last 60 % of the k_verify_fused launches
fused = [i for i, r in enumerate(rows) if r[2] == "k_verify_fused"]
lo = fused[int(len(fused) * 0.4)]
rows = rows[lo:]
gaps = defaultdict(list)
busy_end = rows[0][1]
for prev, cur in zip(rows, rows[1:]):
    gap = cur[0] - busy_end                      # idle time of the device before `cur` starts (negative: overlap)
    if gap > 0:
        gaps[(prev[2], cur[2])].append(gap / 1e3)
    busy_end = max(busy_end, cur[1])
n_steps = sum(1 for r in rows if r[2] == "k_verify_fused")
span = (rows[-1][1] - rows[0][0]) / 1e3
print("%d steps, %.1f us per step on the device timeline" % (n_steps, span / n_steps))
tot = 0.0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    per = sum(v) / n_steps
    tot += per
    print("  idle %6.1f us/step  (%4d x mean %5.1f us)  %s -> %s" % (per, len(v), sum(v) / len(v), k[0], k[1]))
print("  idle total %.1f us/step" % tot)
