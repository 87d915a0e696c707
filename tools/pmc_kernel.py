#!/usr/bin/env python3
"""Print per-kernel averages of the counters in a rocprofv3 --pmc output directory.
Usage: python tools/pmc_kernel.py <dir> [kernel substring]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        acc[(m.group(1) if m else r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, c), v in sorted(acc.items()):
    if flt in k:
        print("%-24s %-28s %16.1f  (%d launches)" % (k, c, sum(v) / len(v), len(v)))
