#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (kernel stats + PMC passes) into profiles/<tag>_*.{csv,json}.
Usage: python tools/summarize_prof.py <tag> <stats_dir> [<fetch_dir> <write_dir>]"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

def short(name):
    m = re.search(r"(k_[a-z0-9_]+(?:<[0-9a-z, ]+>)?)", name)
    return m.group(1) if m else name[:40]

def pmc(dirname):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

tag, stats = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
out = {"kernels": []}
for f in glob.glob(os.path.join(stats, "**", "*_kernel_stats.csv"), recursive=True):
    shutil.copy(f, "profiles/%s_kernel_stats.csv" % tag)
    for r in csv.DictReader(open(f)):
        out["kernels"].append({"kernel": short(r["Name"]), "calls": int(r["Calls"]),
                               "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])})
# per-launch durations from the kernel trace: median and the mean without the first quarter (warm-up launches)
dur = defaultdict(list)
for f in glob.glob(os.path.join(stats, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k in out["kernels"]:
    d = [x[1] for x in sorted(dur.get(k["kernel"], []))]
    if d:
        k["median_us"] = sorted(d)[len(d) // 2] / 1e3
        tail = d[len(d) // 4:]
        k["avg_after_warmup_us"] = sum(tail) / len(tail) / 1e3
if len(sys.argv) >= 5:
    fe, wr = pmc(sys.argv[3]), pmc(sys.argv[4])
    out["pmc_per_launch"] = {}
    for (k, c), (v, n) in sorted(list(fe.items()) + list(wr.items())):
        # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section)
        out["pmc_per_launch"].setdefault(k, {})[c + "_KiB"] = v
        out["pmc_per_launch"][k]["launches"] = n
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
for k in out["kernels"]:
    print("%-22s calls %3d avg %10.1f us  %5.1f %%  median %8.1f us  after warm-up %8.1f us" % (
        k["kernel"], k["calls"], k["avg_us"], k["pct"], k.get("median_us", 0.0), k.get("avg_after_warmup_us", 0.0)))
for k, v in out.get("pmc_per_launch", {}).items():
    print(k, v)
