#!/usr/bin/env python3
"""Per-launch averages of one kernel's SQ counters from rocprofv3 --pmc output directories -> profiles/<tag>_sq_*.json.
Usage: python tools/sq_json.py <tag> <kernel> <sq_dir> [<sq_waits_dir>]

<sq_dir>: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE.  SIMD-cycles available = 1024 SIMDs x GRBM_GUI_ACTIVE / 8 (the counter is summed
over the 8 XCDs); valu_busy = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / SIMD-cycles; mfma_busy =
SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles.
<sq_waits_dir>: SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES (+ LDS / SALU instruction counts):
the three disjoint wavefront-cycle buckets parked / issue-stalled / issuing as fractions of SQ_WAVE_CYCLES."""
import csv, glob, json, os, sys
from collections import defaultdict


def averages(dirname, kernel):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in sorted(acc.items())}, (max(map(len, acc.values())) if acc else 0)


tag, kernel, sq = sys.argv[1], sys.argv[2], sys.argv[3]
import re
fname = re.sub(r"_+$", "", re.sub(r"[^A-Za-z0-9]+", "_", kernel))      # ("k_ba_pass<1, true, true>" -> k_ba_pass_1_true_true)
d, n = averages(sq, kernel)
if d:
    simd_cycles = 1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0
    d["valu_busy_estimate"] = d["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
    d["mfma_busy_estimate"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    d["_note"] = "%s, averages over %d launches of one rocprofv3 --pmc pass (tools/profile_round.sh)" % (kernel, n)
    json.dump(d, open("profiles/%s_sq_%s.json" % (tag, fname), "w"), indent=1)
if len(sys.argv) > 4:
    w, n = averages(sys.argv[4], kernel)
    if w:
        wc = w["SQ_WAVE_CYCLES"]
        w["frac_wait_any (parked: s_waitcnt / barrier)"] = w["SQ_WAIT_ANY"] / wc
        w["frac_wait_inst_any (issue stall)"] = w["SQ_WAIT_INST_ANY"] / wc
        w["frac_active_inst_any"] = w["SQ_ACTIVE_INST_ANY"] / wc
        w["_note"] = "%s, wavefront-cycle accounting (quad-cycles), averages over %d launches" % (kernel, n)
        json.dump(w, open("profiles/%s_sq_waits_%s.json" % (tag, fname), "w"), indent=1)
