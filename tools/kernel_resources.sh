#!/bin/bash
# Compiles one translation unit of csrc/ for gfx950 with -Rpass-analysis=kernel-resource-usage and prints one line
# per kernel: VGPRs, AGPRs, scratch bytes per lane, occupancy, LDS.  Runs without a GPU.
# usage: tools/kernel_resources.sh k_verify.hip [filter-regex]
cd "$(dirname "$0")/../multi_robot_slam_separators_amd/csrc" || exit 1
f=${1:-k_verify.hip}; pat=${2:-.}
canon="-fno-slp-vectorize"; case "$f" in k_verify.hip|k_extract.hip|k_gftt.hip|k_lk.hip) canon="$canon -ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed $canon \
  -c "$f" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = n.split("(")[0].replace("void ", "")
    print("%-44s vgpr %4s agpr %3s sgpr %4s scratch %4s occ %2s lds %6s" % (n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
' | grep -E "$pat"
rm -f /tmp/kres_$$.o
