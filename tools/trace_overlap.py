"""Overlap of the verification kernels of consecutive steps, from a `rocprofv3 --kernel-trace --output-format csv` trace of
bench.py: per kernel name the launch count, mean duration, LDS / VGPR footprint, and how much of the chain kernel's
(k_chain / k_verify_fused) running time had a matching kernel of ANOTHER step running beside it.
usage: python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    for k in ("k_match_split", "k_chain_pnp", "k_chain", "k_verify_fused", "k_nn_filter", "k_nn_refine", "k_spec_pairs", "k_compact"):
        if k in n: return k
    return n[:40]
by = {}
for r in rows:
    by.setdefault(short(r["Kernel_Name"]), []).append(r)
for k, v in sorted(by.items(), key=lambda kv: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1]))[:10]:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v]
    r0 = v[-1]
    print("%-16s n=%4d mean %8.1f us  lds %6s  vgpr %4s  grid %s wg %s" % (k, len(v), sum(d) / len(d) / 1e3, r0.get("LDS_Block_Size"), r0.get("VGPR_Count"), r0.get("Grid_Size_X", r0.get("Grid_Size")), r0.get("Workgroup_Size_X", r0.get("Workgroup_Size"))))
def iv(k): return sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in by.get(k, []))
m = iv("k_match_split") or iv("k_verify_fused")
for ck in ("k_chain", "k_chain_pnp"):
    c = iv(ck)[len(iv(ck)) // 2:]
    if not c: continue
    tot = ov = 0
    for s, e in c:
        tot += e - s
        for a, b in m:
            if b <= s or a >= e: continue
            ov += min(e, b) - max(s, a)
    print("%s: %.1f %% of its running time beside a matching kernel (second half of the trace, %d launches)" % (ck, 100.0 * ov / max(tot, 1), len(c)))
