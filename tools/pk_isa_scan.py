#!/usr/bin/env python3
"""Lists, per kernel of a csrc translation unit, the packed-f32 instructions (v_pk_mul/add/fma_f32: four passes of 16
lanes on gfx950) that read an SGPR operand which a scalar instruction REWRITES within the next few instructions, and those
whose SGPR operand was written by v_readlane within the previous few -- the two shapes found around the pose composition
of the guided pass in the build that produced the position-dependent pose of round 3.  Runs without
a GPU (hipcc -S).
  tools/pk_isa_scan.py [--tu k_verify.hip] [--slp] [--pre-fix] [--window 4] [--fail] [--strict]
--slp: compile WITHOUT -fno-slp-vectorize; --pre-fix: also without the register barriers of guided_body (the build of
the symptom); --fail: exit 1 when the product flags leave any hit (the build gate tests/test_host_logic.py runs).
--strict (round 5, the invariant the product build holds): list EVERY packed-f32 instruction (v_pk_*_f32) of the
translation unit that has a scalar register among its sources -- not only those whose register is rewritten nearby.  The
mechanism of round 3's symptom was never isolated, so the gate no longer leans on a guess about it: the canonical-
arithmetic translation units are built with -fno-slp-vectorize AND -fno-vectorize, the matcher's explicit packed adds are
gone, and the product flags leave no such instruction at all (0 packed-f32 instructions of any kind in k_verify.hip)."""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multi_robot_slam_separators_amd", "csrc")


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return [int(m.group(1))] if m else []


def scan(asm_path, window, strict=False):
    funcs, cur = collections.OrderedDict(), None
    for ln in open(asm_path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
        elif cur is not None and ln.startswith("\t") and not ln.startswith("\t."):
            cur.append(ln.strip())
    out = []
    for name, ins in funcs.items():
        for i, l in enumerate(ins):
            op = l.split()[0]
            if not re.match(r"v_pk_\w+_f32", op):
                continue
            toks = [t.strip(",") for t in re.split(r"[ ,]+", l)[1:]]
            src = set(r for t in toks[1:] for r in sregs(t))
            if not src:
                continue
            if strict:
                out.append((name, i, "scalar source", l, "-"))
                continue
            for j in range(i + 1, min(len(ins), i + 1 + window)):       # rewritten right behind the read
                o2 = ins[j].split()[0]
                t2 = [t.strip(",") for t in re.split(r"[ ,]+", ins[j])[1:]]
                if o2.startswith(("s_", "v_readlane", "v_readfirstlane")) and t2 and src & set(sregs(t2[0])):
                    out.append((name, i, "rewritten %d later" % (j - i), l, ins[j]))
            for j in range(max(0, i - window), i):                         # written by a lane read right in front
                o2 = ins[j].split()[0]
                t2 = [t.strip(",") for t in re.split(r"[ ,]+", ins[j])[1:]]
                if o2.startswith("v_readlane") and t2 and src & set(sregs(t2[0])):
                    out.append((name, i, "v_readlane %d earlier" % (i - j), l, ins[j]))
    return out, sum(len(v) for v in funcs.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tu", default="k_verify.hip")
    ap.add_argument("--slp", action="store_true")
    ap.add_argument("--pre-fix", action="store_true")
    ap.add_argument("--window", type=int, default=4)
    ap.add_argument("--fail", action="store_true")
    ap.add_argument("--strict", action="store_true")
    a = ap.parse_args()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-pass-failed"]
    if a.tu in ("k_verify.hip", "k_extract.hip", "k_gftt.hip", "k_lk.hip"):
        flags.append("-ffp-contract=off")
        if not (a.slp or a.pre_fix):
            flags.append("-fno-vectorize")      # (csrc/Makefile: CANON)
    if not (a.slp or a.pre_fix):
        flags.append("-fno-slp-vectorize")
    if a.pre_fix:
        flags.append("-DSF_NO_PK_BARRIERS")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "tu.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", os.path.join(CSRC, a.tu), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        hits, n_ins = scan(asm, a.window, a.strict)
    names = subprocess.run(["c++filt"] + [h[0] for h in hits], capture_output=True, text=True).stdout.split("\n")
    print("%s  flags: %s  (%d instructions)" % (a.tu, " ".join(flags[6:]) or "-", n_ins))
    per = collections.Counter()
    for h, n in zip(hits, names):
        short = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
        per[short] += 1
        print("  %-44s @%-6d %-22s %s   <-  %s" % (short[:44], h[1], h[2], h[3], h[4]))
    if a.strict:
        print("  total: %d packed-f32 instructions with a scalar source register" % len(hits))
    else:
        print("  total: %d packed-f32 reads of an SGPR that is rewritten / was lane-read within %d instructions" % (len(hits), a.window))
    if a.fail and hits:
        sys.exit(1)


if __name__ == "__main__":
    main()
