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


def vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_mfma_asm(asm_path, states=12):
    """Round 5: the matcher's pipelined scan issues its MFMAs inside asm statements (k_match.hip, mf_pipe_*), where the
    compiler neither sees the matrix instruction nor pads for it.  An MFMA result may be touched by anything but the next
    accumulating MFMA only `states` wait states later (8-pass XDL: 12); the statements keep that distance for their own
    readers.  What they cannot control is COMPILER code behind the statement: a register copy, a spill or an address
    computation allocated into a result register.  This lists every instruction outside an asm statement that reads or
    writes a register of a hand-issued MFMA's result tuple within `states` instructions of that MFMA (each instruction
    counted as one state, s_nop N as N + 1, conservatively ignoring the real issue time of the MFMAs in between)."""
    funcs, cur = collections.OrderedDict(), None
    for ln in open(asm_path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
        elif cur is not None:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                cur.append("#S")
            elif t.startswith(";;#ASMEND"):
                cur.append("#E")
            elif ln.startswith("\t") and not ln.startswith("\t.") and not t.startswith(";"):
                cur.append(t.split(";")[0].strip())
    out, n_mfma = [], 0
    for name, ins in funcs.items():
        in_asm, pending = False, []          # pending: [result registers, states since the MFMA]
        for l in ins:
            if l == "#S":
                in_asm = True; continue
            if l == "#E":
                in_asm = False; continue
            op = l.split()[0]
            toks = [t.strip(",") for t in re.split(r"[ ,]+", l)[1:]]
            regs = set()
            for t in toks:
                regs |= vregs(t)
            if not in_asm:
                for dst, st in pending:
                    if st < states and regs & dst and not op.startswith("v_mfma"):
                        out.append((name, st, "compiler code on a pending MFMA result", l, "v[%d:%d]" % (min(dst), max(dst))))
            step = (int(toks[0]) + 1) if op == "s_nop" and toks and toks[0].isdigit() else 1
            pending = [[d, st + step] for d, st in pending if st + step < states]
            if in_asm and op.startswith("v_mfma") and toks:
                n_mfma += 1
                pending = [[d, st] for d, st in pending if d != vregs(toks[0])]
                pending.append([vregs(toks[0]), 0])
    return out, n_mfma


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tu", default="k_verify.hip")
    ap.add_argument("--slp", action="store_true")
    ap.add_argument("--pre-fix", action="store_true")
    ap.add_argument("--window", type=int, default=4)
    ap.add_argument("--fail", action="store_true")
    ap.add_argument("--strict", action="store_true")
    ap.add_argument("--mfma-asm", action="store_true")
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
        mf_hits, n_mf = scan_mfma_asm(asm) if a.mfma_asm else ([], 0)
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
    if a.mfma_asm:
        mnames = subprocess.run(["c++filt"] + [h[0] for h in mf_hits], capture_output=True, text=True).stdout.split("\n")
        for h, n in zip(mf_hits, mnames):
            short = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
            print("  %-44s %2d states behind the MFMA: %s   [%s]" % (short[:44], h[1], h[3], h[4]))
        print("  hand-issued MFMAs (inside asm statements): %d; compiler instructions on a pending result: %d" % (n_mf, len(mf_hits)))
    if a.fail and (hits or mf_hits):
        sys.exit(1)


if __name__ == "__main__":
    main()
