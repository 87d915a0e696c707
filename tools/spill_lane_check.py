#!/usr/bin/env python3
"""Checks the SGPR spill lanes of ONE kernel in two builds of the same source: `shared.s` (the default build: stack slot
colouring lets spill slots whose live ranges do not overlap share a VGPR lane) against `unshared.s` (-mllvm
-no-stack-slot-sharing: every spill slot its own lane).  The two instruction streams are the same apart from the lane
numbers, so the n-th spill event (v_writelane / v_readlane on a spill VGPR) of one is the n-th of the other and the
unshared build names the slot every event of the shared build belongs to.  A reaching-definitions pass over the shared
build's control-flow graph then reports every reload (v_readlane) that a write of a DIFFERENT slot to the same lane can
reach: a reload that returns another value's bits.
usage: tools/spill_lane_check.py shared_kernel.s unshared_kernel.s   (kernel bodies cut out of hipcc -S --cuda-device-only)"""
import re, sys, collections

def parse(path):
    ins = []
    for l in open(path):
        s = l.split(';')[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r'^(\.LBB\w+):', s)
        if m:
            ins.append(('L', m.group(1)))
        elif s.startswith('\t') and not s.strip().startswith('.'):
            ins.append(('I', s.strip()))
    return ins

def spill_vgprs(ins):
    c = collections.Counter()
    for k, s in ins:
        m = re.match(r'v_writelane_b32 (v\d+), s\d+, \d+', s) if k == 'I' else None
        if m:
            c[m.group(1)] += 1
    return {v for v, n in c.items() if n >= 4}

def events(ins, vg):
    out = []
    for i, (k, s) in enumerate(ins):
        if k != 'I':
            continue
        m = re.match(r'v_writelane_b32 (v\d+), (s\d+), (\d+)', s)
        if m and m.group(1) in vg:
            out.append(['W', m.group(2), (m.group(1), int(m.group(3))), i]); continue
        m = re.match(r'v_readlane_b32 (s\d+), (v\d+), (\d+)', s)
        if m and m.group(2) in vg:
            out.append(['R', m.group(1), (m.group(2), int(m.group(3))), i])
    return out

A, B = parse(sys.argv[1]), parse(sys.argv[2])
ea, eb = events(A, spill_vgprs(A)), events(B, spill_vgprs(B))
print("spill events: shared %d, unshared %d" % (len(ea), len(eb)))
# align: same (kind, sgpr) within a small window (the schedulers differ by a few swaps)
used, slot_of = set(), {}
for i, e in enumerate(ea):
    for j in range(max(0, i - 8), min(len(eb), i + 9)):
        if j not in used and eb[j][0] == e[0] and eb[j][1] == e[1]:
            used.add(j); slot_of[e[3]] = eb[j][2]; break
    else:
        print("unaligned event", e)
# a lane of the unshared build that several shared lanes map to, or vice versa, is expected; what matters is below
# CFG of the shared build
labels = {s: i for i, (k, s) in enumerate(A) if k == 'L'}
n = len(A)
succ = [[] for _ in range(n)]
for i, (k, s) in enumerate(A):
    if k == 'L':
        succ[i].append(i + 1); continue
    op = s.split()[0]
    if op == 's_endpgm':
        continue
    if op == 's_branch':
        succ[i].append(labels[s.split()[1]]); continue
    if op.startswith('s_cbranch'):
        succ[i].append(labels[s.split()[1]])
    if i + 1 < n:
        succ[i].append(i + 1)
ev_at = {e[3]: e for e in ea}
# reaching writes per lane: forward dataflow, state = {lane: frozenset(write instr idx)}
IN = [None] * n
IN[0] = {}
work = collections.deque([0])
def merge(a, b):
    ch = False
    for k, v in b.items():
        if k not in a: a[k] = set(v); ch = True
        elif not v <= a[k]: a[k] |= v; ch = True
    return ch
while work:
    i = work.popleft()
    st = {k: set(v) for k, v in IN[i].items()}
    e = ev_at.get(i)
    if e and e[0] == 'W':
        st[e[2]] = {i}
    for j in succ[i]:
        if IN[j] is None:
            IN[j] = {k: set(v) for k, v in st.items()}; work.append(j)
        elif merge(IN[j], st):
            work.append(j)
bad = 0
for e in ea:
    if e[0] != 'R' or IN[e[3]] is None:
        continue
    want = slot_of.get(e[3])
    reach = IN[e[3]].get(e[2], set())
    wrong = sorted(w for w in reach if slot_of.get(w) != want)
    if wrong or not reach:
        bad += 1
        print("reload at instr %d (%s <- %s lane %d, slot %s): reached by write(s) of other slots: %s" %
              (e[3], e[1], e[2][0], e[2][1], want, [(w, A[w][1], slot_of.get(w)) for w in wrong] or "NONE"))
print("reloads checked: %d, suspicious: %d" % (sum(1 for e in ea if e[0] == 'R'), bad))
