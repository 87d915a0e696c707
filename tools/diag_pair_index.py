"""Diagnostic: does a pair's result depend on its position in the batch?  Reproduces tests/test_gpu_config_scale.py::
test_configs1_full_step's inputs, verifies the candidates in match order, then every candidate again at batch positions
0..7 (padding in front), and holds the differing ones against the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from multi_robot_slam_separators_amd import _abi, lib, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402
import test_gpu_config_scale as cs  # noqa: E402

n, k, cols, dim = 10000, 500, 32, 4096
p = synth.camera_params()
p.iterations = 500
p.netvlad_dimensions = dim
p.netvlad_max_matches_nb = n
p.max_features = k
p.store_capacity = 2 * n
d = cs.gen_pairs(2101, n, k, cols)
loc, rec, partner = cs.gen_netvlad(2102, n, n, dim, 0.6, aligned=True)
RB = _abi.RESULT_DTYPE.itemsize
with lib.SeparatorFinder(p) as f:
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    sa, sb = cs.add_store(f, d, "a", k, cols), cs.add_store(f, d, "b", k, cols)
    f.nn_append_received_device(rec.data_ptr(), n, dim)
    f.nn_append_local_device(loc.data_ptr(), n, dim)
    f.nn_set_precision(0)
    m0 = f.nn_find_matches(cap=n)
    nm = len(m0)
    fr = torch.from_numpy((sa + m0["idx_other"]).astype(np.int32)).to(cs.DEV)
    to = torch.from_numpy((sb + m0["idx_local"]).astype(np.int32)).to(cs.DEV)
    outs = []
    lists = {}
    dbg = os.environ.get("SF_DEBUG_CORR") == "1"
    for shift in range(8):
        frs = torch.cat([fr[:shift], fr]).contiguous()
        tos = torch.cat([to[:shift], to]).contiguous()
        out = torch.zeros((nm + shift, RB), dtype=torch.uint8, device=cs.DEV)
        f.verify_pairs_device(frs.data_ptr(), tos.data_ptr(), nm + shift, out.data_ptr())
        torch.cuda.synchronize()
        outs.append(cs.results_of(out[shift:], nm).copy())
        if dbg and shift > 0:
            cur = outs[-1]
            for i in range(nm):
                if cur[i].tobytes() != outs[0][i].tobytes() and len(lists) < 12:
                    lists[(shift, i)] = [f.debug_correspondences(i + shift, ps) for ps in (1, 2)] + [f.debug_pass_state(i + shift, 1)]
                    if os.environ.get("SF_DIAG"):
                        lists[(shift, i)].append(f.debug_guided_points(i + shift))
    if dbg:
        # reference lists of the same candidates from a clean pass (shift 0 layout)
        out = torch.zeros((nm, RB), dtype=torch.uint8, device=cs.DEV)
        f.verify_pairs_device(fr.data_ptr(), to.data_ptr(), nm, out.data_ptr())
        torch.cuda.synchronize()
        chk = cs.results_of(out, nm).copy()
        for (shift, i), got in lists.items():
            ref = [f.debug_correspondences(i, ps) for ps in (1, 2)]
            t_ref = f.debug_pass_state(i, 1)
            if os.environ.get("SF_DIAG") and len(got) > 3:
                g0, g1 = got[3]
                r0, r1 = f.debug_guided_points(i)
                du = np.nonzero(g0[:k] != r0[:k])[0]
                do = np.nonzero(g1[:k] != r1[:k])[0]
                print("candidate %d shift %d: points whose projection differs: %s; whose (count, last, in-image, octave) differ: %s" % (
                    i, shift, du.tolist()[:24], do.tolist()[:24]))
                for q in do[:6]:
                    print("    point %d: bad  oi %d last %d inimg %d oct %d u,v %s | good oi %d last %d inimg %d oct %d u,v %s" % (
                        q, g1[q] >> 48, (g1[q] >> 32) & 0xFFFF, (g1[q] >> 31) & 1, g1[q] & 0xFFFF,
                        np.array([g0[q] >> 32, g0[q] & 0xFFFFFFFF], dtype=np.uint32).view(np.float32).tolist(),
                        r1[q] >> 48, (r1[q] >> 32) & 0xFFFF, (r1[q] >> 31) & 1, r1[q] & 0xFFFF,
                        np.array([r0[q] >> 32, r0[q] & 0xFFFFFFFF], dtype=np.uint32).view(np.float32).tolist()))
            print("candidate %d shift %d: pass-1 pose identical to the reference run: %s (inliers %d / %d)" % (
                i, shift, got[2][0].tobytes() == t_ref[0].tobytes(), got[2][2], t_ref[2]))
            if got[2][0].tobytes() != t_ref[0].tobytes():
                print("    bad  T1 = %s\n    good T1 = %s" % (got[2][0].ravel().tolist(), t_ref[0].ravel().tolist()))
            same_now = chk[i].tobytes() == outs[0][i].tobytes()
            for ps in (0, 1):
                a = set(zip(got[ps][0].tolist(), got[ps][1].tolist())); b = set(zip(ref[ps][0].tolist(), ref[ps][1].tolist()))
                print("candidate %d shift %d pass %d: bad run %d pairs, reference %d (reference pass clean: %s); missing %s extra %s" % (
                    i, shift, ps + 1, len(a), len(b), same_now, sorted(b - a)[:12], sorted(a - b)[:12]))
    base = outs[0]
    print("diagnostic counters:", f.debug_counters().tolist())
    bad = sorted(set(i for s in range(1, 8) for i in range(nm) if outs[s][i].tobytes() != base[i].tobytes()))
    print("%d candidates; %d differ between batch positions" % (nm, len(bad)))
    for i in bad[:6]:
        ia, ib = int(m0["idx_other"][i]), int(m0["idx_local"][i])
        o = cs.oracle_pair(pyoracle, p, d, ia, ib)
        print("candidate %d = (A[%d], B[%d])  oracle: inl %d m %d inl1 %d m1 %d pos %s" % (
            i, ia, ib, o["inliers"], o["matches"], o["inliers_pass1"], o["matches_pass1"], o["position"].tolist()))
        for s in range(8):
            r = outs[s][i]
            print("   shift %d (pair & 3 = %d): inl %d m %d inl1 %d m1 %d pos %s %s" % (
                s, (i + s) & 3, r["inliers"], r["matches"], r["inliers_pass1"], r["matches_pass1"], r["position"].tolist(),
                "== oracle" if r.tobytes() == o.tobytes() else ""))
