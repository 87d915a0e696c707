#!/usr/bin/env python3
"""A/B the matching-kernel geometries in ONE process, interleaved rounds (guide rule 24).
Usage: python tools/ab_match.py [n_pairs] [k] [cols] variants..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_robot_slam_separators_amd import lib, synth, _abi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 500
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 32
variants = [int(v) for v in sys.argv[4:]] or [2256, 1256, 4128, 2128, 4256, 8064, 4064]
d = synth.make_store_batch(7, n, k=k, cols=cols, true_frac=0.2)
dev = torch.device("cuda:0")
def up(x):
    x = np.ascontiguousarray(x)
    return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
T = {key: up(d[key]) for key in ("desc_a", "desc_b", "xyz_a", "xyz_b", "kp_a", "kp_b")}
finders = {}
for v in variants:
    os.environ["SF_MATCH_VARIANT"] = str(v)
    p = synth.camera_params(); p.iterations = 500; p.max_features = k; p.store_capacity = 2 * n
    f = lib.SeparatorFinder(p)
    a = f.store_add_keyframes_device(n, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
    b = f.store_add_keyframes_device(n, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
    f.synchronize()
    finders[v] = (f, np.arange(n, dtype=np.int32) + a, np.arange(n, dtype=np.int32) + b)
ref = None
times = {v: [] for v in variants}
for rnd in range(7):
    for v in variants:
        f, fs, ts = finders[v]
        f.prof_reset(); f.prof_enable(True)
        res = f.verify_pairs(fs, ts)
        pr = f.prof_get(); f.prof_enable(False)
        times[v].append(pr["k_match_global"][1])
        if ref is None:
            ref = res.tobytes()
        assert res.tobytes() == ref, "variant %d changes results" % v
print("n=%d k=%d cols=%d accepted=%d" % (n, k, cols, int(np.frombuffer(ref, dtype=_abi.RESULT_DTYPE)["success"].sum())))
for v in variants:
    t = np.array(times[v][1:])
    print("variant NQ=%d NT=%3d : median %.3f ms  min %.3f ms  -> %.2f M pairs/s" % (v // 1000, v % 1000, np.median(t), t.min(), n / np.median(t) / 1e3))
