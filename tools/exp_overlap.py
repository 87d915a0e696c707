#!/usr/bin/env python3
"""Experiment: would pipelining verification chunks on two streams (RANSAC/guided of chunk i overlapping
the VALU-bound match kernel of chunk i+1) pay?  Crude version with two handles (own streams + stores)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_robot_slam_separators_amd import lib, synth, _abi
n, k, cols = 10000, 500, 32
d = synth.make_store_batch(7, n, k=k, cols=cols, true_frac=0.2)
dev = torch.device("cuda:0")
def up(x):
    x = np.ascontiguousarray(x)
    return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
T = {key: up(d[key]) for key in ("desc_a", "desc_b", "xyz_a", "xyz_b", "kp_a", "kp_b")}
hs = []
for i in range(2):
    p = synth.camera_params(); p.iterations = 500; p.max_features = k; p.store_capacity = 2 * n
    f = lib.SeparatorFinder(p)            # own non-blocking stream
    a = f.store_add_keyframes_device(n, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
    b = f.store_add_keyframes_device(n, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
    f.synchronize(); hs.append((f, a, b))
d_from = torch.arange(n, dtype=torch.int32, device=dev) + hs[0][1]
d_to = torch.arange(n, dtype=torch.int32, device=dev) + hs[0][2]
d_res = torch.empty((n, 368), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def run(nchunks, two, stagger=False):
    """nchunks equal chunks, alternating between the two handles/streams when `two`; with `stagger` the
    first chunk is half-size, so that the two streams run out of phase (one matching while the other is in
    its motion-estimation kernels) instead of in lock-step."""
    t0 = time.perf_counter()
    sz = (n + nchunks - 1) // nchunks
    bounds, o = [], 0
    if stagger:
        bounds.append((0, sz // 2)); o = sz // 2
    while o < n:
        bounds.append((o, min(sz, n - o))); o += sz
    for c, (o, m) in enumerate(bounds):
        f = hs[c % 2 if two else 0][0]
        f.verify_pairs_device(d_from.data_ptr() + 4 * o, d_to.data_ptr() + 4 * o, m, d_res.data_ptr() + 368 * o)
    for f, _, _ in hs: f.synchronize()
    return (time.perf_counter() - t0) * 1e3
print("SF_FUSED =", os.environ.get("SF_FUSED", "1"))
for cfg in [(1, False), (2, False), (2, True), (2, True, True), (4, True), (4, True, True), (8, True, True), (16, True, True)]:
    run(*cfg)
    ts = [run(*cfg) for _ in range(15)]
    print("chunks %2d  two-streams %-5s stagger %-5s : median %.3f ms" % (cfg[0], cfg[1], len(cfg) > 2, np.median(ts)))
