#!/usr/bin/env python3
"""Cumulative phase cost of k_pnp by truncation (diagnostic; results of truncated runs are invalid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_robot_slam_separators_amd import lib, synth
n, k, cols = 8192, 500, 32
d = synth.make_store_batch(7, n, k=k, cols=cols, true_frac=0.2)
dev = torch.device("cuda:0")
def up(x):
    x = np.ascontiguousarray(x)
    return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
T = {key: up(d[key]) for key in ("desc_a", "desc_b", "xyz_a", "xyz_b", "kp_a", "kp_b")}
names = {1: "gather", 2: "+hypotheses+replay", 3: "+best model+mask", 4: "+LM", 0: "full (+covariance)"}
for stop in (1, 2, 3, 4, 0):
    os.environ["SF_RANSAC_STOP"] = str(stop)
    p = synth.camera_params(); p.iterations = 500; p.estimation_type = 1; p.max_features = k; p.store_capacity = 2 * n
    f = lib.SeparatorFinder(p)
    a = f.store_add_keyframes_device(n, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
    b = f.store_add_keyframes_device(n, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
    fs, ts = np.arange(n, dtype=np.int32) + a, np.arange(n, dtype=np.int32) + b
    ts_ = []
    for r in range(5):
        f.prof_reset(); f.prof_enable(True); f.verify_pairs(fs, ts); pr = f.prof_get(); f.prof_enable(False)
        ts_.append(pr["k_ransac(pass1)"][1])
    print("stop %d %-22s k_pnp(pass1) %.3f ms" % (stop, names[stop], np.median(ts_[1:])))
    f.close()
