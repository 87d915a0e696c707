#!/usr/bin/env python3
"""Uncontended per-workgroup latency of the motion-estimation chain: the stage kernels (SF_FUSED=0) on 256
TRUE pairs = one workgroup per CU, so each kernel's duration is the latency of one pair's stage.  (The tail
of k_verify_fused is this chain for the last surviving pairs.)  Diagnostic only."""
import os, sys
import numpy as np
os.environ["SF_FUSED"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_robot_slam_separators_amd import lib, synth
n, k, cols = 256, 500, 32
d = synth.make_store_batch(11, n, k=k, cols=cols, true_frac=1.0)
dev = torch.device("cuda:0")
def up(x):
    x = np.ascontiguousarray(x)
    return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
T = {key: up(d[key]) for key in ("desc_a", "desc_b", "xyz_a", "xyz_b", "kp_a", "kp_b")}
stops = [int(x) for x in sys.argv[1:]] or [0]        # optional: SF_RANSAC_STOP phases to truncate after (0 = full)
for est, stop in [(e, st) for e in (0, 1) for st in stops]:
    os.environ["SF_RANSAC_STOP"] = str(stop)
    p = synth.camera_params(); p.iterations = 500; p.max_features = k; p.store_capacity = 2 * n; p.estimation_type = est
    with lib.SeparatorFinder(p) as f:
        a = f.store_add_keyframes_device(n, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
        b = f.store_add_keyframes_device(n, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
        fs, ts = np.arange(n, dtype=np.int32) + a, np.arange(n, dtype=np.int32) + b
        acc = {}
        for r in range(6):
            f.prof_reset(); f.prof_enable(True); res = f.verify_pairs(fs, ts); pr = f.prof_get(); f.prof_enable(False)
            if r >= 1:
                for kname, (cnt, ms) in pr.items():
                    if cnt: acc.setdefault(kname, []).append(ms * 1e3)
        print("estimator %s, stop %d, %d true pairs (%d accepted): per-stage latency in us:" % ("3D-3D" if est == 0 else "PnP", stop, n, int(res["success"].sum())),
              {kn: round(float(np.median(v)), 1) for kn, v in acc.items()})
