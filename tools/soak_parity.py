#!/usr/bin/env python3
"""Parity soak: many random candidate pairs through the C-ABI vs the oracle; reports how many results
are bit-identical and fails on any integer-output difference.
Usage: soak_parity.py [rounds] [pairs] [estimator: 3d3d | pnp | mixed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from multi_robot_slam_separators_amd import lib, synth, _abi
from oracle import pyoracle
from test_gpu_fuzz import corrupt, random_frame

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 60
estimator = sys.argv[3] if len(sys.argv) > 3 else "3d3d"
tot = exact = succ = 0
t0 = time.time()
for rd in range(rounds):
    rng = np.random.default_rng(50000 + rd)
    cols = int(rng.choice([8, 16, 32, 64]))
    p = synth.camera_params()
    p.iterations = int(rng.choice([30, 100, 300, 500, 1000]))
    p.min_inliers = int(rng.choice([3, 5, 8, 20]))
    p.nndr = float(rng.choice([0.5, 0.6, 0.8, 0.95]))
    p.guess_win_size = int(rng.choice([1, 5, 20, 40, 100]))
    p.refine_iterations = int(rng.choice([0, 1, 5, 8]))
    p.ransac_adaptive_stop = int(rng.integers(0, 2))
    p.inlier_distance = float(rng.choice([0.02, 0.1, 0.5]))
    p.refine_sigma = float(rng.choice([1.5, 3.0]))
    p.seed = int(rng.integers(0, 2**40))
    p.max_features = 64
    if estimator == "pnp" or (estimator == "mixed" and rd % 2 == 1):
        p.estimation_type = 1
        p.pnp_reproj_error = float(rng.choice([0.5, 2.0, 4.0, 10.0]))
        p.pnp_refine_iterations = int(rng.choice([0, 0, 1, 3, 5]))
        if rng.random() < 0.1:
            p.image_width = 0        # uncalibrated: the estimation never runs
    # the adjacent branches (Reg/Force3DoF, Vis/ForwardEstOnly = false) in a third of the rounds each
    if rng.random() < 0.33:
        p.force_3dof = 1
    if rng.random() < 0.33:
        p.forward_est_only = 0       # (both estimators; with the adjustment below in a quarter of those rounds)
    if p.image_width > 0 and rng.random() < 0.25:
        p.bundle_adjustment = 1
        p.stereo_baseline = float(rng.choice([0.0, 0.12]))
        p.ba_iterations = int(rng.choice([0, 1, 5, 20]))
    A, B = [], []
    for i in range(npairs):
        k = int(rng.choice([0, 1, 3, 9, 64, 100, 255, 256, 257, 500, 777, 1024]))
        a = random_frame(rng, k, cols)
        if k >= 8 and rng.random() < 0.65:
            b, _ = synth.make_true_partner(rng, a, synth.random_transform(rng, 35, 2.0), overlap=float(rng.uniform(0.05, 1.0)),
                                           noise=float(rng.uniform(0, 0.08)), flip=float(rng.uniform(0, 0.15)))
        else:
            b = random_frame(rng, int(rng.choice([0, 2, 64, 300, 500])), cols)
        A.append(corrupt(rng, a)); B.append(corrupt(rng, b))
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    ref = pyoracle.estimate_transform_batch(p, A, B, pyoracle.num_threads())
    for i in range(npairs):
        for key in ("success", "pass1_success", "pass2_guided", "inliers", "matches", "inliers_pass1", "matches_pass1"):
            assert got[i][key] == ref[i][key], (rd, i, key, got[i][key], ref[i][key])
        same = got[i].tobytes() == ref[i].tobytes()
        if not same:
            dp = np.abs(got[i]["position"] - ref[i]["position"]).max()
            dq = np.abs(got[i]["orientation"] - ref[i]["orientation"]).max()
            dc = np.abs(got[i]["covariance"] - ref[i]["covariance"]).max()
            print("round %d pair %d not bit-identical: dpos %.3e dquat %.3e dcov %.3e" % (rd, i, dp, dq, dc), flush=True)
            assert dp <= 1e-4 and dq <= 1e-3
        tot += 1; exact += int(same); succ += int(ref[i]["success"])
    if rd % 5 == 4:
        print("round %d: %d pairs, %d bit-identical, %d separators, %.0f s" % (rd + 1, tot, exact, succ, time.time() - t0), flush=True)
print("SOAK DONE: %d pairs, %d bit-identical (%.4f %%), %d accepted" % (tot, exact, 100.0 * exact / tot, succ))
