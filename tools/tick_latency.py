#!/usr/bin/env python3
"""Latency of ONE tick of the reference's loop at its own scale (BASELINE configs[0] shape: 2 robots x
200 keyframes, 128-D NetVLAD, 500 features, <= 20 candidates per tick), GPU library vs CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_robot_slam_separators_amd import lib, synth, _abi
from oracle import pyoracle
n_kf, k, dim = 200, 500, 128
p = synth.camera_params(); p.iterations = 300; p.netvlad_dimensions = dim; p.max_features = k
d = synth.make_store_batch(3, n_kf, k=k, cols=32, true_frac=0.3)
rng = np.random.default_rng(4)
a = rng.standard_normal((n_kf, dim)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
b = a + rng.standard_normal((n_kf, dim)).astype(np.float32) * np.float32(0.05 / np.sqrt(dim)); b /= np.linalg.norm(b, axis=1, keepdims=True)
A = [_abi.FeatureArrays(d["desc_a"][i], d["xyz_a"][i], d["kp_a"][i]) for i in range(n_kf)]
B = [_abi.FeatureArrays(d["desc_b"][i], d["xyz_b"][i], d["kp_b"][i]) for i in range(n_kf)]
with lib.SeparatorFinder(p) as f:
    sa = [f.store_add_keyframe(x) for x in A]; sb = [f.store_add_keyframe(x) for x in B]
    f.nn_append_local(b); f.nn_append_received(a)
    def tick():
        m = f.nn_find_matches()                      # <= netvlad_max_matches_nb = 20 candidates
        r = f.verify_pairs([sa[j] for j in m["idx_other"]], [sb[i] for i in m["idx_local"]])
        return m, r
    tick()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); m, r = tick(); ts.append(time.perf_counter() - t0)
    print("GPU  : one tick (NN %dx%dx%d + %d verifications): median %.3f ms" % (n_kf, n_kf, dim, len(m), np.median(ts) * 1e3))
t0 = time.perf_counter()
mo, _, _ = pyoracle.find_matches(b.astype(np.float64), a.astype(np.float64), netvlad_distance=p.netvlad_distance, max_matches_nb=20)
ro = [pyoracle.estimate_transform(p, A[int(x["idx_other"])], B[int(x["idx_local"])]) for x in mo]
t1 = time.perf_counter()
print("CPU  : same tick on the oracle, 1 thread: %.1f ms" % ((t1 - t0) * 1e3))
assert np.array_equal(m["idx_local"], mo["idx_local"]) and all(int(x["success"]) == int(y["success"]) for x, y in zip(r, ro))
print("reference: the loop runs at 0.3 Hz (find_separators.py:17), i.e. one such tick per 3333 ms")
