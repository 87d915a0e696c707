#!/usr/bin/env python3
"""Where does a bench step spend host time?  (diagnostic, not part of the bench contract)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multi_robot_slam_separators_amd import _abi, lib, synth

n_kf, k, cols, dim = (int(sys.argv[1]) if len(sys.argv) > 1 else 10000), 500, 32, 4096
p = synth.camera_params(); p.iterations = 500; p.netvlad_dimensions = dim; p.netvlad_max_matches_nb = n_kf
p.nn_precision = 1; p.max_features = k; p.store_capacity = 2 * n_kf
feats, nv_a, nv_b, _ = bench.generate_inputs(12345, n_kf, k, cols, dim, 0.2)
dev = torch.device("cuda:0")
f = lib.SeparatorFinder(p); f.set_stream(torch.cuda.current_stream().cuda_stream)
def up(x):
    x = np.ascontiguousarray(x)
    return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
slots = {}
for w in "ab":
    first = None
    for s in range(0, n_kf, 2048):
        e = min(n_kf, s + 2048)
        a, b, c = up(feats["desc_" + w][s:e]), up(feats["xyz_" + w][s:e]), up(feats["kp_" + w][s:e])
        fs = f.store_add_keyframes_device(e - s, k, cols, a.data_ptr(), b.data_ptr(), c.data_ptr()); torch.cuda.synchronize()
        first = fs if first is None else first
    slots[w] = first
ta, tb = up(nv_a), up(nv_b)
f.nn_append_received_device(ta.data_ptr(), n_kf, dim); f.nn_append_local_device(tb.data_ptr(), n_kf, dim); torch.cuda.synchronize()
d_from = torch.empty(n_kf, dtype=torch.int32, device=dev); d_to = torch.empty_like(d_from)
d_res = torch.empty((n_kf, 368), dtype=torch.uint8, device=dev)
h_res = torch.empty((n_kf, 368), dtype=torch.uint8).pin_memory()
h_flags = torch.empty(n_kf, dtype=torch.bool).pin_memory()
OFF = _abi.RESULT_DTYPE.fields["success"][1]
T = {"nn": [], "idx": [], "verify_launch": [], "verify_wait": [], "d2h_pageable": [], "d2h_pinned": [],
     "accepted_compaction+copies (bench step tail)": []}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    t0 = time.perf_counter(); m = f.nn_find_matches(cap=n_kf); t1 = time.perf_counter()
    n = len(m)
    hf = torch.from_numpy(m["idx_other"].astype(np.int32) + slots["a"]); ht = torch.from_numpy(m["idx_local"].astype(np.int32) + slots["b"])
    d_from[:n].copy_(hf); d_to[:n].copy_(ht); t2 = time.perf_counter()
    f.verify_pairs_device(d_from.data_ptr(), d_to.data_ptr(), n, d_res.data_ptr()); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    x = d_res[:n].cpu(); t5 = time.perf_counter()
    h_res[:n].copy_(d_res[:n], non_blocking=True); torch.cuda.synchronize(); t6 = time.perf_counter()
    res2d = d_res[:n]; succ = res2d[:, OFF] != 0; acc = res2d[succ]
    h_flags[:n].copy_(succ, non_blocking=True); h_res[: acc.shape[0]].copy_(acc, non_blocking=True); torch.cuda.synchronize(); t7 = time.perf_counter()
    for key, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)): T[key].append(v * 1e3)
for key, v in T.items(): print("%-46s median %.3f ms" % (key, np.median(v[2:])))
