#!/usr/bin/env python3
"""Time of the NN stage's fp16 filter alone (HIP events around the kernel, the handle's own bracketing):
N x N x 4096 unit rows with 20 % planted neighbours, the shape of BASELINE configs[1]'s NN stage.

usage: tools/nn_filter_time.py [N] [reps] [full]
  full = 1: SF_OPT_NN_FULL_FILTER (the prefix ladder forced to the whole descriptor)
Environment (timing experiments of k_nn.hip): SF_NN_K128_LDS_PANEL=1 (round-2 kernel); with a library built by
`make -C multi_robot_slam_separators_amd/csrc NN_ABLATION=1` (touch k_nn.hip first) also SF_NN_K128_ABL=1|3|5|7|8|24
(bit 0 no hit scan, bit 1 no LDS reads, bit 2 no next-tile DMA, bit 3 no tile loop, bit 4 no prologue DMA: timing
only, results are wrong)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_robot_slam_separators_amd import lib, _abi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
full = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dim = 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
a = torch.randn(n, dim, device=dev, generator=g); a /= a.norm(dim=1, keepdim=True)
b = torch.randn(n, dim, device=dev, generator=g); b /= b.norm(dim=1, keepdim=True)
k = n // 5
b[:k] = a[:k] + 0.03 * torch.randn(k, dim, device=dev, generator=g) / dim ** 0.5
b[:k] /= b[:k].norm(dim=1, keepdim=True)
p = _abi.default_params(); p.netvlad_distance = 0.13; p.netvlad_max_matches_nb = n; p.netvlad_dimensions = dim
p.nn_precision = 1
with lib.SeparatorFinder(p) as f:
    f.nn_append_local_device(a.data_ptr(), n, dim)
    f.nn_append_received_device(b.data_ptr(), n, dim)
    if full:
        f.set_option(_abi.SF_OPT_NN_FULL_FILTER, 1)
    for _ in range(5):
        m = f.nn_find_matches()
    f.prof_enable(True); f.prof_reset()
    for _ in range(reps):
        m = f.nn_find_matches()
    pr = f.prof_get()
    f.prof_enable(False)
    kd = f.nn_last_filter_dims()
    for name, (cnt, ms) in pr.items():
        if cnt:
            print("%-20s %4d launches  %.4f ms each" % (name, cnt, ms / cnt))
    nl = (n + 127) // 128 * 128
    fl = 2.0 * nl * nl * kd
    t = pr["k_nn_filter_f16"][1] / max(1, pr["k_nn_filter_f16"][0]) * 1e-3
    print("N = %d, filter level %d dims, %d matches; filter %.1f us = %.3f PFLOP/s (padded rows) = %.1f %% of 2.5 PF"
          % (n, kd, len(m), t * 1e6, fl / t / 1e15, 100 * fl / t / 2.5e15))
