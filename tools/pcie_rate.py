#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point sf_estimate_transform_batch (the literal
drop-in for one estimate_transformation service call) -- reported in DESIGN.md, never as `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_robot_slam_separators_amd import _abi, lib, synth
p = synth.camera_params(); p.iterations = 500
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
A, B, is_true, _ = synth.make_pairs(3, N, k=500, cols=32, true_frac=0.2)
mb = lambda n: n * 2 * 500 * (32 + 12 + 28) / 1e6
with lib.SeparatorFinder(p) as f:
    f.estimate_transform_batch(A[:8], B[:8])
    for n in (64, 512, N):
        f.estimate_transform_batch(A[:n], B[:n])                      # sizes the staging buffers
        t0 = time.perf_counter(); r = f.estimate_transform_batch(A[:n], B[:n]); t1 = time.perf_counter()
        fa, ta = _abi.features_array(A[:n]), _abi.features_array(B[:n])
        res = np.zeros(n, dtype=_abi.RESULT_DTYPE)
        ts = []
        for rep in range(5):
            t2 = time.perf_counter(); rc = f._L.sf_estimate_transform_batch(f._h, fa, ta, n, res.ctypes.data); t3 = time.perf_counter()
            assert rc == 0
            ts.append(t3 - t2)
        tc = float(np.median(ts))
        print("batch of %4d host-buffer pairs (%.1f MB over PCIe): C call %.2f ms -> %.0f pairs/s, %.1f GB/s; "
              "through the Python binding %.2f ms -> %.0f pairs/s" % (
                  n, mb(n), tc * 1e3, n / tc, mb(n) / 1e3 / tc, (t1 - t0) * 1e3, n / (t1 - t0)), flush=True)
        assert np.array_equal(r["success"].astype(bool), is_true[:n]) and res.tobytes() == r.tobytes()
    t0 = time.perf_counter()
    for i in range(64):
        f.estimate_transform(A[i], B[i])
    t1 = time.perf_counter()
    print("single service calls: %.3f ms per call" % ((t1 - t0) / 64 * 1e3))
