#!/usr/bin/env python3
"""Throughput of the float32-descriptor (SURF / SIFT type, sf_params.desc_type = 1) verification path: n candidate pairs of
K features with `dims`-dimensional float rows resident in the store, one verify call (stage kernels: exact squared-L2 kNN-2
on the VALU in both matching passes, then the same motion estimation as the binary path), pairs per second from HIP events;
the first `check` pairs are compared with the oracle byte for byte.  The binary twin of the same frames runs beside it.
usage: python tools/bench_l2.py [n_pairs=2048] [k=500] [dims=64] [check=24]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from multi_robot_slam_separators_amd import _abi, lib, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402


def run(n, k, dims, check, floats):
    p = synth.camera_params()
    p.iterations = 500
    p.max_features = k
    p.store_capacity = 2 * n
    if floats:
        p.desc_type = 1
        p.desc_bytes = 4 * dims
    A, B, is_true, _ = synth.make_pairs(4242, n, k=k, cols=32, true_frac=0.2)
    if floats:
        rng = np.random.default_rng(7)
        A = [synth.float_descriptors(a, dims, rng, 0.05) for a in A]
        B = [synth.float_descriptors(b, dims, rng, 0.05) for b in B]
    dev = torch.device("cuda:0")
    with lib.SeparatorFinder(p, device=0) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        sa = [f.store_add_keyframe(a) for a in A]
        sb = [f.store_add_keyframe(b) for b in B]
        d_from = torch.tensor(sa, dtype=torch.int32, device=dev)
        d_to = torch.tensor(sb, dtype=torch.int32, device=dev)
        d_out = torch.empty((n, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        for _ in range(2):
            f.verify_pairs_device(d_from.data_ptr(), d_to.data_ptr(), n, d_out.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            f.verify_pairs_device(d_from.data_ptr(), d_to.data_ptr(), n, d_out.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
        same = sum(int(res[i].tobytes() == pyoracle.estimate_transform(p, A[i], B[i]).tobytes()) for i in range(check))
    ok = int(((res["success"] != 0) == is_true).sum())
    print("%-28s %5d pairs x K = %d: %8.3f ms per call = %9.0f pairs/s; decisions = ground truth %d / %d; first %d results "
          "byte-identical to the oracle: %d" % (("float32 x %d (squared L2)" % dims) if floats else "binary 256 bit (Hamming)",
                                                n, k, ms, n / (ms * 1e-3), ok, n, check, same))


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    n, k, dims, check = (a + [2048, 500, 64, 24][len(a):])[:4]
    run(n, k, dims, check, True)
    run(n, k, 128, check, True) if dims != 128 else None
    run(n, k, dims, check, False)
