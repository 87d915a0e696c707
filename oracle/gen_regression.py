#!/usr/bin/env python3
"""Regression vectors of the verification stage (tests/golden/verify_regression.npz).

NOT reference outputs: the matching / RANSAC / PnP arithmetic of the reference lives in un-vendored
rtabmap / PCL / OpenCV (PARITY UNPINNED, see sf_oracle.h).  These vectors freeze the canonical
arithmetic of DESIGN.md section 4 as this oracle states it today, so that a compiler upgrade or an
edit that silently changes a rounding shows up as a byte difference -- in the oracle (CPU test) and in
the HIP kernels (GPU test) alike.

Run from the repo root:  python oracle/gen_regression.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multi_robot_slam_separators_amd import _abi, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "verify_regression.npz")


def params(estimation_type):
    p = synth.camera_params()
    p.iterations = 200
    p.estimation_type = estimation_type
    return p


def main():
    A, B, is_true, _ = synth.make_pairs(20260101, 6, k=120, cols=32, true_frac=1.0, overlap=0.5)
    A2, B2, f2, _ = synth.make_pairs(20260102, 2, k=120, cols=32, true_frac=0.0)
    A, B, is_true = A + A2, B + B2, np.concatenate([is_true, f2])
    B[1].xyz[::3] = np.nan                   # non-finite points on the "to" side
    A[2].xyz[5] = 0.0                        # a zero point: dropped by 3D-3D, kept by PnP
    A[3].kpts["octave"][::2] = 1             # octave filter of the guided pass
    B[3].kpts["octave"][::2] = 1
    data = {"n": np.int32(len(A)), "is_true": is_true}
    for i, (a, b) in enumerate(zip(A, B)):
        for w, f in (("a", a), ("b", b)):
            data["desc_%s%d" % (w, i)] = f.desc
            data["xyz_%s%d" % (w, i)] = f.xyz
            data["kp_%s%d" % (w, i)] = f.kpts
    for est in (0, 1):
        p = params(est)
        res = np.stack([pyoracle.estimate_transform(p, a, b) for a, b in zip(A, B)])
        data["result_est%d" % est] = res.view(np.uint8).reshape(len(A), -1)
        print("estimation_type %d: success %s" % (est, res["success"].tolist()))
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
