"""Phase timestamps of k_verify_fused's per-pair pipeline (diagnostic build libsepfinder_trace.so, csrc `make trace`).

Loads the trace build through SEPFINDER_LIB, verifies n pairs of the bench shape (20 % true revisits) in one
launch, reads the [pair][48] wall-clock stamps thread 0 of every workgroup left, and prints, for the pairs that
ran the whole chain, the mean / median / p90 duration of every phase -- under load (the 10 000-pair launch) and
uncontended (a launch with fewer surviving pairs than CUs).  Also the launch's wall time from HIP events.

usage: python tools/chain_trace.py [n_pairs=10000] [k=500] [cols=32] [iterations=500] [pnp]
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
os.environ["SEPFINDER_LIB"] = os.path.join(ROOT, "multi_robot_slam_separators_amd", "libsepfinder_trace.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from multi_robot_slam_separators_amd import _abi, lib, synth  # noqa: E402

PHASES = [
    ("match", 0, 1),
    ("match stage rows", 0, 32), ("match scan (wave 0)", 32, 33), ("match scan (wait others)", 33, 34),
    ("match compact", 34, 35), ("match header", 35, 1), ("split: match kernel, whole", 0, 36),
    ("r1 gather", 1, 2), ("r1 pca", 2, 3), ("r1 hypotheses", 3, 4), ("r1 select", 4, 5), ("r1 refine", 5, 6),
    ("r1 variance+out", 6, 7),
    ("guided bucket", 7, 8), ("guided search", 8, 9), ("guided compact", 9, 10),
    ("r2 gather", 10, 11), ("r2 pca", 11, 12), ("r2 hypotheses", 12, 13), ("r2 select", 13, 14),
    ("r2 refine", 14, 15), ("r2 variance+out", 15, 16), ("finalize", 16, 17),
    ("r1 round0 fit", 3, 18), ("r1 round0 count", 18, 19), ("r1 round0 replay", 19, 20),
    ("r2 round0 fit", 12, 21), ("r2 round0 count", 21, 22), ("r2 round0 replay", 22, 23),
    ("guided A0 project", 8, 28), ("guided A0 scan", 28, 29), ("guided A0 handover", 29, 30), ("guided A1 (round 2)", 30, 24),
    ("guided A record", 8, 24), ("guided B hamming", 24, 25), ("guided B second", 25, 26), ("guided C decide", 26, 9),
    ("CHAIN (1 -> 17)", 1, 17), ("WHOLE (0 -> 17)", 0, 17),
]


PHASES_PNP = [
    ("match kernel (whole)", 0, 36),
    ("p1 gather + bearings", 1, 37), ("p1 RANSAC", 37, 38), ("p1 mask + first solve", 38, 39),
    ("p1 refinement rounds", 39, 40), ("p1 pose + covariance", 40, 41),
    ("guided matching", 41, 42),
    ("p2 gather + bearings", 42, 43), ("p2 RANSAC", 43, 44), ("p2 mask + first solve", 44, 45),
    ("p2 refinement rounds", 45, 46), ("p2 pose + covariance", 46, 47),
    ("CHAIN (1 -> 17)", 1, 17),
]


def run(n, k, cols, iters, true_frac, label, est=0):
    p = synth.camera_params()
    p.estimation_type = est
    p.iterations = iters
    p.max_features = k
    p.desc_bytes = cols
    p.store_capacity = 2 * n
    feats = synth.make_store_batch(777, n, k=k, cols=cols, true_frac=true_frac)
    dev = torch.device("cuda:0")
    f = lib.SeparatorFinder(p, device=0)
    f.set_stream(torch.cuda.current_stream().cuda_stream)

    def up(x):
        x = np.ascontiguousarray(x)
        if x.dtype.fields:
            x = x.view(np.uint8)
        return torch.from_numpy(x).to(dev)
    first = {}
    for which in ("a", "b"):
        fs0 = None
        for s in range(0, n, 2048):
            e = min(n, s + 2048)
            td, tx, tk = up(feats["desc_" + which][s:e]), up(feats["xyz_" + which][s:e]), up(feats["kp_" + which][s:e])
            fs = f.store_add_keyframes_device(e - s, k, cols, td.data_ptr(), tx.data_ptr(), tk.data_ptr())
            torch.cuda.synchronize()
            fs0 = fs if fs0 is None else fs0
        first[which] = fs0
    d_from = torch.arange(first["a"], first["a"] + n, dtype=torch.int32, device=dev)
    d_to = torch.arange(first["b"], first["b"] + n, dtype=torch.int32, device=dev)
    d_out = torch.empty((n, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    for _ in range(3):
        f.verify_pairs_device(d_from.data_ptr(), d_to.data_ptr(), n, d_out.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f.verify_pairs_device(d_from.data_ptr(), d_to.data_ptr(), n, d_out.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    L = lib.load()
    L.sf_debug_chain_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    L.sf_debug_chain_trace.restype = ctypes.c_int
    tr = np.zeros((n, 48), np.uint64)
    rc = L.sf_debug_chain_trace(f._h, tr.ctypes.data, n)
    assert rc == 0, rc
    res = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
    if est == 1:
        us = tr.astype(np.float64) / 100.0
        full = (tr[:, 47] != 0) & (tr[:, 41] != 0)     # pairs whose chain ran both PnP passes to the end
        print("== %s (PnP chain): %d pairs, K=%d, %d iterations: launch %.3f ms, %d accepted, %d full chains" % (
            label, n, k, iters, e0.elapsed_time(e1), int(res["success"].sum()), int(full.sum())))
        for name, a, b in PHASES_PNP:
            d = (us[:, b] - us[:, a])[full]
            print("   %-24s mean %7.2f  median %7.2f  p90 %7.2f us" % (name, d.mean(), np.median(d), np.percentile(d, 90)))
        f.close()
        return
    full = (tr[:, 16] != 0) & (tr[:, 7] != 0)          # pairs that ran both RANSAC passes to the end
    us = tr.astype(np.float64) / 100.0                 # 100 MHz wall clock -> microseconds
    print("== %s: %d pairs, K=%d, %d B, %d iterations: launch %.3f ms, %d accepted, %d full chains" % (
        label, n, k, cols, iters, e0.elapsed_time(e1), int(res["success"].sum()), int(full.sum())))
    t0 = us[:, 0].min()
    print("   kernel span from stamps: %.1f us; match-only pairs: mean %.1f us" % (
        us[:, 17].max() - t0, float(np.mean((us[:, 1] - us[:, 0])[~full]))))
    ncand = tr[full, 27].astype(np.int64)
    print("   guided candidates per pair: mean %.0f  p90 %.0f  max %d" % (ncand.mean(), np.percentile(ncand, 90), ncand.max()))
    dbg = tr[full, 31]
    print("   guided scan, wavefront 0 / round 0: busiest lane's trips mean %.1f max %d; entries read by its 64 lanes mean %.0f" % (
        (dbg >> 32).mean(), int((dbg >> 32).max()), (dbg & 0xFFFFFFFF).mean()))
    for name, a, b in PHASES:
        d = (us[:, b] - us[:, a])[full]
        print("   %-18s mean %7.2f  median %7.2f  p90 %7.2f us" % (name, d.mean(), np.median(d), np.percentile(d, 90)))
    # when did chains run relative to the launch: start / end percentiles
    st, en = us[full, 1] - t0, us[full, 17] - t0
    print("   chain start p10/p50/p90: %.0f / %.0f / %.0f us;  chain end p50/p90/max: %.0f / %.0f / %.0f us" % (
        np.percentile(st, 10), np.percentile(st, 50), np.percentile(st, 90),
        np.percentile(en, 50), np.percentile(en, 90), en.max()))
    f.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    cols = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    est = 1 if (len(sys.argv) > 5 and sys.argv[5] == "pnp") else 0
    run(n, k, cols, iters, 0.2, "loaded", est)
    run(600, k, cols, iters, 0.2, "uncontended (~120 chains on 256 CUs)", est)


if __name__ == "__main__":
    main()
