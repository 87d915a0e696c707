"""Parity soak of the feature path: random image sizes / textures / parameters through
sf_detect_corners_device, sf_stereo_correspondences_device (random stereo pairs, windows, level counts, iteration
limits) and sf_extract_keyframe_device against the oracle, byte for byte.
usage: python tools/soak_features.py [rounds=40]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from multi_robot_slam_separators_amd import _abi, lib, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tests import extract_cases as ec  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    p = synth.camera_params()
    p.max_features = 4096
    p.store_capacity = rounds + 8
    f = lib.SeparatorFinder(p, device=0)
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    bad = 0
    for rd in range(rounds):
        rng = np.random.default_rng(900 + rd)
        w, h = int(rng.integers(80, 900)), int(rng.integers(70, 600))
        image = ec.make_case(rd, n=1, width=w, height=h, pad=int(rng.integers(0, 9)))[0]
        if rd % 5 == 0:                                   # quantised texture: many exactly equal responses
            image = (image // 32 * 32).astype(np.uint8)
        maxc = int(rng.choice([0, 50, 500, 1000, 3000]))
        q = float(rng.choice([0.0005, 0.001, 0.01, 0.1]))
        md = float(rng.choice([0.0, 1.0, 2.4, 3.0, 7.0, 12.5]))
        pitch = image.strides[0]
        base = np.lib.stride_tricks.as_strided(image, shape=(h, pitch), strides=(pitch, 1)) if pitch != w else image
        d_img = torch.from_numpy(np.ascontiguousarray(base)).to(dev)
        cap = w * h
        d_kp = torch.zeros((cap, 28), dtype=torch.uint8, device=dev)
        n = f.detect_corners_device(d_img.data_ptr(), w, h, pitch, maxc, q, md, d_kp.data_ptr(), cap)
        torch.cuda.synchronize()
        kp = np.frombuffer(d_kp.cpu().numpy().tobytes(), dtype=_abi.KEYPOINT_DTYPE)[:n]
        ref = pyoracle.detect_corners(image, maxc, q, md)
        ok = n == len(ref) and kp.tobytes() == ref.tobytes()
        # extraction of those corners with planted disparities
        if n > 0 and w >= 120 and h >= 120:
            n = min(n, 2000)                              # (a keyframe holds at most 4096 features)
            ref = ref[:n]
            rx = (ref["x"] - rng.uniform(-5.0, 60.0, n)).astype(np.float32)
            st = (rng.random(n) > 0.1).astype(np.uint8)
            cam = _abi.stereo_camera(400.0, 410.0, w / 2.0, h / 2.0, 0.1, cx_right=float(rng.choice([0.0, w / 2.0 + 3])),
                                     local_transform=None if rd % 2 else [[0, 0, 1, 0.1], [-1, 0, 0, 0], [0, -1, 0, 0.2]],
                                     min_depth=float(rng.choice([0.0, 0.5])), max_depth=float(rng.choice([0.0, 8.0])))
            nb = int(rng.choice([16, 32, 64]))
            tests = ec.brief_tests(rd, nb)
            f.brief_set_pattern(tests)
            d_rx, d_st = torch.from_numpy(rx).to(dev), torch.from_numpy(st).to(dev)
            d_desc = torch.zeros((n, nb), dtype=torch.uint8, device=dev)
            d_xyz = torch.zeros((n, 3), dtype=torch.float32, device=dev)
            f.store_clear()
            slot, rows = f.extract_keyframe_device(d_img.data_ptr(), w, h, pitch, d_kp.data_ptr(), d_rx.data_ptr(),
                                                   d_st.data_ptr(), n, cam, d_desc.data_ptr(), d_xyz.data_ptr(), None)
            torch.cuda.synchronize()
            d, pz, k = pyoracle.extract_keyframe(image, ref, rx, st, cam, tests)
            got = d_xyz.cpu().numpy()[:rows]
            ok = ok and rows == len(d) and d_desc.cpu().numpy()[:rows].tobytes() == d.tobytes() and \
                np.array_equal(np.isnan(got), np.isnan(pz)) and got[~np.isnan(got)].tobytes() == pz[~np.isnan(pz)].tobytes()
        # stereo correspondence of detector corners on a random rectified pair
        left, right, _ = ec.make_stereo_pair(rd, width=w, height=h, pad=int(rng.integers(0, 9)),
                                             max_disp=float(rng.uniform(3.0, max(4.0, min(60.0, w / 5)))))
        kl = pyoracle.detect_corners(left, int(rng.choice([100, 1000, 2500])), 0.001, 3.0)
        kl["x"] += rng.uniform(-0.5, 0.5, len(kl)).astype(np.float32) * float(rng.choice([0.0, 1.0]))
        kl["x"] = np.clip(kl["x"], 0, w - 1)
        win = [(15, 3), (15, 3), (21, 21), (5, 5), (9, 3), (31, 31), (3, 7)][int(rng.integers(0, 7))]
        prm = _abi.stereo_flow_params(win_width=win[0], win_height=win[1], max_level=int(rng.integers(0, 7)),
                                      iterations=int(rng.choice([1, 10, 30, 100])), epsilon=float(rng.choice([0.0, 0.01, 0.03])),
                                      min_disparity=float(rng.choice([0.0, 0.5])), max_disparity=float(rng.choice([16.0, 128.0])))
        lp = left.strides[0]

        def up(img):
            b = np.lib.stride_tricks.as_strided(img, shape=(h, lp), strides=(lp, 1)) if lp != w else img
            return torch.from_numpy(np.ascontiguousarray(b)).to(dev)
        dl, dr = up(left), up(right)
        nk = len(kl)
        d_k = torch.from_numpy(np.frombuffer(kl.tobytes() + b"\0" * 28, np.uint8).copy()).to(dev)
        d_xy = torch.zeros((nk + 1, 2), dtype=torch.float32, device=dev)
        d_s = torch.zeros(nk + 1, dtype=torch.uint8, device=dev)
        d_e = torch.zeros(nk + 1, dtype=torch.float32, device=dev)
        f.stereo_correspondences_device(dl.data_ptr(), dr.data_ptr(), w, h, lp, d_k.data_ptr(), nk, d_xy.data_ptr(),
                                        d_s.data_ptr(), None, d_e.data_ptr(), params=prm)
        torch.cuda.synchronize()
        xy0, st0, er0 = pyoracle.stereo_correspondences(left, right, kl, prm)
        ok_lk = d_xy.cpu().numpy()[:nk].tobytes() == xy0.tobytes() and np.array_equal(d_s.cpu().numpy()[:nk], st0) and \
            d_e.cpu().numpy()[:nk].tobytes() == er0.tobytes()
        if not ok_lk:
            print("round %d: stereo correspondence MISMATCH (%d x %d, window %s, levels %d, %d corners)" % (
                rd, w, h, win, prm.max_level, nk), flush=True)
        ok = ok and ok_lk
        bad += 0 if ok else 1
        if not ok or rd % 10 == 9:
            print("round %d: %d x %d, maxc %d q %g md %g: %d corners %s" % (rd, w, h, maxc, q, md, n, "OK" if ok else "MISMATCH"),
                  flush=True)
    print("FEATURE SOAK DONE: %d rounds, %d mismatching" % (rounds, bad))
    f.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
