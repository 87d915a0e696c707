"""Latency of sf_extract_keyframe_device for one stereo keyframe: synthetic 752 x 480 image,
n corners, 32-byte BRIEF; asynchronous calls timed with HIP events over `reps` keyframes.
usage: python tools/extract_latency.py [n_corners=600] [reps=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from multi_robot_slam_separators_amd import lib, synth  # noqa: E402
from tests import extract_cases as ec  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    image, kp, rx, st, cam = ec.make_case(3, n=n, pad=0)
    dev = torch.device("cuda:0")
    p = synth.camera_params()
    p.max_features = max(1024, n)
    p.store_capacity = reps + 16
    f = lib.SeparatorFinder(p, device=0)
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = torch.from_numpy(np.ascontiguousarray(image)).to(dev)
    d_kp = torch.from_numpy(kp.view(np.uint8)).to(dev)
    d_rx = torch.from_numpy(rx).to(dev)
    d_st = torch.from_numpy(st).to(dev)
    h, w = image.shape
    for _ in range(5):
        f.extract_keyframe_device(d_img.data_ptr(), w, h, w, d_kp.data_ptr(), d_rx.data_ptr(), d_st.data_ptr(), n, cam,
                                  want_rows=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f.extract_keyframe_device(d_img.data_ptr(), w, h, w, d_kp.data_ptr(), d_rx.data_ptr(), d_st.data_ptr(), n, cam,
                                  want_rows=False)
    e1.record()
    torch.cuda.synchronize()
    slot, rows = f.extract_keyframe_device(d_img.data_ptr(), w, h, w, d_kp.data_ptr(), d_rx.data_ptr(), d_st.data_ptr(), n,
                                           cam)
    print("%d x %d image, %d corners (%d kept): %.1f us per keyframe (asynchronous, %d back to back)" % (
        w, h, n, rows, e0.elapsed_time(e1) * 1e3 / reps, reps))
    # the detector in front of it (two synchronisations inside the call: candidate count, corner count)
    import time
    d_out = torch.zeros((4096, 28), dtype=torch.uint8, device=dev)
    for _ in range(3):
        nc = f.detect_corners_device(d_img.data_ptr(), w, h, w, 1000, 0.001, 3.0, d_out.data_ptr(), 4096)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        nc = f.detect_corners_device(d_img.data_ptr(), w, h, w, 1000, 0.001, 3.0, d_out.data_ptr(), 4096)
    torch.cuda.synchronize()
    print("sf_detect_corners_device (max 1000, quality 0.001, minDistance 3): %d corners, %.1f us per call (wall, "
          "synchronous)" % (nc, (time.perf_counter() - t0) * 1e6 / 50))
    # the stereo correspondence between the two (pyramids of both images + one tracker launch), on a real pair
    left, right, _ = ec.make_stereo_pair(5, pad=0)
    dl, dr = torch.from_numpy(np.ascontiguousarray(left)).to(dev), torch.from_numpy(np.ascontiguousarray(right)).to(dev)
    nc = f.detect_corners_device(dl.data_ptr(), w, h, w, 1000, 0.001, 3.0, d_out.data_ptr(), 4096)
    d_xy = torch.zeros((nc, 2), dtype=torch.float32, device=dev)
    d_s = torch.zeros(nc, dtype=torch.uint8, device=dev)
    d_x = torch.zeros(nc, dtype=torch.float32, device=dev)
    for _ in range(5):
        f.stereo_correspondences_device(dl.data_ptr(), dr.data_ptr(), w, h, w, d_out.data_ptr(), nc, d_xy.data_ptr(),
                                        d_s.data_ptr(), d_x.data_ptr())
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f.stereo_correspondences_device(dl.data_ptr(), dr.data_ptr(), w, h, w, d_out.data_ptr(), nc, d_xy.data_ptr(),
                                        d_s.data_ptr(), d_x.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    print("sf_stereo_correspondences_device (15 x 3 window, 6 levels, <= 30 steps): %d corners, %d tracked, %.1f us per "
          "call (asynchronous, %d back to back)" % (nc, int(d_s.sum()), e0.elapsed_time(e1) * 1e3 / reps, reps))
    f.close()


if __name__ == "__main__":
    main()
