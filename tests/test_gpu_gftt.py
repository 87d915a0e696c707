"""GPU: sf_detect_corners_device (csrc/k_gftt.hip, SURVEY section 8 row f3 -- the detector) against the CPU oracle:
the same corners in the same order, byte for byte; and pixels -> corners -> store slot without leaving the device."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, lib, synth
from oracle import pyoracle
from tests import extract_cases as ec

pytestmark = pytest.mark.gpu


@pytest.fixture()
def finder():
    import torch
    p = synth.camera_params()
    p.max_features = 2048
    f = lib.SeparatorFinder(p, device=0)
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    yield f
    f.close()


def detect(f, torch, image, maxc, q, md, cap=None):
    dev = torch.device("cuda:0")
    h, w = image.shape
    pitch = image.strides[0]
    base = np.lib.stride_tricks.as_strided(image, shape=(h, pitch), strides=(pitch, 1)) if pitch != w else image
    d_img = torch.from_numpy(np.ascontiguousarray(base)).to(dev)
    cap = w * h if cap is None else cap
    d_kp = torch.zeros((max(cap, 1), 28), dtype=torch.uint8, device=dev)
    n = f.detect_corners_device(d_img.data_ptr(), w, h, pitch, maxc, q, md, d_kp.data_ptr(), cap)
    torch.cuda.synchronize()
    kp = np.frombuffer(d_kp.cpu().numpy().tobytes(), dtype=_abi.KEYPOINT_DTYPE)[:min(n, cap)]
    return n, kp, d_img, d_kp


@pytest.mark.parametrize("seed,shape,maxc,q,md", [
    (1, (120, 160), 1000, 0.001, 3.0), (2, (97, 131), 200, 0.01, 7.0), (3, (64, 64), 0, 0.001, 1.0),
    (4, (480, 752), 1000, 0.001, 3.0), (5, (50, 70), 50, 0.05, 0.0), (6, (33, 35), 1000, 0.001, 2.4),
    (7, (720, 1280), 2000, 0.001, 3.0), (8, (480, 752), 0, 0.0005, 2.0),
    (9, (1080, 1920), 1500, 0.001, 4.5),          # the taken-corner bitmap does not fit LDS: per-cell lists in HBM
    (10, (300, 400), 400, 0.001, 30.0),           # a large minDistance: 59 rows of the bitmap per test
])
def test_corners_equal_oracle(finder, seed, shape, maxc, q, md):
    import torch
    image = ec.make_case(seed, n=1, width=shape[1], height=shape[0])[0]
    n, kp, _, _ = detect(finder, torch, image, maxc, q, md)
    ref = pyoracle.detect_corners(image, maxc, q, md)
    assert n == len(ref) and n > 0
    assert kp.tobytes() == ref.tobytes()


def test_flat_image_ties_and_small_capacity(finder):
    import torch
    n, kp, _, _ = detect(finder, torch, np.full((40, 50), 9, np.uint8), 100, 0.01, 3.0)
    assert n == 0
    img = np.zeros((60, 80), np.uint8)
    img[10:14, 10:14] = 255
    img[40:44, 50:54] = 255                      # two identical blobs: equal responses, address order decides
    n, kp, _, _ = detect(finder, torch, img, 0, 0.5, 3.0)
    ref = pyoracle.detect_corners(img, 0, 0.5, 3.0)
    assert n == len(ref) and kp.tobytes() == ref.tobytes()
    n2, kp2, _, _ = detect(finder, torch, img, 0, 0.5, 3.0, cap=3)          # more corners than the caller has room for
    assert n2 == n and len(kp2) == 3 and kp2.tobytes() == ref[:3].tobytes()
    with pytest.raises(lib.SepfinderError):
        detect(finder, torch, img, 10, 0.0, 3.0)


def test_pixels_to_store_slot_on_the_device(finder):
    """detector -> (stereo correspondence stands in: planted disparities) -> extraction, all on device pointers; the
    slot equals what the oracle chain produces from the same image."""
    import torch
    dev = torch.device("cuda:0")
    image, _, _, _, cam = ec.make_case(31, n=1)
    h, w = image.shape
    n, kp, d_img, d_kp = detect(finder, torch, image, 1000, 0.001, 3.0)
    assert n > 400
    rng = np.random.default_rng(3)
    rx = (kp["x"] - rng.uniform(2.0, 40.0, n)).astype(np.float32)
    st = (rng.random(n) > 0.1).astype(np.uint8)
    tests = ec.brief_tests(9, 32)
    finder.brief_set_pattern(tests)
    d_rx, d_st = torch.from_numpy(rx).to(dev), torch.from_numpy(st).to(dev)
    d_desc = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    d_xyz = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    d_ko = torch.zeros((n, 28), dtype=torch.uint8, device=dev)
    slot, rows = finder.extract_keyframe_device(d_img.data_ptr(), w, h, image.strides[0], d_kp.data_ptr(), d_rx.data_ptr(),
                                                d_st.data_ptr(), n, cam, d_desc.data_ptr(), d_xyz.data_ptr(), d_ko.data_ptr())
    torch.cuda.synchronize()
    ref_kp = pyoracle.detect_corners(image, 1000, 0.001, 3.0)
    d, p, k = pyoracle.extract_keyframe(image, ref_kp, rx, st, cam, tests)
    assert rows == len(d) and d_desc.cpu().numpy()[:rows].tobytes() == d.tobytes()
    got = d_xyz.cpu().numpy()[:rows]
    assert np.array_equal(np.isnan(got), np.isnan(p)) and got[~np.isnan(got)].tobytes() == p[~np.isnan(p)].tobytes()
