"""GPU: sf_extract_keyframe_device (csrc/k_extract.hip) against the CPU oracle, byte for
byte: the wire copies (descriptors, 3D points, keypoints), the kept-row count, and -- through the verification
path -- the store slot the kernel filled."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, lib, synth
from oracle import pyoracle
from tests import extract_cases as ec

pytestmark = pytest.mark.gpu


def _up(torch, a, dev):
    a = np.ascontiguousarray(a)
    if a.dtype.fields:
        a = a.view(np.uint8)
    return torch.from_numpy(a).to(dev)


def run_extract(f, torch, image, kp, rx, st, cam, n_out=None):
    dev = torch.device("cuda:0")
    h, w = image.shape
    pitch = image.strides[0]
    base = np.lib.stride_tricks.as_strided(image, shape=(h, pitch), strides=(pitch, 1)) if pitch != w else image
    d_img = _up(torch, np.ascontiguousarray(base), dev)
    n = len(kp)
    d_kp = _up(torch, kp, dev) if n else None
    d_rx = _up(torch, rx, dev) if rx is not None and n else None
    d_st = _up(torch, st, dev) if st is not None and n else None
    nb = f.brief_get_pattern().shape[0] // 8
    d_desc = torch.zeros((max(n, 1), nb), dtype=torch.uint8, device=dev)
    d_xyz = torch.zeros((max(n, 1), 3), dtype=torch.float32, device=dev)
    d_kout = torch.zeros((max(n, 1), 28), dtype=torch.uint8, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    slot, rows = f.extract_keyframe_device(ptr(d_img), w, h, pitch, ptr(d_kp), ptr(d_rx), ptr(d_st), n, cam,
                                           d_desc.data_ptr(), d_xyz.data_ptr(), d_kout.data_ptr())
    torch.cuda.synchronize()
    desc = d_desc.cpu().numpy()[:rows]
    xyz = d_xyz.cpu().numpy()[:rows]
    kout = np.frombuffer(d_kout.cpu().numpy().tobytes(), dtype=_abi.KEYPOINT_DTYPE)[:rows]
    return slot, rows, desc, xyz, kout


@pytest.fixture()
def finder():
    import torch
    p = synth.camera_params()
    p.max_features = 2048
    f = lib.SeparatorFinder(p, device=0)
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    yield f
    f.close()


@pytest.mark.parametrize("seed,kw", [
    (1, {}), (2, dict(min_depth=0.8, max_depth=12.0)), (3, dict(identity=True)), (4, dict(max_depth=5.0, identity=True)),
    (5, dict(no_stereo=True)), (6, dict(n=0)), (7, dict(n=1500, width=1280, height=720)), (9, dict(n=257, width=300, height=200)),
])
@pytest.mark.parametrize("nbytes", [32, 64])
def test_extract_equals_oracle(finder, seed, kw, nbytes):
    import torch
    image, kp, rx, st, cam = ec.make_case(seed, **kw)
    tests = ec.brief_tests(100 + seed, nbytes)
    finder.brief_set_pattern(tests)
    slot, rows, desc, xyz, kout = run_extract(finder, torch, image, kp, rx, st, cam)
    d, p, k = pyoracle.extract_keyframe(image, kp, rx, st, cam, tests)
    assert rows == len(d)
    assert desc.tobytes() == d.tobytes()
    assert kout.tobytes() == k.tobytes()
    assert np.array_equal(np.isnan(xyz), np.isnan(p))
    assert xyz[~np.isnan(xyz)].tobytes() == p[~np.isnan(p)].tobytes()
    assert finder.store_size() == slot + 1


def test_store_slot_is_what_the_wire_copy_says(finder):
    """The slot filled by the kernel verifies against a keyframe exactly like the same features ingested from the host."""
    import torch
    image, kp, rx, st, cam = ec.make_case(21, n=900, max_depth=20.0)
    tests = ec.brief_tests(5, 32)
    finder.brief_set_pattern(tests)
    slot, rows, desc, xyz, kout = run_extract(finder, torch, image, kp, rx, st, cam)
    assert rows > 300
    host = _abi.FeatureArrays(desc, xyz, kout)
    slot_host = finder.store_add_keyframe(host)
    # a second view of the same scene: the same corners moved by a rigid motion, some descriptor bits flipped
    rng = np.random.default_rng(77)
    other, _ = synth.make_true_partner(rng, host, synth.random_transform(rng))
    slot_o = finder.store_add_keyframe(other)
    dev = torch.device("cuda:0")
    fr = torch.tensor([slot, slot_host, slot_o, slot_o], dtype=torch.int32, device=dev)
    to = torch.tensor([slot_o, slot_o, slot, slot_host], dtype=torch.int32, device=dev)
    out = torch.zeros((4, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    finder.verify_pairs_device(fr.data_ptr(), to.data_ptr(), 4, out.data_ptr())
    torch.cuda.synchronize()
    res = out.cpu().numpy()
    assert res[0].tobytes() == res[1].tobytes()
    assert res[2].tobytes() == res[3].tobytes()
    # and against itself the extracted keyframe is a perfect revisit
    fr2 = torch.tensor([slot], dtype=torch.int32, device=dev)
    to2 = torch.tensor([slot_host], dtype=torch.int32, device=dev)
    out2 = torch.zeros((1, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    finder.verify_pairs_device(fr2.data_ptr(), to2.data_ptr(), 1, out2.data_ptr())
    torch.cuda.synchronize()
    r = np.frombuffer(out2.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)[0]
    assert r["success"] == 1 and r["inliers"] > 100
    assert np.allclose(r["position"], 0.0, atol=1e-5)


def test_default_pattern_and_errors(finder):
    t = finder.brief_get_pattern()
    assert t.shape == (256, 4) and t.dtype == np.int8 and np.abs(t).max() <= 24 and len(np.unique(t, axis=0)) > 200
    bad = t.copy()
    bad[3, 1] = 30
    with pytest.raises(lib.SepfinderError):
        finder.brief_set_pattern(bad)
    with pytest.raises(lib.SepfinderError):
        finder.brief_set_pattern(t[:8 * 20])       # 20-byte descriptors do not exist
