"""CPU: the C restatement of the keyframe feature extraction (oracle/sf_oracle_extract.c)
against an independent numpy restatement, byte for byte, over the edge cases of tests/extract_cases.py."""
import numpy as np
import pytest

from oracle import pyoracle
from tests import extract_cases as ec


def same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("seed,kw", [
    (1, {}), (2, dict(min_depth=0.8, max_depth=12.0)), (3, dict(identity=True)), (4, dict(max_depth=5.0, identity=True)),
    (5, dict(no_stereo=True)), (6, dict(n=0)), (7, dict(n=1500, width=1280, height=720)), (8, dict(min_depth=-1.0)),
])
@pytest.mark.parametrize("nbytes", [32, 16, 64])
def test_extract_matches_numpy(seed, kw, nbytes):
    image, kp, rx, st, cam = ec.make_case(seed, **kw)
    tests = ec.brief_tests(100 + seed, nbytes)
    d, p, k = pyoracle.extract_keyframe(image, kp, rx, st, cam, tests)
    d2, p2, k2 = ec.numpy_extract(image, kp, rx, st, cam, tests)
    assert same(d, d2)
    assert same(k, k2)
    assert p.shape == p2.shape and np.array_equal(np.isnan(p), np.isnan(p2))
    assert p[~np.isnan(p)].tobytes() == p2[~np.isnan(p2)].tobytes()
    if not kw.get("n", 1) == 0:
        assert 0 < len(d) <= len(kp)


def test_border_and_rounding_rules():
    image, kp, rx, st, cam = ec.make_case(11, n=8)
    h, w = image.shape
    kp["x"][:] = [27.99, 28.0, w - 28.0, w - 28.01, 100.5, 100.49, 300.0, 300.0]
    kp["y"][:] = [100, 100, 100, 100, 28.0, h - 28.01, 27.9, h - 28.0]
    d, p, k = pyoracle.extract_keyframe(image, kp, None, None, cam, ec.brief_tests(1, 32))
    # kept: x = 28.0, x = w - 28.01, the two y-valid ones
    assert [float(v) for v in k["x"]] == [28.0, np.float32(w - 28.01), 100.5, np.float32(100.49)]
    assert np.isnan(p).all()


def test_rejects_offsets_outside_patch():
    image, kp, rx, st, cam = ec.make_case(12, n=4)
    t = ec.brief_tests(1, 32)
    t[5, 2] = 25
    with pytest.raises(RuntimeError):
        pyoracle.extract_keyframe(image, kp, rx, st, cam, t)
