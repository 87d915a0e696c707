"""Float32 local descriptors (sf_params.desc_type 1: SURF / SIFT rows) through the verification path: brute-force kNN-2 on
the L2 distance with the NNDR / uniqueness / window rules of the binary path -- north_star's "ORB/SURF ... Hamming/L2
matching" behind PKG/src/myRegistrationVis.cpp:826-895 (global: squared distances, VWDictionary::addNewWords [upstream])
and :476-825 (guided: cv::BFMatcher(NORM_L2) distances).  The reference's own wire cannot carry such rows
(MsgConversion.cpp:113-129 always builds CV_8U), so there is no reference output; the GPU kernels (k_match_global<., L2>,
k_guided<., L2>: exact float32 arithmetic on the VALU) must equal the oracle's restatement byte for byte, both
estimators, 64 and 128 dimensions, ragged and degenerate frames, ties between equal distances."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, lib, synth

pytestmark = pytest.mark.gpu


def _params(dims, est=0, iterations=100):
    p = synth.camera_params()
    p.iterations = iterations
    p.estimation_type = est
    p.desc_type = 1
    p.desc_bytes = 4 * dims
    p.max_features = 256
    return p


def _pairs(seed, n, k, dims, jitter, true_frac=0.5):
    A, B, is_true, Ts = synth.make_pairs(seed, n, k=k, cols=32, true_frac=true_frac)
    rng = np.random.default_rng(seed + 1)
    return ([synth.float_descriptors(a, dims, rng, jitter) for a in A],
            [synth.float_descriptors(b, dims, rng, jitter) for b in B], is_true, Ts)


@pytest.mark.parametrize("est", [0, 1])
@pytest.mark.parametrize("dims,jitter", [(64, 0.0), (64, 0.05), (128, 0.05), (128, 0.0)])
def test_float_descriptors_equal_the_oracle(oracle, dims, jitter, est):
    p = _params(dims, est)
    A, B, is_true, Ts = _pairs(300 + dims + est, 24, 200, dims, jitter)
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    accepted = 0
    for i in range(len(A)):
        o = oracle.estimate_transform(p, A[i], B[i])
        assert got[i].tobytes() == o.tobytes(), (i, {k: (got[i][k], o[k]) for k in ("success", "inliers", "matches", "inliers_pass1", "pass2_guided")})
        accepted += int(got[i]["success"])
        if is_true[i] and got[i]["success"]:
            assert synth.pose_error(got[i], Ts[i])[0] < (0.1 if est == 0 else 0.4)      # (200 features: PnP is the looser one)
    assert accepted >= np.sum(is_true) - 2 and accepted < len(A)        # the true pairs are found, the false ones are not


@pytest.mark.parametrize("dims", [64, 128])
def test_float_descriptor_edge_cases(oracle, dims):
    """Empty / single-row / two-row frames, identical rows on both sides (every distance 0: ties, NNDR 0 < 0.6 * 0 is
    false), rows with NaN (never the nearest), frames of different sizes (at 128 dimensions the scan walks the "from" rows
    in blocks of 16: 15 / 16 / 17 / 33 rows); the store refuses a row width other than 64 or 128 dimensions and a binary
    keyframe in a float handle's width class."""
    p = _params(dims)
    rng = np.random.default_rng(5)
    A, B, _, _ = _pairs(77, 6, 120, dims, 0.02, true_frac=1.0)
    cases = []
    a, b = A[0], B[0]
    cut = lambda fa, n: _abi.FeatureArrays(fa.desc[:n].view(np.float32).reshape(n, -1) if n else np.zeros((0, dims), np.float32),
                                           fa.xyz[:n], fa.kpts[:n])
    cases.append((cut(a, 0), b)); cases.append((a, cut(b, 0))); cases.append((cut(a, 1), b)); cases.append((cut(a, 2), cut(b, 2)))
    cases.append((cut(a, 120), cut(b, 50)))
    for n in (15, 16, 17, 33):
        cases.append((cut(a, n), b))
    same = _abi.FeatureArrays(np.tile(a.desc[:1].view(np.float32), (120, 1)), a.xyz, a.kpts)
    cases.append((same, same))
    d = a.desc.view(np.float32).reshape(120, -1).copy(); d[::7] = np.nan
    cases.append((_abi.FeatureArrays(d, a.xyz, a.kpts), b))
    cases.append((A[1], B[1]))
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch([c[0] for c in cases], [c[1] for c in cases])
        for i, (x, y) in enumerate(cases):
            o = oracle.estimate_transform(p, x, y)
            assert got[i].tobytes() == o.tobytes(), i
        assert got[-1]["success"] == 1
        bad = _abi.FeatureArrays(np.zeros((10, 48), np.float32), a.xyz[:10], a.kpts[:10])
        with pytest.raises(lib.SepfinderError):
            f.store_add_keyframe(bad)


def test_empty_first_keyframe_on_a_default_width_float_handle(oracle):
    """A float-descriptor handle created with the width sf_default_params leaves (32, a binary width) must accept an EMPTY
    keyframe as its first store entry (the reference tolerates empty keyframes: the fake-words path) -- round 4 derived a
    32-byte float row from it and the store refused the keyframe.  The width is taken as 64 dimensions until the first
    non-empty keyframe arrives."""
    dims = 64
    p = _params(dims)
    p.desc_bytes = 32                       # (what sf_default_params writes)
    A, B, _, _ = _pairs(78, 2, 120, dims, 0.02, true_frac=1.0)
    empty = _abi.FeatureArrays(np.zeros((0, 0), np.float32), A[0].xyz[:0], A[0].kpts[:0])
    with lib.SeparatorFinder(p) as f:
        s0 = f.store_add_keyframe(empty)
        s1 = f.store_add_keyframe(A[0])
        s2 = f.store_add_keyframe(B[0])
        got = f.verify_pairs(np.array([s1, s0], np.int32), np.array([s2, s2], np.int32))
        o = oracle.estimate_transform(_params(dims), A[0], B[0])
        assert got[0].tobytes() == o.tobytes() and got[0]["success"] == 1
        assert got[1]["success"] == 0 and got[1]["matches"] == 0
