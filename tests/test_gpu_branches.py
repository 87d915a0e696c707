"""The adjacent branches of the verification path against the oracle, through the C-ABI:
  Reg/Force3DoF          myRegistration.cpp:245-248, 269-276; myRegistrationVis.cpp:1100-1102, 1141-1143
  Vis/ForwardEstOnly=0   myRegistrationVis.cpp:936-978 (second direction), 1155-1189 (union of ids), 1376-1394 (merge)
Integer outputs bit-exact, poses within 1e-4 m / 1e-3 rad, covariance 1e-9 relative (the bars of test_gpu_verify)."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, synth

from test_gpu_verify import assert_result_parity

pytestmark = pytest.mark.gpu


def _planar_pairs(seed, n, k=300):
    """True pairs whose motion is planar (yaw, x, y) mixed with false ones and general 6-DoF ones."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    A, B = [], []
    for i in range(n):
        a = synth.make_keyframe(rng, k, 32)
        if i % 3 == 0:
            b = synth.make_keyframe(rng, k, 32)
        else:
            T = np.eye(4)
            if i % 3 == 1:
                T[:3, :3] = Rotation.from_rotvec([0, 0, rng.uniform(-0.5, 0.5)]).as_matrix()
                T[:2, 3] = rng.uniform(-1, 1, size=2)
            else:
                T = synth.random_transform(rng)
            b, _ = synth.make_true_partner(rng, a, T, 0.5, 0.01, 0.03)
        A.append(a); B.append(b)
    return A, B


@pytest.mark.parametrize("est,ba", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("fused", ["1", "0", "2"])
def test_force_3dof_parity(monkeypatch, oracle, est, ba, fused):
    """All pipelines (fused kernel / stage kernels / split pipeline), both estimators, with and without the bundle
    adjustment between the two to3DoF applications of a pass."""
    from multi_robot_slam_separators_amd import lib
    A, B = _planar_pairs(91 + est, 24)
    p = synth.camera_params()
    p.iterations = 200
    p.estimation_type = est
    p.force_3dof = 1
    p.bundle_adjustment = ba
    p.stereo_baseline = 0.12 if ba else 0.0
    monkeypatch.setenv("SF_FUSED", fused)
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    n_ok = 0
    for i in range(len(A)):
        o = oracle.estimate_transform(p, A[i], B[i])
        assert_result_parity(got[i], o, "est %d ba %d fused %s pair %d" % (est, ba, fused, i))
        if o["success"]:
            n_ok += 1
            assert got[i]["position"][2] == 0.0 and got[i]["orientation"][0] == 0.0 and got[i]["orientation"][1] == 0.0
    assert n_ok >= 5


@pytest.mark.parametrize("est", [0, 1])
def test_bidirectional_parity(oracle, est):
    """Vis/ForwardEstOnly = false (myRegistrationVis.cpp:936-978, 1155-1189, 1376-1394) with both estimators: the backward
    estimate (frames' roles swapped), the unions of inliers and matches, the half-way interpolation -- byte for byte the
    oracle's; with the PnP estimator also pairs one of whose frames has no 3D points (one direction's gate closed)."""
    from multi_robot_slam_separators_amd import lib
    A, B, is_true, _ = synth.make_pairs(92, 40, k=300, true_frac=0.6)
    if est == 1:
        A = A + [synth.without_3d(a) for a in A[:6]] + A[6:12]
        B = B + B[:6] + [synth.without_3d(b) for b in B[6:12]]
    for f3 in (0, 1):
        p = synth.camera_params()
        p.iterations = 200
        p.estimation_type = est
        p.forward_est_only = 0
        p.force_3dof = f3
        with lib.SeparatorFinder(p) as f:
            f.prof_enable(True)
            got = f.estimate_transform_batch(A, B)
            prof = f.prof_get()
        assert prof["k_verify_fused"][0] == 0        # both directions run on the stage kernels
        p_fwd = _abi.copy_params(p)
        p_fwd.forward_est_only = 1
        n_ok = grew = back_only = 0
        for i in range(len(A)):
            o = oracle.estimate_transform(p, A[i], B[i])
            assert_result_parity(got[i], o, "est %d 3dof %d pair %d" % (est, f3, i))
            n_ok += int(o["success"])
            of = oracle.estimate_transform(p_fwd, A[i], B[i])
            assert o["inliers_pass1"] >= of["inliers_pass1"]        # a union
            grew += int(o["success"] and o["position"].tobytes() != of["position"].tobytes())
            back_only += int(o["success"] and not of["success"])
        assert n_ok >= (3 if f3 else 15)
        assert f3 or grew >= 10                      # the backward estimate really enters the pose (interpolation)
        assert est == 0 or f3 or back_only >= 2      # "from" without 3D points: only the backward PnP can run


@pytest.mark.parametrize("est", [0, 1])
def test_bidirectional_with_bundle_adjustment_parity(oracle, est):
    """Vis/ForwardEstOnly = false with bundle adjustment (myRegistrationVis.cpp:1155-1197, :1369): the forward transform
    refined over the union of both directions' inliers, the backward transform dropped; where there is no forward
    transform the directions merge as without the adjustment.  Byte for byte the oracle's."""
    from multi_robot_slam_separators_amd import lib
    A, B, is_true, _ = synth.make_pairs(93, 32, k=300, true_frac=0.6)
    if est == 1:
        A = A + [synth.without_3d(a) for a in A[:5]] + A[5:10]
        B = B + B[:5] + [synth.without_3d(b) for b in B[5:10]]
    p = synth.camera_params()
    p.iterations = 200
    p.estimation_type = est
    p.forward_est_only = 0
    p.bundle_adjustment = 1
    p.stereo_baseline = 0.12
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    p_fwd = _abi.copy_params(p)
    p_fwd.forward_est_only = 1
    p_noba = _abi.copy_params(p)
    p_noba.bundle_adjustment = 0
    n_ok = differs_fwd = differs_noba = 0
    for i in range(len(A)):
        o = oracle.estimate_transform(p, A[i], B[i])
        assert_result_parity(got[i], o, "est %d pair %d" % (est, i))
        n_ok += int(o["success"])
        differs_fwd += int(o.tobytes() != oracle.estimate_transform(p_fwd, A[i], B[i]).tobytes())
        differs_noba += int(o.tobytes() != oracle.estimate_transform(p_noba, A[i], B[i]).tobytes())
    assert n_ok >= 12
    # both options really act (3D-3D on clean synthetic pairs: the two directions find the same inliers, so the union --
    # and with it the adjusted transform -- is the forward-only one)
    assert differs_noba >= 8 and (est == 0 or differs_fwd >= 8)
