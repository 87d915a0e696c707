"""GPU parity of the two-view bundle adjustment (sf_params.bundle_adjustment = 1; myRegistrationVis.cpp:1192-1370,
csrc/k_ba.hip) against the oracle (oracle/sf_oracle_ba.c), through the C-ABI: byte-identical results for both
estimators, on the BASELINE per-pair shapes, mono and stereo residuals, fused and stage kernels, and the edge cases
around the call (outlier words, the min-inliers re-check, frames without 3D in the "to" frame)."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, synth

pytestmark = pytest.mark.gpu


def ba_params(est=0, baseline=0.12, iters=300):
    p = synth.camera_params()
    p.iterations = iters
    p.estimation_type = est
    p.bundle_adjustment = 1
    p.stereo_baseline = baseline
    return p


@pytest.mark.parametrize("est", [0, 1], ids=["3d3d", "pnp"])
@pytest.mark.parametrize("baseline", [0.12, 0.0], ids=["stereo", "mono"])
def test_ba_batch_equals_oracle(oracle, est, baseline):
    from multi_robot_slam_separators_amd import lib
    p = ba_params(est, baseline)
    A, B, is_true, Ts = synth.make_pairs(5100 + est, 32, k=400, cols=32, true_frac=0.5)
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    ref = oracle.estimate_transform_batch(p, A, B, oracle.num_threads())
    n_same = sum(got[i].tobytes() == ref[i].tobytes() for i in range(len(A)))
    for i in range(len(A)):
        for k in ("success", "pass1_success", "pass2_guided", "inliers", "matches", "inliers_pass1", "matches_pass1"):
            assert got[i][k] == ref[i][k], (i, k, got[i][k], ref[i][k])
        assert np.allclose(got[i]["position"], ref[i]["position"], atol=1e-4)
        assert np.allclose(got[i]["orientation"], ref[i]["orientation"], atol=1e-3)
    print("bundle adjustment est=%d baseline=%.2f: %d / %d results byte-identical to the oracle" % (est, baseline, n_same, len(A)))
    assert n_same == len(A)
    assert got["success"][is_true].all() and not got["success"][~is_true].any()
    # it really ran: the pose differs from the run without bundle adjustment
    q = _abi.copy_params(p)
    q.bundle_adjustment = 0
    with lib.SeparatorFinder(q) as f:
        plain = f.estimate_transform_batch(A, B)
    moved = sum(not np.array_equal(plain[i]["position"], got[i]["position"]) for i in np.nonzero(is_true)[0])
    assert moved >= int(is_true.sum()) - 1


@pytest.mark.parametrize("name,k,cols,iters", [("configs[1]", 500, 32, 500), ("configs[2]", 1000, 32, 2000),
                                                ("configs[4]", 500, 64, 500)])
@pytest.mark.parametrize("est", [0, 1], ids=["3d3d", "pnp"])
def test_ba_on_the_baseline_shapes(oracle, name, k, cols, iters, est):
    from multi_robot_slam_separators_amd import lib
    p = ba_params(est, 0.12, iters)
    p.max_features = k
    A, B, is_true, _ = synth.make_pairs(5200 + k + cols + est, 6, k=k, cols=cols, true_frac=0.5)
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    ref = oracle.estimate_transform_batch(p, A, B, oracle.num_threads())
    for i in range(len(A)):
        assert got[i].tobytes() == ref[i].tobytes(), (name, est, i)
    assert got["success"][is_true].all()


def test_ba_fused_equals_stage_kernels_and_edge_cases(oracle, monkeypatch):
    from multi_robot_slam_separators_amd import lib
    p = ba_params(0, 0.12)
    rng = np.random.default_rng(8)
    A, B, is_true, _ = synth.make_pairs(5300, 20, k=300, cols=32, true_frac=0.6)
    # a "to" frame without 3D points (mono observations of camera 2), keypoints that disagree with their 3D points
    # (sbaOutliers), and a pair left with fewer than min_inliers words after the outliers are removed
    a = synth.make_keyframe(rng, 120)
    b, gt = synth.make_true_partner(rng, a, synth.random_transform(rng), overlap=0.6, noise=0.01, flip=0.02)
    bad = _abi.FeatureArrays(b.desc, b.xyz, b.kpts.copy())
    hit = np.nonzero(gt >= 0)[0]
    bad.kpts["x"][hit[:10]] += 45.0
    bad.kpts["y"][hit[:10]] -= 45.0
    A += [a]
    B += [bad]
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SF_FUSED", fused)
        with lib.SeparatorFinder(p) as f:
            out[fused] = f.estimate_transform_batch(A, B)
    assert out["1"].tobytes() == out["0"].tobytes()
    ref = oracle.estimate_transform_batch(p, A, B, 1)
    for i in range(len(A)):
        assert out["1"][i].tobytes() == ref[i].tobytes(), i
    plain = _abi.copy_params(p)
    plain.bundle_adjustment = 0
    r_bad_plain = oracle.estimate_transform(plain, a, bad)
    # outlier words left the pass-1 inliers (pass 2 re-matches around the refined pose, so only pass 1 is comparable)
    assert out["1"][-1]["success"] == 1 and out["1"][-1]["inliers_pass1"] < r_bad_plain["inliers_pass1"]
    # :1331-1336 with min_inliers between what the estimate found and what the adjustment leaves, the pass is null
    q = _abi.copy_params(p)
    q.min_inliers = int(r_bad_plain["inliers_pass1"]) - 3
    monkeypatch.setenv("SF_FUSED", "1")
    with lib.SeparatorFinder(q) as f:
        g = f.estimate_transform(a, bad)
    o = oracle.estimate_transform(q, a, bad)
    assert g.tobytes() == o.tobytes() and g["pass1_success"] == 0 and g["success"] == 0
    q.bundle_adjustment = 0
    assert oracle.estimate_transform(q, a, bad)["pass1_success"] == 1


def test_ba_needs_a_calibrated_camera():
    from multi_robot_slam_separators_amd import lib
    p = ba_params()
    p.image_width = 0
    with pytest.raises(lib.SepfinderError):
        lib.SeparatorFinder(p)
    p = ba_params()
    p.bundle_adjustment = 2          # cvsba: not implemented
    with pytest.raises(lib.SepfinderError):
        lib.SeparatorFinder(p)
