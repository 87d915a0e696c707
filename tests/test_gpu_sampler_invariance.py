"""Sampler invariance of the verification result (the condition SURVEY.md section 7 attached to replacing PCL's
mt19937 / cv::RNG by a keyed hash, DESIGN.md section 3 deviations 3 and 9): different sampler seeds, and the adaptive
stop on or off, draw different hypotheses -- the DECISION (success), the size of the final inlier set and the pose
(within BASELINE.json's 1e-4 m / 1e-3 rad) should not depend on them.  Run on the per-pair shapes of BASELINE
configs[1] / [2] / [4] with both estimators; the number of pairs that violate each property is printed and bounded,
not tuned away.

What the first run of this test showed (MI355X, round 2; numbers in DESIGN.md section 3):
  * 3D-3D (north_star's estimator): refineModel is a fixed-point iteration from the winning hypothesis and lands on
    the same inlier set whatever hypothesis won: 0 pairs differ in decision, inlier count or pose.
  * PnP with pnp_refine_iterations = 0 (cv::solvePnPRansac as rtabmap calls it by default): the result is the
    iterative solve on the inliers OF THE WINNING HYPOTHESIS, with no re-selection -- so inlier count and pose depend
    on the sampler BY CONSTRUCTION of the upstream algorithm (it does with cv::RNG too); only the decision is
    invariant.  The spread is bounded here (centimetres, milliradians: the estimator's own noise level).
  * PnP with rtabmap's refinement rounds (pnp_refine_iterations > 0) re-selects inliers like refineModel does."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, synth

pytestmark = pytest.mark.gpu

SEEDS = (12345, 1, 2, 3, 999, 2024, 77777, 424242)


def _quat_angle(q1, q2):
    d = abs(float(np.dot(q1, q2))) / max(float(np.linalg.norm(q1) * np.linalg.norm(q2)), 1e-300)
    return 2.0 * np.arccos(np.clip(d, -1.0, 1.0))


@pytest.mark.parametrize("name,k,cols,iters", [("configs[1]", 500, 32, 500), ("configs[2]", 1000, 32, 2000),
                                                ("configs[4]", 500, 64, 500)])
@pytest.mark.parametrize("est", [0, 1, 2], ids=["3d3d", "pnp", "pnp_refined"])
def test_result_does_not_depend_on_the_sampler(name, k, cols, iters, est):
    from multi_robot_slam_separators_amd import lib
    n = 40
    A, B, is_true, Ts = synth.make_pairs(31000 + k + cols, n, k=k, cols=cols, true_frac=0.6)
    runs = []
    for adaptive in (1, 0):
        for seed in SEEDS:
            p = synth.camera_params()
            p.iterations = iters
            p.max_features = k
            p.estimation_type = min(est, 1)
            p.pnp_refine_iterations = 5 if est == 2 else 0
            p.seed = seed
            p.ransac_adaptive_stop = adaptive
            with lib.SeparatorFinder(p) as f:
                runs.append(f.estimate_transform_batch(A, B))
    ref = runs[0]
    bad_success = bad_inliers = bad_pose = 0
    worst_t = worst_r = 0.0
    for i in range(n):
        s = {int(r[i]["success"]) for r in runs}
        if len(s) > 1:
            bad_success += 1
            continue
        if not ref[i]["success"]:
            continue
        if len({int(r[i]["inliers"]) for r in runs}) > 1:
            bad_inliers += 1
        dt = max(float(np.linalg.norm(r[i]["position"] - ref[i]["position"])) for r in runs)
        dr = max(_quat_angle(r[i]["orientation"], ref[i]["orientation"]) for r in runs)
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        if dt > 1e-4 or dr > 1e-3:
            bad_pose += 1
    n_acc = int(ref["success"].sum())
    print("%s est=%d: %d pairs (%d accepted) x %d sampler settings: decision differs on %d, inlier count on %d, pose "
          "beyond 1e-4 m / 1e-3 rad on %d (worst %.2e m, %.2e rad)" % (
              name, est, n, n_acc, len(runs), bad_success, bad_inliers, bad_pose, worst_t, worst_r))
    assert ref["success"][is_true].all() and not ref["success"][~is_true].any()
    if est == 2:
        # rtabmap's PnP refinement rounds (off by default) re-select with a 3-sigma reprojection threshold from the
        # winning hypothesis' pose: reported, loosely bounded -- on the first run 4 of 40 decisions depended on the seed
        assert bad_success <= n // 4
        return
    assert bad_success == 0                      # the accept / reject decision never depends on the sampler
    if est == 0:
        # the fixed point of the refinement could differ by a threshold-borderline correspondence: bounded
        assert bad_inliers <= 2 and bad_pose <= 2, (bad_inliers, bad_pose, n_acc)
        assert worst_t < 5e-3 and worst_r < 5e-3
    else:
        # upstream's PnP result is a function of the winning hypothesis' inlier set (see the module docstring)
        assert worst_t < 0.25 and worst_r < 0.03
