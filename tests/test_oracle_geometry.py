"""Oracle checks for the matching / RANSAC stage.  The reference has no fixtures for this stage
("parity unpinned", oracle/sf_oracle.h), so the oracle is validated against (i) an independent
numpy restatement of the matcher semantics, (ii) numpy SVD Kabsch, (iii) planted ground truth."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, synth


# ---------------------------------------------------------------------------------------------
# independent numpy restatement of myRegistrationVis.cpp:826-895 (brute-force dictionary)
# ---------------------------------------------------------------------------------------------
def np_match_global(df, dt, nndr):
    bf = np.unpackbits(df, axis=1).astype(np.int32)
    bt = np.unpackbits(dt, axis=1).astype(np.int32)
    ham = bt.sum(1)[:, None] + bf.sum(1)[None, :] - 2 * bt @ bf.T        # [to][from]
    order = np.argsort(ham, axis=1, kind="stable")
    d1 = np.take_along_axis(ham, order[:, :1], 1)[:, 0]
    d2 = np.take_along_axis(ham, order[:, 1:2], 1)[:, 0]
    acc = ~(d1.astype(np.float32) > np.float32(nndr) * d2.astype(np.float32))
    ids = np.where(acc, order[:, 0], -1)
    cf, ct = [], []
    for f in range(df.shape[0]):
        ts = np.nonzero(ids == f)[0]
        if len(ts) == 1:
            cf.append(f)
            ct.append(ts[0])
    n_unique_to = int(np.sum(ids < 0)) + len(cf)
    return np.array(cf), np.array(ct), n_unique_to


@pytest.mark.parametrize("seed,kf,kt,cols,nndr", [(1, 64, 64, 32, 0.6), (2, 200, 150, 32, 0.8),
                                                   (3, 37, 91, 64, 0.7), (4, 128, 128, 16, 0.9)])
def test_match_global_vs_numpy(oracle, seed, kf, kt, cols, nndr):
    rng = np.random.default_rng(seed)
    df = rng.integers(0, 256, size=(kf, cols), dtype=np.uint8)
    dt = rng.integers(0, 256, size=(kt, cols), dtype=np.uint8)
    n_copy = min(kf, kt) // 2
    src = rng.permutation(kf)[:n_copy]
    dst = rng.permutation(kt)[:n_copy]
    dt[dst] = synth.flip_bits(rng, df[src], 0.04)
    dt[dst[:4]] = df[src[0]]          # four "to" rows hit the same "from" word -> all dropped
    cf, ct, wf, wt, wt2 = oracle.match_global(df, dt, nndr)
    ecf, ect, eu = np_match_global(df, dt, nndr)
    assert np.array_equal(cf, ecf) and np.array_equal(ct, ect)
    assert wf == kf and wt == eu and wt2 == eu
    assert src[0] not in cf


def test_match_global_edge_cases(oracle):
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, size=(10, 32), dtype=np.uint8)
    # single "from" word: knn returns one neighbour -> always rejected (VWDictionary NNDR rule)
    cf, ct, wf, wt, wt2 = oracle.match_global(d[:1], d, 0.6)
    assert len(cf) == 0 and wt2 == 10
    # empty "to": no wordsTo
    cf, ct, wf, wt, wt2 = oracle.match_global(d, d[:0].reshape(0, 32), 0.6)
    assert len(cf) == 0 and wt2 == 0 and wf == 10
    # identical frames: every row matches itself with d1 = 0
    cf, ct, wf, wt, wt2 = oracle.match_global(d, d, 0.6)
    assert np.array_equal(cf, np.arange(10)) and np.array_equal(ct, np.arange(10))
    # no 3D on the "to" side -> words3To empty
    cf, ct, wf, wt, wt2 = oracle.match_global(d, d, 0.6, True, False)
    assert wt == 0 and wt2 == 10


# ---------------------------------------------------------------------------------------------
def kabsch(src, dst):
    mp, mq = src.mean(0), dst.mean(0)
    H = (src - mp).T @ (dst - mq)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, mq - R @ mp


@pytest.mark.parametrize("n", [3, 4, 10, 300])
def test_fit_rigid_equals_svd_kabsch(oracle, n):
    rng = np.random.default_rng(n)
    for _ in range(20):
        src = rng.normal(size=(n, 3)) * 5
        T = synth.random_transform(rng, 170.0, 5.0)
        dst = src @ T[:3, :3].T + T[:3, 3] + rng.normal(scale=0.01, size=(n, 3))
        R, t = oracle.fit_rigid(src, dst)
        Rk, tk = kabsch(src, dst)
        assert np.allclose(R, Rk, atol=1e-9) and np.allclose(t, tk, atol=1e-8)
        assert abs(np.linalg.det(R) - 1) < 1e-12


def test_fit_rigid_reflection_case(oracle):
    # planar, mirrored target: the optimal PROPER rotation must be returned (det = +1)
    src = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float64)
    dst = src * np.array([1, -1, 1])
    R, t = oracle.fit_rigid(src, dst)
    Rk, tk = kabsch(src, dst)
    assert abs(np.linalg.det(R) - 1) < 1e-12
    res = np.linalg.norm(src @ R.T + t - dst)
    resk = np.linalg.norm(src @ Rk.T + tk - dst)
    assert res <= resk + 1e-9


def test_canon_log(oracle):
    for x in [1e-300, 1e-12, 0.01, 0.010000000000000009, 0.3, 0.5, 0.7071, 0.99999, 1.0, 1.5, 2.0,
              1e6, 1 - 2.2e-16]:
        assert abs(oracle.canon_log(x) - np.log(x)) <= 4e-16 * max(1.0, abs(np.log(x)))


def test_sample_triplet_distinct_uniform(oracle):
    for m in [3, 4, 5, 17, 200, 1000]:
        seen = np.zeros(m, dtype=np.int64)
        for it in range(400):
            s = oracle.sample_triplet(12345, it, 0, m)
            assert len(set(s.tolist())) == 3 and s.max() < m
            seen[s] += 1
        if m <= 17:
            assert seen.min() > 0
    # keyed by (seed, iteration, attempt) only
    assert np.array_equal(oracle.sample_triplet(1, 2, 3, 50), oracle.sample_triplet(1, 2, 3, 50))
    assert not np.array_equal(oracle.sample_triplet(1, 2, 3, 50), oracle.sample_triplet(1, 2, 4, 50))


# ---------------------------------------------------------------------------------------------
def params(iters=500):
    p = synth.camera_params()
    p.iterations = iters
    return p


def test_motion_recovers_planted_transform(oracle):
    rng = np.random.default_rng(5)
    p = params()
    for trial in range(5):
        m = 150
        T = synth.random_transform(rng)
        pts_to = synth.make_points(rng, (m,)).astype(np.float64)
        pts_from = pts_to @ T[:3, :3].T + T[:3, 3] + rng.normal(scale=0.01, size=(m, 3))
        out = rng.permutation(m)[:60]                     # 40 % gross outliers
        pts_from[out] = synth.make_points(rng, (60,))
        idx = np.arange(m, dtype=np.uint16)
        mo, mask = oracle.estimate_motion_3d3d(p, pts_from.astype(np.float32),
                                               pts_to.astype(np.float32), idx, idx)
        assert not mo.is_null and mo.matches == m
        inl = np.ones(m, bool)
        inl[out] = False
        assert mask[inl].sum() >= 0.97 * inl.sum()      # planted inliers recovered
        assert mask[out].sum() <= 2
        Tm = np.eye(4)
        Tm[:3] = np.array(mo.transform).reshape(3, 4)
        assert np.linalg.norm(Tm[:3, 3] - T[:3, 3]) < 0.02
        assert np.arccos(np.clip((np.trace(Tm[:3, :3].T @ T[:3, :3]) - 1) / 2, -1, 1)) < 5e-3
        assert 0 < mo.variance < 0.01


def test_motion_rejects_garbage_and_small_sets(oracle):
    rng = np.random.default_rng(6)
    p = params()
    a = synth.make_points(rng, (80,))
    b = synth.make_points(rng, (80,))
    idx = np.arange(80, dtype=np.uint16)
    mo, _ = oracle.estimate_motion_3d3d(p, a, b, idx, idx)
    assert mo.is_null and mo.matches == 80
    # fewer correspondences than min_inliers: RANSAC not run, covariance stays identity
    mo, _ = oracle.estimate_motion_3d3d(p, a, b, idx[:4], idx[:4])
    assert mo.is_null and mo.variance == 1.0 and mo.inliers == 0
    # non-finite and all-zero points are filtered (util3d::findCorrespondences)
    a2 = a.copy()
    a2[:10] = np.nan
    a2[10:20] = 0
    mo, _ = oracle.estimate_motion_3d3d(p, a2, a2.copy(), idx, idx)
    assert mo.matches == 60 and not mo.is_null and mo.inliers == 60


def test_adaptive_stop_runs_fewer_iterations(oracle):
    rng = np.random.default_rng(7)
    p = params()
    m = 100
    T = synth.random_transform(rng)
    b = synth.make_points(rng, (m,)).astype(np.float64)
    a = (b @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    idx = np.arange(m, dtype=np.uint16)
    mo, _ = oracle.estimate_motion_3d3d(p, a, b.astype(np.float32), idx, idx)
    assert mo.ransac_iterations_run < 10 and mo.inliers == m     # all-inlier data: k collapses
    q = _abi.copy_params(p)
    q.ransac_adaptive_stop = 0
    mo2, _ = oracle.estimate_motion_3d3d(q, a, b.astype(np.float32), idx, idx)
    assert mo2.ransac_iterations_run == p.iterations + 1            # PCL: iterations_ > max


def test_estimate_transform_true_and_false_pairs(oracle):
    p = params()
    A, B, is_true, Ts = synth.make_pairs(11, 16, k=300, cols=32, true_frac=0.5)
    n_true_ok = 0
    for a, b, tr, T in zip(A, B, is_true, Ts):
        r = oracle.estimate_transform(p, a, b)
        if tr:
            assert r["success"] == 1 and r["pass1_success"] == 1 and r["pass2_guided"] == 1
            dt, dr = synth.pose_error(r, T)
            assert dt < 0.05 and dr < 0.01
            assert r["orientation"][3] >= 0
            c = r["covariance"].reshape(6, 6)
            assert np.all(np.diag(c) >= 1e-9) and np.count_nonzero(c - np.diag(np.diag(c))) == 0
            n_true_ok += 1
        else:
            assert r["success"] == 0 and r["inliers"] == 0
            assert np.all(r["position"] == 0) and np.all(r["orientation"] == 0)
            assert np.array_equal(r["covariance"].reshape(6, 6), np.eye(6))
    assert n_true_ok >= 4


def test_estimate_transform_edge_cases(oracle):
    p = params()
    rng = np.random.default_rng(3)
    a = synth.make_keyframe(rng, 50)
    empty = _abi.FeatureArrays(np.zeros((0, 32), np.uint8), np.zeros((0, 3), np.float32),
                               np.zeros(0, _abi.KEYPOINT_DTYPE))
    for f, t in [(a, empty), (empty, a), (empty, empty)]:
        r = oracle.estimate_transform(p, f, t)
        assert r["success"] == 0 and np.array_equal(r["covariance"].reshape(6, 6), np.eye(6))
    # identical frames -> exact identity transform is still a success; pass 2 sees an identity
    # guess and falls back to global matching (myRegistrationVis.cpp:477)
    r = oracle.estimate_transform(p, a, a)
    assert r["success"] == 1 and r["pass2_guided"] == 0 and r["inliers"] == 50
    assert np.allclose(r["position"], 0, atol=1e-5) and abs(r["orientation"][3] - 1) < 1e-6
    assert r["covariance"][0] == 1e-9                     # variance 0 clamped (myRegistration.cpp:284)
    # size mismatch the reference asserts on -> SF_EINVAL
    bad = _abi.FeatureArrays(a.desc, a.xyz[:10], a.kpts)
    with pytest.raises(RuntimeError):
        oracle.estimate_transform(p, bad, a)
    # uncalibrated camera: pass 2 cannot be guided
    q = _abi.copy_params(p)
    q.image_width = 0
    A, B, is_true, Ts = synth.make_pairs(12, 1, k=200, true_frac=1.0)
    r = oracle.estimate_transform(q, A[0], B[0])
    assert r["success"] == 1 and r["pass2_guided"] == 0 and r["inliers"] == r["inliers_pass1"]


def test_guided_matching_semantics(oracle):
    p = params()
    rng = np.random.default_rng(21)
    a = synth.make_keyframe(rng, 120)
    T = synth.random_transform(rng, 5.0, 0.3)
    b, gt = synth.make_true_partner(rng, a, T, overlap=0.6, noise=0.0, flip=0.02)
    guess = T[:3].astype(np.float32)
    cf, ct, wf, wt, wt2, outside = oracle.match_guided(p, guess, a, b)
    assert not outside and wf == 120 and wt == 120
    assert np.all(np.diff(cf.astype(int)) > 0)                    # ascending "from" ids
    assert len(set(ct.tolist())) == len(ct)                        # each "to" claimed once
    good = sum(int(gt[t] == f) for f, t in zip(cf, ct))
    assert good >= 0.8 * len(cf) and good >= 30
    # octave filter: change every "to" octave -> no candidates at all
    b2 = _abi.FeatureArrays(b.desc, b.xyz, b.kpts.copy())
    b2.kpts["octave"] = 3
    cf2, *_ = oracle.match_guided(p, guess, a, b2)
    assert len(cf2) == 0
    # packed SIFT octave: only the low byte, sign-extended, is compared (:709-710)
    b3 = _abi.FeatureArrays(b.desc, b.xyz, b.kpts.copy())
    b3.kpts["octave"] = 0x7F00
    cf3, ct3, *_ = oracle.match_guided(p, guess, a, b3)
    assert np.array_equal(cf3, cf) and np.array_equal(ct3, ct)
    # a guess looking the other way: everything projects outside
    back = np.eye(4)
    back[:3, :3] = np.diag([-1, -1, 1])
    *_, outside = oracle.match_guided(p, back[:3].astype(np.float32), a, b)
    assert outside == 1


def _load_regression():
    import os
    from multi_robot_slam_separators_amd import _abi
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "verify_regression.npz"))
    n = int(z["n"])
    A = [_abi.FeatureArrays(z["desc_a%d" % i], z["xyz_a%d" % i], z["kp_a%d" % i]) for i in range(n)]
    B = [_abi.FeatureArrays(z["desc_b%d" % i], z["xyz_b%d" % i], z["kp_b%d" % i]) for i in range(n)]
    return z, A, B


def test_verification_regression_vectors(oracle):
    """tests/golden/verify_regression.npz (oracle/gen_regression.py): the canonical arithmetic of the
    matching / RANSAC / PnP restatement is frozen byte for byte.  These are regression vectors of the
    restatement, not reference outputs (that stage is PARITY UNPINNED)."""
    from multi_robot_slam_separators_amd import synth
    z, A, B = _load_regression()
    for est in (0, 1):
        p = synth.camera_params()
        p.iterations = 200
        p.estimation_type = est
        want = z["result_est%d" % est]
        for i in range(len(A)):
            got = oracle.estimate_transform(p, A[i], B[i])
            assert got.tobytes() == want[i].tobytes(), "estimation_type %d pair %d" % (est, i)
            assert bool(got["success"]) == bool(z["is_true"][i])


# ---- Reg/Force3DoF and Vis/ForwardEstOnly = false (the adjacent branches of myRegistration.cpp:245-276 and
# myRegistrationVis.cpp:936-978 / 1155-1189 / 1376-1394) -----------------------------------------------------------
def _rot(axis, ang):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis) * ang).as_matrix()


def _T(Rm, t):
    T = np.zeros((3, 4), dtype=np.float32)
    T[:, :3] = Rm
    T[:, 3] = t
    return T


def test_to3dof_is_x_y_yaw(oracle):
    """Transform::to3DoF [upstream]: Transform(x, y, 0, 0, 0, yaw), yaw = atan2(r21, r11) (pcl::getEulerAngles)."""
    import ctypes as C
    L = oracle.lib()
    L.sfo_to3dof.argtypes = [C.c_void_p]
    rng = np.random.default_rng(4)
    for _ in range(50):
        Rm = _rot(rng.normal(size=3), rng.uniform(0, 3.0))
        T = _T(Rm, rng.normal(size=3))
        yaw = np.arctan2(np.float32(Rm[1, 0]), np.float32(Rm[0, 0]))
        want = _T(_rot([0, 0, 1], yaw), [T[0, 3], T[1, 3], 0.0])
        got = T.copy()
        L.sfo_to3dof(got.ctypes.data)
        assert np.abs(got - want).max() < 5e-7
        assert got[2, 3] == 0 and got[2, 2] == 1 and got[0, 3] == T[0, 3] and got[1, 3] == T[1, 3]
    I = _T(np.eye(3), [0, 0, 0])
    got = I.copy()
    L.sfo_to3dof(got.ctypes.data)
    assert np.array_equal(got, I)                 # the identity guess of pass 1 stays the identity (isIdentity, :477)


def test_interpolate_half_is_slerp(oracle):
    """Transform::interpolate(0.5, other) [upstream]: Eigen slerp of the two rotations, midpoint of the translations."""
    import ctypes as C
    from scipy.spatial.transform import Rotation, Slerp
    L = oracle.lib()
    L.sfo_interpolate_half.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    for k in range(60):
        Ra = _rot(rng.normal(size=3), rng.uniform(0, 3.1))
        Rb = Ra @ _rot(rng.normal(size=3), rng.uniform(0, 0.5 if k % 2 else 3.0))
        A, B = _T(Ra, rng.normal(size=3)), _T(Rb, rng.normal(size=3))
        out = np.zeros((3, 4), dtype=np.float32)
        L.sfo_interpolate_half(A.ctypes.data, B.ctypes.data, out.ctypes.data)
        mid = Slerp([0, 1], Rotation.from_matrix(np.stack([A[:, :3], B[:, :3]]).astype(np.float64)))(0.5).as_matrix()
        assert np.abs(out[:, :3] - mid).max() < 2e-6
        assert np.abs(out[:, 3] - 0.5 * (A[:, 3] + B[:, 3])).max() < 1e-6


def test_force_3dof_results_are_planar(oracle):
    A, B, is_true, Ts = synth.make_pairs(77, 12, k=300, true_frac=0.75)
    for est in (0, 1):
        p = synth.camera_params()
        p.iterations = 200
        p.estimation_type = est
        p.force_3dof = 1
        n_ok = 0
        for a, b in zip(A, B):
            r = oracle.estimate_transform(p, a, b)
            if r["success"]:
                n_ok += 1
                assert r["position"][2] == 0.0
                assert abs(r["orientation"][0]) < 1e-7 and abs(r["orientation"][1]) < 1e-7     # rotation about z
        # (a general 6-DoF motion squeezed into 3 DoF as the guess of pass 2 usually loses the pair: that is the
        #  reference's behaviour for a non-planar robot; at least the calls run and decide)
        assert n_ok >= 0


def test_planar_motion_survives_force_3dof(oracle):
    """Pairs whose true motion IS planar (yaw + x, y): with Reg/Force3DoF the result stays close to the 6-DoF estimate."""
    rng = np.random.default_rng(12)
    p0 = synth.camera_params()
    p0.iterations = 300
    p1 = _abi.copy_params(p0)
    p1.force_3dof = 1
    hits = 0
    for _ in range(8):
        a = synth.make_keyframe(rng, 300, 32)
        T = np.eye(4)
        T[:3, :3] = _rot([0, 0, 1], rng.uniform(-0.4, 0.4))
        T[:2, 3] = rng.uniform(-1, 1, size=2)
        b, _ = synth.make_true_partner(rng, a, T, 0.5, 0.01, 0.03)
        r0, r1 = oracle.estimate_transform(p0, a, b), oracle.estimate_transform(p1, a, b)
        if r0["success"] and r1["success"]:
            hits += 1
            assert np.linalg.norm(r0["position"] - r1["position"]) < 0.05
    assert hits >= 5


def test_bidirectional_estimate(oracle):
    """Vis/ForwardEstOnly = false: inliers = union of both directions' (>= the forward estimate's), the pose stays
    within the noise of the forward-only one; PnP / bundle adjustment with it are refused."""
    A, B, is_true, Ts = synth.make_pairs(78, 16, k=300, true_frac=0.75)
    p0 = synth.camera_params()
    p0.iterations = 200
    p1 = _abi.copy_params(p0)
    p1.forward_est_only = 0
    both = 0
    for a, b in zip(A, B):
        r0, r1 = oracle.estimate_transform(p0, a, b), oracle.estimate_transform(p1, a, b)
        assert r1["matches_pass1"] == r0["matches_pass1"]
        assert r1["inliers_pass1"] >= r0["inliers_pass1"]
        if r0["success"] and r1["success"]:
            both += 1
            assert np.linalg.norm(r0["position"] - r1["position"]) < 0.05
    assert both >= 6


@pytest.mark.parametrize("est,ba", [(1, 0), (0, 1), (1, 1)])
def test_bidirectional_estimate_with_pnp_and_bundle_adjustment(oracle, est, ba):
    """Vis/ForwardEstOnly = false with the PnP estimator and / or bundle adjustment (myRegistrationVis.cpp:936-978,
    1155-1197, 1369, 1376-1394): the backward estimate swaps the frames' roles; inliers and matches are unions; with
    bundle adjustment the forward transform is refined over the union and the backward transform is dropped, without it
    the result is the half-way interpolation.  A frame without 3D points takes one direction out (PnP: the one whose 3D
    side it would be)."""
    A, B, is_true, Ts = synth.make_pairs(79, 16, k=300, true_frac=0.75)
    p0 = synth.camera_params()
    p0.iterations = 200
    p0.estimation_type = est
    p0.bundle_adjustment = ba
    p1 = _abi.copy_params(p0)
    p1.forward_est_only = 0
    both = moved = 0
    for a, b in zip(A, B):
        r0, r1 = oracle.estimate_transform(p0, a, b), oracle.estimate_transform(p1, a, b)
        assert r1["matches_pass1"] >= r0["matches_pass1"]
        if not ba:
            assert r1["inliers_pass1"] >= r0["inliers_pass1"]
        if r0["success"] and r1["success"]:
            both += 1
            assert np.linalg.norm(r0["position"] - r1["position"]) < (0.15 if est else 0.08)
            moved += int(r0["position"].tobytes() != r1["position"].tobytes())
    assert both >= 6 and (ba or moved >= 4)       # (with bundle adjustment the backward estimate only adds inliers)
    if est == 1:
        # "to" without 3D points: only the forward PnP can run -> the forward-only result, bit for bit; "from" without:
        # only the backward one -> the inverse of the backward transform (forward-only finds nothing)
        n_back = 0
        for a, b in zip(A[:8], B[:8]):
            b2 = synth.without_3d(b)
            assert oracle.estimate_transform(p1, a, b2).tobytes() == oracle.estimate_transform(p0, a, b2).tobytes()
            a2 = synth.without_3d(a)
            assert not oracle.estimate_transform(p0, a2, b)["success"]
            n_back += int(oracle.estimate_transform(p1, a2, b)["success"])
        assert n_back >= 3 or ba
