"""GPU parity tests of the verification path (matching + RANSAC + two-pass driver), through the
C-ABI, against the CPU oracle on identical seeded inputs.

Bars: correspondences, match / inlier counts and success flags are integer work -> bit-exact;
poses within BASELINE.json's 1e-4 m / 1e-3 rad; covariance within 1e-9 relative."""
import ctypes as C

import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, synth

pytestmark = pytest.mark.gpu

POS_TOL = 1e-4   # metres   (BASELINE.json north_star)
ROT_TOL = 1e-3   # radians


@pytest.fixture(scope="module")
def finder():
    from multi_robot_slam_separators_amd import lib
    p = synth.camera_params()
    p.iterations = 500
    f = lib.SeparatorFinder(p)
    # the tests that inspect correspondence lists (sf_debug_correspondences) need them copied out of LDS; that the
    # option changes no result byte is checked by test_debug_option_changes_no_result and the fuzz test
    f.set_option(_abi.SF_OPT_DEBUG_CORR, 1)
    yield f
    f.close()


def quat_angle(q1, q2):
    d = abs(float(np.dot(q1, q2))) / max(np.linalg.norm(q1) * np.linalg.norm(q2), 1e-300)
    return 2.0 * np.arccos(np.clip(d, -1.0, 1.0))


def assert_result_parity(g, o, ctx=""):
    for k in ("success", "pass1_success", "pass2_guided", "inliers", "matches", "inliers_pass1",
              "matches_pass1"):
        assert g[k] == o[k], "%s %s: gpu %s oracle %s" % (ctx, k, g[k], o[k])
    if o["success"]:
        assert np.linalg.norm(g["position"] - o["position"]) <= POS_TOL, ctx
        assert quat_angle(g["orientation"], o["orientation"]) <= ROT_TOL, ctx
    else:
        assert np.all(g["position"] == 0) and np.all(g["orientation"] == 0)
    assert np.allclose(g["covariance"], o["covariance"], rtol=1e-9, atol=0), ctx


def test_batch_parity_mixed_pairs(finder, oracle):
    A, B, is_true, Ts = synth.make_pairs(2024, 48, k=500, cols=32, true_frac=0.4)
    got = finder.estimate_transform_batch(A, B)
    n_exact = 0
    for i in range(len(A)):
        o, c1, c2 = oracle.estimate_transform(finder.params, A[i], B[i], debug=True)
        g1 = finder.debug_correspondences(i, 1)
        assert np.array_equal(g1[0], c1[0]) and np.array_equal(g1[1], c1[1]), "pass-1 correspondences %d" % i
        if o["pass2_guided"]:
            g2 = finder.debug_correspondences(i, 2)
            assert np.array_equal(g2[0], c2[0]) and np.array_equal(g2[1], c2[1]), "pass-2 correspondences %d" % i
        assert_result_parity(got[i], o, "pair %d" % i)
        if is_true[i]:
            assert got[i]["success"] == 1
            dt, dr = synth.pose_error(got[i], Ts[i])
            assert dt < 0.05 and dr < 0.01
        n_exact += int(np.array_equal(got[i]["position"], o["position"])
                       and np.array_equal(got[i]["orientation"], o["orientation"])
                       and np.array_equal(got[i]["covariance"], o["covariance"]))
    # canonical arithmetic: the GPU and the oracle are expected to agree bit for bit
    print("bit-identical results: %d / %d" % (n_exact, len(A)))
    assert n_exact >= len(A) - 2


def test_single_call_equals_batch_entry(finder, oracle):
    A, B, _, _ = synth.make_pairs(7, 3, k=300, true_frac=1.0)
    batch = finder.estimate_transform_batch(A, B)
    for i in range(3):
        single = finder.estimate_transform(A[i], B[i])
        assert single.tobytes() == batch[i].tobytes()    # n = 1 reproduces one service call


@pytest.mark.parametrize("k,cols,iters", [(1000, 32, 2000), (500, 64, 500), (200, 16, 300), (77, 32, 100), (2000, 32, 300)])
def test_other_configs(oracle, k, cols, iters):
    from multi_robot_slam_separators_amd import lib
    p = synth.camera_params()
    p.iterations = iters
    p.max_features = k
    A, B, is_true, _ = synth.make_pairs(100 + k, 10, k=k, cols=cols, true_frac=0.5)
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    for i in range(len(A)):
        assert_result_parity(got[i], oracle.estimate_transform(p, A[i], B[i]), "k=%d pair %d" % (k, i))
    assert got["success"][is_true].all()


def test_store_resident_pairs(finder, oracle):
    rng = np.random.default_rng(5)
    A, B, is_true, _ = synth.make_pairs(55, 12, k=400, true_frac=0.5)
    finder.store_clear()
    slots_a = [finder.store_add_keyframe(a) for a in A]
    slots_b = [finder.store_add_keyframe(b) for b in B]
    assert finder.store_size() == 24
    # arbitrary pairing, including cross pairs and a keyframe against itself
    fs = slots_a + [slots_a[0], slots_b[3], slots_a[2]]
    ts = slots_b + [slots_a[0], slots_a[3], slots_b[5]]
    feats = {s: f for s, f in zip(slots_a + slots_b, A + B)}
    got = finder.verify_pairs(fs, ts)
    for i, (s, t) in enumerate(zip(fs, ts)):
        assert_result_parity(got[i], oracle.estimate_transform(finder.params, feats[s], feats[t]), "slot pair %d" % i)
    with pytest.raises(Exception):
        finder.verify_pairs([0], [999])
    finder.store_clear()
    _ = rng


def test_edge_cases(finder, oracle):
    rng = np.random.default_rng(3)
    p = finder.params
    a = synth.make_keyframe(rng, 60)
    empty = _abi.FeatureArrays(np.zeros((0, 32), np.uint8), np.zeros((0, 3), np.float32),
                               np.zeros(0, _abi.KEYPOINT_DTYPE))
    one = synth.make_keyframe(rng, 1)
    no3d = _abi.FeatureArrays(a.desc, np.zeros((0, 3), np.float32), a.kpts)
    nan3d = _abi.FeatureArrays(a.desc, a.xyz.copy(), a.kpts)
    nan3d.xyz[::2] = np.nan
    nan3d.xyz[1] = 0.0
    cases = [(a, empty), (empty, a), (empty, empty), (a, a), (one, a), (a, one), (a, no3d), (no3d, a),
             (nan3d, a), (a, nan3d), (nan3d, nan3d)]
    got = finder.estimate_transform_batch([c[0] for c in cases], [c[1] for c in cases])
    for i, (f, t) in enumerate(cases):
        assert_result_parity(got[i], oracle.estimate_transform(p, f, t), "edge %d" % i)
    assert got[3]["success"] == 1 and got[3]["covariance"][0] == 1e-9
    # size mismatches the reference UASSERTs on -> SF_EINVAL, never a crash
    bad = _abi.FeatureArrays(a.desc, a.xyz[:10], a.kpts)
    with pytest.raises(Exception):
        finder.estimate_transform(bad, a)
    bad2 = _abi.FeatureArrays(a.desc, a.xyz, a.kpts[:5])
    with pytest.raises(Exception):
        finder.estimate_transform(a, bad2)
    wide = synth.make_keyframe(rng, 60, cols=64)
    with pytest.raises(Exception):
        finder.estimate_transform(a, wide)


def test_duplicate_descriptors_and_ties(finder, oracle):
    # many identical descriptors: NNDR ties (d1 == d2) must reject, d1 = d2 = 0 accepts lowest index
    rng = np.random.default_rng(9)
    a = synth.make_keyframe(rng, 128)
    a.desc[10:20] = a.desc[10]
    b, _ = synth.make_true_partner(rng, a, synth.random_transform(rng), overlap=0.7, noise=0.01, flip=0.0)
    b.desc[:5] = a.desc[10]
    got = finder.estimate_transform(a, b)
    assert_result_parity(got, oracle.estimate_transform(finder.params, a, b), "ties")
    g = finder.debug_correspondences(0, 1)
    o = oracle.match_global(a.desc, b.desc, finder.params.nndr)
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])


def test_parameter_variants(oracle):
    from multi_robot_slam_separators_amd import lib
    A, B, _, _ = synth.make_pairs(31, 8, k=300, true_frac=0.75)
    variants = []
    for kw in [dict(ransac_adaptive_stop=0), dict(refine_iterations=0), dict(guess_win_size=0),
               dict(min_inliers=50), dict(nndr=0.9), dict(image_width=0), dict(seed=999),
               dict(inlier_distance=0.02), dict(guess_win_size=5), dict(refine_sigma=1.0)]:
        p = synth.camera_params()
        p.iterations = 200
        for k, v in kw.items():
            setattr(p, k, v)
        variants.append((kw, p))
    for kw, p in variants:
        with lib.SeparatorFinder(p) as f:
            got = f.estimate_transform_batch(A, B)
        for i in range(len(A)):
            assert_result_parity(got[i], oracle.estimate_transform(p, A[i], B[i]), "%s pair %d" % (kw, i))


def test_unsupported_estimators_are_rejected():
    from multi_robot_slam_separators_amd import lib
    p = synth.camera_params()
    p.estimation_type = 2          # epipolar geometry (myRegistrationVis.cpp:979-1054) is not implemented
    with pytest.raises(lib.SepfinderError):
        lib.SeparatorFinder(p)


def test_octave_and_window_semantics(finder, oracle):
    rng = np.random.default_rng(77)
    a = synth.make_keyframe(rng, 200)
    T = synth.random_transform(rng, 8.0, 0.5)
    b, _ = synth.make_true_partner(rng, a, T, overlap=0.6, noise=0.01, flip=0.03)
    a.kpts["octave"] = rng.integers(0, 3, size=200)
    b.kpts["octave"] = rng.integers(0, 3, size=200) | 0x100   # only the low byte is compared
    got = finder.estimate_transform(a, b)
    o, c1, c2 = oracle.estimate_transform(finder.params, a, b, debug=True)
    assert_result_parity(got, o, "octaves")
    g2 = finder.debug_correspondences(0, 2)
    assert np.array_equal(g2[0], c2[0]) and np.array_equal(g2[1], c2[1])


@pytest.mark.parametrize("k", [300, 900])
def test_guided_pass_with_more_candidates_than_its_list_holds(oracle, k):
    """Every feature of both frames inside a 40-pixel spot: each projected point of the guess-guided pass has (almost)
    every keypoint of the other frame inside its window -- tens of thousands of (from, to) combinations against the 2 048
    the candidate-parallel search can record, so the per-lane search answers.  Same correspondences, same result."""
    rng = np.random.default_rng(500 + k)

    def narrow(f):
        xyz = f.xyz.copy()
        xyz[:, 1] *= 0.06
        xyz[:, 2] *= 0.06                                   # (base frame: x forward) -> projections within ~20 px of the centre
        u, v = synth.project_base_points(xyz)
        kp = f.kpts.copy()
        kp["x"], kp["y"] = u, v
        return _abi.FeatureArrays(f.desc, xyz, kp)

    a = narrow(synth.make_keyframe(rng, k))
    T = synth.random_transform(rng, 0.5, 0.03)
    b, _ = synth.make_true_partner(rng, a, T, overlap=0.7, noise=0.002, flip=0.02)
    far = np.abs(b.kpts["x"] - synth.CX) > 40.0             # the partner's NEW points come from the whole image: squeeze them too
    xyz = b.xyz.copy()
    xyz[far, 1] *= 0.06
    xyz[far, 2] *= 0.06
    u, v = synth.project_base_points(xyz)
    kp = b.kpts.copy()
    kp["x"], kp["y"] = u, v
    b = _abi.FeatureArrays(b.desc, xyz, kp)
    # (the guess moves a projection by a few pixels at most: pairs within 12 px of each other are inside the 20 px window)
    du = a.kpts["x"][:, None] - b.kpts["x"][None, :]
    dv = a.kpts["y"][:, None] - b.kpts["y"][None, :]
    assert int((du * du + dv * dv < 12.0 ** 2).sum()) > 4 * 2048
    from multi_robot_slam_separators_amd import lib
    p = synth.camera_params()
    p.iterations = 300
    with lib.SeparatorFinder(p) as f:
        f.set_option(_abi.SF_OPT_DEBUG_CORR, 1)
        got = f.estimate_transform(a, b)
        o, c1, c2 = oracle.estimate_transform(f.params, a, b, debug=True)
        assert_result_parity(got, o, "spot %d" % k)
        g2 = f.debug_correspondences(0, 2)
        assert np.array_equal(g2[0], c2[0]) and np.array_equal(g2[1], c2[1])
        assert len(c2[0]) > 0                                # the guided pass ran and matched


def test_size_independent_properties_full_config(finder):
    """BASELINE configs[1] sizes (K=500, 256-bit, 500 iterations) on a larger batch: properties that
    need no oracle -- planted transforms recovered, false pairs rejected, batch order irrelevant."""
    d = synth.make_store_batch(4242, 256, k=500, cols=32, true_frac=0.25)
    import torch
    finder.store_clear()
    dev = torch.device("cuda:0")
    tens = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.uint8) if v.dtype.fields else v).to(dev)
            for k, v in d.items() if k not in ("is_true", "T_gt")}
    torch.cuda.synchronize()
    n = 256
    first_a = finder.store_add_keyframes_device(n, 500, 32, tens["desc_a"].data_ptr(), tens["xyz_a"].data_ptr(),
                                                tens["kp_a"].data_ptr())
    first_b = finder.store_add_keyframes_device(n, 500, 32, tens["desc_b"].data_ptr(), tens["xyz_b"].data_ptr(),
                                                tens["kp_b"].data_ptr())
    fs = np.arange(n, dtype=np.int32) + first_a
    ts = np.arange(n, dtype=np.int32) + first_b
    res = finder.verify_pairs(fs, ts)
    assert np.array_equal(res["success"].astype(bool), d["is_true"])
    for i in np.nonzero(d["is_true"])[0]:
        dt, dr = synth.pose_error(res[i], d["T_gt"][i])
        assert dt < 0.03 and dr < 5e-3
        assert res[i]["inliers"] >= 20 and res[i]["inliers_pass1"] >= 150
    perm = np.random.default_rng(1).permutation(n)
    res2 = finder.verify_pairs(fs[perm], ts[perm])
    assert res2.tobytes() == res[perm].tobytes()            # results do not depend on batch position
    # symmetry: swapping from/to inverts the pose (up to the RANSAC tolerance)
    j = int(np.nonzero(d["is_true"])[0][0])
    r_ab = finder.verify_pairs([fs[j]], [ts[j]])[0]
    r_ba = finder.verify_pairs([ts[j]], [fs[j]])[0]
    assert r_ba["success"] == 1
    Tinv = np.linalg.inv(d["T_gt"][j])
    dt, dr = synth.pose_error(r_ba, Tinv)
    assert dt < 0.03 and dr < 5e-3 and r_ab["success"] == 1
    finder.store_clear()


def test_rccl_allgather_separators_single_rank(finder):
    """The C-ABI's own RCCL exchange (sf_comm_*, sf_allgather_separators) with a 1-rank communicator:
    counts + records come back unchanged.  (Multi-rank runs need one GPU per rank; the N > 1 logic is
    covered by the gloo tests and by the driver's multi-GPU bench.)"""
    import torch
    from multi_robot_slam_separators_amd import lib
    A, B, _, _ = synth.make_pairs(41, 6, k=200, true_frac=0.5)
    res = finder.estimate_transform_batch(A, B)
    sep = lib.pack_separators(res, 0, 1, np.arange(6), np.arange(6) + 10, np.arange(6), np.arange(6))
    acc = sep[sep["transform_est_success"] == 1]
    d_local = torch.from_numpy(acc.view(np.uint8).reshape(len(acc), -1).copy()).cuda()
    cap = 8
    d_all = torch.zeros((cap, _abi.SEPARATOR_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    finder.comm_init(lib.comm_unique_id(), 0, 1)
    try:
        counts = finder.allgather_separators(d_local.data_ptr(), len(acc), d_all.data_ptr(), cap, 1)
        assert counts.tolist() == [len(acc)]
        back = np.frombuffer(d_all.cpu().numpy().tobytes(), dtype=_abi.SEPARATOR_DTYPE)[: len(acc)]
        assert back.tobytes() == acc.tobytes()
        with pytest.raises(lib.SepfinderError):
            finder.allgather_separators(d_local.data_ptr(), cap + 1, d_all.data_ptr(), cap, 1)
        # the no-host form: header slot with a device-stamped count in front of the records, one collective
        RB = _abi.SEPARATOR_DTYPE.itemsize
        d_send = torch.zeros((cap + 1, RB), dtype=torch.uint8, device="cuda")
        d_send[1: 1 + len(acc)] = d_local
        d_send[0, :4].view(torch.int32).fill_(len(acc))          # stamped on the device
        d_blk = torch.zeros((cap + 1, RB), dtype=torch.uint8, device="cuda")
        finder.set_stream(torch.cuda.current_stream().cuda_stream)
        finder.allgather_separators_device(d_send.data_ptr(), d_blk.data_ptr(), cap)
        torch.cuda.synchronize()
        assert int(d_blk[0, :4].view(torch.int32).item()) == len(acc)
        assert d_blk[1: 1 + len(acc)].cpu().numpy().tobytes() == acc.tobytes()
    finally:
        finder.comm_destroy()


def test_torch_rccl_allgather_records_single_rank():
    """bench.py's exchange (torch.distributed, backend "nccl" = RCCL) on a 1-rank group: the same
    dist.allgather_records call the N > 1 bench makes, on device tensors."""
    import os
    import socket
    import torch
    import torch.distributed as td
    from multi_robot_slam_separators_amd import dist
    if td.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rec = torch.arange(5 * 368, dtype=torch.int64, device=dev).remainder(251).to(torch.uint8).reshape(5, 368)
        out, counts = dist.allgather_records(rec)
        assert counts == [5] and torch.equal(out, rec)
        out, counts = dist.allgather_records(rec[:0])
        assert counts == [0] and out.shape == (0, 368)
        for cap in (8, 5, 2):        # the one-collective variant, with and without the overflow fallback
            out, counts = dist.allgather_records_fixed(rec, cap)
            assert counts == [5] and torch.equal(out, rec)
        for cap in (8, 2):           # the persistent-buffer form bench.py runs every step (device buffers, RCCL)
            ex = dist.RecordExchange(368, 16, cap, dev)
            for k in (5, 3, 0):
                ex.payload[:k] = rec[:k]
                ex.exchange(k)
                torch.cuda.synchronize()
                assert ex.counts() == [k]
                out, counts = ex.all_gathered()
                assert counts == [k] and torch.equal(out, rec[:k]), (cap, k)
        # the count stamped by the producer on the device (bench.py: sf_compact_accepted_device_async writes records
        # and count straight into the send buffer), the collective left in flight beside a copy to the host
        from multi_robot_slam_separators_amd import lib as _lib
        ex = dist.RecordExchange(368, 16, 8, dev)
        big = torch.zeros((12, 368), dtype=torch.uint8, device=dev)
        big[:, 0] = torch.arange(12, device=dev).to(torch.uint8)
        big[::2, _abi.RESULT_DTYPE.fields["success"][1]] = 1
        with _lib.SeparatorFinder(synth.camera_params()) as f2:
            f2.set_stream(torch.cuda.current_stream().cuda_stream)
            f2.compact_accepted_device_async(big.data_ptr(), 12, ex.payload.data_ptr(), None, ex.count_ptr)
            ex.exchange(None, finish=False)
            host = torch.empty((6, 368), dtype=torch.uint8).pin_memory()
            host.copy_(ex.payload[:6], non_blocking=True)
            ex.finish()
            torch.cuda.synchronize()
            out, counts = ex.all_gathered()
            assert counts == [6] and torch.equal(out, big[::2]) and torch.equal(host, big[::2].cpu())
        t = torch.tensor([3.5], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        td.barrier()
        assert float(t.item()) == 3.5
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("est", [0, 1])
def test_verification_regression_vectors_on_gpu(est):
    """The HIP kernels reproduce tests/golden/verify_regression.npz byte for byte (both estimators)."""
    from multi_robot_slam_separators_amd import lib
    from test_oracle_geometry import _load_regression
    z, A, B = _load_regression()
    p = synth.camera_params()
    p.iterations = 200
    p.estimation_type = est
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    want = z["result_est%d" % est]
    for i in range(len(A)):
        assert got[i].tobytes() == want[i].tobytes(), "estimation_type %d pair %d" % (est, i)


@pytest.mark.parametrize("est", [0, 1])
def test_fused_pipeline_equals_stage_kernels(monkeypatch, est):
    """k_verify_fused (one launch per chunk) and the five stage kernels (SF_FUSED=0) run the same
    per-pair bodies: results, correspondences and counts must be identical byte for byte."""
    from multi_robot_slam_separators_amd import lib
    A, B, is_true, _ = synth.make_pairs(515, 40, k=500, cols=32, true_frac=0.4)
    A2, B2, _, _ = synth.make_pairs(516, 12, k=300, cols=64, true_frac=0.5)
    for AA, BB in ((A, B), (A2, B2)):
        p = synth.camera_params()
        p.iterations = 300
        p.estimation_type = est
        out = {}
        monkeypatch.setenv("SF_DEBUG_CORR", "1")      # (the fused kernel keeps the lists in LDS otherwise)
        for fused in ("1", "0"):
            monkeypatch.setenv("SF_FUSED", fused)
            with lib.SeparatorFinder(p) as f:
                f.prof_enable(True)
                res = f.estimate_transform_batch(AA, BB)
                corr = [f.debug_correspondences(i, w) for i in range(len(AA)) for w in (1, 2)]
                prof = f.prof_get()
            out[fused] = (res, corr, prof)
        if est == 0:      # (the PnP estimator always runs the stage kernels)
            assert out["1"][2]["k_verify_fused"][0] >= 1 and out["1"][2]["k_match_global"][0] == 0
        assert out["0"][2]["k_verify_fused"][0] == 0 and out["0"][2]["k_match_global"][0] >= 1
        assert out["1"][0].tobytes() == out["0"][0].tobytes()
        for c1, c0 in zip(out["1"][1], out["0"][1]):
            assert np.array_equal(c1[0], c0[0]) and np.array_equal(c1[1], c0[1])


def test_split_pipeline_equals_fused_kernel(monkeypatch):
    """SF_FUSED=2 (k_match_split over all pairs + k_chain over the survivors, the pass-1 lists of the survivors taking
    one trip through HBM) runs the same per-pair bodies as k_verify_fused: identical bytes, identical lists, on both
    descriptor widths, with ragged / empty frames in the batch and with bundle adjustment on."""
    from multi_robot_slam_separators_amd import lib
    from test_gpu_fuzz import random_frame
    rng = np.random.default_rng(7)
    A, B, _, _ = synth.make_pairs(615, 60, k=500, cols=32, true_frac=0.4)
    A += [random_frame(rng, 0, 32), random_frame(rng, 37, 32), A[0]]
    B += [random_frame(rng, 20, 32), random_frame(rng, 0, 32), A[0]]
    A2, B2, _, _ = synth.make_pairs(616, 16, k=300, cols=64, true_frac=0.5)
    for AA, BB, ba in ((A, B, 0), (A2, B2, 0), (A[:24], B[:24], 1)):
        p = synth.camera_params()
        p.iterations = 300
        p.bundle_adjustment = ba
        p.stereo_baseline = 0.12 if ba else 0.0
        out = {}
        monkeypatch.setenv("SF_DEBUG_CORR", "1")
        for mode in ("1", "2"):
            monkeypatch.setenv("SF_FUSED", mode)
            with lib.SeparatorFinder(p) as f:
                f.prof_enable(True)
                res = f.estimate_transform_batch(AA, BB)
                corr = [f.debug_correspondences(i, w) for i in range(len(AA)) for w in (1, 2)]
                prof = f.prof_get()
            out[mode] = (res, corr, prof)
        # (the split's matching launch; with the bundle adjustment on, a launch of its own since round 5, the chain is cut
        #  around it and the handle's default form is the split one too)
        assert out["2"][2]["k_match_global"][0] >= 1 and (out["1"][2]["k_match_global"][0] == 0) == (ba == 0)
        if ba:
            assert out["1"][2]["k_ba_pass"][0] >= 2 and out["2"][2]["k_ba_pass"][0] >= 2
        assert out["1"][0].tobytes() == out["2"][0].tobytes()
        for c1, c2 in zip(out["1"][1], out["2"][1]):
            assert np.array_equal(c1[0], c2[0]) and np.array_equal(c1[1], c2[1])


def test_pipelined_scan_at_its_tile_boundaries(monkeypatch, oracle):
    """k_match_split's scan is software-pipelined inside a wavefront (k_match.hip, mf_pipe_*: two accumulator tuples, the
    top-2 update of a column tile issued in the gaps of the next tile's MFMAs, the ragged last "from" tile and groups of
    fewer than two column tiles on the non-pipelined statements).  Every combination of "from" rows around the 32-row tile
    edges (no full tile, exactly one, one + a ragged one, ...) and of "to" rows around the group sizes (one column tile
    per wavefront or none, two, four, more than one group) against the oracle, correspondence lists included."""
    from multi_robot_slam_separators_amd import lib
    from test_gpu_fuzz import random_frame
    rng = np.random.default_rng(2029)
    base_a, base_b, _, _ = synth.make_pairs(91, 1, k=520, cols=32, true_frac=1.0)
    a, b = base_a[0], base_b[0]
    cut = lambda fa, n: _abi.FeatureArrays(fa.desc[:n], fa.xyz[:n], fa.kpts[:n])
    kfs = (1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 129, 500, 511, 512, 520)
    kts = (1, 31, 32, 33, 64, 127, 128, 129, 255, 256, 257, 300, 500, 512, 520)
    A, B = [], []
    for i, kf in enumerate(kfs):            # a diagonal + two off-diagonals of the grid, and random frames of the same sizes
        for kt in (kts[i], kts[(i + 5) % len(kts)], kts[(i + 10) % len(kts)]):
            A.append(cut(a, kf)); B.append(cut(b, kt))
            A.append(random_frame(rng, kf, 32)); B.append(random_frame(rng, kt, 32))
    p = synth.camera_params()
    p.iterations = 200
    p.min_inliers = 5
    monkeypatch.setenv("SF_FUSED", "2")
    monkeypatch.setenv("SF_DEBUG_CORR", "1")
    with lib.SeparatorFinder(p) as f:
        f.prof_enable(True)
        got = f.estimate_transform_batch(A, B)
        assert f.prof_get()["k_match_global"][0] >= 1          # (the split form's matching launch)
        for i in range(len(A)):
            o, c1, c2 = oracle.estimate_transform(p, A[i], B[i], debug=True)
            g1 = f.debug_correspondences(i, 1)
            assert np.array_equal(g1[0], c1[0]) and np.array_equal(g1[1], c1[1]), (i, len(A[i].desc), len(B[i].desc))
            assert got[i].tobytes() == o.tobytes(), (i, len(A[i].desc), len(B[i].desc))
    assert sum(int(g["success"]) for g in got) >= 5


@pytest.mark.parametrize("est", [0, 1])
@pytest.mark.parametrize("ba", [0, 1])
def test_chain_widths_give_the_same_bytes(monkeypatch, ba, est):
    """The survivors' chains on one, two or four wavefronts (SF_CHAIN_NW: k_chain<W, NW, PART>; SF_CHAIN_PNP_NW:
    k_chain_pnp<W, PART, NW>, whose narrow forms also recompute the bearings instead of keeping them in LDS) and the
    bundle adjustment on one, two or four (SF_BA_NW; k_ba_pass<NW, ...>): the sums keep the canonical 256-lane order
    whatever the width, so results and pass-2 lists are the same bytes -- and equal the oracle's."""
    from multi_robot_slam_separators_amd import lib
    from oracle import pyoracle
    from test_gpu_fuzz import random_frame
    rng = np.random.default_rng(17)
    A, B, _, _ = synth.make_pairs(715, 40, k=500, cols=32, true_frac=0.5)
    A += [random_frame(rng, 0, 32), random_frame(rng, 37, 32), A[0], A[1]]
    B += [random_frame(rng, 20, 32), random_frame(rng, 0, 32), A[0], B[1]]
    p = synth.camera_params()
    p.iterations = 300
    p.estimation_type = est
    p.bundle_adjustment = ba
    p.stereo_baseline = 0.12 if ba else 0.0
    monkeypatch.setenv("SF_FUSED", "2")
    monkeypatch.setenv("SF_DEBUG_CORR", "1")
    out = {}
    for nw in ("4", "2", "1"):
        monkeypatch.setenv("SF_CHAIN_NW", nw)
        monkeypatch.setenv("SF_CHAIN_PNP_NW", nw)
        monkeypatch.setenv("SF_BA_NW", nw)
        with lib.SeparatorFinder(p) as f:
            res = f.estimate_transform_batch(A, B)
            corr = [f.debug_correspondences(i, 2) for i in range(len(A))]
        out[nw] = (res, corr)
    for nw in ("2", "1"):
        assert out[nw][0].tobytes() == out["4"][0].tobytes(), nw
        for c4, cn in zip(out["4"][1], out[nw][1]):
            assert np.array_equal(c4[0], cn[0]) and np.array_equal(c4[1], cn[1])
    ref = pyoracle.estimate_transform_batch(p, A, B, pyoracle.num_threads())
    for i in range(len(A)):
        assert out["1"][0][i].tobytes() == ref[i].tobytes(), i
    assert int(out["1"][0]["success"].sum()) >= 15


@pytest.mark.parametrize("fused", ["1", "0"])
def test_matrix_core_matcher_equals_valu_matcher(monkeypatch, fused):
    """The Hamming table on the fp4 matrix cores (default) and on the VALU (SF_MATCH_MFMA=0) must give the
    same correspondences -- exact +-1 products, lowest-index tie rule carried in the accumulator's fraction
    -- for ragged sizes (rows not a multiple of the 32-row tile, fewer rows than one tile, 1 and 0 rows),
    both descriptor widths, short descriptors and duplicated descriptors (distance ties)."""
    from multi_robot_slam_separators_amd import lib
    from test_gpu_fuzz import random_frame
    rng = np.random.default_rng(99)
    for cols in (32, 64, 16):
        A, B = [], []
        for ka, kb in ((500, 500), (257, 31), (31, 257), (33, 500), (1, 40), (40, 1), (0, 10), (10, 0), (2, 2), (96, 64)):
            a = random_frame(rng, ka, cols)
            if ka >= 8 and kb >= 8:
                b, _ = synth.make_true_partner(rng, a, synth.random_transform(rng, 20, 1.0), overlap=0.6, noise=0.02,
                                               flip=0.05)
                if kb < ka:
                    b = _abi.FeatureArrays(b.desc[:kb].copy(), b.xyz[:kb].copy(), b.kpts[:kb].copy())
                elif kb > ka:      # pad with unrelated features
                    e = random_frame(rng, kb - ka, cols)
                    b = _abi.FeatureArrays(np.concatenate([b.desc, e.desc]), np.concatenate([b.xyz, e.xyz]),
                                           np.concatenate([b.kpts, e.kpts]))
            else:
                b = random_frame(rng, kb, cols)
            if ka >= 40:      # exact duplicates inside the "from" frame: equal best distances, lowest row wins
                a.desc[ka // 2] = a.desc[3]
                a.desc[ka - 1] = a.desc[3]
            A.append(a); B.append(b)
        p = synth.camera_params()
        p.iterations = 100
        p.max_features = 500
        out = {}
        for mf in ("1", "0"):
            monkeypatch.setenv("SF_MATCH_MFMA", mf)
            monkeypatch.setenv("SF_FUSED", fused)
            monkeypatch.setenv("SF_DEBUG_CORR", "1")
            with lib.SeparatorFinder(p) as f:
                res = f.estimate_transform_batch(A, B)
                corr = [f.debug_correspondences(i, w) for i in range(len(A)) for w in (1, 2)]
            out[mf] = (res, corr)
        assert out["1"][0].tobytes() == out["0"][0].tobytes(), cols
        for c1, c0 in zip(out["1"][1], out["0"][1]):
            assert np.array_equal(c1[0], c0[0]) and np.array_equal(c1[1], c0[1])
        assert sum(len(c[0]) for c in out["1"][1]) > 100      # the comparison is not vacuous


@pytest.mark.parametrize("est", [0, 1])
def test_two_stream_batches_equal_single_stream(monkeypatch, est):
    """SF_OVERLAP=1 cuts a batch in two halves that run the stage kernels on two streams (the second on a
    shadow workspace): results and correspondences (both halves, through sf_debug_correspondences) must be
    identical to the single-stream path, also across repeated calls and odd batch sizes."""
    from multi_robot_slam_separators_amd import lib
    A, B, is_true, _ = synth.make_pairs(731, 37, k=300, cols=32, true_frac=0.5)
    p = synth.camera_params()
    p.iterations = 200
    p.estimation_type = est
    out = {}
    for mode in ("single", "two"):
        monkeypatch.setenv("SF_OVERLAP", "1" if mode == "two" else "0")
        monkeypatch.setenv("SF_OVERLAP_MIN", "2")
        monkeypatch.setenv("SF_DEBUG_CORR", "1")
        with lib.SeparatorFinder(p) as f:
            f.prof_enable(True)
            res = [f.estimate_transform_batch(A[:n], B[:n]) for n in (37, 5, 36)]
            corr = [f.debug_correspondences(i, w) for i in range(36) for w in (1, 2)]
            prof = f.prof_get()
        out[mode] = (res, corr, prof)
    # launch sequences of the single-stream path: one k_verify_fused each (3D-3D), or one matching launch + one chain
    # launch each (PnP: k_match_split + k_chain_pnp, or the stage kernels); the two-stream path runs two per batch
    ps = out["single"][2]
    sequences = ps["k_match_global"][0] if ps["k_match_global"][0] else ps["k_verify_fused"][0]
    assert out["two"][2]["k_match_global"][0] == 2 * sequences
    for r2, r1 in zip(out["two"][0], out["single"][0]):
        assert r2.tobytes() == r1.tobytes()
    for c2, c1 in zip(out["two"][1], out["single"][1]):
        assert np.array_equal(c2[0], c1[0]) and np.array_equal(c2[1], c1[1])
    assert out["single"][0][0]["success"][is_true].all()


@pytest.mark.parametrize("est", [0, 1])
def test_accepted_results_streamed_from_the_kernel(est):
    """sf_accept_stream_*: the fused kernel writes every accepted result into a pinned block as it becomes final.  The
    streamed records, keyed by their index, are exactly the accepted results the gathered form returns -- on the
    speculative path; the fallback paths report streamed = 0 and leave the block alone."""
    import torch
    from multi_robot_slam_separators_amd import lib
    n_kf, k, cols, dim = 96, 200, 32, 512
    feats = synth.make_store_batch(81, n_kf, k=k, cols=cols, true_frac=0.5)
    rng = np.random.default_rng(8)
    nv_a = rng.normal(size=(n_kf, dim)); nv_a /= np.linalg.norm(nv_a, axis=1, keepdims=True)
    nv_b = nv_a + 0.002 * rng.normal(size=(n_kf, dim)); nv_b /= np.linalg.norm(nv_b, axis=1, keepdims=True)
    nv_b[10] = nv_b[11] = nv_b[12]            # rows sharing a column: candidates that are not matches
    dev = torch.device("cuda:0")
    p = synth.camera_params()
    p.estimation_type = est
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.netvlad_distance = 0.13
    p.iterations = 200
    p.max_features = k

    def up(x):
        x = np.ascontiguousarray(x)
        return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        T = {key: up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}
        sa = f.store_add_keyframes_device(n_kf, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
        sb = f.store_add_keyframes_device(n_kf, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
        torch.cuda.synchronize()
        f.nn_append_received(nv_a); f.nn_append_local(nv_b)
        cap = 512
        blocks = []
        for which in (0, 1):
            rec = torch.zeros((cap, 368), dtype=torch.uint8).pin_memory()
            idx = torch.full((cap,), -1, dtype=torch.int32).pin_memory()
            fl = torch.full((cap,), 7, dtype=torch.uint8).pin_memory()
            if which == 0:
                f.accept_stream_set(which, rec.data_ptr(), idx.data_ptr(), fl.data_ptr(), cap)
            else:                              # a second (device) copy of every record and a caller-owned counter
                rec2 = torch.zeros((cap, 368), dtype=torch.uint8, device=dev)
                ctr = torch.zeros(2, dtype=torch.int32, device=dev)
                f.accept_stream_set(which, rec.data_ptr(), idx.data_ptr(), fl.data_ptr(), cap, rec2.data_ptr(), ctr.data_ptr())
            blocks.append((rec, idx, fl))
        d_res = torch.zeros((n_kf, 368), dtype=torch.uint8, device=dev)
        for rep in range(4):
            which = rep & 1
            rec, idx, fl = blocks[which]
            rec.zero_(); idx.fill_(-1); fl.fill_(7)
            if which == 1:
                rec2.zero_(); ctr.zero_()
            f.accept_stream_select(which)
            m = f.find_matches_and_verify_device(sa, sb, d_res.data_ptr(), cap=n_kf)
            torch.cuda.synchronize()
            streamed, pairs = f.accept_stream_status()
            assert streamed and pairs >= len(m)
            r, ix, n = f.last_match_results()
            index_of_match = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_int32)), shape=(n,)).copy()
            res = np.frombuffer(d_res.cpu().numpy()[:n].tobytes(), dtype=_abi.RESULT_DTYPE)
            flags = fl.numpy()[:pairs]
            assert set(np.unique(flags)) <= {0, 1}
            n_acc = int(flags.sum())
            got_idx = idx.numpy()[:n_acc]
            assert (got_idx >= 0).all() and (idx.numpy()[n_acc:] == -1).all() and len(set(got_idx.tolist())) == n_acc
            recs = np.frombuffer(rec.numpy()[:n_acc].tobytes(), dtype=_abi.RESULT_DTYPE)
            assert recs["success"].all() and (flags[got_idx] == 1).all()
            if which == 1:
                assert int(ctr[0]) == n_acc and int(ctr[1]) == 0
                assert torch.equal(rec2[:n_acc].cpu(), rec[:n_acc]) and not rec2[n_acc:].any()
            by_index = {int(j): recs[i] for i, j in enumerate(got_idx)}
            acc_matches = [i for i in range(n) if res["success"][i]]
            assert len(acc_matches) >= 10 and n_acc >= len(acc_matches)
            for i in range(n):
                j = int(index_of_match[i])
                assert bool(flags[j]) == bool(res["success"][i])
                if res["success"][i]:
                    assert by_index[j].tobytes() == res[i].tobytes()
        # off: nothing is written
        rec, idx, fl = blocks[0]
        rec.zero_(); idx.fill_(-1); fl.fill_(7)
        f.accept_stream_select(-1)
        f.find_matches_and_verify_device(sa, sb, d_res.data_ptr(), cap=n_kf)
        torch.cuda.synchronize()
        assert f.accept_stream_status() == (False, 0) and (idx.numpy() == -1).all() and (fl.numpy() == 7).all()
        # the fallback (a small cap: no speculation) does not stream either
        f.accept_stream_select(0)
        f.find_matches_and_verify_device(sa, sb, d_res.data_ptr(), cap=7)
        torch.cuda.synchronize()
        assert f.accept_stream_status()[0] is False and (idx.numpy() == -1).all()
        # a block with fewer record slots than the launch has pairs could lose accepted results: it is not armed
        # (streamed = 0, untouched) and the caller takes the compaction
        f.accept_stream_set(0, rec.data_ptr(), idx.data_ptr(), fl.data_ptr(), n_kf)     # < n_kf + n_kf / 8 + 256 slots
        f.accept_stream_select(0)
        m = f.find_matches_and_verify_device(sa, sb, d_res.data_ptr(), cap=n_kf)
        torch.cuda.synchronize()
        assert len(m) > 10 and f.accept_stream_status()[0] is False and (idx.numpy() == -1).all() and (fl.numpy() == 7).all()


def test_indexed_compaction_of_the_last_match_results():
    """sf_find_matches_and_verify_device with d_out = NULL + sf_last_match_results + the indexed compaction give the
    accepted records, flags and count of the gathered form (d_out given, then sf_compact_accepted_device), on the
    speculative path and on the fallback (small cap: no speculation, results in an internal block)."""
    import torch
    from multi_robot_slam_separators_amd import lib
    n_kf, k, cols, dim = 96, 200, 32, 512
    feats = synth.make_store_batch(78, n_kf, k=k, cols=cols, true_frac=0.5)
    rng = np.random.default_rng(6)
    nv_a = rng.normal(size=(n_kf, dim)); nv_a /= np.linalg.norm(nv_a, axis=1, keepdims=True)
    nv_b = nv_a + 0.002 * rng.normal(size=(n_kf, dim)); nv_b /= np.linalg.norm(nv_b, axis=1, keepdims=True)
    dev = torch.device("cuda:0")
    for cap, maxnb in ((n_kf, n_kf), (10, 10)):
        p = synth.camera_params()
        p.netvlad_dimensions = dim
        p.netvlad_max_matches_nb = maxnb
        p.netvlad_distance = 0.13
        p.iterations = 200
        p.max_features = k
        with lib.SeparatorFinder(p) as f:
            f.set_stream(torch.cuda.current_stream().cuda_stream)

            def up(x):
                x = np.ascontiguousarray(x)
                return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
            T = {key: up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}   # kept alive
            sa = f.store_add_keyframes_device(n_kf, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
            sb = f.store_add_keyframes_device(n_kf, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
            torch.cuda.synchronize()
            f.nn_append_received(nv_a); f.nn_append_local(nv_b)
            d_res = torch.zeros((n_kf, 368), dtype=torch.uint8, device=dev)
            m1 = f.find_matches_and_verify_device(sa, sb, d_res.data_ptr(), cap=cap)
            acc1 = torch.zeros((n_kf, 368), dtype=torch.uint8, device=dev)
            fl1 = torch.zeros(n_kf, dtype=torch.uint8, device=dev)
            k1 = f.compact_accepted_device(d_res.data_ptr(), len(m1), acc1.data_ptr(), fl1.data_ptr())
            m2 = f.find_matches_and_verify_device(sa, sb, None, cap=cap)
            r, ix, n2 = f.last_match_results()
            assert n2 == len(m2) == len(m1) >= 8 and m2.tobytes() == m1.tobytes() and r
            assert (ix is not None) == (cap == n_kf)          # an index list only on the speculative path
            acc2 = torch.zeros((n_kf, 368), dtype=torch.uint8, device=dev)
            fl2 = torch.zeros(n_kf, dtype=torch.uint8, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            f.compact_accepted_indexed_device_async(r, ix, n2, acc2.data_ptr(), fl2.data_ptr(), cnt.data_ptr())
            torch.cuda.synchronize()
            assert int(cnt[0]) == k1 and k1 >= 1
            assert torch.equal(acc2[:k1], acc1[:k1]) and torch.equal(fl2[:n2], fl1[:n2])
            # mirrored: every output twice (a device block and a pinned host block)
            acc3 = torch.zeros((n_kf, 368), dtype=torch.uint8, device=dev)
            fl3 = torch.zeros(n_kf, dtype=torch.uint8, device=dev)
            cnt3 = torch.zeros(1, dtype=torch.int32, device=dev)
            h_blk = torch.zeros(16 + n_kf + n_kf * 368, dtype=torch.uint8).pin_memory()
            hb = h_blk.data_ptr()
            f.compact_accepted_indexed_mirrored_device_async(r, ix, n2, acc3.data_ptr(), fl3.data_ptr(), cnt3.data_ptr(),
                                                             hb + 16 + n_kf, hb + 16, hb)
            torch.cuda.synchronize()
            assert int(cnt3[0]) == k1 == int(h_blk[:4].view(torch.int32)[0])
            assert torch.equal(acc3[:k1], acc1[:k1]) and torch.equal(fl3[:n2], fl1[:n2])
            assert torch.equal(h_blk[16 + n_kf: 16 + n_kf + k1 * 368].view(k1, 368), acc1[:k1].cpu())
            assert torch.equal(h_blk[16: 16 + n2], fl1[:n2].cpu())


@pytest.mark.parametrize("est", [0, 1])
def test_find_matches_and_verify_speculative_equals_two_calls(est):
    """sf_find_matches_and_verify_device (speculative verification of the NN candidates while the host walks
    them) returns the matches of sf_nn_find_matches and, per match, the bytes sf_verify_matches_device writes --
    with masked rows / columns, ignored pairs, several local rows sharing one nearest column (the walk keeps
    the first), rows without a candidate, and on the fallback paths (small cap; exact NN precision)."""
    import torch
    from multi_robot_slam_separators_amd import lib
    n_kf, k, cols, dim = 96, 200, 32, 512
    feats = synth.make_store_batch(77, n_kf, k=k, cols=cols, true_frac=0.5)
    rng = np.random.default_rng(5)
    nv_a = rng.normal(size=(n_kf, dim)); nv_a /= np.linalg.norm(nv_a, axis=1, keepdims=True)
    nv_b = nv_a + 0.002 * rng.normal(size=(n_kf, dim)); nv_b /= np.linalg.norm(nv_b, axis=1, keepdims=True)
    nv_b[10] = nv_b[11] = nv_b[12]            # three local rows whose nearest received column is 12
    nv_b[40:44] = rng.normal(size=(4, dim)) * 3.0      # rows with no candidate under the threshold
    p = synth.camera_params()
    p.iterations = 200
    p.estimation_type = est
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.netvlad_distance = 0.13
    p.max_features = k
    dev = torch.device("cuda:0")

    def up(x):
        x = np.ascontiguousarray(x)
        return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)

    for precision, cap in ((1, n_kf), (1, 7), (0, n_kf)):
        p.nn_precision = precision
        with lib.SeparatorFinder(p) as f:
            f.set_stream(torch.cuda.current_stream().cuda_stream)
            T = {key: up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}   # kept alive
            sa = f.store_add_keyframes_device(n_kf, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
            sb = f.store_add_keyframes_device(n_kf, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
            torch.cuda.synchronize()
            f.nn_append_received(nv_a)
            f.nn_append_local(nv_b)
            f.nn_mark_local_used(3); f.nn_mark_other_used(5); f.nn_ignore_pair(20, 20); f.nn_ignore_pair(21, 22)
            d1 = torch.zeros((n_kf, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
            d2 = torch.zeros_like(d1)
            for rep in range(2):                       # twice: buffers and events are reused
                m1 = f.nn_find_matches(cap=cap)
                f.verify_matches_device(m1, sa, sb, d1.data_ptr())
                torch.cuda.synchronize()
                m2 = f.find_matches_and_verify_device(sa, sb, d2.data_ptr(), cap=cap)
                torch.cuda.synchronize()
                assert m1.tobytes() == m2.tobytes(), (precision, cap)
                n = len(m1)
                assert n > 0 and bytes(d1[:n].cpu().numpy()) == bytes(d2[:n].cpu().numpy()), (precision, cap, rep)
            if cap == n_kf:
                rows = m1["idx_local"].tolist()
                assert 3 not in rows and 20 not in rows and not set(range(40, 44)) & set(rows)
                assert sum(r in (10, 11, 12) for r in rows) == 1
                res = np.frombuffer(bytes(d2[:n].cpu().numpy()), dtype=_abi.RESULT_DTYPE)
                assert res["success"].sum() >= 10


def test_find_matches_and_verify_when_the_speculation_overflows():
    """More NN candidates than speculative pair slots (every local row has four received columns under the
    threshold): the speculative verification is abandoned and the call must still return exactly what the two
    separate calls return."""
    import torch
    from multi_robot_slam_separators_amd import lib
    n_l, k, cols, dim = 600, 64, 32, 512
    feats = synth.make_store_batch(78, n_l, k=k, cols=cols, true_frac=0.5)
    rng = np.random.default_rng(6)
    base = rng.normal(size=(n_l, dim)); base /= np.linalg.norm(base, axis=1, keepdims=True)
    recv = np.concatenate([base + 0.001 * rng.normal(size=base.shape) for _ in range(4)], axis=0)   # 4 copies each
    recv /= np.linalg.norm(recv, axis=1, keepdims=True)
    p = synth.camera_params()
    p.iterations = 100
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_l
    p.max_features = k
    p.nn_precision = 1
    dev = torch.device("cuda:0")

    def up(x):
        x = np.ascontiguousarray(x)
        return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)

    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        T = {key: up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}
        # the received database has 4 n_l rows: keyframe slot = column modulo n_l is not needed here, the store
        # simply holds 4 copies of robot A's keyframes so that every column has a slot
        sa = None
        for _ in range(4):
            s0 = f.store_add_keyframes_device(n_l, k, cols, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
            sa = s0 if sa is None else sa
        sb = f.store_add_keyframes_device(n_l, k, cols, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
        torch.cuda.synchronize()
        f.nn_append_received(recv)
        f.nn_append_local(base)
        d1 = torch.zeros((n_l, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        d2 = torch.zeros_like(d1)
        m1 = f.nn_find_matches(cap=n_l)
        f.verify_matches_device(m1, sa, sb, d1.data_ptr())
        torch.cuda.synchronize()
        m2 = f.find_matches_and_verify_device(sa, sb, d2.data_ptr(), cap=n_l)
        torch.cuda.synchronize()
        assert len(m1) == n_l and m1.tobytes() == m2.tobytes()
        assert bytes(d1.cpu().numpy()) == bytes(d2.cpu().numpy())
        res = np.frombuffer(bytes(d2.cpu().numpy()), dtype=_abi.RESULT_DTYPE)
        assert res["success"].sum() > n_l // 4


@pytest.mark.parametrize("name,k,cols,iters", [("configs[1]", 500, 32, 500), ("configs[2]", 1000, 32, 2000),
                                                ("configs[4]", 500, 64, 500)])
@pytest.mark.parametrize("est", [0, 1])
def test_baseline_config_shapes_against_the_oracle(oracle, name, k, cols, iters, est):
    """The per-pair shapes of BASELINE.json's configs (features per keyframe, descriptor width, RANSAC iterations)
    on a handful of pairs each, both estimators: the GPU result equals the oracle's byte for byte."""
    from multi_robot_slam_separators_amd import lib
    A, B, is_true, _ = synth.make_pairs(1000 + k + cols + est, 6, k=k, cols=cols, true_frac=0.5)
    p = synth.camera_params()
    p.iterations = iters
    p.max_features = k
    p.estimation_type = est
    with lib.SeparatorFinder(p) as f:
        got = f.estimate_transform_batch(A, B)
    ref = oracle.estimate_transform_batch(p, A, B, oracle.num_threads())
    for i in range(len(A)):
        assert got[i].tobytes() == ref[i].tobytes(), (name, est, i)
    assert got["success"][is_true].all()


def test_execution_options_on_a_live_handle():
    """sf_set_option switches matcher / fusion / two-stream execution between calls of ONE handle; every
    combination returns the same bytes; an unknown option is SF_EINVAL."""
    from multi_robot_slam_separators_amd import lib
    A, B, is_true, _ = synth.make_pairs(912, 24, k=400, cols=32, true_frac=0.5)
    p = synth.camera_params()
    p.iterations = 200
    with lib.SeparatorFinder(p) as f:
        base = f.estimate_transform_batch(A, B)
        assert base["success"][is_true].all()
        for mfma, fused, dbg in ((0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1), (1, 0, 1)):
            f.set_option(_abi.SF_OPT_MATCH_MFMA, mfma)
            f.set_option(_abi.SF_OPT_FUSED, fused)
            f.set_option(_abi.SF_OPT_DEBUG_CORR, dbg)     # 1: correspondence lists also copied to the workspace
            f.set_option(_abi.SF_OPT_CHAIN_WAVES, 1)      # (round 1's option: accepted, no effect)
            f.prof_reset(); f.prof_enable(True)
            got = f.estimate_transform_batch(A, B)
            prof = f.prof_get()
            assert got.tobytes() == base.tobytes(), (mfma, fused, dbg)
            assert (prof["k_verify_fused"][0] > 0) == bool(fused)
            if fused and not dbg:       # the lists stayed in LDS: asking for them is an error, not garbage
                with pytest.raises(lib.SepfinderError):
                    f.debug_correspondences(0, 1)
            else:
                assert len(f.debug_correspondences(0, 1)[0]) >= 0
        with pytest.raises(Exception):
            f.set_option(99, 1)


def test_fused_pipeline_with_more_than_64k_of_lds(oracle):
    """K = 700 (kcap 1024) and 6 000 iterations: the fused kernel's dynamic LDS exceeds 64 KiB (RANSAC
    count table), which needs the raised per-kernel limit."""
    from multi_robot_slam_separators_amd import lib
    p = synth.camera_params()
    p.iterations = 6000
    p.ransac_adaptive_stop = 0
    p.max_features = 700
    A, B, is_true, _ = synth.make_pairs(808, 6, k=700, cols=32, true_frac=0.5)
    with lib.SeparatorFinder(p) as f:
        f.prof_enable(True)
        got = f.estimate_transform_batch(A, B)
        assert f.prof_get()["k_verify_fused"][0] >= 1
    for i in range(len(A)):
        assert_result_parity(got[i], oracle.estimate_transform(p, A[i], B[i]), "pair %d" % i)
    assert got["success"][is_true].all()


def test_verify_matches_and_compaction_entry_points(finder):
    """sf_verify_matches_device (candidates straight from an NN query) and sf_compact_accepted_device (ordered
    accepted-only compaction) against the explicit slot lists / a numpy mask."""
    import torch
    rng = np.random.default_rng(9)
    A, B, is_true, _ = synth.make_pairs(901, 30, k=300, true_frac=0.5)
    finder.store_clear()
    sa = [finder.store_add_keyframe(a) for a in A][0]
    sb = [finder.store_add_keyframe(b) for b in B][0]
    # a synthetic match list: local (computing robot B) index i <-> other (querying robot A) index perm[i]
    perm = np.arange(30)
    perm[20:] = 20 + rng.permutation(10)           # the first 20 candidates keep their planted partner
    m = np.zeros(30, dtype=_abi.MATCH_DTYPE)
    m["idx_local"], m["idx_other"], m["distance"] = np.arange(30), perm, rng.random(30)
    dev = torch.device("cuda", 0)
    d_res = torch.zeros((30, 368), dtype=torch.uint8, device=dev)
    d_acc = torch.zeros((30, 368), dtype=torch.uint8, device=dev)
    d_flags = torch.zeros(30, dtype=torch.uint8, device=dev)
    assert finder.verify_matches_device(m, sa, sb, d_res.data_ptr()) == 30
    n_acc = finder.compact_accepted_device(d_res.data_ptr(), 30, d_acc.data_ptr(), d_flags.data_ptr())
    got = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
    want = finder.verify_pairs([sa + int(p) for p in perm], [sb + i for i in range(30)])
    assert got.tobytes() == want.tobytes()
    ok = want["success"] != 0
    assert n_acc == int(ok.sum()) and n_acc >= 1
    assert np.array_equal(d_flags.cpu().numpy().astype(bool), ok)
    acc = np.frombuffer(d_acc[:n_acc].cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
    assert acc.tobytes() == want[ok].tobytes()                     # candidate order preserved
    # more than one 1024-candidate chunk, no flags, nothing accepted / everything accepted
    big = torch.zeros((2500, 368), dtype=torch.uint8, device=dev)
    out = torch.zeros((2500, 368), dtype=torch.uint8, device=dev)
    assert finder.compact_accepted_device(big.data_ptr(), 2500, out.data_ptr()) == 0
    off = _abi.RESULT_DTYPE.fields["success"][1]
    big[::3, off] = 1
    big[:, 0] = torch.arange(2500, device=dev).remainder(251).to(torch.uint8)
    k = finder.compact_accepted_device(big.data_ptr(), 2500, out.data_ptr())
    assert k == len(range(0, 2500, 3)) and torch.equal(out[:k], big[::3])
    # the asynchronous form: same records and flags, the count left on the device
    out2 = torch.zeros_like(out)
    flags2 = torch.zeros(2500, dtype=torch.uint8, device=dev)
    cnt = torch.full((2,), -7, dtype=torch.int32, device=dev)
    finder.compact_accepted_device_async(big.data_ptr(), 2500, out2.data_ptr(), flags2.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    assert cnt.tolist() == [k, -7] and torch.equal(out2[:k], big[::3])
    assert torch.equal(flags2.bool(), big[:, off] != 0)
    finder.compact_accepted_device_async(big.data_ptr(), 0, out2.data_ptr(), None, cnt.data_ptr())
    torch.cuda.synchronize()
    assert cnt.tolist() == [0, -7]
    # repeated launches of the one-kernel form (epoch tags), sizes around the chunk, a size that grows the chunk-state
    # array, one beyond 128 chunks (the two-kernel form) and back again
    rng = np.random.default_rng(8)
    for n_big in (1, 1023, 1024, 1025, 5000, 70_000, 140_000, 3000):
        flags = rng.random(n_big) < 0.3
        src = torch.zeros((n_big, 368), dtype=torch.uint8, device=dev)
        src[:, off] = torch.from_numpy(flags.astype(np.uint8)).to(dev)
        src[:, 1] = torch.arange(n_big, device=dev).remainder(253).to(torch.uint8)
        dst = torch.zeros_like(src)
        kk = finder.compact_accepted_device(src.data_ptr(), n_big, dst.data_ptr())
        assert kk == int(flags.sum())
        assert torch.equal(dst[:kk], src[torch.from_numpy(flags).to(dev)])
    with pytest.raises(Exception):
        m_bad = m.copy(); m_bad["idx_other"][0] = 10_000
        finder.verify_matches_device(m_bad, sa, sb, d_res.data_ptr())
    finder.store_clear()


def test_empty_keyframe_on_a_512_bit_store(oracle):
    """Featureless frames exist in the reference (no corners -> rows = cols = 0).  On a store of 64-byte
    descriptors the per-tick sf_store_add_keyframe used to default such a frame to the 32-byte width class and
    reject it; both ingest paths must accept it and verify it as the oracle does."""
    from multi_robot_slam_separators_amd import lib
    rng = np.random.default_rng(77)
    p = synth.camera_params()
    p.desc_bytes = 64
    a = synth.make_keyframe(rng, 120, cols=64)
    empty = _abi.FeatureArrays(np.zeros((0, 0), np.uint8), np.zeros((0, 3), np.float32),
                               np.zeros(0, _abi.KEYPOINT_DTYPE))
    with lib.SeparatorFinder(p) as f:
        s0 = f.store_add_keyframe(a)
        s1 = f.store_add_keyframe(empty)       # per-tick path (one keyframe per call)
        s2 = f.store_add_keyframe(a)
        got = f.verify_pairs([s0, s1, s0], [s1, s0, s2])
        assert got[2]["success"] == 1
        for g, (x, y) in zip(got, [(a, empty), (empty, a), (a, a)]):
            assert_result_parity(g, oracle.estimate_transform(p, x, y))
        got_b = f.estimate_transform_batch([a, empty], [empty, a])   # batch path
        assert got_b[0].tobytes() == got[0].tobytes() and got_b[1].tobytes() == got[1].tobytes()


def test_host_batch_ingest_reuses_its_pool():
    """Several large host-buffer batches through one handle (the packing workers and the pinned staging belong to the
    handle and are re-used), interleaved with small ones; results must not depend on the batch they travelled in."""
    from multi_robot_slam_separators_amd import lib
    p = synth.camera_params()
    p.iterations = 100
    A, B, is_true, _ = synth.make_pairs(4242, 160, k=500, cols=32, true_frac=0.3)
    with lib.SeparatorFinder(p) as f:
        ref = f.estimate_transform_batch(A, B)          # > 4 MB of staged features: the worker pool starts
        again = f.estimate_transform_batch(A, B)        # ... and is re-used
        small = f.estimate_transform_batch(A[:3], B[:3])   # inline packing on the calling thread
        rev = f.estimate_transform_batch(A[::-1], B[::-1])
    assert ref.tobytes() == again.tobytes()
    assert small.tobytes() == ref[:3].tobytes()
    assert rev[::-1].tobytes() == ref.tobytes()
    assert ref["success"][is_true].all()
