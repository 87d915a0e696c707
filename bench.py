#!/usr/bin/env python3
"""Benchmark of the separator-finder hot path on MI355X (contract: see the round prompt).

One STEP = one full inter-robot matching pass at BASELINE.json configs[1]:
  1. NetVLAD NN search of robot B's N keyframes against robot A's N descriptors (dense
     N x N x 4096 fp32 distance matrix on the matrix cores with the row arg-min fused, float64
     re-evaluation of the row minima, the sequential top-K walk of data_handler.py:191-205);
  2. geometric verification of EVERY candidate the NN stage returns (both registration passes of
     stereoCamGeometricTools.cpp:122-178: Hamming kNN-2 + NNDR, RANSAC 3D-3D with 500
     iterations + refinement, guess-guided re-matching, RANSAC again);
  3. every candidate's success flag and the ACCEPTED separator records go to (pinned) host memory;
     with N > 1 ranks the accepted records are first all-gathered over RCCL (ragged, two-phase).
`value` = candidate pairs verified per second over the whole step, all ranks.  Inputs (both
robots' NetVLAD databases and keyframe feature stores) are resident in HBM before the timed
region.

N > 1 (`python bench.py --gpus N` starts the N ranks itself when WORLD_SIZE is unset; under torchrun the
environment's WORLD_SIZE must equal --gpus):
  --partition robot-pairs (default, "weak"): every rank owns an independent robot pair of the same size and the
      accepted separators are all-gathered (one RCCL collective per step, out of two alternating send buffers: the
      collective of step k runs beside the verification of step k + 1);
  --partition 8e ("strong"): SURVEY.md section 8(e) as written -- ONE robot pair's step cut over the ranks: local NN
      rows in contiguous blocks against the replicated received database, all-gather of the per-row minima, the
      walk replicated, candidate p to rank p mod G over a replicated keyframe store, flags + accepted records
      all-gathered and interleaved back into candidate order (multi_robot_slam_separators_amd/sharded.py; the same
      orchestration is run by gloo ranks in tests/test_sharded_step.py).  --robots R flattens R(R-1)/2 robot pairs
      into one candidate list first (BASELINE configs[4]: 5 robots).
  --workload cfg4: BASELINE configs[3], verification only: --pairs candidate pairs (default 1 000 000) of the
      configs[1] shape round-robin over the ranks, accepted separators all-gathered.
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "candidate keyframe-pair verifications/sec (NetVLAD NN + ORB match + RANSAC) @1/2/4/8 GPU"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak
MFMA_F16_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: bf16/fp16 dense matrix peak
MFMA_FP4_PEAK_TF = 10000.0   # MI355X_MICROARCH.md: fp6/fp4 dense matrix peak (~10 PF; the 20 PF spec is 2:1 sparse)


def bytes_per_pair(k, cols):
    """compulsory HBM traffic of one verification = both keyframes'
    descriptors + 3D points read once + one result written: 2*K*(C+12) + 352."""
    return 2 * k * (cols + 12) + 352


def pmc_traffic(kernel_prefix, pairs_per_launch, any_size=False):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_summary.json; separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).
    gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts half of a coalesced stream
    (calibrated here on k_nn_copy_rows / k_ingest, see profiles/README.md)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")))
    for path in reversed(files):
        try:
            pm = json.load(open(path)).get("pmc_per_launch", {})
        except Exception:
            continue
        for name, v in pm.items():
            if name.startswith(kernel_prefix) and "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
                ppl = v.get("pairs_per_launch", 10000.0)     # summaries before r01k: 10 000 pairs per launch
                if abs(ppl - pairs_per_launch) > 1 and not any_size:
                    return None
                return {"bytes": (2.0 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024.0,
                        "source": os.path.basename(path), "kernel": name,
                        "note": "measured at %d pairs per launch; (2*FETCH_SIZE + WRITE_SIZE) KiB" % ppl}
    return None


def sq_evidence(kernel_prefix):
    """VALU-busy estimate of the dominant kernel from the committed SQ counter passes (profiles/*_sq_*.json)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if (kernel_prefix.replace("k_match_global", "k_match") in os.path.basename(path) and "valu_busy_estimate" in d
                and "SQ_INSTS_MFMA" not in d):      # passes of the VALU matcher only
            return {"frac": d["valu_busy_estimate"], "SQ_INSTS_VALU": d.get("SQ_INSTS_VALU"),
                    "GRBM_GUI_ACTIVE": d.get("GRBM_GUI_ACTIVE"), "source": os.path.basename(path)}
    return None


def sq_counters(kernel_prefix):
    """The newest committed SQ counter pass of a kernel (profiles/*_sq_<kernel>.json), whole."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_*.json")), reverse=True):
        if kernel_prefix.replace("k_match_global", "k_match") in os.path.basename(path):
            try:
                d = json.load(open(path))
            except Exception:
                continue
            d.pop("_note", None)
            d["source"] = os.path.basename(path)
            return d
    return None


def compute_note(dom, k, cols, pairs_per_launch, launch_ms):
    """What actually bounds the dominant kernel: on-chip issue, not HBM (SURVEY section 8(d) asks for both).
    Default build: the K x K Hamming table runs on the fp4 matrix cores (+-1 encoding, exact); with
    SF_MATCH_MFMA=0 on the VALU (xor + popcount)."""
    per_s = pairs_per_launch * k * k / (launch_ms * 1e-3) if launch_ms > 0 else 0.0
    if os.environ.get("SF_MATCH_MFMA", "1") != "0":
        flop = 2.0 * k * k * cols * 8          # one multiply-add per descriptor bit pair
        tf = pairs_per_launch * flop / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0
        scan = ("four resident column tiles per wavefront (three workgroups per CU): per 32-row tile and SIMD the 16 MFMAs hold "
                "the fp4 pipe for 226 ns; the vector issue port is held 1.06 ns by each of the scan's 102 vector instructions and "
                "9.3 ns by each MFMA (tools/ubench/mfma_port.hip): 257 ns -- over the launch the port is ~70 % busy, the pipe "
                "~50 %, the rest is dependent latency at three wavefronts per SIMD (DESIGN.md section 5)" if dom == "k_match_split" else
                "per 32-row tile and SIMD the 8 MFMAs hold the fp4 pipe for 113 ns; the vector issue port is held 1.06 ns by each "
                "of the scan's 60 vector instructions and 9.3 ns by each MFMA (tools/ubench/mfma_port.hip): 138 ns "
                "(DESIGN.md section 5; profiles/r03m_fewer_valu_no_gain.log)")
        return {"note": "matching = v_mfma_f32_32x32x64_f8f6f4 over +-1-encoded descriptor bits (2*K*K*bits flop per pair, "
                        "rows unpadded) + 1.25 VALU ops per table cell for the top-2 scan: " + scan
                        + ("; the launch also holds both motion-estimation chains of the surviving pairs, which are "
                           "latency-bound" if dom == "k_verify_fused" else ""),
                "matrix_core": {"achieved": tf, "peak": MFMA_FP4_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_FP4_PEAK_TF},
                "descriptor_pairs_per_s": per_s, "counters": sq_counters(dom)}
    return {"note": "VALU matcher (SF_MATCH_MFMA=0): per 256-bit descriptor pair 8 v_xor (full rate) + 8 v_bcnt_u32_b32 "
                    "(HALF rate on gfx950, tools/ubench/valu_rate.hip) + 6 16-bit min/max",
            "descriptor_pairs_per_s": per_s, "valu_busy_from_counters": sq_evidence(dom)}


def generate_inputs(seed, n_kf, k, cols, dim, true_frac):
    from multi_robot_slam_separators_amd import synth
    t0 = time.time()
    feats = synth.make_store_batch(seed, n_kf, k=k, cols=cols, true_frac=true_frac)
    rng = np.random.default_rng(seed + 1)
    # robot A's NetVLAD rows (the "received" database on robot B); B's row j is a perceptual alias
    # of A's row j (distance ~0.05 < netvlad_distance), so the NN stage proposes all N pairs and the
    # geometric stage has to sort the 20 % true revisits from the 80 % aliases.
    a = rng.standard_normal((n_kf, dim), dtype=np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + rng.standard_normal((n_kf, dim), dtype=np.float32) * np.float32(0.05 / np.sqrt(dim))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return feats, a, b, time.time() - t0


def measure_next_rows(dev):
    """SURVEY section 8(f) rows 3 and 4 beside the headline (rank 0, N = 1, not timed into `value`): one stereo keyframe
    pixels -> store slot (corner detection, stereo correspondence, descriptors + 3D points) and one NetVLAD inference,
    each with its CPU restatement (oracle/, one thread for the features, torch fp32 on this process' cores for the
    network) timed on the same input."""
    import torch
    from multi_robot_slam_separators_amd import _abi, lib, synth
    from oracle import pyoracle
    from tests import extract_cases as ec
    out = {}
    p = synth.camera_params()
    p.max_features = 1024
    p.store_capacity = 600
    g = lib.SeparatorFinder(p, device=dev.index or 0)
    try:
        g.set_stream(torch.cuda.current_stream().cuda_stream)
        left, right, _ = ec.make_stereo_pair(5, pad=0)
        h, w = left.shape
        cam = _abi.stereo_camera(460.0, 458.0, 367.2, 248.4, 0.11)
        tests = ec.brief_tests(9, 32)
        g.brief_set_pattern(tests)
        dl = torch.from_numpy(np.ascontiguousarray(left)).to(dev)
        dr = torch.from_numpy(np.ascontiguousarray(right)).to(dev)
        d_kp = torch.zeros((1000, 28), dtype=torch.uint8, device=dev)
        d_xy = torch.zeros((1000, 2), dtype=torch.float32, device=dev)
        d_rx = torch.zeros(1000, dtype=torch.float32, device=dev)
        d_st = torch.zeros(1000, dtype=torch.uint8, device=dev)

        def keyframe():
            n = g.detect_corners_device(dl.data_ptr(), w, h, w, 1000, 0.001, 3.0, d_kp.data_ptr(), 1000)
            g.stereo_correspondences_device(dl.data_ptr(), dr.data_ptr(), w, h, w, d_kp.data_ptr(), n, d_xy.data_ptr(),
                                            d_st.data_ptr(), d_rx.data_ptr())
            return g.extract_keyframe_device(dl.data_ptr(), w, h, w, d_kp.data_ptr(), d_rx.data_ptr(), d_st.data_ptr(), n, cam)
        for _ in range(5):
            keyframe()
        torch.cuda.synchronize()
        reps = 100
        t1 = time.perf_counter()
        for _ in range(reps):
            slot, rows = keyframe()
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t1) / reps * 1e3
        t1 = time.perf_counter()
        kp0 = pyoracle.detect_corners(left, 1000, 0.001, 3.0)
        t2 = time.perf_counter()
        xy0, st0, _ = pyoracle.stereo_correspondences(left, right, kp0)
        t3 = time.perf_counter()
        d0, p0, k0 = pyoracle.extract_keyframe(left, kp0, np.ascontiguousarray(xy0[:, 0]), st0, cam, tests)
        t4 = time.perf_counter()
        # a batch of 64 keyframes in one launch sequence (counts on the device, no host wait between the stages)
        nb = 64
        stride = ((h * w + 255) // 256) * 256
        Lb = torch.zeros((nb, stride), dtype=torch.uint8, device=dev)
        Rb = torch.zeros((nb, stride), dtype=torch.uint8, device=dev)
        for i in range(nb):
            li, ri, _ = ec.make_stereo_pair(100 + i % 8, pad=0)
            Lb[i, : h * w] = torch.from_numpy(np.ascontiguousarray(li).reshape(-1)).to(dev)
            Rb[i, : h * w] = torch.from_numpy(np.ascontiguousarray(ri).reshape(-1)).to(dev)
        d_rows = torch.zeros(nb, dtype=torch.int32, device=dev)
        batch_ms = None
        for rep in range(4):
            g.store_clear()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            g.get_features_and_descriptor_batch_device(Lb.data_ptr(), Rb.data_ptr(), nb, w, h, w, stride, cam, None, None,
                                                       d_rows.data_ptr())
            torch.cuda.synchronize()
            if rep:
                batch_ms = min(batch_ms or 1e9, (time.perf_counter() - t5) / nb * 1e3)
        out["keyframe_features"] = {
            "ms_per_keyframe_in_a_batch_of_64": batch_ms, "batch_features_kept_mean": float(d_rows.float().mean().item()),
            "what": "752 x 480 stereo pair -> 1000 corners (goodFeaturesToTrack) -> pyramidal LK -> BRIEF-32 + stereo 3D, "
                    "written into the device-resident store; synchronous calls, pixels resident in HBM",
            "ms_per_keyframe": gpu_ms, "features_kept": int(rows), "rows_equal_cpu_restatement": bool(rows == len(d0)),
            "cpu_baseline": {"kind": "port", "cores": 1, "ms_per_keyframe": (t4 - t1) * 1e3,
                             "detect_ms": (t2 - t1) * 1e3, "stereo_ms": (t3 - t2) * 1e3, "extract_ms": (t4 - t3) * 1e3}}
    except Exception as e:
        print("bench: keyframe-feature measurement failed: %r" % (e,), file=sys.stderr)
    try:
        from oracle import netvlad_torch as nv
        wts = nv.random_weights(3, clusters=64, pca_dim=4096)
        g.netvlad_load(wts)
        H, W = 480, 640
        img = np.random.default_rng(2).uniform(0, 255, size=(H, W, 3)).astype(np.float32)
        d_img = torch.from_numpy(img).to(dev)
        d_out = torch.zeros(128, dtype=torch.float32, device=dev)
        for _ in range(3):
            g.netvlad_infer_device(d_img.data_ptr(), W, H, d_out.data_ptr(), 128)
        torch.cuda.synchronize()
        reps = 20
        t1 = time.perf_counter()
        for _ in range(reps):
            g.netvlad_infer_device(d_img.data_ptr(), W, H, d_out.data_ptr(), 128)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t1) / reps * 1e3
        # a batch of netvlad_batch_size = 3 images (data_handler.py:149-156): one pass over the WPCA weights for the three
        d_img3 = torch.stack([d_img, d_img.flip(0), d_img.flip(1)]).contiguous()
        d_out3 = torch.zeros((3, 128), dtype=torch.float32, device=dev)
        g.netvlad_infer_batch_device(d_img3.data_ptr(), 3, W, H, d_out3.data_ptr(), 128)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            g.netvlad_infer_batch_device(d_img3.data_ptr(), 3, W, H, d_out3.data_ptr(), 128)
        torch.cuda.synchronize()
        gpu_ms3 = (time.perf_counter() - t1) / reps / 3 * 1e3
        batch_same = bool(torch.equal(d_out3[0], d_out))
        t1 = time.perf_counter()
        want = nv.netvlad(img, wts)
        cpu_ms = (time.perf_counter() - t1) * 1e3
        flop = 0.0
        hh, ww = H, W
        for i, (ci, co) in enumerate(_abi.VGG16_CONVS):
            flop += 2.0 * hh * ww * 9 * ci * co
            if nv.POOL[i]:
                hh, ww = hh // 2, ww // 2
        out["netvlad_inference"] = {
            "what": "VGG16 + NetVLAD + WPCA (4096) on a 640 x 480 image, random weights of the published shapes, first "
                    "128 dimensions kept (data_handler.py:157-158)",
            "ms_per_image": gpu_ms, "ms_per_image_in_a_batch_of_3": gpu_ms3, "batch_bits_equal_single_image": batch_same,
            "trunk_gflop": flop / 1e9, "tflops_fp32_equivalent": flop / (gpu_ms * 1e-3) / 1e12,
            "max_abs_error_vs_cpu_fp32": float(np.abs(d_out.cpu().numpy() - want[:128]).max()),
            "cpu_baseline": {"kind": "port", "cores": torch.get_num_threads(), "ms_per_image": cpu_ms,
                             "what": "PyTorch fp32 CPU evaluation of the same network (oracle/netvlad_torch.py)"}}
    except Exception as e:
        print("bench: NetVLAD measurement failed: %r" % (e,), file=sys.stderr)
    g.close()
    return out


def cpu_baseline(params, feats, nv_a, nv_b, n_kf, sample_pairs, sample_rows):
    """The oracle (kind "port") timed on this box's host cores on a bounded sample of the same
    workload; scaled to the full step."""
    from multi_robot_slam_separators_amd import _abi
    from oracle import pyoracle
    threads = pyoracle.num_threads()
    S = min(sample_pairs, n_kf)
    A = [_abi.FeatureArrays(feats["desc_a"][i], feats["xyz_a"][i], feats["kp_a"][i]) for i in range(S)]
    B = [_abi.FeatureArrays(feats["desc_b"][i], feats["xyz_b"][i], feats["kp_b"][i]) for i in range(S)]
    pyoracle.estimate_transform_batch(params, A[:8], B[:8], threads)      # warm-up
    t0 = time.time()
    res = pyoracle.estimate_transform_batch(params, A, B, threads)
    t_ver = time.time() - t0
    R = min(sample_rows, n_kf)
    loc = nv_b[:R].astype(np.float64)
    rec = nv_a.astype(np.float64)
    t0 = time.time()
    pyoracle.find_matches(loc, rec, netvlad_distance=params.netvlad_distance, max_matches_nb=R)
    t_nn = time.time() - t0
    t_full = t_ver * (n_kf / S) + t_nn * (n_kf / R)
    # single-thread figure (the reference's nodes are single-threaded, stereoCamGeometricTools.cpp:212)
    S1 = min(96, S)
    t0 = time.time()
    pyoracle.estimate_transform_batch(params, A[:S1], B[:S1], 1)
    t_one = time.time() - t0
    return {
        "value": n_kf / t_full, "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": "%d of %d candidate pairs verified in %.2f s + NN rows %d of %d x %d x %d in %.2f s, "
                  "both scaled to the full step; OpenMP over pairs / rows, one thread per CPU of this process' "
                  "cgroup share" % (
                      S, n_kf, t_ver, R, n_kf, n_kf, nv_a.shape[1], t_nn),
        "verify_pairs_per_s": S / t_ver, "accepted_in_sample": int(res["success"].sum()),
        "verify_pairs_per_s_single_thread": S1 / t_one,
        "_results": res,          # (popped by the caller: the oracle's results of pairs 0 .. S-1, for the parity count)
    }


def upload_store(f, feats, dev, n_kf, k, cols):
    """Both robots' keyframes of one robot pair into the handle's device store; returns (slot_a, slot_b)."""
    import torch

    def up(x):
        x = np.ascontiguousarray(x)
        if x.dtype.fields:
            x = x.view(np.uint8)
        return torch.from_numpy(x).to(dev)
    CH = 2048
    first = {}
    for which in ("a", "b"):
        fs0 = None
        for s in range(0, n_kf, CH):
            e = min(n_kf, s + CH)
            td_, tx, tk = up(feats["desc_" + which][s:e]), up(feats["xyz_" + which][s:e]), up(feats["kp_" + which][s:e])
            fs = f.store_add_keyframes_device(e - s, k, cols, td_.data_ptr(), tx.data_ptr(), tk.data_ptr())
            torch.cuda.synchronize()
            fs0 = fs if fs0 is None else fs0
        first[which] = fs0
    return first["a"], first["b"]


def set_estimator(p, args):
    """--estimator / --bundle-adjustment / --forward-est-only -> sf_params (myRegistrationVis.cpp:52-71 reads these as
    rtabmap's Vis/EstimationType, Vis/BundleAdjustment, Vis/ForwardEstOnly).  The adjustment takes stereo residuals
    (baseline 0.12 m: the reference's camera is a stereo pair, stereoCamGeometricTools.cpp:56-76)."""
    p.estimation_type = 1 if args.estimator == "pnp" else 0
    if args.bundle_adjustment:
        p.bundle_adjustment = 1
        p.stereo_baseline = 0.12
    p.forward_est_only = int(args.forward_est_only)


def estimator_text(args):
    return ("3D-3D" if args.estimator == "3d3d" else "PnP") + (" + two-view bundle adjustment" if args.bundle_adjustment else "") + \
        (", both directions (Vis/ForwardEstOnly = false)" if not args.forward_est_only else "")


def run_partition_8e(args, rank, world, dev, dev_index, coll_dev, dist_on):
    """the step(s) of R(R-1)/2 robot pairs cut over the ranks (strong scaling).  Prints its own
    JSON line (rank 0)."""
    import torch
    import torch.distributed as td
    from multi_robot_slam_separators_amd import _abi, dist, lib, sharded, synth
    n_kf, k, cols, dim = args.keyframes, args.features, args.desc_bytes, args.dim
    n_rp = args.robots * (args.robots - 1) // 2
    p = synth.camera_params()
    p.iterations = args.iterations
    set_estimator(p, args)
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.nn_precision = args.nn_precision
    p.max_features = k
    p.desc_bytes = cols
    p.store_capacity = 2 * n_kf
    lo, hi = sharded.row_blocks(n_kf, world)[rank]
    steps, truths = [], []
    for rp in range(n_rp):
        feats, nv_a, nv_b, _ = generate_inputs(12345 + rp, n_kf, k, cols, dim, args.true_frac)   # SAME on every rank
        f = lib.SeparatorFinder(p, device=dev_index)
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        slot_a, slot_b = upload_store(f, feats, dev, n_kf, k, cols)        # replicated store
        ta = torch.from_numpy(nv_a).to(dev)
        tb = torch.from_numpy(np.ascontiguousarray(nv_b[lo:hi])).to(dev)
        if args.netvlad_f16:
            ta, tb = ta.to(torch.float16), tb.to(torch.float16)
            f.nn_append_received_f16_device(ta.data_ptr(), n_kf, dim)
            if hi > lo:
                f.nn_append_local_f16_device(tb.data_ptr(), hi - lo, dim)
        else:
            f.nn_append_received_device(ta.data_ptr(), n_kf, dim)
            if hi > lo:
                f.nn_append_local_device(tb.data_ptr(), hi - lo, dim)
        torch.cuda.synchronize()
        be = sharded.GpuShardBackend(f, lo, n_kf, slot_a, slot_b, n_kf, dev, world)
        steps.append(sharded.ShardedStep(be, rank, world, n_kf, coll_dev, accept_cap=n_kf // (4 * world) + 256))
        truths.append(feats["is_true"])
        del feats

    def step():
        out = [st.step() for st in steps]          # one robot pair after the other: each is cut over all the ranks
        return out

    for _ in range(args.warmup):
        step()
    if dist_on:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_pairs = 0
    for _ in range(args.steps):
        last = step()
        n_pairs += sum(len(m) for m, _, _ in last)
    torch.cuda.synchronize()
    if dist_on:
        td.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    if dist_on:
        td.all_reduce(t, op=td.ReduceOp.MAX)
    elapsed = float(t.item())
    # the node's output = the single-GPU output: every candidate's decision equals the ground truth on every rank
    correct = total = accepted = 0
    for (m, flags, acc), truth in zip(last, truths):
        want = truth[m["idx_local"]] & (m["idx_local"] == m["idx_other"])
        correct += int((flags == want).sum())
        total += len(m)
        accepted += int(acc.shape[0])
        rec = np.frombuffer(acc.tobytes(), dtype=_abi.RESULT_DTYPE)
        assert bool(rec["success"].all()) and len(rec) == int(flags.sum())
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": n_pairs / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "u8+f32+f64" if args.nn_precision == 0 else "u8+f16+f32+f64", "data": "synthetic",
            "config": {"workload": "SURVEY 8(e) partition: %d robots = %d robot pair(s) x %d keyframes, %d-D %s NetVLAD, "
                                   "%d x %d-bit ORB, <= %d RANSAC hypotheses (PCL adaptive stop, p = 0.99); local NN rows in "
                                   "%d contiguous blocks + all-gather of row minima + replicated walk; candidate p -> rank "
                                   "p mod %d over a replicated store; flags + accepted records all-gathered and interleaved"
                                   % (args.robots, n_rp, n_kf, dim, "fp16" if args.netvlad_f16 else "fp32", k, cols * 8,
                                      args.iterations, world, world),
                       "pairs_per_step": n_pairs / args.steps, "parallelism": "section 8(e): row-sharded NN, pairs p mod G"},
            "check": {"decisions_matching_ground_truth": correct, "of": total, "accepted_last_step": accepted,
                      "host_waits_per_robot_pair_step": steps[0].waits},
        }))
    for st in steps:
        st.b.f.close()


def run_cfg4(args, rank, world, dev, dev_index, coll_dev, dist_on):
    """BASELINE configs[3]: --pairs candidate pairs of the configs[1] shape round-robin over the ranks, verification
    only; the accepted separators are all-gathered (dist.RecordExchange).  The pairs re-use a replicated store of
    --keyframes keyframes per robot (pair p = keyframe p mod N of robot A against the same of robot B)."""
    import torch
    import torch.distributed as td
    from multi_robot_slam_separators_amd import _abi, dist, lib, synth
    n_kf, k, cols = args.keyframes, args.features, args.desc_bytes
    p = synth.camera_params()
    p.iterations = args.iterations
    set_estimator(p, args)
    p.max_features = k
    p.desc_bytes = cols
    p.store_capacity = 2 * n_kf
    feats = synth.make_store_batch(12345, n_kf, k=k, cols=cols, true_frac=args.true_frac)     # replicated
    f = lib.SeparatorFinder(p, device=dev_index)
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    slot_a, slot_b = upload_store(f, feats, dev, n_kf, k, cols)
    mine = dist.shard_pairs(args.pairs, rank, world)                  # p mod G
    d_from = torch.from_numpy((slot_a + mine % n_kf).astype(np.int32)).to(dev)
    d_to = torch.from_numpy((slot_b + mine % n_kf).astype(np.int32)).to(dev)
    n = len(mine)
    RB = _abi.RESULT_DTYPE.itemsize
    d_res = torch.empty((n, RB), dtype=torch.uint8, device=dev)
    exch = dist.RecordExchange(RB, n, n // 4 + 1024, coll_dev) if dist_on else None
    d_acc = torch.empty((n, RB), dtype=torch.uint8, device=dev)
    d_flags = torch.empty(n, dtype=torch.bool, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    h_cnt = torch.zeros(1, dtype=torch.int32).pin_memory()

    def step():
        f.verify_pairs_device(d_from.data_ptr(), d_to.data_ptr(), n, d_res.data_ptr())
        if exch is not None and coll_dev.type == "cuda":
            f.compact_accepted_device_async(d_res.data_ptr(), n, exch.payload.data_ptr(), d_flags.data_ptr(), exch.count_ptr)
            exch.exchange(None)
            h_cnt.copy_(exch.send[0, :4].view(torch.int32), non_blocking=True)
        elif exch is not None:
            n_acc = f.compact_accepted_device(d_res.data_ptr(), n, d_acc.data_ptr(), d_flags.data_ptr())
            exch.payload[:n_acc].copy_(d_acc[:n_acc])
            exch.exchange(n_acc)
            h_cnt[0] = n_acc
        else:
            f.compact_accepted_device_async(d_res.data_ptr(), n, d_acc.data_ptr(), d_flags.data_ptr(), d_cnt.data_ptr())
            h_cnt.copy_(d_cnt, non_blocking=True)
        torch.cuda.synchronize()
        return int(h_cnt[0])

    for _ in range(args.warmup):
        step()
    f.prof_reset()
    f.prof_enable(True)
    if dist_on:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_acc = step()
    torch.cuda.synchronize()
    if dist_on:
        td.barrier()
    elapsed = time.perf_counter() - t0
    prof = f.prof_get()
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    if dist_on:
        td.all_reduce(t, op=td.ReduceOp.MAX)
    elapsed = float(t.item())
    truth = feats["is_true"][mine % n_kf]
    ok = bool((d_flags.cpu().numpy() == truth).all())
    gathered = sum(exch.counts()) if exch is not None else n_acc
    if rank == 0:
        nm, tm = prof.get("k_verify_fused", (0, 0.0))
        bpp = bytes_per_pair(k, cols)
        launch_ms = tm / max(nm, 1)
        ppl = n * args.steps / max(nm, 1)
        ach = ppl * bpp / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        print(json.dumps({
            "metric": METRIC, "value": args.pairs * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8+f32+f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d candidate pairs (%d x %d-bit ORB per keyframe, <= %d RANSAC "
                                   "hypotheses with PCL's adaptive stop, both passes, %.0f %% true) round-robin over %d rank(s), "
                                   "verification only, replicated store of 2 x %d keyframes, accepted separators "
                                   "all-gathered" % (args.pairs, k, cols * 8, args.iterations, 100 * args.true_frac, world, n_kf),
                       "pairs_per_step_per_gpu": n, "parallelism": "pairs p mod G"},
            "roofline": {"kernel": "k_verify_fused", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None, "bytes_per_pair": bpp, "pairs_per_launch": ppl,
                         "avg_launch_ms": launch_ms},
            "check": {"decisions_matching_ground_truth_rank0": ok, "accepted_rank0": n_acc,
                      "accepted_separators_gathered_per_step": gathered},
        }))
    f.close()


def run_cfg3(args, rank, world, dev, dev_index, coll_dev, dist_on):
    """BASELINE configs[2]: 3 robots = 3 robot pairs, 100 000 keyframes per robot, 4096-D NetVLAD, 1000 x 256-bit ORB per
    keyframe, <= 2000 RANSAC hypotheses per pass.  One step = one find-and-verify pass (sf_step_issue / sf_step_retire)
    of EVERY robot pair: 100 000 x 100 000 x 4096 NN + 100 000 candidate verifications each (every local row has a
    perceptual alias under netvlad_distance, a fifth of them are true revisits).  Each robot pair lives in its own handle
    (its two NetVLAD databases: 3.3 GB fp32, its replicated keyframe store: 12 GB).  N > 1: one such topology per rank
    (weak scaling, no data-path collective).  The keyframes of a robot are `--cfg3-base` generated frames tiled on the
    device (every slot its own HBM bytes; generating 300 000 x 1000 features on the host would take minutes)."""
    import torch
    import torch.distributed as td
    from multi_robot_slam_separators_amd import _abi, lib, synth
    n_kf, k, cols, dim, base = args.keyframes, args.features, args.desc_bytes, args.dim, args.cfg3_base
    n_rp = args.robots * (args.robots - 1) // 2
    p = synth.camera_params()
    p.iterations = args.iterations
    set_estimator(p, args)
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.nn_precision = args.nn_precision
    p.max_features = k
    p.desc_bytes = cols
    p.store_capacity = 2 * n_kf
    t_gen0 = time.time()
    reps = (n_kf + base - 1) // base

    def up(x):
        x = np.ascontiguousarray(x)
        return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)

    handles, truths, feats0, nv0 = [], [], None, None
    for rp in range(n_rp):
        feats = synth.make_store_batch(12345 + 97 * rank + rp, base, k=k, cols=cols, true_frac=args.true_frac)
        f = lib.SeparatorFinder(p, device=dev_index)
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.set_option(_abi.SF_OPT_STEP_DEPTH, 2)          # (a step is tens of milliseconds: two in flight per robot pair)
        slots = []
        for which in ("a", "b"):
            d0, x0, k0 = up(feats["desc_" + which]), up(feats["xyz_" + which]), up(feats["kp_" + which]).view(base, -1)
            CH = 8 * base
            first = None
            for s0 in range(0, n_kf, CH):                # tile the base frames over the robot's keyframes, CH slots at a time
                m = min(CH, n_kf - s0)
                r_ = (m + base - 1) // base
                dd = d0.repeat(r_, 1, 1)[:m].contiguous(); xx = x0.repeat(r_, 1, 1)[:m].contiguous(); kk = k0.repeat(r_, 1)[:m].contiguous()
                fs = f.store_add_keyframes_device(m, k, cols, dd.data_ptr(), xx.data_ptr(), kk.data_ptr())
                torch.cuda.synchronize()
                first = fs if first is None else first
                del dd, xx, kk
            slots.append(first)
        g = torch.Generator(device=dev)
        g.manual_seed(777 + 31 * rank + rp)
        a = torch.randn((n_kf, dim), generator=g, device=dev, dtype=torch.float32)
        a /= a.norm(dim=1, keepdim=True)
        b = a + torch.randn((n_kf, dim), generator=g, device=dev, dtype=torch.float32) * (0.05 / np.sqrt(dim))
        b /= b.norm(dim=1, keepdim=True)
        f.nn_append_received_device(a.data_ptr(), n_kf, dim)     # robot A's descriptors, as received by B
        f.nn_append_local_device(b.data_ptr(), n_kf, dim)        # robot B's own
        torch.cuda.synchronize()
        if rp == 0:
            feats0 = feats
            nv0 = (a[:, :].cpu().numpy(), b[: args.cpu_sample_rows].cpu().numpy())
        del a, b
        handles.append((f, slots[0], slots[1]))
        truths.append(np.tile(feats["is_true"], reps)[:n_kf])
    t_gen = time.time() - t_gen0
    inflight = [0] * n_rp
    state = {"pairs": 0, "last": [None] * n_rp}

    def retire(rp, copy=False):
        m, rom, recs, info = handles[rp][0].step_retire(copy=copy)
        inflight[rp] -= 1
        state["pairs"] += info["n_matches"]
        state["last"][rp] = (m, rom, info)

    def step():
        for rp, (f, sa, sb) in enumerate(handles):
            if inflight[rp] >= 2:
                retire(rp)
            f.step_issue(sa, sb)
            inflight[rp] += 1

    def drain(copy=False):
        for rp in range(n_rp):
            while inflight[rp]:
                retire(rp, copy)

    for _ in range(args.warmup):
        step()
    drain()
    f0 = handles[0][0]
    f0.prof_reset(); f0.prof_select(("k_verify_fused", "k_match_global", "k_nn_filter_f16")); f0.prof_enable(True)
    state["pairs"] = 0
    if dist_on:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain(copy=True)
    torch.cuda.synchronize()
    if dist_on:
        td.barrier()
    elapsed = time.perf_counter() - t0
    prof = f0.prof_get()
    f0.prof_enable(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    npairs = torch.tensor([state["pairs"]], dtype=torch.float64, device=coll_dev)
    if dist_on:
        td.all_reduce(t, op=td.ReduceOp.MAX)
        td.all_reduce(npairs, op=td.ReduceOp.SUM)
    elapsed, total_pairs = float(t.item()), float(npairs.item())
    correct = total = accepted = 0
    for rp in range(n_rp):
        m, rom, info = state["last"][rp]
        want = truths[rp][m["idx_local"]] & (m["idx_local"] == m["idx_other"])
        correct += int(((rom >= 0) == want).sum())
        total += len(m)
        accepted += info["n_accepted"]
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # the oracle on a bounded sample: verification of the first pairs of robot pair 0 + the NN rows of a few queries
        from oracle import pyoracle
        threads = pyoracle.num_threads()
        S = min(args.cpu_sample_pairs, base, 1500)
        A = [_abi.FeatureArrays(feats0["desc_a"][i], feats0["xyz_a"][i], feats0["kp_a"][i]) for i in range(S)]
        B = [_abi.FeatureArrays(feats0["desc_b"][i], feats0["xyz_b"][i], feats0["kp_b"][i]) for i in range(S)]
        pyoracle.estimate_transform_batch(p, A[:8], B[:8], threads)
        tv = time.time(); res = pyoracle.estimate_transform_batch(p, A, B, threads); tv = time.time() - tv
        R = min(args.cpu_sample_rows, 128)
        tn = time.time()
        pyoracle.find_matches(nv0[1][:R].astype(np.float64), nv0[0].astype(np.float64), netvlad_distance=p.netvlad_distance,
                              max_matches_nb=R)
        tn = time.time() - tn
        t_full = tv * (n_kf / S) + tn * (n_kf / R)
        cpu = {"value": n_kf / t_full, "unit": "pairs/s", "cores": threads, "kind": "port",
               "sample": "%d of %d candidate pairs of one robot pair verified in %.2f s + NN rows %d of %d x %d x %d in %.2f s, both "
                         "scaled to that robot pair's step; OpenMP over pairs / rows" % (S, n_kf, tv, R, n_kf, n_kf, dim, tn),
               "verify_pairs_per_s": S / tv, "accepted_in_sample": int(res["success"].sum())}
    if rank == 0:
        nm, tm = prof.get("k_verify_fused", (0, 0.0))
        nf, tf_ = prof.get("k_nn_filter_f16", (0, 0.0))
        bpp = bytes_per_pair(k, cols)
        launch_ms = tm / max(nm, 1)
        ach = n_kf * bpp / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        flop4 = 2.0 * k * k * cols * 8 * n_kf
        filt_ms = tf_ / max(nf, 1)
        kd = f0.nn_last_filter_dims()
        print(json.dumps({
            "metric": METRIC, "value": total_pairs / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8+f16+f32+f64" if args.nn_precision == 1 else "u8+f32+f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 1xMI355X per rank, %d robots = %d robot pairs x %d keyframes per robot, "
                                   "%d-D fp32 NetVLAD, %d x %d-bit ORB per keyframe, <= %d RANSAC hypotheses per pass with PCL's "
                                   "adaptive stop (%s), both registration passes, %.0f %% true revisits, every local row has an "
                                   "alias under netvlad_distance (all rows become candidates); keyframes = %d generated frames "
                                   "per robot tiled on the device; NN filter contracted %d of %d dimensions"
                                   % (args.robots, n_rp, n_kf, dim, k, cols * 8, args.iterations, args.estimator,
                                      100 * args.true_frac, base, kd, dim),
                       "pairs_per_step_per_gpu": total_pairs / args.steps / world, "parallelism": "single GPU" if world == 1 else
                       "one topology per rank"},
            "roofline": {"kernel": "k_verify_fused (WIDE form: K = 1000)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "bytes_per_pair": bpp,
                         "pairs_per_launch": float(n_kf), "avg_launch_ms": launch_ms,
                         "compute": {"matrix_core": {"achieved": flop4 / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0,
                                                      "peak": MFMA_FP4_PEAK_TF, "unit": "TFLOP/s",
                                                      "frac": flop4 / (launch_ms * 1e-3) / 1e12 / MFMA_FP4_PEAK_TF if launch_ms > 0 else 0.0},
                                     "note": "K x K Hamming table on the fp4 matrix cores (2*K*K*bits flop per pair) + the "
                                             "motion-estimation chains of the surviving fifth inside the same launch"}},
            "roofline_nn": {"kernel": "k_nn_filter_f16", "bound": "mfma", "avg_launch_ms": filt_ms, "contracted_dims": kd,
                            "achieved": 2.0 * n_kf * n_kf * kd / (filt_ms * 1e-3) / 1e12 if filt_ms > 0 else 0.0,
                            "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                            "frac": 2.0 * n_kf * n_kf * kd / (filt_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF if filt_ms > 0 else 0.0},
            "cpu_baseline": cpu,
            "check": {"decisions_matching_ground_truth": correct, "of": total, "accepted_last_step": accepted},
            "input_generation_s": t_gen,
        }))
    for f, _, _ in handles:
        f.close()


def wait_ranks(procs):
    """Exit code of a set of rank processes.  A rank that dies (e.g. fewer GPUs than ranks) leaves the others waiting in
    the rendezvous or in a collective: as soon as one exits with an error the rest are ended (these exact children, by
    their handles) and its code is returned."""
    import time as _time
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:
                    q.terminate()
        if live:
            _time.sleep(0.05 if rc == 0 else 0.5)
            if rc != 0:
                for q in live:
                    if q.poll() is None:
                        q.kill()
    return rc


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher: one child process per GPU with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), rank 0's stdout (the JSON line) passed through.  The parent has not
    imported torch or touched HIP."""
    import socket
    import subprocess
    if (os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", "")
            or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD")):
        # under rocprofv3 the preloaded tool has initialised the GPU in THIS process already: starting the ranks from
        # here would be the fork + exec of GPU work from a GPU-initialised parent that this pool forbids
        raise SystemExit("bench.py --gpus %d: running under a rocprofiler preload -- profile ONE rank per rocprofv3 "
                         "invocation (the python program itself behind `--`, RANK / WORLD_SIZE set by hand)" % n)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    raise SystemExit(wait_ranks(procs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--keyframes", type=int, default=None, help="keyframes per robot (configs[1]: 10k; --workload cfg3: 100k)")
    ap.add_argument("--features", type=int, default=None, help="features per keyframe (500; --workload cfg3: 1000)")
    ap.add_argument("--desc-bytes", type=int, default=32)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=None, help="RANSAC hypotheses per pass at most (500; --workload cfg3: 2000)")
    ap.add_argument("--cfg3-base", type=int, default=4000, help="--workload cfg3: generated keyframes per robot (tiled on the device)")
    ap.add_argument("--true-frac", type=float, default=0.2)
    ap.add_argument("--nn-precision", type=int, default=1,
                    help="1 = fp16 MFMA filter with rigorous error band + exact f64 refinement (identical "
                         "matches); 0 = fp32 MFMA ranking of every column")
    ap.add_argument("--estimator", choices=("3d3d", "pnp"), default="3d3d",
                    help="motion estimator of both registration passes: 3d3d = RANSAC 3D->3D (north_star, "
                         "myRegistrationVis.cpp:1113-1152), pnp = RANSAC 3D->2D (:1055-1112, rtabmap's default)")
    ap.add_argument("--bundle-adjustment", action="store_true",
                    help="two-view bundle adjustment behind each pass's estimate (myRegistrationVis.cpp:1192-1370; rtabmap's "
                         "Vis/BundleAdjustment = 1, its default where g2o is present); with --estimator pnp this is the flow "
                         "the reference most likely runs as shipped (SURVEY.md section 9)")
    ap.add_argument("--forward-est-only", type=int, choices=(0, 1), default=1,
                    help="0: Vis/ForwardEstOnly = false (both directions estimated and merged, myRegistrationVis.cpp:936-978)")
    ap.add_argument("--strict", action="store_true",
                    help="time BASELINE configs[1] read to the letter: every pass evaluates all iterations + 1 hypotheses "
                         "(ransac_adaptive_stop = 0) and the NN filter contracts the full descriptor length "
                         "(SF_OPT_NN_FULL_FILTER) -- the configuration value_strict of the default line measures; its "
                         "dominant kernel is the fp16 MFMA filter and the roofline of the line is that kernel's")
    ap.add_argument("--netvlad-f16", action="store_true",
                    help="NetVLAD descriptors handed over in fp16 (BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined-extra", action="store_true",
                    help="skip the informational two-stream pipelined measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps: none of the untimed comparison runs (fp32 NN ranking, VALU matcher, "
                         "pipelined) -- what the profiling rounds use, so that every launch they see is a timed one")
    ap.add_argument("--cpu-sample-pairs", type=int, default=10000)
    ap.add_argument("--cpu-sample-rows", type=int, default=1024)
    ap.add_argument("--partition", choices=("robot-pairs", "8e"), default="robot-pairs",
                    help="N > 1: robot-pairs = one independent robot pair per rank (weak scaling); 8e = one robot "
                         "pair's step cut over the ranks as SURVEY.md section 8(e) writes it (strong scaling)")
    ap.add_argument("--robots", type=int, default=None,
                    help="--partition 8e: robots in the topology; their R(R-1)/2 robot pairs are flattened into one "
                         "candidate list before the round-robin (BASELINE configs[4]: 5)")
    ap.add_argument("--workload", choices=("cfg2", "cfg3", "cfg4"), default="cfg2",
                    help="cfg2 = NN + verification step (configs[1], the metric's configuration); cfg3 = configs[2]: 3 robot "
                         "pairs x 100k keyframes, K = 1000, 2000 iterations; cfg4 = configs[3]: --pairs candidate pairs "
                         "round-robin over the ranks, verification only")
    ap.add_argument("--pairs", type=int, default=1000000, help="--workload cfg4: candidate pairs per step (all ranks)")
    args = ap.parse_args()
    big = args.workload == "cfg3"
    args.keyframes = args.keyframes if args.keyframes is not None else (100000 if big else 10000)
    args.features = args.features if args.features is not None else (1000 if big else 500)
    args.iterations = args.iterations if args.iterations is not None else (2000 if big else 500)
    args.robots = args.robots if args.robots is not None else (3 if big else 2)

    # ---- N > 1 without a launcher: start the N ranks here, BEFORE anything touches the GPU (a process that has
    # initialised HIP must never exec or fork GPU work; this parent only waits for its children) ----------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %s: start it as `python -m "
                         "torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d "
                         "...` or let `python bench.py --gpus %d` start the ranks itself"
                         % (args.gpus, os.environ["WORLD_SIZE"], args.gpus, args.gpus, args.gpus))

    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with RCCL's streams in the process
    # the library's second stream would share a queue with its first and lose the overlap it exists for
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not args.no_extras:
        # every step of a multi-rank run is a collective: the informational comparison runs behind the timed region
        # (other NN precision, fixed iterations, VALU matcher, ...) would add hundreds of them, each a chance for the
        # ranks to part ways on an error path; the N > 1 line carries the contract's fields only
        args.no_extras = True
    import torch
    import torch.distributed as td
    from multi_robot_slam_separators_amd.hostinfo import cpu_share
    # torch sizes its intra-op pool from the host's CPU count; on a box that grants a cgroup share of the
    # host that oversubscribes the quota and CFS throttling stalls the process (periodic 40-80 ms gaps)
    torch.set_num_threads(max(1, min(8, cpu_share())))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the separator-finder path has no CPU fallback")
    backend = os.environ.get("BENCH_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on one GPU
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # BENCH_FORCE_DIST=1: run the N > 1 code path (process group, separator exchange) with whatever world size
    # the launcher gave, including 1 -- the rehearsal of the RCCL path on a one-GPU box
    dist_on = world > 1 or os.environ.get("BENCH_FORCE_DIST") is not None
    if dist_on:
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")      # (RCCL's stream: see `xstream` below)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            td.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from multi_robot_slam_separators_amd import _abi, dist, lib, synth

    if args.workload in ("cfg3", "cfg4") or args.partition == "8e":
        (run_cfg3 if args.workload == "cfg3" else run_cfg4 if args.workload == "cfg4" else run_partition_8e)(
            args, rank, world, dev, dev_index, coll_dev, dist_on)
        if dist_on:
            td.destroy_process_group()
        return

    n_kf, k, cols, dim = args.keyframes, args.features, args.desc_bytes, args.dim
    p = synth.camera_params()
    p.iterations = args.iterations
    set_estimator(p, args)
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf            # batch operation: walk every row
    p.nn_precision = args.nn_precision
    p.max_features = k
    p.desc_bytes = cols
    p.store_capacity = 2 * n_kf
    if args.strict:
        p.ransac_adaptive_stop = 0
    feats, nv_a, nv_b, t_gen = generate_inputs(12345 + rank, n_kf, k, cols, dim, args.true_frac)

    f = lib.SeparatorFinder(p, device=dev_index)
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.strict:
        f.set_option(_abi.SF_OPT_NN_FULL_FILTER, 1)
    # ---- make everything resident in HBM (untimed) -------------------------------------------------
    def up(x):
        x = np.ascontiguousarray(x)
        if x.dtype.fields:
            x = x.view(np.uint8)
        return torch.from_numpy(x).to(dev)
    CH = 2048
    slot_a = slot_b = None
    for which in ("a", "b"):
        first = None
        for s in range(0, n_kf, CH):
            e = min(n_kf, s + CH)
            td_, tx, tk = up(feats["desc_" + which][s:e]), up(feats["xyz_" + which][s:e]), up(feats["kp_" + which][s:e])
            fs = f.store_add_keyframes_device(e - s, k, cols, td_.data_ptr(), tx.data_ptr(), tk.data_ptr())
            torch.cuda.synchronize()
            first = fs if first is None else first
        if which == "a":
            slot_a = first
        else:
            slot_b = first
    ta, tb = up(nv_a), up(nv_b)
    if args.netvlad_f16:      # BASELINE configs[4]: NetVLAD shipped in fp16 (exactly representable in the fp32 database)
        ta, tb = ta.to(torch.float16), tb.to(torch.float16)
        nn_append_received, nn_append_local = "nn_append_received_f16_device", "nn_append_local_f16_device"
    else:
        nn_append_received, nn_append_local = "nn_append_received_device", "nn_append_local_device"
    getattr(f, nn_append_received)(ta.data_ptr(), n_kf, dim)    # robot A's descriptors, as received by B
    getattr(f, nn_append_local)(tb.data_ptr(), n_kf, dim)       # robot B's own descriptors
    torch.cuda.synchronize()

    d_res = torch.empty((n_kf, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    # count | per-candidate flags | accepted records packed in ONE device buffer (and its pinned mirror), so that the
    # count, the flags and a speculative prefix of the records reach the host with ONE copy
    RB = _abi.RESULT_DTYPE.itemsize
    flags_off, acc_off = 64, 64 + (n_kf + 63) // 64 * 64
    d_pack = torch.zeros(acc_off + n_kf * RB, dtype=torch.uint8, device=dev)
    h_pack = torch.zeros(acc_off + n_kf * RB, dtype=torch.uint8).pin_memory()
    d_cnt = d_pack[:4].view(torch.int32)
    d_flags = d_pack[flags_off: flags_off + n_kf].view(torch.bool)
    d_acc = d_pack[acc_off:].view(n_kf, RB)
    h_res = h_pack[acc_off:].view(n_kf, RB)
    # N > 1: the accepted separators of every rank are all-gathered on the devices (persistent buffers, one
    # collective per step, capacity for a 25 % acceptance rate + slack; overflow falls back to two-phase);
    # each rank hands ITS OWN accepted separators to the host, so the node delivers every record once.
    exch = None
    if dist_on:
        # (rows for every slot a speculative verification may accept: a smaller mirror switches the step to the compaction)
        exch = dist.RecordExchange(_abi.RESULT_DTYPE.itemsize, n_kf + n_kf // 8 + 256, n_kf // 4 + 256, coll_dev)
    h_flags = h_pack[flags_off: flags_off + n_kf].view(torch.bool)
    h_cnt = h_pack[:4].view(torch.int32)
    spec_cap = n_kf // 4 + 256
    OFF_SUCCESS = _abi.RESULT_DTYPE.fields["success"][1]
    state = {"pairs": 0, "accepted": 0, "last": None}

    # single GPU: the one-synchronisation form of a step (`step()` below: warm-up and comparison runs) lets the compaction
    # kernel write count, flags and accepted records STRAIGHT into a pinned host block
    zero_copy = exch is None
    hp = h_pack.data_ptr()
    h_cnt_np = h_cnt.numpy()                                  # (a view of the pinned word: no tensor indexing per step)
    # THE TIMED STEP is the library's begin / retire pair (include/sepfinder.h: sf_step_issue / sf_step_retire, the loop
    # body of find_separators.py:59-133): sf_step_issue queues the NN filter, the exact re-evaluation, the row minima,
    # the argsort + walk of find_matches (on the device) and the verification, with the accepted separators streaming
    # from inside the verification kernel into a pinned block of the handle, and returns without waiting;
    # sf_step_retire hands out the oldest step.  `depth` steps are kept in flight; all K steps are issued and retired
    # inside the timed region.  examples/bench_cli.cpp runs the same loop from C++ (no torch).
    # N > 1 (RCCL): a step's accepted separators are all-gathered when the step is RETIRED -- every accepted record also
    # lands in a device buffer of the step's own block (sf_step_result.d_records), from which `retire` copies them into the
    # send buffer of one of two alternating exchanges (device-to-device, on a stream of the exchanges' own) and starts
    # the collective, which runs on RCCL's stream beside the steps in flight; only the REUSE of an exchange's buffers, two
    # retires later, waits for it.  The steps themselves run exactly as at N = 1 (same ring, same streams): no buffer of a
    # collective is tied to a step in flight.  (Rounds 2-3 mirrored the records into the send buffer from inside the
    # verification kernel -- sf_step_mirror_pair, still in the library and its tests -- which chained step k + 2 behind
    # the collective of step k.)  With gloo (CPU collectives: the rehearsal of the N > 1 path on one GPU) the steps are
    # not pipelined: `step()`, one synchronisation per step.
    dist_cuda = exch is not None and coll_dev.type == "cuda"
    pipelined = exch is None or dist_cuda
    exchs = [exch]
    retired = [0]
    # the exchanges' own stream, at the highest priority like RCCL's (TORCH_NCCL_HIGH_PRIORITY, set in main()): their
    # launches are a handful of small copies per step that must not wait behind a verification's dispatch (with both at
    # the default priority a step took 0.488 ms at world size 1, with both raised 0.462 -- profiles/r04v_placement, run p; N = 1: 0.446)
    xstream = torch.cuda.Stream(priority=-1) if dist_on and coll_dev.type == "cuda" else None
    if pipelined and dist_cuda and os.environ.get("BENCH_ONE_EXCHANGE_BUFFER") is None:
        exchs.append(dist.RecordExchange(_abi.RESULT_DTYPE.itemsize, n_kf + n_kf // 8 + 256, n_kf // 4 + 256, coll_dev))
    inflight = [0]

    # steps kept in flight: the library's ring (SF_OPT_STEP_DEPTH, default 6)
    depth = int(os.environ.get("SF_STEP_DEPTH", "6"))
    t_issue, t_retire = [], []

    def retire(copy=False):
        ts = time.perf_counter()
        m, rom, recs, info = f.step_retire(copy=copy)
        t_retire.append((time.perf_counter() - ts) * 1e3)
        inflight[0] -= 1
        state["pairs"] += info["n_matches"]
        state["last"] = (m, rom, recs, info)
        if exch is None:
            state["gathered"] = info["n_accepted"]
        if dist_cuda:
            # the retired step's separators -> the node: copy into the send buffer of this retire's exchange, stamp the
            # count, ONE all-gather (async: it runs beside the steps in flight)
            # (all of it on a stream of its own: on the handle's stream -- which is also the first of the streams the steps
            #  are dealt over -- the copy, the header fill and the wait for the previous collective would queue behind and
            #  in front of every third step)
            ex = exchs[retired[0] % len(exchs)]
            retired[0] += 1
            with torch.cuda.stream(xstream):
                ex.finish()                   # the collective that last used these buffers (two retires ago)
                nrec = min(info["n_records"], int(ex.payload.shape[0]))
                if nrec:
                    f.memcpy_device_async(ex.payload.data_ptr(), info["d_records"], nrec * RB, xstream.cuda_stream)
                ex.exchange(nrec, finish=False)
            state["exch_last"] = ex

    def issue(k):
        """Step k enters the pipeline; the oldest step is retired first when the ring is full (its outputs were queued
        `depth` steps ago: the device has `depth - 1` steps of work queued while the host looks at them)."""
        if inflight[0] >= depth:
            retire()
        ts = time.perf_counter()
        f.step_issue(slot_a, slot_b)
        t_issue.append((time.perf_counter() - ts) * 1e3)
        inflight[0] += 1

    def drain_exchanges():
        """Every collective in flight finished, its header rows on the host; `gathered` = the last step's node-wide count."""
        if dist_cuda:
            with torch.cuda.stream(xstream):
                for ex in exchs:
                    ex.finish()
            torch.cuda.synchronize()
            state["gathered"] = sum(state["exch_last"].counts())

    def materialize_last():
        """The last retired step's separators in MATCH order + every match's flag, in the form the checks below take."""
        m, rom, recs, info = state["last"]
        ordered = np.ascontiguousarray(recs[rom[rom >= 0]])
        state["flags_last"] = (rom >= 0).copy()
        state["streamed_last"] = info["streamed"]
        state["last"] = (m.copy(), torch.from_numpy(ordered.view(np.uint8).reshape(-1, RB).copy()), len(m))

    def step():
        """One step with ONE synchronisation at its end, through the building blocks of include/sf_experimental.h (what
        value_one_synchronisation_per_step reports; also the warm-up, the comparison runs and the gloo rehearsal)."""
        # NN kernels, then -- in ONE library call -- the verification of the candidates: the NN filter's candidates are
        # verified speculatively on the device while the host reduces them to row minima, sorts and walks them
        # (data_handler.py:187-205); the walk's matches then pick their results (sf_api.hip).  Single GPU: only the
        # ACCEPTED separators leave the device (d_out = NULL), the compaction reads them through the index list the call
        # left behind.
        m = f.find_matches_and_verify_device(slot_a, slot_b, None if zero_copy else d_res.data_ptr(), cap=n_kf)
        n = len(m)
        # every candidate's success flag goes back to the two robots involved (failures feed the ignore list,
        # data_handler.py:406-408); only ACCEPTED separators are exchanged between GPUs / handed to the back-end
        # (data_handler.py:352-368).  The compaction leaves its count on the device.
        if exch is not None and coll_dev.type != "cuda":      # gloo rehearsal: the collective runs on CPU tensors
            n_acc = f.compact_accepted_device(d_res.data_ptr(), n, d_acc.data_ptr(), d_flags.data_ptr())
            acc = d_acc
            exch.payload[:n_acc].copy_(acc[:n_acc])
            exch.exchange(n_acc)
            d_cnt.fill_(n_acc)                                # (the packed copy below carries it to the host)
        elif exch is not None:
            # compaction writes records AND count straight into the exchange's send buffer
            acc, d_cnt_view = exch.payload, exch.send[0, :4].view(torch.int32)
            f.compact_accepted_device_async(d_res.data_ptr(), n, acc.data_ptr(), d_flags.data_ptr(), exch.count_ptr)
            exch.exchange(None, finish=False)                 # ONE all-gather, in flight beside the copies below
        else:
            acc = d_acc
            r_ptr, r_idx, r_n = f.last_match_results()
            f.compact_accepted_indexed_device_async(r_ptr, r_idx, n, hp + acc_off, hp + flags_off, hp)
        k_spec = n if zero_copy else min(n, spec_cap)
        if not zero_copy and acc is d_acc:
            # count + flags + a speculative prefix of the accepted separators (capacity for a 25 % acceptance rate): one
            # copy to the pinned mirror
            nb = acc_off + k_spec * RB
            h_pack[:nb].copy_(d_pack[:nb], non_blocking=True)
        elif not zero_copy:
            h_cnt.copy_(d_cnt_view, non_blocking=True)
            h_flags[:n].copy_(d_flags[:n], non_blocking=True)
            h_res[:k_spec].copy_(acc[:k_spec], non_blocking=True)   # accepted separators delivered to the host (pinned)
        if exch is not None:
            exch.finish()                                     # the copies above ran beside the collective
        if zero_copy:
            f.synchronize()                                   # (everything of the step is on the handle's streams)
            n_acc = int(h_cnt_np[0])
        else:
            torch.cuda.synchronize()
            n_acc = int(h_cnt[0])
        if n_acc > k_spec:                                    # more accepted than the speculative prefix held
            h_res[k_spec:n_acc].copy_(acc[k_spec:n_acc], non_blocking=True)
            torch.cuda.synchronize()
        host = h_res[:n_acc]
        state["pairs"] += n
        state["last"] = (m, host, n)
        state["gathered"] = sum(exch.counts()) if exch is not None else n_acc
        return n

    for _ in range(args.warmup):
        step()
    # a garbage collection inside the timed region shows up as one 8 - 10 ms step, and one right in front of it lets
    # the device idle long enough to drop its clocks (the first timed step then takes 7 ms): collect here, in front
    # of the last warm-up steps, and keep the collector off until the timed region has ended
    import gc
    gc.collect()
    gc.disable()
    # (the overlapped form, its second pinned block and the profiler's timing events are warmed too: the first
    # hipEventCreate of a process can cost milliseconds)
    f.prof_select(("k_verify_fused", "k_match_global", "k_ba_pass"))
    f.prof_enable(True)
    # bounded self-warm-up: the driver's few warm-up steps leave the clocks un-ramped (round 2: 0.53 ms for a kernel
    # that takes 0.47 once warm).  Steps are run, in the PIPELINED form of the timed region, until three consecutive ones
    # agree within 3 % -- but at least 100 (~50 ms of load) and at most 150: the device needs tens of milliseconds of the
    # overlapped load to settle (a 20-step timed region behind 5-20 such steps: median step 0.52 ms, 18.6 M pairs/s;
    # behind 100: 0.485 ms, 19.5 M; behind 400: the same -- round 3, one box), and host-side step times converge long
    # before that
    warm_ts = []
    sw_min = int(os.environ.get("BENCH_SELF_WARMUP_MIN", "100"))
    sw_max = int(os.environ.get("BENCH_SELF_WARMUP_MAX", "150"))
    for step_i in range(sw_max):
        ts = time.perf_counter()
        if pipelined:
            issue(step_i)
        else:
            step()
        warm_ts.append(time.perf_counter() - ts)
        # (with several ranks every step is a collective: all ranks must run the SAME number of warm-up steps, so the
        #  count is fixed there -- a break decided by a rank's own timings would leave the others waiting in an all-gather)
        if dist_on:
            if len(warm_ts) >= sw_min:
                break
        elif len(warm_ts) >= sw_min and max(warm_ts[-3:]) < 1.03 * min(warm_ts[-3:]):
            break
    if pipelined:
        while inflight[0]:
            retire()
        drain_exchanges()
    f.prof_enable(False)
    # HIP events over the timed region bracket ONLY the kernel the roofline prices (two timing events per launch
    # cost host time and a marker on the queue: with every kernel bracketed a step took 0.594 ms instead of 0.568);
    # the other kernels are surveyed in a short pass after the timed region
    dominant = (("k_verify_fused", "k_match_global") + (("k_ba_pass",) if args.bundle_adjustment else ())
                + (("k_nn_filter_f16",) if args.strict else ()))
    f.prof_reset()
    f.prof_select(dominant)
    f.prof_enable(True)
    state["pairs"] = 0
    del t_issue[:], t_retire[:]
    # The contract's bracket: barrier + synchronize, then EXACTLY K steps issued and retired, then synchronize + barrier.
    # Nothing else sits between the warm-up's last retire and t0 (round 3 drained, re-armed the profiler and collected
    # here: the device idled long enough for its first timed step to take 7 ms on the driver's box).
    if dist_on:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms = []
    if pipelined:
        for step_i in range(args.steps):
            ts = time.perf_counter()
            issue(step_i)                 # (retires step k - depth first when the ring is full)
            step_ms.append((time.perf_counter() - ts) * 1e3)
        while inflight[0] > 1:
            retire()
        retire(copy=True)                 # (waits for the last step's verification)
        drain_exchanges()                 # (the collectives' stream and the header copies too)
    else:
        for _ in range(args.steps):
            ts = time.perf_counter()
            step()                      # (ends with the step's one synchronisation)
            step_ms.append((time.perf_counter() - ts) * 1e3)
    torch.cuda.synchronize()
    if dist_on:
        td.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if pipelined:
        materialize_last()                # (check infrastructure: the last step's separators in match order)
        if dist_cuda:
            exch = state["exch_last"]     # (the checks below read the LAST step's exchange)
    timed_issue_ms, timed_retire_ms = list(t_issue), list(t_retire)
    prof = f.prof_get()
    f.prof_enable(False)
    # survey of every kernel (not timed): per-kernel HIP-event times of `survey_steps` further steps
    survey_steps = min(args.steps, 20)
    pairs_timed = state["pairs"]
    lm, lh, ln = state["last"]
    last_timed = (lm.copy(), lh.clone(), ln)      # (the survey steps rewrite the pinned block the view points into)
    f.prof_reset()
    f.prof_select(None)
    f.prof_enable(True)
    if pipelined:
        # (the form of the timed region: the library picks the verification's form per call -- inside overlapped steps the
        #  split one, k_match_split + k_chain -- so the survey runs the step pair too)
        for step_i in range(survey_steps):
            issue(step_i)
        while inflight[0]:
            retire()
        drain_exchanges()
    else:
        for _ in range(survey_steps):
            step()
    torch.cuda.synchronize()
    prof_all = f.prof_get()
    f.prof_enable(False)
    # ... and the dominant kernel ALONE on the chip (one step at a time, one stream): inside the timed region up to
    # SF_OPT_STEP_LANES launches of it run beside each other and share the CUs, so a launch's wall time there is a multiple
    # of what it needs by itself
    prof_alone = None
    if pipelined and not dist_on:
        try:
            f.set_option(_abi.SF_OPT_STEP_LANES, 1)
            f.prof_reset(); f.prof_select(dominant); f.prof_enable(True)
            for _ in range(8):
                f.step_issue(slot_a, slot_b)
                f.step_retire()
            torch.cuda.synchronize()
            prof_alone = f.prof_get()
            f.prof_enable(False)
            f.set_option(_abi.SF_OPT_STEP_LANES, int(os.environ.get("SF_STEP_LANES", "3")))
        except Exception as e:
            print("bench: exclusive-launch pass failed: %r" % (e,), file=sys.stderr)
    state["pairs"] = pairs_timed
    state["last"] = last_timed
    filter_dims = f.nn_last_filter_dims()      # prefix length the fp16 filter contracted (0: exact path)

    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    npairs = torch.tensor([state["pairs"]], dtype=torch.float64, device=coll_dev)
    if dist_on:
        td.all_reduce(t, op=td.ReduceOp.MAX)
        td.all_reduce(npairs, op=td.ReduceOp.SUM)
    elapsed = float(t.item())
    total_pairs = float(npairs.item())

    # ---- the same step with the fp32-ranking NN stage, for reference (untimed by the driver) -------
    # (the comparison runs below are informational: none of them may take the headline line down)
    alt = None
    if args.nn_precision == 1 and not args.no_extras:
        try:
            f.nn_set_precision(0)
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_alt = 0
            for _ in range(3):
                n_alt += step()
            torch.cuda.synchronize()
            alt = n_alt / (time.perf_counter() - t1)
            alt_m = state["last"][0]
        except Exception as e:
            print("bench: fp32-ranking comparison run failed: %r" % (e,), file=sys.stderr)
            alt = None
        f.nn_set_precision(1)
        step()

    # ---- the same steps with ONE synchronisation per step (no overlap of consecutive steps), untimed ----
    alt_sync = None
    if pipelined:
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_alt = 0
            for _ in range(min(args.steps, 50)):
                n_alt += step()
            torch.cuda.synchronize()
            alt_sync = n_alt / (time.perf_counter() - t1) * world
        except Exception as e:
            print("bench: one-synchronisation-per-step comparison run failed: %r" % (e,), file=sys.stderr)
        state["pairs"] = pairs_timed
        state["last"] = last_timed

    def pipe_rate(ff, sa_, sb_, n_steps):
        """Pairs per second of `n_steps` steps of the timed region's form (sf_step_issue / sf_step_retire, `depth` in
        flight) on handle ff, behind three warm-up steps; returns (rate, accepted separators of the last step)."""
        infl, n_pairs, last_acc = 0, 0, 0
        for _ in range(3):
            ff.step_issue(sa_, sb_)
            ff.step_retire()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_steps):
            if infl >= depth:
                info = ff.step_retire()[3]
                n_pairs += info["n_matches"]; last_acc = info["n_accepted"]; infl -= 1
            ff.step_issue(sa_, sb_)
            infl += 1
        while infl:
            info = ff.step_retire()[3]
            n_pairs += info["n_matches"]; last_acc = info["n_accepted"]; infl -= 1
        torch.cuda.synchronize()
        return n_pairs / (time.perf_counter() - t1) * world, last_acc

    # ---- the same steps with the prefix ladder of the NN filter forced to the FULL descriptor length (the cost on a
    # data set whose prefixes are uninformative; the timed steps contract `filter_dims` dimensions), untimed ----
    alt_full = None
    if args.nn_precision == 1 and not args.no_extras and pipelined and not dist_on:
        try:
            f.set_option(_abi.SF_OPT_NN_FULL_FILTER, 1)
            rate, _ = pipe_rate(f, slot_a, slot_b, 20)
            alt_full = {"value": rate, "contracted_dims": f.nn_last_filter_dims(), "steps": 20}
        except Exception as e:
            print("bench: full-length filter comparison run failed: %r" % (e,), file=sys.stderr)
        f.set_option(_abi.SF_OPT_NN_FULL_FILTER, 0)
        step()

    # ---- the same steps with EXACTLY iterations + 1 hypotheses per pass (ransac_adaptive_stop = 0), untimed: a second
    # handle with the same stores and databases (parameters are fixed at sf_create); and the STRICT reading of the
    # configuration: fixed hypotheses AND the full-length filter together ----
    alt_fixed = alt_strict = None
    if not args.no_extras and pipelined and not dist_on:
        try:
            q = _abi.copy_params(p)
            q.ransac_adaptive_stop = 0
            f2 = lib.SeparatorFinder(q, device=dev_index)
            f2.set_stream(torch.cuda.current_stream().cuda_stream)
            sa2, sb2 = upload_store(f2, feats, dev, n_kf, k, cols)
            getattr(f2, nn_append_received)(ta.data_ptr(), n_kf, dim)
            getattr(f2, nn_append_local)(tb.data_ptr(), n_kf, dim)
            torch.cuda.synchronize()
            rate, acc2 = pipe_rate(f2, sa2, sb2, 20)
            alt_fixed = {"value": rate, "accepted_last_step": acc2, "hypotheses_per_pass": args.iterations + 1, "steps": 20}
            if args.nn_precision == 1:
                f2.set_option(_abi.SF_OPT_NN_FULL_FILTER, 1)
                pipe_rate(f2, sa2, sb2, 6)
                f2.prof_reset(); f2.prof_select(None); f2.prof_enable(True)
                rate_p, _ = pipe_rate(f2, sa2, sb2, 10)          # (every kernel bracketed: a survey, not the rate)
                prof_strict = f2.prof_get()
                f2.prof_enable(False)
                rate, acc2 = pipe_rate(f2, sa2, sb2, 30)
                alt_strict = {"value": rate, "accepted_last_step": acc2, "hypotheses_per_pass": args.iterations + 1,
                              "kernel_ms_per_step": {kn: ms / 13.0 for kn, (cnt, ms) in prof_strict.items() if cnt},
                              "contracted_dims": f2.nn_last_filter_dims(), "steps": 30,
                              "what": "every pass evaluates all %d hypotheses (no adaptive stop) AND the NN filter contracts "
                                      "the full descriptor length -- the configuration read to the letter" % (args.iterations + 1)}
            f2.close()
        except Exception as e:
            print("bench: fixed-iteration / strict comparison runs failed: %r" % (e,), file=sys.stderr)

    # ---- the same step with the VALU matcher (xor + popcount; north_star's literal kernel mix), untimed ----
    alt_valu = None
    if os.environ.get("SF_MATCH_MFMA", "1") != "0" and not args.no_extras:
        try:
            f.set_option(_abi.SF_OPT_MATCH_MFMA, 0)
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_alt = 0
            for _ in range(5):
                n_alt += step()
            torch.cuda.synchronize()
            alt_valu = n_alt / (time.perf_counter() - t1)
        except Exception as e:
            print("bench: VALU-matcher comparison run failed: %r" % (e,), file=sys.stderr)
            alt_valu = None
        f.set_option(_abi.SF_OPT_MATCH_MFMA, 1)
        step()

    # ---- SURVEY 8(d) "incl. H2D of features": candidate pairs handed over as HOST buffers (the wire layout of
    # EstTransform.srv), verified, results back in host memory -- packing, PCIe both ways and the verification ----
    pcie = None
    if world == 1 and not args.no_extras:
        try:
            S = min(2048, n_kf)
            A_h = [_abi.FeatureArrays(feats["desc_a"][i], feats["xyz_a"][i], feats["kp_a"][i]) for i in range(S)]
            B_h = [_abi.FeatureArrays(feats["desc_b"][i], feats["xyz_b"][i], feats["kp_b"][i]) for i in range(S)]
            f.estimate_transform_batch(A_h[:256], B_h[:256])
            fa_h, fb_h = _abi.features_array(A_h), _abi.features_array(B_h)    # (the sf_features arrays a C host holds)
            res_h = np.zeros(S, dtype=_abi.RESULT_DTYPE)

            def call_h():
                f._check(f._L.sf_estimate_transform_batch(f._h, fa_h, fb_h, S, res_h.ctypes.data))
            call_h()
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                call_h()
            dt = (time.perf_counter() - t1) / reps
            pcie = {"value": S / dt, "unit": "pairs/s", "pairs_per_call": S, "ms_per_call": dt * 1e3,
                    "accepted": int(res_h["success"].sum()),
                    "what": "sf_estimate_transform_batch on host buffers: features packed into pinned staging, H2D, "
                            "verification, results D2H; never `value`"}
            del A_h, B_h
        except Exception as e:
            print("bench: PCIe-inclusive run failed: %r" % (e,), file=sys.stderr)

    # ---- informational: the same steps software-pipelined over two streams (untimed by the driver) ----
    # A deployment that serves a stream of independent batches can run the NN stage of batch i+1 (MFMA +
    # HBM + host walk, on a second handle with its own stream) while batch i is being verified.
    # `value` above is NOT measured this way: its steps run strictly one after the other.
    piped = None
    if world == 1 and args.nn_precision == 1 and not args.no_pipelined_extra and not args.no_extras:
        f_nn = lib.SeparatorFinder(p, device=dev_index)          # own non-blocking stream
        getattr(f_nn, nn_append_received)(ta.data_ptr(), n_kf, dim)
        getattr(f_nn, nn_append_local)(tb.data_ptr(), n_kf, dim)
        f_nn.synchronize()

        def launch_verify(m):
            f.verify_matches_device(m, slot_a, slot_b, d_res.data_ptr())   # asynchronous on the verification stream
            return len(m)

        h_res_p = torch.empty((n_kf, _abi.RESULT_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
        h_flags_p = torch.empty(n_kf, dtype=torch.bool).pin_memory()

        def finish(n):
            n_acc = f.compact_accepted_device(d_res.data_ptr(), n, d_acc.data_ptr(), d_flags.data_ptr())
            h_flags_p[:n].copy_(d_flags[:n], non_blocking=True)
            h_res_p[:n_acc].copy_(d_acc[:n_acc], non_blocking=True)
            torch.cuda.synchronize()
            return int(n_acc)

        def run_piped(k_steps):
            pairs = 0
            m = f_nn.nn_find_matches(cap=n_kf)
            for i in range(k_steps):
                n = launch_verify(m)                      # asynchronous on the verification stream
                if i + 1 < k_steps:
                    m_next = f_nn.nn_find_matches(cap=n_kf)   # overlaps the verification of batch i
                got = finish(n)
                pairs += n
                if i + 1 < k_steps:
                    m = m_next
            return pairs, got

        try:
            run_piped(3)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_p, got_p = run_piped(max(10, args.steps))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            piped = {"value": n_p / dt, "unit": "pairs/s", "ms_per_step": dt / max(10, args.steps) * 1e3,
                     "accepted_last_step": got_p,
                     "note": "NN stage of batch i+1 on a second handle/stream while batch i is verified; "
                             "informational (round 1's two-handle form); `value` overlaps consecutive steps on ONE handle, see steps_overlap"}
        except Exception as e:   # informational only: never let it take the headline line down
            piped = {"error": repr(e)}
        f_nn.close()
    # ---- NN stage alone on SURVEY 8(d)'s generator (5 % planted revisits at distance ~0.05, 95 % independent rows),
    # untimed: queries/s and the matrix-core rate of whatever prefix level the filter settles on ----
    nn_only = None
    if world == 1 and not args.no_extras:
        try:
            loc, oth, planted = synth.make_netvlad(777, n_kf, n_kf, dim, planted_frac=0.05)
            f.nn_reset()
            tl, to = torch.from_numpy(loc).to(dev), torch.from_numpy(oth).to(dev)
            f.nn_append_local_device(tl.data_ptr(), n_kf, dim)
            f.nn_append_received_device(to.data_ptr(), n_kf, dim)
            torch.cuda.synchronize()
            for _ in range(3):
                mm = f.nn_find_matches(cap=n_kf)
            f.prof_reset(); f.prof_select(None); f.prof_enable(True)
            t1 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                mm = f.nn_find_matches(cap=n_kf)
            dtn = time.perf_counter() - t1
            prn = f.prof_get()
            f.prof_enable(False)
            kd = f.nn_last_filter_dims() or dim
            kern = "k_nn_filter_f16" if args.nn_precision == 1 else "k_nn_argmin"
            kms = prn[kern][1] / max(prn[kern][0], 1)
            nn_only = {"generator": "5 %% planted revisits (distance ~0.05), 95 %% independent unit rows; %d x %d x %d" % (n_kf, n_kf, dim),
                       "queries_per_s": n_kf * reps / dtn, "ms_per_query_block": dtn / reps * 1e3,
                       "matches_found": int(len(mm)), "planted": int((planted >= 0).sum()),
                       "contracted_dims": kd,
                       "kernel": kern + ("_k128r" if kern == "k_nn_filter_f16" and kd == 128 else ""), "kernel_ms": kms,
                       "kernel_tflops": 2.0 * n_kf * n_kf * kd / (kms * 1e-3) / 1e12 if kms > 0 else 0.0}
            del tl, to
        except Exception as e:
            print("bench: NN-only run failed: %r" % (e,), file=sys.stderr)
    del ta, tb

    # ---- the rows next to the hot path: keyframe features and NetVLAD inference ----
    next_rows = None
    if world == 1 and rank == 0 and not args.no_extras and not args.no_cpu_baseline:
        try:
            next_rows = measure_next_rows(dev)
        except Exception as e:
            print("bench: next-rows measurement failed: %r" % (e,), file=sys.stderr)

    # ---- sanity of the timed work (rank 0): the separators found are the planted revisits -----------
    m, host, n = state["last"]
    flags = state["flags_last"] if state.get("flags_last") is not None else h_flags[:n].numpy().copy()
    truth = feats["is_true"][m["idx_local"]]
    same = m["idx_local"] == m["idx_other"]
    accepted = int(flags.sum())
    correct = int((flags == (truth & same)).sum())
    sep = np.frombuffer(host.numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
    if exch is not None:      # outside the timed region: every gathered record is an accepted separator
        allrec, cts = exch.all_gathered()
        gat = np.frombuffer(allrec.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
        mine = gat[sum(cts[:rank]): sum(cts[:rank + 1])]
        # (streamed records arrive in completion order and cover every accepted CANDIDATE: a superset of the matches')
        have = set(r.tobytes() for r in mine)
        all_ok = (bool(gat["success"].all()) and len(gat) == state["gathered"] and len(mine) >= len(sep)
                  and all(r.tobytes() in have for r in sep))
    else:
        all_ok = bool(sep["success"].all()) and len(sep) == state["gathered"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        pairs_per_step = total_pairs / args.steps / world
        assert k == args.features and cols == args.desc_bytes      # (a loop variable once shadowed k: 5 544 B per pair)
        bpp = bytes_per_pair(k, cols)
        # dominant kernel: the fused per-pair pipeline (matching + both RANSAC passes + guided matching +
        # result: the whole verification the 44 352 B/pair figure of SURVEY 8(d) describes), or the
        # matching kernel when the stage kernels run (PnP estimator, SF_FUSED=0)
        dom = "k_verify_fused" if prof.get("k_verify_fused", (0, 0.0))[0] > 0 else "k_match_global"
        # the split form of the 3D-3D verification (the library's choice inside overlapped steps, SF_OPT_STEP_SPLIT; always
        # with SF_FUSED=2): k_match_split over every candidate -- the dominant kernel, priced with ITS share of the
        # 44 352 B (both keyframes' descriptors + the result record: 2*K*C + 352) -- and k_chain over the survivors (the
        # 3D points, 2*K*12 B per surviving pair, and the latency-bound motion-estimation chains), reported beside it
        split_form = (args.estimator == "3d3d" and prof.get("k_verify_fused", (0, 0.0))[0] > 0
                      and prof.get("k_match_global", (0, 0.0))[0] > 0)
        chain_ms = None
        if split_form:
            chain_ms = prof["k_verify_fused"][1] / max(prof["k_verify_fused"][0], 1)
            dom = "k_match_global"
        nm, tm = prof[dom]
        match_ms = tm / max(nm, 1)
        # pairs one launch of the dominant kernel processes (big batches are cut in two halves on two streams)
        pairs_per_launch = pairs_per_step * args.steps / max(nm, 1)
        dom_name = "k_match_split" if split_form else dom
        bpp_dom = (2 * k * cols + 352) if split_form else bpp
        pmc = pmc_traffic(dom_name, pairs_per_launch)
        ach = pairs_per_launch * bpp_dom / (match_ms * 1e-3) / 1e9 if match_ms > 0 else 0.0
        lanes_in_use = 1 if not pipelined else int(os.environ.get("SF_STEP_LANES", "3"))
        alone_roof = None
        if prof_alone is not None and prof_alone.get(dom, (0, 0.0))[0] > 0:
            a_ms = prof_alone[dom][1] / prof_alone[dom][0]
            a_ach = pairs_per_launch * bpp_dom / (a_ms * 1e-3) / 1e9
            alone_roof = {"avg_launch_ms": a_ms, "launches": prof_alone[dom][0], "achieved": a_ach, "unit": "GB/s",
                          "frac": a_ach / HBM_PEAK_GBS}
        nn_kernel, nn_peak = (("k_nn_filter_f16", MFMA_F16_PEAK_TF) if args.nn_precision == 1
                              else ("k_nn_argmin", MFMA_F32_PEAK_TF))
        nn_n, nn_t = prof_all[nn_kernel]
        nn_ms = nn_t / max(nn_n, 1)
        # flops the launched kernel really performs: the fp16 filter contracts a PREFIX of the descriptor
        # (k_nn.hip, nn_run_filter: adaptive 128 / 512 / full), the fp32 ranking kernel all D dimensions
        k_eff = dim if args.nn_precision == 0 else (filter_dims or dim)
        nn_tf = 2.0 * n_kf * n_kf * k_eff / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0
        def survey_name(kname):
            if args.estimator == "pnp":
                return kname.replace("k_ransac", "k_pnp")
            if split_form and pipelined:      # (the profiling slots of the library keep the stage names)
                return {"k_match_global": "k_match_split", "k_verify_fused": "k_chain"}.get(kname, kname)
            return kname

        out = {
            "metric": METRIC,
            "value": total_pairs / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8+f32+f64" if args.nn_precision == 0 else "u8+f16+f32+f64",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[1]: 1xMI355X per rank, 2 robots x %d keyframes, %d-D %s NetVLAD, "
                            "%d x %d-bit ORB per keyframe, " + ("EXACTLY %d + 1 RANSAC hypotheses per pass (no adaptive stop: the "
                            "configuration read to the letter, --strict)" if args.strict else
                            "<= %d RANSAC hypotheses per pass with PCL's adaptive stop "
                            "(p = 0.99; it ends inside the first 16-hypothesis round on these correspondences -- the "
                            "fixed-count figure is value_fixed_iterations)") + " (%s), both registration passes, %.0f %% true "
                            "revisits, every row has a perceptual alias under netvlad_distance (all %d rows become "
                            "candidates); NN filter contracted %s of %d dimensions (" + ("the full length, --strict" if args.strict
                            else "adaptive prefix ladder -- the full-length figure is value_full_length_filter") + ")") % (
                                n_kf, dim, "fp16" if args.netvlad_f16 else "fp32", k, cols * 8, args.iterations,
                                estimator_text(args), 100 * args.true_frac, n_kf,
                                (filter_dims or dim) if args.nn_precision == 1 else dim, dim),
                "pairs_per_step_per_gpu": pairs_per_step,
                "parallelism": "pairs sharded by robot pair, 1 rank per GPU" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": dom_name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": (pmc or {}).get("bytes"),
                "traffic_measured_in_this_run": False,
                "traffic_source": pmc,
                "compute": compute_note(dom_name, k, cols, pairs_per_launch, match_ms),
                "bytes_per_pair": bpp_dom, "pairs_per_launch": pairs_per_launch, "avg_launch_ms": match_ms,
                "launches_per_step": nm / args.steps,
                # `achieved` / `frac` above: algorithmic bytes of a launch over its average duration INSIDE the timed
                # region (HIP events), where the launches of the steps in flight share the chip; the same kernel alone
                # on the chip and the step as a whole:
                "launches_sharing_the_chip": lanes_in_use,
                "alone": alone_roof,
                "whole_step": {"bytes_per_step": pairs_per_step * bpp, "ms_per_step": ms_per_step,
                               "achieved": pairs_per_step * bpp / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                               "frac": pairs_per_step * bpp / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            },
            "verification_form": ("split: k_match_split over every candidate + k_chain over the survivors" if split_form
                                  else "split (PnP): k_match_split + k_chain_pnp (the library's k_verify_fused profiling slot)"
                                  if args.estimator == "pnp" and dom == "k_verify_fused"
                                  else "fused: k_verify_fused" if dom == "k_verify_fused" else "stage kernels"),
            "roofline_nn": {
                "kernel": nn_kernel + ("_k128r" if nn_kernel == "k_nn_filter_f16" and k_eff == 128 else ""),
                "bound": "mfma", "achieved": nn_tf, "peak": nn_peak,
                "unit": "TFLOP/s", "frac": nn_tf / nn_peak, "avg_launch_ms": nn_ms, "contracted_dims": k_eff,
            },
            "kernel_ms_per_step": {survey_name(kname): (ms / survey_steps) for kname, (cnt, ms) in prof_all.items()},
            "kernel_ms_per_step_source": "HIP events around every kernel in %d steps AFTER the timed region, in the timed "
                                         "region's form (kernels of the two steps in flight run beside each other, so a "
                                         "kernel's figure includes what it lost to its neighbour); the timed region "
                                         "brackets only %s" % (survey_steps,
                                                               "k_match_split and k_chain" if split_form else dom),
            "check": {"accepted_last_step": accepted, "decisions_matching_ground_truth": correct, "of": int(n),
                      "accepted_separators_gathered_per_step": state.get("gathered", 0),
                      "gathered_records_all_accepted": all_ok},
            "input_generation_s": t_gen,
            # host wall time of each timed step's issue (which first retires the oldest step once the ring is full); the
            # slowest step is named, with what its issue and the retire inside it took
            "step_ms_spread": {"min": float(np.min(step_ms)), "median": float(np.median(step_ms)),
                               "p90": float(np.percentile(step_ms, 90)), "max": float(np.max(step_ms)),
                               "argmax": int(np.argmax(step_ms)),
                               "issue_ms_of_argmax": (timed_issue_ms[int(np.argmax(step_ms))]
                                                      if pipelined and len(timed_issue_ms) > int(np.argmax(step_ms)) else None),
                               "issue_ms_median": float(np.median(timed_issue_ms)) if pipelined and timed_issue_ms else None,
                               "retire_ms_median": float(np.median(timed_retire_ms)) if pipelined and timed_retire_ms else None,
                               "retire_ms_max": float(np.max(timed_retire_ms)) if pipelined and timed_retire_ms else None,
                               "steps_in_flight": depth if pipelined else 1},
        }
        if split_form:
            whole_ms = match_ms + chain_ms
            out["roofline"]["chain_kernel"] = {
                "kernel": "k_chain", "avg_launch_ms": chain_ms,
                "note": "RANSAC, guess-guided matching, RANSAC and the result of the pairs whose matching found enough "
                        "correspondences (a fifth of the candidates here): latency-bound fp64 chains, 2*K*12 B of 3D points "
                        "per surviving pair"}
            out["roofline"]["whole_verification"] = {
                "bytes_per_pair": bpp, "ms_per_launch_pair": whole_ms,
                "achieved": pairs_per_launch * bpp / (whole_ms * 1e-3) / 1e9 if whole_ms > 0 else 0.0, "unit": "GB/s",
                "note": "SURVEY section 8(d)'s 44 352 B per pair over the two kernels' launch times added (they overlap the "
                        "neighbouring step's kernels, not each other)"}
        if args.bundle_adjustment and prof.get("k_ba_pass", (0, 0.0))[0] > 0:
            # The as-shipped flow's dominant kernel is the bundle adjustment (k_ba_pass: one launch sequence per pass behind
            # the estimate).  Unit = one adjusted pass of one surviving pair; its compulsory bytes: the pass's correspondence
            # list and the estimate's inlier bytes (4 + 1 B per correspondence), per inlier word both 3D points and both
            # keypoints (12 + 12 + 16 + 16 B), the pass state read and written (2 x 80 B) -- DESIGN.md section 5.  The counts
            # come from the last timed step's accepted separators (matches / inliers of both passes).
            n_ba, t_ba = prof["k_ba_pass"]
            ba_ms = t_ba / max(n_ba, 1)
            b1 = 5.0 * sep["matches_pass1"].astype(np.float64) + 56.0 * sep["inliers_pass1"] + 160.0
            b2 = 5.0 * sep["matches"].astype(np.float64) + 56.0 * sep["inliers"] + 160.0
            units = float(len(sep))
            bytes_per_launch = float(b1.sum() + b2.sum()) / 2.0
            ba_ach = bytes_per_launch / (ba_ms * 1e-3) / 1e9 if ba_ms > 0 else 0.0
            out["roofline_matching_kernel"] = out["roofline"]
            out["roofline"] = {
                "kernel": "k_ba_pass", "bound": "hbm", "achieved": ba_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ba_ach / HBM_PEAK_GBS, "traffic": (pmc_traffic("k_ba_pass<%d, %s, true>" % (int(os.environ.get("SF_BA_NW", "1")), "true" if args.estimator == "pnp" else "false"),
                                         units, any_size=True) or {}).get("bytes"),
                "traffic_measured_in_this_run": False,
                "units_per_launch": units,
                "unit_of_work": "one adjusted pass of one surviving pair (two launch sequences per step: pass 1, pass 2)",
                "bytes_per_unit": bytes_per_launch / max(units, 1.0), "avg_launch_ms": ba_ms,
                "launches_per_step": n_ba / args.steps, "launches_sharing_the_chip": lanes_in_use,
                "mean_words_per_unit": float((sep["inliers_pass1"].sum() + sep["inliers"].sum()) / max(2.0 * units, 1.0)),
                "compute": {"note": "the adjustment is fp64 vector arithmetic (Schur-complement Levenberg-Marquardt, ~1 000 fp64 "
                                    "instructions per word and evaluation, no FMA contraction: canonical order), not memory: "
                                    "HBM is the reporting basis BASELINE asks for, the vector unit is what bounds it",
                            "counters": sq_counters("k_ba_pass")},
                "whole_step": out["roofline_matching_kernel"]["whole_step"],
            }
        if args.strict and prof.get("k_nn_filter_f16", (0, 0.0))[0] > 0:
            n_f, t_f = prof["k_nn_filter_f16"]
            fms = t_f / n_f
            ftf = 2.0 * n_kf * n_kf * dim / (fms * 1e-3) / 1e12
            out["roofline_verification_kernel"] = out["roofline"]
            out["roofline"] = {"kernel": "k_nn_filter_f16", "bound": "mfma", "achieved": ftf, "peak": MFMA_F16_PEAK_TF,
                               "unit": "TFLOP/s", "frac": ftf / MFMA_F16_PEAK_TF, "avg_launch_ms": fms,
                               "flop_per_launch": 2.0 * n_kf * n_kf * dim, "launches_per_step": n_f / args.steps,
                               "traffic": (pmc_traffic("k_nn_filter_f16", 0, any_size=True) or {}).get("bytes"),
                               "traffic_measured_in_this_run": False, "launches_sharing_the_chip": lanes_in_use,
                               "whole_step": out["roofline_verification_kernel"]["whole_step"]}
        out["self_warmup_steps"] = len(warm_ts)
        out["steps_overlap"] = bool(pipelined)
        out["timed_step_entry_points"] = "sf_step_issue + sf_step_retire" if pipelined else "sf_experimental.h building blocks"
        # SF_OPT_STEP_OVERLAP (the library's default, SF_STEP_OVERLAP=0 turns it off): the two steps in flight on two
        # streams -- unless the separators are mirrored into an exchange buffer (N > 1), where one stream orders the
        # collective behind the step
        out["steps_on_several_streams"] = bool(pipelined and os.environ.get("SF_STEP_OVERLAP", "1") != "0")
        out["stream_placement"] = f.stream_placement()
        if dist_cuda:
            out["exchange_buffers"] = len(exchs)
        if exch is not None:
            # who took part in the separator exchange, read from the gathered header rows of the last step's all-gather
            # (every rank stamps its own number there), and what one collective moves
            seen = exch.ranks_seen()
            out["collective"] = {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend,
                                 "op": "all_gather_into_tensor, one per step, started at retire",
                                 "ranks_in_allgather": len(seen), "ranks_seen": seen,
                                 "bytes_per_step": exch.bytes_per_exchange(),
                                 "records_per_rank_last_step": exch.counts()}
        out["accepted_separators_streamed_from_the_kernel"] = bool(state.get("streamed_last", False))
        if alt_sync is not None:
            out["value_one_synchronisation_per_step"] = alt_sync
        if alt_fixed is not None:
            out["value_fixed_iterations"] = alt_fixed["value"]
            out["fixed_iterations"] = alt_fixed
        if alt_full is not None:
            out["value_full_length_filter"] = alt_full["value"]
            out["full_length_filter"] = alt_full
        if alt_strict is not None:
            out["value_strict"] = alt_strict["value"]
            out["strict"] = alt_strict
            # the strict form's dominant kernel: the fp16 MFMA filter over the full descriptor length (2 N N D flop per
            # launch against the dense fp16 peak); its time per launch from the survey above (every kernel bracketed, the
            # steps in flight sharing the chip); `python bench.py --strict` times this form as the line's own value
            ks = alt_strict.get("kernel_ms_per_step", {})
            if ks.get("k_nn_filter_f16"):
                fms = ks["k_nn_filter_f16"]
                ftf = 2.0 * n_kf * n_kf * dim / (fms * 1e-3) / 1e12
                dom_s = max(ks, key=ks.get)
                out["roofline_strict"] = {"kernel": "k_nn_filter_f16", "bound": "mfma", "achieved": ftf, "peak": MFMA_F16_PEAK_TF,
                                          "unit": "TFLOP/s", "frac": ftf / MFMA_F16_PEAK_TF, "avg_launch_ms": fms,
                                          "traffic": None, "longest_kernel_of_the_survey": dom_s,
                                          "ms_per_step": 1e3 * pairs_per_step / alt_strict["value"] if alt_strict["value"] else None}
        if nn_only is not None:
            out["nn_only_survey_8d_generator"] = nn_only
        if next_rows:
            out["next_rows"] = next_rows
        if piped is not None:
            out["pipelined_two_streams"] = piped
        if pcie is not None:
            out["value_pcie_inclusive"] = pcie["value"]
            out["pcie_inclusive"] = pcie
        if alt_valu is not None:
            # the Hamming table by xor + popcount on the VALU instead of the fp4 matrix cores (identical outputs)
            out["value_with_valu_matcher"] = alt_valu * world
        if alt is not None:
            out["value_with_fp32_nn_ranking"] = alt * world
            out["check"]["nn_matches_identical_fp32_vs_f16filter"] = bool(
                np.array_equal(alt_m["idx_local"], m["idx_local"]) and np.array_equal(alt_m["idx_other"], m["idx_other"])
                and np.array_equal(alt_m["distance"], m["distance"]))
        if world == 1 and not args.no_cpu_baseline:
            # parity of the timed work itself: every accepted separator of the last step whose pair the oracle sample
            # covers is compared with the oracle's result of that pair, byte for byte (368 B each)
            out["cpu_baseline"] = cpu_baseline(p, feats, nv_a, nv_b, n_kf, args.cpu_sample_pairs,
                                               args.cpu_sample_rows)
            ores = out["cpu_baseline"].pop("_results")
            acc_idx = m["idx_local"][flags.astype(bool) & same]          # pair index of every accepted separator
            acc_rec = sep[(same[flags.astype(bool)])] if len(sep) == int(flags.sum()) else sep[:0]
            inside = acc_idx < len(ores)
            ident = sum(1 for rec, j in zip(acc_rec[inside], acc_idx[inside]) if rec.tobytes() == ores[j].tobytes())
            out["parity_in_this_run"] = {"accepted_separators_compared_with_the_oracle": int(inside.sum()),
                                         "byte_identical": int(ident)}
        print(json.dumps(out))
    # teardown order: the exchanges' tensors and the process group have seen the handle's second stream (collectives
    # were handed over on it); they go first, while that stream still exists
    exch = None
    exchs.clear()
    state.clear()
    gc.collect()
    torch.cuda.synchronize()
    if dist_on:
        td.destroy_process_group()
    f.close()


if __name__ == "__main__":
    main()
