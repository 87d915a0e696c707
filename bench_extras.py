"""bench_extras.py -- everything of the benchmark that is NOT the contract's timed step (bench.py): the roofline /
evidence helpers, the CPU baseline, the rows next to the hot path, the other BASELINE workloads (--partition 8e,
--workload cfg3 / cfg4, each printing its own line) and the untimed comparison runs behind the timed region
(run_untimed_comparisons: fp32 NN ranking, one synchronisation per step, full-length filter, fixed iterations, the
strict form with its kernel survey, VALU matcher, PCIe-inclusive rate, two-handle pipeline, NN stage alone, next rows).
Split out of bench.py in round 5; nothing here is inside a timed region of the default line."""
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
        if "cfg3" in os.path.basename(path):     # (profiles of the other workload: K = 1000, 100 000 pairs per launch)
            continue
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
        scan = ("four resident column tiles per wavefront (three workgroups per CU), the top-2 update of tile j - 1 issued in the "
                "gaps of tile j's four MFMAs from a second accumulator tuple (k_match.hip, mf_pipe_*): a tile pair holds the "
                "matrix pipe for 4 x 32 cycles and the vector issue port for 4 x 8 + 26 x 4.  Counters of the launch alone on the "
                "chip (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x GRBM_GUI_ACTIVE / 8): pipe 75 % busy at K = 1000 "
                "(profiles/r05s_cfg3_split_sq_k_match_split.json, sustained clock 1.69 GHz), ~58 % at K = 500 where staging, "
                "compaction and dispatch are a quarter of a pair's workgroup time" if dom == "k_match_split" else
                "per 32-row tile and SIMD the 8 MFMAs hold the fp4 pipe for 8 x 32 cycles; the vector issue port is held 4 cycles "
                "by each of the scan's 60 vector instructions and 8 by each MFMA (MI355X_MICROARCH.md, cycle constants; "
                "tools/ubench/mfma_port.hip) (DESIGN.md section 5; profiles/r03m_fewer_valu_no_gain.log)")
        return {"note": "matching = v_mfma_f32_32x32x64_f8f6f4 over +-1-encoded descriptor bits (2*K*K*bits flop per pair, "
                        "rows unpadded) + 1.25 VALU ops per table cell for the top-2 scan: " + scan
                        + ("; the launch also holds both motion-estimation chains of the surviving pairs, which are "
                           "latency-bound" if dom == "k_verify_fused" else ""),
                "matrix_core": {"achieved": tf, "peak": MFMA_FP4_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_FP4_PEAK_TF},
                "descriptor_pairs_per_s": per_s, "counters": sq_counters(dom)}
    return {"note": "VALU matcher (SF_MATCH_MFMA=0): per 256-bit descriptor pair 8 v_xor (full rate) + 8 v_bcnt_u32_b32 "
                    "(HALF rate on gfx950, tools/ubench/valu_rate.hip) + 6 16-bit min/max",
            "descriptor_pairs_per_s": per_s, "valu_busy_from_counters": sq_evidence(dom)}


def matrix_pipe_roofline(hbm_roof, k, cols):
    """The dominant kernel of the 3D-3D lines (k_match_split, or k_verify_fused whose matching part it is) is bound by the
    fp4 matrix pipe, not by HBM: its roofline object on THAT basis -- algorithmic flops (2 * K * K * bits per pair: one
    multiply-add per pair of descriptor bits of the K x K Hamming table) of a launch over its average duration, against
    the dense fp4 MFMA peak of MI355X_MICROARCH.md -- with SURVEY section 8(d)'s HBM basis (bytes per pair) kept whole
    under "hbm_basis".  `traffic` stays the HBM bytes per launch of the PMC passes."""
    mc = hbm_roof["compute"]["matrix_core"]
    flop = 2.0 * k * k * cols * 8
    alone = hbm_roof.get("alone")
    out = {"kernel": hbm_roof["kernel"], "bound": "mfma", "achieved": mc["achieved"], "peak": mc["peak"],
           "unit": "TFLOP/s", "frac": mc["frac"], "traffic": hbm_roof.get("traffic"),
           "traffic_measured_in_this_run": False, "traffic_source": hbm_roof.get("traffic_source"),
           "flop_per_pair": flop, "pairs_per_launch": hbm_roof["pairs_per_launch"],
           "avg_launch_ms": hbm_roof["avg_launch_ms"], "launches_per_step": hbm_roof["launches_per_step"],
           "launches_sharing_the_chip": hbm_roof["launches_sharing_the_chip"],
           "basis": "v_mfma_f32_32x32x64_f8f6f4 on +-1-coded descriptor bits: 2*K*K*bits flop per pair (rows unpadded), "
                    "peak = dense fp4 at the nominal clock; `achieved` inside the timed region, where the launches of the "
                    "steps in flight share the chip -- `alone` is the same kernel by itself",
           "alone": None if not alone else {
               "avg_launch_ms": alone["avg_launch_ms"], "launches": alone["launches"],
               "achieved": hbm_roof["pairs_per_launch"] * flop / (alone["avg_launch_ms"] * 1e-3) / 1e12, "unit": "TFLOP/s",
               "frac": hbm_roof["pairs_per_launch"] * flop / (alone["avg_launch_ms"] * 1e-3) / 1e12 / mc["peak"]},
           "compute": hbm_roof["compute"],
           "hbm_basis": {kk: hbm_roof[kk] for kk in ("bound", "achieved", "peak", "unit", "frac", "bytes_per_pair", "alone",
                                                     "whole_step") if kk in hbm_roof}}
    for kk in ("chain_kernel", "whole_verification"):
        if kk in hbm_roof:
            out[kk] = hbm_roof[kk]
    return out


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
            # the verification launch is bound by the fp4 matrix pipe (its matching part: 75 % busy alone on the chip in the
            # split form's matcher, profiles/r05s_cfg3_split_sq_k_match_split.json); SURVEY 8(d)'s HBM basis beside it
            "roofline": {"kernel": "k_verify_fused (WIDE form: K = 1000)", "bound": "mfma",
                         "achieved": flop4 / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0, "peak": MFMA_FP4_PEAK_TF,
                         "unit": "TFLOP/s",
                         "frac": flop4 / (launch_ms * 1e-3) / 1e12 / MFMA_FP4_PEAK_TF if launch_ms > 0 else 0.0,
                         "traffic": None, "flop_per_pair": 2.0 * k * k * cols * 8,
                         "pairs_per_launch": float(n_kf), "avg_launch_ms": launch_ms,
                         "basis": "K x K Hamming table on the fp4 matrix cores (2*K*K*bits flop per pair, rows unpadded) + the "
                                  "motion-estimation chains of the surviving fifth inside the same launch; three launches (one "
                                  "per robot pair) share the chip inside the timed region",
                         "hbm_basis": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ach / HBM_PEAK_GBS, "bytes_per_pair": bpp}},
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



def run_untimed_comparisons(ctx):
    """The informational comparison runs behind the timed region of the default line (none of them may take the headline
    line down: each is wrapped).  ctx: the names of bench.main() they read; returns a dict of what the line reports."""
    _abi = ctx._abi
    args = ctx.args
    cols = ctx.cols
    d_acc = ctx.d_acc
    d_flags = ctx.d_flags
    d_res = ctx.d_res
    depth = ctx.depth
    dev = ctx.dev
    dev_index = ctx.dev_index
    dim = ctx.dim
    dist_on = ctx.dist_on
    f = ctx.f
    feats = ctx.feats
    k = ctx.k
    last_timed = ctx.last_timed
    lib = ctx.lib
    n_kf = ctx.n_kf
    nn_append_local = ctx.nn_append_local
    nn_append_received = ctx.nn_append_received
    p = ctx.p
    pairs_timed = ctx.pairs_timed
    pipelined = ctx.pipelined
    rank = ctx.rank
    slot_a = ctx.slot_a
    slot_b = ctx.slot_b
    state = ctx.state
    step = ctx.step
    synth = ctx.synth
    ta = ctx.ta
    tb = ctx.tb
    torch = ctx.torch
    world = ctx.world
    alt_m = None
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

    return dict(alt=alt, alt_fixed=alt_fixed, alt_full=alt_full, alt_m=alt_m, alt_strict=alt_strict, alt_sync=alt_sync, alt_valu=alt_valu, next_rows=next_rows, nn_only=nn_only, pcie=pcie, piped=piped)

