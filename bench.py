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
region.  Weak scaling: every rank owns an independent robot pair of the same size.
"""
import argparse
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


def pmc_traffic(kernel_prefix, pairs_per_launch):
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
                if abs(ppl - pairs_per_launch) > 1:
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
        return {"note": "matching = v_mfma_f32_32x32x64_f8f6f4 over +-1-encoded descriptor bits (2*K*K*bits flop per pair, "
                        "rows unpadded) + 1.25 VALU ops per table cell for the top-2 scan"
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
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--keyframes", type=int, default=10000, help="keyframes per robot (configs[1]: 10k)")
    ap.add_argument("--features", type=int, default=500)
    ap.add_argument("--desc-bytes", type=int, default=32)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=500)
    ap.add_argument("--true-frac", type=float, default=0.2)
    ap.add_argument("--nn-precision", type=int, default=1,
                    help="1 = fp16 MFMA filter with rigorous error band + exact f64 refinement (identical "
                         "matches); 0 = fp32 MFMA ranking of every column")
    ap.add_argument("--estimator", choices=("3d3d", "pnp"), default="3d3d",
                    help="motion estimator of both registration passes: 3d3d = RANSAC 3D->3D (north_star, "
                         "myRegistrationVis.cpp:1113-1152), pnp = RANSAC 3D->2D (:1055-1112, rtabmap's default)")
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
    args = ap.parse_args()

    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with RCCL's streams in the process
    # the library's second stream would share a queue with its first and lose the overlap it exists for
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            td.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from multi_robot_slam_separators_amd import _abi, dist, lib, synth

    n_kf, k, cols, dim = args.keyframes, args.features, args.desc_bytes, args.dim
    p = synth.camera_params()
    p.iterations = args.iterations
    p.estimation_type = 1 if args.estimator == "pnp" else 0
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf            # batch operation: walk every row
    p.nn_precision = args.nn_precision
    p.max_features = k
    p.desc_bytes = cols
    p.store_capacity = 2 * n_kf
    feats, nv_a, nv_b, t_gen = generate_inputs(12345 + rank, n_kf, k, cols, dim, args.true_frac)

    f = lib.SeparatorFinder(p, device=dev_index)
    f.set_stream(torch.cuda.current_stream().cuda_stream)

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
        exch = dist.RecordExchange(_abi.RESULT_DTYPE.itemsize, n_kf, n_kf // 4 + 256, coll_dev)
    h_flags = h_pack[flags_off: flags_off + n_kf].view(torch.bool)
    h_cnt = h_pack[:4].view(torch.int32)
    spec_cap = n_kf // 4 + 256
    OFF_SUCCESS = _abi.RESULT_DTYPE.fields["success"][1]
    state = {"pairs": 0, "accepted": 0, "last": None}

    trace = os.environ.get("BENCH_STEP_TRACE") is not None    # per-phase wall times of a step on stderr
    two_calls = os.environ.get("BENCH_TWO_CALLS") is not None

    def step():
        t_0 = time.perf_counter()
        # NN kernels, then -- in ONE library call -- the verification of the candidates: the NN filter's
        # candidates are verified speculatively on the device while the host reduces them to row minima, sorts
        # and walks them (data_handler.py:187-205); the walk's matches then pick their results (sf_api.hip).
        # BENCH_TWO_CALLS=1: the two separate calls (sf_nn_find_matches, then sf_verify_matches_device).
        if two_calls:
            m = f.nn_find_matches(cap=n_kf)                   # NN kernels + row minima to host + walk
            t_1 = time.perf_counter()
            f.verify_matches_device(m, slot_a, slot_b, d_res.data_ptr())
        else:
            m = f.find_matches_and_verify_device(slot_a, slot_b, d_res.data_ptr(), cap=n_kf)
            t_1 = time.perf_counter()
        n = len(m)
        # pair (from = querying robot A's keyframe idx_other, to = computing robot B's idx_local),
        # find_separators.py:85-91
        if trace:
            t_1b = t_2 = time.perf_counter()
            torch.cuda.synchronize()
            t_3 = time.perf_counter()
        # every candidate's success flag goes back to the two robots involved (failures feed the ignore
        # list, data_handler.py:406-408); only ACCEPTED separators are exchanged between GPUs / handed to
        # the back-end (data_handler.py:352-368).
        # The compaction leaves its count on the device; count, flags and a speculative prefix of the accepted
        # records (capacity for a 25 % acceptance rate) go to the host behind the step's ONE synchronisation.
        if exch is not None and coll_dev.type != "cuda":      # gloo rehearsal: the collective runs on CPU tensors
            n_acc = f.compact_accepted_device(d_res.data_ptr(), n, d_acc.data_ptr(), d_flags.data_ptr())
            acc = d_acc
            exch.payload[:n_acc].copy_(acc[:n_acc])
            exch.exchange(n_acc)
            d_cnt.fill_(n_acc)                                # (the packed copy below carries it to the host)
        else:
            if exch is not None:
                # compaction writes records AND count straight into the exchange's send buffer
                acc, cnt_ptr, d_cnt_view = exch.payload, exch.count_ptr, exch.send[0, :4].view(torch.int32)
            else:
                acc, cnt_ptr, d_cnt_view = d_acc, d_cnt.data_ptr(), d_cnt
            f.compact_accepted_device_async(d_res.data_ptr(), n, acc.data_ptr(), d_flags.data_ptr(), cnt_ptr)
            if exch is not None:
                exch.exchange(None, finish=False)             # ONE all-gather, in flight beside the copies below
        k_spec = min(n, spec_cap)
        if acc is d_acc:
            # count + flags + the speculative prefix of the accepted separators: one copy to the pinned mirror
            nb = acc_off + k_spec * RB
            h_pack[:nb].copy_(d_pack[:nb], non_blocking=True)
        else:
            if exch is not None and coll_dev.type == "cuda":
                h_cnt.copy_(d_cnt_view, non_blocking=True)
            h_flags[:n].copy_(d_flags[:n], non_blocking=True)
            h_res[:k_spec].copy_(acc[:k_spec], non_blocking=True)   # accepted separators delivered to the host (pinned)
        if exch is not None:
            exch.finish()                                     # the copies above ran beside the collective
        torch.cuda.synchronize()
        n_acc = int(h_cnt[0])
        if n_acc > k_spec:                                    # more accepted than the speculative prefix held
            h_res[k_spec:n_acc].copy_(acc[k_spec:n_acc], non_blocking=True)
            torch.cuda.synchronize()
        host = h_res[:n_acc]
        if trace:
            t_4 = time.perf_counter()
            pr = f.prof_get()
            kv = sum(v[1] for kname, v in pr.items() if kname.startswith("k_verify") or kname.startswith("k_match")
                     or kname.startswith("k_ransac") or kname.startswith("k_guided"))
            print("[bench step] nn (+ verify launch) %.3f ms, - %.3f, - %.3f, verify wait %.3f, accepted-only "
                  "gather + copies %.3f; verification kernels so far %.3f ms (hipEvents)"
                  % ((t_1 - t_0) * 1e3, (t_1b - t_1) * 1e3, (t_2 - t_1b) * 1e3, (t_3 - t_2) * 1e3, (t_4 - t_3) * 1e3, kv),
                  file=sys.stderr)
        state["pairs"] += n
        state["last"] = (m, host, n)
        state["gathered"] = sum(exch.counts()) if exch is not None else n_acc
        return n

    for _ in range(args.warmup):
        step()
    f.prof_reset()
    f.prof_enable(True)
    state["pairs"] = 0
    if dist_on:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist_on:
        td.barrier()
    elapsed = time.perf_counter() - t0
    prof = f.prof_get()
    f.prof_enable(False)
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
                             "informational, `value` runs its steps strictly in sequence"}
        except Exception as e:   # informational only: never let it take the headline line down
            piped = {"error": repr(e)}
        f_nn.close()
    del ta, tb

    # ---- sanity of the timed work (rank 0): the separators found are the planted revisits -----------
    m, host, n = state["last"]
    flags = h_flags[:n].numpy().copy()
    truth = feats["is_true"][m["idx_local"]]
    same = m["idx_local"] == m["idx_other"]
    accepted = int(flags.sum())
    correct = int((flags == (truth & same)).sum())
    sep = np.frombuffer(host.numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
    if exch is not None:      # outside the timed region: every gathered record is an accepted separator
        allrec, cts = exch.all_gathered()
        gat = np.frombuffer(allrec.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
        mine = gat[sum(cts[:rank]): sum(cts[:rank + 1])]
        all_ok = (bool(gat["success"].all()) and len(gat) == state["gathered"] and len(mine) == len(sep)
                  and mine.tobytes() == sep.tobytes())
    else:
        all_ok = bool(sep["success"].all()) and len(sep) == state["gathered"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        pairs_per_step = total_pairs / args.steps / world
        bpp = bytes_per_pair(k, cols)
        # dominant kernel: the fused per-pair pipeline (matching + both RANSAC passes + guided matching +
        # result: the whole verification the 44 352 B/pair figure of SURVEY 8(d) describes), or the
        # matching kernel when the stage kernels run (PnP estimator, SF_FUSED=0)
        dom = "k_verify_fused" if prof.get("k_verify_fused", (0, 0.0))[0] > 0 else "k_match_global"
        nm, tm = prof[dom]
        match_ms = tm / max(nm, 1)
        # pairs one launch of the dominant kernel processes (big batches are cut in two halves on two streams)
        pairs_per_launch = pairs_per_step * args.steps / max(nm, 1)
        pmc = pmc_traffic(dom, pairs_per_launch)
        ach = pairs_per_launch * bpp / (match_ms * 1e-3) / 1e9 if match_ms > 0 else 0.0
        nn_kernel, nn_peak = (("k_nn_filter_f16", MFMA_F16_PEAK_TF) if args.nn_precision == 1
                              else ("k_nn_argmin", MFMA_F32_PEAK_TF))
        nn_n, nn_t = prof[nn_kernel]
        nn_ms = nn_t / max(nn_n, 1)
        # flops the launched kernel really performs: the fp16 filter contracts a PREFIX of the descriptor
        # (k_nn.hip, nn_run_filter: adaptive 128 / 512 / full), the fp32 ranking kernel all D dimensions
        k_eff = dim if args.nn_precision == 0 else (filter_dims or dim)
        nn_tf = 2.0 * n_kf * n_kf * k_eff / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0
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
                "workload": "BASELINE configs[1]: 1xMI355X per rank, 2 robots x %d keyframes, %d-D fp32 NetVLAD, "
                            "%d x %d-bit ORB per keyframe, %d RANSAC iterations (%s), both registration passes, "
                            "%.0f %% true revisits" % (n_kf, dim, k, cols * 8, args.iterations,
                                                       "3D-3D" if args.estimator == "3d3d" else "PnP",
                                                       100 * args.true_frac),
                "pairs_per_step_per_gpu": pairs_per_step,
                "parallelism": "pairs sharded by robot pair, 1 rank per GPU" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": (pmc or {}).get("bytes"),
                "traffic_source": pmc,
                "compute": compute_note(dom, k, cols, pairs_per_launch, match_ms),
                "bytes_per_pair": bpp, "pairs_per_launch": pairs_per_launch, "avg_launch_ms": match_ms,
                "launches_per_step": nm / args.steps,
            },
            "roofline_nn": {
                "kernel": nn_kernel, "bound": "mfma", "achieved": nn_tf, "peak": nn_peak,
                "unit": "TFLOP/s", "frac": nn_tf / nn_peak, "avg_launch_ms": nn_ms, "contracted_dims": k_eff,
            },
            "kernel_ms_per_step": {(kname.replace("k_ransac", "k_pnp") if args.estimator == "pnp" else kname):
                                   (ms / args.steps) for kname, (cnt, ms) in prof.items()},
            "check": {"accepted_last_step": accepted, "decisions_matching_ground_truth": correct, "of": int(n),
                      "accepted_separators_gathered_per_step": state.get("gathered", 0),
                      "gathered_records_all_accepted": all_ok},
            "input_generation_s": t_gen,
        }
        if piped is not None:
            out["pipelined_two_streams"] = piped
        if alt_valu is not None:
            # the Hamming table by xor + popcount on the VALU instead of the fp4 matrix cores (identical outputs)
            out["value_with_valu_matcher"] = alt_valu * world
        if alt is not None:
            out["value_with_fp32_nn_ranking"] = alt * world
            out["check"]["nn_matches_identical_fp32_vs_f16filter"] = bool(
                np.array_equal(alt_m["idx_local"], m["idx_local"]) and np.array_equal(alt_m["idx_other"], m["idx_other"])
                and np.array_equal(alt_m["distance"], m["distance"]))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, feats, nv_a, nv_b, n_kf, args.cpu_sample_pairs,
                                               args.cpu_sample_rows)
        print(json.dumps(out))
    f.close()
    if dist_on:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
