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

from bench_extras import (ROOT, METRIC, HBM_PEAK_GBS, MFMA_F32_PEAK_TF, MFMA_F16_PEAK_TF, MFMA_FP4_PEAK_TF, bytes_per_pair,
                          pmc_traffic, sq_evidence, sq_counters, compute_note, matrix_pipe_roofline, generate_inputs, measure_next_rows, cpu_baseline,
                          upload_store, set_estimator, estimator_text, run_partition_8e, run_cfg4, run_cfg3,
                          run_untimed_comparisons)


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
            # (two lanes, ONE step in flight: the library keeps the form it uses inside overlapped steps -- the split one for
            #  3D-3D -- and nothing runs beside the step's own kernels, which depend on each other)
            f.set_option(_abi.SF_OPT_STEP_LANES, 2)
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

    # ---- the untimed comparison runs (bench_extras.py): other NN precision, one synchronisation per step, full-length
    # filter, fixed iterations, the strict form, VALU matcher, PCIe-inclusive rate, NN stage alone, the next rows ----
    import types
    X = run_untimed_comparisons(types.SimpleNamespace(
        _abi=_abi, args=args, cols=cols, d_acc=d_acc, d_flags=d_flags, d_res=d_res, depth=depth, dev=dev, dev_index=dev_index, dim=dim, dist_on=dist_on, f=f, feats=feats, k=k, last_timed=last_timed, lib=lib, n_kf=n_kf, nn_append_local=nn_append_local, nn_append_received=nn_append_received, p=p, pairs_timed=pairs_timed, pipelined=pipelined, rank=rank, slot_a=slot_a, slot_b=slot_b, state=state, step=step, synth=synth, ta=ta, tb=tb, torch=torch, world=world))
    alt, alt_fixed, alt_full, alt_m, alt_strict, alt_sync, alt_valu, next_rows, nn_only, pcie, piped = (X["alt"], X["alt_fixed"], X["alt_full"], X["alt_m"], X["alt_strict"], X["alt_sync"], X["alt_valu"], X["next_rows"], X["nn_only"], X["pcie"], X["piped"])
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
        if (args.estimator == "3d3d" and dom_name in ("k_match_split", "k_verify_fused")
                and os.environ.get("SF_MATCH_MFMA", "1") != "0"):
            # the matching kernel is bound by the fp4 matrix pipe (75 % busy at K = 1000, 58 % at K = 500 alone on the chip,
            # profiles/r05s_*, r05w_*): that is the roofline it is priced against; SURVEY 8(d)'s HBM basis stays beside it
            out["roofline"] = matrix_pipe_roofline(out["roofline"], k, cols)
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
                "whole_step": (out["roofline_matching_kernel"].get("whole_step")
                               or out["roofline_matching_kernel"]["hbm_basis"]["whole_step"]),
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
                               "whole_step": (out["roofline_verification_kernel"].get("whole_step")
                                              or out["roofline_verification_kernel"]["hbm_basis"]["whole_step"])}
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
