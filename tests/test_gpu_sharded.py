"""SURVEY.md section 8(e) on the GPU: sf_nn_row_minima_device (the NN kernels of a block of local rows, minima left in
device memory, no walk) against the minima sf_nn_find_matches reports; sharded.ShardedStep over GpuShardBackend -- the
code `bench.py --partition 8e` runs -- against the single-GPU calls; and that bench line itself with a world-1 RCCL
process group (every collective of the step is issued through RCCL)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from multi_robot_slam_separators_amd import _abi, lib, sharded, synth

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
RB = _abi.RESULT_DTYPE.itemsize
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _descriptors(seed, n, dim):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, dim)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + 0.002 * rng.normal(size=(n, dim)); b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[10] = b[11] = b[12]                      # three local rows whose nearest received column is 12
    b[40:44] = rng.normal(size=(4, dim)) * 3.0  # rows with no candidate under the threshold
    return a, b


@pytest.mark.parametrize("precision,dim", [(1, 512), (1, 128), (0, 512)])
def test_row_minima_device_equals_find_matches(precision, dim):
    n = 300
    a, b = _descriptors(5 + dim, n, dim)
    p = synth.camera_params()
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n
    p.nn_precision = precision
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.nn_append_received(a)
        f.nn_append_local(b)
        f.nn_mark_local_used(3); f.nn_mark_other_used(5); f.nn_ignore_pair(20, 20); f.nn_ignore_pair(21, 22)
        f.nn_find_matches(cap=n)
        d_ref, a_ref = f.nn_last_row_minima()
        dmin = torch.full((n,), -1.0, dtype=torch.float64, device=DEV)
        darg = torch.full((n,), -1, dtype=torch.int32, device=DEV)
        st = torch.full((1,), 7, dtype=torch.int32, device=DEV)
        for _ in range(2):                     # the second call finds the coefficient / fp16 caches warm
            f.nn_row_minima_device(dmin.data_ptr(), darg.data_ptr(), st.data_ptr())
            torch.cuda.synchronize()
            assert int(st.item()) == 0
            assert dmin.cpu().numpy().tobytes() == d_ref.tobytes()
            assert np.array_equal(darg.cpu().numpy(), a_ref)
        if precision == 1:
            assert np.isinf(d_ref[40:44]).all() and np.isinf(d_ref[3])


def test_row_minima_device_reports_a_dense_candidate_set():
    """Every pair under the threshold: the prefix level's sparse limit (8 n + 4096 candidates) is exceeded, the status
    word says so, and the synchronous path still answers."""
    n, dim = 1024, 512
    rng = np.random.default_rng(3)
    base = rng.normal(size=dim); base /= np.linalg.norm(base)
    a = base + 1e-3 * rng.normal(size=(n, dim)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    p = synth.camera_params()
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n
    p.nn_precision = 1
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.nn_append_received(a)
        f.nn_append_local(a[::-1].copy())
        dmin = torch.zeros(n, dtype=torch.float64, device=DEV)
        darg = torch.zeros(n, dtype=torch.int32, device=DEV)
        st = torch.zeros(1, dtype=torch.int32, device=DEV)
        f.nn_row_minima_device(dmin.data_ptr(), darg.data_ptr(), st.data_ptr())
        torch.cuda.synchronize()
        assert int(st.item()) == 1
        m = f.nn_find_matches(cap=n)
        d, arg = f.nn_last_row_minima()
        assert len(m) > 0 and np.isfinite(d).all()


def _store(f, feats, n_kf, k):
    def up(x):
        x = np.ascontiguousarray(x)
        return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(DEV)
    T = {key: up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}
    sa = f.store_add_keyframes_device(n_kf, k, 32, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
    sb = f.store_add_keyframes_device(n_kf, k, 32, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
    torch.cuda.synchronize()
    return sa, sb, T


def test_sharded_step_on_one_gpu_equals_the_separate_calls():
    """One rank, no process group (the collectives are copies): candidate list, flags and accepted records of
    ShardedStep equal sf_nn_find_matches + sf_verify_matches_device; the step waits ONCE (device walk)."""
    n_kf, k, dim = 96, 200, 512
    feats = synth.make_store_batch(31, n_kf, k=k, cols=32, true_frac=0.5)
    a, b = _descriptors(32, n_kf, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        sa, sb, keep = _store(f, feats, n_kf, k)
        f.nn_append_received(a)
        f.nn_append_local(b)
        m_ref = f.nn_find_matches(cap=n_kf)
        d = torch.zeros((len(m_ref), RB), dtype=torch.uint8, device=DEV)
        f.verify_matches_device(m_ref, sa, sb, d.data_ptr())
        torch.cuda.synchronize()
        res = np.frombuffer(d.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
        ok = res["success"].astype(bool)
        assert 5 <= ok.sum() < len(m_ref)
        be = sharded.GpuShardBackend(f, 0, n_kf, sa, sb, n_kf, DEV, 1)
        st = sharded.ShardedStep(be, 0, 1, n_kf, DEV, accept_cap=4)     # small block: the overflow path runs as well
        for _ in range(2):
            m, flags, acc = st.step()
            assert st.waits == 1
            assert m.tobytes() == m_ref.tobytes()
            assert np.array_equal(flags, ok)
            assert acc.tobytes() == res[ok].tobytes()
        st2 = sharded.ShardedStep(be, 0, 1, n_kf, DEV)
        m, flags, acc = st2.step()
        assert m.tobytes() == m_ref.tobytes() and np.array_equal(flags, ok) and acc.tobytes() == res[ok].tobytes()


def test_partition_8e_bench_line_over_rccl_world_1():
    """`bench.py --partition 8e` with a world-1 RCCL process group (BENCH_FORCE_DIST=1): the minima block and the
    record block go through ncclAllGather; every decision matches the ground truth and a step waits once."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--partition", "8e", "--keyframes", "1500",
                          "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["scaling"] == "strong"
    assert j["check"]["decisions_matching_ground_truth"] == j["check"]["of"] > 1000
    assert j["check"]["host_waits_per_robot_pair_step"] == 1


@pytest.mark.parametrize("form", ["two_buffers", "one_buffer"])
def test_robot_pairs_bench_line_over_rccl_world_1(form):
    """The default N > 1 path of `bench.py` (every rank its own robot pair, accepted separators all-gathered) with a
    world-1 RCCL process group: the steps run as at N = 1 (same ring, same streams) and every RETIRED step's separators
    go from the step block's device copy (sf_step_result.d_records) into the send buffer of one of two alternating
    exchanges (or of one); every gathered record is an accepted separator of the last step and every decision matches
    the ground truth."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534",
               BENCH_SELF_WARMUP_MIN="7", BENCH_SELF_WARMUP_MAX="7")       # (an odd count: the parity must not matter)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_ONE_EXCHANGE_BUFFER"):
        env.pop(k, None)
    if form == "one_buffer":
        env["BENCH_ONE_EXCHANGE_BUFFER"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--keyframes", "1500", "--steps", "9",
                          "--warmup", "2", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["scaling"] == "weak"
    assert j["check"]["decisions_matching_ground_truth"] == j["check"]["of"] > 1000
    assert j["check"]["gathered_records_all_accepted"] is True
    assert j["check"]["accepted_separators_gathered_per_step"] >= j["check"]["accepted_last_step"] > 100
    assert j["exchange_buffers"] == (1 if form == "one_buffer" else 2)
    assert j["steps_on_several_streams"] is True
