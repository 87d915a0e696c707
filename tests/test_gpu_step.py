"""sf_step_issue / sf_step_retire (include/sepfinder.h): the loop body of the reference's caller
(PKG/scripts/find_separators.py:59-133 -- s_find_matches_query, one s_ans_est_transform per returned candidate, then the
per-candidate outcome) as a begin / retire pair, against the separate calls it replaces (sf_nn_find_matches +
sf_verify_matches_device), byte for byte."""
import numpy as np
import pytest
import torch

from multi_robot_slam_separators_amd import _abi, lib, synth

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
RB = _abi.RESULT_DTYPE.itemsize


def _world(seed, n_kf, k, dim):
    feats = synth.make_store_batch(seed, n_kf, k=k, cols=32, true_frac=0.5)
    rng = np.random.default_rng(seed + 1)
    nv_a = rng.normal(size=(n_kf, dim)); nv_a /= np.linalg.norm(nv_a, axis=1, keepdims=True)
    nv_b = nv_a + 0.002 * rng.normal(size=(n_kf, dim)); nv_b /= np.linalg.norm(nv_b, axis=1, keepdims=True)
    nv_b[10] = nv_b[11] = nv_b[12]                      # three local rows whose nearest received column is 12
    nv_b[40:44] = rng.normal(size=(4, dim)) * 3.0       # rows with no candidate under the threshold
    return feats, nv_a, nv_b


def _up(x):
    x = np.ascontiguousarray(x)
    return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(DEV)


def _fill(f, feats, nv_a, nv_b, n_kf, k):
    T = {key: _up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}
    sa = f.store_add_keyframes_device(n_kf, k, 32, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
    sb = f.store_add_keyframes_device(n_kf, k, 32, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
    torch.cuda.synchronize()
    f.nn_append_received(nv_a)
    f.nn_append_local(nv_b)
    return sa, sb, T


def _two_calls(f, sa, sb, cap):
    m = f.nn_find_matches(cap=cap)
    d = torch.zeros((max(len(m), 1), RB), dtype=torch.uint8, device=DEV)
    f.verify_matches_device(m, sa, sb, d.data_ptr())
    torch.cuda.synchronize()
    return m, np.frombuffer(d[: len(m)].cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE).copy()


def _check_step(out, m_ref, res_ref, want_streamed):
    m, rom, recs, info = out
    assert info["n_matches"] == len(m_ref) and m.tobytes() == m_ref.tobytes()
    assert info["streamed"] == want_streamed
    ok = res_ref["success"].astype(bool)
    assert np.array_equal(rom >= 0, ok) and info["n_accepted"] == int(ok.sum())
    assert info["n_records"] >= info["n_accepted"] and len(recs) == info["n_records"]
    used = rom[rom >= 0]
    assert len(set(used.tolist())) == len(used) and (used < info["n_records"]).all()
    for i in np.nonzero(ok)[0]:
        assert recs[rom[i]].tobytes() == res_ref[i].tobytes(), "match %d" % i
    assert recs["success"].all()          # only accepted results are ever delivered


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("est", [0, 1])
def test_step_pair_equals_the_separate_calls(est, overlap):
    """Batch mode (the walk may return every local row: speculative verification, separators streamed out of the kernel)
    and the reference's cadence (netvlad_max_matches_nb = 20: no speculation, ordered compaction), both estimators, with
    masked rows / columns and ignored pairs; steps overlapped two deep as a host would run them."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(177 + est, n_kf, k, dim)
    for max_nb, want_streamed in ((n_kf, True), (20, False)):
        p = synth.camera_params()
        p.iterations = 200
        p.estimation_type = est
        p.netvlad_dimensions = dim
        p.netvlad_max_matches_nb = max_nb
        p.max_features = k
        with lib.SeparatorFinder(p) as f:
            f.set_stream(torch.cuda.current_stream().cuda_stream)
            f.set_option(_abi.SF_OPT_STEP_OVERLAP, overlap)       # the two steps in flight on two streams / on one
            sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
            f.nn_mark_local_used(3); f.nn_mark_other_used(5); f.nn_ignore_pair(20, 20); f.nn_ignore_pair(21, 22)
            m_ref, res_ref = _two_calls(f, sa, sb, max_nb)
            assert len(m_ref) == min(max_nb, len(m_ref)) and res_ref["success"].sum() >= 3
            # one step at a time
            f.step_issue(sa, sb)
            _check_step(f.step_retire(copy=True), m_ref, res_ref, want_streamed)
            # two in flight, retired oldest first; the third issue is refused until one is retired
            f.step_issue(sa, sb)
            f.step_issue(sa, sb)
            with pytest.raises(lib.SepfinderError):
                f.step_issue(sa, sb)
            for _ in range(6):
                _check_step(f.step_retire(copy=True), m_ref, res_ref, want_streamed)
                f.step_issue(sa, sb)
            # the state the caller feeds back between ticks (data_handler.py:402-408) takes effect in the next step
            first = f.step_retire(copy=True)
            _check_step(first, m_ref, res_ref, want_streamed)
            _check_step(f.step_retire(copy=True), m_ref, res_ref, want_streamed)
            with pytest.raises(lib.SepfinderError):
                f.step_retire()
            row, col = int(m_ref["idx_local"][0]), int(m_ref["idx_other"][0])
            f.nn_mark_local_used(row); f.nn_mark_other_used(col)
            m2, res2 = _two_calls(f, sa, sb, max_nb)
            assert row not in m2["idx_local"].tolist() and col not in m2["idx_other"].tolist()
            f.step_issue(sa, sb)
            _check_step(f.step_retire(copy=True), m2, res2, want_streamed)


@pytest.mark.parametrize("form", ["forced_two_streams", "forced_one_stream", "auto", "auto_one_stream", "off"])
def test_step_pair_in_the_split_form(form, monkeypatch):
    """The 3D-3D verification as one matching launch + one chain launch over the survivors (k_match_split + k_chain)
    instead of the fused kernel: everywhere with SF_FUSED=2 (read at sf_create); by the library's own choice inside
    overlapped steps (SF_OPT_STEP_SPLIT, default on) -- not when the steps share one stream, not with the option off.
    The chain kernel streams the accepted separators like the fused one does, and a step's matches, flags and records
    are the separate calls', byte for byte, in every form."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(577, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    monkeypatch.delenv("SF_FUSED", raising=False)
    monkeypatch.delenv("SF_STEP_SPLIT", raising=False)
    monkeypatch.delenv("SF_STEP_SPLIT_MIN", raising=False)
    with lib.SeparatorFinder(p) as f:              # the reference results: the fused kernel through the separate calls
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m_ref, res_ref = _two_calls(f, sa, sb, n_kf)
    assert res_ref["success"].sum() >= 3
    if form.startswith("forced"):
        monkeypatch.setenv("SF_FUSED", "2")
    monkeypatch.setenv("SF_STEP_SPLIT_MIN", "1")       # (by default only queries of >= 2048 candidates switch form)
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.set_option(_abi.SF_OPT_STEP_OVERLAP, 0 if form.endswith("one_stream") else 1)
        if form == "off":
            f.set_option(_abi.SF_OPT_STEP_SPLIT, 0)
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m2, res2 = _two_calls(f, sa, sb, n_kf)     # the separate calls in this handle's form: the same bytes
        assert m2.tobytes() == m_ref.tobytes() and res2.tobytes() == res_ref.tobytes()
        f.prof_reset()
        f.prof_enable(True)
        f.step_issue(sa, sb)
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        f.step_issue(sa, sb)
        for _ in range(4):
            f.step_issue(sa, sb)
            _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        torch.cuda.synchronize()
        prof = f.prof_get()
        f.prof_enable(False)
        split_ran = prof["k_match_global"][0] > 0          # (the matching launch's profiling slot)
        assert split_ran == (form in ("forced_two_streams", "forced_one_stream", "auto")), (form, prof)
        assert prof["k_verify_fused"][0] == 6              # the fused kernel, or the chain launch in its slot


def test_step_with_a_mirror_for_the_exchange():
    """sf_step_mirror: every accepted record also lands in a device buffer (an all-gather's send buffer) and the slot
    counter is the caller's device word; a mirror too small for the query's verified candidates switches the query to
    the compaction -- nothing is dropped."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(277, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m_ref, res_ref = _two_calls(f, sa, sb, n_kf)
        n_acc = int(res_ref["success"].sum())
        for cap, want_streamed in ((n_kf + n_kf // 8 + 256, True), (n_kf, False)):
            send = torch.zeros((cap + 1, RB), dtype=torch.uint8, device=DEV)       # slot 0 = header (count)
            f.step_mirror(send[1:].data_ptr(), send.data_ptr(), cap)
            for rep in range(3):
                send[0].zero_()
                f.step_issue(sa, sb)
                out = f.step_retire(copy=True)
                _check_step(out, m_ref, res_ref, want_streamed)
                torch.cuda.synchronize()
                cnt = int(send[0, :4].view(torch.int32).item())
                assert cnt == out[3]["n_records"] >= n_acc
                mirror = np.frombuffer(send[1: 1 + cnt].cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
                assert mirror.tobytes() == out[2][:cnt].tobytes()
            f.step_mirror(None, None, 0)
        f.step_issue(sa, sb)
        with pytest.raises(lib.SepfinderError):
            f.step_mirror(None, None, 0)            # not while a step is in flight
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)


def test_step_with_two_alternating_mirrors():
    """sf_step_mirror_pair: steps alternate between two device destinations (the first one issued after the call
    writes the even pair), also with two steps in flight -- each buffer holds exactly its own step's records, and a
    buffer is not touched by the step that uses the other one."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(281, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m_ref, res_ref = _two_calls(f, sa, sb, n_kf)
        n_acc = int(res_ref["success"].sum())
        assert n_acc > 0
        cap = n_kf + n_kf // 8 + 256
        send = [torch.zeros((cap + 1, RB), dtype=torch.uint8, device=DEV) for _ in range(2)]
        f.step_issue(sa, sb)                        # an odd number of steps before the call: the call restarts the parity
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        f.step_mirror_pair((send[0][1:].data_ptr(), send[0].data_ptr()), (send[1][1:].data_ptr(), send[1].data_ptr()), cap)

        def check_buffer(b, out):
            cnt = int(send[b][0, :4].view(torch.int32).item())
            assert cnt == out[3]["n_records"] >= n_acc
            got = np.frombuffer(send[b][1: 1 + cnt].cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
            assert got.tobytes() == out[2][:cnt].tobytes()

        # one step at a time: even, odd, even; the other buffer stays as it was (all 0xEE)
        for step in range(3):
            b = step & 1
            send[b].zero_()
            send[b ^ 1].fill_(0xEE)
            f.step_issue(sa, sb)
            out = f.step_retire(copy=True)
            _check_step(out, m_ref, res_ref, True)
            torch.cuda.synchronize()
            check_buffer(b, out)
            assert bool((send[b ^ 1] == 0xEE).all())
        # two in flight (the parity continues: this is step 3 -> odd)
        send[1].zero_()
        f.step_issue(sa, sb)
        send[0].zero_()
        f.step_issue(sa, sb)
        out_odd = f.step_retire(copy=True)
        out_even = f.step_retire(copy=True)
        _check_step(out_odd, m_ref, res_ref, True)
        _check_step(out_even, m_ref, res_ref, True)
        torch.cuda.synchronize()
        check_buffer(1, out_odd)
        check_buffer(0, out_even)
        with pytest.raises(lib.SepfinderError):
            f.step_mirror_pair((send[0][1:].data_ptr(), send[0].data_ptr()), (0, 0), cap)     # both pairs or none
        # sf_step_mirror_streams: the odd steps on the handle's second stream; the caller's fills go to the same streams
        f.step_mirror_pair((send[0][1:].data_ptr(), send[0].data_ptr()), (send[1][1:].data_ptr(), send[1].data_ptr()), cap)
        s_even, s_odd = f.step_mirror_streams()
        assert s_odd != s_even                     # (SF_OPT_STEP_OVERLAP is the default)
        lanes = [torch.cuda.current_stream(), torch.cuda.ExternalStream(s_odd)]
        outs = []
        for step in range(6):
            b = step & 1
            with torch.cuda.stream(lanes[b]):
                send[b].zero_()
                f.step_issue(sa, sb)
            if step:
                outs.append(f.step_retire(copy=True))      # the step before: two are in flight at this point
        outs.append(f.step_retire(copy=True))
        assert len(outs) == 6
        for out in outs:
            _check_step(out, m_ref, res_ref, True)
        torch.cuda.synchronize()
        check_buffer(0, outs[-2])                  # step 4 (even) and step 5 (odd): the last writers of the two buffers
        check_buffer(1, outs[-1])
        with pytest.raises(lib.SepfinderError):
            f.step_mirror(send[0][1:].data_ptr(), send[0].data_ptr(), cap) or f.step_mirror_streams()   # one mirror: no lanes
        f.step_mirror(None, None, 0)
        f.step_issue(sa, sb)
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)


def test_step_on_an_empty_candidate_list_and_an_empty_database():
    n_kf, k, dim = 32, 64, 128
    feats, nv_a, nv_b = _world(377, 48, k, dim)
    p = synth.camera_params()
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = 48
    p.netvlad_distance = 1e-6               # nothing is under the threshold
    p.max_features = k
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        with pytest.raises(lib.SepfinderError):
            f.step_issue(0, 0)              # data_handler.py:308 guards the empty database
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, 48, k)
        f.step_issue(sa, sb)
        m, rom, recs, info = f.step_retire()
        assert info["n_matches"] == 0 and info["n_accepted"] == 0 and len(m) == 0 and len(rom) == 0


def test_overlapped_steps_and_a_growing_store():
    """SF_OPT_STEP_OVERLAP: keyframes appended (the store re-allocated) and descriptors appended while two steps are in
    flight on two streams; the steps in flight and the steps after see consistent databases."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(477, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = 4 * n_kf
    p.max_features = k
    p.store_capacity = 2 * n_kf                       # exactly full after _fill: the next keyframe re-allocates
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.set_option(_abi.SF_OPT_STEP_OVERLAP, 1)
        sa, sb, T = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m_ref, res_ref = _two_calls(f, sa, sb, 4 * n_kf)
        f.step_issue(sa, sb)
        f.step_issue(sa, sb)                          # on the second stream
        extra = f.store_add_keyframes_device(8, k, 32, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
        assert extra == 2 * n_kf
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        for _ in range(4):                            # steps after the re-allocation, both lanes
            f.step_issue(sa, sb)
            f.step_issue(sa, sb)
            _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
            _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        # new received descriptors (written through the handle's stream) are seen by a step on the second stream
        f.step_issue(sa, sb)                          # lane 0
        more = nv_a[:5] + 1e-4
        f.nn_append_received(more / np.linalg.norm(more, axis=1, keepdims=True))
        m2, res2 = _two_calls(f, sa, sb, 4 * n_kf)
        f.step_issue(sa, sb)                          # lane 1
        _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        _check_step(f.step_retire(copy=True), m2, res2, True)
