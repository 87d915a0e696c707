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


@pytest.mark.parametrize("walk", [1, 2, 0])
@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("est", [0, 1])
def test_step_pair_equals_the_separate_calls(est, overlap, walk):
    """Batch mode (the walk may return every local row) and the reference's cadence (netvlad_max_matches_nb = 20), both
    estimators, with masked rows / columns and ignored pairs; steps overlapped two deep as a host would run them.
    walk = 1: the device-resident step (no host wait in sf_step_issue; batch mode verifies every filter candidate beside
    the walk on a second stream -- SF_OPT_STEP_SPECULATE --, the reference's cadence takes the walk's matches from the
    device list; the accepted separators stream in both modes); walk = 2: the same with SF_OPT_STEP_SPECULATE off (NN ->
    walk -> verification on one stream in both modes); walk = 0: round 3's form (host walk inside
    sf_step_issue; speculative verification in batch mode, ordered compaction at the reference's cadence)."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(177 + est, n_kf, k, dim)
    for max_nb, want_streamed in ((n_kf, True), (20, bool(walk))):
        p = synth.camera_params()
        p.iterations = 200
        p.estimation_type = est
        p.netvlad_dimensions = dim
        p.netvlad_max_matches_nb = max_nb
        p.max_features = k
        with lib.SeparatorFinder(p) as f:
            f.set_stream(torch.cuda.current_stream().cuda_stream)
            f.set_option(_abi.SF_OPT_STEP_OVERLAP, overlap)       # the two steps in flight on two streams / on one
            f.set_option(_abi.SF_OPT_STEP_DEVICE_WALK, int(walk != 0))
            f.set_option(_abi.SF_OPT_STEP_SPECULATE, int(walk != 2))
            f.set_option(_abi.SF_OPT_STEP_DEPTH, 2)
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


@pytest.mark.parametrize("form", ["forced_lanes", "forced_one_stream", "opt_in", "opt_in_one_stream", "default", "opt_out"])
def test_step_pair_in_the_split_form(form, monkeypatch):
    """The 3D-3D verification as one matching launch + one chain launch over the survivors (k_match_split + k_chain)
    instead of the fused kernel: everywhere with SF_FUSED=2 (read at sf_create); with SF_OPT_STEP_SPLIT on, by the
    library's own choice inside steps that are dealt over several streams -- not when the steps share one stream.  The
    option is on by default (round 5: the matching launch's scan is software-pipelined); 0 keeps the fused kernel.  The
    chain kernel streams the accepted
    separators like the fused one does, and a step's matches, flags and records are the separate calls', byte for byte,
    in every form."""
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
        if form.startswith("opt_in"):
            f.set_option(_abi.SF_OPT_STEP_SPLIT, 1)
        if form == "opt_out":
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
        assert split_ran == (form in ("forced_lanes", "forced_one_stream", "opt_in", "default")), (form, prof)
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
        assert n_acc > 8
        # a mirror with a slot for every pair slot of the query: streamed; smaller than the query (also smaller than the
        # accepted separators: 8 slots): the compaction, whose mirror writes stop at the mirror's capacity -- the host
        # block still receives every record and the mirror's count word carries the full count (the exchange's overflow
        # signal); round 3 wrote past the mirror here
        for cap, want_streamed in ((n_kf + n_kf // 8 + 256, True), (n_kf, True), (8, False)):
            send = torch.zeros((cap + 1 + 64, RB), dtype=torch.uint8, device=DEV)       # slot 0 = header (count)
            send[1 + cap:].fill_(0xEE)                                             # guard rows behind the mirror
            f.step_mirror(send[1:].data_ptr(), send.data_ptr(), cap)
            for rep in range(3):
                send[0].zero_()
                f.step_issue(sa, sb)
                out = f.step_retire(copy=True)
                _check_step(out, m_ref, res_ref, want_streamed)
                torch.cuda.synchronize()
                cnt = int(send[0, :4].view(torch.int32).item())
                assert cnt == out[3]["n_records"] >= n_acc
                held = min(cnt, cap)
                mirror = np.frombuffer(send[1: 1 + held].cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
                assert mirror.tobytes() == out[2][:held].tobytes()
                assert bool((send[1 + cap:] == 0xEE).all()), "a write past the mirror's capacity"
            f.step_mirror(None, None, 0)
        # round 3's form of the step (host walk, speculative verification of n_kf * 9 / 8 + 256 slots): a mirror of n_kf
        # slots is smaller than that query -> the compaction
        f.set_option(_abi.SF_OPT_STEP_DEVICE_WALK, 0)
        send = torch.zeros((n_kf + 1, RB), dtype=torch.uint8, device=DEV)
        f.step_mirror(send[1:].data_ptr(), send.data_ptr(), n_kf)
        send[0].zero_()
        f.step_issue(sa, sb)
        out = f.step_retire(copy=True)
        _check_step(out, m_ref, res_ref, False)
        torch.cuda.synchronize()
        cnt = int(send[0, :4].view(torch.int32).item())
        assert cnt == out[3]["n_records"] == n_acc
        assert np.frombuffer(send[1: 1 + cnt].cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE).tobytes() == out[2][:cnt].tobytes()
        f.step_mirror(None, None, 0)
        f.set_option(_abi.SF_OPT_STEP_DEVICE_WALK, 1)
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
        # the ring's default depth (6) must not apply with a mirror set: a THIRD step in flight would zero and overwrite the
        # buffer of the oldest one before its collective has read it -- the library refuses it (round 4 admitted it)
        send[1].zero_()
        f.step_issue(sa, sb)
        send[0].zero_()
        f.step_issue(sa, sb)
        with pytest.raises(lib.SepfinderError, match="2-buffer mirror"):
            f.step_issue(sa, sb)
        out_odd = f.step_retire(copy=True)
        send[1].zero_()
        f.step_issue(sa, sb)                      # (one retired: one may enter)
        out_even = f.step_retire(copy=True)
        out_odd2 = f.step_retire(copy=True)
        for out in (out_odd, out_even, out_odd2):
            _check_step(out, m_ref, res_ref, True)
        torch.cuda.synchronize()
        check_buffer(0, out_even)
        check_buffer(1, out_odd2)
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


@pytest.mark.parametrize("speculate", [1, 0])
@pytest.mark.parametrize("depth,lanes", [(1, 1), (4, 2), (4, 3), (8, 4), (16, 2)])
def test_step_ring_depth_and_lanes(depth, lanes, speculate):
    """SF_OPT_STEP_DEPTH steps in flight dealt over SF_OPT_STEP_LANES streams: the (depth + 1)-th issue is refused, steps
    retire oldest first with the separate calls' bytes, and what sf_step_retire hands out stays valid -- without a copy --
    through further issues until the next retire."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(677, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.set_option(_abi.SF_OPT_STEP_DEPTH, depth)
        f.set_option(_abi.SF_OPT_STEP_LANES, lanes)
        f.set_option(_abi.SF_OPT_STEP_SPECULATE, speculate)
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m_ref, res_ref = _two_calls(f, sa, sb, n_kf)
        for _ in range(depth):
            f.step_issue(sa, sb)
        with pytest.raises(lib.SepfinderError):
            f.step_issue(sa, sb)
        with pytest.raises(lib.SepfinderError):
            f.set_option(_abi.SF_OPT_STEP_DEPTH, 2)          # not while steps are in flight
        for _ in range(3 * depth + 2):
            view = f.step_retire(copy=False)
            f.step_issue(sa, sb)                            # (the ring has depth + 1 blocks: the view is still the step's)
            _check_step(view, m_ref, res_ref, True)
        for _ in range(depth):
            _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        with pytest.raises(lib.SepfinderError):
            f.step_retire()


@pytest.mark.parametrize("max_nb", [200, 400])       # 200 < local rows: the serial device form; 400: the speculative one
def test_step_falls_back_when_the_filter_level_is_too_dense(oracle, max_nb):
    """Every vector shares its first 512 dimensions: at the 128-dimension prefix level the handle starts on every pair is
    a candidate, which only the DEVICE notices now (status word of the walk).  sf_step_retire then runs that query through
    the prefix ladder itself; the ladder settles on the full length and the following steps stay on the device."""
    rng = np.random.default_rng(21)
    dim, n_l, n_r, k = 4096, 260, 300, 128
    shared = rng.normal(size=512).astype(np.float32) * 0.03
    a = rng.normal(size=(n_l, dim)).astype(np.float32) / np.sqrt(dim)
    b = rng.normal(size=(n_r, dim)).astype(np.float32) / np.sqrt(dim)
    a[:, :512] = shared
    b[:, :512] = shared
    b[:80] = a[100:180] + rng.normal(size=(80, dim)).astype(np.float32) * (0.03 / np.sqrt(dim))
    b[:80, :512] = shared
    feats = synth.make_store_batch(5, n_r, k=k, cols=32, true_frac=0.5)
    p = synth.camera_params()
    p.iterations = 100
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = max_nb
    p.max_features = k
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        T = {key: _up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}
        sa = f.store_add_keyframes_device(n_r, k, 32, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
        sb = f.store_add_keyframes_device(n_r, k, 32, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
        torch.cuda.synchronize()
        f.nn_append_local(a)
        f.nn_append_received(b)
        # three steps queued before anybody knows the level is too dense: all three fall back, in order
        for _ in range(3):
            f.step_issue(sa, sb)
        outs = [f.step_retire(copy=True) for _ in range(3)]
        assert f.nn_last_filter_dims() == dim                # the ladder went to the full length
        mo, _, _ = oracle.find_matches(a.astype(np.float64), b.astype(np.float64), netvlad_distance=p.netvlad_distance,
                                       max_matches_nb=max_nb)
        assert len(mo) == 80
        m_ref, res_ref = _two_calls(f, sa, sb, max_nb)
        assert np.array_equal(m_ref["idx_local"], mo["idx_local"]) and np.array_equal(m_ref["idx_other"], mo["idx_other"])
        for out in outs:
            m, rom, recs, info = out
            assert m.tobytes() == m_ref.tobytes()
            assert np.array_equal(rom >= 0, res_ref["success"].astype(bool))
            for i in np.nonzero(rom >= 0)[0]:
                assert recs[rom[i]].tobytes() == res_ref[i].tobytes()
        # the level is settled: these stay on the device (streamed, no fallback) and give the same bytes
        for _ in range(3):
            f.step_issue(sa, sb)
        for _ in range(3):
            _check_step(f.step_retire(copy=True), m_ref, res_ref, True)


@pytest.mark.parametrize("n", [65537, 131073, 140000])
def test_step_queries_across_the_form_and_chunk_boundaries(n):
    """Overlapped steps (two streams) whose query has more candidates than the split form takes (65 536) and than one
    launch sequence holds (131 072): every chunk must run in the form the workspace was reserved for.  Round 3 decided
    the form per chunk: at 140 000 candidates the second chunk (8 928 pairs) took the split form -- which writes
    correspondence lists -- on a workspace reserved for the fused form, which has none: a device out-of-bounds write
    (commit 4822be5 fixed the decision, this is its regression test).  Results against sf_verify_pairs_device of the
    same pairs."""
    k, dim, base = 64, 128, 500
    feats = synth.make_store_batch(91, base, k=k, cols=32, true_frac=0.5)
    rng = np.random.default_rng(92)
    nv_a = rng.normal(size=(n, dim)).astype(np.float32); nv_a /= np.linalg.norm(nv_a, axis=1, keepdims=True)
    nv_b = nv_a + (0.002 * rng.normal(size=(n, dim))).astype(np.float32); nv_b /= np.linalg.norm(nv_b, axis=1, keepdims=True)
    p = synth.camera_params()
    p.iterations = 50
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n
    p.max_features = k
    p.store_capacity = 2 * n
    reps = (n + base - 1) // base
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        slots = []
        for which in ("a", "b"):        # keyframe i of a robot = base keyframe i mod 500 (tiled on the device)
            d = _up(feats["desc_" + which]).repeat(reps, 1, 1)[:n].contiguous()
            x = _up(feats["xyz_" + which]).repeat(reps, 1, 1)[:n].contiguous()
            kp = _up(feats["kp_" + which]).view(base, -1).repeat(reps, 1)[:n].contiguous()
            slots.append(f.store_add_keyframes_device(n, k, 32, d.data_ptr(), x.data_ptr(), kp.data_ptr()))
            torch.cuda.synchronize()
            del d, x, kp
        sa, sb = slots
        ta, tb = torch.from_numpy(nv_a).to(DEV), torch.from_numpy(nv_b).to(DEV)
        f.nn_append_received_device(ta.data_ptr(), n, dim)
        f.nn_append_local_device(tb.data_ptr(), n, dim)
        torch.cuda.synchronize()
        f.step_issue(sa, sb)
        f.step_issue(sa, sb)                         # two steps in flight on two streams: the overlapped form
        outs = [f.step_retire(copy=True), f.step_retire(copy=True)]
        m = outs[0][0]
        assert len(m) == n and np.array_equal(np.sort(m["idx_local"]), np.arange(n)) and np.array_equal(m["idx_local"], m["idx_other"])
        d_from = torch.from_numpy((sa + m["idx_other"]).astype(np.int32)).to(DEV)
        d_to = torch.from_numpy((sb + m["idx_local"]).astype(np.int32)).to(DEV)
        d_res = torch.zeros((n, RB), dtype=torch.uint8, device=DEV)
        f.verify_pairs_device(d_from.data_ptr(), d_to.data_ptr(), n, d_res.data_ptr())
        torch.cuda.synchronize()
        ref = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE)
        ok = ref["success"].astype(bool)
        assert 0.2 * n < ok.sum() < 0.8 * n
        for mm, rom, recs, info in outs:
            assert mm.tobytes() == m.tobytes()
            assert np.array_equal(rom >= 0, ok) and info["n_accepted"] == int(ok.sum())
            assert recs[rom[ok]].tobytes() == ref[ok].tobytes()


def test_step_result_carries_a_device_copy_of_the_records():
    """sf_step_result.d_records: the accepted records of a retired step once more in device memory, same slots as the
    pinned host block (what a multi-GPU host hands to its all-gather after the retire); absent while a mirror is set."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(877, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m_ref, res_ref = _two_calls(f, sa, sb, n_kf)
        for walk in (1, 0):
            f.set_option(_abi.SF_OPT_STEP_DEVICE_WALK, walk)
            for _ in range(3):
                f.step_issue(sa, sb)
            for _ in range(3):
                out = f.step_retire(copy=True)
                _check_step(out, m_ref, res_ref, True)
                n = out[3]["n_records"]
                assert out[3]["d_records"] != 0 and n > 0
                d = torch.zeros((n, RB), dtype=torch.uint8, device=DEV)
                f.memcpy_device_async(d.data_ptr(), out[3]["d_records"], n * RB)      # (None: the handle's stream)
                f.synchronize()
                assert d.cpu().numpy().tobytes() == out[2].tobytes()
        send = torch.zeros((n_kf + n_kf // 8 + 257, RB), dtype=torch.uint8, device=DEV)
        f.step_mirror(send[1:].data_ptr(), send.data_ptr(), n_kf + n_kf // 8 + 256)
        f.step_issue(sa, sb)
        assert f.step_retire()[3]["d_records"] == 0
        f.step_mirror(None, None, 0)


def test_streams_of_the_step_pipeline_are_placed_by_measurement():
    """The step pipeline's streams are picked from candidates by measuring which of them dispatch beside each other
    (sf_api.hip place_streams): the report says so once steps on several streams have run, names a class for every lane's
    main stream, and the steps' bytes are the separate calls' -- also with the streams given in another order (a foreign
    stream used first shifts every later stream's hardware queue)."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(977, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    foreign = torch.cuda.Stream()
    with torch.cuda.stream(foreign):
        torch.zeros(1 << 16, dtype=torch.uint8, device=DEV).clone()
    foreign.synchronize()
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        assert "not measured" in f.stream_placement()
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        m_ref, res_ref = _two_calls(f, sa, sb, n_kf)
        for _ in range(6):
            f.step_issue(sa, sb)
        for _ in range(6):
            _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        rep = f.stream_placement()
        assert rep.startswith("placement:"), rep
        if "classes," in rep:             # (measured: at least two classes, a class per lane's main stream)
            assert int(rep.split("placement: ")[1].split(" classes")[0]) >= 2, rep
            assert "main classes" in rep and "second streams: class" in rep, rep
    # the explicit form: the host picks the (quiet) moment -- before the first step -- and the steps find the streams placed
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.streams_prepare()
        rep0 = f.stream_placement()
        assert rep0.startswith("placement:") and "not measured" not in rep0, rep0
        f.streams_prepare()               # idempotent
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        for _ in range(4):
            f.step_issue(sa, sb)
        for _ in range(4):
            _check_step(f.step_retire(copy=True), m_ref, res_ref, True)
        assert f.stream_placement() == rep0


def test_a_copy_out_of_d_records_is_ordered_before_the_blocks_reuse():
    """sf_memcpy_device_async out of sf_step_result.d_records on a stream of the caller's that runs LATE (a long launch
    queued in front of every copy): the step that reuses the block must wait for the copy.  Two queries with different
    separators alternate through a ring of two, so a block is rewritten with OTHER records one retire later."""
    n_kf, k, dim = 96, 200, 512
    feats, nv_a, nv_b = _world(1077, n_kf, k, dim)
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = dim
    p.netvlad_max_matches_nb = n_kf
    p.max_features = k
    side = torch.cuda.Stream()
    ballast = torch.zeros((4096, 4096), dtype=torch.float32, device=DEV)
    with lib.SeparatorFinder(p) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.set_option(_abi.SF_OPT_STEP_DEPTH, 2)
        sa, sb, keep = _fill(f, feats, nv_a, nv_b, n_kf, k)
        # the second query: the keyframes' roles swapped (the same matches, the inverse poses: other record bytes); a
        # block is reused three steps later, i.e. by the OTHER query
        copies, want = [], []
        for it in range(24):
            if it % 2 == 0:
                f.step_issue(sa, sb)
            else:
                f.step_issue(sb, sa)
            if it >= 1:
                out = f.step_retire(copy=True)
                n = out[3]["n_records"]
                assert n > 0
                d = torch.zeros((n, RB), dtype=torch.uint8, device=DEV)
                with torch.cuda.stream(side):
                    for _ in range(3):
                        ballast @ ballast                      # ~ms of work in front of the copy
                    f.memcpy_device_async(d.data_ptr(), out[3]["d_records"], n * RB, side.cuda_stream)
                copies.append(d)
                want.append(out[2].tobytes())
        f.step_retire()
        torch.cuda.synchronize()
        assert len(set(want)) >= 2                             # (the two queries' records really differ)
        for d, w in zip(copies, want):
            assert d.cpu().numpy().tobytes() == w
