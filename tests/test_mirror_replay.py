"""2-robot replay through the host-side mirrors of the reference's interface
(DataHandler / StereoCamGeometricTools / find_separators): CPU run on the oracle backend checks
the host logic; the GPU run must exchange IDENTICAL ReceiveSeparators requests."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import _abi, synth
from multi_robot_slam_separators_amd.messages import FindMatchesRequest

from oracle_backend import OracleBackend
from replay import make_world, run_replay


def params():
    p = synth.camera_params()
    p.iterations = 200
    p.netvlad_dimensions = 128
    return p


@pytest.fixture(scope="module")
def oracle_run():
    p = params()
    world = make_world(77)
    return world, p, run_replay(world, OracleBackend, p)


def test_replay_host_logic_on_oracle(oracle_run):
    world, p, (exchanged, (dhA, dhB), log) = oracle_run
    reqs = [r for _, r in exchanged if r is not None]
    assert len(reqs) >= 6
    n_ok = sum(sum(r.transform_est_success) for r in reqs)
    n_fail = sum(len(r.transform_est_success) - sum(r.transform_est_success) for r in reqs)
    assert n_ok >= 5 and n_fail >= 2                      # revisits accepted, aliases rejected
    # masks are updated only on the COMPUTING robot (data_handler.py:402-408)
    used_on_B = [f for d, r in exchanged if r is not None and d == "A->B"
                 for f, s in zip(r.frames_kepts_ids_to, r.transform_est_success) if s]
    assert dhB.local_kf_already_used == used_on_B
    ign_on_B = [[t, f] for d, r in exchanged if r is not None and d == "A->B"
                for f, t, s in zip(r.frames_kepts_ids_from, r.frames_kepts_ids_to, r.transform_est_success) if not s]
    assert dhB.frames_kept_pairs_ignored == ign_on_B
    # a used / ignored pair is never proposed again
    seen = set()
    for d, r in exchanged:
        if r is None:
            continue
        for f, t in zip(r.frames_kepts_ids_from, r.frames_kepts_ids_to):
            assert (d, f, t) not in seen
            seen.add((d, f, t))
    # incremental shipping: every descriptor sent exactly once (find_separators.py:59-68)
    assert dhA.nb_descriptors_already_sent == len(dhA.local_descriptors) == len(dhB.received_descriptors)
    # kf ids are mapped through kf_ids_of_frames_kept (data_handler.py:440-441)
    for d, r in exchanged:
        if r is None:
            continue
        mult_from, mult_to = (2, 3) if d == "A->B" else (3, 2)
        off_from, off_to = (0, 1) if d == "A->B" else (1, 0)
        assert r.kf_ids_from == [mult_from * f + off_from for f in r.frames_kepts_ids_from]
        assert r.kf_ids_to == [mult_to * t + off_to for t in r.frames_kepts_ids_to]
    # back-end log only ever receives accepted separators
    for side in log.values():
        for kept in side:
            assert all(kept.transform_est_success)


def test_find_matches_service_empty_database(oracle_run):
    _, p, _ = oracle_run
    from multi_robot_slam_separators_amd.data_handler import DataHandler
    dh = DataHandler(OracleBackend(p), 0, 1, 128)
    resp = dh.find_matches_service(FindMatchesRequest(np.zeros(256)))
    assert resp.kf_ids_computing_robot == [] and len(dh.received_descriptors) == 2   # :308-311


def test_int16_wire_limit():
    from multi_robot_slam_separators_amd.messages import check_int16
    with pytest.raises(OverflowError):
        check_int16([40000], "kf id")


@pytest.mark.gpu
def test_replay_gpu_exchanges_identical_separators(oracle_run):
    from multi_robot_slam_separators_amd import lib
    from multi_robot_slam_separators_amd.data_handler import FinderBackend
    world, p, (ref_exchanged, (rA, rB), ref_log) = oracle_run
    finders = []

    def make_backend(pp):
        f = lib.SeparatorFinder(pp)
        finders.append(f)
        return FinderBackend(f)

    exchanged, (dhA, dhB), log = run_replay(world, make_backend, p)
    assert len(exchanged) == len(ref_exchanged)
    for (d, r), (d0, r0) in zip(exchanged, ref_exchanged):
        assert d == d0 and (r is None) == (r0 is None)
        if r is None:
            continue
        for fld in ("robot_from_id", "robot_to_id", "kf_ids_from", "kf_ids_to", "frames_kepts_ids_from",
                    "frames_kepts_ids_to", "transform_est_success"):
            assert getattr(r, fld) == getattr(r0, fld), fld          # integer / bool fields: exact
        for s, s0 in zip(r.separators, r0.separators):
            assert np.linalg.norm(s.pose.position - s0.pose.position) <= 1e-4
            assert np.abs(s.pose.orientation - s0.pose.orientation).max() <= 1e-3
            assert np.allclose(s.covariance, s0.covariance, rtol=1e-9)
    assert dhA.local_kf_already_used == rA.local_kf_already_used
    assert dhB.frames_kept_pairs_ignored == rB.frames_kept_pairs_ignored
    for f in finders:
        f.close()
