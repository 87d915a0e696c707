"""Replays a `tests/golden/refsession_*.npz` fixture -- a multi-tick session of one computing robot recorded by
EXECUTING the reference's find_matches_service / receive_separators_service (oracle/gen_golden.py,
oracle/ref_exec.py) -- through (a) a raw NN backend (append / mask / find_matches calls, what the C-ABI exposes)
and (b) this repository's DataHandler mirror on top of such a backend, and asserts that every tick returns what the
reference returned.  Shared by the CPU tests (oracle backend) and the GPU tests (libsepfinder backend)."""
import glob
import os

import numpy as np

from multi_robot_slam_separators_amd.data_handler import DataHandler
from multi_robot_slam_separators_amd.messages import FindMatchesRequest, GeomFeatures, ReceiveSeparatorsRequest

SESSIONS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refsession_*.npz")))


def replay_backend(g, backend):
    """Raw backend calls in the order the reference mutates its state."""
    for t in range(int(g["ticks"])):
        nl, nr = g["t%d_new_local" % t].astype(np.float64), g["t%d_new_received" % t].astype(np.float64)
        if nl.shape[0]:
            backend.nn_append_local(nl)
        if nr.shape[0]:
            backend.nn_append_received(nr)
        want = np.stack([g["t%d_frames_computing" % t], g["t%d_frames_querying" % t]], axis=1)
        if t == 0 and want.shape[0] == 0 and (nl.shape[0] == 0 or nr.shape[0] == 0):
            got = np.zeros((0, 2), np.int32)     # the reference returns early on an empty database (:308-311)
        else:
            got = np.array(backend.find_matches(), dtype=np.int32).reshape(-1, 2)
        assert np.array_equal(got, want), "tick %d: matches differ from the reference's" % t
        for (il, io), ok in zip(want, g["t%d_success" % t]):
            if ok:
                backend.mark_local_used(int(il))
                backend.mark_other_used(int(io))
            else:
                backend.ignore_pair(int(il), int(io))


def replay_mirror(g, backend):
    """The same session through the DataHandler mirror (find_matches_service / receive_separators_service)."""
    dim = int(g["dim"])
    h = DataHandler(backend, local_robot_id=0, other_robot_id=1, netvlad_dimensions=dim)
    feats = GeomFeatures(np.zeros((0, 32), np.uint8), np.zeros((0, 3), np.float32), np.zeros(0))
    for t in range(int(g["ticks"])):
        for row, kf in zip(g["t%d_new_local" % t].astype(np.float64), g["t%d_new_local_kf_ids" % t]):
            h.add_keyframe(row, feats, kf_id=int(kf))
        resp = h.find_matches_service(FindMatchesRequest(g["t%d_new_received" % t].astype(np.float64).reshape(-1)))
        assert list(resp.frames_kept_ids_computing_robot) == g["t%d_frames_computing" % t].tolist(), t
        assert list(resp.frames_kept_ids_querying_robot) == g["t%d_frames_querying" % t].tolist(), t
        assert list(resp.kf_ids_computing_robot) == g["t%d_kf_ids_computing" % t].tolist(), t
        req = ReceiveSeparatorsRequest(
            robot_from_id=1, robot_to_id=0,
            kf_ids_from=[1000 + int(q) for q in resp.frames_kept_ids_querying_robot],
            kf_ids_to=list(resp.kf_ids_computing_robot),
            frames_kepts_ids_from=list(resp.frames_kept_ids_querying_robot),
            frames_kepts_ids_to=list(resp.frames_kept_ids_computing_robot),
            transform_est_success=[bool(s) for s in g["t%d_success" % t]],
            separators=[None] * len(resp.kf_ids_computing_robot))
        h.receive_separators_service(req)
        assert list(h.local_kf_already_used) == g["t%d_local_used_after" % t].tolist(), t
        assert list(h.other_kf_already_used) == g["t%d_other_used_after" % t].tolist(), t
        assert np.array_equal(np.array(h.frames_kept_pairs_ignored, dtype=np.int32).reshape(-1, 2),
                              g["t%d_ignored_after" % t]), t
    found = np.array([(a, b) for a, b, _ in h.separators_found], dtype=np.int32).reshape(-1, 2)
    assert np.array_equal(found, g["separators_found_kf"])
