"""Host-side mirror of the reference's DataHandler for the hot path only
(PKG/scripts/data_handler.py; PKG = ros_ws/src/multi_robot_separators): same method names,
argument meaning and state (`local_kf_already_used`, `other_kf_already_used`,
`frames_kept_pairs_ignored`, `kf_ids_of_frames_kept`, `nb_descriptors_already_sent`), with the
arithmetic delegated to the MI355X library through a small backend interface.

Out of scope and therefore absent: image queues, keyframe selection, TensorFlow NetVLAD
inference, GPS logging, rospy plumbing.  Keyframes enter through `add_keyframe()` with their
already-computed NetVLAD descriptor and geometric features (what `get_keyframes` +
`compute_descriptors` produce in the reference, data_handler.py:143-164,212-295).
"""
import numpy as np

from . import _abi
from .messages import (FindMatchesResponse, GeomFeatures, PoseWithCovariance,
                       ReceiveSeparatorsRequest, check_int16)


class FinderBackend:
    """Adapter: the backend interface DataHandler / StereoCamGeometricTools use, served by a
    lib.SeparatorFinder (the product).  tests/ provide an oracle-backed twin for comparison."""

    def __init__(self, finder, cam=None, detector=None, stereo_flow=None):
        """cam (sf_stereo_camera), detector / stereo_flow (None: rtabmap's defaults): only get_features needs them."""
        self.f = finder
        self.cam, self.detector, self.stereo_flow = cam, detector, stereo_flow
        self.slots = []               # store slot of every keyframe get_features produced, in call order

    def nn_append_local(self, rows):
        self.f.nn_append_local(rows)

    def nn_append_received(self, rows):
        self.f.nn_append_received(rows)

    def mark_local_used(self, i):
        self.f.nn_mark_local_used(i)

    def mark_other_used(self, j):
        self.f.nn_mark_other_used(j)

    def ignore_pair(self, i, j):
        self.f.nn_ignore_pair(i, j)

    def find_matches(self):
        m = self.f.nn_find_matches()
        return [(int(r["idx_local"]), int(r["idx_other"])) for r in m]

    def estimate_transform(self, f_from, f_to):
        return self.f.estimate_transform(f_from, f_to)

    def get_features(self, left, right):
        if self.cam is None:
            raise ValueError("FinderBackend needs the stereo camera model for get_features")
        desc, xyz, kp, slot = self.f.get_features_and_descriptor(left, right, self.cam, self.detector, self.stereo_flow)
        self.slots.append(slot)
        return desc, xyz, kp


class DataHandler:
    def __init__(self, backend, local_robot_id, other_robot_id, netvlad_dimensions=128,
                 send_estimates_of_poses=False, add_separators_pose_graph=None):
        self.backend = backend
        self.local_robot_id = local_robot_id          # data_handler.py:88-89
        self.other_robot_id = other_robot_id
        self.netvlad_dimensions = netvlad_dimensions  # :97
        self.send_estimates_of_poses = send_estimates_of_poses
        # the external back-end service `add_separators_pose_graph` (:84-85): any callable
        self.s_add_seps_pose_graph = add_separators_pose_graph or (lambda *a: None)
        self.local_descriptors = []                   # :40
        self.received_descriptors = []
        self.geometric_feats = []                     # :268
        self.kf_ids_of_frames_kept = []               # :287
        self.local_kf_already_used = []               # :402-405
        self.other_kf_already_used = []
        self.frames_kept_pairs_ignored = []           # :437-438
        self.separators_found = []
        self.nb_descriptors_already_sent = 0          # find_separators.py:59,68
        self.nb_kf_odom = 0

    # ---- ingestion (stands in for get_keyframes + compute_descriptors) -------------------------
    def add_keyframe(self, netvlad_descriptor, geometric_feats, kf_id=None):
        d = np.asarray(netvlad_descriptor, dtype=np.float64).reshape(-1)
        d = d[: self.netvlad_dimensions]              # :157-158 keep the first netvlad_dimensions
        if d.size != self.netvlad_dimensions:
            raise ValueError("NetVLAD descriptor shorter than netvlad_dimensions")
        self.nb_kf_odom += 1
        kf_id = self.nb_kf_odom - 1 if kf_id is None else kf_id
        check_int16([kf_id, len(self.local_descriptors)], "keyframe id")
        self.local_descriptors.append(d)
        self.backend.nn_append_local(d.reshape(1, -1))
        self.geometric_feats.append(geometric_feats)
        self.kf_ids_of_frames_kept.append(kf_id)

    # ---- data_handler.py:166-209 ----------------------------------------------------------------
    def find_matches(self):
        return self.backend.find_matches()

    # ---- data_handler.py:297-337 ----------------------------------------------------------------
    def find_matches_service(self, find_matches_req):
        new = np.asarray(find_matches_req.new_netvlad_descriptors, dtype=np.float64).reshape(
            -1, self.netvlad_dimensions)              # :300-301
        if new.shape[0]:
            self.received_descriptors.extend(new)
            self.backend.nn_append_received(new)
        if not (len(self.received_descriptors) > 0 and len(self.local_descriptors) > 0):   # :308-311
            return FindMatchesResponse()
        matches = self.find_matches()
        resp = FindMatchesResponse()
        for idx_local, idx_other in matches:          # :316-325
            feats = self.get_geom_features(idx_local)
            if not feats:
                continue
            resp.frames_kept_ids_computing_robot.append(idx_local)
            resp.frames_kept_ids_querying_robot.append(idx_other)
            resp.descriptors_vec.append(feats.descriptors)
            resp.kpts3D_vec.append(feats.kpts3D)
            resp.kpts_vec.append(feats.kpts)
        resp.kf_ids_computing_robot = self.get_kf_ids_from_frames_kept_ids(
            resp.frames_kept_ids_computing_robot)     # :327-328
        check_int16(resp.kf_ids_computing_robot + resp.frames_kept_ids_computing_robot
                    + resp.frames_kept_ids_querying_robot, "id")
        return resp

    # ---- data_handler.py:339-370 ----------------------------------------------------------------
    def found_separators_local(self, kf_ids_from, kf_ids_to, frames_kept_ids_from, frames_kept_ids_to,
                               pose_estimates_from, pose_estimates_to, transform_est_success, separators):
        kept = ReceiveSeparatorsRequest(self.local_robot_id, self.other_robot_id)
        for i in range(len(kf_ids_from)):
            if transform_est_success[i]:
                kept.kf_ids_from.append(kf_ids_from[i])
                kept.kf_ids_to.append(kf_ids_to[i])
                kept.separators.append(separators[i])
                kept.transform_est_success.append(transform_est_success[i])
                kept.frames_kepts_ids_from.append(frames_kept_ids_from[i])
                kept.frames_kepts_ids_to.append(frames_kept_ids_to[i])
                self.separators_found.append((kf_ids_from[i], kf_ids_to[i], separators[i]))
        self.s_add_seps_pose_graph(kept)
        return kept

    # ---- data_handler.py:373-419 ----------------------------------------------------------------
    def receive_separators_service(self, req):
        kept = ReceiveSeparatorsRequest(req.robot_from_id, req.robot_to_id)
        for i in range(len(req.kf_ids_from)):
            if req.transform_est_success[i]:
                kept.frames_kepts_ids_from.append(req.frames_kepts_ids_from[i])
                kept.frames_kepts_ids_to.append(req.frames_kepts_ids_to[i])
                kept.kf_ids_from.append(req.kf_ids_from[i])
                kept.kf_ids_to.append(req.kf_ids_to[i])
                kept.separators.append(req.separators[i])
                kept.transform_est_success.append(req.transform_est_success[i])
                self.separators_found.append((req.kf_ids_to[i], req.kf_ids_from[i], req.separators[i]))
                # only the COMPUTING robot updates its masks (:402-405)
                self.local_kf_already_used.append(req.frames_kepts_ids_to[i])
                self.backend.mark_local_used(req.frames_kepts_ids_to[i])
                self.other_kf_already_used.append(req.frames_kepts_ids_from[i])
                self.backend.mark_other_used(req.frames_kepts_ids_from[i])
            else:
                self.add_frames_kept_pairs_to_ignore(req.frames_kepts_ids_to[i], req.frames_kepts_ids_from[i])
        self.s_add_seps_pose_graph(kept)              # :411-415
        return True

    def get_geom_features(self, id):                  # :421-422
        return self.geometric_feats[id]

    def add_frames_kept_pairs_to_ignore(self, id_local, id_other):   # :437-438
        self.frames_kept_pairs_ignored.append([id_local, id_other])
        self.backend.ignore_pair(id_local, id_other)

    def get_kf_ids_from_frames_kept_ids(self, frames_kept_ids):       # :440-441
        return [int(self.kf_ids_of_frames_kept[i]) for i in frames_kept_ids]


def geom_features_from_arrays(fa: _abi.FeatureArrays) -> GeomFeatures:
    return GeomFeatures(fa.desc, fa.xyz, fa.kpts)


__all__ = ["DataHandler", "FinderBackend", "geom_features_from_arrays", "PoseWithCovariance"]
