#!/usr/bin/env python
# NOT RUN IN THIS REPOSITORY (no ROS in the image) -- see ros/README.md.
"""ROS1 shim: the `find_matches_compute` and `receive_separators_py` servers of the reference
(PKG/scripts/find_separators.py:30-34) answered from libsepfinder.so.

The NetVLAD nearest-neighbour search of DataHandler.find_matches (PKG/scripts/data_handler.py:166-209: masked
scipy cdist, row minima, argsort, sequential walk) and the bookkeeping that feeds it (:157-158, :300-301, :402-408,
:437-438) run on the GPU behind the C-ABI (sf_nn_append_*, sf_nn_mark_*, sf_nn_ignore_pair, sf_nn_find_matches); the
ROS side -- message types, service names, the queues, the 0.3 Hz loop of find_separators.py -- is the reference's.
`multi_robot_slam_separators_amd.data_handler.DataHandler` is the ROS-free mirror of the reference class that this
repository tests (tests/test_mirror_replay.py); this file only converts between its plain records and the generated
ROS messages.

Keyframes enter through `add_keyframe(netvlad_descriptor, GetFeatsAndDesc response, kf_id)`: call it where the
reference appends to `local_descriptors` / `geometric_feats` (data_handler.py:157-164 and :268-287)."""
import numpy as np
import rospy
from geometry_msgs.msg import PoseWithCovariance
from multi_robot_separators.msg import Descriptors, KeyPoint3DVec, KeyPointVec
from multi_robot_separators.srv import (FindMatches, FindMatchesResponse, ReceiveSeparators,
                                        ReceiveSeparatorsResponse)
from rtabmap_ros.msg import KeyPoint, Point3f

from multi_robot_slam_separators_amd import _abi, lib
from multi_robot_slam_separators_amd.data_handler import DataHandler, FinderBackend
from multi_robot_slam_separators_amd.messages import FindMatchesRequest, GeomFeatures, ReceiveSeparatorsRequest


def feats_from_ros(resp):
    """GetFeatsAndDesc response (or the three messages of one FindMatches entry) -> GeomFeatures of numpy arrays."""
    d = np.frombuffer(bytes(bytearray(resp.descriptors.data)), dtype=np.uint8).reshape(resp.descriptors.rows,
                                                                                       resp.descriptors.cols)
    xyz = np.array([[p.x, p.y, p.z] for p in resp.kpts3D.kpts3DVec], dtype=np.float32).reshape(-1, 3)
    kp = np.zeros(len(resp.kpts.kptsVec), dtype=_abi.KEYPOINT_DTYPE)
    for i, k in enumerate(resp.kpts.kptsVec):
        kp[i] = (k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id)
    return GeomFeatures(d, xyz, kp)


def feats_to_ros(g):
    d = Descriptors(rows=g.descriptors.shape[0], cols=g.descriptors.shape[1], data=g.descriptors.tobytes())
    p3 = KeyPoint3DVec(size=len(g.kpts3D), kpts3DVec=[Point3f(float(x), float(y), float(z)) for x, y, z in g.kpts3D])
    kv = KeyPointVec(size=len(g.kpts))
    for k in g.kpts:
        m = KeyPoint()
        m.pt.x, m.pt.y, m.size, m.angle, m.response, m.octave, m.class_id = (
            float(k["x"]), float(k["y"]), float(k["size"]), float(k["angle"]), float(k["response"]), int(k["octave"]),
            int(k["class_id"]))
        kv.kptsVec.append(m)
    return d, p3, kv


class SepfinderDataHandler(object):
    def __init__(self):
        p = _abi.default_params()
        p.netvlad_distance = rospy.get_param("netvlad_distance")              # data_handler.py:96-100
        p.netvlad_dimensions = rospy.get_param("netvlad_dimensions")
        p.netvlad_max_matches_nb = rospy.get_param("netvlad_max_matches_nb")
        p.min_inliers = rospy.get_param("separators_min_inliers")
        self.finder = lib.SeparatorFinder(p, device=rospy.get_param("~gpu", 0))
        add_seps = rospy.ServiceProxy("add_separators_pose_graph", ReceiveSeparators)        # :84-85
        self.dh = DataHandler(FinderBackend(self.finder), rospy.get_param("local_robot_id"),
                              rospy.get_param("other_robot_id"), p.netvlad_dimensions,
                              add_separators_pose_graph=lambda kept: add_seps(self._separators_to_ros(kept)))

    def add_keyframe(self, netvlad_descriptor, feats_resp, kf_id):
        self.dh.add_keyframe(netvlad_descriptor, feats_from_ros(feats_resp), kf_id)

    # data_handler.py:297-337
    def find_matches_service(self, req):
        r = self.dh.find_matches_service(FindMatchesRequest(np.asarray(req.new_netvlad_descriptors, dtype=np.float64)))
        out = FindMatchesResponse()
        out.kf_ids_computing_robot = list(r.kf_ids_computing_robot)
        out.frames_kept_ids_computing_robot = list(r.frames_kept_ids_computing_robot)
        out.frames_kept_ids_querying_robot = list(r.frames_kept_ids_querying_robot)
        for d, p3, kv in zip(r.descriptors_vec, r.kpts3D_vec, r.kpts_vec):
            md, mp, mk = feats_to_ros(GeomFeatures(d, p3, kv))
            out.descriptors_vec.append(md); out.kpts3D_vec.append(mp); out.kpts_vec.append(mk)
        return out

    # data_handler.py:373-419
    def receive_separators_service(self, req):
        plain = ReceiveSeparatorsRequest(req.robot_from_id, req.robot_to_id, list(req.kf_ids_from), list(req.kf_ids_to),
                                         list(req.frames_kepts_ids_from), list(req.frames_kepts_ids_to),
                                         list(req.pose_estimates_from), list(req.pose_estimates_to),
                                         list(req.transform_est_success), list(req.separators))
        return ReceiveSeparatorsResponse(bool(self.dh.receive_separators_service(plain)))

    @staticmethod
    def _separators_to_ros(kept):
        from multi_robot_separators.srv import ReceiveSeparatorsRequest as RosReq
        return RosReq(kept.robot_from_id, kept.robot_to_id, kept.kf_ids_from, kept.kf_ids_to, kept.frames_kepts_ids_from,
                      kept.frames_kepts_ids_to, [], [], kept.transform_est_success, kept.separators)


if __name__ == "__main__":
    rospy.init_node("find_separators", anonymous=False)
    handler = SepfinderDataHandler()
    s_find_matches = rospy.Service("find_matches_compute", FindMatches, handler.find_matches_service)
    s_receive_separators = rospy.Service("receive_separators_py", ReceiveSeparators, handler.receive_separators_service)
    rospy.spin()
