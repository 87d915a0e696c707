"""Plain-Python stand-ins for the reference's ROS message / service types (field names and
meanings exactly as in PKG/msg/*.msg and PKG/srv/*.srv; PKG = ros_ws/src/multi_robot_separators).
ROS itself is out of scope (and absent from this image): these carry the same payloads between
the host-side mirrors so the hot path can be driven and checked end to end."""
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import _abi


@dataclass
class Pose:                       # geometry_msgs/Pose
    position: np.ndarray = field(default_factory=lambda: np.zeros(3))
    orientation: np.ndarray = field(default_factory=lambda: np.zeros(4))   # x, y, z, w


@dataclass
class PoseWithCovariance:         # geometry_msgs/PoseWithCovariance
    pose: Pose = field(default_factory=Pose)
    covariance: np.ndarray = field(default_factory=lambda: np.zeros(36))

    @staticmethod
    def from_result(r):
        return PoseWithCovariance(Pose(np.array(r["position"], dtype=np.float64),
                                       np.array(r["orientation"], dtype=np.float64)),
                                  np.array(r["covariance"], dtype=np.float64))


@dataclass
class GeomFeatures:               # GetFeatsAndDesc.srv response: Descriptors + KeyPoint3DVec + KeyPointVec
    descriptors: np.ndarray       # rows x cols uint8            (Descriptors.msg)
    kpts3D: np.ndarray            # size x 3 float32             (KeyPoint3DVec.msg)
    kpts: np.ndarray              # size KEYPOINT_DTYPE records  (KeyPointVec.msg)

    def arrays(self) -> _abi.FeatureArrays:
        return _abi.FeatureArrays(self.descriptors, self.kpts3D, self.kpts)


@dataclass
class GetFeatsAndDescRequest:     # GetFeatsAndDesc.srv:1-2 (sensor_msgs/Image x 2, MONO8 after cv_bridge)
    image_left: np.ndarray        # h x w uint8
    image_right: np.ndarray


@dataclass
class FindMatchesRequest:         # FindMatches.srv:1
    new_netvlad_descriptors: np.ndarray   # float64[] (flat)


@dataclass
class FindMatchesResponse:        # FindMatches.srv:3-9
    kf_ids_computing_robot: List[int] = field(default_factory=list)
    frames_kept_ids_computing_robot: List[int] = field(default_factory=list)
    frames_kept_ids_querying_robot: List[int] = field(default_factory=list)
    descriptors_vec: list = field(default_factory=list)
    kpts3D_vec: list = field(default_factory=list)
    kpts_vec: list = field(default_factory=list)
    pose_estimates: list = field(default_factory=list)


@dataclass
class EstTransformRequest:        # EstTransform.srv:1-6
    descriptorsFrom: np.ndarray
    descriptorsTo: np.ndarray
    kptsFrom3D: np.ndarray
    kptsTo3D: np.ndarray
    kptsFrom: np.ndarray
    kptsTo: np.ndarray


@dataclass
class EstTransformResponse:       # EstTransform.srv:8-9
    poseWithCov: PoseWithCovariance = field(default_factory=PoseWithCovariance)
    success: bool = False


@dataclass
class ReceiveSeparatorsRequest:   # ReceiveSeparators.srv:1-10
    robot_from_id: int = 0
    robot_to_id: int = 0
    kf_ids_from: List[int] = field(default_factory=list)
    kf_ids_to: List[int] = field(default_factory=list)
    frames_kepts_ids_from: List[int] = field(default_factory=list)
    frames_kepts_ids_to: List[int] = field(default_factory=list)
    pose_estimates_from: list = field(default_factory=list)
    pose_estimates_to: list = field(default_factory=list)
    transform_est_success: List[bool] = field(default_factory=list)
    separators: List[PoseWithCovariance] = field(default_factory=list)


def check_int16(values, what):
    """kf / frame ids are int16 on the wire (FindMatches.srv:3-5, ReceiveSeparators.srv:3-6)."""
    for v in values:
        if not -32768 <= int(v) <= 32767:
            raise OverflowError("%s %d does not fit the int16 wire type" % (what, v))
