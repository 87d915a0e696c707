"""ROS1 wire format (TCPROS message body serialisation) of the hot path's service payloads, without ROS.

Rules (ROS1 message serialisation): little endian; primitives packed with no padding; a
variable-length array is a uint32 element count followed by the elements; fixed-length arrays
(`float64[36]`) have no count; `bool` is one byte; nested messages are inlined.  Message definitions:
PKG/msg/{Descriptors,KeyPoint3DVec,KeyPointVec}.msg, PKG/srv/{FindMatches,EstTransform,
ReceiveSeparators}.srv (PKG = ros_ws/src/multi_robot_separators), rtabmap_ros/KeyPoint
(pt.x, pt.y, size, angle, response: float32; octave, class_id: int32 -- fields as used at
PKG/src/MsgConversion.cpp:50-56), rtabmap_ros/Point3f (3 x float32),
geometry_msgs/PoseWithCovariance (7 x float64 + float64[36] = 344 bytes, the constant the reference's
own tools/evaluate_communication.py:88,113 uses).

Purpose: request dumps recorded from a live system (rosbag / a tap on the relay node,
PKG/src/communication.cpp) can be replayed through the MI355X library byte for byte, and the mirrors
in this package can emit payloads a ROS node would accept.
"""
import struct

import numpy as np

from . import _abi
from .messages import (EstTransformRequest, EstTransformResponse, FindMatchesRequest, FindMatchesResponse,
                       Pose, PoseWithCovariance, ReceiveSeparatorsRequest)

POSE_WITH_COV_BYTES = 344
KEYPOINT_BYTES = 28
POINT3F_BYTES = 12


class Reader:
    def __init__(self, buf):
        self.b = memoryview(bytes(buf))
        self.o = 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("truncated ROS message (need %d bytes at offset %d of %d)" % (n, self.o, len(self.b)))
        v = self.b[self.o: self.o + n]
        self.o += n
        return v

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def array(self, dtype, count=None):
        n = self.u32() if count is None else count
        dt = np.dtype(dtype)
        return np.frombuffer(self.take(n * dt.itemsize), dtype=dt).copy()

    def done(self):
        if self.o != len(self.b):
            raise ValueError("%d trailing bytes after ROS message" % (len(self.b) - self.o))


def _arr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return struct.pack("<I", a.size) + a.tobytes()


# ---- geometry_msgs/PoseWithCovariance ---------------------------------------------------------------
def pack_pose_with_cov(p):
    pos = np.asarray(p.pose.position, dtype="<f8").reshape(3)
    q = np.asarray(p.pose.orientation, dtype="<f8").reshape(4)
    cov = np.asarray(p.covariance, dtype="<f8").reshape(36)
    return pos.tobytes() + q.tobytes() + cov.tobytes()


def read_pose_with_cov(r):
    v = r.array("<f8", 43)
    return PoseWithCovariance(Pose(v[0:3].copy(), v[3:7].copy()), v[7:43].copy())


# ---- multi_robot_separators/Descriptors, KeyPoint3DVec, KeyPointVec ------------------------------------
def pack_descriptors(desc):
    d = np.ascontiguousarray(desc, dtype=np.uint8)
    rows, cols = (d.shape if d.ndim == 2 else (0, 0))
    if rows > 65535 or cols > 65535:
        raise OverflowError("Descriptors.rows / cols are uint16 on the wire")
    return struct.pack("<HH", rows, cols) + _arr(d.reshape(-1), np.uint8)


def read_descriptors(r):
    rows, cols = struct.unpack("<HH", r.take(4))
    data = r.array(np.uint8)
    if data.size != rows * cols:
        raise ValueError("Descriptors: %d bytes for %d x %d" % (data.size, rows, cols))
    return data.reshape(rows, cols)


def pack_kpts3d(xyz):
    x = np.ascontiguousarray(xyz, dtype="<f4").reshape(-1, 3)
    if x.shape[0] > _abi.SF_MAX_FEATURES:
        raise OverflowError("KeyPoint3DVec.size is int16 on the wire")
    return struct.pack("<h", x.shape[0]) + struct.pack("<I", x.shape[0]) + x.tobytes()


def read_kpts3d(r):
    (size,) = struct.unpack("<h", r.take(2))
    n = r.u32()
    x = r.array("<f4", 3 * n).reshape(n, 3)
    return x[:size] if size <= n else x      # keypoints3DFromROS reads msg.size entries (MsgConversion.cpp:8-18)


def pack_kpts(kpts):
    k = np.ascontiguousarray(kpts, dtype=_abi.KEYPOINT_DTYPE)
    if k.shape[0] > _abi.SF_MAX_FEATURES:
        raise OverflowError("KeyPointVec.size is int16 on the wire")
    return struct.pack("<h", k.shape[0]) + struct.pack("<I", k.shape[0]) + k.tobytes()


def read_kpts(r):
    (size,) = struct.unpack("<h", r.take(2))
    n = r.u32()
    k = r.array(_abi.KEYPOINT_DTYPE, n)
    return k[:size] if size <= n else k


# ---- FindMatches.srv ----------------------------------------------------------------------------------
def serialize_find_matches_request(req):
    return _arr(req.new_netvlad_descriptors, "<f8")


def deserialize_find_matches_request(buf):
    r = Reader(buf)
    v = r.array("<f8")
    r.done()
    return FindMatchesRequest(v)


def serialize_find_matches_response(res):
    out = [_arr(res.kf_ids_computing_robot, "<i2"), _arr(res.frames_kept_ids_computing_robot, "<i2"),
           _arr(res.frames_kept_ids_querying_robot, "<i2")]
    out.append(struct.pack("<I", len(res.descriptors_vec)) + b"".join(pack_descriptors(d) for d in res.descriptors_vec))
    out.append(struct.pack("<I", len(res.kpts3D_vec)) + b"".join(pack_kpts3d(x) for x in res.kpts3D_vec))
    out.append(struct.pack("<I", len(res.kpts_vec)) + b"".join(pack_kpts(k) for k in res.kpts_vec))
    out.append(struct.pack("<I", len(res.pose_estimates)) + b"".join(pack_pose_with_cov(p) for p in res.pose_estimates))
    return b"".join(out)


def deserialize_find_matches_response(buf):
    r = Reader(buf)
    res = FindMatchesResponse()
    res.kf_ids_computing_robot = r.array("<i2").tolist()
    res.frames_kept_ids_computing_robot = r.array("<i2").tolist()
    res.frames_kept_ids_querying_robot = r.array("<i2").tolist()
    res.descriptors_vec = [read_descriptors(r) for _ in range(r.u32())]
    res.kpts3D_vec = [read_kpts3d(r) for _ in range(r.u32())]
    res.kpts_vec = [read_kpts(r) for _ in range(r.u32())]
    res.pose_estimates = [read_pose_with_cov(r) for _ in range(r.u32())]
    r.done()
    return res


# ---- EstTransform.srv ----------------------------------------------------------------------------------
def serialize_est_transform_request(req):
    return (pack_descriptors(req.descriptorsFrom) + pack_descriptors(req.descriptorsTo)
            + pack_kpts3d(req.kptsFrom3D) + pack_kpts3d(req.kptsTo3D) + pack_kpts(req.kptsFrom) + pack_kpts(req.kptsTo))


def deserialize_est_transform_request(buf):
    r = Reader(buf)
    d_from, d_to = read_descriptors(r), read_descriptors(r)
    x_from, x_to = read_kpts3d(r), read_kpts3d(r)
    k_from, k_to = read_kpts(r), read_kpts(r)
    r.done()
    return EstTransformRequest(d_from, d_to, x_from, x_to, k_from, k_to)


def serialize_est_transform_response(res):
    return pack_pose_with_cov(res.poseWithCov) + struct.pack("<?", bool(res.success))


def deserialize_est_transform_response(buf):
    r = Reader(buf)
    p = read_pose_with_cov(r)
    (ok,) = struct.unpack("<?", r.take(1))
    r.done()
    return EstTransformResponse(p, bool(ok))


# ---- ReceiveSeparators.srv ------------------------------------------------------------------------------
def serialize_receive_separators_request(req):
    for v in (req.robot_from_id, req.robot_to_id):
        if not -128 <= int(v) <= 127:
            raise OverflowError("robot ids are int8 on the wire (ReceiveSeparators.srv:1-2)")
    out = [struct.pack("<bb", int(req.robot_from_id), int(req.robot_to_id)),
           _arr(req.kf_ids_from, "<i2"), _arr(req.kf_ids_to, "<i2"),
           _arr(req.frames_kepts_ids_from, "<i2"), _arr(req.frames_kepts_ids_to, "<i2")]
    for lst in (req.pose_estimates_from, req.pose_estimates_to):
        out.append(struct.pack("<I", len(lst)) + b"".join(pack_pose_with_cov(p) for p in lst))
    out.append(_arr(np.asarray(req.transform_est_success, dtype=bool), np.uint8))
    out.append(struct.pack("<I", len(req.separators)) + b"".join(pack_pose_with_cov(p) for p in req.separators))
    return b"".join(out)


def deserialize_receive_separators_request(buf):
    r = Reader(buf)
    a, b = struct.unpack("<bb", r.take(2))
    req = ReceiveSeparatorsRequest(a, b)
    req.kf_ids_from = r.array("<i2").tolist()
    req.kf_ids_to = r.array("<i2").tolist()
    req.frames_kepts_ids_from = r.array("<i2").tolist()
    req.frames_kepts_ids_to = r.array("<i2").tolist()
    req.pose_estimates_from = [read_pose_with_cov(r) for _ in range(r.u32())]
    req.pose_estimates_to = [read_pose_with_cov(r) for _ in range(r.u32())]
    req.transform_est_success = [bool(x) for x in r.array(np.uint8)]
    req.separators = [read_pose_with_cov(r) for _ in range(r.u32())]
    r.done()
    return req


# ---- request dumps: length-prefixed records ---------------------------------------------------------------
def write_dump(path, kind, payloads):
    """A dump file = magic, kind string, then uint32-length-prefixed serialized requests (the framing
    TCPROS itself uses for a service request body)."""
    with open(path, "wb") as f:
        k = kind.encode()
        f.write(b"SFDUMP1\0" + struct.pack("<I", len(k)) + k)
        for p in payloads:
            f.write(struct.pack("<I", len(p)) + p)


def read_dump(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"SFDUMP1\0":
        raise ValueError("not a sepfinder request dump")
    (n,) = struct.unpack("<I", data[8:12])
    kind = data[12:12 + n].decode()
    o = 12 + n
    out = []
    while o < len(data):
        (ln,) = struct.unpack("<I", data[o:o + 4])
        out.append(data[o + 4:o + 4 + ln])
        o += 4 + ln
    return kind, out
