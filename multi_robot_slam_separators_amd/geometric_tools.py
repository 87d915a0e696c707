"""Host-side mirror of the reference's geometry node for the hot path
(PKG/src/stereoCamGeometricTools.cpp:122-178; PKG = ros_ws/src/multi_robot_separators).
`getFeaturesAndDescriptor` (feature extraction, :100-120) is out of scope."""
from . import _abi
from .messages import EstTransformResponse, PoseWithCovariance


class StereoCamGeometricTools:
    def __init__(self, backend):
        """backend.estimate_transform(FeatureArrays from, FeatureArrays to) -> sf_result record.
        The camera model `cam_` and Vis/MinInliers (stereoCamGeometricTools.cpp:76,87) live in the
        backend's sf_params."""
        self.backend = backend

    def estimateTransformation(self, req):
        """EstTransform.srv: always returns (the reference handler returns true at :177);
        a failed estimation is success=False with an all-zero pose (:168-175, MsgConversion.cpp:77-80).
        Size mismatches the reference UASSERT-aborts on raise instead."""
        f_from = _abi.FeatureArrays(req.descriptorsFrom, req.kptsFrom3D, req.kptsFrom)   # :135-140
        f_to = _abi.FeatureArrays(req.descriptorsTo, req.kptsTo3D, req.kptsTo)
        r = self.backend.estimate_transform(f_from, f_to)
        return EstTransformResponse(PoseWithCovariance.from_result(r), bool(r["success"]))
