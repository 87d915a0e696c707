"""Host-side mirror of the reference's geometry node (PKG/src/stereoCamGeometricTools.cpp; PKG =
ros_ws/src/multi_robot_separators): `estimateTransformation` (:122-178, the hot path) and
`getFeaturesAndDescriptor` (:100-120)."""
from . import _abi
from .messages import EstTransformResponse, GeomFeatures, PoseWithCovariance


class StereoCamGeometricTools:
    def __init__(self, backend):
        """backend.estimate_transform(FeatureArrays from, FeatureArrays to) -> sf_result record.
        The camera model `cam_` and Vis/MinInliers (stereoCamGeometricTools.cpp:76,87) live in the
        backend's sf_params."""
        self.backend = backend

    def getFeaturesAndDescriptor(self, req):
        """GetFeatsAndDesc.srv: req.image_left / req.image_right are MONO8 images (uint8 [h, w], what
        cv_bridge::toCvCopy yields at :104-105).  Returns the response's three arrays (:116-118); the handler itself
        always returns true (:119)."""
        desc, xyz, kp = self.backend.get_features(req.image_left, req.image_right)
        return GeomFeatures(desc, xyz, kp)

    def estimateTransformation(self, req):
        """EstTransform.srv: always returns (the reference handler returns true at :177);
        a failed estimation is success=False with an all-zero pose (:168-175, MsgConversion.cpp:77-80).
        Size mismatches the reference UASSERT-aborts on raise instead."""
        f_from = _abi.FeatureArrays(req.descriptorsFrom, req.kptsFrom3D, req.kptsFrom)   # :135-140
        f_to = _abi.FeatureArrays(req.descriptorsTo, req.kptsTo3D, req.kptsTo)
        r = self.backend.estimate_transform(f_from, f_to)
        return EstTransformResponse(PoseWithCovariance.from_result(r), bool(r["success"]))
