// NOT COMPILED IN THIS REPOSITORY (no ROS in the image) -- see ros/README.md.
//
// ROS1 node serving the reference's two geometric services from libsepfinder.so:
//   get_features_and_descriptor  (multi_robot_separators/GetFeatsAndDesc)  <- sf_get_features_and_descriptor
//   estimate_transformation      (multi_robot_separators/EstTransform)     <- sf_estimate_transform
// It stands where PKG/src/stereoCamGeometricTools.cpp stands in the reference: same node parameters, same topics read
// at start-up (left/camera_info, right/camera_info, the tf between the optical frames and `frame_id`), same service
// names and types, so PKG/scripts/data_handler.py and find_separators.py call it unchanged.  No rtabmap, OpenCV or
// cv_bridge: MONO8 images are taken from sensor_msgs/Image directly, the message arrays are borrowed without a copy
// (rtabmap_ros/Point3f = 3 packed float32, rtabmap_ros/KeyPoint = the 28-byte sf_keypoint, PKG/src/MsgConversion.cpp:50-56).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/image_encodings.h>
#include <tf/transform_listener.h>

#include <multi_robot_separators/EstTransform.h>
#include <multi_robot_separators/GetFeatsAndDesc.h>

#include <sepfinder.h>

namespace {

// 3x4 row-major float from a tf transform (what rtabmap_ros::transformFromTF builds, stereoCamGeometricTools.cpp:56)
void rows_from_tf(const tf::Transform& t, float out[12]) {
  const tf::Matrix3x3& R = t.getBasis();
  const tf::Vector3& o = t.getOrigin();
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out[4 * i + j] = (float)R[i][j];
    out[4 * i + 3] = (float)o[i];
  }
}

class SepfinderGeometricTools {
 public:
  SepfinderGeometricTools(ros::NodeHandle& n, const sensor_msgs::CameraInfo& info_l, const sensor_msgs::CameraInfo& info_r,
                          const std::string& frame_id, bool stereo_from_tf, int min_inliers) {
    // --- stereoCamGeometricTools.cpp:36-76: local transform (base -> left optical frame) and the stereo baseline ---
    tf::TransformListener listener;
    float local[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    tf::StampedTransform st;
    try {
      listener.waitForTransform(frame_id, info_l.header.frame_id, info_l.header.stamp, ros::Duration(1.0));
      listener.lookupTransform(frame_id, info_l.header.frame_id, info_l.header.stamp, st);
      rows_from_tf(st, local);
    } catch (tf::TransformException& ex) {
      ROS_ERROR("%s", ex.what());
    }
    // P = [fx 0 cx Tx; 0 fy cy 0; 0 0 1 0], Tx = -fx * baseline for the right camera (sensor_msgs/CameraInfo)
    double baseline = info_r.P[0] != 0.0 ? -info_r.P[3] / info_r.P[0] : 0.0;
    if (stereo_from_tf) {
      try {
        listener.lookupTransform(info_r.header.frame_id, info_l.header.frame_id, info_l.header.stamp, st);
        baseline = std::abs((double)st.getOrigin().x());
      } catch (tf::TransformException& ex) {
        ROS_ERROR("%s", ex.what());
      }
    }

    sf_params p;
    sf_default_params(&p);                       // rtabmap's compiled-in Vis/* defaults (myRegistrationVis.cpp:52-71)
    p.min_inliers = min_inliers;                 // ParametersPair(kVisMinInliers(), ...), :87
    int est = 0;
    n.param("estimation_type", est, 0);          // Vis/EstimationType: 0 = 3D->3D, 1 = PnP
    p.estimation_type = est;
    p.fx = info_l.P[0]; p.fy = info_l.P[5]; p.cx = info_l.P[2]; p.cy = info_l.P[6];
    p.image_width = (int32_t)info_l.width;
    p.image_height = (int32_t)info_l.height;
    std::memcpy(p.local_transform, local, sizeof(local));
    p.stereo_baseline = (float)baseline;
    if (sf_create(&p, /*device=*/0, &sf_) != SF_OK) throw std::runtime_error(sf_last_error(nullptr));

    cam_.fx = (float)p.fx; cam_.fy = (float)p.fy; cam_.cx = (float)p.cx; cam_.cy = (float)p.cy;
    cam_.cx_right = (float)info_r.P[2];
    cam_.baseline = (float)baseline;
    cam_.min_depth = 0.f; cam_.max_depth = 0.f;                     // Vis/MinDepth, Vis/MaxDepth (rtabmap: 0, 0)
    std::memcpy(cam_.local_transform, local, sizeof(local));

    // OpenCV's BRIEF point pairs, if the integrator provides them (512 tests x 4 int8: x1 y1 x2 y2), see ros/README.md
    std::string pattern_file;
    if (n.getParam("brief_pattern_file", pattern_file) && !pattern_file.empty()) {
      std::ifstream in(pattern_file.c_str(), std::ios::binary);
      std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
      const int bytes = (int)raw.size() / (8 * 4);                  // descriptor bytes = tests / 8
      if (raw.empty() || sf_brief_set_pattern(sf_, reinterpret_cast<const int8_t*>(raw.data()), bytes) != SF_OK)
        ROS_ERROR("brief_pattern_file %s: %s", pattern_file.c_str(), sf_last_error(sf_));
    } else {
      ROS_WARN("no brief_pattern_file: descriptors will not match a robot that runs the reference's OpenCV BRIEF");
    }
  }

  ~SepfinderGeometricTools() { if (sf_) sf_destroy(sf_); }

  // stereoCamGeometricTools.cpp:100-120
  bool getFeaturesAndDescriptor(multi_robot_separators::GetFeatsAndDesc::Request& req,
                                multi_robot_separators::GetFeatsAndDesc::Response& res) {
    const sensor_msgs::Image &l = req.image_left, &r = req.image_right;
    if (l.encoding != sensor_msgs::image_encodings::MONO8 || r.encoding != sensor_msgs::image_encodings::MONO8 ||
        l.width != r.width || l.height != r.height || l.step != r.step) {
      ROS_ERROR("get_features_and_descriptor: rectified MONO8 pair of one size expected (%s / %s)", l.encoding.c_str(),
                r.encoding.c_str());
      return false;       // (cv_bridge::toCvCopy(..., MONO8) converts other encodings in the reference: convert upstream)
    }
    int32_t bytes = 0;
    sf_brief_get_pattern(sf_, nullptr, 0, &bytes);
    const int cap = 32767;                                            // KeyPointVec.size is an int16
    std::vector<uint8_t> desc((size_t)cap * bytes);
    res.kpts3D.kpts3DVec.resize(cap);
    res.kpts.kptsVec.resize(cap);
    int32_t rows = 0;
    static_assert(sizeof(rtabmap_ros::Point3f) == 12 && sizeof(rtabmap_ros::KeyPoint) == sizeof(sf_keypoint), "message layout");
    const int rc = sf_get_features_and_descriptor(sf_, l.data.data(), r.data.data(), (int32_t)l.width, (int32_t)l.height,
                                                  (int32_t)l.step, &cam_, /*det=*/nullptr, /*flow=*/nullptr, desc.data(),
                                                  reinterpret_cast<float*>(res.kpts3D.kpts3DVec.data()),
                                                  reinterpret_cast<sf_keypoint*>(res.kpts.kptsVec.data()), cap, &rows,
                                                  /*slot_out=*/nullptr);
    if (rc != SF_OK) { ROS_ERROR("sf_get_features_and_descriptor: %s", sf_last_error(sf_)); return false; }
    rows = std::min(rows, cap);
    res.descriptors.rows = (uint16_t)rows;                             // MsgConversion.cpp:100-111
    res.descriptors.cols = (uint16_t)bytes;
    res.descriptors.data.assign(desc.begin(), desc.begin() + (size_t)rows * bytes);
    res.kpts3D.kpts3DVec.resize(rows);  res.kpts3D.size = (int16_t)rows;
    res.kpts.kptsVec.resize(rows);      res.kpts.size = (int16_t)rows;
    return true;
  }

  // stereoCamGeometricTools.cpp:122-176
  bool estimateTransformation(multi_robot_separators::EstTransform::Request& req,
                              multi_robot_separators::EstTransform::Response& res) {
    const sf_features from = view(req.descriptorsFrom, req.kptsFrom3D, req.kptsFrom);
    const sf_features to = view(req.descriptorsTo, req.kptsTo3D, req.kptsTo);
    sf_result out;
    const int rc = sf_estimate_transform(sf_, &from, &to, &out);
    if (rc != SF_OK) { ROS_ERROR("sf_estimate_transform: %s", sf_last_error(sf_)); return false; }
    for (int i = 0; i < 36; ++i) res.poseWithCov.covariance[i] = out.covariance[i];       // covToFloat64Msg
    res.poseWithCov.pose.position.x = out.position[0];                                    // transformToPoseMsg
    res.poseWithCov.pose.position.y = out.position[1];
    res.poseWithCov.pose.position.z = out.position[2];
    res.poseWithCov.pose.orientation.x = out.orientation[0];
    res.poseWithCov.pose.orientation.y = out.orientation[1];
    res.poseWithCov.pose.orientation.z = out.orientation[2];
    res.poseWithCov.pose.orientation.w = out.orientation[3];
    res.success = out.success != 0;
    return true;
  }

 private:
  static sf_features view(const multi_robot_separators::Descriptors& d, const multi_robot_separators::KeyPoint3DVec& p3,
                          const multi_robot_separators::KeyPointVec& kp) {
    sf_features f;
    std::memset(&f, 0, sizeof(f));
    f.desc = d.data.data(); f.rows = d.rows; f.cols = d.cols;
    f.xyz = reinterpret_cast<const float*>(p3.kpts3DVec.data()); f.n3d = p3.size;
    f.kpts = reinterpret_cast<const sf_keypoint*>(kp.kptsVec.data()); f.nkp = kp.size;
    return f;
  }

  sf_handle sf_ = nullptr;
  sf_stereo_camera cam_;
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "stereo_cam_geometric_tools_node");     // the reference's node name: launch files keep working
  ros::NodeHandle n;
  bool stereo_from_tf = false;
  std::string frame_id;
  int min_inliers = 5;
  if (!n.getParam("estimate_stereo_transform_from_tf", stereo_from_tf)) ROS_ERROR("Couldn't find estimate_stereo_transform_from_tf param");
  if (!n.getParam("frame_id", frame_id)) ROS_ERROR("Couldn't find frame ID");
  if (!n.getParam("separators_min_inliers", min_inliers)) ROS_ERROR("Couldn't find separators_min_inliers param");
  sensor_msgs::CameraInfoConstPtr il = ros::topic::waitForMessage<sensor_msgs::CameraInfo>("left/camera_info", n);
  sensor_msgs::CameraInfoConstPtr ir = ros::topic::waitForMessage<sensor_msgs::CameraInfo>("right/camera_info", n);
  SepfinderGeometricTools node(n, *il, *ir, frame_id, stereo_from_tf, min_inliers);
  ros::ServiceServer s1 = n.advertiseService("get_features_and_descriptor", &SepfinderGeometricTools::getFeaturesAndDescriptor, &node);
  ros::ServiceServer s2 = n.advertiseService("estimate_transformation", &SepfinderGeometricTools::estimateTransformation, &node);
  ROS_INFO("Stereo camera geometric tools ready (libsepfinder, ABI %d)", sf_abi_version());
  ros::spin();
  return 0;
}
