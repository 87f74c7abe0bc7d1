// messages.h -- plain-old-data mirrors of the reference's ROS wire surface (contract only; no ROS
// runtime is built).  Field for field:
//   msg/GraspMsg.msg:5-16, msg/GraspListMsg.msg:1-2, msg/SamplesMsg.msg, msg/CloudIndexed.msg,
//   msg/CloudSized.msg, srv/FindGrasps.srv:9-34.
// Topic / service names the node uses: publisher "grasps" (queue 10, grasp_detection_node.cpp:63),
// service "find_grasps" (:58), default cloud topic "/camera/depth_registered/points" (:31).
#ifndef AGILE_GRASP2_MESSAGES_H
#define AGILE_GRASP2_MESSAGES_H

#include <cstdint>
#include <string>
#include <vector>

namespace agile_grasp2 {

struct Point { double x = 0, y = 0, z = 0; };    // geometry_msgs/Point
struct Vector3 { double x = 0, y = 0, z = 0; };  // geometry_msgs/Vector3
struct Header {                                   // std_msgs/Header
  uint32_t seq = 0;
  uint32_t stamp_sec = 0, stamp_nsec = 0;
  std::string frame_id;
};

struct GraspMsg {  // msg/GraspMsg.msg
  Point surface, bottom, top;
  Vector3 axis, approach, binormal;
  float width = 0.f;  // std_msgs/Float32
  float score = 0.f;  // std_msgs/Float32
};

struct GraspListMsg {  // msg/GraspListMsg.msg
  Header header;
  std::vector<GraspMsg> grasps;
};

struct SamplesMsg {  // msg/SamplesMsg.msg
  Header header;
  std::vector<Point> samples;
};

struct Int64 { int64_t data = 0; };               // std_msgs/Int64

// sensor_msgs/PointCloud2, the fields the node reads (grasp_detection_node.cpp:216-275): the cloud
// itself travels as a CloudCamera on this side, so only the layout description is mirrored.
struct PointField {
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
};
struct PointCloud2 {
  Header header;
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = false;
};

struct CloudIndexed {  // msg/CloudIndexed.msg:1-2
  PointCloud2 cloud;
  std::vector<Int64> indices;   // read as msg.indices[i].data, grasp_detector.cpp:353-361
};

struct CloudSized {  // msg/CloudSized.msg:1-2
  PointCloud2 cloud;
  Int64 size_left;
};

struct FindGraspsRequest {  // srv/FindGrasps.srv:9-30
  int32_t grasps_signal = 0;       // 0: whole cloud, 1: r-ball, 2: indices
  int32_t num_samples = 0;         // 0: launch-file value
  int32_t min_handle_inliers = 0;
  bool calculate_antipodal = false;
  Vector3 centroid;
  float radius = 0.f;
  std::vector<int64_t> indices;
};

struct FindGraspsResponse {  // srv/FindGrasps.srv:33-34
  GraspListMsg grasps_msg;
};
struct FindGrasps {  // the names roscpp generates for the service (grasp_detection_node.h:157)
  typedef FindGraspsRequest Request;
  typedef FindGraspsResponse Response;
};

// Fixed 152-byte little-endian payload of one GraspMsg (3 Points + 3 Vector3 as f64, width and
// score as f32) -- the body a ROS serializer emits for this message.
inline void serialize(const GraspMsg& m, std::vector<uint8_t>& out) {
  const double d[18] = {m.surface.x, m.surface.y, m.surface.z, m.bottom.x, m.bottom.y, m.bottom.z,
                        m.top.x, m.top.y, m.top.z, m.axis.x, m.axis.y, m.axis.z,
                        m.approach.x, m.approach.y, m.approach.z, m.binormal.x, m.binormal.y, m.binormal.z};
  const uint8_t* p = reinterpret_cast<const uint8_t*>(d);
  out.insert(out.end(), p, p + sizeof(d));
  const float f[2] = {m.width, m.score};
  p = reinterpret_cast<const uint8_t*>(f);
  out.insert(out.end(), p, p + sizeof(f));
}

}  // namespace agile_grasp2

#endif  // AGILE_GRASP2_MESSAGES_H
