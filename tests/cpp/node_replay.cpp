// node_replay.cpp -- the call sequence of the reference's ROS node against the host mirror.
//
// The reference's GraspDetectionNode (src/nodes/grasp_detection_node.cpp) is the one caller of the
// hot path.  This file replays what its member functions do with GraspDetector / ImportanceSampling
// / CloudCamera / the messages -- constructor :16-66, detectGraspPosesInFile :98-120,
// detectGraspPosesInTopic :123-143, graspsServiceCallback :146-201, getSamplesInBall :204-213, the
// cloud / samples callbacks :216-293, createGraspListMsg :306-313 -- with every call spelled as the
// node spells it, so that it compiles only if the mirror headers offer those names and signatures.
// What a maintainer still has to change when switching the node over is exactly what is marked
// "ROS" below: the constructor argument (a Params struct instead of
// ros::NodeHandle&) and the conversion of sensor_msgs/PointCloud2 into a point cloud.
//
//   node_replay <cloud.f32> <params.txt>        (runs on the GPU; no arguments: usage, exit 2)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "agile_grasp2/cloud_camera.h"
#include "agile_grasp2/grasp_detector.h"
#include "agile_grasp2/importance_sampling.h"
#include "agile_grasp2/messages.h"

typedef ag2::PointCloudRGB PointCloudRGBA;  // nodes/grasp_detection_node.h:71 (same point type)

class GraspDetectionNode {
 public:
  // ROS: the reference takes ros::NodeHandle& and reads the parameters from it (:16-66)
  explicit GraspDetectionNode(const GraspDetector::Params& node, bool use_importance_sampling,
                              bool has_samples_topic)
      : cloud_(new PointCloudRGBA), cloud_normals_(new PointCloudNormal), size_left_cloud_(0), has_cloud_(false),
        has_normals_(false), has_samples_(true), use_importance_sampling_(use_importance_sampling) {
    importance_sampling_ = new ImportanceSampling(node);
    grasp_detector_ = new GraspDetector(node);
    if (has_samples_topic) {
      has_samples_ = false;
      grasp_detector_->setUseIncomingSamples(true);
    }
  }
  ~GraspDetectionNode() {
    delete grasp_detector_;
    delete importance_sampling_;
  }

  // one turn of run() (:69-95) without the ROS spin: returns the message the node would publish
  bool runOnce(agile_grasp2::GraspListMsg* grasps_msg) {
    if (has_cloud_ && ((grasp_detector_->getUseIncomingSamples() && has_samples_) ||
                       !grasp_detector_->getUseIncomingSamples())) {
      std::vector<GraspHypothesis> grasps = detectGraspPosesInTopic();
      *grasps_msg = createGraspListMsg(grasps);
      has_cloud_ = false;
      has_samples_ = false;
      return true;
    }
    return false;
  }

  std::vector<GraspHypothesis> detectGraspPosesInFile(const std::string& file_name_left,
                                                      const std::string& file_name_right) {
    CloudCamera* cloud_cam;
    if (file_name_right.length() == 0)
      cloud_cam = new CloudCamera(file_name_left);
    else
      cloud_cam = new CloudCamera(file_name_left, file_name_right);
    grasp_detector_->preprocessPointCloud(*cloud_cam);
    std::vector<GraspHypothesis> grasps;
    if (use_importance_sampling_)
      grasps = importance_sampling_->detectGraspPoses(*cloud_cam);
    else
      grasps = grasp_detector_->detectGraspPoses(*cloud_cam);
    delete cloud_cam;
    return grasps;
  }

  std::vector<GraspHypothesis> detectGraspPosesInTopic() {
    CloudCamera* cloud_cam;
    if (has_normals_)
      cloud_cam = new CloudCamera(cloud_normals_, size_left_cloud_);
    else
      cloud_cam = new CloudCamera(cloud_, size_left_cloud_);
    std::vector<GraspHypothesis> grasps;
    if (use_importance_sampling_)
      grasps = importance_sampling_->detectGraspPoses(*cloud_cam);
    else
      grasps = grasp_detector_->detectGraspPoses(*cloud_cam);
    delete cloud_cam;
    return grasps;
  }

  bool graspsServiceCallback(agile_grasp2::FindGrasps::Request& req, agile_grasp2::FindGrasps::Response& resp) {
    if (!has_cloud_) return false;
    CloudCamera* cloud_cam;
    if (has_normals_)
      cloud_cam = new CloudCamera(cloud_normals_, size_left_cloud_);
    else
      cloud_cam = new CloudCamera(cloud_, size_left_cloud_);
    if (req.grasps_signal == ALL_POINTS) {
      if (req.num_samples == 0) {
        grasp_detector_->preprocessPointCloud(*cloud_cam);
      } else {
        grasp_detector_->setNumSamples(req.num_samples);
        grasp_detector_->preprocessPointCloud(*cloud_cam);
      }
    } else if (req.grasps_signal == RADIUS) {
      ag2::PointXYZRGBA centroid;
      centroid.x = (float)req.centroid.x;
      centroid.y = (float)req.centroid.y;
      centroid.z = (float)req.centroid.z;
      std::vector<int> indices_ball = getSamplesInBall(cloud_cam->getCloudOriginal(), centroid, req.radius);
      cloud_cam->setSampleIndices(indices_ball);
    } else if (req.grasps_signal == INDICES) {
      std::vector<int> indices(req.indices.size());
      for (size_t i = 0; i < req.indices.size(); i++) indices[i] = (int)req.indices[i];
      cloud_cam->setSampleIndices(indices);
    }
    std::vector<GraspHypothesis> hands = grasp_detector_->detectGraspPoses(*cloud_cam);
    resp.grasps_msg = createGraspListMsg(hands);  // (the reference: "TODO: fill response", :196)
    delete cloud_cam;
    return true;
  }

  // ROS: the reference converts sensor_msgs/PointCloud2 with pcl::fromROSMsg (:216-236); the clouds
  // arrive here already converted
  void cloud_callback(const PointCloudRGBA::Ptr& msg) {
    if (!has_cloud_) {
      cloud_ = msg;
      size_left_cloud_ = (int)cloud_->size();
      has_cloud_ = true;
    }
  }
  void cloud_indexed_callback(const agile_grasp2::CloudIndexed& msg, const PointCloudRGBA::Ptr& converted) {
    if (!has_cloud_) {
      cloud_ = converted;
      size_left_cloud_ = (int)cloud_->size();
      grasp_detector_->setIndicesFromMsg(msg);
      has_cloud_ = true;
    }
  }
  void cloud_sized_callback(const agile_grasp2::CloudSized& msg, const PointCloudRGBA::Ptr& converted) {
    if (!has_cloud_) {
      cloud_ = converted;
      size_left_cloud_ = (int)msg.size_left.data;
      has_cloud_ = true;
    }
  }
  void samples_callback(const agile_grasp2::SamplesMsg& msg) {
    if (!has_samples_) {
      grasp_detector_->setSamplesMsg(msg);
      has_samples_ = true;
    }
  }
  agile_grasp2::GraspListMsg createGraspListMsg(const std::vector<GraspHypothesis>& hands) {
    agile_grasp2::GraspListMsg msg;
    for (size_t i = 0; i < hands.size(); i++) msg.grasps.push_back(hands[i].convertToGraspMsg());
    return msg;
  }
  GraspDetector& detector() { return *grasp_detector_; }
  void rearm() { has_cloud_ = false; }

  static const int ALL_POINTS = 0, RADIUS = 1, INDICES = 2;

 private:
  // the reference builds a kd-tree for one query (:204-213); same strict float test
  std::vector<int> getSamplesInBall(const PointCloudRGBA::Ptr& cloud, const ag2::PointXYZRGBA& centroid,
                                    float radius) {
    std::vector<int> indices;
    for (size_t i = 0; i < cloud->size(); i++) {
      const float dx = cloud->points[i].x - centroid.x, dy = cloud->points[i].y - centroid.y,
                  dz = cloud->points[i].z - centroid.z;
      if ((dx * dx + dy * dy) + dz * dz < radius * radius) indices.push_back((int)i);
    }
    return indices;
  }

  PointCloudRGBA::Ptr cloud_;
  PointCloudNormal::Ptr cloud_normals_;
  int size_left_cloud_;
  bool has_cloud_, has_normals_, has_samples_;
  bool use_importance_sampling_;
  GraspDetector* grasp_detector_;
  ImportanceSampling* importance_sampling_;
};

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s cloud.f32 params.txt\n", argv[0]);
    return 2;
  }
  std::ifstream cf(argv[1], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(cf)), std::istreambuf_iterator<char>());
  std::vector<float> xyz(raw.size() / 4);
  std::memcpy(xyz.data(), raw.data(), xyz.size() * 4);
  std::ifstream pf(argv[2]);
  const std::string ptext((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
  GraspDetector::Params prm;
  std::string err;
  if (!GraspDetector::Params::fromKeyValueText(ptext, &prm, &err)) {
    fprintf(stderr, "params: %s\n", err.c_str());
    return 2;
  }
  PointCloudRGBA::Ptr cloud(new PointCloudRGBA);
  cloud->points.resize(xyz.size() / 3);
  for (size_t i = 0; i < cloud->size(); i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  GraspDetectionNode node(prm, false, false);
  // the handle search of the detector is reachable and live (grasp_detector.h:123)
  const int inl = node.detector().getHandleSearch().getMinInliers();
  node.detector().getHandleSearch().setMinInliers(inl);

  // 1. CLOUD_INDEXED topic: every 40th point is a sample
  agile_grasp2::CloudIndexed cmsg;
  for (size_t i = 0; i < cloud->size(); i += 40) {
    agile_grasp2::Int64 v;
    v.data = (int64_t)i;
    cmsg.indices.push_back(v);
  }
  node.cloud_indexed_callback(cmsg, cloud);
  agile_grasp2::GraspListMsg topic_msg;
  if (!node.runOnce(&topic_msg)) return 3;

  // 2. the service, three request kinds
  size_t n_srv[3] = {0, 0, 0};
  for (int sig = 0; sig < 3; sig++) {
    node.rearm();
    node.cloud_callback(cloud);
    agile_grasp2::FindGrasps::Request req;
    agile_grasp2::FindGrasps::Response resp;
    req.grasps_signal = sig;
    req.num_samples = (sig == 0) ? 200 : 0;
    if (sig == GraspDetectionNode::RADIUS) {
      const ag2::PointXYZRGBA& c = cloud->points[cloud->size() / 2];
      req.centroid.x = c.x;
      req.centroid.y = c.y;
      req.centroid.z = c.z;
      req.radius = 0.05f;
    }
    if (sig == GraspDetectionNode::INDICES)
      for (const agile_grasp2::Int64& v : cmsg.indices) req.indices.push_back(v.data);
    if (!node.graspsServiceCallback(req, resp)) return 4;
    n_srv[sig] = resp.grasps_msg.grasps.size();
  }
  printf("node replay ok: topic %zu grasps; service all=%zu radius=%zu indices=%zu\n", topic_msg.grasps.size(),
         n_srv[0], n_srv[1], n_srv[2]);
  return 0;
}
