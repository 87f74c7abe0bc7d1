// grasp_hypothesis.h -- host mirror of GraspHypothesis
// (include/agile_grasp2/grasp_hypothesis.h:62-312, src/agile_grasp2/grasp_hypothesis.cpp:40-52).
#ifndef AGILE_GRASP2_GRASP_HYPOTHESIS_H
#define AGILE_GRASP2_GRASP_HYPOTHESIS_H

#include <vector>

#include "ag2_c.h"
#include "agile_grasp2/messages.h"
#include "agile_grasp2/types.h"

class GraspHypothesis {
 public:
  GraspHypothesis() : cam_source_(-1), grasp_width_(0), score_(0), full_antipodal_(false), half_antipodal_(false) {}
  // grasp_hypothesis.h:71-79
  GraspHypothesis(const ag2::Vector3d& axis, const ag2::Vector3d& approach, const ag2::Vector3d& binormal,
                  const ag2::Vector3d& surface, const ag2::Vector3d& bottom, const ag2::Vector3d& top,
                  double width, const ag2::Matrix3Xd& points_for_learning,
                  const ag2::Matrix3Xd& normals_for_learning, const ag2::MatrixXi& camera_source_for_learning)
      : cam_source_(-1), axis_(axis), approach_(approach), binormal_(binormal), grasp_surface_(surface),
        grasp_bottom_(bottom), grasp_top_(top), grasp_width_(width), score_(0.0), full_antipodal_(false),
        half_antipodal_(false), points_for_learning_(points_for_learning),
        normals_for_learning_(normals_for_learning), camera_source_for_learning_(camera_source_for_learning) {}
  // from a C-ABI record (+ optional point lists)
  explicit GraspHypothesis(const ag2_hypothesis& r);
  ag2_hypothesis toRecord() const;  // the fixed part, for C-ABI calls that take hypotheses

  agile_grasp2::GraspMsg convertToGraspMsg() const;  // grasp_hypothesis.cpp:40-52

  const ag2::Vector3d& getApproach() const { return approach_; }
  const ag2::Vector3d& getAxis() const { return axis_; }
  const ag2::Vector3d& getBinormal() const { return binormal_; }
  bool isFullAntipodal() const { return full_antipodal_; }
  bool isHalfAntipodal() const { return half_antipodal_; }
  const ag2::Vector3d& getGraspBottom() const { return grasp_bottom_; }
  const ag2::Vector3d& getGraspSurface() const { return grasp_surface_; }
  const ag2::Vector3d& getGraspTop() const { return grasp_top_; }
  double getGraspWidth() const { return grasp_width_; }
  double getScore() const { return score_; }
  int getCamSource() const { return cam_source_; }
  const ag2::Matrix3Xd& getPointsForLearning() const { return points_for_learning_; }
  const ag2::Matrix3Xd& getNormalsForLearning() const { return normals_for_learning_; }
  const std::vector<int>& getIndicesPointsForLearningCam1() const { return indices_cam1_; }
  const std::vector<int>& getIndicesPointsForLearningCam2() const { return indices_cam2_; }
  void setFullAntipodal(bool b) { full_antipodal_ = b; }
  void setHalfAntipodal(bool b) { half_antipodal_ = b; }
  void setGraspWidth(double w) { grasp_width_ = w; }
  void setGraspBottom(const ag2::Vector3d& v) { grasp_bottom_ = v; }
  void setGraspSurface(const ag2::Vector3d& v) { grasp_surface_ = v; }
  void setGraspTop(const ag2::Vector3d& v) { grasp_top_ = v; }
  void setScore(double s) { score_ = s; }
  void setPointsForLearning(ag2::Matrix3Xd pts, ag2::Matrix3Xd nrm) {
    points_for_learning_ = std::move(pts);
    normals_for_learning_ = std::move(nrm);
  }
  // position in the fixed-slot candidate table (sample slot, orientation); -1 if unknown
  int getSampleSlot() const { return sample_slot_; }
  int getOrientation() const { return orientation_; }

 protected:
  int cam_source_;
  ag2::Vector3d axis_, approach_, binormal_, grasp_surface_, grasp_bottom_, grasp_top_;
  double grasp_width_, score_;
  bool full_antipodal_, half_antipodal_;
  ag2::Matrix3Xd points_for_learning_, normals_for_learning_;
  ag2::MatrixXi camera_source_for_learning_;
  std::vector<int> indices_cam1_, indices_cam2_;
  int sample_slot_ = -1, orientation_ = -1;
};

#endif  // AGILE_GRASP2_GRASP_HYPOTHESIS_H
