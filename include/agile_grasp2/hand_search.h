// hand_search.h -- host mirror of HandSearch (include/agile_grasp2/hand_search.h:55-181,
// src/agile_grasp2/hand_search.cpp).  Same Parameters fields, same setters, same
// generateHypotheses signature; the body runs K0-K3 on the GPU through the C-ABI.
#ifndef AGILE_GRASP2_HAND_SEARCH_H
#define AGILE_GRASP2_HAND_SEARCH_H

#include <memory>
#include <vector>

#include "ag2_c.h"
#include "agile_grasp2/cloud_camera.h"
#include "agile_grasp2/grasp_hypothesis.h"
#include "agile_grasp2/types.h"

namespace ag2 {
// RAII owner of one ag2_ctx (one GPU).  Shared by HandSearch / Learning / Classifier inside a
// GraspDetector so that the cloud, normals and point lists stay resident between stages.
class Context {
 public:
  Context(const ag2_params& p, int device);
  ~Context();
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  ag2_ctx* get() const { return c_; }
  bool ok() const { return c_ != nullptr; }
  const ag2_params& params() const { return p_; }
  // Which Classifier's blobs ag2_lenet_load last packed into this context (Classifier::generation(); 0 = none):
  // packing is 14.5 MB of host work + H2D, done once per (context, classifier), not once per call.
  uint64_t weights_generation = 0;
 private:
  ag2_ctx* c_;
  ag2_params p_;
};
}  // namespace ag2

class HandSearch {
 public:
  struct Parameters {  // hand_search.h:72-91
    double nn_radius_taubin_ = 0.01;
    int num_threads_ = 1;  // kept for API parity; the GPU path ignores it
    int num_samples_ = 1000;
    ag2::Matrix4d cam_tf_left_, cam_tf_right_;
    double nn_radius_hands_ = 0.1;
    int num_orientations_ = 8;
    double finger_width_ = 0.01;
    double hand_outer_diameter_ = 0.09;
    double hand_depth_ = 0.06;
    double hand_height_ = 0.02;
    double init_bite_ = 0.015;
  };

  HandSearch() {}
  explicit HandSearch(const Parameters& params) : params_(params) {}
  // hand_search.h:114-118: the 7-argument form fixes nn_radius_taubin_ = 0.03 and nn_radius_hands_ =
  // 0.08; it leaves num_orientations_ and the camera poses uninitialised in the reference -- here
  // they keep the Parameters defaults (8 orientations, identity poses).
  HandSearch(double finger_width, double hand_outer_diameter, double hand_depth, double hand_height,
             double init_bite, int num_threads, int num_samples) {
    params_.finger_width_ = finger_width;
    params_.hand_outer_diameter_ = hand_outer_diameter;
    params_.hand_depth_ = hand_depth;
    params_.hand_height_ = hand_height;
    params_.init_bite_ = init_bite;
    params_.num_threads_ = num_threads;
    params_.num_samples_ = num_samples;
    params_.nn_radius_taubin_ = 0.03;
    params_.nn_radius_hands_ = 0.08;
  }

  // hand_search.cpp:4-61.  antipodal_mode and forces_PSD are ignored by the reference body too;
  // plots_* have no effect (no visualisation is built).  Returns hypotheses in sample order, each
  // with getPointsForLearning()/getNormalsForLearning() filled.  Empty vector + message on stderr
  // on any error (the reference's observable behaviour for an empty cloud).
  std::vector<GraspHypothesis> generateHypotheses(const CloudCamera& cloud_cam, int antipodal_mode,
                                                  bool use_samples, bool forces_PSD = false,
                                                  bool plots_normals = false, bool plots_samples = false);

  void setParameters(const Parameters& params) { params_ = params; ctx_.reset(); }  // :64-80
  void setCamTfLeft(const ag2::Matrix4d& m) { params_.cam_tf_left_ = m; ctx_.reset(); }
  void setCamTfRight(const ag2::Matrix4d& m) { params_.cam_tf_right_ = m; ctx_.reset(); }
  const Parameters& getParameters() const { return params_; }

  // not in the reference: seed of the neighbour draw that replaces rand() (hand_search.cpp:130),
  // the GPU to use, and access to the context for the stages that follow.
  void setSeed(uint64_t seed) { seed_ = seed; }
  void setDevice(int device) { device_ = device; ctx_.reset(); }
  std::shared_ptr<ag2::Context> context() const { return ctx_; }
  // Fill an ag2_params from Parameters (+ defaults for everything HandSearch does not own).
  static ag2_params toAbiParams(const Parameters& p, int n_cams);
  // Upload the processed cloud of cloud_cam (xyz, camera source, optional normals) into ctx and
  // compute normals if absent.  Returns 0 or an AG2_ERR_* code.
  static int uploadCloud(ag2_ctx* ctx, const CloudCamera& cloud_cam);

 private:
  Parameters params_;
  uint64_t seed_ = 0;
  int device_ = 0;
  std::shared_ptr<ag2::Context> ctx_;
};

#endif  // AGILE_GRASP2_HAND_SEARCH_H
