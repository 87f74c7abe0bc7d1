// handle_search.h -- host mirror of HandleSearch (include/agile_grasp2/handle_search.h:57-170,
// src/agile_grasp2/handle_search.cpp:4-80): grasp clustering behind the scoring.
// findClusters(hand_list) -- the form every caller in the reference uses (grasp_detector.cpp:231,
// importance_sampling.cpp:107) -- runs on the GPU (ag2_find_clusters).  remove_inliers = true is
// order-dependent across hands and stays plain C++.  findHandles / Handle are only reached from
// src/tests/test_cnn.cpp:164 and are not built.
#ifndef AGILE_GRASP2_HANDLE_SEARCH_H
#define AGILE_GRASP2_HANDLE_SEARCH_H

#include <memory>
#include <vector>

#include "agile_grasp2/grasp_hypothesis.h"
#include "agile_grasp2/hand_search.h"

class HandleSearch {
 public:
  // handle_search.h:66-67
  std::vector<GraspHypothesis> findClusters(const std::vector<GraspHypothesis>& hand_list,
                                            bool remove_inliers = false);

  int getMinInliers() const { return min_inliers_; }                           // :78-81
  void setMinInliers(int min_inliers) { min_inliers_ = min_inliers; }          // :83-86
  void setMinLength(double min_length) { min_length_ = min_length; }           // :88-91
  void setReuseInliers(bool reuse_inliers) { reuse_inliers_ = reuse_inliers; }  // :93-96

  void setContext(std::shared_ptr<ag2::Context> ctx) { ctx_ = std::move(ctx); }

 private:
  bool reuse_inliers_ = false;
  int min_inliers_ = 0;
  double min_length_ = 0.0;
  std::shared_ptr<ag2::Context> ctx_;
};

#endif  // AGILE_GRASP2_HANDLE_SEARCH_H
