// importance_sampling.h -- host mirror of ImportanceSampling
// (include/agile_grasp2/importance_sampling.h, src/agile_grasp2/importance_sampling.cpp:30-157): a caller
// that re-enters the hot path num_iterations times with num_samples off-surface xyz samples drawn
// around the grasps found so far.  The loop is host control flow; every round is one ag2_detect on
// the cloud, search grid and normals that are already in HBM (the reference rebuilds the kd-tree
// and recomputes all normals each round, hand_search.cpp:11-29).
//
// Differences, all forced by reproducibility or by bugs in the reference:
//  * the draws use the library's counter-based generator keyed by Params::seed (the reference seeds
//    boost::mt19937 from wall time and mixes in rand(), importance_sampling.cpp:58-59,85,125);
//  * re-entries always use the xyz samples of the round (the reference's re-entry honours
//    use_incoming_samples_, so with the default `false` every round silently re-evaluates the
//    initial indices, grasp_detector.cpp:137);
//  * no plotting.
#ifndef AGILE_GRASP2_IMPORTANCE_SAMPLING_H
#define AGILE_GRASP2_IMPORTANCE_SAMPLING_H

#include <vector>

#include "agile_grasp2/grasp_detector.h"

class ImportanceSampling : public GraspDetector {
 public:
  static const int SUM = 1, MAX = 2;  // sampling methods, importance_sampling.cpp:5-6
  // standard parameters, importance_sampling.cpp:9-15
  static const int NUM_ITERATIONS = 5, NUM_SAMPLES = 50, NUM_INIT_SAMPLES = 100, METHOD = MAX;
  static constexpr double PROB_RAND_SAMPLES = 0.3, RADIUS = 0.02;

  explicit ImportanceSampling(const Params& params);  // importance_sampling.cpp:18-27

  // importance_sampling.cpp:30-118
  std::vector<GraspHypothesis> detectGraspPoses(const CloudCamera& cloud_cam_in);

  void setNumIterations(int n) { num_iterations_ = n; }
  void setNumSamplesPerIteration(int n) { num_samples_is_ = n; }
  void setProbRandSamples(double p) { prob_rand_samples_ = p; }
  void setRadius(double r) { radius_ = r; }
  void setSamplingMethod(int m) { sampling_method_ = m; }
  // the xyz samples of every round of the last call (3 x num_samples each), for inspection
  const std::vector<ag2::Matrix3Xd>& lastSampleRounds() const { return rounds_; }
  int lastInitialCount() const { return n_initial_; }

 private:
  double gaussian(uint64_t round, uint64_t* counter) const;
  int num_iterations_, num_samples_is_, num_init_samples_;
  double prob_rand_samples_, radius_;
  int sampling_method_;
  std::vector<ag2::Matrix3Xd> rounds_;
  int n_initial_ = 0;
};

#endif  // AGILE_GRASP2_IMPORTANCE_SAMPLING_H
