// learning.h -- host mirror of Learning (include/agile_grasp2/learning.h:61-183,
// src/agile_grasp2/learning.cpp:4-33,143-209): renders 60x60x3 grasp images (K4) on the GPU.
#ifndef AGILE_GRASP2_LEARNING_H
#define AGILE_GRASP2_LEARNING_H

#include <memory>
#include <vector>

#include "agile_grasp2/grasp_hypothesis.h"
#include "agile_grasp2/hand_search.h"
#include "agile_grasp2/types.h"

class Learning {
 public:
  Learning() : num_horizontal_cells_(60), num_vertical_cells_(60), num_threads_(1) {}
  // learning.h:75.  Only size 60 is supported (the network's input geometry,
  // grasp_detector.cpp:56); other sizes yield an empty result + a message.
  Learning(int size, int num_threads)
      : num_horizontal_cells_(size), num_vertical_cells_(size), num_threads_(num_threads) {}

  // learning.cpp:4-33.  cam_pos is unused by the reference as well; is_plotting / is_storing have no
  // effect here.  One CV_8UC3-like ag2::Image (60 x 60 x 3, HWC) per hypothesis.
  std::vector<ag2::Image> createGraspImages(const std::vector<GraspHypothesis>& hands_list,
                                            const ag2::Matrix3Xd& cam_pos, bool is_plotting = false,
                                            bool is_storing = false);

  void setContext(std::shared_ptr<ag2::Context> ctx) { ctx_ = std::move(ctx); }

 private:
  int num_horizontal_cells_, num_vertical_cells_, num_threads_;
  std::shared_ptr<ag2::Context> ctx_;
};

#endif  // AGILE_GRASP2_LEARNING_H
