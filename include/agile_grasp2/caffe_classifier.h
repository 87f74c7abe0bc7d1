// caffe_classifier.h -- host mirror of Classifier (include/agile_grasp2/caffe_classifier.h:55-89,
// src/agile_grasp2/caffe_classifier.cpp): batched LeNet scoring (K5) on the GPU, no Caffe.
#ifndef AGILE_GRASP2_CAFFE_CLASSIFIER_H
#define AGILE_GRASP2_CAFFE_CLASSIFIER_H

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "agile_grasp2/hand_search.h"
#include "agile_grasp2/types.h"

typedef std::pair<std::string, float> Prediction;  // caffe_classifier.h:55

class Classifier {
 public:
  // caffe_classifier.cpp:4-35.  model_file: the prototxt is only checked for existence when given
  // (the architecture is fixed: caffe/test_1batch2.prototxt).  trained_file: either a Caffe
  // ".caffemodel" (binary NetParameter, read with a hand-written protobuf wire-format walker: new
  // `layer` and legacy V1 `layers` records, packed or unpacked float / double blobs) or the flat
  // ".ag2w" container (magic "AG2W", then the eight blobs conv1 w,b / conv2 w,b / ip1 w,b /
  // ip2 w,b as little-endian float32 in Caffe blob order).  label_file: one label per line
  // (caffe/labels.txt).  Never aborts: on failure ok() is false and ClassifyBatch returns an empty vector.
  Classifier(const std::string& model_file, const std::string& trained_file,
             const std::string& label_file);
  // The reader by itself: layers conv1, conv2, ip1, ip2 -> blobs[0..7]; false + *err on any mismatch.
  static bool readCaffeModel(const std::string& path, std::vector<float> blobs[8], std::string* err);

  // caffe_classifier.cpp:70-91: per image [(label_0, ip2[0]), (label_1, ip2[1])], raw logits.
  std::vector<std::vector<Prediction>> ClassifyBatch(const std::vector<ag2::Image>& imgs, int num_classes);
  // caffe_classifier.cpp:57-67 (single image; use_softmax is ignored: ip2 is returned, as in the
  // batch path the detector uses).
  std::vector<Prediction> Classify(const ag2::Image& img, bool use_softmax = false);

  bool ok() const { return ok_; }
  const std::string& error() const { return err_; }
  void setContext(std::shared_ptr<ag2::Context> ctx);
  const std::vector<float>& blob(int i) const { return blobs_[i]; }
  const std::vector<std::string>& labels() const { return labels_; }
  // process-unique id of this object's (immutable) blob set: what a context remembers having packed
  uint64_t generation() const { return generation_; }
  // ag2_lenet_load of the blobs into `ctx` unless that context already holds THIS classifier's weights
  int loadInto(ag2::Context& ctx) const;

 private:
  bool ensureLoaded();
  bool ok_ = false, uploaded_ = false;
  std::string err_;
  std::vector<std::string> labels_;
  std::vector<float> blobs_[8];
  uint64_t generation_ = 0;
  std::shared_ptr<ag2::Context> ctx_;
};

#endif  // AGILE_GRASP2_CAFFE_CLASSIFIER_H
