// test_host_api.cpp -- drives the C++ host mirror (include/agile_grasp2/*.h) the way the reference's
// GraspDetectionNode drives the reference classes (src/nodes/grasp_detection_node.cpp:98-143):
//   CloudCamera -> GraspDetector::detectGraspPoses, and the stage-by-stage path
//   HandSearch::generateHypotheses -> Learning::createGraspImages -> Classifier::ClassifyBatch.
// Inputs and outputs are flat binary files so that tests/test_cpp_host.py can compare them with the
// C-ABI results and with the oracle.
//
//   test_host_api <cloud.f32> <idx.i32> <params.txt> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "agile_grasp2/caffe_classifier.h"
#include "agile_grasp2/cloud_camera.h"
#include "agile_grasp2/grasp_detector.h"
#include "agile_grasp2/hand_search.h"
#include "agile_grasp2/importance_sampling.h"
#include "agile_grasp2/learning.h"

template <class T>
static std::vector<T> read_all(const char* path) {
  std::ifstream f(path, std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<T> out(raw.size() / sizeof(T));
  std::memcpy(out.data(), raw.data(), out.size() * sizeof(T));
  return out;
}

template <class T>
static void put(std::ofstream& f, const T* p, size_t n) {
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

// test_host_api --preprocess <raw.f32> <params.txt> <out.bin>: the node's own sequence for a raw
// cloud (grasp_detection_node.cpp:98-143): preprocessPointCloud -> detectGraspPoses.
static int run_preprocess(const char* raw_path, const char* params_path, const char* out_path) {
  const std::vector<float> xyz = read_all<float>(raw_path);
  std::ifstream pf(params_path);
  const std::string ptext((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
  GraspDetector::Params prm;
  std::string err;
  if (!GraspDetector::Params::fromKeyValueText(ptext, &prm, &err)) {
    fprintf(stderr, "params: %s\n", err.c_str());
    return 2;
  }
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(xyz.size() / 3);
  for (size_t i = 0; i < cloud->size(); i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  CloudCamera cc(cloud, (int)cloud->size());
  GraspDetector det(prm);
  det.preprocessPointCloud(cc);
  const std::vector<GraspHypothesis> sel = det.detectGraspPoses(cc);
  std::ofstream out(out_path, std::ios::binary);
  const int64_t m = (int64_t)cc.getCloudProcessed()->size(), k = (int64_t)cc.getSampleIndices().size(),
                ns = (int64_t)sel.size();
  put(out, &m, 1);
  for (const ag2::PointXYZRGBA& p : cc.getCloudProcessed()->points) put(out, &p.x, 3);
  put(out, &k, 1);
  for (int i : cc.getSampleIndices()) {
    const int32_t v = i;
    put(out, &v, 1);
  }
  put(out, &ns, 1);
  for (const GraspHypothesis& h : sel) {
    const int32_t so[2] = {h.getSampleSlot(), h.getOrientation()};
    const double sc = h.getScore();
    put(out, so, 2);
    put(out, &sc, 1);
  }
  printf("preprocess ok: %lld raw -> %lld points, %lld samples, %lld grasps, preprocess %.3f ms\n",
         (long long)cloud->size(), (long long)m, (long long)k, (long long)ns,
         (double)det.lastStageTimes().preprocess_ms);
  return 0;
}

// test_host_api --frames <raw.f32> <params.txt> <out.bin>: GraspDetector::detectGraspPosesInFrame three times on
// the same raw cloud (the first call runs step by step, the second at fixed shapes, the third replays the graph)
// beside preprocessPointCloud + detectGraspPoses on a second detector.  Output: for each of the four runs
// int64 n, then {int32 slot, int32 orientation, double score} per grasp.
static int run_frames(const char* raw_path, const char* params_path, const char* out_path) {
  const std::vector<float> xyz = read_all<float>(raw_path);
  std::ifstream pf(params_path);
  const std::string ptext((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
  GraspDetector::Params prm;
  std::string err;
  if (!GraspDetector::Params::fromKeyValueText(ptext, &prm, &err)) {
    fprintf(stderr, "params: %s\n", err.c_str());
    return 2;
  }
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(xyz.size() / 3);
  for (size_t i = 0; i < cloud->size(); i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  std::ofstream out(out_path, std::ios::binary);
  auto dump = [&](const std::vector<GraspHypothesis>& sel) {
    const int64_t ns = (int64_t)sel.size();
    put(out, &ns, 1);
    for (const GraspHypothesis& h : sel) {
      const int32_t so[2] = {h.getSampleSlot(), h.getOrientation()};
      const double sc = h.getScore();
      put(out, so, 2);
      put(out, &sc, 1);
    }
  };
  GraspDetector frames(prm);
  size_t total = 0;
  for (int k = 0; k < 3; k++) {
    const std::vector<GraspHypothesis> sel = frames.detectGraspPosesInFrame(cloud);
    if (!frames.lastError().empty()) {
      fprintf(stderr, "frame %d: %s\n", k, frames.lastError().c_str());
      return 3;
    }
    total += sel.size();
    dump(sel);
  }
  GraspDetector two(prm);
  CloudCamera cc(cloud, (int)cloud->size());
  two.preprocessPointCloud(cc);
  dump(two.detectGraspPoses(cc));
  printf("frames ok: %zu grasps in three frames\n", total);
  return 0;
}

// test_host_api --importance <cloud.f32> <idx.i32> <params.txt> <out.bin>:
// ImportanceSampling::detectGraspPoses (importance_sampling.cpp:30-118) on a preprocessed cloud
static int run_importance(const char* cloud_path, const char* idx_path, const char* params_path,
                          const char* out_path) {
  const std::vector<float> xyz = read_all<float>(cloud_path);
  const std::vector<int32_t> idx = read_all<int32_t>(idx_path);
  std::ifstream pf(params_path);
  const std::string ptext((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
  GraspDetector::Params prm;
  std::string err;
  if (!GraspDetector::Params::fromKeyValueText(ptext, &prm, &err)) {
    fprintf(stderr, "params: %s\n", err.c_str());
    return 2;
  }
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(xyz.size() / 3);
  for (size_t i = 0; i < cloud->size(); i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  CloudCamera cc(cloud, (int)cloud->size());
  cc.setSampleIndices(std::vector<int>(idx.begin(), idx.end()));
  ImportanceSampling is(prm);
  is.setNumIterations(3);
  is.setNumSamplesPerIteration(40);
  const std::vector<GraspHypothesis> hands = is.detectGraspPoses(cc);
  std::ofstream out(out_path, std::ios::binary);
  const int64_t n0 = is.lastInitialCount(), nr = (int64_t)is.lastSampleRounds().size(), nh = (int64_t)hands.size();
  put(out, &n0, 1);
  put(out, &nr, 1);
  for (const ag2::Matrix3Xd& m : is.lastSampleRounds()) {
    const int64_t s = m.cols();
    put(out, &s, 1);
    put(out, m.data(), (size_t)(3 * s));
  }
  put(out, &nh, 1);
  for (const GraspHypothesis& h : hands) {
    const int32_t so[2] = {h.getSampleSlot(), h.getOrientation()};
    const double v[4] = {h.getScore(), h.getGraspBottom()(0), h.getGraspBottom()(1), h.getGraspBottom()(2)};
    put(out, so, 2);
    put(out, v, 4);
  }
  printf("importance ok: %lld initial, %lld rounds, %lld hands\n", (long long)n0, (long long)nr, (long long)nh);
  return 0;
}

// test_host_api --caffemodel <file> <out.bin>: the eight blobs, concatenated float32
static int run_caffemodel(const char* path, const char* out_path) {
  std::vector<float> blobs[8];
  std::string err;
  if (!Classifier::readCaffeModel(path, blobs, &err)) {
    fprintf(stderr, "caffemodel: %s\n", err.c_str());
    return 3;
  }
  std::ofstream out(out_path, std::ios::binary);
  for (int b = 0; b < 8; b++) put(out, blobs[b].data(), blobs[b].size());
  return 0;
}

// test_host_api --pcd <file> <out.bin>: int64 n, then n x (x, y, z) float32 (CloudCamera(filename))
static int run_pcd(const char* path, const char* out_path) {
  CloudCamera cc{std::string(path)};
  std::ofstream out(out_path, std::ios::binary);
  const int64_t n = (int64_t)cc.getCloudProcessed()->size();
  put(out, &n, 1);
  for (const ag2::PointXYZRGBA& p : cc.getCloudProcessed()->points) put(out, &p.x, 3);
  const int64_t rows = cc.getCameraSource().rows(), cols = cc.getCameraSource().cols();
  put(out, &rows, 1);
  put(out, &cols, 1);
  return 0;
}

// test_host_api --constants <out.txt>: what the mirror derives from the reference-held constants,
// no GPU touched: default camera poses (grasp_detector.cpp:108-126) and the finger slots / hand
// angles / deepen depths of the launch-file hand (finger_hand.cpp:7-12, hand_search.cpp:179-180).
static int run_constants(const char* out_path) {
  GraspDetector::Params prm;  // defaults: no camera_pose => Baxter matrices
  prm.antipodal_mode = GraspDetector::NONE;  // no classifier files needed
  GraspDetector det(prm);
  ag2::Matrix4d l, r;
  det.cameraPoses(&l, &r);
  FILE* f = fopen(out_path, "w");
  if (!f) return 3;
  fprintf(f, "cam_tf_left");
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) fprintf(f, " %.17g", l(i, j));
  fprintf(f, "\ncam_tf_right");
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) fprintf(f, " %.17g", r(i, j));
  HandSearch::Parameters hp;
  hp.init_bite_ = 0.01;  // launch/file_detect_grasps.launch
  const ag2_params ap = HandSearch::toAbiParams(hp, 1);
  double fs[20], ang[32], dep[32];
  int32_t nd = 0;
  if (ag2_hand_constants(&ap, fs, ang, dep, &nd)) return 4;
  fprintf(f, "\nfinger_spacing");
  for (int i = 0; i < 20; i++) fprintf(f, " %.17g", fs[i]);
  fprintf(f, "\nangles");
  for (int i = 0; i < ap.num_orientations; i++) fprintf(f, " %.17g", ang[i]);
  fprintf(f, "\ndepths");
  for (int i = 0; i < nd; i++) fprintf(f, " %.17g", dep[i]);
  HandSearch seven(0.01, 0.09, 0.06, 0.02, 0.01, 4, 500);  // hand_search.h:114-118
  fprintf(f, "\nseven_arg_ctor %.17g %.17g %d %d\n", seven.getParameters().nn_radius_taubin_,
          seven.getParameters().nn_radius_hands_, seven.getParameters().num_threads_,
          seven.getParameters().num_samples_);
  fclose(f);
  return 0;
}

// test_host_api --modes cloud.f32 idx.i32 params.txt out.bin: GraspDetector::detectGraspPoses in the
// antipodal_mode / min_inliers the parameter file gives (grasp_detector.cpp:163-252: NONE returns the
// pruned hypotheses before the clustering and the selection, GEOMETRIC keeps the full-antipodal ones,
// clusters them when min_inliers > 0 and takes num_selected).  Output: int64 n, then per hand
// {int32 slot, int32 orientation, int32 full_antipodal, int32 half_antipodal, double score, double bottom[3]}.
static int run_modes(const char* cloud_path, const char* idx_path, const char* params_path, const char* out_path,
                     const char* normals_path) {
  const std::vector<float> xyz = read_all<float>(cloud_path);
  const std::vector<float> nrm = normals_path ? read_all<float>(normals_path) : std::vector<float>();
  const std::vector<int32_t> idx = read_all<int32_t>(idx_path);
  std::ifstream pf(params_path);
  const std::string ptext((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
  GraspDetector::Params prm;
  std::string err;
  if (!GraspDetector::Params::fromKeyValueText(ptext, &prm, &err)) {
    fprintf(stderr, "params: %s\n", err.c_str());
    return 2;
  }
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(xyz.size() / 3);
  for (size_t i = 0; i < cloud->size(); i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  // a cloud that brings its normals (cloud_camera.cpp:4-32) -- or one without (:35-51)
  PointCloudNormal::Ptr cloud_n(new PointCloudNormal);
  if (!nrm.empty()) {
    if (nrm.size() != xyz.size()) return 2;
    cloud_n->points.resize(xyz.size() / 3);
    for (size_t i = 0; i < cloud_n->size(); i++) {
      cloud_n->points[i].x = xyz[3 * i];
      cloud_n->points[i].y = xyz[3 * i + 1];
      cloud_n->points[i].z = xyz[3 * i + 2];
      cloud_n->points[i].normal_x = nrm[3 * i];
      cloud_n->points[i].normal_y = nrm[3 * i + 1];
      cloud_n->points[i].normal_z = nrm[3 * i + 2];
    }
  }
  CloudCamera cc = nrm.empty() ? CloudCamera(cloud, (int)cloud->size()) : CloudCamera(cloud_n, (int)cloud_n->size());
  cc.setSampleIndices(std::vector<int>(idx.begin(), idx.end()));
  GraspDetector det(prm);
  std::vector<GraspHypothesis> hands = det.detectGraspPoses(cc);
  if (!det.lastError().empty()) {
    fprintf(stderr, "detect: %s\n", det.lastError().c_str());
    return 3;
  }
  // a second call on the same detector (contexts, packed weights and peers are reused): the same hands
  const std::vector<GraspHypothesis> again = det.detectGraspPoses(cc);
  if (again.size() != hands.size()) {
    fprintf(stderr, "second call: %zu hands, first %zu\n", again.size(), hands.size());
    return 4;
  }
  for (size_t i = 0; i < hands.size(); i++)
    if (again[i].getSampleSlot() != hands[i].getSampleSlot() || again[i].getOrientation() != hands[i].getOrientation() ||
        again[i].getScore() != hands[i].getScore()) {
      fprintf(stderr, "second call differs at hand %zu\n", i);
      return 4;
    }
  printf("tile_points");
  for (size_t m : det.lastTilePoints()) printf(" %zu", m);
  printf("\n");
  std::ofstream out(out_path, std::ios::binary);
  const int64_t n = (int64_t)hands.size();
  put(out, &n, 1);
  for (const GraspHypothesis& h : hands) {
    const int32_t ids[4] = {h.getSampleSlot(), h.getOrientation(), h.isFullAntipodal() ? 1 : 0,
                            h.isHalfAntipodal() ? 1 : 0};
    put(out, ids, 4);
    const double v[4] = {h.getScore(), h.getGraspBottom()(0), h.getGraspBottom()(1), h.getGraspBottom()(2)};
    put(out, v, 4);
  }
  printf("modes ok: mode %d, min_inliers %d, %lld hands\n", prm.antipodal_mode, prm.min_inliers, (long long)n);
  return 0;
}

int main(int argc, char** argv) {
  if ((argc == 6 || argc == 7) && std::string(argv[1]) == "--modes")
    return run_modes(argv[2], argv[3], argv[4], argv[5], argc == 7 ? argv[6] : nullptr);
  if (argc == 3 && std::string(argv[1]) == "--constants") return run_constants(argv[2]);
  if (argc == 5 && std::string(argv[1]) == "--preprocess") return run_preprocess(argv[2], argv[3], argv[4]);
  if (argc == 5 && std::string(argv[1]) == "--frames") return run_frames(argv[2], argv[3], argv[4]);
  if (argc == 4 && std::string(argv[1]) == "--caffemodel") return run_caffemodel(argv[2], argv[3]);
  if (argc == 4 && std::string(argv[1]) == "--pcd") return run_pcd(argv[2], argv[3]);
  if (argc == 6 && std::string(argv[1]) == "--importance") return run_importance(argv[2], argv[3], argv[4], argv[5]);
  if (argc != 5) {
    fprintf(stderr, "usage: %s cloud.f32 idx.i32 params.txt out.bin | --preprocess raw.f32 params.txt out.bin\n",
            argv[0]);
    return 2;
  }
  const std::vector<float> xyz = read_all<float>(argv[1]);
  const std::vector<int32_t> idx = read_all<int32_t>(argv[2]);
  std::ifstream pf(argv[3]);
  const std::string ptext((std::istreambuf_iterator<char>(pf)), std::istreambuf_iterator<char>());
  GraspDetector::Params prm;
  std::string err;
  if (!GraspDetector::Params::fromKeyValueText(ptext, &prm, &err)) {
    fprintf(stderr, "params: %s\n", err.c_str());
    return 2;
  }

  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(xyz.size() / 3);
  for (size_t i = 0; i < cloud->size(); i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  CloudCamera cc(cloud, (int)cloud->size());  // one camera
  cc.setSampleIndices(std::vector<int>(idx.begin(), idx.end()));

  // ---- stage by stage, as src/tests/test_cnn.cpp does ----------------------------------------
  HandSearch::Parameters hp;
  hp.nn_radius_taubin_ = prm.nn_radius_taubin;
  hp.nn_radius_hands_ = prm.nn_radius_hands;
  hp.num_orientations_ = prm.num_orientations;
  hp.finger_width_ = prm.finger_width;
  hp.hand_outer_diameter_ = prm.hand_outer_diameter;
  hp.hand_depth_ = prm.hand_depth;
  hp.hand_height_ = prm.hand_height;
  hp.init_bite_ = prm.init_bite;
  if (prm.camera_pose.size() != 16) {
    fprintf(stderr, "params: camera_pose needs 16 values, got %zu\n", prm.camera_pose.size());
    return 2;
  }
  for (int k = 0; k < 3; k++) hp.cam_tf_left_(k, 3) = hp.cam_tf_right_(k, 3) = prm.camera_pose[4 * k + 3];
  HandSearch hs(hp);
  hs.setSeed(prm.seed);
  const std::vector<GraspHypothesis> hyps = hs.generateHypotheses(cc, 0, false);
  Learning learning(60, 1);
  learning.setContext(hs.context());
  const std::vector<ag2::Image> images = learning.createGraspImages(hyps, ag2::Matrix3Xd());
  Classifier clf(prm.model_file, prm.trained_file, prm.label_file);
  if (!clf.ok()) {
    fprintf(stderr, "classifier: %s\n", clf.error().c_str());
    return 3;
  }
  clf.setContext(hs.context());
  const std::vector<std::vector<Prediction>> pred = clf.ClassifyBatch(images, 2);

  // ---- the detector entry the node calls -------------------------------------------------------
  GraspDetector det(prm);
  const std::vector<GraspHypothesis> sel = det.detectGraspPoses(cc);
  const agile_grasp2::GraspListMsg msg = GraspDetector::createGraspListMsg(sel);

  std::ofstream out(argv[4], std::ios::binary);
  const int64_t nh = (int64_t)hyps.size(), ni = (int64_t)images.size(), np = (int64_t)pred.size(),
                ns = (int64_t)sel.size();
  put(out, &nh, 1);
  for (const GraspHypothesis& h : hyps) {
    const double rec[20] = {h.getAxis()(0), h.getAxis()(1), h.getAxis()(2), h.getApproach()(0), h.getApproach()(1),
                            h.getApproach()(2), h.getBinormal()(0), h.getBinormal()(1), h.getBinormal()(2),
                            h.getGraspSurface()(0), h.getGraspSurface()(1), h.getGraspSurface()(2),
                            h.getGraspBottom()(0), h.getGraspBottom()(1), h.getGraspBottom()(2),
                            h.getGraspTop()(0), h.getGraspTop()(1), h.getGraspTop()(2), h.getGraspWidth(),
                            (double)(h.isHalfAntipodal() + 2 * h.isFullAntipodal())};
    put(out, rec, 20);
    const int64_t p = h.getPointsForLearning().cols();
    put(out, &p, 1);
  }
  put(out, &ni, 1);
  for (const ag2::Image& im : images) put(out, im.data.data(), im.data.size());
  put(out, &np, 1);
  for (const auto& pr : pred) {
    const float l[2] = {pr[0].second, pr[1].second};
    put(out, l, 2);
  }
  put(out, &ns, 1);
  std::vector<uint8_t> wire;
  for (const agile_grasp2::GraspMsg& g : msg.grasps) agile_grasp2::serialize(g, wire);
  put(out, wire.data(), wire.size());
  for (const GraspHypothesis& h : sel) {
    const int32_t so[2] = {h.getSampleSlot(), h.getOrientation()};
    put(out, so, 2);
  }
  printf("host api ok: %lld hypotheses, %lld images, %lld selected, label0=%s\n", (long long)nh, (long long)ni,
         (long long)ns, pred.empty() ? "-" : pred[0][0].first.c_str());
  return 0;
}
