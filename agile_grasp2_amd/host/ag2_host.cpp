// ag2_host.cpp -- C++ host mirror of the reference's public classes for the hot path, sitting on
// the C-ABI of libag2hip.so (include/ag2_c.h).  Same class / method names, argument meaning and
// error behaviour (empty result + message, never an abort) as the reference; see the headers under
// include/agile_grasp2/ for the reference file:line each method mirrors.  No compute happens here
// beyond preprocessing and glue: normals, frames, hand search, images and LeNet run on the GPU.
#include <algorithm>
#include <atomic>
#include <array>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>

#include "agile_grasp2/caffe_classifier.h"
#include "agile_grasp2/cloud_camera.h"
#include "agile_grasp2/grasp_detector.h"
#include "agile_grasp2/grasp_hypothesis.h"
#include "agile_grasp2/hand_search.h"
#include "agile_grasp2/handle_search.h"
#include "agile_grasp2/importance_sampling.h"
#include "agile_grasp2/learning.h"

using ag2::Matrix3Xd;
using ag2::Matrix4d;
using ag2::MatrixXi;
using ag2::Vector3d;

// ------------------------------------------------------------------------------------------------
// small matrix helpers
// ------------------------------------------------------------------------------------------------
Matrix4d Matrix4d::operator*(const Matrix4d& o) const {
  Matrix4d r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0.0;
      for (int k = 0; k < 4; k++) s += m[i][k] * o.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}

Matrix4d Matrix4d::inverse() const {  // Gauss-Jordan with partial pivoting
  double a[4][8];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      a[i][j] = m[i][j];
      a[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; c++) {
    int piv = c;
    for (int r = c + 1; r < 4; r++)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 8; j++) std::swap(a[c][j], a[piv][j]);
    const double d = a[c][c];
    if (d == 0.0) return Matrix4d();
    for (int j = 0; j < 8; j++) a[c][j] /= d;
    for (int r = 0; r < 4; r++)
      if (r != c) {
        const double f = a[r][c];
        for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
      }
  }
  Matrix4d r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.m[i][j] = a[i][4 + j];
  return r;
}

namespace {
// the library's counter-based draw (csrc/ag2_device.h draw_u64), for the host twins of GPU steps
uint64_t draw_u64(uint64_t seed, uint64_t slot, uint64_t j) {
  uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (slot + 1ull));
  x += 0xD1B54A32D192ED03ull * (j + 1ull);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// CloudCamera
// ------------------------------------------------------------------------------------------------
CloudCamera::CloudCamera() : cloud_processed_(new PointCloudRGB), cloud_original_(new PointCloudRGB) {}

static MatrixXi source_matrix(size_t n, int size_left) {
  if ((size_t)size_left == n) return MatrixXi::Zero(1, (int)n);  // cloud_camera.cpp:14-17 (zeros!)
  MatrixXi s = MatrixXi::Zero(2, (int)n);
  for (int i = 0; i < size_left; i++) s(0, i) = 1;
  for (size_t i = (size_t)size_left; i < n; i++) s(1, (int)i) = 1;
  return s;
}

CloudCamera::CloudCamera(const PointCloudNormal::Ptr& cloud, int size_left_cloud)
    : cloud_processed_(new PointCloudRGB), cloud_original_(new PointCloudRGB) {
  const size_t n = cloud->size();
  cloud_original_->points.resize(n);
  for (size_t i = 0; i < n; i++) {
    ag2::PointXYZRGBA& d = cloud_original_->points[i];
    const ag2::PointXYZRGBNormal& s = cloud->points[i];
    d.x = s.x; d.y = s.y; d.z = s.z; d.rgba = s.rgba;
  }
  *cloud_processed_ = *cloud_original_;
  camera_source_ = source_matrix(n, size_left_cloud);
  normals_.resize(3, (int)n);
  for (size_t i = 0; i < n; i++) {
    normals_(0, (int)i) = cloud->points[i].normal_x;
    normals_(1, (int)i) = cloud->points[i].normal_y;
    normals_(2, (int)i) = cloud->points[i].normal_z;
  }
}

CloudCamera::CloudCamera(const PointCloudRGB::Ptr& cloud, int size_left_cloud)
    : cloud_processed_(cloud), cloud_original_(cloud) {
  camera_source_ = source_matrix(cloud->size(), size_left_cloud);
}

CloudCamera::CloudCamera(const std::string& filename) : cloud_processed_(new PointCloudRGB) {
  cloud_processed_ = loadPointCloudFromFile(filename);
  cloud_original_ = cloud_processed_;
  camera_source_ = MatrixXi::Ones(1, (int)cloud_processed_->size());  // cloud_camera.cpp:59
}

CloudCamera::CloudCamera(const std::string& filename_left, const std::string& filename_right)
    : cloud_processed_(new PointCloudRGB) {
  PointCloudRGB::Ptr l = loadPointCloudFromFile(filename_left), r = loadPointCloudFromFile(filename_right);
  cloud_processed_->points = l->points;
  cloud_processed_->points.insert(cloud_processed_->points.end(), r->points.begin(), r->points.end());
  cloud_original_ = cloud_processed_;
  camera_source_ = MatrixXi::Zero(2, (int)cloud_processed_->size());
  for (size_t i = 0; i < l->size(); i++) camera_source_(0, (int)i) = 1;
  for (size_t i = 0; i < r->size(); i++) camera_source_(1, (int)(l->size() + i)) = 1;
}

void CloudCamera::filterWorkspace(const std::vector<double>& ws) {
  if (ws.size() < 6) return;
  std::vector<int> keep;
  for (size_t i = 0; i < cloud_processed_->size(); i++) {
    const ag2::PointXYZRGBA& p = cloud_processed_->points[i];
    if (p.x > ws[0] && p.x < ws[1] && p.y > ws[2] && p.y < ws[3] && p.z > ws[4] && p.z < ws[5])
      keep.push_back((int)i);
  }
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(keep.size());
  MatrixXi src(camera_source_.rows(), (int)keep.size());
  for (size_t i = 0; i < keep.size(); i++) {
    cloud->points[i] = cloud_processed_->points[keep[i]];
    for (int c = 0; c < camera_source_.rows(); c++) src(c, (int)i) = camera_source_(c, keep[i]);
  }
  if (normals_.cols() > 0) {
    Matrix3Xd n(3, (int)keep.size());
    for (size_t i = 0; i < keep.size(); i++)
      for (int k = 0; k < 3; k++) n(k, (int)i) = normals_(k, keep[i]);
    normals_ = n;
  }
  cloud_processed_ = cloud;
  camera_source_ = src;
}

void CloudCamera::voxelizeCloud(double cell_size) {
  const size_t n = cloud_processed_->size();
  if (n == 0) return;
  float mn[3] = {cloud_processed_->points[0].x, cloud_processed_->points[0].y, cloud_processed_->points[0].z};
  for (size_t i = 1; i < n; i++) {
    const ag2::PointXYZRGBA& p = cloud_processed_->points[i];
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
  }
  const float cell = (float)cell_size;
  // std::set with the reference's comparator == map ordered by (ix, iy, iz); value = first index
  std::map<std::array<int, 3>, int> bins;
  for (size_t i = 0; i < n; i++) {
    const ag2::PointXYZRGBA& p = cloud_processed_->points[i];
    const std::array<int, 3> v = {(int)std::floor((p.x - mn[0]) / cell), (int)std::floor((p.y - mn[1]) / cell),
                                  (int)std::floor((p.z - mn[2]) / cell)};
    bins.insert(std::make_pair(v, (int)i));  // keeps the first point that hit the voxel
  }
  // :140-141 pushes the first-hit indices in scan order, :149-152 reads them by position in set order
  std::vector<int> idx_cam_source;
  idx_cam_source.reserve(bins.size());
  for (const auto& kv : bins) idx_cam_source.push_back(kv.second);
  std::sort(idx_cam_source.begin(), idx_cam_source.end());
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(bins.size());
  MatrixXi src(camera_source_.rows(), (int)bins.size());
  int i = 0;
  for (const auto& kv : bins) {
    ag2::PointXYZRGBA& d = cloud->points[i];
    d.x = (float)kv.first[0] * cell + mn[0];
    d.y = (float)kv.first[1] * cell + mn[1];
    d.z = (float)kv.first[2] * cell + mn[2];
    for (int c = 0; c < camera_source_.rows(); c++)
      src(c, i) = (camera_source_(c, idx_cam_source[i]) == 1) ? 1 : 0;
    i++;
  }
  cloud_processed_ = cloud;
  camera_source_ = src;
  normals_ = Matrix3Xd();  // the reference does not carry normals through voxelisation either
}

void CloudCamera::subsampleUniformly(int num_samples, uint64_t seed) {
  const int n = (int)cloud_processed_->size();
  const int k = std::min(num_samples, n);
  std::vector<int> idx(k);
  if (k == n) {
    for (int i = 0; i < n; i++) idx[i] = i;
  } else {  // the k smallest keys (draw(seed, i), i): same draw as ag2_subsample_uniformly
    std::vector<std::pair<uint64_t, int>> keys(n);
    for (int i = 0; i < n; i++) keys[i] = std::make_pair(draw_u64(seed, 0xFFFFFFFFFFFFFFF0ull, (uint64_t)i), i);
    std::nth_element(keys.begin(), keys.begin() + k, keys.end());
    for (int i = 0; i < k; i++) idx[i] = keys[i].second;
    std::sort(idx.begin(), idx.end());  // pcl::RandomSample returns ascending indices
  }
  sample_indices_ = idx;
}

void CloudCamera::adoptProcessed(const PointCloudRGB::Ptr& cloud, const MatrixXi& camera_source,
                                 const Matrix3Xd& normals) {
  cloud_processed_ = cloud;
  camera_source_ = camera_source;
  normals_ = normals;
}

void CloudCamera::subsampleSamples(const agile_grasp2::SamplesMsg& msg, int num_samples, uint64_t seed) {
  const int n = (int)msg.samples.size();
  std::vector<int> seq(n);
  for (int i = 0; i < n; i++) seq[i] = i;
  int k = n;
  if (num_samples < n) {
    uint64_t s = seed ^ 0x1234567887654321ull;
    for (int i = n - 1; i > 0; i--) std::swap(seq[i], seq[(int)(splitmix(s) % (uint64_t)(i + 1))]);
    k = num_samples;
  }
  samples_.resize(3, k);
  for (int i = 0; i < k; i++) {
    samples_(0, i) = msg.samples[seq[i]].x;
    samples_(1, i) = msg.samples[seq[i]].y;
    samples_(2, i) = msg.samples[seq[i]].z;
  }
}

void CloudCamera::setSamples(const agile_grasp2::SamplesMsg& msg) {
  samples_.resize(3, (int)msg.samples.size());
  for (size_t i = 0; i < msg.samples.size(); i++) {
    samples_(0, (int)i) = msg.samples[i].x;
    samples_(1, (int)i) = msg.samples[i].y;
    samples_(2, (int)i) = msg.samples[i].z;
  }
}

// PCD reader (ASCII / binary, un-compressed): fields x y z (float32) and optionally rgb / rgba.
PointCloudRGB::Ptr CloudCamera::loadPointCloudFromFile(const std::string& filename) {
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  std::ifstream f(filename.c_str(), std::ios::binary);
  if (!f) {
    fprintf(stderr, "Couldn't read .pcd file: %s\n", filename.c_str());  // cloud_camera.cpp:236
    return cloud;
  }
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  size_t npts = 0;
  std::string data_mode, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string key;
    is >> key;
    if (key == "FIELDS") { std::string s; while (is >> s) fields.push_back(s); }
    else if (key == "SIZE") { int v; while (is >> v) sizes.push_back(v); }
    else if (key == "TYPE") { char c; while (is >> c) types.push_back(c); }
    else if (key == "COUNT") { int v; while (is >> v) counts.push_back(v); }
    else if (key == "POINTS") { is >> npts; }
    else if (key == "DATA") { is >> data_mode; break; }
  }
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (fields.empty() || sizes.size() != fields.size() || (data_mode != "ascii" && data_mode != "binary")) {
    fprintf(stderr, "Couldn't read .pcd file: %s (unsupported header)\n", filename.c_str());
    return cloud;
  }
  int off = 0, ox = -1, oy = -1, oz = -1, orgb = -1, ix = -1, iy = -1, iz = -1, irgb = -1, col = 0;
  for (size_t k = 0; k < fields.size(); k++) {
    if (fields[k] == "x") { ox = off; ix = col; }
    if (fields[k] == "y") { oy = off; iy = col; }
    if (fields[k] == "z") { oz = off; iz = col; }
    if (fields[k] == "rgb" || fields[k] == "rgba") { orgb = off; irgb = col; }
    off += sizes[k] * counts[k];
    col += counts[k];
  }
  if (ox < 0 || oy < 0 || oz < 0) {
    fprintf(stderr, "Couldn't read .pcd file: %s (no x y z fields)\n", filename.c_str());
    return cloud;
  }
  cloud->points.resize(npts);
  if (data_mode == "binary") {
    std::vector<char> rec((size_t)off);
    for (size_t i = 0; i < npts; i++) {
      if (!f.read(rec.data(), off)) { cloud->points.resize(i); break; }
      ag2::PointXYZRGBA& p = cloud->points[i];
      std::memcpy(&p.x, &rec[ox], 4); std::memcpy(&p.y, &rec[oy], 4); std::memcpy(&p.z, &rec[oz], 4);
      if (orgb >= 0) std::memcpy(&p.rgba, &rec[orgb], 4);
    }
  } else {
    for (size_t i = 0; i < npts; i++) {
      if (!std::getline(f, line)) { cloud->points.resize(i); break; }
      std::istringstream is(line);
      std::vector<std::string> tok;
      std::string s;
      while (is >> s) tok.push_back(s);
      if ((int)tok.size() < col) { cloud->points.resize(i); break; }
      ag2::PointXYZRGBA& p = cloud->points[i];
      p.x = std::strtof(tok[ix].c_str(), nullptr);
      p.y = std::strtof(tok[iy].c_str(), nullptr);
      p.z = std::strtof(tok[iz].c_str(), nullptr);
      if (irgb >= 0) {
        const float v = std::strtof(tok[irgb].c_str(), nullptr);
        std::memcpy(&p.rgba, &v, 4);
      }
    }
  }
  return cloud;
}

// ------------------------------------------------------------------------------------------------
// GraspHypothesis
// ------------------------------------------------------------------------------------------------
GraspHypothesis::GraspHypothesis(const ag2_hypothesis& r)
    : cam_source_(-1), axis_(r.axis[0], r.axis[1], r.axis[2]),
      approach_(r.approach[0], r.approach[1], r.approach[2]),
      binormal_(r.binormal[0], r.binormal[1], r.binormal[2]),
      grasp_surface_(r.surface[0], r.surface[1], r.surface[2]),
      grasp_bottom_(r.bottom[0], r.bottom[1], r.bottom[2]), grasp_top_(r.top[0], r.top[1], r.top[2]),
      grasp_width_(r.width), score_(r.score), full_antipodal_(r.full_antipodal != 0),
      half_antipodal_(r.half_antipodal != 0), sample_slot_(r.sample_slot), orientation_(r.orientation) {}

ag2_hypothesis GraspHypothesis::toRecord() const {
  ag2_hypothesis r;
  memset(&r, 0, sizeof(r));
  for (int k = 0; k < 3; k++) {
    r.axis[k] = axis_(k);
    r.approach[k] = approach_(k);
    r.binormal[k] = binormal_(k);
    r.surface[k] = grasp_surface_(k);
    r.bottom[k] = grasp_bottom_(k);
    r.top[k] = grasp_top_(k);
  }
  r.width = grasp_width_;
  r.score = score_;
  r.sample_slot = sample_slot_;
  r.orientation = orientation_;
  r.half_antipodal = half_antipodal_ ? 1 : 0;
  r.full_antipodal = full_antipodal_ ? 1 : 0;
  r.n_points = points_for_learning_.cols();
  return r;
}

// ------------------------------------------------------------------------------------------------
// HandleSearch
// ------------------------------------------------------------------------------------------------
std::vector<GraspHypothesis> HandleSearch::findClusters(const std::vector<GraspHypothesis>& hand_list,
                                                        bool remove_inliers) {
  std::vector<GraspHypothesis> out;
  const size_t n = hand_list.size();
  if (n == 0) return out;
  if (min_inliers_ < 1) {
    fprintf(stderr, "HandleSearch::findClusters: min_inliers must be >= 1 (the reference divides by the inlier "
                    "count, handle_search.cpp:66)\n");
    return out;
  }
  auto moved = [&](size_t i, const double delta[3], double score) {
    GraspHypothesis h = hand_list[i];  // :69-75
    h.setGraspSurface(Vector3d(h.getGraspSurface()(0) + delta[0], h.getGraspSurface()(1) + delta[1],
                               h.getGraspSurface()(2) + delta[2]));
    h.setGraspBottom(Vector3d(h.getGraspBottom()(0) + delta[0], h.getGraspBottom()(1) + delta[1],
                              h.getGraspBottom()(2) + delta[2]));
    h.setGraspTop(Vector3d(h.getGraspTop()(0) + delta[0], h.getGraspTop()(1) + delta[1],
                           h.getGraspTop()(2) + delta[2]));
    h.setScore(score);
    return h;
  };
  if (!remove_inliers) {  // every caller in the reference: on the GPU
    if (!ctx_) {
      ag2_params p;
      ag2_default_params(&p);
      ctx_.reset(new ag2::Context(p, 0));
    }
    if (!ctx_->ok()) {
      fprintf(stderr, "HandleSearch: could not create a GPU context (no CPU fallback)\n");
      return out;
    }
    std::vector<ag2_hypothesis> recs(n), res(n);
    for (size_t i = 0; i < n; i++) {
      recs[i] = hand_list[i].toRecord();
      recs[i].n_points = (int32_t)i;  // carried through verbatim: position in hand_list
    }
    size_t k = 0;
    if (ag2_find_clusters(ctx_->get(), recs.data(), n, min_inliers_, res.data(), n, &k)) {
      fprintf(stderr, "HandleSearch::findClusters: %s\n", ag2_last_error(ctx_->get()));
      return out;
    }
    out.reserve(k);
    for (size_t q = 0; q < k; q++) {
      const size_t i = (size_t)res[q].n_points;
      const double delta[3] = {0.0, 0.0, 0.0};
      GraspHypothesis h = moved(i, delta, res[q].score);
      h.setGraspSurface(Vector3d(res[q].surface[0], res[q].surface[1], res[q].surface[2]));
      h.setGraspBottom(Vector3d(res[q].bottom[0], res[q].bottom[1], res[q].bottom[2]));
      h.setGraspTop(Vector3d(res[q].top[0], res[q].top[1], res[q].top[2]));
      out.push_back(h);
    }
    return out;
  }
  // remove_inliers = true: has_used couples the iterations (handle_search.cpp:14-22,31,58-59)
  const double cos_thresh = std::cos(15.0 * M_PI / 180.0);
  std::vector<char> has_used(n, 0);
  for (size_t i = 0; i < n; i++) {
    const Vector3d &a = hand_list[i].getAxis(), &b = hand_list[i].getGraspBottom();
    double P[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) P[r][c] = ((r == c) ? 1.0 : 0.0) - a(r) * a(c);
    int cnt = 0;
    double sum[3] = {0.0, 0.0, 0.0}, ssum = 0.0;
    for (size_t j = 0; j < n; j++) {
      if (i == j || has_used[j]) continue;
      const Vector3d &t = hand_list[j].getAxis(), &u = hand_list[j].getGraspBottom();
      const double aligned = (a(0) * t(0) + a(1) * t(1)) + a(2) * t(2);
      const double d0 = b(0) - u(0), d1 = b(1) - u(1), d2 = b(2) - u(2);
      const double mag = std::sqrt((d0 * d0 + d1 * d1) + d2 * d2);
      const double p0 = (P[0][0] * d0 + P[0][1] * d1) + P[0][2] * d2;
      const double p1 = (P[1][0] * d0 + P[1][1] * d1) + P[1][2] * d2;
      const double p2 = (P[2][0] * d0 + P[2][1] * d1) + P[2][2] * d2;
      const double pm = std::sqrt((p0 * p0 + p1 * p1) + p2 * p2);
      if (std::fabs(aligned) > cos_thresh && mag <= 0.05 && pm <= 0.005) {
        cnt++;
        sum[0] += u(0); sum[1] += u(1); sum[2] += u(2);
        ssum += hand_list[j].getScore();
        has_used[j] = 1;
      }
    }
    if (cnt >= min_inliers_) {
      const double delta[3] = {sum[0] / cnt - b(0), sum[1] / cnt - b(1), sum[2] / cnt - b(2)};
      out.push_back(moved(i, delta, ssum / cnt));
    }
  }
  return out;
}

agile_grasp2::GraspMsg GraspHypothesis::convertToGraspMsg() const {
  agile_grasp2::GraspMsg m;
  m.surface = {grasp_surface_(0), grasp_surface_(1), grasp_surface_(2)};
  m.bottom = {grasp_bottom_(0), grasp_bottom_(1), grasp_bottom_(2)};
  m.top = {grasp_top_(0), grasp_top_(1), grasp_top_(2)};
  m.axis = {axis_(0), axis_(1), axis_(2)};
  m.approach = {approach_(0), approach_(1), approach_(2)};
  m.binormal = {binormal_(0), binormal_(1), binormal_(2)};
  m.width = (float)grasp_width_;
  m.score = (float)score_;
  return m;
}

// ------------------------------------------------------------------------------------------------
// Context / HandSearch
// ------------------------------------------------------------------------------------------------
ag2::Context::Context(const ag2_params& p, int device) : c_(ag2_create(&p, device)), p_(p) {}
ag2::Context::~Context() { if (c_) ag2_destroy(c_); }

ag2_params HandSearch::toAbiParams(const Parameters& hp, int n_cams) {
  ag2_params p;
  ag2_default_params(&p);
  p.finger_width = hp.finger_width_;
  p.hand_outer_diameter = hp.hand_outer_diameter_;
  p.hand_depth = hp.hand_depth_;
  p.hand_height = hp.hand_height_;
  p.init_bite = hp.init_bite_;
  p.nn_radius_taubin = hp.nn_radius_taubin_;
  p.nn_radius_hands = hp.nn_radius_hands_;
  p.num_orientations = hp.num_orientations_;
  p.num_threads = hp.num_threads_;
  p.n_cams = n_cams;
  for (int k = 0; k < 3; k++) {  // local_frame.cpp:8-12: translation column of the camera poses
    p.cam_origin[0][k] = hp.cam_tf_left_(k, 3);
    p.cam_origin[1][k] = hp.cam_tf_right_(k, 3);
  }
  return p;
}

int HandSearch::uploadCloud(ag2_ctx* ctx, const CloudCamera& cc) {
  const PointCloudRGB::Ptr& cloud = cc.getCloudProcessed();
  const size_t n = cloud->size();
  const MatrixXi& src = cc.getCameraSource();
  const Matrix3Xd& nrm = cc.getNormals();
  const bool has_n = (size_t)nrm.cols() == n && n > 0;
  int rc = ag2_set_cloud(ctx, n ? &cloud->points[0].x : nullptr, n, sizeof(ag2::PointXYZRGBA),
                         src.cols() == (int)n && n ? src.data() : nullptr, std::max(1, src.rows()),
                         has_n ? nrm.data() : nullptr);
  if (rc) return rc;
  if (!has_n) rc = ag2_compute_normals(ctx);  // hand_search.cpp:20-29
  return rc;
}

std::vector<GraspHypothesis> HandSearch::generateHypotheses(const CloudCamera& cloud_cam, int, bool use_samples,
                                                            bool, bool, bool) {
  std::vector<GraspHypothesis> out;
  const int n_cams = std::max(1, cloud_cam.getCameraSource().rows());
  if (!ctx_ || ctx_->params().n_cams != n_cams) {
    ctx_.reset(new ag2::Context(toAbiParams(params_, n_cams), device_));
    if (!ctx_->ok()) {
      fprintf(stderr, "HandSearch: could not create a GPU context (no CPU fallback)\n");
      ctx_.reset();
      return out;
    }
  }
  ag2_ctx* c = ctx_->get();
  int rc = uploadCloud(c, cloud_cam);
  const size_t s = use_samples ? (size_t)cloud_cam.getSamples().cols() : cloud_cam.getSampleIndices().size();
  std::vector<ag2_hypothesis> recs(std::max<size_t>(1, s * (size_t)params_.num_orientations_));
  size_t n = 0;
  if (!rc) {
    std::vector<int32_t> idx(cloud_cam.getSampleIndices().begin(), cloud_cam.getSampleIndices().end());
    rc = ag2_generate_hypotheses(c, use_samples ? nullptr : idx.data(),
                                 use_samples ? cloud_cam.getSamples().data() : nullptr, s, 0, seed_,
                                 recs.data(), recs.size(), &n);
  }
  if (rc) {
    fprintf(stderr, "HandSearch::generateHypotheses: %s\n", ag2_last_error(c));
    return out;
  }
  out.reserve(n);
  for (size_t h = 0; h < n; h++) {
    GraspHypothesis g(recs[h]);
    Matrix3Xd pts(3, recs[h].n_points), nrm(3, recs[h].n_points);
    if (ag2_hyp_points(c, h, pts.data(), nrm.data()) == 0) g.setPointsForLearning(std::move(pts), std::move(nrm));
    out.push_back(std::move(g));
  }
  return out;
}

// ------------------------------------------------------------------------------------------------
// Learning
// ------------------------------------------------------------------------------------------------
std::vector<ag2::Image> Learning::createGraspImages(const std::vector<GraspHypothesis>& hands,
                                                    const Matrix3Xd&, bool, bool) {
  std::vector<ag2::Image> out;
  if (hands.empty()) return out;
  if (num_horizontal_cells_ != 60 || num_vertical_cells_ != 60) {
    fprintf(stderr, "Learning: only 60x60 grasp images are supported\n");
    return out;
  }
  if (!ctx_) {
    ag2_params p;
    ag2_default_params(&p);
    ctx_.reset(new ag2::Context(p, 0));
  }
  if (!ctx_->ok()) {
    fprintf(stderr, "Learning: could not create a GPU context (no CPU fallback)\n");
    return out;
  }
  std::vector<int64_t> offs(hands.size() + 1, 0);
  for (size_t i = 0; i < hands.size(); i++) offs[i + 1] = offs[i] + hands[i].getPointsForLearning().cols();
  std::vector<double> pts((size_t)std::max<int64_t>(offs.back(), 1) * 3), nrm(pts.size());
  for (size_t i = 0; i < hands.size(); i++) {
    const Matrix3Xd& p = hands[i].getPointsForLearning();
    const Matrix3Xd& q = hands[i].getNormalsForLearning();
    std::copy(p.d.begin(), p.d.end(), pts.begin() + offs[i] * 3);
    std::copy(q.d.begin(), q.d.end(), nrm.begin() + offs[i] * 3);
  }
  std::vector<uint8_t> raw(hands.size() * 10800);
  const int rc = ag2_render_images_from_points(ctx_->get(), hands.size(), offs.data(), pts.data(), nrm.data(),
                                               raw.data());
  if (rc) {
    fprintf(stderr, "Learning::createGraspImages: %s\n", ag2_last_error(ctx_->get()));
    return out;
  }
  out.resize(hands.size());
  for (size_t i = 0; i < hands.size(); i++) {
    out[i].rows = out[i].cols = 60;
    out[i].chans = 3;
    out[i].data.assign(raw.begin() + i * 10800, raw.begin() + (i + 1) * 10800);
  }
  return out;
}

// ------------------------------------------------------------------------------------------------
// Classifier
// ------------------------------------------------------------------------------------------------
static const size_t kBlobSizes[8] = {20 * 3 * 25, 20, 50 * 20 * 25, 50, 500 * 7200, 500, 2 * 500, 2};

// ---- .caffemodel reader: a walker over the protobuf wire format (protoc is not needed) ---------------
// Field numbers from BVLC caffe.proto: NetParameter.layer = 100 (LayerParameter: name = 1, blobs = 7),
// legacy NetParameter.layers = 2 (V1LayerParameter: name = 4, blobs = 6); BlobProto: data = 5 (float,
// packed or not), double_data = 8, shape = 7, legacy num/channels/height/width = 1..4 (ignored: only
// the element count is checked against the fixed architecture caffe/test_1batch2.prototxt).
namespace {
struct Pb {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  Pb(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
  bool more() const { return ok && p < end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) { ok = false; return 0; }
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  Pb sub() {  // length-delimited payload
    const uint64_t len = varint();
    if (!ok || len > (uint64_t)(end - p)) { ok = false; return Pb(end, end); }
    Pb s(p, p + len);
    p += len;
    return s;
  }
  void skip(int wire) {
    if (wire == 0) (void)varint();
    else if (wire == 1) { if (end - p < 8) ok = false; else p += 8; }
    else if (wire == 2) (void)sub();
    else if (wire == 5) { if (end - p < 4) ok = false; else p += 4; }
    else ok = false;  // groups are not used by caffe.proto
  }
};

bool parse_blob(Pb b, std::vector<float>* out) {
  out->clear();
  while (b.more()) {
    const uint64_t key = b.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == 5 && wire == 2) {         // packed repeated float
      Pb d = b.sub();
      const size_t n = (size_t)(d.end - d.p) / 4;
      const size_t at = out->size();
      out->resize(at + n);
      if (n) std::memcpy(out->data() + at, d.p, n * 4);  // little-endian host (x86-64)
    } else if (field == 5 && wire == 5) {  // one unpacked float
      if (b.end - b.p < 4) return false;
      float v;
      std::memcpy(&v, b.p, 4);
      b.p += 4;
      out->push_back(v);
    } else if (field == 8 && wire == 2) {  // packed double_data
      Pb d = b.sub();
      for (; d.end - d.p >= 8; d.p += 8) {
        double v;
        std::memcpy(&v, d.p, 8);
        out->push_back((float)v);
      }
    } else if (field == 8 && wire == 1) {
      if (b.end - b.p < 8) return false;
      double v;
      std::memcpy(&v, b.p, 8);
      b.p += 8;
      out->push_back((float)v);
    } else {
      b.skip(wire);
    }
  }
  return b.ok;
}

bool parse_layer(Pb l, int name_field, int blobs_field, std::string* name, std::vector<std::vector<float>>* blobs) {
  while (l.more()) {
    const uint64_t key = l.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == name_field && wire == 2) {
      Pb s = l.sub();
      name->assign((const char*)s.p, (size_t)(s.end - s.p));
    } else if (field == blobs_field && wire == 2) {
      blobs->emplace_back();
      if (!parse_blob(l.sub(), &blobs->back())) return false;
    } else {
      l.skip(wire);
    }
  }
  return l.ok;
}
}  // namespace

bool Classifier::readCaffeModel(const std::string& path, std::vector<float> blobs[8], std::string* err) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) { *err = "cannot open " + path; return false; }
  const std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  Pb net((const uint8_t*)raw.data(), (const uint8_t*)raw.data() + raw.size());
  static const char* kLayers[4] = {"conv1", "conv2", "ip1", "ip2"};
  bool seen[4] = {false, false, false, false};
  while (net.more()) {
    const uint64_t key = net.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if ((field == 100 || field == 2) && wire == 2) {
      std::string name;
      std::vector<std::vector<float>> lb;
      if (!parse_layer(net.sub(), field == 100 ? 1 : 4, field == 100 ? 7 : 6, &name, &lb)) {
        *err = "malformed layer record in " + path;
        return false;
      }
      for (int k = 0; k < 4; k++) {
        if (name != kLayers[k] || lb.empty()) continue;  // pooling / relu / softmax carry no blobs
        if (lb.size() != 2 || lb[0].size() != kBlobSizes[2 * k] || lb[1].size() != kBlobSizes[2 * k + 1]) {
          *err = "layer " + name + " does not have the blob sizes of caffe/test_1batch2.prototxt";
          return false;
        }
        blobs[2 * k] = lb[0];
        blobs[2 * k + 1] = lb[1];
        seen[k] = true;
      }
    } else {
      net.skip(wire);
    }
  }
  if (!net.ok) { *err = "not a protobuf NetParameter: " + path; return false; }
  for (int k = 0; k < 4; k++)
    if (!seen[k]) { *err = std::string("layer ") + kLayers[k] + " with weights not found in " + path; return false; }
  return true;
}

Classifier::Classifier(const std::string& model_file, const std::string& trained_file,
                       const std::string& label_file) {
  if (!model_file.empty()) {
    std::ifstream m(model_file.c_str());
    if (!m) { err_ = "cannot open model file " + model_file; return; }
  }
  std::ifstream f(trained_file.c_str(), std::ios::binary);
  char magic[4] = {0, 0, 0, 0};
  if (!f || !f.read(magic, 4)) {
    err_ = "cannot read weights file " + trained_file;
    return;
  }
  if (std::memcmp(magic, "AG2W", 4) != 0) {  // CopyTrainedLayersFrom, caffe_classifier.cpp:14
    f.close();
    if (!readCaffeModel(trained_file, blobs_, &err_)) return;
  } else {
    for (int b = 0; b < 8; b++) {
      blobs_[b].resize(kBlobSizes[b]);
      if (!f.read(reinterpret_cast<char*>(blobs_[b].data()), (std::streamsize)(kBlobSizes[b] * 4))) {
        err_ = "truncated weights file " + trained_file;
        return;
      }
    }
  }
  std::ifstream l(label_file.c_str());
  if (!l) { err_ = "Unable to open labels file " + label_file; return; }  // caffe_classifier.cpp:27
  std::string line;
  while (std::getline(l, line)) labels_.push_back(line);
  if (labels_.size() != 2) { err_ = "Number of labels is different from the output layer dimension."; return; }
  static std::atomic<uint64_t> next_generation{0};
  generation_ = ++next_generation;
  ok_ = true;
}

int Classifier::loadInto(ag2::Context& ctx) const {
  if (!ok_ || !ctx.ok()) return AG2_ERR_STATE;
  if (ctx.weights_generation == generation_) return 0;  // this context already holds these blobs, packed
  const int rc = ag2_lenet_load(ctx.get(), blobs_[0].data(), blobs_[1].data(), blobs_[2].data(), blobs_[3].data(),
                                blobs_[4].data(), blobs_[5].data(), blobs_[6].data(), blobs_[7].data());
  ctx.weights_generation = rc ? 0 : generation_;
  return rc;
}

void Classifier::setContext(std::shared_ptr<ag2::Context> ctx) {
  ctx_ = std::move(ctx);
  uploaded_ = false;
}

bool Classifier::ensureLoaded() {
  if (!ok_) return false;
  if (!ctx_) {
    ag2_params p;
    ag2_default_params(&p);
    ctx_.reset(new ag2::Context(p, 0));
  }
  if (!ctx_->ok()) { err_ = "could not create a GPU context (no CPU fallback)"; return false; }
  if (!uploaded_) {
    if (loadInto(*ctx_)) { err_ = ag2_last_error(ctx_->get()); return false; }
    uploaded_ = true;
  }
  return true;
}

std::vector<std::vector<Prediction>> Classifier::ClassifyBatch(const std::vector<ag2::Image>& imgs, int num_classes) {
  std::vector<std::vector<Prediction>> out;
  if (imgs.empty() || num_classes != 2) return out;
  if (!ensureLoaded()) {
    fprintf(stderr, "Classifier: %s\n", err_.c_str());
    return out;
  }
  std::vector<uint8_t> raw(imgs.size() * 10800);
  for (size_t i = 0; i < imgs.size(); i++) {
    if (imgs[i].rows != 60 || imgs[i].cols != 60 || imgs[i].chans != 3 || imgs[i].data.size() != 10800) {
      fprintf(stderr, "Classifier: image %zu is not 60x60x3 uint8\n", i);
      return out;
    }
    std::copy(imgs[i].data.begin(), imgs[i].data.end(), raw.begin() + i * 10800);
  }
  std::vector<float> logits(imgs.size() * 2);
  if (ag2_lenet_forward(ctx_->get(), raw.data(), imgs.size(), logits.data())) {
    fprintf(stderr, "Classifier: %s\n", ag2_last_error(ctx_->get()));
    return out;
  }
  out.resize(imgs.size());
  for (size_t i = 0; i < imgs.size(); i++)
    for (int k = 0; k < 2; k++) out[i].push_back(std::make_pair(labels_[k], logits[2 * i + k]));
  return out;
}

std::vector<Prediction> Classifier::Classify(const ag2::Image& img, bool) {
  std::vector<std::vector<Prediction>> r = ClassifyBatch(std::vector<ag2::Image>(1, img), 2);
  return r.empty() ? std::vector<Prediction>() : r[0];
}

// ------------------------------------------------------------------------------------------------
// GraspDetector
// ------------------------------------------------------------------------------------------------
namespace {
std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n\""), b = s.find_last_not_of(" \t\r\n\"");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::vector<double> parse_list(const std::string& v) {
  std::vector<double> out;
  std::string t = v;
  for (char& ch : t)
    if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
  std::istringstream is(t);
  double d;
  while (is >> d) out.push_back(d);
  return out;
}
bool parse_bool(const std::string& v) { return v == "true" || v == "1" || v == "True"; }

bool set_param(GraspDetector::Params* p, const std::string& k, const std::string& v, std::string* err) {
#define AG2_NUM(name) if (k == #name) { p->name = (decltype(p->name))std::atof(v.c_str()); return true; }
#define AG2_BOOL(name) if (k == #name) { p->name = parse_bool(v); return true; }
#define AG2_STR(name) if (k == #name) { p->name = v; return true; }
  if (k == "workspace") { p->workspace = parse_list(v); return true; }
  if (k == "camera_pose") { p->camera_pose = parse_list(v); return true; }
  if (k == "gripper_width_range") { p->gripper_width_range = parse_list(v); return true; }
  if (k == "tiling") {
    if (v == "spatial" || v == "1") p->tiling = GraspDetector::Params::TILING_SPATIAL;
    else if (v == "replicate" || v == "0") p->tiling = GraspDetector::Params::TILING_REPLICATE;
    else { *err = "tiling must be replicate or spatial"; return false; }
    return true;
  }
  if (k == "devices") {
    p->devices.clear();
    for (double d : parse_list(v)) p->devices.push_back((int)d);
    return true;
  }
  if (k == "sample_indices") {
    p->sample_indices.clear();
    for (double d : parse_list(v)) p->sample_indices.push_back((int)d);
    return true;
  }
  AG2_NUM(num_samples) AG2_NUM(num_threads) AG2_NUM(nn_radius_taubin) AG2_NUM(nn_radius_hands)
  AG2_NUM(num_orientations) AG2_BOOL(voxelize) AG2_BOOL(filter_half_grasps) AG2_NUM(finger_width)
  AG2_NUM(hand_outer_diameter) AG2_NUM(hand_depth) AG2_NUM(hand_height) AG2_NUM(init_bite)
  AG2_NUM(antipodal_mode) AG2_STR(model_file) AG2_STR(trained_file) AG2_STR(label_file)
  AG2_NUM(min_score_diff) AG2_NUM(batch_size) AG2_NUM(min_inliers) AG2_NUM(min_length)
  AG2_BOOL(reuse_inliers) AG2_NUM(num_selected) AG2_NUM(plot_mode) AG2_BOOL(only_plot_output)
  AG2_NUM(device) AG2_NUM(seed)
#undef AG2_NUM
#undef AG2_BOOL
#undef AG2_STR
  // node-level parameters of the reference that this library does not consume
  static const char* ignored[] = {"cloud_type", "cloud_file_name", "cloud_topic", "samples_topic",
                                  "use_importance_sampling", "use_service", "images_directory",
                                  "normal_estimation_method", "rviz_topic", "num_init_samples",
                                  "num_iterations", "num_samples_per_iteration", "prob_rand_samples",
                                  "std", "sampling_method", "visualize_rounds"};
  for (const char* ig : ignored)
    if (k == ig) return true;
  if (err) *err = "unknown parameter: " + k;
  return false;
}
}  // namespace

bool GraspDetector::Params::fromKeyValueText(const std::string& text, Params* out, std::string* err) {
  std::istringstream is(text);
  std::string line;
  while (std::getline(is, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    const size_t eq = line.find('=');
    if (eq == std::string::npos) {
      if (!trim(line).empty()) { if (err) *err = "expected name=value: " + line; return false; }
      continue;
    }
    if (!set_param(out, trim(line.substr(0, eq)), trim(line.substr(eq + 1)), err)) return false;
  }
  return true;
}

bool GraspDetector::Params::fromLaunchXml(const std::string& xml, Params* out, std::string* err) {
  // strip <!-- comments -->
  std::string s;
  for (size_t i = 0; i < xml.size();) {
    if (xml.compare(i, 4, "<!--") == 0) {
      const size_t e = xml.find("-->", i + 4);
      i = (e == std::string::npos) ? xml.size() : e + 3;
    } else {
      s.push_back(xml[i++]);
    }
  }
  auto attr = [](const std::string& tag, const std::string& name) {
    const size_t a = tag.find(name + "=\"");
    if (a == std::string::npos) return std::string();
    const size_t b = a + name.size() + 2, e = tag.find('"', b);
    return e == std::string::npos ? std::string() : tag.substr(b, e - b);
  };
  size_t pos = 0;
  while ((pos = s.find("<param ", pos)) != std::string::npos) {
    const size_t e = s.find('>', pos);
    if (e == std::string::npos) break;
    const std::string tag = s.substr(pos, e - pos);
    std::string v = attr(tag, "value");
    const size_t find_macro = v.find("$(find agile_grasp2)");
    if (find_macro != std::string::npos) v.replace(find_macro, 20, ".");
    if (!set_param(out, attr(tag, "name"), v, err)) return false;
    pos = e;
  }
  pos = 0;
  while ((pos = s.find("<rosparam ", pos)) != std::string::npos) {
    const size_t e = s.find('>', pos), c = s.find("</rosparam>", pos);
    if (e == std::string::npos || c == std::string::npos) break;
    if (!set_param(out, attr(s.substr(pos, e - pos), "param"), trim(s.substr(e + 1, c - e - 1)), err)) return false;
    pos = c;
  }
  return true;
}

GraspDetector::GraspDetector(const Params& params) : p_(params), num_samples_(params.num_samples) {
  indices_ = p_.sample_indices;
  if (!indices_.empty()) num_samples_ = (int)indices_.size();  // grasp_detector.cpp:24-29
  if (p_.antipodal_mode == PREDICTION)
    classifier_.reset(new Classifier(p_.model_file, p_.trained_file, p_.label_file));
  learning_.reset(new Learning(60, p_.num_threads));  // grasp_detector.cpp:56
  handle_search_.setMinInliers(p_.min_inliers);       // grasp_detector.cpp:58-67
  handle_search_.setMinLength(p_.min_length);
  handle_search_.setReuseInliers(p_.reuse_inliers);
}

GraspDetector::~GraspDetector() {}

void GraspDetector::setIndicesFromMsg(const agile_grasp2::CloudIndexed& msg) {  // grasp_detector.cpp:353-361
  indices_.resize(msg.indices.size());
  for (size_t i = 0; i < indices_.size(); i++) indices_[i] = (int)msg.indices[i].data;
}

void GraspDetector::cameraPoses(Matrix4d* left, Matrix4d* right) const {
  if (p_.camera_pose.size() == 16) {  // grasp_detector.cpp:129-136
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) (*left)(i, j) = p_.camera_pose[4 * i + j];
    *right = Matrix4d::Identity();
    return;
  }
  // camera poses of the 2-camera Baxter setup, grasp_detector.cpp:108-126
  Matrix4d base_tf, sqrt_tf;
  const double b[16] = {0, 0.445417, 0.895323, 0.215, 1, 0, 0, -0.015, 0, 0.895323, -0.445417, 0.23, 0, 0, 0, 1};
  const double q[16] = {0.9366, -0.0162, 0.3500, -0.2863, 0.0151, 0.9999, 0.0058, 0.0058,
                        -0.3501, -0.0002, 0.9367, 0.0554, 0, 0, 0, 1};
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      base_tf(i, j) = b[4 * i + j];
      sqrt_tf(i, j) = q[4 * i + j];
    }
  *left = base_tf * sqrt_tf.inverse();
  *right = base_tf * sqrt_tf;
}

ag2_params GraspDetector::abiParams(int n_cams) const {
  HandSearch::Parameters hp;
  hp.nn_radius_taubin_ = p_.nn_radius_taubin;
  hp.nn_radius_hands_ = p_.nn_radius_hands;
  hp.num_threads_ = p_.num_threads;
  hp.num_samples_ = num_samples_;
  hp.num_orientations_ = p_.num_orientations;
  hp.finger_width_ = p_.finger_width;
  hp.hand_outer_diameter_ = p_.hand_outer_diameter;
  hp.hand_depth_ = p_.hand_depth;
  hp.hand_height_ = p_.hand_height;
  hp.init_bite_ = p_.init_bite;
  cameraPoses(&hp.cam_tf_left_, &hp.cam_tf_right_);
  ag2_params ap = HandSearch::toAbiParams(hp, n_cams);
  ap.filter_half_grasps = p_.filter_half_grasps ? 1 : 0;
  for (int i = 0; i < 6 && i < (int)p_.workspace.size(); i++) ap.workspace[i] = p_.workspace[i];
  if (p_.gripper_width_range.size() >= 2) {
    ap.min_aperture = p_.gripper_width_range[0];
    ap.max_aperture = p_.gripper_width_range[1];
  }
  ap.min_score_diff = p_.min_score_diff;
  ap.num_selected = p_.num_selected;
  return ap;
}

std::shared_ptr<ag2::Context> GraspDetector::contextFor(int n_cams) {
  if (ctx_ && ctx_cams_ == n_cams) return ctx_;
  const ag2_params ap = abiParams(n_cams);
  ctx_.reset(new ag2::Context(ap, p_.devices.empty() ? p_.device : p_.devices[0]));
  peers_.clear();
  ctx_cams_ = n_cams;
  resident_ctx_ = nullptr;
  if (!ctx_->ok()) {
    err_ = "could not create a GPU context (no CPU fallback)";
    ctx_.reset();
    return ctx_;
  }
  if (classifier_) classifier_->setContext(ctx_);
  learning_->setContext(ctx_);
  return ctx_;
}

std::vector<GraspHypothesis> GraspDetector::detectGraspPoses(const CloudCamera& cloud_cam, bool clusters_grasps) {
  return detectImpl(cloud_cam, clusters_grasps, nullptr, false);
}

std::vector<GraspHypothesis> GraspDetector::detectImpl(const CloudCamera& cloud_cam, bool clusters_grasps,
                                                       const Matrix3Xd* samples_xyz, bool cloud_is_resident) {
  std::vector<GraspHypothesis> out;
  const int hs_inliers = handle_search_.getMinInliers();
  const int min_inliers = (clusters_grasps && hs_inliers > 0) ? hs_inliers : 0;  // :228-236
  if (cloud_cam.getCloudOriginal()->size() == 0) {  // grasp_detector.cpp:86-91
    fprintf(stderr, "Point cloud is empty!\n");
    return out;
  }
  const int n_cams = std::max(1, cloud_cam.getCameraSource().rows());
  std::shared_ptr<ag2::Context> ctx = contextFor(n_cams);
  if (!ctx) {
    fprintf(stderr, "GraspDetector: %s\n", err_.c_str());
    return out;
  }
  ag2_ctx* c = ctx->get();
  int rc = 0;
  // The N-device forms (Params::devices, PREDICTION, index samples): with spatial tiles no device -- the first
  // included -- holds the whole cloud, so the upload below is left out.
  const bool index_samples = !samples_xyz && !use_incoming_samples_;
  const bool spatial = p_.tiling == Params::TILING_SPATIAL && p_.antipodal_mode == PREDICTION && index_samples;
  const bool multi = p_.devices.size() > 1 && p_.antipodal_mode == PREDICTION && index_samples &&
                     !cloud_cam.getSampleIndices().empty();
  const bool same_cloud = c == resident_ctx_ && cloud_cam.getCloudProcessed().get() == resident_cloud_ &&
                          cloud_cam.getCloudProcessed()->size() == resident_n_;
  if (multi && spatial) {
    resident_ctx_ = nullptr;  // (the first device will hold its tile, not this cloud)
  } else if (same_cloud && cloud_is_resident && resident_normals_) {
    // ImportanceSampling re-entry: grid and normals of this cloud are already in the context
  } else if (same_cloud && !cloud_is_resident && !resident_uploaded_here_) {
    // preprocessPointCloud left exactly this cloud (and its grid) in the context
    if (!resident_normals_) rc = ag2_compute_normals(c);  // hand_search.cpp:20-29
    resident_normals_ = true;
  } else {
    rc = HandSearch::uploadCloud(c, cloud_cam);
    resident_ctx_ = rc ? nullptr : c;
    resident_cloud_ = cloud_cam.getCloudProcessed().get();
    resident_n_ = cloud_cam.getCloudProcessed()->size();
    resident_normals_ = true;
    resident_uploaded_here_ = true;  // a public detectGraspPoses call always uploads again
  }
  const bool use_samples = samples_xyz ? true : use_incoming_samples_;
  const Matrix3Xd& sample_mat = samples_xyz ? *samples_xyz : cloud_cam.getSamples();
  const size_t s = use_samples ? (size_t)sample_mat.cols() : cloud_cam.getSampleIndices().size();
  std::vector<int32_t> idx(cloud_cam.getSampleIndices().begin(), cloud_cam.getSampleIndices().end());
  if (spatial && !use_samples) {  // the order tiles shard: the same for every number of devices, one included
    int axis = 0;
    idx = orderSamplesAlongLongestAxis(cloud_cam, idx, &axis);
  }
  const int32_t* pidx = use_samples ? nullptr : idx.data();
  const double* pxyz = use_samples ? sample_mat.data() : nullptr;
  const double dummy_xyz[3] = {0.0, 0.0, 0.0};
  if (use_samples && s == 0) pxyz = dummy_xyz;
  const int32_t dummy = 0;
  if (!use_samples && idx.empty()) pidx = &dummy;  // zero samples: a valid (empty) request
  const bool do_prune = indices_.empty();          // grasp_detector.cpp:149-160
  const size_t cap = std::max<size_t>(1, s * (size_t)p_.num_orientations);
  std::vector<ag2_hypothesis> recs(cap);
  size_t n = 0;
  if (!rc) {
    if (p_.antipodal_mode == PREDICTION) {
      if (!classifier_ || !classifier_->ok()) {
        fprintf(stderr, "GraspDetector: classifier not available: %s\n",
                classifier_ ? classifier_->error().c_str() : "not constructed");
        return out;
      }
      rc = classifier_->loadInto(*ctx);  // (packed once per context and classifier, not once per call)
      if (!rc) rc = ag2_set_min_inliers(c, min_inliers);  // clustering between threshold and top-k, on the GPU
      if (!rc && multi && s > 0) {
        if (!detectOnDevices(cloud_cam, idx, do_prune, min_inliers, &recs, &n)) {
          fprintf(stderr, "GraspDetector::detectGraspPoses: %s\n", err_.c_str());
          return out;
        }
      } else if (!rc) {
        rc = ag2_detect(c, pidx, pxyz, s, 0, p_.seed, do_prune ? 1 : 0, recs.data(), cap, &n, nullptr, 0, nullptr);
      }
    } else {
      rc = ag2_generate_hypotheses(c, pidx, pxyz, s, 0, p_.seed, recs.data(), cap, &n);
      if (!rc) {
        std::vector<uint8_t> keep(n, 1);
        if (do_prune && n) rc = ag2_prune(c, keep.data(), n);
        size_t m = 0;
        for (size_t h = 0; h < n; h++) {
          const bool sel = keep[h] && (p_.antipodal_mode == NONE || recs[h].full_antipodal);  // :163-221
          if (sel) recs[m++] = recs[h];
        }
        n = m;
        // antipodal_mode NONE returns the pruned hypotheses as they are -- no clustering, no top-k
        // (grasp_detector.cpp:170-176 returns hands_filtered before steps 4 and 5)
        if (p_.antipodal_mode != NONE && min_inliers > 0 && n) {  // 4. grasp clusters, :228-236
          std::vector<ag2_hypothesis> clustered(n);
          size_t k = 0;
          rc = ag2_find_clusters(c, recs.data(), n, min_inliers, clustered.data(), n, &k);
          if (!rc) std::copy(clustered.begin(), clustered.begin() + (long)k, recs.begin());
          if (!rc) n = k;
        }
        if (p_.antipodal_mode == GEOMETRIC) {  // :239-252 (scores are all zero: stable order kept)
          if ((int)n > p_.num_selected) n = (size_t)p_.num_selected;
        }
      }
    }
  }
  if (rc) {
    err_ = ag2_last_error(c);
    fprintf(stderr, "GraspDetector::detectGraspPoses: %s\n", err_.c_str());
    return out;
  }
  (void)ag2_get_stage_times(c, &times_);
  (void)ag2_get_counters(c, &counters_);
  out.reserve(n);
  for (size_t h = 0; h < n; h++) out.push_back(GraspHypothesis(recs[h]));
  return out;
}

std::vector<GraspHypothesis> GraspDetector::detectGraspPosesInFrame(const PointCloudRGB::Ptr& raw_cloud) {
  std::vector<GraspHypothesis> out;
  if (!raw_cloud || raw_cloud->size() == 0) {  // grasp_detector.cpp:86-91
    fprintf(stderr, "Point cloud is empty!\n");
    return out;
  }
  const bool one_call = p_.antipodal_mode == PREDICTION && p_.voxelize && !use_incoming_samples_ && indices_.empty() &&
                        p_.devices.size() <= 1 && num_samples_ > 0 && classifier_ && classifier_->ok();
  if (!one_call) {  // the two calls, as the file path of the node makes them (grasp_detection_node.cpp:97-121)
    CloudCamera cc(raw_cloud, (int)raw_cloud->size());
    preprocessPointCloud(cc);
    return detectGraspPoses(cc);
  }
  std::shared_ptr<ag2::Context> ctx = contextFor(1);
  if (!ctx) {
    fprintf(stderr, "GraspDetector: %s\n", err_.c_str());
    return out;
  }
  ag2_ctx* c = ctx->get();
  int rc = 0;
  rc = classifier_->loadInto(*ctx);  // once per (context, classifier): the context remembers (a context that
                                     // contextFor() re-creates starts without weights, whatever its address)
  const int hs_inliers = handle_search_.getMinInliers();
  if (!rc) rc = ag2_set_min_inliers(c, hs_inliers > 0 ? hs_inliers : 0);  // (the clustering is part of the captured sequence)
  const size_t cap = std::max<size_t>(1, (size_t)num_samples_ * (size_t)p_.num_orientations);
  std::vector<ag2_hypothesis> recs(cap);
  size_t n = 0, n_scored = 0, n_vox = 0;
  if (!rc)
    rc = ag2_detect_frame_raw(c, &raw_cloud->points[0].x, /*on_device=*/0, raw_cloud->size(), sizeof(ag2::PointXYZRGBA),
                              p_.workspace.size() >= 6 ? 1 : 0, voxel_size_, (size_t)num_samples_, p_.seed, p_.seed,
                              /*do_prune=*/1, recs.data(), recs.size(), &n, &n_scored, &n_vox);
  resident_ctx_ = nullptr;  // (the context's cloud is this frame's processed cloud, not a CloudCamera's)
  if (rc) {
    err_ = ag2_last_error(c);
    fprintf(stderr, "GraspDetector::detectGraspPosesInFrame: %s\n", err_.c_str());
    return out;
  }
  (void)ag2_get_stage_times(c, &times_);
  (void)ag2_get_counters(c, &counters_);
  out.reserve(n);
  for (size_t h = 0; h < n; h++) out.push_back(GraspHypothesis(recs[h]));
  return out;
}

// ---- Params::devices: N GPUs from the one process of the node ---------------------------------------------------
// Replicated cloud (TILING_REPLICATE): every device holds the whole cloud (a 300 k-point cloud is 4.8 MB) and takes
// a contiguous range of the sample list.  Spatial tiles (TILING_SPATIAL, BASELINE configuration 4): the list --
// already in ascending order along the cloud's longest axis -- is cut into ranges of equal summed neighbour
// counts and every device holds only the points of its samples' interval +- the halo, binned against the whole
// cloud's minimum: grid + normals shard as well.  Either way a device runs grid, normals and detect for its range
// with the samples' positions in the WHOLE list as slot base, so its hypotheses are those of the one-device run
// (hand_search.cpp:194-228: no state crosses samples; the draws are keyed by the slot).  The ranks leave their
// scored candidates above the threshold on the first device (ag2_gather_selected: peer copies over xGMI) and that
// device clusters (when min_inliers > 0) and takes the top num_selected over the concatenation, which is in
// sample order (grasp_detector.cpp:228-252).  LeNet weights are packed into a device's context ONCE
// (Classifier::loadInto), not once per call.
namespace {

// float per-axis minimum over the finite points: the unsplit run's grid origin (sharding.cloud_origin)
bool cloud_extent(const PointCloudRGB& cloud, float mn[3], float mx[3]) {
  bool any = false;
  for (const ag2::PointXYZRGBA& p : cloud.points) {
    if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) continue;
    const float v[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; a++) {
      mn[a] = any ? std::min(mn[a], v[a]) : v[a];
      mx[a] = any ? std::max(mx[a], v[a]) : v[a];
    }
    any = true;
  }
  return any;
}
inline float coord(const ag2::PointXYZRGBA& p, int axis) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }

// sharding.sample_costs + balanced_bounds: G + 1 boundaries of contiguous ranges of the ordered sample list with
// (nearly) equal summed cost, the cost of a sample being the number of cloud points whose coordinate along the
// axis lies within nn_radius_hands of its own -- the 1-D marginal of K2, which drives the hand sweep (SURVEY.md
// section 8e: balance on sum K2, not on the sample count).  Every rank gets a sample while there are enough.
std::vector<size_t> balanced_bounds(const PointCloudRGB& cloud, const std::vector<int32_t>& ordered, int axis,
                                    double radius, size_t G) {
  const size_t n = ordered.size();
  std::vector<size_t> b(G + 1, 0);
  if (n == 0) return b;
  if (n <= G) {
    for (size_t g = 0; g <= G; g++) b[g] = std::min(g, n);
    return b;
  }
  std::vector<double> xs;
  xs.reserve(cloud.points.size());
  for (const ag2::PointXYZRGBA& p : cloud.points) {
    const double x = (double)coord(p, axis);
    if (std::isfinite(x)) xs.push_back(x);
  }
  std::sort(xs.begin(), xs.end());
  std::vector<double> cum(n);
  double run = 0.0;
  for (size_t i = 0; i < n; i++) {
    const double sx = (double)coord(cloud.points[(size_t)ordered[i]], axis);
    const double cost = (double)(std::lower_bound(xs.begin(), xs.end(), sx + radius) -
                                 std::lower_bound(xs.begin(), xs.end(), sx - radius));
    run += std::max(cost, 1.0);
    cum[i] = run;
  }
  b[G] = n;
  for (size_t g = 1; g < G; g++) {
    const double target = cum[n - 1] * (double)g / (double)G;
    b[g] = (size_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin()) + 1;
  }
  for (size_t g = 1; g < G; g++) b[g] = std::min(std::max(b[g], b[g - 1] + 1), n - (G - g));
  return b;
}

}  // namespace

std::vector<int32_t> GraspDetector::orderSamplesAlongLongestAxis(const CloudCamera& cloud_cam,
                                                                 const std::vector<int32_t>& idx, int* axis) {
  const PointCloudRGB& cloud = *cloud_cam.getCloudProcessed();
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  *axis = 0;
  if (cloud_extent(cloud, mn, mx))
    for (int a = 1; a < 3; a++)
      if (mx[a] - mn[a] > mx[*axis] - mn[*axis]) *axis = a;
  std::vector<int32_t> ordered(idx);
  const int ax = *axis;
  std::stable_sort(ordered.begin(), ordered.end(), [&](int32_t a, int32_t b) {
    return coord(cloud.points[(size_t)a], ax) < coord(cloud.points[(size_t)b], ax);
  });
  return ordered;
}

// Host-only view of the tile plan (no GPU is touched): what TILING_SPATIAL does with a cloud and a sample list --
// the ordered list, the ranges of equal summed neighbour counts, and per rank the points its tile holds.  Used by the
// CPU tests to hold this port against agile_grasp2_amd/sharding.py; a site can use it to size its devices.
extern "C" int ag2host_tile_plan(const float* xyz, size_t n, const int32_t* idx, size_t s, int world, double radius_hands,
                                 double halo, int32_t* ordered_out, int64_t* bounds_out, int64_t* tile_points_out,
                                 int32_t* axis_out) {
  if (!xyz || !idx || world < 1 || !ordered_out || !bounds_out) return -1;
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(n);
  for (size_t i = 0; i < n; i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  CloudCamera cc(cloud, (int)n);
  int axis = 0;
  const std::vector<int32_t> ordered = GraspDetector::orderSamplesAlongLongestAxis(cc, std::vector<int32_t>(idx, idx + s), &axis);
  std::copy(ordered.begin(), ordered.end(), ordered_out);
  const std::vector<size_t> b = balanced_bounds(*cloud, ordered, axis, radius_hands, (size_t)world);
  for (int g = 0; g <= world; g++) bounds_out[g] = (int64_t)b[(size_t)g];
  if (axis_out) *axis_out = axis;
  if (tile_points_out)
    for (int g = 0; g < world; g++) {
      int64_t m = 0;
      if (b[(size_t)g + 1] > b[(size_t)g]) {
        double xlo = 0.0, xhi = 0.0;
        for (size_t i = b[(size_t)g]; i < b[(size_t)g + 1]; i++) {
          const double x = (double)coord(cloud->points[(size_t)ordered[i]], axis);
          xlo = (i == b[(size_t)g]) ? x : std::min(xlo, x);
          xhi = (i == b[(size_t)g]) ? x : std::max(xhi, x);
        }
        for (size_t i = 0; i < n; i++) {
          const double x = (double)coord(cloud->points[i], axis);
          if (x >= xlo - halo && x <= xhi + halo) m++;
        }
      }
      tile_points_out[g] = m;
    }
  return 0;
}

bool GraspDetector::detectOnDevices(const CloudCamera& cloud_cam, const std::vector<int32_t>& idx, bool do_prune,
                                    int min_inliers, std::vector<ag2_hypothesis>* recs, size_t* n) {
  const size_t s = idx.size();
  const bool spatial = p_.tiling == Params::TILING_SPATIAL;
  const size_t G = std::min(p_.devices.size(), std::max<size_t>(s, 1));  // (fewer samples than devices: the last ones idle)
  const int n_cams = std::max(1, cloud_cam.getCameraSource().rows());
  if (peers_.size() != p_.devices.size() - 1 || peers_cams_ != n_cams) {
    peers_.clear();
    const ag2_params ap = abiParams(n_cams);
    for (size_t g = 1; g < p_.devices.size(); g++) {
      std::shared_ptr<ag2::Context> pc(new ag2::Context(ap, p_.devices[g]));
      if (!pc->ok()) {
        err_ = "could not create a GPU context on device " + std::to_string(p_.devices[g]) + " (no CPU fallback)";
        peers_.clear();
        return false;
      }
      peers_.push_back(pc);
    }
    peers_cams_ = n_cams;
  }
  std::vector<ag2::Context*> cx(G, ctx_.get());
  for (size_t g = 1; g < G; g++) cx[g] = peers_[g - 1].get();
  ag2_ctx* root = cx[0]->get();
  const PointCloudRGB& cloud = *cloud_cam.getCloudProcessed();
  const size_t n_pts = cloud.points.size();
  const MatrixXi& src = cloud_cam.getCameraSource();
  const Matrix3Xd& nrm = cloud_cam.getNormals();
  const bool has_src = src.cols() == (int)n_pts && n_pts > 0;
  const bool has_n = (size_t)nrm.cols() == n_pts && n_pts > 0;
  // the ranges of the sample list, and (spatial) the interval of the axis every rank has to hold
  std::vector<size_t> lo(G + 1);
  int axis = 0;
  float origin[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  if (spatial) {
    if (cloud_extent(cloud, origin, mx))
      for (int a = 1; a < 3; a++)
        if (mx[a] - origin[a] > mx[axis] - origin[axis]) axis = a;
    lo = balanced_bounds(cloud, idx, axis, p_.nn_radius_hands, G);
  } else {
    for (size_t g = 0; g <= G; g++) lo[g] = s * g / G;
  }
  size_t longest = 0;
  for (size_t g = 0; g < G; g++) longest = std::max(longest, lo[g + 1] - lo[g]);
  const size_t cap_records = std::max<size_t>(1, longest * (size_t)p_.num_orientations);  // every slot: never cut
  // (A device's detect runs in one trip from its second call on, at the shapes its previous call left; when they
  // did not hold -- the merge reads that from the gathered headers: AG2_ERR_RETRY -- the whole step is repeated
  // once more, step by step.)
  for (int attempt = 0; attempt < 3; attempt++) {
  if (ag2_gather_begin(root, G, cap_records)) {
    err_ = ag2_last_error(root);
    return false;
  }
  // sharding.tile_halo: what a sample reads of the cloud (its hand neighbourhood, those points' normals), plus
  // the rounding of the f32 coordinate differences the radius tests see
  const double halo = std::max(p_.nn_radius_hands, p_.nn_radius_taubin) + cx[0]->params().normals_radius + 1e-5;
  std::vector<int> rcs(G, 0);
  std::vector<std::string> msgs(G);
  tile_points_.assign(G, 0);
  auto run = [&](size_t g) {
    ag2::Context& ctx = *cx[g];
    ag2_ctx* c = ctx.get();
    int rc = 0;
    const size_t len = lo[g + 1] - lo[g];
    std::vector<int32_t> local(idx.begin() + (long)lo[g], idx.begin() + (long)lo[g + 1]);
    if (spatial) {
      // sharding.tile_points: the points of [min x - halo, max x + halo] of this rank's samples, in their original
      // relative order (an order-preserving subset keeps the canonical (cell, index) neighbour order)
      double xlo = 0.0, xhi = 0.0;
      for (size_t i = 0; i < len; i++) {
        const double x = (double)coord(cloud.points[(size_t)local[i]], axis);
        xlo = i ? std::min(xlo, x) : x;
        xhi = i ? std::max(xhi, x) : x;
      }
      xlo -= halo;
      xhi += halo;
      std::vector<int32_t> new_index(n_pts, -1);
      std::vector<float> xyz;
      std::vector<int32_t> cam;
      std::vector<double> nn;
      size_t m = 0;
      for (size_t i = 0; i < n_pts; i++) {
        const double x = (double)coord(cloud.points[i], axis);
        if (!(x >= xlo && x <= xhi)) continue;  // (NaN compares false: dropped, never a neighbour)
        new_index[i] = (int32_t)m++;
        xyz.push_back(cloud.points[i].x);
        xyz.push_back(cloud.points[i].y);
        xyz.push_back(cloud.points[i].z);
        if (has_src)
          for (int k = 0; k < n_cams; k++) cam.push_back(src(k, (int)i));
        if (has_n)
          for (int k = 0; k < 3; k++) nn.push_back(nrm(k, (int)i));
      }
      for (size_t i = 0; i < len; i++) local[i] = new_index[(size_t)local[i]];
      tile_points_[g] = m;
      rc = ag2_set_grid_origin(c, origin);
      if (!rc)
        rc = ag2_set_cloud(c, xyz.data(), m, 12, has_src ? cam.data() : nullptr, n_cams, has_n ? nn.data() : nullptr);
      if (!rc && !has_n) rc = ag2_compute_normals(c);
    } else if (g > 0) {  // (replicated cloud: the first device was set up by the caller)
      tile_points_[g] = n_pts;
      rc = HandSearch::uploadCloud(c, cloud_cam);
    } else {
      tile_points_[g] = n_pts;
    }
    if (!rc) rc = classifier_->loadInto(ctx);
    if (!rc) rc = ag2_set_min_inliers(c, min_inliers);
    size_t ns = 0, nsc = 0;
    const int32_t dummy = 0;
    if (!rc)  // selected == NULL, cap == 0: no local read-back, the merge selects
      rc = ag2_detect(c, len ? local.data() : &dummy, nullptr, len, (uint64_t)lo[g], p_.seed, do_prune ? 1 : 0,
                      nullptr, 0, &ns, nullptr, 0, &nsc);
    if (!rc) rc = ag2_gather_selected(root, c, g);
    if (spatial) (void)ag2_set_grid_origin(c, nullptr);  // (the context goes back to binning against its own cloud)
    rcs[g] = rc;
    if (rc) msgs[g] = ag2_last_error(c);
  };
  std::vector<std::thread> th;
  for (size_t g = 1; g < G; g++) th.emplace_back(run, g);
  run(0);
  for (std::thread& t : th) t.join();
  if (spatial) resident_ctx_ = nullptr;  // (the first device holds a tile now)
  for (size_t g = 0; g < G; g++)
    if (rcs[g]) {
      err_ = "device " + std::to_string(p_.devices[g]) + " (rank " + std::to_string(g) + "): " + msgs[g];
      return false;
    }
  const size_t k_cap = (p_.num_selected >= 0) ? std::min<size_t>((size_t)p_.num_selected, G * cap_records) : G * cap_records;
  recs->resize(std::max<size_t>(1, k_cap));
  size_t n_total = 0;
  const int mrc = ag2_merge_gathered(root, recs->data(), recs->size(), n, &n_total);
  if (mrc == AG2_ERR_RETRY) continue;
  if (mrc) {
    err_ = ag2_last_error(root);
    return false;
  }
  return true;
  }
  err_ = "the devices' shapes did not settle in three attempts";
  return false;
}

// Steps 1-2 (+ the uniform draw of step 3) on the GPU through ag2_preprocess_cloud; the processed
// cloud is read back into cloud_cam (getCloudProcessed / getCameraSource keep their meaning) and
// stays resident in the context for the detectGraspPoses call that follows.
bool GraspDetector::preprocessOnDevice(CloudCamera& cloud_cam) {
  const int n_cams = std::max(1, cloud_cam.getCameraSource().rows());
  std::shared_ptr<ag2::Context> ctx = contextFor(n_cams);
  if (!ctx) {
    fprintf(stderr, "GraspDetector::preprocessPointCloud: %s\n", err_.c_str());
    return false;
  }
  ag2_ctx* c = ctx->get();
  const PointCloudRGB::Ptr& raw = cloud_cam.getCloudProcessed();
  const size_t n = raw->size();
  const MatrixXi& src = cloud_cam.getCameraSource();
  const Matrix3Xd& nrm = cloud_cam.getNormals();
  const bool carry_normals = !p_.voxelize && n > 0 && (size_t)nrm.cols() == n;
  size_t m = 0;
  int rc = ag2_preprocess_cloud(c, n ? &raw->points[0].x : nullptr, n, sizeof(ag2::PointXYZRGBA),
                                src.cols() == (int)n && n ? src.data() : nullptr, n_cams,
                                carry_normals ? nrm.data() : nullptr, p_.workspace.size() >= 6 ? 1 : 0,
                                p_.voxelize ? 1 : 0, voxel_size_, 0, &m);
  std::vector<float> xyz(3 * std::max<size_t>(m, 1));
  MatrixXi cam(n_cams, (int)m);
  if (!rc) rc = ag2_get_cloud(c, xyz.data(), m ? cam.data() : nullptr, m, &m);
  Matrix3Xd out_n;
  if (!rc && carry_normals && m) {
    out_n.resize(3, (int)m);
    rc = ag2_get_normals(c, out_n.data());
  }
  if (rc) {
    err_ = ag2_last_error(c);
    fprintf(stderr, "GraspDetector::preprocessPointCloud: %s\n", err_.c_str());
    return false;
  }
  PointCloudRGB::Ptr cloud(new PointCloudRGB);
  cloud->points.resize(m);
  for (size_t i = 0; i < m; i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
  }
  cloud_cam.adoptProcessed(cloud, cam, out_n);
  resident_cloud_ = cloud.get();
  resident_n_ = m;
  resident_ctx_ = c;
  resident_normals_ = carry_normals && m;
  resident_uploaded_here_ = false;
  return true;
}

void GraspDetector::preprocessPointCloud(CloudCamera& cloud_cam) {
  if (indices_.empty()) {
    if (!preprocessOnDevice(cloud_cam)) return;  // 1.-2. :292-303; no host fallback: the error is printed
    const int n = (int)cloud_cam.getCloudProcessed()->size();
    if (use_incoming_samples_) {                                             // 3. :306-321
      agile_grasp2::SamplesMsg filtered;
      for (const agile_grasp2::Point& p : samples_msg_.samples)
        if (p_.workspace.size() < 6 ||
            (p.x > p_.workspace[0] && p.x < p_.workspace[1] && p.y > p_.workspace[2] && p.y < p_.workspace[3] &&
             p.z > p_.workspace[4] && p.z < p_.workspace[5]))
          filtered.samples.push_back(p);
      cloud_cam.subsampleSamples(filtered, num_samples_, p_.seed);
    } else if (num_samples_ > n) {                                           // :322-330
      std::vector<int> all(n);
      for (int i = 0; i < n; i++) all[i] = i;
      cloud_cam.setSampleIndices(all);
    } else {
      std::vector<int32_t> idx((size_t)std::max(num_samples_, 1));           // :331-335, drawn on the GPU
      size_t k = 0;
      ag2_ctx* c = const_cast<ag2_ctx*>(resident_ctx_);
      if (ag2_subsample_uniformly(c, (size_t)std::max(num_samples_, 0), p_.seed, idx.data(), idx.size(), &k)) {
        err_ = ag2_last_error(c);
        fprintf(stderr, "GraspDetector::preprocessPointCloud: %s\n", err_.c_str());
        return;
      }
      cloud_cam.setSampleIndices(std::vector<int>(idx.begin(), idx.begin() + (long)k));
    }
  } else {
    if (num_samples_ != (int)indices_.size() && num_samples_ < (int)cloud_cam.getCloudOriginal()->size()) {
      std::vector<int> r(num_samples_);                                      // :339-345 (rand() -> seeded)
      uint64_t s = p_.seed ^ 0x0F0F0F0F12345678ull;
      for (int i = 0; i < num_samples_; i++) r[i] = indices_[(size_t)(splitmix(s) % indices_.size())];
      cloud_cam.setSampleIndices(r);
    } else {
      cloud_cam.setSampleIndices(indices_);
    }
  }
}

std::vector<GraspHypothesis> GraspDetector::pruneGraspsOnHandParameters(const std::vector<GraspHypothesis>& hands,
                                                                        float min_x, float max_x, float min_y,
                                                                        float max_y, float min_z) {
  std::vector<GraspHypothesis> out;  // grasp_detector.cpp:363-395 (host twin of the in-kernel predicate)
  const double hw = 0.5 * p_.hand_outer_diameter;
  const double min_ap = p_.gripper_width_range.size() >= 2 ? p_.gripper_width_range[0] : 0.03;
  const double max_ap = p_.gripper_width_range.size() >= 2 ? p_.gripper_width_range[1] : 0.07;
  for (const GraspHypothesis& h : hands) {
    if (p_.filter_half_grasps && !h.isHalfAntipodal()) continue;
    double mn[3], mx[3];
    for (int a = 0; a < 3; a++) {
      const double c5[5] = {h.getGraspBottom()(a) + hw * h.getBinormal()(a), h.getGraspBottom()(a) - hw * h.getBinormal()(a),
                            h.getGraspTop()(a) + hw * h.getBinormal()(a), h.getGraspTop()(a) - hw * h.getBinormal()(a),
                            h.getGraspBottom()(a) - 0.10 * h.getApproach()(a)};
      mn[a] = mx[a] = c5[0];
      for (int k = 1; k < 5; k++) {
        mn[a] = std::min(mn[a], c5[k]);
        mx[a] = std::max(mx[a], c5[k]);
      }
    }
    const double ap = h.getGraspWidth();
    if (ap >= min_ap && ap <= max_ap && mn[2] >= (double)min_z && mn[1] >= (double)min_y && mx[1] <= (double)max_y &&
        mn[0] >= (double)min_x && mx[0] <= (double)max_x)
      out.push_back(h);
  }
  return out;
}

agile_grasp2::GraspListMsg GraspDetector::createGraspListMsg(const std::vector<GraspHypothesis>& hands) {
  agile_grasp2::GraspListMsg msg;
  for (const GraspHypothesis& h : hands) msg.grasps.push_back(h.convertToGraspMsg());
  return msg;
}

bool GraspDetector::findGrasps(const CloudCamera& cloud_in, const agile_grasp2::FindGraspsRequest& req,
                               agile_grasp2::FindGraspsResponse* resp) {
  CloudCamera cc = cloud_in;
  const std::vector<int> saved = indices_;
  const int saved_n = num_samples_;
  if (req.num_samples > 0) num_samples_ = req.num_samples;
  if (req.grasps_signal == 2) {        // samples given by indices, grasp_detection_node.cpp:178-186
    indices_.assign(req.indices.begin(), req.indices.end());
  } else if (req.grasps_signal == 1) {  // samples drawn from an r-ball, :158-176 / :204-213
    indices_.clear();
    const float r2 = req.radius * req.radius;
    const PointCloudRGB::Ptr& cl = cc.getCloudProcessed();
    for (size_t i = 0; i < cl->size(); i++) {
      const float dx = cl->points[i].x - (float)req.centroid.x, dy = cl->points[i].y - (float)req.centroid.y,
                  dz = cl->points[i].z - (float)req.centroid.z;
      if ((dx * dx + dy * dy) + dz * dz < r2) indices_.push_back((int)i);
    }
  } else {
    indices_.clear();
  }
  preprocessPointCloud(cc);
  const int saved_mode = p_.antipodal_mode;
  if (req.calculate_antipodal) p_.antipodal_mode = GEOMETRIC;
  const std::vector<GraspHypothesis> hands = detectGraspPoses(cc);
  p_.antipodal_mode = saved_mode;
  indices_ = saved;
  num_samples_ = saved_n;
  if (resp) resp->grasps_msg = createGraspListMsg(hands);  // filled (the reference leaves it empty, :196)
  return true;
}

// ------------------------------------------------------------------------------------------------
// ImportanceSampling
// ------------------------------------------------------------------------------------------------
ImportanceSampling::ImportanceSampling(const Params& params)
    : GraspDetector(params), num_iterations_(NUM_ITERATIONS), num_samples_is_(NUM_SAMPLES),
      num_init_samples_(NUM_INIT_SAMPLES), prob_rand_samples_(PROB_RAND_SAMPLES), radius_(RADIUS),
      sampling_method_(METHOD) {}

// standard normal deviate (Box-Muller) from the counter-based generator, stream = round
double ImportanceSampling::gaussian(uint64_t round, uint64_t* counter) const {
  const uint64_t stream = 0xFFFFFFFFFFFF0000ull + round;
  const double u1 = (double)((draw_u64(params().seed, stream, (*counter)++) >> 11) + 1ull) * (1.0 / 9007199254740992.0);
  const double u2 = (double)(draw_u64(params().seed, stream, (*counter)++) >> 11) * (1.0 / 9007199254740992.0);
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
}

std::vector<GraspHypothesis> ImportanceSampling::detectGraspPoses(const CloudCamera& cloud_cam_in) {
  const CloudCamera& cloud_cam = cloud_cam_in;
  rounds_.clear();
  // 1. initial grasp hypotheses (importance_sampling.cpp:38)
  std::vector<GraspHypothesis> hands = detectImpl(cloud_cam, false, nullptr, false);
  n_initial_ = (int)hands.size();
  if (hands.empty()) return hands;  // :40-43
  const PointCloudRGB::Ptr& cloud = cloud_cam.getCloudProcessed();
  // 2. (:50-62)
  const int num_rand_samples = (int)(prob_rand_samples_ * num_samples_is_);
  const int num_gauss_samples = num_samples_is_ - num_rand_samples;
  const double sigma = radius_;
  const double term = 1.0 / std::sqrt(std::pow(2.0 * M_PI, 3.0) * std::pow(sigma, 3.0));
  for (int it = 0; it < num_iterations_; it++) {  // 3. (:65-101)
    Matrix3Xd samples(3, num_samples_is_);
    uint64_t counter = 0;
    const uint64_t stream = 0xFFFFFFFFFFFF0000ull + (uint64_t)it;
    auto pick = [&](size_t n) { return (size_t)(draw_u64(params().seed, stream, counter++) % (uint64_t)n); };
    int j = 0, guard = 0;
    while (j < num_gauss_samples && guard < 1000000) {
      guard++;
      const size_t idx = pick(hands.size());  // :125 / :139
      double x[3];
      for (int k = 0; k < 3; k++) x[k] = hands[idx].getGraspSurface()(k) + gaussian((uint64_t)it, &counter) * sigma;
      bool accept = true;
      if (sampling_method_ == MAX) {  // rejection sampling, :135-165
        auto dens = [&](const GraspHypothesis& h) {
          const double d0 = x[0] - h.getGraspSurface()(0), d1 = x[1] - h.getGraspSurface()(1),
                       d2 = x[2] - h.getGraspSurface()(2);
          return term * std::exp((-1.0 / (2.0 * sigma)) * ((d0 * d0 + d1 * d1) + d2 * d2));
        };
        double maxp = 0.0;
        for (const GraspHypothesis& h : hands) maxp = std::max(maxp, dens(h));
        accept = dens(hands[idx]) >= maxp;
      }
      if (accept) {
        for (int k = 0; k < 3; k++) samples(k, j) = x[k];
        j++;
      }
    }
    for (int q = num_samples_is_ - num_rand_samples; q < num_samples_is_; q++) {  // 3.2 (:83-90)
      const ag2::PointXYZRGBA& p = cloud->points[pick(cloud->size())];
      samples(0, q) = (double)p.x;
      samples(1, q) = (double)p.y;
      samples(2, q) = (double)p.z;
    }
    rounds_.push_back(samples);
    // 3.3 (:93-95): one ag2_detect on the resident cloud, grid and normals
    const std::vector<GraspHypothesis> hands_new = detectImpl(cloud_cam, false, &samples, true);
    hands.insert(hands.end(), hands_new.begin(), hands_new.end());
  }
  if (getHandleSearch().getMinInliers() > 0) {  // :104-108
    getHandleSearch().setContext(context());
    hands = getHandleSearch().findClusters(hands);
  }
  return hands;
}
