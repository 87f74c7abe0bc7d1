// types.h -- minimal value types standing in for Eigen / PCL / OpenCV in the host mirror.
//
// The reference's public methods take Eigen matrices, pcl::PointCloud pointers and cv::Mat
// (include/agile_grasp2/*.h).  None of those libraries exist in this build, so the mirror uses the
// small types below, with the same member spelling where the reference's callers touch them
// (operator(), col(), cols(), points, size(), rows/cols/data).  A site that has the real libraries
// converts at this one seam (INTEGRATION.md shows the three-line adapters).
#ifndef AGILE_GRASP2_TYPES_H
#define AGILE_GRASP2_TYPES_H

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace ag2 {

struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double dot(const Vector3d& o) const { return (v[0] * o.v[0] + v[1] * o.v[1]) + v[2] * o.v[2]; }
};

struct Matrix4d {
  double m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  double& operator()(int r, int c) { return m[r][c]; }
  double operator()(int r, int c) const { return m[r][c]; }
  static Matrix4d Identity() { return Matrix4d(); }
  Matrix4d operator*(const Matrix4d& o) const;
  Matrix4d inverse() const;  // general 4x4 (camera poses are rigid, but no assumption is made)
};

// Column-major 3 x N matrix of doubles (Eigen::Matrix3Xd).
struct Matrix3Xd {
  std::vector<double> d;
  Matrix3Xd() = default;
  Matrix3Xd(int rows, int cols) : d((size_t)3 * cols, 0.0) { (void)rows; }
  int rows() const { return 3; }
  int cols() const { return (int)(d.size() / 3); }
  void resize(int rows, int cols) { (void)rows; d.assign((size_t)3 * cols, 0.0); }
  double& operator()(int r, int c) { return d[(size_t)3 * c + r]; }
  double operator()(int r, int c) const { return d[(size_t)3 * c + r]; }
  Vector3d col(int c) const { return Vector3d(d[3 * (size_t)c], d[3 * (size_t)c + 1], d[3 * (size_t)c + 2]); }
  const double* data() const { return d.data(); }
  double* data() { return d.data(); }
};

// Column-major R x N matrix of ints (Eigen::MatrixXi), used for camera_source_.
struct MatrixXi {
  int r = 0;
  std::vector<int32_t> d;
  MatrixXi() = default;
  MatrixXi(int rows, int cols) : r(rows), d((size_t)rows * cols, 0) {}
  static MatrixXi Zero(int rows, int cols) { return MatrixXi(rows, cols); }
  static MatrixXi Ones(int rows, int cols) {
    MatrixXi m(rows, cols);
    std::fill(m.d.begin(), m.d.end(), 1);
    return m;
  }
  int rows() const { return r; }
  int cols() const { return r ? (int)(d.size() / r) : 0; }
  int32_t& operator()(int row, int c) { return d[(size_t)c * r + row]; }
  int32_t operator()(int row, int c) const { return d[(size_t)c * r + row]; }
  const int32_t* data() const { return d.data(); }
  int32_t* data() { return d.data(); }
};

// pcl::PointXYZRGBA: 32 bytes, xyz at offset 0 (what ag2_set_cloud's stride argument is for).
struct PointXYZRGBA {
  float x = 0, y = 0, z = 0, pad0 = 1.f;
  uint32_t rgba = 0;
  uint32_t pad1[3] = {0, 0, 0};
};
static_assert(sizeof(PointXYZRGBA) == 32, "PointXYZRGBA must mirror PCL's 32-byte layout");

// pcl::PointXYZRGBNormal subset used by CloudCamera(const PointCloudNormal::Ptr&, int).
struct PointXYZRGBNormal {
  float x = 0, y = 0, z = 0, pad0 = 1.f;
  float normal_x = 0, normal_y = 0, normal_z = 0, pad1 = 0;
  uint32_t rgba = 0;
  float curvature = 0;
  uint32_t pad2[2] = {0, 0};
};

template <class P>
struct PointCloud {
  std::vector<P> points;
  size_t size() const { return points.size(); }
  typedef std::shared_ptr<PointCloud<P>> Ptr;
};
typedef PointCloud<PointXYZRGBA> PointCloudRGB;
typedef PointCloud<PointXYZRGBNormal> PointCloudNormal;

// cv::Mat stand-in for the 60x60x3 CV_8UC3 grasp images (interleaved HWC).
struct Image {
  int rows = 0, cols = 0, chans = 3;
  std::vector<uint8_t> data;
  int channels() const { return chans; }
  bool empty() const { return data.empty(); }
};

}  // namespace ag2

#endif  // AGILE_GRASP2_TYPES_H
