/*
 * ag2_oracle.cpp -- CPU restatement of the agile_grasp2 hot path
 *   sample -> normals -> local frames -> hand search -> grasp image -> LeNet score -> select.
 *
 * TEST INFRASTRUCTURE ONLY (see ag2_oracle.h).  PARITY UNPINNED against the real reference: it has
 * no tests/fixtures and cannot be built here; third-party numerics (PCL, Eigen, OpenCV, Caffe) are
 * restated from their documented behaviour.  All paths cited below are relative to /root/reference.
 *
 * Arithmetic contract (shared with the HIP kernels, which must reproduce it bit for bit):
 *   - only IEEE-754 +,-,*,/,sqrt,floor and comparisons on the parity-critical path; every
 *     expression is written with explicit parenthesisation and the file is compiled with
 *     -ffp-contract=off, so no FMA contraction and no reassociation happens;
 *   - cos/sin of the hand angles and of the friction cone are computed ONCE on the host (libm) and
 *     enter as constants;
 *   - neighbour lists are enumerated in ascending (grid cell key, original point index) order --
 *     the "canonical order".  The reference gets FLANN's distance-sorted order; order only
 *     influences (a) which neighbour a random draw picks and (b) the last ulp of per-pixel normal
 *     sums, and the reference's own draw is rand()-based and irreproducible.
 */
#include "ag2_oracle.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// small fixed-size helpers (explicit evaluation order everywhere)
// ------------------------------------------------------------------------------------------------
struct V3 {
  double x, y, z;
};
inline double dot3(const V3& a, const V3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross3(const V3& a, const V3& b) {
  // Eigen::MatrixBase::cross: (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 neg3(const V3& a) { return V3{-a.x, -a.y, -a.z}; }
inline double norm3(const V3& a) { return std::sqrt(dot3(a, a)); }

// Cyclic Jacobi eigen-solver for a symmetric 3x3 (stands in for Eigen::EigenSolver at
// src/agile_grasp2/local_frame.cpp:30-32 and for pcl::eigen33 inside pcl::NormalEstimationOMP,
// hand_search.cpp:85-92).  A is overwritten; V's columns are the eigenvectors, d the eigenvalues.
// Only + - * / sqrt fabs: bit-reproducible on the GPU.
void jacobi3(double A[3][3], double V[3][3], double d[3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  static const int PQ[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};  // p, q, r(the other index)
  for (int sweep = 0; sweep < 12; sweep++) {
    if (A[0][1] == 0.0 && A[0][2] == 0.0 && A[1][2] == 0.0) break;
    for (int e = 0; e < 3; e++) {
      const int p = PQ[e][0], q = PQ[e][1], r = PQ[e][2];
      const double apq = A[p][q];
      if (apq == 0.0) continue;
      const double app = A[p][p], aqq = A[q][q];
      // Converged pair: with |apq| <= 2^-60 |aqq - app| the rotation angle is below 2^-60, so the
      // rotation would change neither the diagonal (by t*apq < 2^-120 of the gap) nor, at f64
      // resolution, the eigenvectors (c == 1.0 exactly, s < 2^-60): annihilate without rotating.
      if (std::fabs(apq) <= 0x1p-60 * std::fabs(aqq - app)) {
        A[p][q] = 0.0;
        A[q][p] = 0.0;
        continue;
      }
      const double theta = (aqq - app) / (2.0 * apq);
      double t = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      if (theta < 0.0) t = -t;
      const double c = 1.0 / std::sqrt(t * t + 1.0);
      const double s = t * c;
      A[p][p] = app - t * apq;
      A[q][q] = aqq + t * apq;
      A[p][q] = 0.0;
      A[q][p] = 0.0;
      const double arp = A[r][p], arq = A[r][q];
      const double nrp = c * arp - s * arq;
      const double nrq = s * arp + c * arq;
      A[r][p] = nrp;
      A[p][r] = nrp;
      A[r][q] = nrq;
      A[q][r] = nrq;
      for (int k = 0; k < 3; k++) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
      }
    }
  }
  d[0] = A[0][0];
  d[1] = A[1][1];
  d[2] = A[2][2];
}

inline int argmin3(const double d[3]) {  // first minimum, as Eigen's minCoeff(&index)
  int m = 0;
  if (d[1] < d[m]) m = 1;
  if (d[2] < d[m]) m = 2;
  return m;
}

// Counter-based RNG replacing rand() at hand_search.cpp:130 (the reference's draw is not
// reproducible: shared rand() state inside an OpenMP loop).  Keyed by (seed, global slot, draw).
inline uint64_t draw_u64(uint64_t seed, uint64_t slot, uint64_t j) {
  uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (slot + 1ull));
  x += 0xD1B54A32D192ED03ull * (j + 1ull);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

inline bool finite3f(float a, float b, float c) {
  return std::isfinite(a) && std::isfinite(b) && std::isfinite(c);
}

// ------------------------------------------------------------------------------------------------
// Uniform grid: exact fixed-radius search replacing pcl::KdTreeFLANN (hand_search.cpp:11-12,
// :122, :201, :268).  radiusSearch semantics restated: squared L2 distance in float, strictly
// below (float)(r*r), query point included.
// ------------------------------------------------------------------------------------------------
struct Grid {
  float o[3] = {0, 0, 0};
  float inv = 100.0f;
  int dims[3] = {0, 0, 0};
  int n_valid = 0;
  std::vector<int32_t> perm;        // sorted position -> original index
  std::vector<int32_t> inv_perm;    // original index -> sorted position (or -1)
  std::vector<float> sx, sy, sz;    // coordinates in sorted order
  std::vector<int64_t> cell_start;  // ncells + 1

  inline int cell_of(float v, int a) const { return (int)std::floor((v - o[a]) * inv); }
  inline int64_t key(int cx, int cy, int cz) const {
    return ((int64_t)cz * dims[1] + cy) * dims[0] + cx;
  }

  // origin: nullptr = per-axis minimum of the finite points; else the caller's (a spatial tile passes
  // the minimum of the whole cloud so that it bins its points exactly as the whole cloud would)
  void build(const std::vector<float>& x, const std::vector<float>& y, const std::vector<float>& z,
             double cell, const float* origin = nullptr) {
    const size_t n = x.size();
    float mn[3] = {std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(),
                   std::numeric_limits<float>::infinity()};
    float mx[3] = {-mn[0], -mn[1], -mn[2]};
    n_valid = 0;
    for (size_t i = 0; i < n; i++) {
      if (!finite3f(x[i], y[i], z[i])) continue;
      n_valid++;
      mn[0] = std::min(mn[0], x[i]); mx[0] = std::max(mx[0], x[i]);
      mn[1] = std::min(mn[1], y[i]); mx[1] = std::max(mx[1], y[i]);
      mn[2] = std::min(mn[2], z[i]); mx[2] = std::max(mx[2], z[i]);
    }
    inv = 1.0f / (float)cell;
    perm.assign(n, 0);
    inv_perm.assign(n, -1);
    sx.assign(n, 0); sy.assign(n, 0); sz.assign(n, 0);
    if (n_valid == 0) {
      dims[0] = dims[1] = dims[2] = 0;
      cell_start.assign(1, 0);
      for (size_t i = 0; i < n; i++) perm[i] = (int32_t)i;
      return;
    }
    for (int a = 0; a < 3; a++) o[a] = origin ? origin[a] : mn[a];
    for (int a = 0; a < 3; a++) dims[a] = cell_of(mx[a], a) + 1;
    const int64_t ncells = (int64_t)dims[0] * dims[1] * dims[2];
    std::vector<int64_t> keys(n, ncells);  // invalid points sort to the end
    cell_start.assign(ncells + 1, 0);
    for (size_t i = 0; i < n; i++) {
      if (!finite3f(x[i], y[i], z[i])) continue;
      keys[i] = key(cell_of(x[i], 0), cell_of(y[i], 1), cell_of(z[i], 2));
      cell_start[keys[i] + 1]++;
    }
    for (int64_t c = 0; c < ncells; c++) cell_start[c + 1] += cell_start[c];
    std::vector<int64_t> fill(cell_start.begin(), cell_start.end() - 1);
    int64_t tail = n_valid;
    for (size_t i = 0; i < n; i++) {  // ascending i => stable => (key, index) order
      const int64_t pos = (keys[i] < ncells) ? fill[keys[i]]++ : tail++;
      perm[pos] = (int32_t)i;
      if (keys[i] < ncells) inv_perm[i] = (int32_t)pos;
      sx[pos] = x[i]; sy[pos] = y[i]; sz[pos] = z[i];
    }
  }

  // Calls f(sorted_position) for every point within r of q, ascending sorted position.
  template <class F>
  void radius(const float q[3], double r, F&& f) const {
    if (n_valid == 0) return;
    const float r2f = (float)(r * r);
    const float rq = (float)r * 1.001f;
    int lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
      lo[a] = std::max(cell_of(q[a] - rq, a), 0);
      hi[a] = std::min(cell_of(q[a] + rq, a), dims[a] - 1);
      if (lo[a] > hi[a]) return;
    }
    for (int cz = lo[2]; cz <= hi[2]; cz++)
      for (int cy = lo[1]; cy <= hi[1]; cy++) {
        const int64_t b = cell_start[key(lo[0], cy, cz)];
        const int64_t e = cell_start[key(hi[0], cy, cz) + 1];
        for (int64_t j = b; j < e; j++) {
          const float dx = sx[j] - q[0], dy = sy[j] - q[1], dz = sz[j] - q[2];
          const float d2 = (dx * dx + dy * dy) + dz * dz;
          if (d2 < r2f) f((int32_t)j);
        }
      }
  }
};

struct Hyp {
  ag2o_hypothesis rec;
  std::vector<double> pts;  // 3 x P, unit-box scaled (hand_search.cpp:399-409)
  std::vector<double> nrm;  // 3 x P, rotated normals
};

struct LeNet {
  bool loaded = false;
  std::vector<float> c1w, c1b, c2w, c2b, f1w, f1b, f2w, f2b;
};

}  // namespace

struct ag2o_ctx {
  ag2o_params p;
  std::string err;
  size_t n = 0;
  std::vector<float> x, y, z;          // original order
  std::vector<int32_t> cam;            // n_cams x n col-major
  int n_cams = 1;
  Grid grid;
  bool has_normals = false;
  std::vector<float> nx, ny, nz;       // sorted order, float (see set_cloud)
  float min_z = 0.f;                   // pcl::getMinMax3D, grasp_detector.cpp:152-153
  std::vector<Hyp> hyps;
  LeNet net;
  ag2o_counters cnt;
  // derived hand constants
  double fs[20], fsr[20];              // finger_spacing_(i), finger_spacing_(i) + finger_width_
  std::vector<double> cos_t, sin_t;    // per orientation
  std::vector<double> depths;          // deepenHand depth sequence
  int min_inliers = 0;                 // HandleSearch::setMinInliers; grasp_detector.cpp:59-65
  bool has_origin = false;             // ag2o_set_grid_origin
  float origin[3] = {0, 0, 0};
};

namespace {

void derive(ag2o_ctx* c) {
  const ag2o_params& p = c->p;
  // FingerHand ctor, src/agile_grasp2/finger_hand.cpp:7-12: fs_half = LinSpaced(10, 0, od - fw),
  // finger_spacing = [fs_half - od + fw, fs_half].  Eigen's LinSpaced(i) = low + i*step,
  // step = (high-low)/(n-1).
  const int n = 10;
  const double high = p.hand_outer_diameter - p.finger_width;
  const double step = (high - 0.0) / (double)(n - 1);
  for (int i = 0; i < n; i++) {
    const double h = 0.0 + (double)i * step;
    c->fs[i] = (h - p.hand_outer_diameter) + p.finger_width;
    c->fs[n + i] = h;
  }
  for (int i = 0; i < 2 * n; i++) c->fsr[i] = c->fs[i] + p.finger_width;
  // hand_search.cpp:179-180: angles0 = LinSpaced(R+1, -pi/2, pi/2); first R entries.
  const int R = p.num_orientations;
  c->cos_t.resize(R);
  c->sin_t.resize(R);
  const double low = -1.0 * M_PI / 2.0, hi = M_PI / 2.0;
  const double astep = (hi - low) / (double)R;
  for (int i = 0; i < R; i++) {
    const double a = low + (double)i * astep;
    c->cos_t[i] = std::cos(a);
    c->sin_t[i] = std::sin(a);
  }
  // finger_hand.cpp:118-122: for (depth = min + 0.005; depth <= max; depth += 0.005), f64 accum.
  c->depths.clear();
  const double dstep = 0.005;
  for (double d = p.init_bite + dstep; d <= p.hand_depth; d += dstep) c->depths.push_back(d);
}

// ------------------------------------------------------------------------------------------------
// pcl::NormalEstimationOMP restated (hand_search.cpp:83-94).  Per point: neighbours within
// normals_radius (self included); < 3 neighbours => NaN; single-pass raw moments accumulated in
// float in canonical order (PCL 1.7 computeMeanAndCovarianceMatrix); covariance = E[xx^T] - mm^T;
// eigenvector of the smallest eigenvalue (Jacobi in f64 instead of pcl::eigen33's closed form);
// flipped so that n . (viewpoint - p) >= 0 with viewpoint (0,0,0) (hand_search.cpp:88).
// ------------------------------------------------------------------------------------------------
void normal_at(const ag2o_ctx* c, int32_t pos, float out[3], int64_t* k1) {
  const Grid& g = c->grid;
  const float q[3] = {g.sx[pos], g.sy[pos], g.sz[pos]};
  float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int cnt = 0;
  g.radius(q, c->p.normals_radius, [&](int32_t j) {
    const float px = g.sx[j], py = g.sy[j], pz = g.sz[j];
    a[0] = a[0] + px * px;
    a[1] = a[1] + px * py;
    a[2] = a[2] + px * pz;
    a[3] = a[3] + py * py;
    a[4] = a[4] + py * pz;
    a[5] = a[5] + pz * pz;
    a[6] = a[6] + px;
    a[7] = a[7] + py;
    a[8] = a[8] + pz;
    cnt++;
  });
  *k1 += cnt;
  const float nanf_ = std::numeric_limits<float>::quiet_NaN();
  if (cnt < 3) {
    out[0] = out[1] = out[2] = nanf_;
    return;
  }
  const float fc = (float)cnt;
  for (int i = 0; i < 9; i++) a[i] = a[i] / fc;
  const float c00 = a[0] - a[6] * a[6], c01 = a[1] - a[6] * a[7], c02 = a[2] - a[6] * a[8];
  const float c11 = a[3] - a[7] * a[7], c12 = a[4] - a[7] * a[8], c22 = a[5] - a[8] * a[8];
  double A[3][3] = {{(double)c00, (double)c01, (double)c02},
                    {(double)c01, (double)c11, (double)c12},
                    {(double)c02, (double)c12, (double)c22}};
  double V[3][3], d[3];
  jacobi3(A, V, d);
  const int m = argmin3(d);
  V3 v{V[0][m], V[1][m], V[2][m]};
  const double nv = norm3(v);
  float fx = (float)(v.x / nv), fy = (float)(v.y / nv), fz = (float)(v.z / nv);
  // pcl::flipNormalTowardsViewpoint with vp = 0: vp - p, cos_theta = (vp-p).n; flip if < 0.
  const float vx = 0.0f - q[0], vy = 0.0f - q[1], vz = 0.0f - q[2];
  const float ct = (vx * fx + vy * fy) + vz * fz;
  if (ct < 0.0f) {
    fx = -fx; fy = -fy; fz = -fz;
  }
  out[0] = fx; out[1] = fy; out[2] = fz;
}

struct Frame {
  bool valid = false;
  V3 sample, normal, binormal, curv;
};

// HandSearch::calculateLocalFrames (hand_search.cpp:97-170 / :238-317) +
// LocalFrame::findAverageNormalAxis (local_frame.cpp:26-59).
Frame local_frame(const ag2o_ctx* c, const float q[3], uint64_t slot, uint64_t seed, int64_t* k1) {
  Frame fr;
  const Grid& g = c->grid;
  std::vector<int32_t> nb;
  g.radius(q, c->p.nn_radius_taubin, [&](int32_t j) {
    // NaN-normal neighbours are dropped from the draw (documented fix).
    if (finite3f(c->nx[j], c->ny[j], c->nz[j])) nb.push_back(j);
  });
  *k1 += (int64_t)nb.size();
  if (nb.empty()) return fr;                                      // hand_search.cpp:122
  const int m = std::min(50, (int)nb.size());                     // :124-125
  double N[50][3];
  int votes[2] = {0, 0};
  for (int j = 0; j < m; j++) {
    const uint64_t u = draw_u64(seed, slot, (uint64_t)j);
    const int32_t pick = nb[(size_t)(u % (uint64_t)nb.size())];  // :130 rand() % size
    const double vx = (double)c->nx[pick], vy = (double)c->ny[pick], vz = (double)c->nz[pick];
    const int32_t orig = g.perm[pick];
    for (int cam = 0; cam < c->n_cams; cam++)                     // :137-141
      if (c->cam[(size_t)orig * c->n_cams + cam] == 1) votes[cam]++;
    const double mag = std::sqrt((vx * vx + vy * vy) + vz * vz);  // :148-149
    N[j][0] = vx / mag; N[j][1] = vy / mag; N[j][2] = vz / mag;
  }
  int majority = 0;                                               // :146 maxCoeff (first max)
  if (c->n_cams > 1 && votes[1] > votes[0]) majority = 1;

  // local_frame.cpp:29 M = normals * normals^T
  double M[3][3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double acc = 0.0;
      for (int j = 0; j < m; j++) acc = acc + N[j][a] * N[j][b];
      M[a][b] = acc;
    }
  // symmetrise exactly (M[a][b] and M[b][a] are computed by the same products, so they are equal)
  double V[3][3], d[3];
  jacobi3(M, V, d);
  const int mi = argmin3(d);                                      // :36-38
  V3 cv{V[0][mi], V[1][mi], V[2][mi]};
  const double ncv = norm3(cv);
  cv = V3{cv.x / ncv, cv.y / ncv, cv.z / ncv};
  // :42 argmax_j sum_i (n_i . n_j)^6   (pow(.,6) restated as ((g*g)*(g*g))*(g*g))
  int max_index = 0;
  double best = -1.0;
  for (int j = 0; j < m; j++) {
    double acc = 0.0;
    for (int i = 0; i < m; i++) {
      const double gij = (N[i][0] * N[j][0] + N[i][1] * N[j][1]) + N[i][2] * N[j][2];
      const double g2 = gij * gij;
      acc = acc + (g2 * g2) * g2;
    }
    if (acc > best) {
      best = acc;
      max_index = j;
    }
  }
  // :43-45 normal = normalize((I - c c^T) n_max)
  const double cc[3] = {cv.x, cv.y, cv.z};
  double P[3][3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) P[a][b] = ((a == b) ? 1.0 : 0.0) - cc[a] * cc[b];
  const double* nm = N[max_index];
  V3 np{(P[0][0] * nm[0] + P[0][1] * nm[1]) + P[0][2] * nm[2],
        (P[1][0] * nm[0] + P[1][1] * nm[1]) + P[1][2] * nm[2],
        (P[2][0] * nm[0] + P[2][1] * nm[1]) + P[2][2] * nm[2]};
  const double nn = norm3(np);
  V3 normal{np.x / nn, np.y / nn, np.z / nn};
  V3 binormal = cross3(cv, normal);                               // :48
  const V3 sample{(double)q[0], (double)q[1], (double)q[2]};
  const V3 v{sample.x - c->p.cam_origin[majority][0], sample.y - c->p.cam_origin[majority][1],
             sample.z - c->p.cam_origin[majority][2]};            // :51
  if (dot3(normal, v) > 0.0) normal = neg3(normal);               // :52-53
  if (dot3(binormal, v) > 0.0) binormal = neg3(binormal);         // :54-55
  fr.valid = true;
  fr.sample = sample;
  fr.normal = normal;
  fr.binormal = binormal;
  fr.curv = cross3(normal, binormal);                             // :58
  return fr;
}

// Antipodal::evaluateGrasp, src/agile_grasp2/antipodal.cpp:8-84.  pts/nrm are 3 x P col-major.
int antipodal_label(const double* pts, const double* nrm, int P, double thresh, double cos_fc) {
  double mnx = pts[0], mxx = pts[0];
  for (int j = 1; j < P; j++) {
    mnx = std::min(mnx, pts[3 * j]);
    mxx = std::max(mxx, pts[3 * j]);
  }
  const double lt = mnx + thresh, rt = mxx - thresh;              // :16-17
  int nl = 0, nr = 0;
  double lmaxy = 0, lminy = 0, lmaxz = 0, lminz = 0, rmaxy = 0, rminy = 0, rmaxz = 0, rminz = 0;
  for (int j = 0; j < P; j++) {
    const double ux = pts[3 * j], uy = pts[3 * j + 1], uz = pts[3 * j + 2];
    const double yx = nrm[3 * j], yy = nrm[3 * j + 1], yz = nrm[3 * j + 2];
    const double ldot = (-1.0 * yx + 0.0 * yy) + 0.0 * yz;        // :20-25 l^T normals
    const double rdot = (1.0 * yx + 0.0 * yy) + 0.0 * yz;
    if (ldot > cos_fc && ux < lt) {                               // :38-39
      if (nl == 0) { lmaxy = lminy = uy; lmaxz = lminz = uz; }
      else { lmaxy = std::max(lmaxy, uy); lminy = std::min(lminy, uy);
             lmaxz = std::max(lmaxz, uz); lminz = std::min(lminz, uz); }
      nl++;
    }
    if (rdot > cos_fc && ux > rt) {                               // :40-41
      if (nr == 0) { rmaxy = rminy = uy; rmaxz = rminz = uz; }
      else { rmaxy = std::max(rmaxy, uy); rminy = std::min(rminy, uy);
             rmaxz = std::max(rmaxz, uz); rminz = std::min(rminz, uz); }
      nr++;
    }
  }
  int result = 0;
  if (nl > 0 || nr > 0) result = 1;                               // :48-51
  if (nl > 0 && nr > 0) {                                         // :54-81
    const double top_y = std::min(lmaxy, rmaxy), bot_y = std::max(lminy, rminy);
    const double top_z = std::min(lmaxz, rmaxz), bot_z = std::max(lminz, rminz);
    if (top_y > bot_y && top_z > bot_z) result = 2;
  }
  return result;
}

// HandSearch::evaluateHands (hand_search.cpp:173-235) for one frame + HandSearch::calculateHand
// (:319-426) + FingerHand (finger_hand.cpp:17-214, :313-325).
void sweep_sample(const ag2o_ctx* c, const Frame& fr, const float q[3], int32_t slot,
                  std::vector<Hyp>& out, int64_t* k2, int64_t* kcrop, int64_t* psum) {
  const ag2o_params& p = c->p;
  const Grid& g = c->grid;
  const double hh = p.hand_height;
  const double F[3][3] = {{fr.normal.x, fr.binormal.x, fr.curv.x},
                          {fr.normal.y, fr.binormal.y, fr.curv.y},
                          {fr.normal.z, fr.binormal.z, fr.curv.z}};  // :325-326
  std::vector<double> P, Q;  // cropped points / normals, 3 x K'
  int64_t nk2 = 0;
  g.radius(q, p.nn_radius_hands, [&](int32_t j) {
    nk2++;
    // :209-210 centered = (cloud point - sample) in float, then cast to double
    const double p0 = (double)(g.sx[j] - q[0]), p1 = (double)(g.sy[j] - q[1]),
                 p2 = (double)(g.sz[j] - q[2]);
    const double zf = (F[0][2] * p0 + F[1][2] * p1) + F[2][2] * p2;  // row 2 of frame^T * points
    if (zf > -1.0 * hh && zf < hh) {                                 // :333
      P.push_back(p0); P.push_back(p1); P.push_back(p2);
      Q.push_back((double)c->nx[j]); Q.push_back((double)c->ny[j]); Q.push_back((double)c->nz[j]);
    }
  });
  *k2 += nk2;
  if (nk2 == 0) return;  // :201
  const int K = (int)(P.size() / 3);
  *kcrop += K;
  const int R = p.num_orientations;
  const double cos_fc = std::cos(30.0 * M_PI / 180.0);  // antipodal.cpp:11,23
  std::vector<double> X(3 * (size_t)K);
  for (int oi = 0; oi < R; oi++) {
    const double cs = c->cos_t[oi], sn = c->sin_t[oi];
    const double rot[3][3] = {{cs, -1.0 * sn, 0.0}, {sn, cs, 0.0}, {0.0, 0.0, 1.0}};  // :356
    double Fr[3][3];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        Fr[a][b] = (F[a][0] * rot[0][b] + F[a][1] * rot[1][b]) + F[a][2] * rot[2][b];  // :357
    for (int j = 0; j < K; j++) {                                                       // :358
      const double p0 = P[3 * j], p1 = P[3 * j + 1], p2 = P[3 * j + 2];
      X[3 * j + 0] = (Fr[0][0] * p0 + Fr[1][0] * p1) + Fr[2][0] * p2;
      X[3 * j + 1] = (Fr[0][1] * p0 + Fr[1][1] * p1) + Fr[2][1] * p2;
      X[3 * j + 2] = (Fr[0][2] * p0 + Fr[1][2] * p1) + Fr[2][2] * p2;
    }
    // ---- evaluateFingers(points_rot, init_bite), finger_hand.cpp:17-72
    double top = p.init_bite, bottom = p.init_bite - p.hand_depth;
    bool free_[20];
    for (int k = 0; k < 20; k++) free_[k] = false;
    {
      bool back_collision = false, any = false;
      for (int j = 0; j < K; j++)
        if (X[3 * j + 1] < top) {
          any = true;
          if (X[3 * j + 1] < bottom) back_collision = true;  // :35-36 early return
        }
      if (!back_collision && any) {
        for (int k = 0; k < 20; k++) {
          int num = 0;
          for (int j = 0; j < K; j++)
            if (X[3 * j + 1] < top && X[3 * j] > c->fs[k] && X[3 * j] < c->fsr[k]) num++;
          free_[k] = (num == 0);
        }
      }
    }
    int nfree = 0;
    for (int k = 0; k < 20; k++) nfree += free_[k] ? 1 : 0;
    if (!(nfree > 2)) continue;  // hand_search.cpp:366
    // ---- evaluateHand, finger_hand.cpp:313-325
    int valid[10], nvalid = 0;
    for (int k = 0; k < 10; k++)
      if (free_[k] && free_[10 + k]) valid[nvalid++] = k;
    if (!(nvalid > 0)) continue;  // hand_search.cpp:370
    // ---- deepenHand, finger_hand.cpp:96-134
    const int idx = valid[(int)std::ceil(nvalid / 2.0) - 1];
    for (size_t di = 0; di < c->depths.size(); di++) {
      const double d = c->depths[di];
      const double t_ = d, b_ = d - p.hand_depth;
      bool back = false, blocked = false;
      for (int j = 0; j < K; j++) {
        const double xy = X[3 * j + 1];
        if (xy < t_) {
          if (xy < b_) back = true;
          const double xx = X[3 * j];
          if ((xx > c->fs[idx] && xx < c->fsr[idx]) || (xx > c->fs[10 + idx] && xx < c->fsr[10 + idx]))
            blocked = true;
        }
      }
      if (back || blocked) break;  // :125-126 (sum of the two tested fingers < 2)
      top = t_;
      bottom = b_;
    }
    // ---- computePointsInClosingRegion, finger_hand.cpp:137-180
    const double left = c->fs[idx] + p.finger_width;
    const double right = c->fs[10 + idx];
    const double center = 0.5 * (left + right);
    double surface = X[1];
    for (int j = 1; j < K; j++) surface = std::min(surface, X[3 * j + 1]);  // :158
    std::vector<int> box;
    for (int j = 0; j < K; j++)
      if (X[3 * j + 1] < top && X[3 * j] > left && X[3 * j] < right) box.push_back(j);
    if (box.empty()) continue;  // hand_search.cpp:377-381
    // ---- calculateGraspParameters, finger_hand.cpp:183-199 (columns 0..2)
    Hyp h;
    std::memset(&h.rec, 0, sizeof(h.rec));
    const double smp[3] = {fr.sample.x, fr.sample.y, fr.sample.z};
    const double ys[3] = {surface, bottom, top};
    double* dst[3] = {h.rec.surface, h.rec.bottom, h.rec.top};
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++)
        dst[k][a] = ((Fr[a][0] * center + Fr[a][1] * ys[k]) + Fr[a][2] * 0.0) + smp[a];
    for (int a = 0; a < 3; a++) {  // hand_search.cpp:383-385
      h.rec.binormal[a] = Fr[a][0];
      h.rec.approach[a] = Fr[a][1];
      h.rec.axis[a] = Fr[a][2];
    }
    const int Pn = (int)box.size();
    h.pts.resize(3 * (size_t)Pn);
    h.nrm.resize(3 * (size_t)Pn);
    double mnx = X[3 * box[0]], mxx = mnx;
    for (int b = 0; b < Pn; b++) {
      const int j = box[b];
      mnx = std::min(mnx, X[3 * j]);
      mxx = std::max(mxx, X[3 * j]);
      const double q0 = Q[3 * j], q1 = Q[3 * j + 1], q2 = Q[3 * j + 2];  // :359-360, :394
      h.nrm[3 * b + 0] = (Fr[0][0] * q0 + Fr[1][0] * q1) + Fr[2][0] * q2;
      h.nrm[3 * b + 1] = (Fr[0][1] * q0 + Fr[1][1] * q1) + Fr[2][1] * q2;
      h.nrm[3 * b + 2] = (Fr[0][2] * q0 + Fr[1][2] * q1) + Fr[2][2] * q2;
    }
    h.rec.width = mxx - mnx;  // :397
    // :400-409 scale into the unit box
    const double baseline = 0.1;
    const double left_const = left - 0.5 * (baseline - (right - left));
    const double lower[3] = {left_const, bottom, -1.0 * hh};
    const double scales[3] = {1.0 / baseline, 1.0 / (top - bottom), 1.0 / (2.0 * hh)};
    for (int b = 0; b < Pn; b++) {
      const int j = box[b];
      for (int a = 0; a < 3; a++) h.pts[3 * b + a] = scales[a] * (X[3 * j + a] - lower[a]);
    }
    const int label = antipodal_label(h.pts.data(), h.nrm.data(), Pn, 0.003, cos_fc);  // :415-416
    h.rec.half_antipodal = (label >= 1) ? 1 : 0;
    h.rec.full_antipodal = (label == 2) ? 1 : 0;
    h.rec.score = 0.0;
    h.rec.sample_slot = slot;
    h.rec.orientation = oi;
    h.rec.n_points = Pn;
    *psum += Pn;
    out.push_back(std::move(h));
  }
}

// GraspDetector::pruneGraspsOnHandParameters, grasp_detector.cpp:363-395.  Bounds are float in the
// reference signature (:363-364); min_z from pcl::getMinMax3D (:152-153).
bool prune_keep(const ag2o_ctx* c, const ag2o_hypothesis& h) {
  const ag2o_params& p = c->p;
  if (p.filter_half_grasps && !h.half_antipodal) return false;
  const float min_x = (float)p.workspace[0], max_x = (float)p.workspace[1];
  const float min_y = (float)p.workspace[2], max_y = (float)p.workspace[3];
  const float min_z = c->min_z;
  const double hw = 0.5 * p.hand_outer_diameter;
  double pts[5][3];
  for (int a = 0; a < 3; a++) {
    pts[0][a] = h.bottom[a] + hw * h.binormal[a];
    pts[1][a] = h.bottom[a] - hw * h.binormal[a];
    pts[2][a] = h.top[a] + hw * h.binormal[a];
    pts[3][a] = h.top[a] - hw * h.binormal[a];
    pts[4][a] = h.bottom[a] - 0.10 * h.approach[a];
  }
  double mn[3], mx[3];
  for (int a = 0; a < 3; a++) {
    mn[a] = mx[a] = pts[0][a];
    for (int k = 1; k < 5; k++) {
      mn[a] = std::min(mn[a], pts[k][a]);
      mx[a] = std::max(mx[a], pts[k][a]);
    }
  }
  const double ap = h.width;
  return ap >= p.min_aperture && ap <= p.max_aperture && mn[2] >= (double)min_z &&
         mn[1] >= (double)min_y && mx[1] <= (double)max_y && mn[0] >= (double)min_x &&
         mx[0] <= (double)max_x;
}

// Learning::convertToImageRGB (learning.cpp:143-209) + convertTo(CV_8UC3, 255.0) (:16).
// A scatter formulation of the reference's 3600 x P scan: per cell the normals are summed in
// in-box order, exactly as the reference's inner j-loop (:166-179) visits them.
void render_image(const double* pts, const double* nrm, int P, uint8_t* out_hwc) {
  const int S = 60;
  std::vector<double> acc(3 * S * S, 0.0);
  std::vector<int> cnt(S * S, 0);
  double miny = 0.0;
  if (P > 0) {
    miny = pts[1];
    for (int j = 1; j < P; j++) miny = std::min(miny, pts[3 * j + 1]);  // :148-149
  }
  const double cellsize = 1.0 / (double)S;  // :152
  for (int j = 0; j < P; j++) {
    const double fx = std::floor(pts[3 * j] / cellsize);
    const double fy = std::floor((pts[3 * j + 1] - miny) / cellsize);
    if (!(std::fabs(fx) < 1.0e9) || !(std::fabs(fy) < 1.0e9)) continue;  // NaN / huge: no cell
    const long cell = (long)fx + (long)fy * S;  // :156 (x-cells >= 60 alias into the next row)
    if (cell < 0 || cell >= S * S) continue;
    acc[3 * cell + 0] = acc[3 * cell + 0] + nrm[3 * j + 0];
    acc[3 * cell + 1] = acc[3 * cell + 1] + nrm[3 * j + 1];
    acc[3 * cell + 2] = acc[3 * cell + 2] + nrm[3 * j + 2];
    cnt[cell]++;
  }
  // u8 quantisation is monotone, so quantise-then-dilate == dilate-then-quantise (bit for bit).
  std::vector<uint8_t> img(3 * S * S, 0);  // [row][col][ch], channel order before BGR2RGB
  for (int cell = 0; cell < S * S; cell++) {
    if (cnt[cell] == 0) continue;
    const double ax = acc[3 * cell], ay = acc[3 * cell + 1], az = acc[3 * cell + 2];
    const double s = 1.0 / std::sqrt((ax * ax + ay * ay) + az * az);  // :183
    const double v[3] = {std::fabs(s * ax), std::fabs(s * ay), std::fabs(s * az)};
    const int row = S - 1 - cell / S, col = cell % S;  // :188-190
    for (int ch = 0; ch < 3; ch++) {
      const float f = (float)v[ch];          // cv::Vec3f from doubles
      const float t = f * 255.0f;            // convertTo(..., 255.0): float multiply
      uint8_t u = 0;
      if (t == t) {                          // NaN -> 0 (documented: library-dependent in OpenCV)
        const float r = std::nearbyint(t);   // cvRound: round half to even
        u = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
      }
      img[(row * S + col) * 3 + ch] = u;
    }
  }
  // :202-203 3x3 rect dilate (per-channel max, out-of-image taps ignored); :206 swap ch 0 <-> 2
  for (int r = 0; r < S; r++)
    for (int cc = 0; cc < S; cc++)
      for (int ch = 0; ch < 3; ch++) {
        uint8_t m = 0;
        for (int dr = -1; dr <= 1; dr++)
          for (int dc = -1; dc <= 1; dc++) {
            const int rr = r + dr, c2 = cc + dc;
            if (rr < 0 || rr >= S || c2 < 0 || c2 >= S) continue;
            m = std::max(m, img[(rr * S + c2) * 3 + ch]);
          }
        out_hwc[(r * S + cc) * 3 + (2 - ch)] = m;
      }
}

// Caffe layers of caffe/test_1batch2.prototxt:1-92 restated (Classifier::PredictBatch,
// caffe_classifier.cpp:94-127; PreprocessBatch :158-198: u8 HWC -> float planar, no mean/scale).
// Accumulation order per output: (c, ky, kx) ascending for conv, k ascending for inner product.
void lenet_forward(const LeNet& net, const uint8_t* images, size_t n, float* out, int nthreads) {
  const int B = 32;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
  for (long b0 = 0; b0 < (long)n; b0 += B) {
    const int nb = (int)std::min<long>(B, (long)n - b0);
    std::vector<float> in(3 * 60 * 60), c1(20 * 56 * 56), p1(20 * 28 * 28), c2(50 * 24 * 24);
    std::vector<float> p2t((size_t)7200 * B, 0.f), h1((size_t)500 * B);
    for (int bi = 0; bi < nb; bi++) {
      const uint8_t* im = images + (size_t)(b0 + bi) * 10800;
      for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < 3600; i++) in[ch * 3600 + i] = (float)im[i * 3 + ch];
      for (int oc = 0; oc < 20; oc++) {
        float* o = &c1[oc * 3136];
        for (int i = 0; i < 3136; i++) o[i] = 0.f;
        for (int c = 0; c < 3; c++)
          for (int ky = 0; ky < 5; ky++)
            for (int kx = 0; kx < 5; kx++) {
              const float w = net.c1w[((oc * 3 + c) * 5 + ky) * 5 + kx];
              for (int y = 0; y < 56; y++) {
                const float* src = &in[c * 3600 + (y + ky) * 60 + kx];
                float* d = &o[y * 56];
                for (int x = 0; x < 56; x++) d[x] = d[x] + w * src[x];
              }
            }
        const float bias = net.c1b[oc];
        for (int i = 0; i < 3136; i++) o[i] = o[i] + bias;
      }
      for (int oc = 0; oc < 20; oc++)
        for (int y = 0; y < 28; y++)
          for (int x = 0; x < 28; x++) {
            const float* s = &c1[oc * 3136 + (2 * y) * 56 + 2 * x];
            p1[oc * 784 + y * 28 + x] = std::max(std::max(s[0], s[1]), std::max(s[56], s[57]));
          }
      for (int oc = 0; oc < 50; oc++) {
        float* o = &c2[oc * 576];
        for (int i = 0; i < 576; i++) o[i] = 0.f;
        for (int c = 0; c < 20; c++)
          for (int ky = 0; ky < 5; ky++)
            for (int kx = 0; kx < 5; kx++) {
              const float w = net.c2w[((oc * 20 + c) * 5 + ky) * 5 + kx];
              for (int y = 0; y < 24; y++) {
                const float* src = &p1[c * 784 + (y + ky) * 28 + kx];
                float* d = &o[y * 24];
                for (int x = 0; x < 24; x++) d[x] = d[x] + w * src[x];
              }
            }
        const float bias = net.c2b[oc];
        for (int i = 0; i < 576; i++) o[i] = o[i] + bias;
      }
      for (int oc = 0; oc < 50; oc++)
        for (int y = 0; y < 12; y++)
          for (int x = 0; x < 12; x++) {
            const float* s = &c2[oc * 576 + (2 * y) * 24 + 2 * x];
            p2t[(size_t)(oc * 144 + y * 12 + x) * B + bi] =
                std::max(std::max(s[0], s[1]), std::max(s[24], s[25]));
          }
    }
    // ip1 (500 x 7200) + ReLU, batch-innermost so the k-ordered chain vectorises across images
    for (int o = 0; o < 500; o++) {
      float* hrow = &h1[(size_t)o * B];
      for (int bi = 0; bi < B; bi++) hrow[bi] = 0.f;
      const float* w = &net.f1w[(size_t)o * 7200];
      for (int k = 0; k < 7200; k++) {
        const float wk = w[k];
        const float* xs = &p2t[(size_t)k * B];
        for (int bi = 0; bi < B; bi++) hrow[bi] = hrow[bi] + wk * xs[bi];
      }
      const float bias = net.f1b[o];
      for (int bi = 0; bi < B; bi++) hrow[bi] = std::max(hrow[bi] + bias, 0.f);
    }
    for (int bi = 0; bi < nb; bi++)
      for (int o = 0; o < 2; o++) {
        float acc = 0.f;
        for (int k = 0; k < 500; k++) acc = acc + net.f2w[o * 500 + k] * h1[(size_t)k * B + bi];
        out[(size_t)(b0 + bi) * 2 + o] = acc + net.f2b[o];
      }
  }
}

int fail(ag2o_ctx* c, const char* msg) {
  c->err = msg;
  return -1;
}

bool sample_query(const ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t i,
                  float q[3]) {
  if (sample_idx) {
    const int32_t id = sample_idx[i];
    if (id < 0 || (size_t)id >= c->n) return false;
    q[0] = c->x[id]; q[1] = c->y[id]; q[2] = c->z[id];
  } else {
    // hand_search.cpp:261-263: sample.x = samples(0,i) (double -> float)
    q[0] = (float)sample_xyz[3 * i]; q[1] = (float)sample_xyz[3 * i + 1];
    q[2] = (float)sample_xyz[3 * i + 2];
  }
  return finite3f(q[0], q[1], q[2]);
}

int generate(ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
             uint64_t slot_base, uint64_t seed) {
  if (!c->has_normals) return fail(c, "normals missing: call ag2o_compute_normals or pass normals");
  if ((sample_idx == nullptr) == (sample_xyz == nullptr))
    return fail(c, "exactly one of sample_idx / sample_xyz must be given");
  c->hyps.clear();
  std::vector<std::vector<Hyp>> lists(s);
  std::vector<Frame> frames(s);
  int64_t k1 = 0, k2 = 0, kc = 0, ps = 0, nfr = 0;
  const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 8) num_threads(c->p.num_threads) reduction(+ : k1, nfr)
  for (long i = 0; i < (long)s; i++) {
    float q[3];
    if (!sample_query(c, sample_idx, sample_xyz, (size_t)i, q)) continue;
    frames[i] = local_frame(c, q, slot_base + (uint64_t)i, seed, &k1);
    if (frames[i].valid) nfr++;
  }
  const double t1 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->p.num_threads) reduction(+ : k2, kc, ps)
  for (long i = 0; i < (long)s; i++) {
    if (!frames[i].valid) continue;
    float q[3];
    sample_query(c, sample_idx, sample_xyz, (size_t)i, q);
    sweep_sample(c, frames[i], q, (int32_t)(slot_base + (uint64_t)i), lists[i], &k2, &kc, &ps);
  }
  for (size_t i = 0; i < s; i++)  // hand_search.cpp:223-228 concatenate in sample order
    for (auto& h : lists[i]) c->hyps.push_back(std::move(h));
  const double t2 = omp_get_wtime();
  c->cnt.n_samples = (int64_t)s;
  c->cnt.n_frames = nfr;
  c->cnt.n_hypotheses = (int64_t)c->hyps.size();
  c->cnt.sum_k1 += k1;
  c->cnt.sum_k2 = k2;
  c->cnt.sum_kcrop = kc;
  c->cnt.sum_p = ps;
  c->cnt.t_frames = t1 - t0;
  c->cnt.t_hands = t2 - t1;
  return 0;
}

}  // namespace

extern "C" {

void ag2o_default_params(ag2o_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->finger_width = 0.01;
  p->hand_outer_diameter = 0.09;
  p->hand_depth = 0.06;
  p->hand_height = 0.02;
  p->init_bite = 0.015;
  p->nn_radius_taubin = 0.01;
  p->nn_radius_hands = 0.1;
  p->normals_radius = 0.01;
  p->grid_cell = 0.01;
  p->num_orientations = 8;
  p->num_threads = 1;
  p->n_cams = 1;
  p->filter_half_grasps = 1;
  const double ws[6] = {-1e30, 1e30, -1e30, 1e30, -1e30, 1e30};
  for (int i = 0; i < 6; i++) p->workspace[i] = ws[i];
  p->min_aperture = 0.03;
  p->max_aperture = 0.07;
  p->min_score_diff = 500.0;
  p->num_selected = 50;
}

ag2o_ctx* ag2o_create(const ag2o_params* p) {
  ag2o_ctx* c = new ag2o_ctx();
  c->p = *p;
  if (c->p.num_threads < 1) c->p.num_threads = 1;
  if (c->p.num_orientations < 1) c->p.num_orientations = 1;
  std::memset(&c->cnt, 0, sizeof(c->cnt));
  derive(c);
  return c;
}

void ag2o_destroy(ag2o_ctx* c) { delete c; }

int ag2o_hand_constants(const ag2o_params* p, double* finger_spacing20, double* angles,
                        double* depths32, int32_t* n_depths) {
  if (!p || p->num_orientations < 1) return -1;
  ag2o_ctx* c = ag2o_create(p);
  if (finger_spacing20) std::memcpy(finger_spacing20, c->fs, sizeof(c->fs));
  const double low = -1.0 * M_PI / 2.0, astep = (M_PI / 2.0 - low) / (double)c->p.num_orientations;
  if (angles)
    for (int i = 0; i < c->p.num_orientations; i++) angles[i] = low + (double)i * astep;
  const size_t nd = std::min<size_t>(c->depths.size(), 32);
  if (depths32) std::memcpy(depths32, c->depths.data(), nd * sizeof(double));
  if (n_depths) *n_depths = (int32_t)nd;
  ag2o_destroy(c);
  return 0;
}
const char* ag2o_last_error(const ag2o_ctx* c) { return c ? c->err.c_str() : "null context"; }

int ag2o_set_cloud(ag2o_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                   const int32_t* cam_source, int n_cams, const double* normals) {
  if (!c) return -1;
  if (n_cams < 1 || n_cams > 2) return fail(c, "n_cams must be 1 or 2");
  if (stride_bytes < 12 || stride_bytes % 4 != 0) return fail(c, "bad stride");
  c->n = n;
  c->n_cams = n_cams;
  c->x.resize(n); c->y.resize(n); c->z.resize(n);
  const char* base = (const char*)xyz;
  for (size_t i = 0; i < n; i++) {
    const float* pt = (const float*)(base + i * stride_bytes);
    c->x[i] = pt[0]; c->y[i] = pt[1]; c->z[i] = pt[2];
  }
  c->cam.assign(n * (size_t)n_cams, 1);  // cloud_camera.cpp:59 file ctor: ones
  if (cam_source) std::copy(cam_source, cam_source + n * (size_t)n_cams, c->cam.begin());
  if (c->has_origin)
    for (size_t i = 0; i < n; i++)
      if (finite3f(c->x[i], c->y[i], c->z[i]) &&
          (c->x[i] < c->origin[0] || c->y[i] < c->origin[1] || c->z[i] < c->origin[2]))
        return fail(c, "set_cloud: a point lies below the grid origin given to ag2o_set_grid_origin");
  c->grid.build(c->x, c->y, c->z, c->p.grid_cell, c->has_origin ? c->origin : nullptr);
  // pcl::getMinMax3D over finite points, grasp_detector.cpp:152-153 (a tile: the whole cloud's
  // minimum, which is what the caller passed as the origin)
  c->min_z = std::numeric_limits<float>::infinity();
  for (size_t i = 0; i < n; i++)
    if (finite3f(c->x[i], c->y[i], c->z[i])) c->min_z = std::min(c->min_z, c->z[i]);
  if (c->has_origin) c->min_z = c->origin[2];
  c->nx.assign(n, 0.f); c->ny.assign(n, 0.f); c->nz.assign(n, 0.f);
  c->has_normals = false;
  if (normals) {
    // cloud_camera.cpp:27-31: normals come from float PointNormal fields, so they are
    // float-representable; stored as float, widened on use (hand_search.cpp:28, :93).
    for (size_t pos = 0; pos < n; pos++) {
      const size_t i = (size_t)c->grid.perm[pos];
      c->nx[pos] = (float)normals[3 * i]; c->ny[pos] = (float)normals[3 * i + 1];
      c->nz[pos] = (float)normals[3 * i + 2];
    }
    c->has_normals = true;
  }
  std::memset(&c->cnt, 0, sizeof(c->cnt));
  c->cnt.n_points = (int64_t)n;
  c->cnt.n_valid_points = c->grid.n_valid;
  c->hyps.clear();
  return 0;
}

int ag2o_compute_normals(ag2o_ctx* c) {
  if (!c) return -1;
  const double t0 = omp_get_wtime();
  const long nv = c->grid.n_valid;
  int64_t k1 = 0;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(c->p.num_threads) reduction(+ : k1)
  for (long pos = 0; pos < nv; pos++) {
    float o[3];
    normal_at(c, (int32_t)pos, o, &k1);
    c->nx[pos] = o[0]; c->ny[pos] = o[1]; c->nz[pos] = o[2];
  }
  const float nanf_ = std::numeric_limits<float>::quiet_NaN();
  for (size_t pos = (size_t)nv; pos < c->n; pos++) c->nx[pos] = c->ny[pos] = c->nz[pos] = nanf_;
  c->has_normals = true;
  c->cnt.sum_k1 = k1;
  c->cnt.t_normals = omp_get_wtime() - t0;
  return 0;
}

int ag2o_get_normals(ag2o_ctx* c, double* out) {
  if (!c || !c->has_normals) return c ? fail(c, "no normals") : -1;
  for (size_t pos = 0; pos < c->n; pos++) {
    const size_t i = (size_t)c->grid.perm[pos];
    out[3 * i] = (double)c->nx[pos]; out[3 * i + 1] = (double)c->ny[pos];
    out[3 * i + 2] = (double)c->nz[pos];
  }
  return 0;
}

int ag2o_get_grid_perm(ag2o_ctx* c, int32_t* perm) {
  if (!c) return -1;
  std::copy(c->grid.perm.begin(), c->grid.perm.end(), perm);
  return 0;
}

int ag2o_radius_search(ag2o_ctx* c, const float* q, double r, int32_t* out, size_t cap,
                       size_t* n_out) {
  if (!c) return -1;
  size_t k = 0;
  c->grid.radius(q, r, [&](int32_t j) {
    if (k < cap) out[k] = c->grid.perm[j];
    k++;
  });
  *n_out = k;
  return (k > cap) ? fail(c, "radius_search: output capacity too small") : 0;
}

int ag2o_local_frames(ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                      uint64_t slot_base, uint64_t seed, double* fo, int32_t* valid) {
  if (!c || !c->has_normals) return c ? fail(c, "no normals") : -1;
  for (size_t i = 0; i < s; i++) {
    float q[3];
    int64_t k1 = 0;
    Frame f;
    if (sample_query(c, sample_idx, sample_xyz, i, q)) f = local_frame(c, q, slot_base + i, seed, &k1);
    valid[i] = f.valid ? 1 : 0;
    const V3 v[4] = {f.sample, f.normal, f.binormal, f.curv};
    for (int k = 0; k < 4; k++) {
      fo[12 * i + 3 * k] = f.valid ? v[k].x : 0.0;
      fo[12 * i + 3 * k + 1] = f.valid ? v[k].y : 0.0;
      fo[12 * i + 3 * k + 2] = f.valid ? v[k].z : 0.0;
    }
  }
  return 0;
}

int ag2o_generate_hypotheses(ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz,
                             size_t s, uint64_t slot_base, uint64_t seed, ag2o_hypothesis* out,
                             size_t cap, size_t* n_out) {
  if (!c) return -1;
  const int rc = generate(c, sample_idx, sample_xyz, s, slot_base, seed);
  if (rc) return rc;
  *n_out = c->hyps.size();
  if (c->hyps.size() > cap) return fail(c, "generate_hypotheses: output capacity too small");
  for (size_t h = 0; h < c->hyps.size(); h++) out[h] = c->hyps[h].rec;
  return 0;
}

int ag2o_hyp_points(ag2o_ctx* c, size_t h, double* pts, double* nrm) {
  if (!c || h >= c->hyps.size()) return c ? fail(c, "hypothesis index out of range") : -1;
  std::copy(c->hyps[h].pts.begin(), c->hyps[h].pts.end(), pts);
  std::copy(c->hyps[h].nrm.begin(), c->hyps[h].nrm.end(), nrm);
  return 0;
}

int ag2o_prune(ag2o_ctx* c, uint8_t* keep, size_t n) {
  if (!c || n != c->hyps.size()) return c ? fail(c, "prune: size mismatch") : -1;
  for (size_t h = 0; h < n; h++) keep[h] = prune_keep(c, c->hyps[h].rec) ? 1 : 0;
  return 0;
}

int ag2o_render_images(ag2o_ctx* c, size_t first, size_t count, uint8_t* out) {
  if (!c || first + count > c->hyps.size()) return c ? fail(c, "render: range") : -1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(c->p.num_threads)
  for (long i = 0; i < (long)count; i++) {
    const Hyp& h = c->hyps[first + (size_t)i];
    render_image(h.pts.data(), h.nrm.data(), h.rec.n_points, out + (size_t)i * 10800);
  }
  return 0;
}

int ag2o_render_image_from_points(const double* pts, const double* nrm, size_t p, uint8_t* out) {
  render_image(pts, nrm, (int)p, out);
  return 0;
}

int ag2o_lenet_load(ag2o_ctx* c, const float* c1w, const float* c1b, const float* c2w,
                    const float* c2b, const float* f1w, const float* f1b, const float* f2w,
                    const float* f2b) {
  if (!c) return -1;
  LeNet& n = c->net;
  n.c1w.assign(c1w, c1w + 20 * 3 * 25);
  n.c1b.assign(c1b, c1b + 20);
  n.c2w.assign(c2w, c2w + 50 * 20 * 25);
  n.c2b.assign(c2b, c2b + 50);
  n.f1w.assign(f1w, f1w + 500 * 7200);
  n.f1b.assign(f1b, f1b + 500);
  n.f2w.assign(f2w, f2w + 2 * 500);
  n.f2b.assign(f2b, f2b + 2);
  n.loaded = true;
  return 0;
}

int ag2o_lenet_forward(ag2o_ctx* c, const uint8_t* images, size_t n, float* out) {
  if (!c || !c->net.loaded) return c ? fail(c, "lenet weights not loaded") : -1;
  lenet_forward(c->net, images, n, out, c->p.num_threads);
  return 0;
}

int ag2o_detect(ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                uint64_t slot_base, uint64_t seed, int do_prune, ag2o_hypothesis* selected,
                size_t cap, size_t* n_selected, ag2o_hypothesis* scored_all, size_t cap_all,
                size_t* n_scored) {
  if (!c || !c->net.loaded) return c ? fail(c, "lenet weights not loaded") : -1;
  const double t0 = omp_get_wtime();
  const int rc = generate(c, sample_idx, sample_xyz, s, slot_base, seed);
  if (rc) return rc;
  // grasp_detector.cpp:149-160: prune unless explicit indices were configured
  std::vector<size_t> kept;
  for (size_t h = 0; h < c->hyps.size(); h++)
    if (!do_prune || prune_keep(c, c->hyps[h].rec)) kept.push_back(h);
  c->cnt.n_pruned = (int64_t)kept.size();
  const double t1 = omp_get_wtime();
  std::vector<uint8_t> imgs(kept.size() * 10800);
#pragma omp parallel for schedule(dynamic, 16) num_threads(c->p.num_threads)
  for (long i = 0; i < (long)kept.size(); i++) {
    const Hyp& h = c->hyps[kept[(size_t)i]];
    render_image(h.pts.data(), h.nrm.data(), h.rec.n_points, &imgs[(size_t)i * 10800]);
  }
  const double t2 = omp_get_wtime();
  std::vector<float> logits(kept.size() * 2);
  lenet_forward(c->net, imgs.data(), kept.size(), logits.data(), c->p.num_threads);
  const double t3 = omp_get_wtime();
  // grasp_detector.cpp:198-207 (batch slicing bug at :192-204 NOT reproduced: every image scored)
  std::vector<ag2o_hypothesis> anti;
  for (size_t i = 0; i < kept.size(); i++) {
    ag2o_hypothesis r = c->hyps[kept[i]].rec;
    const float score = logits[2 * i + 1] - logits[2 * i];
    r.score = (double)score;
    if (scored_all && i < cap_all) scored_all[i] = r;
    if ((double)score >= c->p.min_score_diff) {
      r.full_antipodal = 1;
      anti.push_back(r);
    }
  }
  if (n_scored) *n_scored = kept.size();
  c->cnt.n_scored = (int64_t)kept.size();
  if (c->min_inliers > 0) {  // grasp_detector.cpp:228-236
    std::vector<ag2o_hypothesis> clustered(anti.size());
    size_t nc = 0;
    ag2o_find_clusters(anti.data(), anti.size(), c->min_inliers, 0, clustered.data(), clustered.size(), &nc);
    clustered.resize(nc);
    anti.swap(clustered);
  }
  // :239-252 top num_selected by score, descending; ties broken by position (stable)
  std::stable_sort(anti.begin(), anti.end(),
                   [](const ag2o_hypothesis& a, const ag2o_hypothesis& b) { return a.score > b.score; });
  size_t k = anti.size();
  if (c->p.num_selected >= 0 && k > (size_t)c->p.num_selected) k = (size_t)c->p.num_selected;
  *n_selected = k;
  c->cnt.n_selected = (int64_t)k;
  c->cnt.t_images = t2 - t1;
  c->cnt.t_lenet = t3 - t2;
  c->cnt.t_total = omp_get_wtime() - t0;
  if (k > cap) return fail(c, "detect: output capacity too small");
  for (size_t i = 0; i < k; i++) selected[i] = anti[i];
  return 0;
}

int ag2o_get_counters(ag2o_ctx* c, ag2o_counters* out) {
  if (!c) return -1;
  *out = c->cnt;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Grasp clustering: HandleSearch::findClusters, handle_search.cpp:4-80.
// Eigen's fixed-size dot / norm / 3x3 product reduce as ((x0 + x1) + x2).
// ------------------------------------------------------------------------------------------------
int ag2o_find_clusters(const ag2o_hypothesis* hands, size_t n, int min_inliers, int remove_inliers,
                       ag2o_hypothesis* out, size_t cap, size_t* n_out) {
  if (!n_out || min_inliers < 1) return -1;
  const double cos_thresh = std::cos(15.0 * M_PI / 180.0);  // :7
  std::vector<char> has_used(n, 0);                          // :14-22
  size_t k = 0;
  for (size_t i = 0; i < n; i++) {
    const double* a = hands[i].axis;
    const double* b = hands[i].bottom;
    double P[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) P[r][c] = ((r == c) ? 1.0 : 0.0) - a[r] * a[c];  // :27,47
    int cnt = 0;
    double sum[3] = {0.0, 0.0, 0.0}, ssum = 0.0;
    for (size_t j = 0; j < n; j++) {
      if (i == j || (remove_inliers && has_used[j])) continue;                       // :31
      const double* t = hands[j].axis;
      const double* u = hands[j].bottom;
      const double aligned = (a[0] * t[0] + a[1] * t[1]) + a[2] * t[2];              // :35
      const double d0 = b[0] - u[0], d1 = b[1] - u[1], d2 = b[2] - u[2];             // :39
      const double mag = std::sqrt((d0 * d0 + d1 * d1) + d2 * d2);                   // :40
      const double p0 = (P[0][0] * d0 + P[0][1] * d1) + P[0][2] * d2;                // :48
      const double p1 = (P[1][0] * d0 + P[1][1] * d1) + P[1][2] * d2;
      const double p2 = (P[2][0] * d0 + P[2][1] * d1) + P[2][2] * d2;
      const double pm = std::sqrt((p0 * p0 + p1 * p1) + p2 * p2);                    // :49
      if (std::fabs(aligned) > cos_thresh && mag <= 0.05 && pm <= 0.005) {           // :36,41,50,52
        cnt++;
        sum[0] = sum[0] + u[0]; sum[1] = sum[1] + u[1]; sum[2] = sum[2] + u[2];      // :56
        ssum = ssum + hands[j].score;                                                // :57
        if (remove_inliers) has_used[j] = 1;                                         // :58-59
      }
    }
    if (cnt >= min_inliers) {                                                        // :64
      ag2o_hypothesis h = hands[i];
      const double nn = (double)cnt;
      for (int c = 0; c < 3; c++) {
        const double delta = sum[c] / nn - b[c];                                     // :66
        h.surface[c] = h.surface[c] + delta;                                         // :71-73
        h.bottom[c] = h.bottom[c] + delta;
        h.top[c] = h.top[c] + delta;
      }
      h.score = ssum / nn;                                                           // :67,74
      if (k < cap && out) out[k] = h;
      k++;
    }
  }
  *n_out = k;
  return (k > cap) ? -2 : 0;
}

int ag2o_set_grid_origin(ag2o_ctx* c, const float* origin3) {
  if (!c) return -1;
  c->has_origin = origin3 != nullptr;
  if (origin3)
    for (int a = 0; a < 3; a++) c->origin[a] = origin3[a];
  return 0;
}

int ag2o_set_min_inliers(ag2o_ctx* c, int min_inliers) {
  if (!c || min_inliers < 0) return -1;
  c->min_inliers = min_inliers;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Preprocessing: GraspDetector::preprocessPointCloud steps 1-3,
// grasp_detector.cpp:285-335.
// ------------------------------------------------------------------------------------------------
int ag2o_preprocess_cloud(ag2o_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                          const int32_t* cam_source, int n_cams, const double* normals,
                          int filter_workspace, int voxelize, double voxel_size, int flags,
                          size_t* n_out) {
  if (!c) return -1;
  if (n_cams < 1 || n_cams > 2) return fail(c, "n_cams must be 1 or 2");
  if (stride_bytes < 12 || stride_bytes % 4 != 0) return fail(c, "bad stride");
  if (voxelize && normals) return fail(c, "normals do not survive voxelisation");
  if (voxelize && !((float)voxel_size > 0.f)) return fail(c, "voxel_size must be positive");
  const char* base = (const char*)xyz;
  const double* ws = c->p.workspace;
  // 1. CloudCamera::filterWorkspace, cloud_camera.cpp:89-121: strict bounds, float coordinate
  //    widened to double for the comparison, order kept.  Non-finite points fail every comparison;
  //    they are dropped when the filter is off as well (no such path in the reference).  The
  //    2-camera source matrix is copied column by column (the linear index at :107 is a bug).
  std::vector<size_t> keep;
  keep.reserve(n);
  for (size_t i = 0; i < n; i++) {
    const float* p = (const float*)(base + i * stride_bytes);
    if (!finite3f(p[0], p[1], p[2])) continue;
    if (filter_workspace &&
        !((double)p[0] > ws[0] && (double)p[0] < ws[1] && (double)p[1] > ws[2] &&
          (double)p[1] < ws[3] && (double)p[2] > ws[4] && (double)p[2] < ws[5]))
      continue;
    keep.push_back(i);
  }
  const size_t m = keep.size();
  std::vector<float> px(3 * m);
  std::vector<int32_t> cam(m * (size_t)n_cams, 1);
  std::vector<double> nrm;
  for (size_t k = 0; k < m; k++) {
    const float* p = (const float*)(base + keep[k] * stride_bytes);
    px[3 * k] = p[0]; px[3 * k + 1] = p[1]; px[3 * k + 2] = p[2];
    if (cam_source)
      for (int j = 0; j < n_cams; j++) cam[k * n_cams + j] = cam_source[keep[k] * n_cams + j];
  }
  if (normals) {
    nrm.resize(3 * m);
    for (size_t k = 0; k < m; k++)
      for (int a = 0; a < 3; a++) nrm[3 * k + a] = normals[3 * keep[k] + a];
  }
  if (!voxelize || m == 0) {
    if (n_out) *n_out = m;
    return ag2o_set_cloud(c, px.data(), m, 12, cam.data(), n_cams, normals ? nrm.data() : nullptr);
  }
  // 2. CloudCamera::voxelizeCloud, cloud_camera.cpp:124-168.  All arithmetic in float: Eigen
  //    converts the double cell_size to the vector's scalar type (:139, :155-157).
  float mn[3] = {px[0], px[1], px[2]};
  for (size_t k = 1; k < m; k++)
    for (int a = 0; a < 3; a++) mn[a] = std::min(mn[a], px[3 * k + a]);  // :127-128
  const float cell = (float)voxel_size;
  struct Bin { int v[3]; size_t first; };
  std::vector<Bin> bins(m);
  for (size_t k = 0; k < m; k++) {
    for (int a = 0; a < 3; a++) bins[k].v[a] = (int)std::floor((px[3 * k + a] - mn[a]) / cell);  // :139
    bins[k].first = k;
  }
  // std::set with UniqueVectorComparator (cloud_camera.h:164-175) == ascending (ix, iy, iz); the
  // index of the point whose insert succeeded is the smallest index in the voxel
  std::sort(bins.begin(), bins.end(), [](const Bin& a, const Bin& b) {
    for (int k = 0; k < 3; k++)
      if (a.v[k] != b.v[k]) return a.v[k] < b.v[k];
    return a.first < b.first;
  });
  size_t u = 0;
  for (size_t k = 0; k < m; k++)
    if (k == 0 || bins[k].v[0] != bins[u - 1].v[0] || bins[k].v[1] != bins[u - 1].v[1] ||
        bins[k].v[2] != bins[u - 1].v[2])
      bins[u++] = bins[k];
  bins.resize(u);
  // :137-141 pushes the first-hit indices in SCAN order, :149-152 reads entry i for the i-th voxel
  // in SET order -- replicated literally (flags bit 0 selects "the first point that hit the voxel")
  std::vector<size_t> scan_order(u);
  for (size_t v = 0; v < u; v++) scan_order[v] = bins[v].first;
  std::sort(scan_order.begin(), scan_order.end());
  std::vector<float> vx(3 * u);
  std::vector<int32_t> vcam(u * (size_t)n_cams);
  for (size_t v = 0; v < u; v++) {
    for (int a = 0; a < 3; a++) vx[3 * v + a] = (float)bins[v].v[a] * cell + mn[a];  // :155-157
    const size_t src = (flags & 1) ? bins[v].first : scan_order[v];
    for (int j = 0; j < n_cams; j++) vcam[v * n_cams + j] = (cam[src * n_cams + j] == 1) ? 1 : 0;
  }
  if (n_out) *n_out = u;
  return ag2o_set_cloud(c, vx.data(), u, 12, vcam.data(), n_cams, nullptr);
}

int ag2o_get_cloud(ag2o_ctx* c, float* xyz_nx3, int32_t* cam_source, size_t cap, size_t* n) {
  if (!c || !n) return -1;
  *n = c->n;
  if (cap < c->n) return fail(c, "get_cloud: capacity too small");
  for (size_t i = 0; i < c->n; i++) {
    if (xyz_nx3) { xyz_nx3[3 * i] = c->x[i]; xyz_nx3[3 * i + 1] = c->y[i]; xyz_nx3[3 * i + 2] = c->z[i]; }
  }
  // as the 0/1 "seen by camera" mask every consumer reduces it to (== 1: hand_search.cpp:139,
  // cloud_camera.cpp:151)
  if (cam_source)
    for (size_t i = 0; i < c->cam.size(); i++) cam_source[i] = (c->cam[i] == 1) ? 1 : 0;
  return 0;
}

// 3. CloudCamera::subsampleUniformly, cloud_camera.cpp:171-178 / grasp_detector.cpp:321-335:
// min(num_samples, n) distinct indices, ascending (pcl::RandomSample's output order).  The
// reference seeds from wall time, so only the distribution can be kept: every point gets the key
// (draw_u64(seed, kSubsampleStream, i), i) and the num_samples smallest keys win -- a uniform
// draw without replacement that needs no sequential state.
int ag2o_subsample_uniformly(ag2o_ctx* c, size_t num_samples, uint64_t seed, int32_t* idx_out,
                             size_t cap, size_t* n_out) {
  if (!c || !n_out) return -1;
  const size_t n = c->n, k = std::min(num_samples, n);
  *n_out = k;
  if (cap < k) return fail(c, "subsample: capacity too small");
  if (k == n) {
    for (size_t i = 0; i < n; i++) idx_out[i] = (int32_t)i;
    return 0;
  }
  std::vector<std::pair<uint64_t, uint32_t>> keys(n);
  for (size_t i = 0; i < n; i++) keys[i] = {draw_u64(seed, 0xFFFFFFFFFFFFFFF0ull, (uint64_t)i), (uint32_t)i};
  std::nth_element(keys.begin(), keys.begin() + (long)k, keys.end());
  for (size_t i = 0; i < k; i++) idx_out[i] = (int32_t)keys[i].second;
  std::sort(idx_out, idx_out + k);
  return 0;
}

}  // extern "C"
