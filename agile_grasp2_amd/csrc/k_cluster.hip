// k_cluster.hip -- grasp clustering, the step right behind the scoring.
//
// Replaces HandleSearch::findClusters(hand_list, remove_inliers = false)
// (src/agile_grasp2/handle_search.cpp:4-80; called at grasp_detector.cpp:228-236 and
// importance_sampling.cpp:107): for every hand i, the hands j != i whose axis is within 15 degrees,
// whose bottom is within 5 cm, and within 5 mm once projected onto the plane orthogonal to i's axis,
// are its inliers; a hand with >= min_inliers of them is kept, moved by (mean inlier bottom - own
// bottom) and given the mean inlier score.
//
// One thread per hand i walks ALL hands j in ascending order -- the reference's own summation order,
// so the f64 sums are bit-identical -- with the (axis, bottom, score) of 256 hands at a time staged
// in LDS and read as wave-uniform broadcasts.  O(H^2) pair tests, ~45 f64 operations each: compute
// bound on the f64 vector pipes for large H, launch-bound for the few hundred hands of a tabletop
// scene.  remove_inliers = true is order-dependent across i (has_used, :31-32,58-59) and stays on
// the host (agile_grasp2_amd/host, HandleSearch::findClusters); no caller in the reference uses it.
#include <math.h>

#include <algorithm>

#include "ag2_internal.h"

namespace ag2 {

constexpr int kCluThreads = 256;

__global__ void __launch_bounds__(kCluThreads) k_cluster(const ag2_hypothesis* __restrict__ hands,
                                                         const unsigned* __restrict__ n_ptr, int n_max,
                                                         int min_inliers, double cos_thresh,
                                                         ag2_hypothesis* __restrict__ moved,
                                                         unsigned* __restrict__ flags) {
  __shared__ double tile[kCluThreads][7];  // axis xyz, bottom xyz, score
  const int n = min((int)*n_ptr, n_max);
  const int i = blockIdx.x * kCluThreads + threadIdx.x;
  if (blockIdx.x * kCluThreads >= n) {  // whole workgroup beyond the list (uniform)
    if (i <= n_max) flags[i] = 0u;
    return;
  }
  const bool live = i < n;
  double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
  if (live) {
    for (int k = 0; k < 3; k++) {
      a[k] = hands[i].axis[k];
      b[k] = hands[i].bottom[k];
    }
  }
  // handle_search.cpp:27 axis_outer_prod, :47 axis_orth_proj = I - a a^T
  double P[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) P[r][cc] = ((r == cc) ? 1.0 : 0.0) - a[r] * a[cc];
  int cnt = 0;
  double sum[3] = {0.0, 0.0, 0.0}, ssum = 0.0;
  for (int j0 = 0; j0 < n; j0 += kCluThreads) {
    __syncthreads();
    const int jl = j0 + threadIdx.x;
    if (jl < n) {
      const ag2_hypothesis& h = hands[jl];
      tile[threadIdx.x][0] = h.axis[0];
      tile[threadIdx.x][1] = h.axis[1];
      tile[threadIdx.x][2] = h.axis[2];
      tile[threadIdx.x][3] = h.bottom[0];
      tile[threadIdx.x][4] = h.bottom[1];
      tile[threadIdx.x][5] = h.bottom[2];
      tile[threadIdx.x][6] = h.score;
    }
    __syncthreads();
    const int m = min(kCluThreads, n - j0);
    for (int jj = 0; jj < m; jj++) {
      if (j0 + jj == i) continue;  // :31
      const double* t = tile[jj];
      const double aligned = (a[0] * t[0] + a[1] * t[1]) + a[2] * t[2];            // :35
      const double d0 = b[0] - t[3], d1 = b[1] - t[4], d2 = b[2] - t[5];           // :39
      const double mag = __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);            // :40
      const double p0 = (P[0][0] * d0 + P[0][1] * d1) + P[0][2] * d2;              // :48
      const double p1 = (P[1][0] * d0 + P[1][1] * d1) + P[1][2] * d2;
      const double p2 = (P[2][0] * d0 + P[2][1] * d1) + P[2][2] * d2;
      const double pm = __builtin_sqrt((p0 * p0 + p1 * p1) + p2 * p2);             // :49
      if (__builtin_fabs(aligned) > cos_thresh && mag <= 0.05 && pm <= 0.005) {    // :36,41,50,52
        cnt++;                                                                      // :55-57
        sum[0] = sum[0] + t[3];
        sum[1] = sum[1] + t[4];
        sum[2] = sum[2] + t[5];
        ssum = ssum + t[6];
      }
    }
  }
  const bool keep = live && cnt >= min_inliers;  // :64
  if (i <= n_max) flags[i] = keep ? 1u : 0u;
  if (keep) {
    ag2_hypothesis h = hands[i];
    const double nn = (double)cnt;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double delta = sum[k] / nn - b[k];  // :66
      h.surface[k] = h.surface[k] + delta;      // :71-73
      h.bottom[k] = h.bottom[k] + delta;
      h.top[k] = h.top[k] + delta;
    }
    h.score = ssum / nn;                        // :67,74
    moved[i] = h;
  }
}

// order-preserving compaction of the kept hands; pref = exclusive prefix of the flags
__global__ void k_cluster_scatter(const ag2_hypothesis* __restrict__ moved,
                                  const unsigned* __restrict__ pref, int n_max,
                                  ag2_hypothesis* __restrict__ out, unsigned* __restrict__ n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    *n_out = pref[n_max];
    *reinterpret_cast<unsigned*>(out + n_max) = pref[n_max];  // trailer: one copy brings records + count
  }
  if (i >= n_max) return;
  const unsigned p = pref[i];
  if (pref[i + 1] != p) out[p] = moved[i];
}

// d_in: up to n_max records, *d_n of them valid.  Result: d_cluster (records), *d_count.
int cluster_async(ag2_ctx* c, const ag2_hypothesis* d_in, size_t n_max, const unsigned* d_n,
                  int min_inliers, unsigned* d_count) {
  AG2_HIP(c, c->d_cluster.reserve(std::max<size_t>(n_max, 1) * sizeof(ag2_hypothesis) + 16));
  AG2_HIP(c, c->d_cluster_tmp.reserve(std::max<size_t>(n_max, 1) * sizeof(ag2_hypothesis)));
  AG2_HIP(c, c->d_flags.reserve((n_max + 1) * 4));
  if (n_max == 0) {
    AG2_HIP(c, hipMemsetAsync(d_count, 0, 4, c->stream));
    return 0;
  }
  const double cos_thresh = cos(15.0 * M_PI / 180.0);  // AXIS_ALIGN_ANGLE_THRESH, handle_search.cpp:7
  const int ni = (int)n_max;
  unsigned* flags = c->d_flags.as<unsigned>();
  hipLaunchKernelGGL(k_cluster, dim3((ni + kCluThreads) / kCluThreads), dim3(kCluThreads), 0, c->stream,
                     d_in, d_n, ni, min_inliers, cos_thresh, c->d_cluster_tmp.as<ag2_hypothesis>(), flags);
  const int rc = scan_exclusive_u32(c, flags, ni + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_cluster_scatter, dim3((ni + 255) / 256), dim3(256), 0, c->stream,
                     c->d_cluster_tmp.as<ag2_hypothesis>(), flags, ni, c->d_cluster.as<ag2_hypothesis>(),
                     d_count);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2

using namespace ag2;

extern "C" {

int ag2_set_min_inliers(ag2_ctx* c, int min_inliers) {
  if (!c) return AG2_ERR_ARG;
  if (min_inliers < 0) return set_err(c, AG2_ERR_ARG, "min_inliers must be >= 0");
  c->min_inliers = min_inliers;
  return 0;
}

int ag2_find_clusters(ag2_ctx* c, const ag2_hypothesis* hands, size_t n, int min_inliers,
                      ag2_hypothesis* out, size_t cap, size_t* n_out) {
  if (!c || !n_out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (min_inliers < 1)
    return set_err(c, AG2_ERR_ARG, "find_clusters: min_inliers must be >= 1 (0 inliers divide by zero, "
                                   "handle_search.cpp:66)");
  if (n > (size_t)1 << 24) return set_err(c, AG2_ERR_CAPACITY, "find_clusters: more than 2^24 hands");
  *n_out = 0;
  if (n == 0) return 0;
  if (!hands) return set_err(c, AG2_ERR_ARG, "find_clusters: hands is NULL");
  AG2_HIP(c, c->d_tmp.reserve(n * sizeof(ag2_hypothesis) + 16));
  unsigned* d_n = (unsigned*)((char*)c->d_tmp.p + n * sizeof(ag2_hypothesis));
  const unsigned nu = (unsigned)n;
  AG2_HIP(c, hipMemcpyAsync(c->d_tmp.p, hands, n * sizeof(ag2_hypothesis), hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, hipMemcpyAsync(d_n, &nu, 4, hipMemcpyHostToDevice, c->stream));
  const int rc = cluster_async(c, c->d_tmp.as<ag2_hypothesis>(), n, d_n, min_inliers, d_n + 1);
  if (rc) return rc;
  unsigned k = 0;
  AG2_HIP(c, hipMemcpyAsync(&k, d_n + 1, 4, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  *n_out = k;
  if (k > cap) return set_err(c, AG2_ERR_CAPACITY, "find_clusters: output capacity too small");
  if (k && !out) return set_err(c, AG2_ERR_ARG, "find_clusters: out is NULL");
  if (k) AG2_HIP(c, hipMemcpy(out, c->d_cluster.p, (size_t)k * sizeof(ag2_hypothesis), hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
