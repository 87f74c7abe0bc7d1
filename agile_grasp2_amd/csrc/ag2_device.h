// ag2_device.h -- device-side helpers shared by the HIP kernels of libag2hip.so (gfx950 only).
//
// Arithmetic contract: everything that feeds a comparison or an index on the geometric path uses
// IEEE-754 + - * / sqrt with the parenthesisation written here; the translation units are compiled
// with -ffp-contract=off so hipcc forms no FMAs.  The same contract is stated (independently) in
// oracle/ag2_oracle.cpp; parity tests require bit-identical labels/indices from the two.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ag2 {

constexpr int kWave = 64;
constexpr int kMaxOrient = 32;
constexpr int kMaxDepths = 32;
constexpr int kMaxRows = 512;    // stencil rows (y,z pairs) of one hand-radius query
constexpr int kImg = 60;         // Learning(60, ...) grasp_detector.cpp:56

struct GridDesc {
  float o[3];
  float inv;
  int dims[3];
  int ncells;
  int n_valid;
  float min_z;   // smallest z of the (whole) cloud: pcl::getMinMax3D, grasp_detector.cpp:152-153
};

// Frame mode (ag2_frame.hip): what changes from frame to frame and would otherwise be a by-value
// kernel argument frozen into the captured graph lives in page-locked host memory the kernels read
// through its device view.
struct FrameArgs {
  unsigned long long seed;
  unsigned long long slot_base;
  unsigned long long sample_seed;  // ag2_detect_frame_raw: seed of the uniform sub-sampling
  unsigned long long seq;          // the frame's sequence number: k_topk writes it behind the results (FrameOut::done_seq)
};

// Per-context constants, built on the host (ag2_context.hip derive_constants) and read through a
// uniform pointer.
struct HandConst {
  double fs[20];                 // finger_spacing_(i), finger_hand.cpp:9-12
  double fsr[20];                // finger_spacing_(i) + finger_width_
  double cos_t[kMaxOrient];      // hand_search.cpp:179-180, :356
  double sin_t[kMaxOrient];
  double depths[kMaxDepths];     // deepenHand depth sequence, finger_hand.cpp:118-122
  double cam_origin[2][3];
  double finger_width, hand_outer_diameter, hand_depth, hand_height, init_bite;
  double cos_fc;                 // cos(30 deg), antipodal.cpp:11,23
  double slot_inv_step;          // 1 / finger-slot spacing (candidate selection only)
  double slot_step;              // (od - fw) / 9: the LinSpaced step, finger_hand.cpp:10
  double min_aperture, max_aperture;
  float ws_min_x, ws_max_x, ws_min_y, ws_max_y;  // float bounds, grasp_detector.cpp:363-364
  float r2_taubin, r2_hands, r2_normals;         // (float)(r*r)
  float rq_taubin, rq_hands, rq_normals;         // conservative per-axis reach
  int R, n_depths, n_cams, filter_half;
  int slot_span, pad0;           // floor(finger_width / spacing) + 1
};

// k_lenet_fc1_x3: split of K (225 chunks of 32) for a batch of `mtiles` 128-image tiles, so that
// small batches still put about two workgroups on every CU.  One rule for the host (exact-size
// launches) and the device (frame mode, batch size known only there): the split decides the order
// in which the partial sums are added, so it must not depend on who chose it.
constexpr int kFc1X3Chunks = 225;
constexpr int kFc1X3MaxSplit = 45;
__host__ __device__ inline int fc1_x3_ksplit(int mtiles) {
  const int splits[7] = {1, 3, 5, 9, 15, 25, 45};
  for (int i = 0; i < 7; i++)
    if ((long long)mtiles * 4 * splits[i] >= 448) return splits[i];
  return kFc1X3MaxSplit;
}

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ double dot3(const V3& a, const V3& b) {
  return (a.x * b.x + a.y * b.y) + a.z * b.z;
}
__device__ __forceinline__ V3 cross3(const V3& a, const V3& b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 neg3(const V3& a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ double norm3(const V3& a) { return __builtin_sqrt(dot3(a, a)); }

__device__ __forceinline__ int cell_of(float v, float o, float inv) {
  return (int)__builtin_floorf((v - o) * inv);
}
__device__ __forceinline__ bool finite3(float a, float b, float c) {
  return __builtin_isfinite(a) && __builtin_isfinite(b) && __builtin_isfinite(c);
}

// Cyclic Jacobi for a symmetric 3x3, written on scalars so everything stays in registers.
// Stands in for Eigen::EigenSolver (local_frame.cpp:30-32) and pcl::eigen33 (inside
// pcl::NormalEstimationOMP, hand_search.cpp:85-92).  Rotation order (0,1), (0,2), (1,2); at most 12
// sweeps; a pair with an exactly-zero off-diagonal is skipped.
struct Sym3 {
  double a00, a01, a02, a11, a12, a22;
};
struct Eig3 {
  double d[3];
  double v[3][3];  // columns = eigenvectors
};

// A converged pair (|apq| <= 2^-60 |aqq - app|: the rotation would change neither the diagonal nor, at
// f64 resolution, the eigenvectors) is annihilated without rotating -- same rule as the oracle.
#define AG2_JACOBI_ROT(APP, AQQ, APQ, ARP, ARQ, VP, VQ)                          \
  if (APQ != 0.0 && __builtin_fabs(APQ) <= 0x1p-60 * __builtin_fabs(AQQ - APP)) { \
    APQ = 0.0;                                                                   \
  } else if (APQ != 0.0) {                                                       \
    const double theta = (AQQ - APP) / (2.0 * APQ);                              \
    double t = 1.0 / (__builtin_fabs(theta) + __builtin_sqrt(theta * theta + 1.0)); \
    if (theta < 0.0) t = -t;                                                     \
    const double c = 1.0 / __builtin_sqrt(t * t + 1.0);                          \
    const double s = t * c;                                                      \
    APP = APP - t * APQ;                                                         \
    AQQ = AQQ + t * APQ;                                                         \
    APQ = 0.0;                                                                   \
    const double arp = ARP, arq = ARQ;                                           \
    ARP = c * arp - s * arq;                                                     \
    ARQ = s * arp + c * arq;                                                     \
    _Pragma("unroll") for (int k = 0; k < 3; k++) {                              \
      const double vkp = e.v[k][VP], vkq = e.v[k][VQ];                           \
      e.v[k][VP] = c * vkp - s * vkq;                                            \
      e.v[k][VQ] = s * vkp + c * vkq;                                            \
    }                                                                            \
  }

__device__ inline Eig3 jacobi3(Sym3 m) {
  Eig3 e;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) e.v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; sweep++) {
    if (m.a01 == 0.0 && m.a02 == 0.0 && m.a12 == 0.0) break;
    // (p,q,r) = (0,1,2): a[r][p] = a02, a[r][q] = a12
    AG2_JACOBI_ROT(m.a00, m.a11, m.a01, m.a02, m.a12, 0, 1)
    // (p,q,r) = (0,2,1): a[r][p] = a01, a[r][q] = a12
    AG2_JACOBI_ROT(m.a00, m.a22, m.a02, m.a01, m.a12, 0, 2)
    // (p,q,r) = (1,2,0): a[r][p] = a01, a[r][q] = a02
    AG2_JACOBI_ROT(m.a11, m.a22, m.a12, m.a01, m.a02, 1, 2)
  }
  e.d[0] = m.a00;
  e.d[1] = m.a11;
  e.d[2] = m.a22;
  return e;
}

__device__ __forceinline__ int argmin3(const double d[3]) {
  int m = 0;
  if (d[1] < d[m]) m = 1;
  if (d[2] < d[m]) m = 2;
  return m;
}

// Counter-based draw replacing rand() % size (hand_search.cpp:130), keyed (seed, slot, draw).
__device__ __forceinline__ uint64_t draw_u64(uint64_t seed, uint64_t slot, uint64_t j) {
  uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (slot + 1ull));
  x += 0xD1B54A32D192ED03ull * (j + 1ull);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// ---- wave / block primitives (64-wide wavefronts) --------------------------------------------
// ---- rank of record i among n records by (score descending, position ascending) ----------------------
// grasp_detector.cpp:239-252 (partial_sort by score; ties by list position, as ag2_detect's host sort
// orders them).  Scores are float differences widened to double, so a record's place in the order is
// one 64-bit key -- the float's bits made monotone, then the complement of the position -- and the rank
// is the number of larger keys: every key staged ONCE in LDS, two instructions per compare.  A list
// longer than kRankKeys, or a score that is not exactly a float, is left to the caller's general loop
// (returns -1; uniform over the workgroup).  `keys` = kRankKeys x 8 bytes of LDS, `flag` one LDS word.
constexpr int kRankKeys = 4096;
__device__ __forceinline__ int rank_by_keys(const ag2_hypothesis* __restrict__ recs, int n, int i,
                                            unsigned long long* keys, int* flag) {
  if (n > kRankKeys) return -1;
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
  if (tid == 0) *flag = 0;
  __syncthreads();
  const int npad = (n + 7) & ~7;
  bool inexact = false;
  for (int j = tid; j < npad; j += nt) {
    unsigned long long k = 0ull;  // padding: never larger than a real key
    if (j < n) {
      const double sd = recs[j].score;
      const float f = (float)sd + 0.0f;  // (-0.0 and +0.0 compare equal as doubles: one key for both)
      inexact = inexact || !((double)f == sd);
      const unsigned b = __float_as_uint(f);
      const unsigned ok = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
      k = ((unsigned long long)ok << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)j);
    }
    keys[j] = k;
  }
  if (inexact) *flag = 1;
  __syncthreads();
  if (*flag) return -1;
  int rank = 0;
  if (i < n) {
    const unsigned long long ki = keys[i];
    for (int j0 = 0; j0 < npad; j0 += 8) {
      unsigned long long v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = keys[j0 + u];
#pragma unroll
      for (int u = 0; u < 8; u++) rank += (v[u] > ki) ? 1 : 0;
    }
  }
  return rank;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// Wave-wide reductions on the DPP cross-lane path (no LDS round trips, unlike __shfl's
// ds_bpermute): quad swaps, row half-mirror / mirror, then row_bcast:15 / row_bcast:31 leave the
// full result in lane 63, which is broadcast with v_readlane.  A lane whose DPP source is masked off
// keeps `old` (bound_ctrl = 0), so `old` is the operation's identity for that lane.
#define AG2_DPP(old, v, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rmask), 0xF, false)
#define AG2_DPP_STEPS(OP, ID)                         \
  v = OP(v, AG2_DPP(ID, v, 0xB1, 0xF));  /* quad_perm [1,0,3,2] */ \
  v = OP(v, AG2_DPP(ID, v, 0x4E, 0xF));  /* quad_perm [2,3,0,1] */ \
  v = OP(v, AG2_DPP(ID, v, 0x141, 0xF)); /* row_half_mirror */     \
  v = OP(v, AG2_DPP(ID, v, 0x140, 0xF)); /* row_mirror */          \
  v = OP(v, AG2_DPP(ID, v, 0x142, 0xA)); /* row_bcast:15 */        \
  v = OP(v, AG2_DPP(ID, v, 0x143, 0xC)); /* row_bcast:31 */

__device__ __forceinline__ int ag2_op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int ag2_op_or(int a, int b) { return a | b; }
__device__ __forceinline__ int ag2_op_min(int a, int b) { return a < b ? a : b; }

__device__ __forceinline__ int wave_sum_i(int v) {
  AG2_DPP_STEPS(ag2_op_add, 0)
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_or_u(unsigned u) {
  int v = (int)u;
  AG2_DPP_STEPS(ag2_op_or, 0)
  return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v) {
  AG2_DPP_STEPS(ag2_op_min, 0x7fffffff)
  return __builtin_amdgcn_readlane(v, 63);
}
// f64 min / max: both dwords travel together through the same DPP pattern
__device__ __forceinline__ double ag2_dpp_d(double idv, double v, const int which, const int rmask) {
  const long long iv = __double_as_longlong(v), ii = __double_as_longlong(idv);
  int lo = (int)iv, hi = (int)(iv >> 32);
  const int ilo = (int)ii, ihi = (int)(ii >> 32);
  switch (which) {
    case 0: lo = AG2_DPP(ilo, lo, 0xB1, 0xF); hi = AG2_DPP(ihi, hi, 0xB1, 0xF); break;
    case 1: lo = AG2_DPP(ilo, lo, 0x4E, 0xF); hi = AG2_DPP(ihi, hi, 0x4E, 0xF); break;
    case 2: lo = AG2_DPP(ilo, lo, 0x141, 0xF); hi = AG2_DPP(ihi, hi, 0x141, 0xF); break;
    case 3: lo = AG2_DPP(ilo, lo, 0x140, 0xF); hi = AG2_DPP(ihi, hi, 0x140, 0xF); break;
    case 4: lo = AG2_DPP(ilo, lo, 0x142, 0xA); hi = AG2_DPP(ihi, hi, 0x142, 0xA); break;
    default: lo = AG2_DPP(ilo, lo, 0x143, 0xC); hi = AG2_DPP(ihi, hi, 0x143, 0xC); break;
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double ag2_bcast63_d(double v) {
  const long long iv = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)iv, 63), hi = __builtin_amdgcn_readlane((int)(iv >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_min_d(double v) {
  const double id = __builtin_inf();
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const double w = ag2_dpp_d(id, v, s, 0);
    v = (w < v) ? w : v;
  }
  return ag2_bcast63_d(v);
}
__device__ __forceinline__ double wave_max_d(double v) {
  const double id = -__builtin_inf();
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const double w = ag2_dpp_d(id, v, s, 0);
    v = (w > v) ? w : v;
  }
  return ag2_bcast63_d(v);
}

}  // namespace ag2
