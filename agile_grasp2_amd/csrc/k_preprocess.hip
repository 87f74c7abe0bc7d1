// k_preprocess.hip -- the step in front of the hot path, on the GPU.
//
// Replaces GraspDetector::preprocessPointCloud steps 1-3 (src/agile_grasp2/grasp_detector.cpp:285-335):
//   1. CloudCamera::filterWorkspace   (cloud_camera.cpp:89-121)  flag -> scan -> ordered compaction
//   2. CloudCamera::voxelizeCloud     (cloud_camera.cpp:124-168) occupancy bitmap in (ix, iy, iz) key
//      order instead of a std::set: marking is one atomicOr per point, the sorted unique voxel list
//      falls out of a popcount scan over the bitmap words -- no sort, no hash table
//   3. CloudCamera::subsampleUniformly (cloud_camera.cpp:171-178) the num_samples smallest of the
//      per-point keys (draw_u64(seed, stream, i), i), found by a 6-digit radix select that never
//      materialises the keys
// The result is written straight into the context's cloud buffer (as ag2_set_cloud would) and the
// sample indices stay on the device for ag2_detect / ag2_generate_hypotheses.
//
// All of it is HBM-bound integer/byte work: per point 16 B read + 4 B flag, 16 B write for the
// survivors; per voxel-grid word 4 B read twice + 4 B rank.
#include <math.h>

#include <algorithm>

#include "ag2_internal.h"
#include "k_grid_common.h"

namespace ag2 {

constexpr uint64_t kSubsampleStream = 0xFFFFFFFFFFFFFFF0ull;  // "slot" of the sub-sampling draws

struct WsBox {
  double b[6];
};
struct VoxDesc {
  float mn[3];
  float cell;
  int dims[3];
  int pad;
};

__global__ void k_pre_init(PreStats* ps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    PreStats z{};
    z.mn[0] = z.mn[1] = z.mn[2] = 0x7fffffff;
    z.mx[0] = z.mx[1] = z.mx[2] = (int)0x80000000;
    *ps = z;
  }
}

// cloud_camera.cpp:94-95: strict bounds, the float coordinate widened to double.  Non-finite
// points fail every comparison; they are dropped when the filter is off as well.
__global__ void __launch_bounds__(256) k_pre_flag_bounds(const float4* __restrict__ raw, int n,
                                                         WsBox ws, int do_filter,
                                                         unsigned* __restrict__ flags,
                                                         PreStats* __restrict__ ps) {
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = raw[i];
    bool keep = finite3(p.x, p.y, p.z);
    if (keep && do_filter)
      keep = (double)p.x > ws.b[0] && (double)p.x < ws.b[1] && (double)p.y > ws.b[2] &&
             (double)p.y < ws.b[3] && (double)p.z > ws.b[4] && (double)p.z < ws.b[5];
    flags[i] = keep ? 1u : 0u;
    if (keep) {
      cnt++;
      const int a[3] = {f2ord(p.x), f2ord(p.y), f2ord(p.z)};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        mn[k] = min(mn[k], a[k]);
        mx[k] = max(mx[k], a[k]);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) flags[n] = 0u;  // the scan leaves the total here
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[k] = min(mn[k], __shfl_xor(mn[k], o, 64));
      mx[k] = max(mx[k], __shfl_xor(mx[k], o, 64));
    }
  }
  cnt = wave_sum_i(cnt);
  __shared__ int part[4][7];
  if (lane_id() == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      part[wave_id()][k] = mn[k];
      part[wave_id()][3 + k] = mx[k];
    }
    part[wave_id()][6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int k = threadIdx.x;
    int v = part[0][k];
    for (int w = 1; w < 4; w++) {
      const int o = part[w][k];
      v = (k < 3) ? min(v, o) : (k < 6 ? max(v, o) : v + o);
    }
    if (k < 3) atomicMin(&ps->mn[k], v);
    else if (k < 6) atomicMax(&ps->mx[k - 3], v);
    else atomicAdd(&ps->n_keep, (unsigned)v);
  }
}

// order-preserving compaction (cloud_camera.cpp:101-118); pref = exclusive prefix of the flags
__global__ void k_pre_compact(const float4* __restrict__ raw, const float4* __restrict__ nrm,
                              const unsigned* __restrict__ pref, int n, float4* __restrict__ dst,
                              float4* __restrict__ nrm_dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned a = pref[i];
  if (pref[i + 1] != a) {
    dst[a] = raw[i];
    if (nrm) nrm_dst[a] = nrm[i];
  }
}

// cloud_camera.cpp:139 + floorVector :212-218, in float like Eigen evaluates it
__device__ __forceinline__ long long vox_key(const float4& p, const VoxDesc& v) {
  const int ix = (int)__builtin_floorf((p.x - v.mn[0]) / v.cell);
  const int iy = (int)__builtin_floorf((p.y - v.mn[1]) / v.cell);
  const int iz = (int)__builtin_floorf((p.z - v.mn[2]) / v.cell);
  return ((long long)ix * v.dims[1] + iy) * v.dims[2] + iz;  // ascending == the set's comparator
}

__global__ void k_vox_mark(const float4* __restrict__ pts, int m, VoxDesc v,
                           unsigned* __restrict__ bitmap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const long long key = vox_key(pts[i], v);
  atomicOr(&bitmap[key >> 5], 1u << (key & 31));
}

__global__ void k_vox_popc(const unsigned* __restrict__ bitmap, int words,
                           unsigned* __restrict__ wrank) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w > words) return;
  wrank[w] = (w < words) ? (unsigned)__popc(bitmap[w]) : 0u;
}

__device__ __forceinline__ unsigned vox_rank(long long key, const unsigned* __restrict__ bitmap,
                                             const unsigned* __restrict__ wrank) {
  const long long w = key >> 5;
  const unsigned below = bitmap[w] & ((1u << (key & 31)) - 1u);
  return wrank[w] + (unsigned)__popc(below);
}

// smallest point index per voxel = the point whose std::set insert succeeded (:140-141)
__global__ void k_vox_first(const float4* __restrict__ pts, int m, VoxDesc v,
                            const unsigned* __restrict__ bitmap,
                            const unsigned* __restrict__ wrank, int* __restrict__ first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  atomicMin(&first[vox_rank(vox_key(pts[i], v), bitmap, wrank)], i);
}

// voxel value = index * cell + min (:155-157), two float roundings; camera mask filled in later
__global__ void k_vox_emit(const unsigned* __restrict__ bitmap, const unsigned* __restrict__ wrank,
                           int words, VoxDesc v, float4* __restrict__ out) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= words) return;
  unsigned bits = bitmap[w];
  unsigned r = wrank[w];
  const long long plane = (long long)v.dims[1] * v.dims[2];
  while (bits) {
    const int b = __ffs((int)bits) - 1;
    bits &= bits - 1u;
    const long long key = ((long long)w << 5) + b;
    const int ix = (int)(key / plane);
    const long long rem = key - (long long)ix * plane;
    const int iy = (int)(rem / v.dims[2]);
    const int iz = (int)(rem - (long long)iy * v.dims[2]);
    out[r++] = make_float4((float)ix * v.cell + v.mn[0], (float)iy * v.cell + v.mn[1],
                           (float)iz * v.cell + v.mn[2], __int_as_float(1));
  }
}

__global__ void k_vox_firsthit_flags(const float4* __restrict__ pts, int m, VoxDesc v,
                                     const unsigned* __restrict__ bitmap,
                                     const unsigned* __restrict__ wrank,
                                     const int* __restrict__ first, unsigned* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > m) return;
  flags[i] = (i < m && first[vox_rank(vox_key(pts[i], v), bitmap, wrank)] == i) ? 1u : 0u;
}

// Literal reference indexing: idx_cam_source is filled in SCAN order (:140-141) but read with the
// voxel's position in SET order (:149-152), so voxel k takes the mask of the k-th first-hit point.
__global__ void k_vox_cam_literal(const float4* __restrict__ pts, int m,
                                  const unsigned* __restrict__ pref, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const unsigned a = pref[i];
  if (pref[i + 1] != a) out[a].w = pts[i].w;
}
// flags bit 0: the mask of the first point that hit the voxel
__global__ void k_vox_cam_owner(const float4* __restrict__ pts, const int* __restrict__ first,
                                int n_vox_max, const unsigned* __restrict__ n_vox,
                                float4* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_vox_max || k >= (int)*n_vox) return;
  out[k].w = pts[first[k]].w;
}

// ---- sub-sampling: radix select of the num_samples-th smallest (hash hi, hash lo, index) ---------
__device__ __forceinline__ void sel_words(uint64_t seed, int i, unsigned W[3]) {
  const uint64_t h = draw_u64(seed, kSubsampleStream, (uint64_t)i);
  W[0] = (unsigned)(h >> 32);
  W[1] = (unsigned)h;
  W[2] = (unsigned)i;
}

__global__ void k_sel_init(PreStats* ps, unsigned k) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    ps->sel_prefix[0] = ps->sel_prefix[1] = ps->sel_prefix[2] = 0u;
    ps->sel_remaining = k;
  }
}

__global__ void k_sel_hist(int n, uint64_t seed, int digit, const PreStats* __restrict__ ps,
                           unsigned* __restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned W[3];
  sel_words(seed, i, W);
  const int w = digit >> 1;
  bool match = true;
  for (int k = 0; k < w; k++) match = match && (W[k] == ps->sel_prefix[k]);
  if (digit & 1) match = match && ((W[w] >> 16) == (ps->sel_prefix[w] >> 16));
  if (match) atomicAdd(&hist[(W[w] >> ((digit & 1) ? 0 : 16)) & 0xFFFFu], 1u);
}

// one workgroup: find the bin where the running count reaches sel_remaining, extend the prefix,
// clear the histogram for the next digit
__global__ void __launch_bounds__(1024) k_sel_pick(unsigned* __restrict__ hist, int digit,
                                                   PreStats* __restrict__ ps) {
  __shared__ unsigned wsum[16];
  const int t = threadIdx.x;
  unsigned loc[64];
  unsigned tot = 0;
#pragma unroll
  for (int k = 0; k < 64; k++) {
    loc[k] = hist[t * 64 + k];
    tot += loc[k];
    hist[t * 64 + k] = 0u;
  }
  unsigned inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = (unsigned)__shfl_up((int)inc, o, 64);
    if (lane_id() >= o) inc += v;
  }
  if (lane_id() == 63) wsum[wave_id()] = inc;
  __syncthreads();
  unsigned woff = 0;
  for (int w = 0; w < wave_id(); w++) woff += wsum[w];
  const unsigned before = woff + inc - tot;  // keys in bins below mine
  const unsigned rem = ps->sel_remaining;
  __syncthreads();                            // everyone has read sel_remaining
  if (rem > before && rem <= before + tot) {
    unsigned run = before;
    for (int k = 0; k < 64; k++) {
      if (rem <= run + loc[k]) {
        const unsigned bin = (unsigned)(t * 64 + k);
        ps->sel_prefix[digit >> 1] |= bin << ((digit & 1) ? 0 : 16);
        ps->sel_remaining = rem - run;
        break;
      }
      run += loc[k];
    }
  }
}

__global__ void k_sel_flags(int n, uint64_t seed, const PreStats* __restrict__ ps,
                            unsigned* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  unsigned f = 0u;
  if (i < n) {
    unsigned W[3];
    sel_words(seed, i, W);
    const unsigned* T = ps->sel_prefix;
    const bool le = (W[0] != T[0]) ? (W[0] < T[0]) : ((W[1] != T[1]) ? (W[1] < T[1]) : (W[2] <= T[2]));
    f = le ? 1u : 0u;
  }
  flags[i] = f;
}

__global__ void k_sel_scatter(const unsigned* __restrict__ pref, int n, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned a = pref[i];
  if (pref[i + 1] != a) out[a] = i;
}

__global__ void k_iota(int n, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

// ================================================================================================
// The front end WITHOUT a host round trip (single-camera clouds, voxel grid on): what a frame of a
// cloud stream runs inside its captured sequence (ag2_detect_frame_raw), and the fast path of
// ag2_preprocess_cloud_device / ag2_subsample_uniformly.
//   k_raw_filter_bounds   pack + workspace test + extent partials of the points that pass
//   k_vox_mark_frame      voxel lattice derived from the partials by the first wave of every workgroup
//                         (as k_cell_count derives the search grid), then one atomicOr per passing point.
//                         No compaction: with one camera the voxel set does not depend on the order of
//                         the points (cloud_camera.cpp:137-141 inserts into a std::set).
//   k_scan_chained<POPC>  rank of every bitmap word (k_grid.hip)
//   k_vox_emit_frame      voxels in ascending (ix, iy, iz) order + padding, the search grid's description,
//                         and the sub-sampling candidates (hash of the voxel's index under the threshold)
//   k_sel_rank / k_sel_emit   exact selection of the num_samples smallest keys among the candidates,
//                         written in ascending index order
// Every shape is a fixed maximum; what a frame does not fit is flagged in PreFrame and the caller repeats
// the frame on the general path below.
// ================================================================================================
__device__ __forceinline__ bool ws_keep(const float4& p, const WsBox& ws, int do_filter) {
  bool keep = finite3(p.x, p.y, p.z);
  if (keep && do_filter)  // cloud_camera.cpp:94-95
    keep = (double)p.x > ws.b[0] && (double)p.x < ws.b[1] && (double)p.y > ws.b[2] &&
           (double)p.y < ws.b[3] && (double)p.z > ws.b[4] && (double)p.z < ws.b[5];
  return keep;
}

constexpr int kRawBlocks = 512;  // extent partials (one 32-B record per workgroup): two waves per SIMD keep HBM busy

// PACK: read the caller's strided buffer, write float4 (w = camera mask 1), pad [n, n_pad) with
// non-finite points; else the float4 cloud is already in `raw`.  st (optional): cleared by block 0.
template <bool PACK>
__global__ void __launch_bounds__(256) k_raw_filter_bounds(const char* __restrict__ src, size_t stride,
                                                           float4* __restrict__ raw, int n, int n_pad,
                                                           WsBox ws, int do_filter, int* __restrict__ part,
                                                           DevStats* st) {
  if (st && blockIdx.x == 0 && threadIdx.x == 0) *st = DevStats{};
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
    float4 p;
    if (PACK) {
      if (i < n) {
        const float* q = (const float*)(src + (size_t)i * stride);
        p = make_float4(q[0], q[1], q[2], __int_as_float(1));
      } else {
        p = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), 0.f);
      }
      raw[i] = p;
    } else {
      p = raw[i];
    }
    if (ws_keep(p, ws, do_filter)) {
      cnt++;
      const int a[3] = {f2ord(p.x), f2ord(p.y), f2ord(p.z)};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        mn[k] = min(mn[k], a[k]);
        mx[k] = max(mx[k], a[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[k] = min(mn[k], __shfl_xor(mn[k], o, 64));
      mx[k] = max(mx[k], __shfl_xor(mx[k], o, 64));
    }
  }
  cnt = wave_sum_i(cnt);
  __shared__ int wpart[4][7];
  if (lane_id() == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      wpart[wave_id()][k] = mn[k];
      wpart[wave_id()][3 + k] = mx[k];
    }
    wpart[wave_id()][6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int k = threadIdx.x;
    int v = wpart[0][k];
    for (int w = 1; w < 4; w++) {
      const int o = wpart[w][k];
      v = (k < 3) ? min(v, o) : (k < 6 ? max(v, o) : v + o);
    }
    part[blockIdx.x * 8 + k] = v;
  }
}

// The voxel lattice of the points that passed, from the extent partials: the float expressions of
// preprocess_resident below (cloud_camera.cpp:127-139), evaluated by every lane of the first wave.
__device__ __forceinline__ PreFrame vox_from_partials(const int* __restrict__ part, int nb, float cell,
                                                      int cap_words) {
  const int lane = lane_id();
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int cnt = 0;
  for (int b = lane; b < nb; b += 64) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = min(mn[a], part[b * 8 + a]);
      mx[a] = max(mx[a], part[b * 8 + 3 + a]);
    }
    cnt += part[b * 8 + 6];
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
    }
  }
  cnt = wave_sum_i(cnt);
  PreFrame f{};
  f.cell = cell;
  f.n_keep = (unsigned)cnt;
  if (cnt > 0) {
    long long ncells = 1;
    bool bad = false;
    for (int a = 0; a < 3; a++) {
      f.mn[a] = ord2f(mn[a]);
      const float top = __builtin_floorf((ord2f(mx[a]) - f.mn[a]) / cell);  // same expression as vox_key
      if (!(top < 2.0e6f)) bad = true;
      f.dims[a] = bad ? 1 : (int)top + 1;
      ncells *= f.dims[a];
      if (ncells > (1ll << 33)) bad = true;
    }
    const long long words = (ncells + 31) >> 5;
    if (bad || words > (long long)cap_words) {
      f.flags = kPreGridTooLarge;
      f.dims[0] = f.dims[1] = f.dims[2] = 0;
    } else {
      f.words = (int)words;
    }
  }
  return f;
}

__global__ void __launch_bounds__(256) k_vox_mark_frame(const float4* __restrict__ raw, int n_pad, WsBox ws,
                                                        int do_filter, const int* __restrict__ part, int nb,
                                                        float cell, int cap_words, PreFrame* __restrict__ pf,
                                                        unsigned* __restrict__ bitmap) {
  __shared__ PreFrame f_sh;
  if (wave_id() == 0) {
    const PreFrame f = vox_from_partials(part, nb, cell, cap_words);
    if (lane_id() == 0) {
      f_sh = f;
      if (blockIdx.x == 0) *pf = f;  // (also clears n_vox, n_cand and the flags of the previous frame)
    }
  }
  __syncthreads();
  if (f_sh.words <= 0) return;  // uniform
  VoxDesc v;
  v.cell = f_sh.cell;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    v.mn[a] = f_sh.mn[a];
    v.dims[a] = f_sh.dims[a];
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  const float4 p = raw[i];
  if (!ws_keep(p, ws, do_filter)) return;
  const long long key = vox_key(p, v);
  // (a load first, skipping the atomic when the bit is already set -- two to three points share a voxel --
  // measured SLOWER: 61 against 41 us for 765 k points in random order; the atomics are not what limits it)
  atomicOr(&bitmap[key >> 5], 1u << (key & 31));
}

// exclusive prefix of v over the 256 threads of the workgroup; *total = the sum (uniform)
__device__ __forceinline__ int block_excl_scan_256(int v, int* total) {
  __shared__ int ws_[4];
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane_id() >= o) inc += t;
  }
  __syncthreads();  // (a previous use of ws_ has been read)
  if (lane_id() == 63) ws_[wave_id()] = inc;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave_id(); w++) woff += ws_[w];
  *total = ws_[0] + ws_[1] + ws_[2] + ws_[3];
  return woff + inc - v;
}

// One thread per bitmap word.  Voxel value = index * cell + min (cloud_camera.cpp:155-157); the voxel's
// position in the output (its rank) is its index in the processed cloud, so the sub-sampling key of that
// index can be formed here.  gf.out != nullptr: thread 0 also leaves the search grid's description (the
// lattice's extent is known: both end voxels of every axis are occupied, by the extreme points).
__global__ void __launch_bounds__(256) k_vox_emit_frame(const unsigned* __restrict__ bitmap,
                                                        const unsigned* __restrict__ wrank, int cap_words,
                                                        PreFrame* __restrict__ pf, float4* __restrict__ out,
                                                        int n_max, unsigned num_samples,
                                                        unsigned long long seed_arg,
                                                        const FrameArgs* __restrict__ fa,
                                                        unsigned long long* __restrict__ cand_h,
                                                        unsigned* __restrict__ cand_i, unsigned cand_cap,
                                                        GridFromParts gf) {
  VoxDesc v;
  v.cell = pf->cell;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    v.mn[a] = pf->mn[a];
    v.dims[a] = pf->dims[a];
  }
  const int words = pf->words;
  const unsigned n_vox = wrank[cap_words];
  const unsigned long long seed = fa ? fa->sample_seed : seed_arg;
  const unsigned long long thr = cand_threshold(num_samples, n_vox);
  const bool want_cand = num_samples > 0u && n_vox > num_samples && n_vox <= (unsigned)n_max;
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned bits0 = (w < words) ? bitmap[w] : 0u;
  const unsigned r0 = (w < words) ? wrank[w] : 0u;
  // (ix, iy, iz) of the word's first voxel by division once per word (32-bit when the lattice allows);
  // the set bits are then walked with carries: two 64-bit divisions per VOXEL were most of this kernel
  int bx = 0, by = 0, bz = 0;
  if (bits0) {
    const long long plane = (long long)v.dims[1] * v.dims[2];
    const long long key0 = (long long)w << 5;
    if (((long long)words << 5) < (1ll << 31)) {
      const unsigned k0 = (unsigned)key0, pl = (unsigned)plane, dz = (unsigned)v.dims[2];
      bx = (int)(k0 / pl);
      const unsigned rem = k0 - (unsigned)bx * pl;
      by = (int)(rem / dz);
      bz = (int)(rem - (unsigned)by * dz);
    } else {
      bx = (int)(key0 / plane);
      const long long rem = key0 - (long long)bx * plane;
      by = (int)(rem / v.dims[2]);
      bz = (int)(rem - (long long)by * v.dims[2]);
    }
  }
  int my = 0;
  {
    unsigned bits = bits0, r = r0;
    int ix = bx, iy = by, iz = bz, at = 0;
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      iz += b - at;
      at = b;
      while (iz >= v.dims[2]) {
        iz -= v.dims[2];
        if (++iy >= v.dims[1]) {
          iy = 0;
          ix++;
        }
      }
      if (r < (unsigned)n_max)
        out[r] = make_float4((float)ix * v.cell + v.mn[0], (float)iy * v.cell + v.mn[1],
                             (float)iz * v.cell + v.mn[2], __int_as_float(1));
      if (want_cand && draw_u64(seed, kSubsampleStream, (uint64_t)r) <= thr) my++;
      r++;
    }
  }
  // candidates: one returning atomic per workgroup
  if (num_samples > 0u) {  // (uniform)
    __shared__ unsigned s_base;
    int total = 0;
    const int off = block_excl_scan_256(my, &total);
    if (total > 0) {  // (uniform)
      if (threadIdx.x == 0) s_base = atomicAdd(&pf->n_cand, (unsigned)total);
      __syncthreads();
      unsigned dst = s_base + (unsigned)off;
      unsigned bits = bits0, r = r0;
      while (bits && my > 0) {
        bits &= bits - 1u;
        const unsigned long long h = draw_u64(seed, kSubsampleStream, (uint64_t)r);
        if (h <= thr) {
          if (dst < cand_cap) {
            cand_h[dst] = h;
            cand_i[dst] = r;
          }
          dst++;
          my--;
        }
        r++;
      }
    }
  }
  // padding behind the cloud: non-finite points, which no later stage sees as points
  const float nanv = __builtin_nanf("");
  for (long long i = (long long)n_vox + blockIdx.x * blockDim.x + threadIdx.x; i < n_max;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = make_float4(nanv, nanv, nanv, 0.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned fl = 0u;
    if (n_vox > (unsigned)n_max) fl |= kPreTooManyVoxels;
    if (num_samples > 0u && n_vox <= num_samples) fl |= kPreAllPoints;
    if (fl) atomicOr(&pf->flags, fl);
    pf->n_vox = n_vox;
    pf->thr = thr;
    if (gf.out) {
      float bmin[3], bmax[3];
      for (int a = 0; a < 3; a++) {
        bmin[a] = v.mn[a];
        bmax[a] = (float)(v.dims[a] - 1) * v.cell + v.mn[a];  // same float expression as the emit above
      }
      *gf.out = grid_from_extent(bmin, bmax, (fl & kPreTooManyVoxels) ? 0 : (int)n_vox, gf);
    }
  }
}

// candidates of the sub-sampling when the cloud already exists (ag2_subsample_uniformly)
__global__ void __launch_bounds__(256) k_sel_candidates(int n, unsigned long long seed, unsigned long long thr,
                                                        PreFrame* __restrict__ pf,
                                                        unsigned long long* __restrict__ cand_h,
                                                        unsigned* __restrict__ cand_i, unsigned cand_cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    pf->thr = thr;
    pf->n_vox = (unsigned)n;
  }
  const unsigned long long h = (i < n) ? draw_u64(seed, kSubsampleStream, (uint64_t)i) : ~0ull;
  const int my = (i < n && h <= thr) ? 1 : 0;
  __shared__ unsigned s_base;
  int total = 0;
  const int off = block_excl_scan_256(my, &total);
  if (total == 0) return;  // uniform
  if (threadIdx.x == 0) s_base = atomicAdd(&pf->n_cand, (unsigned)total);
  __syncthreads();
  const unsigned dst = s_base + (unsigned)off;
  if (my && dst < cand_cap) {
    cand_h[dst] = h;
    cand_i[dst] = (unsigned)i;
  }
}

// rank of every candidate among the candidates by (hash, index); the one of rank k - 1 is the threshold
constexpr int kSelStage = 1024;
__global__ void __launch_bounds__(256) k_sel_rank(const unsigned long long* __restrict__ cand_h,
                                                  const unsigned* __restrict__ cand_i, unsigned cand_cap,
                                                  unsigned k, PreFrame* __restrict__ pf) {
  __shared__ unsigned long long sh[kSelStage];
  __shared__ unsigned si[kSelStage];
  const unsigned n_cand = pf->n_cand;
  const int M = (int)min(n_cand, cand_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned fl = 0u;
    if (n_cand > cand_cap) fl |= kPreCandOverflow;
    if ((unsigned)M < k && !(pf->flags & kPreAllPoints)) fl |= kPreCandShort;
    if (fl) atomicOr(&pf->flags, fl);
  }
  if ((int)(blockIdx.x * blockDim.x) >= M) return;  // uniform
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long hj = (j < M) ? cand_h[j] : ~0ull;
  const unsigned ij = (j < M) ? cand_i[j] : 0xFFFFFFFFu;
  unsigned rank = 0;
  for (int c0 = 0; c0 < M; c0 += kSelStage) {
    __syncthreads();
    for (int t = threadIdx.x; t < kSelStage; t += blockDim.x) {
      const bool in = c0 + t < M;
      sh[t] = in ? cand_h[c0 + t] : ~0ull;       // padding: never smaller than a real key
      si[t] = in ? cand_i[c0 + t] : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int t = 0; t < kSelStage; t += 4) {
      unsigned long long h4[4];
      unsigned i4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        h4[u] = sh[t + u];
        i4[u] = si[t + u];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) rank += (unsigned)(h4[u] < hj) | ((unsigned)(h4[u] == hj) & (unsigned)(i4[u] < ij));
    }
  }
  if (j < M && rank + 1u == k) {
    pf->kth_h = hj;
    pf->kth_i = ij;
  }
}

// the selected indices in ascending order: position = number of selected candidates with a smaller index
__global__ void __launch_bounds__(256) k_sel_emit(const unsigned long long* __restrict__ cand_h,
                                                  const unsigned* __restrict__ cand_i, unsigned cand_cap,
                                                  const PreFrame* __restrict__ pf, int* __restrict__ out) {
  __shared__ unsigned si[kSelStage];
  if (pf->flags != 0u) return;  // uniform: the caller repeats the step on the general path
  const int M = (int)min(pf->n_cand, cand_cap);
  if ((int)(blockIdx.x * blockDim.x) >= M) return;  // uniform
  const unsigned long long kh = pf->kth_h;
  const unsigned ki = pf->kth_i;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long hj = (j < M) ? cand_h[j] : ~0ull;
  const unsigned ij = (j < M) ? cand_i[j] : 0xFFFFFFFFu;
  const bool sel = j < M && (hj < kh || (hj == kh && ij <= ki));
  unsigned pos = 0;
  for (int c0 = 0; c0 < M; c0 += kSelStage) {
    __syncthreads();
    for (int t = threadIdx.x; t < kSelStage; t += blockDim.x) {
      unsigned v = 0xFFFFFFFFu;  // not selected / padding: never smaller than a real index
      if (c0 + t < M) {
        const unsigned long long h = cand_h[c0 + t];
        const unsigned i = cand_i[c0 + t];
        if (h < kh || (h == kh && i <= ki)) v = i;
      }
      si[t] = v;
    }
    __syncthreads();
    for (int t = 0; t < kSelStage; t += 8) {
      unsigned i8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) i8[u] = si[t + u];
#pragma unroll
      for (int u = 0; u < 8; u++) pos += (unsigned)(i8[u] < ij);
    }
  }
  if (sel) out[pos] = (int)ij;
}

// ---- the usual case (candidate list <= 8192 entries): the whole selection in ONE workgroup, O(M) ----
// The two kernels above cost M^2 compares (M ~ 2 400 for 2 000 samples: 80 + 25 us).  Hashes and indices are
// both uniform, so linear buckets sort them: (1) 2048 buckets over the hash range [0, thr] locate the
// bucket of the k-th smallest key, whose few members are ranked exactly; (2) the selected candidates are
// bucketed by index, each bucket (about one entry) is put in order by its owner thread, and the list goes
// out in ascending index order.  Same result as k_sel_rank + k_sel_emit.
constexpr int kSelThreads = 1024;
constexpr int kSelBins = 2048;
constexpr int kSelCapMax = 8192;   // candidates this kernel takes (8 per thread)
constexpr int kSelMembers = 128;   // members of the threshold bucket it can rank (expected: one or two)
__device__ __forceinline__ unsigned sel_block_scan(unsigned v, unsigned* total, unsigned* wsum /* 16 words of LDS */) {
  unsigned inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)inc, o, 64);
    if (lane_id() >= o) inc += t;
  }
  __syncthreads();
  if (lane_id() == 63) wsum[wave_id()] = inc;
  __syncthreads();
  unsigned woff = 0, tot = 0;
  for (int w = 0; w < kSelThreads / 64; w++) {
    const unsigned x = wsum[w];
    if (w < wave_id()) woff += x;
    tot += x;
  }
  *total = tot;
  return woff + inc - v;  // exclusive
}

__global__ void __launch_bounds__(kSelThreads) k_sel_small(const unsigned long long* __restrict__ cand_h,
                                                          const unsigned* __restrict__ cand_i, unsigned cand_cap,
                                                          unsigned k, unsigned n_arg, PreFrame* __restrict__ pf,
                                                          int* __restrict__ out) {
  // (the candidates themselves live in registers, eight per thread; LDS holds the histograms and the output)
  __shared__ unsigned long long mh[kSelMembers];
  __shared__ unsigned mi[kSelMembers];
  __shared__ unsigned hist[kSelBins];
  __shared__ unsigned start[kSelBins + 1];
  __shared__ unsigned wsum[kSelThreads / 64];
  __shared__ unsigned ctl[4];               // threshold bucket, rank inside it, its members, -
  __shared__ unsigned sorted[kSelCapMax];   // the selected indices, bucket by bucket
  const int tid = threadIdx.x;
  const unsigned n_cand = pf->n_cand;
  const unsigned M = min(n_cand, cand_cap);
  const unsigned flags_in = pf->flags;
  unsigned fl = 0u;
  if (n_cand > cand_cap) fl |= kPreCandOverflow;
  if (M < k && !(flags_in & kPreAllPoints)) fl |= kPreCandShort;
  if (fl && tid == 0) atomicOr(&pf->flags, fl);
  if (fl || flags_in) return;  // uniform: the caller repeats the step on the general path
  const unsigned long long thr = pf->thr;
  const unsigned n = n_arg ? n_arg : pf->n_vox;
  // buckets: hash >> hshift < kSelBins for every hash <= thr; index >> ishift < kSelBins for every index < n
  int hshift = 0, ishift = 0;
  while ((thr >> hshift) >= (unsigned long long)kSelBins) hshift++;
  while (((n - 1u) >> ishift) >= (unsigned)kSelBins) ishift++;
  for (int b = tid; b < kSelBins; b += kSelThreads) hist[b] = 0u;
  if (tid < 4) ctl[tid] = 0u;
  __syncthreads();
  unsigned long long myh[kSelCapMax / kSelThreads];
  unsigned myi[kSelCapMax / kSelThreads];
#pragma unroll
  for (int u = 0; u < kSelCapMax / kSelThreads; u++) {
    const unsigned j = (unsigned)tid + (unsigned)u * kSelThreads;
    myh[u] = ~0ull;
    myi[u] = 0xFFFFFFFFu;
    if (j < M) {
      myh[u] = cand_h[j];
      myi[u] = cand_i[j];
      atomicAdd(&hist[(unsigned)(myh[u] >> hshift)], 1u);
    }
  }
  __syncthreads();
  // (1) the bucket that holds the k-th smallest hash
  {
    const unsigned a = hist[2 * tid], b = hist[2 * tid + 1];
    unsigned total = 0;
    const unsigned before = sel_block_scan(a + b, &total, wsum);
    if (k > before && k <= before + a) {
      ctl[0] = 2u * tid;
      ctl[1] = k - before;
    } else if (k > before + a && k <= before + a + b) {
      ctl[0] = 2u * tid + 1u;
      ctl[1] = k - before - a;
    }
  }
  __syncthreads();
  const unsigned B = ctl[0], rB = ctl[1];
#pragma unroll
  for (int u = 0; u < kSelCapMax / kSelThreads; u++) {
    if (myi[u] != 0xFFFFFFFFu && (unsigned)(myh[u] >> hshift) == B) {
      const unsigned m = atomicAdd(&ctl[2], 1u);
      if (m < (unsigned)kSelMembers) {
        mh[m] = myh[u];
        mi[m] = myi[u];
      }
    }
  }
  __syncthreads();
  const unsigned nm = ctl[2];
  if (nm > (unsigned)kSelMembers) {  // (uniform) hashes that do not spread: leave it to the general path
    if (tid == 0) atomicOr(&pf->flags, kPreCandShort);
    return;
  }
  // the member of rank rB (1-based) by (hash, index): every thread finds it (nm is tiny)
  unsigned long long kh = 0ull;
  unsigned ki = 0u;
  for (unsigned a = 0; a < nm; a++) {
    unsigned r = 1u;
    for (unsigned b = 0; b < nm; b++) r += (unsigned)(mh[b] < mh[a]) | ((unsigned)(mh[b] == mh[a]) & (unsigned)(mi[b] < mi[a]));
    if (r == rB) {
      kh = mh[a];
      ki = mi[a];
    }
  }
  if (tid == 0) {
    pf->kth_h = kh;
    pf->kth_i = ki;
  }
  // (2) the selected candidates by index bucket
  for (int b = tid; b < kSelBins; b += kSelThreads) hist[b] = 0u;
  __syncthreads();
  unsigned selmask = 0u;
#pragma unroll
  for (int u = 0; u < kSelCapMax / kSelThreads; u++) {
    const bool sel = myi[u] != 0xFFFFFFFFu && (myh[u] < kh || (myh[u] == kh && myi[u] <= ki));
    if (sel) {
      selmask |= 1u << u;
      atomicAdd(&hist[myi[u] >> ishift], 1u);
    }
  }
  __syncthreads();
  {
    const unsigned a = hist[2 * tid], b = hist[2 * tid + 1];
    unsigned total = 0;
    const unsigned before = sel_block_scan(a + b, &total, wsum);
    start[2 * tid] = before;
    start[2 * tid + 1] = before + a;
    if (tid == kSelThreads - 1) start[kSelBins] = before + a + b;  // == k
  }
  __syncthreads();
  for (int b = tid; b < kSelBins; b += kSelThreads) hist[b] = 0u;  // now: entries placed per bucket
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kSelCapMax / kSelThreads; u++) {
    if (selmask & (1u << u)) {
      const unsigned b = myi[u] >> ishift;
      sorted[start[b] + atomicAdd(&hist[b], 1u)] = myi[u];
    }
  }
  __syncthreads();
  for (int b = tid; b < kSelBins; b += kSelThreads) {  // order inside a bucket: insertion sort over a handful
    const unsigned lo = start[b], hi = start[b + 1];
    for (unsigned x = lo + 1; x < hi; x++) {
      const unsigned v = sorted[x];
      unsigned y = x;
      while (y > lo && sorted[y - 1] > v) {
        sorted[y] = sorted[y - 1];
        y--;
      }
      sorted[y] = v;
    }
  }
  __syncthreads();
  const unsigned n_sel = start[kSelBins];
  for (unsigned x = tid; x < n_sel; x += kSelThreads) out[x] = (int)sorted[x];
}

// the exact selection among the candidates: one workgroup when the list fits it, else the two M^2 kernels
static int launch_selection(ag2_ctx* c, unsigned long long* cand_h, unsigned* cand_i, size_t cand_cap, size_t k,
                            size_t n_arg, PreFrame* pf, int* out) {
  static const bool pair_only = getenv("AG2_SEL_PAIR") != nullptr;  // A/B and tests of the two-kernel form
  if (cand_cap <= (size_t)kSelCapMax && cand_cap > 0 && !pair_only) {
    hipLaunchKernelGGL(k_sel_small, dim3(1), dim3(kSelThreads), 0, c->stream, cand_h, cand_i, (unsigned)cand_cap,
                       (unsigned)k, (unsigned)n_arg, pf, out);
  } else {
    const unsigned sel_blocks = (unsigned)((std::max<size_t>(cand_cap, 1) + 255) / 256);
    hipLaunchKernelGGL(k_sel_rank, dim3(sel_blocks), dim3(256), 0, c->stream, cand_h, cand_i, (unsigned)cand_cap,
                       (unsigned)k, pf);
    hipLaunchKernelGGL(k_sel_emit, dim3(sel_blocks), dim3(256), 0, c->stream, cand_h, cand_i, (unsigned)cand_cap, pf, out);
  }
  AG2_HIP(c, hipGetLastError());
  return 0;
}

static WsBox ws_of(const ag2_ctx* c) {
  WsBox ws;
  for (int k = 0; k < 6; k++) ws.b[k] = c->p.workspace[k];
  return ws;
}

// the pack + workspace test + extent kernel of a frame: reads the caller's buffer, so it runs in front of
// the captured sequence (its pointer and count change with every frame)
int front_pack_raw(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes, const FrontShapes& fs) {
  AG2_HIP(c, c->d_raw.reserve(std::max<size_t>(fs.raw_max, 1) * 16));
  AG2_HIP(c, c->d_bounds.reserve((size_t)kRawBlocks * 8 * 4));
  const int nb = std::min(((int)fs.raw_max + 255) / 256, kRawBlocks);
  hipLaunchKernelGGL(k_raw_filter_bounds<true>, dim3(nb), dim3(256), 0, c->stream, (const char*)d_xyz,
                     stride_bytes, c->d_raw.as<float4>(), (int)n, (int)fs.raw_max, ws_of(c),
                     fs.filter_workspace, c->d_bounds.as<int>(), c->d_stats.as<DevStats>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// Everything of the front end behind that kernel, at the fixed shapes `fs`: voxel grid -> d_xyz_in
// (padded to fs.n_max), search-grid description -> d_griddesc, sample indices -> d_samples.  No host round
// trip, no allocation once the buffers have their size: capturable.
int enqueue_front_frame(ag2_ctx* c, const FrontShapes& fs) {
  const int cap_words = (int)fs.cap_words;
  const size_t ctl_words = (scan_ctl_words(cap_words + 1) + 3) & ~size_t(3);
  const size_t bm_words = ((size_t)cap_words + 3) & ~size_t(3);
  // bitmap and the scan's control words share one buffer: one fill clears both
  AG2_HIP(c, c->d_bitmap.reserve((bm_words + ctl_words) * 4));
  AG2_HIP(c, c->d_wrank.reserve(((size_t)cap_words + 1) * 4));
  AG2_HIP(c, c->d_preframe.reserve(sizeof(PreFrame)));
  AG2_HIP(c, c->d_cand.reserve(std::max<size_t>(fs.cand_cap, 1) * 12));
  AG2_HIP(c, c->d_samples.reserve(std::max<size_t>(fs.num_samples, 1) * 4));
  AG2_HIP(c, c->d_xyz_in.reserve(std::max<size_t>(fs.n_max, 1) * 16));
  AG2_HIP(c, c->d_griddesc.reserve(sizeof(GridDesc)));
  unsigned* bitmap = c->d_bitmap.as<unsigned>();
  PreFrame* pf = c->d_preframe.as<PreFrame>();
  AG2_HIP(c, hipMemsetAsync(bitmap, 0, (bm_words + ctl_words) * 4, c->stream));
  const int nb = std::min(((int)fs.raw_max + 255) / 256, kRawBlocks);
  hipLaunchKernelGGL(k_vox_mark_frame, dim3(((unsigned)fs.raw_max + 255) / 256), dim3(256), 0, c->stream,
                     c->d_raw.as<float4>(), (int)fs.raw_max, ws_of(c), fs.filter_workspace,
                     c->d_bounds.as<int>(), nb, fs.cell, cap_words, pf, bitmap);
  int rc = scan_popc_u32(c, bitmap, cap_words, c->d_wrank.as<unsigned>(), bitmap + bm_words);
  if (rc) return rc;
  GridFromParts gf{};
  gf.inv = 1.0f / (float)c->p.grid_cell;
  gf.origin_set = c->origin_set ? 1 : 0;
  for (int a = 0; a < 3; a++) gf.org[a] = c->origin[a];
  gf.cap_cells = (int)c->fm_cap_cells;
  gf.out = c->d_griddesc.as<GridDesc>();
  unsigned long long* cand_h = c->d_cand.as<unsigned long long>();
  unsigned* cand_i = reinterpret_cast<unsigned*>(cand_h + std::max<size_t>(fs.cand_cap, 1));
  hipLaunchKernelGGL(k_vox_emit_frame, dim3(((unsigned)cap_words + 255) / 256), dim3(256), 0, c->stream, bitmap,
                     c->d_wrank.as<unsigned>(), cap_words, pf, c->d_xyz_in.as<float4>(), (int)fs.n_max,
                     (unsigned)fs.num_samples, 0ull, c->fm_args_dev, cand_h, cand_i, (unsigned)fs.cand_cap, gf);
  AG2_HIP(c, hipGetLastError());
  if (fs.num_samples == 0) return 0;  // (ag2_preprocess_cloud*: no sub-sampling)
  return launch_selection(c, cand_h, cand_i, fs.cand_cap, fs.num_samples, 0, pf, c->d_samples.as<int>());
}

// d_raw (and d_raw_nrm when have_nrm) hold n points
static int preprocess_resident(ag2_ctx* c, size_t n, bool have_cam, bool have_nrm,
                               int filter_workspace, int voxelize, double voxel_size, int flags,
                               size_t* n_out) {
  if (voxelize && have_nrm)
    return set_err(c, AG2_ERR_ARG, "normals do not survive voxelisation (cloud_camera.cpp:124-168)");
  if (voxelize && !((float)voxel_size > 0.f)) return set_err(c, AG2_ERR_ARG, "voxel_size must be positive");
  c->has_cloud = c->has_normals = false;
  AG2_HIP(c, stage_event(c, 14));
  auto finish = [&](size_t m) -> int {
    c->n = m;
    if (n_out) *n_out = m;
    AG2_HIP(c, stage_event(c, 15));
    int rc = after_cloud(c);
    if (rc) return rc;
    if (have_nrm && c->n_valid) {  // d_tmp holds the survivors' normals in cloud order
      rc = gather_normals(c);
      if (rc) return rc;
      c->has_normals = true;
    }
    AG2_HIP(c, stage_sync(c, 15));
    stage_elapsed(c, &c->times.preprocess_ms, 14, 15);
    return 0;
  };
  AG2_HIP(c, c->d_xyz_in.reserve(std::max<size_t>(n, 1) * 16));
  if (n == 0) return finish(0);
  // Fast path (the usual call: one camera, voxel grid on, a bounded workspace): the kernels of the
  // fixed-shape front end above, with ONE read-back at the end (the processed cloud's size) instead of
  // two and no compaction pass.  The workspace bounds the voxel lattice, hence the bitmap.
  static const bool fast_off = getenv("AG2_PRE_GENERAL") != nullptr;  // A/B and tests of the general path
  if (!fast_off && voxelize && !have_cam && !have_nrm && filter_workspace) {
    double cells = 1.0;
    for (int a = 0; a < 3; a++)
      cells *= floor((c->p.workspace[2 * a + 1] - c->p.workspace[2 * a]) / (double)(float)voxel_size) + 2.0;
    if (cells >= 1.0 && cells < 4.0e9) {  // (<= 500 MB of bitmap; an unbounded workspace takes the general path)
      FrontShapes fs{};
      fs.raw_max = n;
      fs.cap_words = (size_t)(cells / 32.0) + 2;
      fs.n_max = n;
      fs.num_samples = 0;
      fs.cand_cap = 0;
      fs.cell = (float)voxel_size;
      fs.filter_workspace = 1;
      AG2_HIP(c, c->d_bounds.reserve((size_t)kRawBlocks * 8 * 4));
      const int nb = std::min(((int)n + 255) / 256, kRawBlocks);
      hipLaunchKernelGGL(k_raw_filter_bounds<false>, dim3(nb), dim3(256), 0, c->stream, (const char*)nullptr,
                         (size_t)0, c->d_raw.as<float4>(), (int)n, (int)n, ws_of(c), 1, c->d_bounds.as<int>(),
                         (DevStats*)nullptr);
      const size_t cap_cells_saved = c->fm_cap_cells;
      const FrameArgs* fa_saved = c->fm_args_dev;
      c->fm_args_dev = nullptr;
      int rc = enqueue_front_frame(c, fs);  // (the grid description it leaves is not used on this path)
      c->fm_args_dev = fa_saved;
      c->fm_cap_cells = cap_cells_saved;
      if (rc) return rc;
      PreFrame hf;
      AG2_HIP(c, hipMemcpyAsync(pin_small(c), c->d_preframe.p, sizeof(hf), hipMemcpyDeviceToHost, c->stream));
      AG2_HIP(c, hipStreamSynchronize(c->stream));
      __builtin_memcpy(&hf, pin_small(c), sizeof(hf));
      if (!(hf.flags & (kPreGridTooLarge | kPreTooManyVoxels))) {
        const size_t m = hf.n_vox;
        for (int a = 0; a < 3; a++) c->last_vox_dims[a] = hf.dims[a];
        for (int a = 0; a < 3 && m; a++) {
          c->known_min[a] = hf.mn[a];
          c->known_max[a] = (float)(hf.dims[a] - 1) * hf.cell + hf.mn[a];
        }
        c->bounds_known = m > 0;
        return finish(m);
      }
      // (cannot happen with a lattice inside the workspace; the general path below decides)
    }
  }
  AG2_HIP(c, c->d_prestats.reserve(sizeof(PreStats)));
  AG2_HIP(c, c->d_pflags.reserve((n + 1) * 4));
  if (voxelize) AG2_HIP(c, c->d_pre.reserve(n * 16));
  if (have_nrm) AG2_HIP(c, c->d_tmp.reserve(n * 16));
  PreStats* ps = c->d_prestats.as<PreStats>();
  unsigned* pf = c->d_pflags.as<unsigned>();
  const float4* raw = c->d_raw.as<float4>();
  float4* kept = voxelize ? c->d_pre.as<float4>() : c->d_xyz_in.as<float4>();
  WsBox ws;
  for (int k = 0; k < 6; k++) ws.b[k] = c->p.workspace[k];
  const int ni = (int)n, g256 = (ni + 255) / 256;
  hipLaunchKernelGGL(k_pre_init, dim3(1), dim3(64), 0, c->stream, ps);
  hipLaunchKernelGGL(k_pre_flag_bounds, dim3(std::min(g256, 1024)), dim3(256), 0, c->stream, raw, ni,
                     ws, filter_workspace ? 1 : 0, pf, ps);
  int rc = scan_exclusive_u32(c, pf, ni + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pre_compact, dim3(g256), dim3(256), 0, c->stream, raw,
                     have_nrm ? c->d_raw_nrm.as<float4>() : (const float4*)nullptr, pf, ni, kept,
                     have_nrm ? c->d_tmp.as<float4>() : (float4*)nullptr);
  PreStats hs;
  AG2_HIP(c, hipMemcpyAsync(pin_small(c), ps, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  __builtin_memcpy(&hs, pin_small(c), sizeof(hs));
  const size_t m = hs.n_keep;
  if (!voxelize || m == 0) {
    if (m) {  // extent of the survivors: the grid build needs no bounds pass of its own
      for (int a = 0; a < 3; a++) {
        c->known_min[a] = ord2f(hs.mn[a]);
        c->known_max[a] = ord2f(hs.mx[a]);
      }
      c->bounds_known = true;
    }
    return finish(m);
  }

  VoxDesc v{};
  v.cell = (float)voxel_size;
  long long ncells = 1;
  for (int a = 0; a < 3; a++) {
    v.mn[a] = ord2f(hs.mn[a]);
    const float mx = ord2f(hs.mx[a]);
    const float top = floorf((mx - v.mn[a]) / v.cell);  // same expression as vox_key
    if (!(top < 2.0e6f)) return set_err(c, AG2_ERR_CAPACITY, "voxel grid too fine for the cloud extent");
    v.dims[a] = (int)top + 1;
    ncells *= v.dims[a];
    if (ncells > (1ll << 33)) return set_err(c, AG2_ERR_CAPACITY, "voxel grid has more than 2^33 cells");
  }
  for (int a = 0; a < 3; a++) c->last_vox_dims[a] = v.dims[a];
  const int words = (int)((ncells + 31) >> 5);
  const int mi = (int)m, gm = (mi + 255) / 256, gw = (words + 256) / 256;
  AG2_HIP(c, c->d_bitmap.reserve((size_t)words * 4));
  AG2_HIP(c, c->d_wrank.reserve(((size_t)words + 1) * 4));
  unsigned* bitmap = c->d_bitmap.as<unsigned>();
  unsigned* wrank = c->d_wrank.as<unsigned>();
  float4* out = c->d_xyz_in.as<float4>();
  AG2_HIP(c, hipMemsetAsync(bitmap, 0, (size_t)words * 4, c->stream));
  hipLaunchKernelGGL(k_vox_mark, dim3(gm), dim3(256), 0, c->stream, kept, mi, v, bitmap);
  hipLaunchKernelGGL(k_vox_popc, dim3(gw), dim3(256), 0, c->stream, bitmap, words, wrank);
  rc = scan_exclusive_u32(c, wrank, words + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_vox_emit, dim3(gw), dim3(256), 0, c->stream, bitmap, wrank, words, v, out);
  if (have_cam) {
    AG2_HIP(c, c->d_first.reserve(m * 4));
    int* first = c->d_first.as<int>();
    AG2_HIP(c, hipMemsetAsync(first, 0x7f, m * 4, c->stream));
    hipLaunchKernelGGL(k_vox_first, dim3(gm), dim3(256), 0, c->stream, kept, mi, v, bitmap, wrank, first);
    if (flags & 1) {
      hipLaunchKernelGGL(k_vox_cam_owner, dim3(gm), dim3(256), 0, c->stream, kept, first, mi,
                         wrank + words, out);
    } else {
      hipLaunchKernelGGL(k_vox_firsthit_flags, dim3((mi + 256) / 256), dim3(256), 0, c->stream, kept, mi,
                         v, bitmap, wrank, first, pf);
      rc = scan_exclusive_u32(c, pf, mi + 1);
      if (rc) return rc;
      hipLaunchKernelGGL(k_vox_cam_literal, dim3(gm), dim3(256), 0, c->stream, kept, mi, pf, out);
    }
  }
  AG2_HIP(c, hipGetLastError());
  unsigned n_vox = 0;
  AG2_HIP(c, hipMemcpyAsync(pin_small(c), wrank + words, 4, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  __builtin_memcpy(&n_vox, pin_small(c), 4);
  // the voxel lattice spans [mn, (dims - 1) * cell + mn] per axis (both end voxels are occupied, by
  // the extreme points); same float expression as k_vox_emit
  for (int a = 0; a < 3; a++) {
    c->known_min[a] = v.mn[a];
    c->known_max[a] = (float)(v.dims[a] - 1) * v.cell + v.mn[a];
  }
  c->bounds_known = n_vox > 0;
  return finish(n_vox);
}

}  // namespace ag2

using namespace ag2;

extern "C" {

int ag2_preprocess_cloud(ag2_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                         const int32_t* cam_source, int n_cams, const double* normals,
                         int filter_workspace, int voxelize, double voxel_size, int flags,
                         size_t* n_out) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (n_cams != c->p.n_cams) return set_err(c, AG2_ERR_ARG, "n_cams differs from ag2_params.n_cams");
  if (stride_bytes < 12 || stride_bytes % 4 != 0) return set_err(c, AG2_ERR_ARG, "bad stride");
  if (n > 0 && !xyz) return set_err(c, AG2_ERR_ARG, "xyz is NULL");
  if (n > (size_t)1 << 30) return set_err(c, AG2_ERR_CAPACITY, "more than 2^30 points");
  std::vector<float> pack(n * 4);
  const char* base = (const char*)xyz;
  for (size_t i = 0; i < n; i++) {
    const float* pt = (const float*)(base + i * stride_bytes);
    int mask = 0;
    for (int cam = 0; cam < n_cams; cam++) {
      const int v = cam_source ? cam_source[i * (size_t)n_cams + cam] : 1;  // cloud_camera.cpp:59
      if (v == 1) mask |= (1 << cam);
    }
    pack[4 * i] = pt[0];
    pack[4 * i + 1] = pt[1];
    pack[4 * i + 2] = pt[2];
    __builtin_memcpy(&pack[4 * i + 3], &mask, 4);
  }
  AG2_HIP(c, c->d_raw.reserve(std::max<size_t>(n, 1) * 16));
  if (n) AG2_HIP(c, hipMemcpyAsync(c->d_raw.p, pack.data(), n * 16, hipMemcpyHostToDevice, c->stream));
  std::vector<float> nf;
  if (normals && n) {  // cloud_camera.cpp:27-31: float PointNormal fields
    nf.assign(n * 4, 0.f);
    for (size_t i = 0; i < n; i++) {
      nf[4 * i] = (float)normals[3 * i];
      nf[4 * i + 1] = (float)normals[3 * i + 1];
      nf[4 * i + 2] = (float)normals[3 * i + 2];
    }
    AG2_HIP(c, c->d_raw_nrm.reserve(n * 16));
    AG2_HIP(c, hipMemcpyAsync(c->d_raw_nrm.p, nf.data(), n * 16, hipMemcpyHostToDevice, c->stream));
  }
  AG2_HIP(c, hipStreamSynchronize(c->stream));  // staging vectors go out of scope
  // without explicit masks every voxel gets the one-camera mask 1 that k_vox_emit writes
  return preprocess_resident(c, n, cam_source != nullptr || n_cams > 1, normals != nullptr && n > 0,
                             filter_workspace, voxelize, voxel_size, flags, n_out);
}

int ag2_preprocess_cloud_device(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes,
                                int filter_workspace, int voxelize, double voxel_size,
                                size_t* n_out) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (c->p.n_cams != 1) return set_err(c, AG2_ERR_ARG, "device clouds are single-camera");
  if (stride_bytes < 12 || stride_bytes % 4 != 0) return set_err(c, AG2_ERR_ARG, "bad stride");
  if (n > 0 && !d_xyz) return set_err(c, AG2_ERR_ARG, "d_xyz is NULL");
  if (n > (size_t)1 << 30) return set_err(c, AG2_ERR_CAPACITY, "more than 2^30 points");
  AG2_HIP(c, c->d_raw.reserve(std::max<size_t>(n, 1) * 16));
  if (n) {
    const int rc = pack_device_xyz(c, d_xyz, n, stride_bytes, c->d_raw.as<float4>());
    if (rc) return rc;
  }
  return preprocess_resident(c, n, false, false, filter_workspace, voxelize, voxel_size, 0, n_out);
}

int ag2_get_cloud(ag2_ctx* c, float* xyz_nx3, int32_t* cam_source, size_t cap, size_t* n) {
  if (!c || !n) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->has_cloud) return set_err(c, AG2_ERR_STATE, "no cloud set");
  *n = c->n;
  if (cap < c->n) return set_err(c, AG2_ERR_CAPACITY, "cloud buffer too small");
  if (c->n == 0) return 0;
  std::vector<float> pack(c->n * 4);
  AG2_HIP(c, hipMemcpyAsync(pack.data(), c->d_xyz_in.p, c->n * 16, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  const int n_cams = c->p.n_cams;
  for (size_t i = 0; i < c->n; i++) {
    if (xyz_nx3) {
      xyz_nx3[3 * i] = pack[4 * i];
      xyz_nx3[3 * i + 1] = pack[4 * i + 1];
      xyz_nx3[3 * i + 2] = pack[4 * i + 2];
    }
    if (cam_source) {
      int mask;
      __builtin_memcpy(&mask, &pack[4 * i + 3], 4);
      for (int cam = 0; cam < n_cams; cam++) cam_source[i * (size_t)n_cams + cam] = (mask >> cam) & 1;
    }
  }
  return 0;
}

int ag2_get_samples(ag2_ctx* c, int32_t* idx, size_t cap, size_t* n) {
  if (!c || !n) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  *n = c->n_resident_samples;
  if (cap < c->n_resident_samples) return set_err(c, AG2_ERR_CAPACITY, "sample buffer too small");
  if (c->n_resident_samples && !idx) return set_err(c, AG2_ERR_ARG, "idx is NULL");
  if (c->n_resident_samples) {
    AG2_HIP(c, hipMemcpyAsync(idx, c->d_samples.p, c->n_resident_samples * 4, hipMemcpyDeviceToHost, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}

int ag2_subsample_uniformly(ag2_ctx* c, size_t num_samples, uint64_t seed, int32_t* idx_out,
                            size_t cap, size_t* n_out) {
  if (!c || !n_out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->has_cloud) return set_err(c, AG2_ERR_STATE, "no cloud set");
  const size_t n = c->n, k = std::min(num_samples, n);  // grasp_detector.cpp:321-335
  *n_out = k;
  c->n_resident_samples = 0;
  if (idx_out && cap < k) return set_err(c, AG2_ERR_CAPACITY, "sample buffer too small");
  if (k == 0) return 0;
  AG2_HIP(c, c->d_samples.reserve(k * 4));
  int* out = c->d_samples.as<int>();
  const int ni = (int)n, g256 = (ni + 255) / 256;
  bool done = false;
  static const bool fast_off = getenv("AG2_PRE_GENERAL") != nullptr;
  if (k < n && !fast_off) {
    // candidates under a hash threshold, then the exact selection among them (three small kernels); the
    // record that comes back says whether the candidate list held what it had to -- if not (p < 1e-12)
    // the radix select below decides
    const size_t cap = cand_capacity(k);
    AG2_HIP(c, c->d_preframe.reserve(sizeof(PreFrame)));
    AG2_HIP(c, c->d_cand.reserve(cap * 12));
    PreFrame* pf = c->d_preframe.as<PreFrame>();
    unsigned long long* cand_h = c->d_cand.as<unsigned long long>();
    unsigned* cand_i = reinterpret_cast<unsigned*>(cand_h + cap);
    AG2_HIP(c, hipMemsetAsync(pf, 0, sizeof(PreFrame), c->stream));
    hipLaunchKernelGGL(k_sel_candidates, dim3(g256), dim3(256), 0, c->stream, ni, (unsigned long long)seed,
                       cand_threshold((unsigned)k, (unsigned)n), pf, cand_h, cand_i, (unsigned)cap);
    AG2_HIP(c, hipGetLastError());
    {
      const int rc = launch_selection(c, cand_h, cand_i, cap, k, n, pf, out);
      if (rc) return rc;
    }
    PreFrame hf;
    AG2_HIP(c, hipMemcpyAsync(pin_small(c), pf, sizeof(hf), hipMemcpyDeviceToHost, c->stream));
    if (idx_out) AG2_HIP(c, hipMemcpyAsync(idx_out, out, k * 4, hipMemcpyDeviceToHost, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    __builtin_memcpy(&hf, pin_small(c), sizeof(hf));
    done = hf.flags == 0u;
    if (done) {
      c->n_resident_samples = k;
      return 0;
    }
  }
  if (k == n) {
    hipLaunchKernelGGL(k_iota, dim3(g256), dim3(256), 0, c->stream, ni, out);
  } else {
    AG2_HIP(c, c->d_prestats.reserve(sizeof(PreStats)));
    AG2_HIP(c, c->d_hist.reserve(65536 * 4));
    AG2_HIP(c, c->d_pflags.reserve((n + 1) * 4));
    PreStats* ps = c->d_prestats.as<PreStats>();
    unsigned* hist = c->d_hist.as<unsigned>();
    unsigned* pf = c->d_pflags.as<unsigned>();
    AG2_HIP(c, hipMemsetAsync(hist, 0, 65536 * 4, c->stream));
    hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(64), 0, c->stream, ps, (unsigned)k);
    for (int digit = 0; digit < 6; digit++) {
      hipLaunchKernelGGL(k_sel_hist, dim3(g256), dim3(256), 0, c->stream, ni, seed, digit, ps, hist);
      hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(1024), 0, c->stream, hist, digit, ps);
    }
    hipLaunchKernelGGL(k_sel_flags, dim3((ni + 256) / 256), dim3(256), 0, c->stream, ni, seed, ps, pf);
    const int rc = scan_exclusive_u32(c, pf, ni + 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sel_scatter, dim3(g256), dim3(256), 0, c->stream, pf, ni, out);
  }
  AG2_HIP(c, hipGetLastError());
  if (idx_out) {
    AG2_HIP(c, hipMemcpyAsync(idx_out, out, k * 4, hipMemcpyDeviceToHost, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->n_resident_samples = k;
  return 0;
}

}  // extern "C"
