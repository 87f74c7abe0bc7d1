// k_grid.hip -- K0: uniform search grid over the cloud.
//
// Replaces pcl::KdTreeFLANN::setInputCloud (src/agile_grasp2/hand_search.cpp:11-12) and the kd-tree
// inside pcl::NormalEstimationOMP (:87).  Points are counting-sorted by cell key (x fastest) with
// ties broken by original index, so a run of x-adjacent cells is one contiguous, coalesced span of
// float4 points and "ascending sorted position" is the canonical neighbour order the oracle uses.
#include "ag2_internal.h"

namespace ag2 {

__global__ void k_init_stats(DevStats* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    DevStats z{};
    z.bounds[0] = z.bounds[1] = z.bounds[2] = 0x7fffffffu;   // min as ordered int
    z.bounds[3] = z.bounds[4] = z.bounds[5] = 0x80000000u;   // max as ordered int
    *st = z;
  }
}

// strided source (12-byte packed xyz or 32-byte pcl::PointXYZRGBA) -> float4, w = camera mask bits
__global__ void k_pack_xyz(const char* __restrict__ src, size_t stride, int n,
                           float4* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = (const float*)(src + (size_t)i * stride);
  dst[i] = make_float4(p[0], p[1], p[2], __int_as_float(1));
}

__global__ void __launch_bounds__(256) k_bounds(const float4* __restrict__ xyz, int n, DevStats* st) {
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = xyz[i];
    if (finite3(p.x, p.y, p.z)) {
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
  // one set of atomics per workgroup (7 contended addresses chip-wide)
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
    if (k < 3) atomicMin((int*)&st->bounds[k], v);
    else if (k < 6) atomicMax((int*)&st->bounds[k], v);
    else atomicAdd(&st->bounds[6], (unsigned)v);
  }
}

__global__ void k_cell_count(const float4* __restrict__ xyz, int n, GridDesc g,
                             int* __restrict__ key, unsigned* __restrict__ cell) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = xyz[i];
  int k = -1;
  if (finite3(p.x, p.y, p.z)) {
    const int cx = cell_of(p.x, g.o[0], g.inv), cy = cell_of(p.y, g.o[1], g.inv),
              cz = cell_of(p.z, g.o[2], g.inv);
    k = (cz * g.dims[1] + cy) * g.dims[0] + cx;
    atomicAdd(&cell[k], 1u);
  }
  key[i] = k;
}

// ---- exclusive scan of uint32 (2048 elements per block) ---------------------------------------
__global__ void __launch_bounds__(256) k_scan_block(unsigned* __restrict__ data, int n,
                                                    unsigned* __restrict__ sums) {
  __shared__ unsigned wsum[4];
  const int base = blockIdx.x * 2048 + threadIdx.x * 8;
  unsigned v[8];
  unsigned tot = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    v[k] = (base + k < n) ? data[base + k] : 0u;
    tot += v[k];
  }
  // inclusive wave scan of tot
  unsigned inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)inc, o, 64);
    if (lane_id() >= o) inc += t;
  }
  if (lane_id() == 63) wsum[wave_id()] = inc;
  __syncthreads();
  unsigned woff = 0;
  for (int w = 0; w < wave_id(); w++) woff += wsum[w];
  unsigned run = woff + inc - tot;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
  if (sums && threadIdx.x == 255) sums[blockIdx.x] = woff + inc;
}

__global__ void k_scan_add(unsigned* __restrict__ data, int n, const unsigned* __restrict__ sums) {
  const int i = blockIdx.x * 2048 + threadIdx.x;
  const unsigned off = sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int j = i + k * 256;
    if (j < n) data[j] += off;
  }
}

static int scan_exclusive(ag2_ctx* c, unsigned* d, int n, unsigned* scratch, size_t scratch_elems) {
  const int nb = (n + 2047) / 2048;
  if (nb <= 1) {
    hipLaunchKernelGGL(k_scan_block, dim3(1), dim3(256), 0, c->stream, d, n, (unsigned*)nullptr);
    return 0;
  }
  if ((size_t)nb > scratch_elems) return set_err(c, AG2_ERR_CAPACITY, "scan scratch too small");
  hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(256), 0, c->stream, d, n, scratch);
  const int rc = scan_exclusive(c, scratch, nb, scratch + nb, scratch_elems - nb);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, c->stream, d, n, scratch);
  return 0;
}

int scan_exclusive_u32(ag2_ctx* c, unsigned* d, int n) {
  const size_t need = ((size_t)n / 2048 + 8) * 2;
  AG2_HIP(c, c->d_scan.reserve(need * sizeof(unsigned)));
  return scan_exclusive(c, d, n, c->d_scan.as<unsigned>(), need);
}

__global__ void k_scatter(const int* __restrict__ key, int n, const unsigned* __restrict__ cell,
                          unsigned* __restrict__ fill, int* __restrict__ perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = key[i];
  if (k < 0) return;
  const unsigned pos = cell[k] + atomicAdd(&fill[k], 1u);
  perm[pos] = i;
}

// Within a cell the scatter order is whatever the atomics produced; restore ascending original
// index so the layout is deterministic ((key, index) order).  Cells hold ~10-30 points on a 3 mm
// voxel cloud: one thread per cell, insertion sort.
// The 256 cells of a workgroup own one contiguous span of perm: it is staged in LDS (coalesced load
// and store), so the dependent compares and moves of the insertion sorts never touch global memory.
constexpr int kSortStage = 6144;
__global__ void __launch_bounds__(256) k_cell_sort(const unsigned* __restrict__ cell, int ncells,
                                                   int* __restrict__ perm) {
  __shared__ int stage[kSortStage];
  const int c0 = blockIdx.x * 256, c = c0 + threadIdx.x;
  const int lo = (int)cell[c0], hi = (int)cell[min(c0 + 256, ncells)];
  const int b = (c < ncells) ? (int)cell[c] : 0, e = (c < ncells) ? (int)cell[c + 1] : 0;
  if ((hi - lo) <= kSortStage / 2) {  // uniform: LDS path (ds_ instructions, no generic pointers)
    // Rank sort, one thread per ELEMENT: every element counts the smaller indices in its own cell
    // (independent compares, no dependent insertion chains) and is written to (cell start + rank).
    // stage[0, n) = values; stage[n, 2n) = packed (cell start - lo) | (cell length << 16).
    const int n = hi - lo;
    for (int i = lo + threadIdx.x; i < hi; i += 256) stage[i - lo] = perm[i];
    for (int i = b; i < e; i++)  // lengths fit 16 bits: n <= kSortStage / 2
      stage[n + (i - lo)] = (b - lo) | ((e - b) << 16);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      const int v = stage[i];
      const int be = stage[n + i];
      const int cb = be & 0xFFFF, cl = be >> 16;
      int rank = 0;
      for (int k = 0; k < cl; k++) rank += (stage[cb + k] < v) ? 1 : 0;
      perm[lo + cb + rank] = v;  // indices within a cell are distinct: ranks are a permutation
    }
  } else if ((hi - lo) <= kSortStage) {  // staged insertion sort
    for (int i = lo + threadIdx.x; i < hi; i += 256) stage[i - lo] = perm[i];
    __syncthreads();
    const int sb = b - lo, se = e - lo;
    for (int i = sb + 1; i < se; i++) {
      const int v = stage[i];
      int j = i - 1;
      while (j >= sb) {
        const int u = stage[j];
        if (!(u > v)) break;
        stage[j + 1] = u;
        j--;
      }
      stage[j + 1] = v;
    }
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += 256) perm[i] = stage[i - lo];
  } else {  // dense cells: in place in global memory
    for (int i = b + 1; i < e; i++) {
      const int v = perm[i];
      int j = i - 1;
      while (j >= b) {
        const int u = perm[j];
        if (!(u > v)) break;
        perm[j + 1] = u;
        j--;
      }
      perm[j + 1] = v;
    }
  }
}

__global__ void k_gather4(const float4* __restrict__ src, const int* __restrict__ perm, int n,
                          float4* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}

int gather_normals(ag2_ctx* c) {
  hipLaunchKernelGGL(k_gather4, dim3(((int)c->n_valid + 255) / 256), dim3(256), 0, c->stream,
                     c->d_tmp.as<float4>(), c->d_perm.as<int>(), (int)c->n_valid,
                     c->d_nrm.as<float4>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

int pack_device_xyz(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes, float4* dst) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_pack_xyz, dim3(((int)n + 255) / 256), dim3(256), 0, c->stream,
                     (const char*)d_xyz, stride_bytes, (int)n, dst);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

int build_grid(ag2_ctx* c) {
  const int n = (int)c->n;
  DevStats* st = c->d_stats.as<DevStats>();
  hipLaunchKernelGGL(k_init_stats, dim3(1), dim3(64), 0, c->stream, st);
  c->n_valid = 0;
  c->grid = GridDesc{};
  if (n == 0) return 0;
  const float4* xyz = c->d_xyz_in.as<float4>();
  float bmin[3], bmax[3];
  if (c->bounds_known) {
    // the front end already knows the extent of what it produced (all points finite): no bounds
    // pass, no host round trip
    c->bounds_known = false;
    for (int a = 0; a < 3; a++) {
      bmin[a] = c->known_min[a];
      bmax[a] = c->known_max[a];
    }
    c->n_valid = (size_t)n;
  } else {
    const int nb = std::min((n + 255) / 256, 128);  // 7 contended atomics per workgroup: keep them few
    hipLaunchKernelGGL(k_bounds, dim3(nb), dim3(256), 0, c->stream, xyz, n, st);
    DevStats hs;
    AG2_HIP(c, hipMemcpyAsync(pin_small(c), st, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    __builtin_memcpy(&hs, pin_small(c), sizeof(hs));
    c->n_valid = hs.bounds[6];
    for (int a = 0; a < 3; a++) {
      bmin[a] = ord2f((int)hs.bounds[a]);
      bmax[a] = ord2f((int)hs.bounds[3 + a]);
    }
  }
  if (c->n_valid == 0) return 0;
  GridDesc g{};
  g.inv = 1.0f / (float)c->p.grid_cell;
  long long ncells = 1;
  for (int a = 0; a < 3; a++) {
    // a spatial tile bins its points from the origin of the WHOLE cloud (ag2_set_grid_origin)
    if (c->origin_set && bmin[a] < c->origin[a])
      return set_err(c, AG2_ERR_ARG, "a point lies below the grid origin given to ag2_set_grid_origin");
    g.o[a] = c->origin_set ? c->origin[a] : bmin[a];
    const float mx = bmax[a];
    g.dims[a] = (int)floorf((mx - g.o[a]) * g.inv) + 1;
    ncells *= g.dims[a];
  }
  c->min_z = g.o[2];  // pcl::getMinMax3D of the (whole) cloud, grasp_detector.cpp:152-153
  if (ncells > (1ll << 30))
    return set_err(c, AG2_ERR_CAPACITY, "search grid has more than 2^30 cells; raise grid_cell");
  g.ncells = (int)ncells;
  g.n_valid = (int)c->n_valid;
  c->grid = g;
  AG2_HIP(c, c->d_key.reserve((size_t)n * 4));
  // cell_start and the per-cell cursors share one buffer so that ONE fill (a multiple of 16 bytes:
  // no tail kernel) clears both
  const size_t cell_words = (((size_t)ncells + 1) + 3) & ~size_t(3);
  AG2_HIP(c, c->d_cell.reserve(2 * cell_words * 4));
  AG2_HIP(c, c->d_perm.reserve((size_t)n * 4));
  AG2_HIP(c, c->d_sorted.reserve((size_t)n * 16));
  AG2_HIP(c, c->d_nrm.reserve((size_t)n * 16));
  unsigned* cell = c->d_cell.as<unsigned>();
  unsigned* fill = cell + cell_words;
  AG2_HIP(c, hipMemsetAsync(cell, 0, 2 * cell_words * 4, c->stream));
  const int g256 = (n + 255) / 256;
  hipLaunchKernelGGL(k_cell_count, dim3(g256), dim3(256), 0, c->stream, xyz, n, g,
                     c->d_key.as<int>(), cell);
  const int rc = scan_exclusive_u32(c, cell, (int)ncells + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scatter, dim3(g256), dim3(256), 0, c->stream, c->d_key.as<int>(), n, cell,
                     fill, c->d_perm.as<int>());
  hipLaunchKernelGGL(k_cell_sort, dim3(((int)ncells + 255) / 256), dim3(256), 0, c->stream, cell,
                     (int)ncells, c->d_perm.as<int>());
  hipLaunchKernelGGL(k_gather4, dim3(((int)c->n_valid + 255) / 256), dim3(256), 0, c->stream, xyz,
                     c->d_perm.as<int>(), (int)c->n_valid, c->d_sorted.as<float4>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
