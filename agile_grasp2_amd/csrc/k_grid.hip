// k_grid.hip -- K0: uniform search grid over the cloud.
//
// Replaces pcl::KdTreeFLANN::setInputCloud (src/agile_grasp2/hand_search.cpp:11-12) and the kd-tree
// inside pcl::NormalEstimationOMP (:87).  Points are counting-sorted by cell key (x fastest) with
// ties broken by original index, so a run of x-adjacent cells is one contiguous, coalesced span of
// float4 points and "ascending sorted position" is the canonical neighbour order the oracle uses.
#include "ag2_internal.h"
#include "k_grid_common.h"

#ifndef AG2_SCAN_PER
#define AG2_SCAN_PER 8
#endif

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

// Extent of the cloud (finite points only), optionally fused with the pack above: every workgroup
// writes ONE partial record (min xyz, max xyz as order-preserving ints, count) that the host reduces
// after the read-back it needs anyway -- no atomics, nothing to initialise.  Block 0 also clears the
// device statistics of the previous cloud.
static_assert(kBoundsBlocks * 32 <= (int)kPinDoneFlag, "extent partials must fit the small read-back area in front of the flags");
// n_pad > n (frame mode): xyz[n .. n_pad) is filled with non-finite points, which no later stage
// ever sees as a point -- the launches of a captured frame run over the fixed maximum n_pad.
template <bool PACK>
__global__ void __launch_bounds__(256) k_bounds(const char* __restrict__ src, size_t stride,
                                                float4* __restrict__ xyz, int n, int n_pad,
                                                int* __restrict__ part, DevStats* st,
                                                int* __restrict__ part_dev, uint4* __restrict__ zero16, int n_zero16) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *st = DevStats{};
  // (the grid build's counters and scan words, cleared here instead of by a fill of their own between this pass
  // and the counting pass: see pack_device_xyz)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_zero16; i += gridDim.x * blockDim.x)
    zero16[i] = make_uint4(0u, 0u, 0u, 0u);
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
      xyz[i] = p;
    } else {
      p = xyz[i];
    }
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
    if (part_dev) part_dev[blockIdx.x * 8 + k] = v;  // (a copy in device memory for the k_cell_count queued behind)
  }
}

// key[i] = (cell key or -1, arrival rank inside the cell): the rank the counting atomic hands out
// places the point in the scatter below without a second round of atomics.
__global__ void __launch_bounds__(256) k_cell_count(const float4* __restrict__ xyz, int n, GridDesc g_arg,
                                                    GridFromParts fp, int2* __restrict__ key,
                                                    unsigned* __restrict__ cell) {
  __shared__ GridDesc g_sh;
  // k_bounds is complete: tell the polling host.  Sound because (a) the partials and the flag live in a block
  // allocated hipHostMallocCoherent (kPinFlags: fine-grained, not held in any device cache -- a store that has
  // been acknowledged is in host memory) and (b) this kernel starts only after every store of k_bounds has been
  // acknowledged (kernels of one stream run in order, a kernel's end waits for its memory operations).  The
  // fence orders this thread's flag store behind that point at system scope.
  if (fp.done_flag && blockIdx.x == 0 && threadIdx.x == 0) {
    __threadfence_system();
    *reinterpret_cast<volatile unsigned*>(fp.done_flag) = fp.done_seq;
  }
  if (fp.part) {  // frame mode (uniform): the description from the extent partials of k_bounds
    if (wave_id() == 0) {
      const GridDesc gd = grid_from_partials(fp);
      if (lane_id() == 0) {
        g_sh = gd;
        if (blockIdx.x == 0) *fp.out = gd;
      }
    }
    __syncthreads();
  }
  // (fp.ready: frame mode behind the GPU front end, which left the description in memory)
  const GridDesc g = fp.part ? g_sh : (fp.ready ? *fp.ready : g_arg);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float4 p = (i < n) ? xyz[i] : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
  int k = -1, r = 0;
  if (i < n && g.ncells > 0 && finite3(p.x, p.y, p.z)) {
    const int cx = cell_of(p.x, g.o[0], g.inv), cy = cell_of(p.y, g.o[1], g.inv),
              cz = cell_of(p.z, g.o[2], g.inv);
    k = (cz * g.dims[1] + cy) * g.dims[0] + cx;
  }
  // Consecutive points of a sensor cloud fall into a handful of cells, so the 64 lanes of a wave
  // would queue up on the same few counters: the lanes that share a cell send ONE atomic (their
  // count) and split the range it returns among themselves by lane order.  The groups are formed
  // first, in registers (one ballot per distinct cell of the wave), and then all the group leaders'
  // atomics go out together: one round trip per wave, however its points are scattered -- a
  // returning atomic inside the grouping loop made a wave of 64 distinct cells wait for nine.
  const int lane = lane_id();
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int lead = lane;
  unsigned within = 0, gsize = 1;
  for (unsigned long long todo = __ballot(k >= 0); todo;) {
    const int l0 = __ffsll((long long)todo) - 1;
    const int k0 = __builtin_amdgcn_readlane(k, l0);
    const unsigned long long same = __ballot(k == k0) & todo;
    if (k == k0) {
      lead = l0;
      within = (unsigned)__popcll(same & below);
      gsize = (unsigned)__popcll(same);
    }
    todo &= ~same;
  }
  unsigned base = 0;
  if (k >= 0 && lane == lead) base = atomicAdd(&cell[k], gsize);
  base = (unsigned)__shfl((int)base, lead, 64);
  r = (int)(base + within);
  if (i < n) key[i] = make_int2(k, r);
}

// ---- exclusive scan of uint32, one pass (kScanTile elements per tile) ----------------------------
// Chained scan with look-back: a workgroup draws its tile from a ticket counter (so every
// predecessor it waits for is already running, whatever the dispatch order), scans the tile,
// publishes the tile total, and wave 0 looks back over the predecessors' records -- 64 at a time --
// until it meets one that already holds an inclusive prefix.  A record is one 64-bit word
// (state << 32 | value; state 1 = tile total, 2 = inclusive prefix), read and written with relaxed
// device-scope atomics: state and value travel together, so no ordering against other memory is
// needed.  ctl[0] = ticket, ctl[2 + 2 * t ...] = record of tile t; all zero before the launch.
// POPC: the input is the population count of src[i] (i < n - 1; element n - 1 counts as zero, so
// data[n - 1] receives the total) -- the voxel front end's rank-per-bitmap-word scan, without a pass that
// writes the counts first.
constexpr int kScanPer = AG2_SCAN_PER;          // elements per thread
constexpr int kScanTile = 256 * kScanPer;       // ... per tile
template <bool POPC>
__global__ void __launch_bounds__(256) k_scan_chained(unsigned* __restrict__ data, int n,
                                                      unsigned* __restrict__ ctl,
                                                      const unsigned* __restrict__ src) {
  __shared__ unsigned wsum[4];
  __shared__ unsigned s_tile, s_prefix;
  unsigned long long* rec = reinterpret_cast<unsigned long long*>(ctl + 2);
  if (threadIdx.x == 0) s_tile = atomicAdd(&ctl[0], 1u);
  __syncthreads();
  const int tile = (int)s_tile;
  const int base = tile * kScanTile + threadIdx.x * kScanPer;
  unsigned v[kScanPer];
  unsigned tot = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; k++) {
    if (POPC) v[k] = (base + k < n - 1) ? (unsigned)__popc(src[base + k]) : 0u;
    else v[k] = (base + k < n) ? data[base + k] : 0u;
    tot += v[k];
  }
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
  if (wave_id() == 0) {
    const unsigned total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (lane_id() == 0)
      __hip_atomic_store(&rec[tile], ((unsigned long long)(tile == 0 ? 2u : 1u) << 32) | total,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned prefix = 0;
    int jb = tile - 1;
    while (jb >= 0) {
      const int j = jb - lane_id();
      // tiles before the first count as an inclusive prefix of zero
      const unsigned long long r = (j >= 0) ? __hip_atomic_load(&rec[j], __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT)
                                            : (2ull << 32);
      const unsigned state = (unsigned)(r >> 32);
      const unsigned long long m2 = __ballot(state == 2u), m0 = __ballot(state == 0u);
      const int first2 = m2 ? (__ffsll((long long)m2) - 1) : 64;
      const unsigned long long need = (first2 >= 63) ? ~0ull : ((2ull << first2) - 1ull);
      if (m0 & need) {  // a predecessor in the window has not published yet: poll again
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      unsigned val = (lane_id() <= first2) ? (unsigned)r : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) val += (unsigned)__shfl_xor((int)val, o, 64);
      prefix += val;
      if (first2 < 64) break;
      jb -= 64;
    }
    if (lane_id() == 0) {
      if (tile > 0)
        __hip_atomic_store(&rec[tile], (2ull << 32) | (unsigned long long)(prefix + total),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
    }
  }
  __syncthreads();
  unsigned run = s_prefix + woff + inc - tot;
#pragma unroll
  for (int k = 0; k < kScanPer; k++) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
}

size_t scan_ctl_words(int n) { return 2 + 2 * ((size_t)n / kScanTile + 1) + 2; }

// zeroed_ctl: scan_ctl_words(n) zeroed words the caller cleared together with its own buffers
// (saves the fill); nullptr = the scan clears its own.
int scan_exclusive_u32(ag2_ctx* c, unsigned* d, int n, unsigned* zeroed_ctl) {
  if (n <= 0) return 0;
  const int nb = (n + kScanTile - 1) / kScanTile;
  if (!zeroed_ctl) {
    const size_t words = scan_ctl_words(n);
    AG2_HIP(c, c->d_scan.reserve(words * sizeof(unsigned)));
    AG2_HIP(c, hipMemsetAsync(c->d_scan.p, 0, words * sizeof(unsigned), c->stream));
    zeroed_ctl = c->d_scan.as<unsigned>();
  }
  hipLaunchKernelGGL(k_scan_chained<false>, dim3(nb), dim3(256), 0, c->stream, d, n, zeroed_ctl,
                     (const unsigned*)nullptr);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// rank[w] = number of set bits in bitmap[0 .. w), w = 0 .. words (rank[words] = all of them)
int scan_popc_u32(ag2_ctx* c, const unsigned* bitmap, int words, unsigned* rank, unsigned* zeroed_ctl) {
  const int n = words + 1;
  const int nb = (n + kScanTile - 1) / kScanTile;
  if (!zeroed_ctl) {
    const size_t w = scan_ctl_words(n);
    AG2_HIP(c, c->d_scan.reserve(w * sizeof(unsigned)));
    AG2_HIP(c, hipMemsetAsync(c->d_scan.p, 0, w * sizeof(unsigned), c->stream));
    zeroed_ctl = c->d_scan.as<unsigned>();
  }
  hipLaunchKernelGGL(k_scan_chained<true>, dim3(nb), dim3(256), 0, c->stream, rank, n, zeroed_ctl, bitmap);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

__global__ void k_scatter(const int2* __restrict__ key, int n, const unsigned* __restrict__ cell,
                          int* __restrict__ perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int2 k = key[i];
  if (k.x < 0) return;
  perm[cell[k.x] + (unsigned)k.y] = i;
}

// Within a cell the arrival order is whatever the atomics produced; restore ascending original
// index so the layout is deterministic ((key, index) order).
// A workgroup owns kSortCells consecutive cells = one contiguous span of perm.  The span is worked through
// in batches of whole cells that fit the LDS stage (usually one batch: cells of a 3 mm voxel cloud
// hold ~10-30 points; un-voxelised clouds with hundreds of points per cell take several), each batch
// rank-sorted with one thread per ELEMENT: every element counts the smaller indices in its own cell
// (independent compares, no dependent insertion chains) and is written to (cell start + rank).
// Only a single cell larger than the stage is sorted in place in global memory by one thread.
// The sorted float4 cloud is written from here as well (sorted[pos] = xyz[perm[pos]]): the final
// position of every element is known at this point, so no separate gather pass re-reads perm.
constexpr int kSortCells = 64;       // cells per workgroup (a 300 k-point voxel cloud has ~30 k cells: at 256 cells
                                     // per workgroup half the CUs had none)
constexpr int kSortStage = 6144;     // ints of LDS: values [0, n) + packed cell (start | length << 16) [n, 2n)
constexpr int kSortBatch = kSortStage / 2;
static_assert(kSortBatch <= 65535, "cell start and length are packed into 16 bits each");
__global__ void __launch_bounds__(256) k_cell_sort(const unsigned* __restrict__ cell, int ncells,
                                                   int* __restrict__ perm,
                                                   const float4* __restrict__ xyz,
                                                   float4* __restrict__ sorted) {
  __shared__ int stage[kSortStage];
  __shared__ int cs[kSortCells + 1];  // sorted position at which each of the workgroup's cells starts
  const int c0 = blockIdx.x * kSortCells, tid = threadIdx.x;
  if (tid <= kSortCells) cs[tid] = (int)cell[min(c0 + tid, ncells)];
  // one all-NaN point behind the sorted cloud: what k_normals reads where a load step reaches past
  // the end of its span (cell[ncells] = number of valid points)
  if (blockIdx.x == 0 && tid == 0) {
    const float nanv = __builtin_nanf("");
    sorted[cell[ncells]] = make_float4(nanv, nanv, nanv, nanv);
  }
  __syncthreads();
  int cb = 0;  // first cell of the batch (everything below is uniform across the workgroup)
  while (cb < kSortCells && cs[kSortCells] > cs[cb]) {
    const int lo = cs[cb];
    // largest ce in [cb, kSortCells] with cs[ce] - lo <= kSortBatch (cs is non-decreasing)
    int a = cb, b = kSortCells;
    while (a < b) {
      const int m = (a + b + 1) >> 1;
      if (cs[m] - lo <= kSortBatch) a = m; else b = m - 1;
    }
    int ce = a;
    if (ce == cb) {  // a single cell larger than the stage: in place, one thread
      ce = cb + 1;
      const int e = cs[ce];
      if (tid == 0) {
        for (int i = lo + 1; i < e; i++) {
          const int v = perm[i];
          int j = i - 1;
          while (j >= lo) {
            const int u = perm[j];
            if (!(u > v)) break;
            perm[j + 1] = u;
            j--;
          }
          perm[j + 1] = v;
        }
      }
      __syncthreads();
      for (int i = lo + tid; i < e; i += 256) sorted[i] = xyz[perm[i]];
    } else {
      const int n = cs[ce] - lo;
      for (int i = tid; i < n; i += 256) stage[i] = perm[lo + i];
      if (tid >= cb && tid < ce) {  // the owner of a cell labels its elements
        const int st = cs[tid] - lo, len = cs[tid + 1] - cs[tid];
        for (int k = 0; k < len; k++) stage[n + st + k] = st | (len << 16);
      }
      __syncthreads();
      for (int i = tid; i < n; i += 256) {
        const int v = stage[i];
        const int be = stage[n + i];
        const int st = be & 0xFFFF, len = (int)((unsigned)be >> 16);
        // (eight independent reads per round trip: a cell of an un-voxelised cloud holds a hundred points and
        // more, and one dependent LDS read per comparison left this kernel waiting for latency)
        int rank = 0;
        const int* cellv = stage + st;
        int k = 0;
        for (; k + 8 <= len; k += 8) {
          int u[8];
#pragma unroll
          for (int q = 0; q < 8; q++) u[q] = cellv[k + q];
#pragma unroll
          for (int q = 0; q < 8; q++) rank += (u[q] < v) ? 1 : 0;
        }
        for (; k < len; k++) rank += (cellv[k] < v) ? 1 : 0;
        perm[lo + st + rank] = v;  // indices within a cell are distinct: ranks are a permutation
        sorted[lo + st + rank] = xyz[v];
      }
      __syncthreads();  // the stage is reused by the next batch
    }
    cb = ce;
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

// with_bounds: the pack also produces the extent partials build_grid needs (c->bounds_blocks > 0
// tells build_grid they are there), one pass over the source instead of two.
int pack_device_xyz(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes, float4* dst,
                    bool with_bounds, size_t n_pad) {
  c->bounds_blocks = 0;
  if (n == 0) return 0;
  if (n_pad < n) n_pad = n;  // (0 = no padding)
  if (with_bounds) {
    const int nb = std::min(((int)n_pad + 255) / 256, kBoundsBlocks);  // (frame mode: a function of n_pad only)
    AG2_HIP(c, c->d_bounds.reserve((size_t)kBoundsBlocks * 8 * 4));
    // Step-by-step use: the host reduces the partials, so the kernel writes them where the host reads
    // them (page-locked memory through its device view): no copy operation in front of the read-back.
    // Frame mode reduces them on the device (k_cell_count).
    int* part = c->d_bounds.as<int>();
    c->bounds_in_pin = false;
    if (!c->fm_on) {
      const int rc = pin_reserve(c, 0);
      if (rc) return rc;
      if (c->h_pin_dev) {
        part = reinterpret_cast<int*>(pin_small_dev(c));
        c->bounds_in_pin = true;
      }
    }
    const bool ahead = !c->fm_on && c->bounds_in_pin && c->cell_bytes_last && c->d_cell.p && n_pad == n;
    const size_t guess = ahead ? std::min(c->d_cell.bytes & ~size_t(15), (c->cell_bytes_last + c->cell_bytes_last / 4 + 15) & ~size_t(15)) : 0;
    hipLaunchKernelGGL(k_bounds<true>, dim3(nb), dim3(256), 0, c->stream, (const char*)d_xyz,
                       stride_bytes, dst, (int)n, (int)n_pad, part, c->d_stats.as<DevStats>(),
                       ahead ? c->d_bounds.as<int>() : (int*)nullptr,
                       ahead ? c->d_cell.as<uint4>() : (uint4*)nullptr, (int)(guess / 16));
    c->bounds_blocks = nb;
    // The grid build clears its cell counters and scan words before it counts -- a fill whose size it learns
    // from the extent.  Done HERE, by the extent pass itself and at the size of the previous cloud's grid plus a
    // quarter, it costs no launch and nothing waits for it: build_grid then finds the words clear (a cloud whose
    // grid outgrows the guess gets its fill as before).
    c->cell_bytes_prezeroed = 0;
    c->cell_count_ahead_cap = 0;
    c->bounds_flag_armed = false;
    // (same box, back to back: 0.6686 against 0.6714 ms per step)
    if (ahead) {
      c->cell_bytes_prezeroed = guess;
      c->cell_prezeroed_at = c->d_cell.p;
      // ... and the counting pass itself: it derives the grid from the extent partials on the device (the
      // expressions build_grid uses: the same cells), so it need not wait for the host either -- the GPU counts
      // while the host reads the extent, and build_grid goes on with the scan.  Valid for grids whose counters and
      // scan words lie inside what was just cleared; a larger grid is left alone (ncells = -1: nothing counted)
      // and takes the ordinary path.
      size_t cap = guess / 4;
      while (cap > 0 && ((((cap + 1) + 3) & ~size_t(3)) + ((scan_ctl_words((int)cap + 1) + 3) & ~size_t(3))) * 4 > guess)
        cap -= std::min<size_t>(cap, 64);
      // (same box, four pairs back to back: 0.6735 against 0.6770 ms per step)
      if (cap > 0 && c->d_key.reserve(n * 8) == hipSuccess && c->d_griddesc.reserve(sizeof(GridDesc)) == hipSuccess) {
        GridFromParts fp{};
        fp.part = c->d_bounds.as<int>();
        fp.nb = nb;
        fp.inv = 1.0f / (float)c->p.grid_cell;
        fp.origin_set = c->origin_set ? 1 : 0;
        for (int a = 0; a < 3; a++) fp.org[a] = c->origin[a];
        fp.cap_cells = (int)std::min<size_t>(cap, (size_t)1 << 30);
        fp.out = c->d_griddesc.as<GridDesc>();
        // (the host polls this word instead of waiting for the stream: written by the first thread of the kernel
        // BEHIND the extent pass, it says the partials are complete -- no fence per workgroup inside k_bounds, which
        // on this GPU is a write-back of the L2 each)
        if (++c->bounds_seq == 0u) c->bounds_seq = 1u;  // (zero is what the flag starts from)
        fp.done_flag = reinterpret_cast<unsigned*>(pin_small_dev(c) + kPinBoundsFlag);
        fp.done_seq = c->bounds_seq;
        c->bounds_flag_armed = true;
        hipLaunchKernelGGL(k_cell_count, dim3(((int)n + 255) / 256), dim3(256), 0, c->stream, dst, (int)n, GridDesc{}, fp,
                           c->d_key.as<int2>(), c->d_cell.as<unsigned>());
        c->cell_count_ahead_cap = fp.cap_cells;
      }
    }
  } else {
    hipLaunchKernelGGL(k_pack_xyz, dim3(((int)n + 255) / 256), dim3(256), 0, c->stream,
                       (const char*)d_xyz, stride_bytes, (int)n, dst);
  }
  AG2_HIP(c, hipGetLastError());
  return 0;
}

int launch_grid_frame(ag2_ctx* c, unsigned* cell, unsigned* zeroed_ctl) {
  c->cell_bytes_prezeroed = 0;  // (this path writes the cell array: nothing cleared ahead survives it)
  const int n = (int)c->fm_n_max, cap = (int)c->fm_cap_cells;
  GridFromParts fp{};
  if (c->fm_grid_ready) {  // the GPU front end left the description in d_griddesc (k_vox_emit_frame)
    fp.ready = c->d_griddesc.as<GridDesc>();
  } else {
    fp.part = c->d_bounds.as<int>();
  }
  fp.nb = std::min((n + 255) / 256, kBoundsBlocks);  // as pack_device_xyz launched k_bounds for n_pad = n
  fp.inv = 1.0f / (float)c->p.grid_cell;
  fp.origin_set = c->origin_set ? 1 : 0;
  for (int a = 0; a < 3; a++) fp.org[a] = c->origin[a];
  fp.cap_cells = cap;
  fp.out = c->d_griddesc.as<GridDesc>();
  const int g256 = (n + 255) / 256;
  hipLaunchKernelGGL(k_cell_count, dim3(g256), dim3(256), 0, c->stream, c->d_xyz_in.as<float4>(), n,
                     GridDesc{}, fp, c->d_key.as<int2>(), cell);
  // the scan and the sort run over the whole table: cells beyond the frame's grid hold no point
  const int rc = scan_exclusive_u32(c, cell, cap + 1, zeroed_ctl);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scatter, dim3(g256), dim3(256), 0, c->stream, c->d_key.as<int2>(), n, cell,
                     c->d_perm.as<int>());
  hipLaunchKernelGGL(k_cell_sort, dim3((cap + kSortCells - 1) / kSortCells), dim3(256), 0, c->stream, cell, cap,
                     c->d_perm.as<int>(), c->d_xyz_in.as<float4>(), c->d_sorted.as<float4>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

int build_grid(ag2_ctx* c) {
  const size_t prezeroed = c->cell_bytes_prezeroed;  // (whatever this call does, the next one starts without it)
  const int counted_ahead_cap = c->cell_count_ahead_cap;
  c->cell_bytes_prezeroed = 0;
  c->cell_count_ahead_cap = 0;
  const int n = (int)c->n;
  DevStats* st = c->d_stats.as<DevStats>();
  c->n_valid = 0;
  c->grid = GridDesc{};
  const int packed_blocks = c->bounds_blocks;
  c->bounds_blocks = 0;
  if (n == 0) {
    hipLaunchKernelGGL(k_init_stats, dim3(1), dim3(64), 0, c->stream, st);
    return 0;
  }
  const float4* xyz = c->d_xyz_in.as<float4>();
  float bmin[3], bmax[3];
  if (c->bounds_known) {
    // the front end already knows the extent of what it produced (all points finite): no bounds
    // pass, no host round trip
    hipLaunchKernelGGL(k_init_stats, dim3(1), dim3(64), 0, c->stream, st);
    c->bounds_known = false;
    for (int a = 0; a < 3; a++) {
      bmin[a] = c->known_min[a];
      bmax[a] = c->known_max[a];
    }
    c->n_valid = (size_t)n;
  } else {
    int nb = packed_blocks;
    if (nb == 0) {  // the cloud did not come through the fused pack: extent pass on its own
      nb = std::min((n + 255) / 256, kBoundsBlocks);
      AG2_HIP(c, c->d_bounds.reserve((size_t)kBoundsBlocks * 8 * 4));
      hipLaunchKernelGGL(k_bounds<false>, dim3(nb), dim3(256), 0, c->stream, (const char*)nullptr,
                         (size_t)0, c->d_xyz_in.as<float4>(), n, n, c->d_bounds.as<int>(), st, (int*)nullptr,
                         (uint4*)nullptr, 0);
    }
    if (packed_blocks && c->bounds_in_pin) {  // the pack kernel wrote them into pin_small itself
      if (c->bounds_flag_armed) {               // ... and the kernel queued behind it raises the flag
        const int rcw = wait_flag(c, kPinBoundsFlag, c->bounds_seq);
        if (rcw) return rcw;
      } else {
        AG2_HIP(c, hipStreamSynchronize(c->stream));
      }
      c->bounds_flag_armed = false;
    } else {
      AG2_HIP(c, hipMemcpyAsync(pin_small(c), c->d_bounds.p, (size_t)nb * 32, hipMemcpyDeviceToHost,
                                c->stream));
      AG2_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->bounds_in_pin = false;
    const int* part = (const int*)pin_small(c);
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    long long cnt = 0;
    for (int b = 0; b < nb; b++) {
      for (int a = 0; a < 3; a++) {
        mn[a] = std::min(mn[a], part[b * 8 + a]);
        mx[a] = std::max(mx[a], part[b * 8 + 3 + a]);
      }
      cnt += part[b * 8 + 6];
    }
    c->n_valid = (size_t)cnt;
    for (int a = 0; a < 3; a++) {
      bmin[a] = ord2f(mn[a]);
      bmax[a] = ord2f(mx[a]);
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
  g.min_z = c->min_z;
  if (ncells > (1ll << 30))
    return set_err(c, AG2_ERR_CAPACITY, "search grid has more than 2^30 cells; raise grid_cell");
  g.ncells = (int)ncells;
  g.n_valid = (int)c->n_valid;
  c->grid = g;
  AG2_HIP(c, c->d_key.reserve((size_t)n * 8));
  // the cell counters and the scan's control words share one buffer so that ONE fill (a multiple of
  // 16 bytes: no tail kernel) clears both
  const size_t cell_words = (((size_t)ncells + 1) + 3) & ~size_t(3);
  const size_t ctl_words = (scan_ctl_words((int)ncells + 1) + 3) & ~size_t(3);
  AG2_HIP(c, c->d_cell.reserve((cell_words + ctl_words) * 4));
  AG2_HIP(c, c->d_perm.reserve((size_t)n * 4));
  AG2_HIP(c, c->d_sorted.reserve(((size_t)n + 1) * 16));  // + the NaN point behind the cloud
  AG2_HIP(c, c->d_nrm.reserve((size_t)n * 16));
  unsigned* cell = c->d_cell.as<unsigned>();
  const size_t clear_bytes = (cell_words + ctl_words) * 4;
  const bool cleared_ahead = prezeroed >= clear_bytes && c->cell_prezeroed_at == (const void*)cell;
  // (counted ahead: pack_device_xyz queued k_cell_count with the grid derived on the device -- this very grid --
  // unless it has more cells than that launch was told to accept)
  const bool counted_ahead = cleared_ahead && counted_ahead_cap > 0 && ncells <= (long long)counted_ahead_cap;
  if (!cleared_ahead) AG2_HIP(c, hipMemsetAsync(cell, 0, clear_bytes, c->stream));
  c->cell_bytes_last = clear_bytes;
  const int g256 = (n + 255) / 256;
  if (!counted_ahead)
    hipLaunchKernelGGL(k_cell_count, dim3(g256), dim3(256), 0, c->stream, xyz, n, g,
                       GridFromParts{}, c->d_key.as<int2>(), cell);
  const int rc = scan_exclusive_u32(c, cell, (int)ncells + 1, cell + cell_words);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scatter, dim3(g256), dim3(256), 0, c->stream, c->d_key.as<int2>(), n, cell,
                     c->d_perm.as<int>());
  hipLaunchKernelGGL(k_cell_sort, dim3(((int)ncells + kSortCells - 1) / kSortCells), dim3(256), 0, c->stream, cell,
                     (int)ncells, c->d_perm.as<int>(), xyz, c->d_sorted.as<float4>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
