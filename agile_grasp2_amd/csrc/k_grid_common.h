// k_grid_common.h -- the search grid's description derived ON THE DEVICE (frame mode), shared by
// k_grid.hip (from the extent partials of k_bounds) and k_preprocess.hip (from the voxel lattice the
// front end just emitted).  Same float expressions as build_grid uses on the host, so the same cells.
#pragma once
#include "ag2_internal.h"

namespace ag2 {

struct GridFromParts {
  const int* part;        // extent partials of k_bounds (one 32-B record per workgroup), or nullptr
  const GridDesc* ready;  // a description already in device memory (the front end wrote it), or nullptr
  int nb;
  float inv;
  int origin_set;
  float org[3];
  int cap_cells;
  GridDesc* out;
  unsigned* done_flag;    // page-locked word (or nullptr): the kernel that derives the grid writes done_seq there first --
  unsigned done_seq;      // it runs behind k_bounds, so the host polling the word finds the extent partials complete
};

// ncells <= 0 tells every consumer "no grid": -1 more cells than the captured table holds, -2 a point
// below the origin given to ag2_set_grid_origin, 0 no finite point.
__device__ __forceinline__ GridDesc grid_from_extent(const float bmin[3], const float bmax[3], int cnt,
                                                     const GridFromParts& f) {
  GridDesc g{};
  g.inv = f.inv;
  if (cnt > 0) {
    long long ncells = 1;
    bool below = false;
    for (int a = 0; a < 3; a++) {
      if (f.origin_set && bmin[a] < f.org[a]) below = true;
      g.o[a] = f.origin_set ? f.org[a] : bmin[a];
      g.dims[a] = (int)__builtin_floorf((bmax[a] - g.o[a]) * g.inv) + 1;
      ncells *= g.dims[a];
      if (ncells > (1ll << 30)) ncells = (1ll << 30) + 1;
    }
    g.min_z = g.o[2];  // pcl::getMinMax3D of the (whole) cloud, grasp_detector.cpp:152-153
    g.n_valid = cnt;
    g.ncells = below ? -2 : (ncells > (long long)f.cap_cells ? -1 : (int)ncells);
    if (g.ncells <= 0) {
      g.n_valid = 0;
      g.dims[0] = g.dims[1] = g.dims[2] = 0;
    }
  }
  return g;
}

// Executed by the first wave of EVERY workgroup of k_cell_count (4 KB of L2-resident partials: cheaper
// than a launch of its own); workgroup 0 leaves the result in memory for the kernels that follow.
__device__ __forceinline__ GridDesc grid_from_partials(const GridFromParts& f) {
  const int lane = lane_id();
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int cnt = 0;
  for (int b = lane; b < f.nb; b += 64) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = min(mn[a], f.part[b * 8 + a]);
      mx[a] = max(mx[a], f.part[b * 8 + 3 + a]);
    }
    cnt += f.part[b * 8 + 6];
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
  const float bmin[3] = {ord2f(mn[0]), ord2f(mn[1]), ord2f(mn[2])};
  const float bmax[3] = {ord2f(mx[0]), ord2f(mx[1]), ord2f(mx[2])};
  return grid_from_extent(bmin, bmax, cnt, f);
}

}  // namespace ag2
