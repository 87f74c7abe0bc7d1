// k_sweep_common.h -- shared by the two hand-sweep kernels (k_sweep.hip: one workgroup per sample;
// k_sweep_wave.hip: one wave per sample): launch arguments, the stencil query, the row culling.
#pragma once
#include "ag2_internal.h"

namespace ag2 {

struct QueryRange {
  int lo[3], hi[3];
  bool empty;
};
__device__ __forceinline__ QueryRange query_range(const GridDesc& g, float qx, float qy, float qz,
                                                  float rq) {
  QueryRange r;
  const float qq[3] = {qx, qy, qz};
  r.empty = (g.n_valid == 0);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    r.lo[a] = max(cell_of(qq[a] - rq, g.o[a], g.inv), 0);
    r.hi[a] = min(cell_of(qq[a] + rq, g.o[a], g.inv), g.dims[a] - 1);
    if (r.lo[a] > r.hi[a]) r.empty = true;
  }
  return r;
}

struct SweepArgs {
  const float4* pts;
  const float4* nrm;
  const unsigned* cell;
  GridDesc g;
  const GridDesc* gp;        // frame mode: g and min_z are read from here
  const FrameArgs* fa;       // frame mode: slot_base is read from here
  const HandConst* hc;
  const float4* sample_q;
  const double* frames;
  const int* frame_ok;
  int n_samples;
  int slot_base;
  ag2_hypothesis* table;     // [n_samples * R]
  long long* tab_off;        // arena offset per slot (-1: no list)
  unsigned char* tab_keep;   // prune flag per slot
  double* arena;             // 6 doubles per in-box point
  long long arena_cap;       // points
  int emit_lists;
  DevStats* st;
  int* overflow;             // stage 0 appends, stage 1 works through it (length: st->n_overflow)
  float* gscratch;           // global variant: 6 * gcap floats per block
  int gcap;
  int* gpos;                 // stage 0: gpos_cap positions per workgroup (lists longer than the LDS stage)
  int gpos_cap;              // ... the longest list stage 0 keeps; longer ones go to the long-list stage
  float min_z;
  int flags;                 // bit0: no row tightening (exact K2 accounting, diagnostic)
  unsigned long long* prof;  // optional per-phase cycle sums (AG2_SWEEP_PROF=1), else nullptr
  // split sweep: the first kernels stop at the gates and queue (sample, orientation) pairs for
  // k_sweep_orient, with the sample's cropped list in the list arena: one float4 per point, the
  // centred coordinates p - q (float, the value the crop computed) and the sorted position as bits
  float4* lists;
  long long list_cap;        // points
  struct SweepPair* pairs;   // [n_samples * R]
};

// A sample's cropped list in the list arena: up to four dense SEGMENTS in canonical order inside one
// reservation (the long-list stage writes it in one pass, every wave its own rows, into room reserved
// from the candidate count: the survivors of a wave's rows are not known before they are written).
// Entry j of the list sits at arena index list_off + j + shift of its segment; a list written after
// its length was known (first stage) is one segment: end = K, shift = 0.
struct ListSegs {
  int end[3];    // list index at which segments 1, 2, 3 begin
  int shift[3];  // arena offset - list index within segments 1, 2, 3 (segment 0: 0)
};
// (taken by VALUE, six scalars: handed a struct in a local variable the compiler kept it in scratch
// memory and turned the selection below into an indexed load from it)
__device__ __forceinline__ int list_slot6(int e0, int e1, int e2, int s0, int s1, int s2, int j) {
  int sh = 0;
  sh = (j >= e0) ? s0 : sh;
  sh = (j >= e1) ? s1 : sh;
  sh = (j >= e2) ? s2 : sh;
  return j + sh;
}

struct SweepPair {
  int t, oi;                 // sample (position in this run's list), orientation
  unsigned hand;             // the 10 hand bits of the orientation (finger_hand.cpp:313-325)
  int K;                     // length of the sample's cropped list
  long long list_off;        // its reservation in the list arena
  ListSegs segs;
  int pad[2];
};
static_assert(sizeof(SweepPair) == 56, "pair queue entry");

template <int NW>
struct Red {
  double d[2][NW][12];
  unsigned u[2][NW][4];
  int i[2][NW][2];
};

// Shrink the x-cell range [cxa, cxb] of stencil row (cy, cz) to what the sphere |p - q| < r and the
// crop slab |curv . (p - q)| < hand_height can reach.  Purely conservative (mg = 0.3 mm of slack on
// every bound, three orders above the float rounding of these few operations): the exact per-point
// tests still decide, so the surviving set and its order are unchanged.  An unreachable row comes back
// with cxb < cxa.
__device__ __forceinline__ void tighten_row(const GridDesc& G, const HandConst& hc, const float4& q,
                                            const double (&F)[3][3], double hh, int cy, int cz, int& cxa,
                                            int& cxb) {
  const float mg = 3.0e-4f;
  const float h = 1.0f / G.inv;
  const float dyl = (G.o[1] + (float)cy * h) - q.y - mg, dyh = dyl + h + 2.0f * mg;
  const float dzl = (G.o[2] + (float)cz * h) - q.z - mg, dzh = dzl + h + 2.0f * mg;
  const float dym = dyl > 0.f ? dyl : (dyh < 0.f ? -dyh : 0.f);
  const float dzm = dzl > 0.f ? dzl : (dzh < 0.f ? -dzh : 0.f);
  const float rm = hc.rq_hands + mg;
  const float rho2 = rm * rm - dym * dym - dzm * dzm;
  bool empty = !(rho2 > 0.f);
  float xa = 0.f, xb = 0.f;
  if (!empty) {
    const float rho = __builtin_sqrtf(rho2);
    xa = -rho;
    xb = rho;
    const float cxn = (float)F[0][2], cyn = (float)F[1][2], czn = (float)F[2][2];
    const float hm = (float)hh + mg;
    const float slo = cyn * (cyn >= 0.f ? dyl : dyh) + czn * (czn >= 0.f ? dzl : dzh);
    const float shi = cyn * (cyn >= 0.f ? dyh : dyl) + czn * (czn >= 0.f ? dzh : dzl);
    const float ulo = -hm - shi - mg, uhi = hm - slo + mg;  // cxn * dx must lie in (ulo, uhi)
    const float acx = __builtin_fabsf(cxn);
    if (uhi < -acx * rho || ulo > acx * rho) {
      empty = true;
    } else if (acx > 1.0e-4f) {
      const float a = ulo / cxn, b2 = uhi / cxn;
      const float sa = (a < b2 ? a : b2) - mg, sb = (a < b2 ? b2 : a) + mg;
      xa = sa > xa ? sa : xa;
      xb = sb < xb ? sb : xb;
      if (xa > xb) empty = true;
    }
  }
  if (empty) {
    cxb = cxa - 1;
  } else {
    cxa = max(cxa, cell_of(q.x + xa - mg, G.o[0], G.inv));
    cxb = min(cxb, cell_of(q.x + xb + mg, G.o[0], G.inv));
  }
}

// k_sweep_orient.hip
int launch_sweep_orient(ag2_ctx* c, const SweepArgs& A, size_t n_slots);

}  // namespace ag2
