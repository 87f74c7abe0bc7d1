// k_sweep_wave.hip -- K3 the hand sweep, ONE WAVE PER SAMPLE.
//
// Same arithmetic as k_sweep.hip (HandSearch::evaluateHands hand_search.cpp:173-235, calculateHand
// :319-426, FingerHand finger_hand.cpp:17-214, :313-325, Antipodal::evaluateGrasp antipodal.cpp:8-84),
// different shape.  k_sweep gives a sample to a 256-thread workgroup: every phase boundary is a
// workgroup barrier and every reduction an LDS hop, ~14 per sample, and the four waves of a workgroup
// sit on four SIMDs that they share with other workgroups, so at every barrier three waves wait for
// the one whose SIMD was busiest.  Here a sample belongs to one 64-lane wave:
//   * no workgroup barrier anywhere -- phase boundaries are wave-synchronous, reductions are DPP;
//   * the cropped list is appended in canonical order in ONE pass (a single wave walks the stencil
//     rows in order, so ballot + popcount give the final offsets at once: no per-piece counts, no
//     scan, no second pass), and the closing-region list likewise;
//   * sixteen samples are in flight per CU instead of four, so the dependent-load chains of one
//     sample overlap the arithmetic of fifteen others.
// Per wave: 2 032 list positions in LDS (8 KB) + 2 KB that hold the piece table of the current 64-row
// round during the crop and the closing-region index list afterwards; list entries beyond the LDS
// part live in a per-wave global slice (L2-resident).  Lists longer than both go to the workgroup
// kernel's global-scratch stage through the overflow queue, exactly as before.
#include "ag2_internal.h"
#include "k_sweep_common.h"

namespace ag2 {

constexpr int kWvWaves = 4;        // independent waves per workgroup
constexpr int kWvWgPerCu = 4;
constexpr int kWvLds = 2032;       // list positions per wave kept in LDS
constexpr int kWvGcap = 12288;     // ... and in the wave's global slice
constexpr int kWvPieces = 256;     // piece table of one batch (pieces of <= 8 consecutive points)
constexpr int kWvBox = 1024;       // closing-region index list (u16), overlays the piece table
constexpr int kWvSliceBytes = kWvLds * 4 + 2048;
static_assert(kWvPieces * 5 <= 2048 && kWvBox * 2 <= 2048, "aux area");
static_assert(kWvLds + kWvGcap <= 65536, "in-box indices are 16 bits");
static_assert(kWvWgPerCu * (kWvWaves * kWvSliceBytes + 256) <= 163840, "LDS budget");

__device__ __forceinline__ void wave_mem_sync() {
  // orders this wave's LDS / global writes before its own later reads (other lanes' data)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
}

template <int RMAX>
__global__ void __launch_bounds__(kWvWaves * kWave, kWvWgPerCu) k_sweep_wave(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ double s_depths[kMaxDepths];
  const int lane = lane_id(), wid = wave_id();
  unsigned char* slice = smem_raw + (size_t)wid * kWvSliceBytes;
  int* POS = reinterpret_cast<int*>(slice);
  unsigned char* aux = slice + kWvLds * 4;
  int* pst = reinterpret_cast<int*>(aux);                       // crop: first point of a piece
  unsigned char* pln = aux + kWvPieces * 4;                     // crop: its length (1 .. 8)
  unsigned short* box16 = reinterpret_cast<unsigned short*>(aux);  // orientations: closing-region list
  int* gpos = A.gpos + ((size_t)blockIdx.x * kWvWaves + wid) * kWvGcap;

  const HandConst& hc = *A.hc;
  const GridDesc G = A.gp ? *A.gp : A.g;
  const float cloud_min_z = A.gp ? G.min_z : A.min_z;
  const int slot_base = A.fa ? (int)A.fa->slot_base : A.slot_base;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int R = hc.R;
  const double hh = hc.hand_height;
  const int n_depths = hc.n_depths;
  if (threadIdx.x < kMaxDepths) s_depths[threadIdx.x] = hc.depths[threadIdx.x];
  __syncthreads();  // the only workgroup barrier: once per kernel, before any wave diverges
  const double hand_depth = hc.hand_depth;
  const double slot_inv_step = hc.slot_inv_step, slot_step = hc.slot_step;
  const double hand_od = hc.hand_outer_diameter, finger_w = hc.finger_width;
  const double slot_ratio = hc.finger_width * hc.slot_inv_step;
  const int slot_span = hc.slot_span;
  const double slot_base0 = hc.fs[0], slot_base1 = hc.fs[10];
  const float r2_hands = hc.r2_hands;
  const bool tighten = (A.flags & 1) == 0;
  const double top0 = hc.init_bite, bottom0 = hc.init_bite - hc.hand_depth;
  // hand angles rounded to f32 for the classification pass (wave-uniform: scalar registers)
  float cosf_t[RMAX], sinf_t[RMAX];
#pragma unroll
  for (int i = 0; i < RMAX; i++) {
    cosf_t[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)hc.cos_t[i])));
    sinf_t[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)hc.sin_t[i])));
  }
  const int grp = lane >> 3, lg = lane & 7;  // 8 lanes share one piece of the crop

  auto pos_at = [&](int j) -> int { return (j < kWvLds) ? POS[j] : gpos[j - kWvLds]; };

  // per-wave statistics, flushed once
  unsigned long long acc_kcrop = 0, acc_k2 = 0, acc_p = 0;
  unsigned acc_hyp = 0, acc_maxp = 0;

  const int n_work = A.n_samples;
  const int grid_w = (int)gridDim.x * kWvWaves;
  int t = (int)blockIdx.x * kWvWaves + wid;
  while (t < n_work) {
    // the next item is requested now and used at the end of this one: the atomic is never waited for
    int nxt = 0;
    if (lane == 0) nxt = grid_w + (int)atomicAdd(&A.st->work_next[0], 1u);
    do {
      if (!A.frame_ok[t]) break;  // uniform
      const float4 q = A.sample_q[t];
      auto ldp = [&](int j, float& x, float& y, float& z) {  // cropped point j, centred on the sample
        const float4 p = A.pts[pos_at(j)];
        x = p.x - q.x;
        y = p.y - q.y;
        z = p.z - q.z;
      };
      const double* fr = A.frames + (size_t)t * 12;
      const double smp[3] = {fr[0], fr[1], fr[2]};
      // frame = [normal binormal curvature_axis] as columns, hand_search.cpp:325-326
      const double F[3][3] = {{fr[3], fr[6], fr[9]}, {fr[4], fr[7], fr[10]}, {fr[5], fr[8], fr[11]}};

      // ---- crop to the +-hand_height slab: stencil rows in canonical order, 64 at a time ----------
      const QueryRange qr = query_range(G, q.x, q.y, q.z, hc.rq_hands);
      const int ny = qr.empty ? 0 : (qr.hi[1] - qr.lo[1] + 1);
      const int nz = qr.empty ? 0 : (qr.hi[2] - qr.lo[2] + 1);
      const int nrows = ny * nz;
      auto classify = [&](const float4& p, float4& d) -> int {  // 0 miss, 1 in radius, 3 + in slab
        d = make_float4(p.x - q.x, p.y - q.y, p.z - q.z, 0.f);
        const float d2 = (d.x * d.x + d.y * d.y) + d.z * d.z;
        if (!(d2 < r2_hands)) return 0;
        // hand_search.cpp:209-210 centred in float then widened; :329-339 crop on row 2 of frame^T p
        const double p0 = (double)d.x, p1 = (double)d.y, p2 = (double)d.z;
        const double zf = (F[0][2] * p0 + F[1][2] * p1) + F[2][2] * p2;
        return (zf > -1.0 * hh && zf < hh) ? 3 : 1;
      };
      int K = 0, k2 = 0;
      bool too_long = false;
      for (int r0 = 0; r0 < nrows && !too_long; r0 += kWave) {
        const int r = r0 + lane;
        int b = 0, len = 0;
        if (r < nrows) {
          const int cz = qr.lo[2] + r / ny, cy = qr.lo[1] + r % ny;
          int cxa = qr.lo[0], cxb = qr.hi[0];
          if (tighten) tighten_row(G, hc, q, F, hh, cy, cz, cxa, cxb);
          if (cxa <= cxb) {
            const int rowbase = (cz * G.dims[1] + cy) * G.dims[0];
            b = (int)A.cell[rowbase + cxa];
            len = (int)A.cell[rowbase + cxb + 1] - b;
          }
        }
        // the rows of this round cut into pieces of <= 8 consecutive points, in row order
        const int np = (len + 7) >> 3;
        int inc = np;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(inc, o, 64);
          if (lane >= o) inc += v;
        }
        const int T = __builtin_amdgcn_readlane(inc, 63);
        const int poff = inc - np;
        for (int pbase = 0; pbase < T && !too_long; pbase += kWvPieces) {
          wave_mem_sync();  // the readers of the previous batch are done
          {  // table of the pieces [pbase, pbase + kWvPieces) -- a row usually contributes one or two
            const int k0 = max(pbase - poff, 0), k1 = min(np, pbase + kWvPieces - poff);
            for (int k = k0; k < k1; k++) {
              pst[poff + k - pbase] = b + 8 * k;
              pln[poff + k - pbase] = (unsigned char)min(8, len - 8 * k);
            }
          }
          wave_mem_sync();
          const int Tb = min(kWvPieces, T - pbase);
          // eight pieces per wave-wide load (16 B per lane, contiguous within a piece), four such loads
          // issued before the first result is used; a step's survivors are in canonical order
          // lane by lane, so ballot + popcount append them at their final offsets
          for (int q0 = 0; q0 < Tb; q0 += 32) {
            int pb[4], pl[4];
            float4 pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int qi = q0 + 8 * u + grp;
              const bool ok = qi < Tb;
              pb[u] = ok ? pst[qi] : 0;
              pl[u] = ok ? (int)pln[qi] : 0;
              pv[u] = A.pts[pb[u] + max(min(lg, pl[u] - 1), 0)];  // unconditional, clamped (no branch)
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              float4 d;
              const int cls = (lg < pl[u]) ? classify(pv[u], d) : 0;
              const unsigned long long mask = __ballot(cls == 3);
              k2 += (cls != 0) ? 1 : 0;
              if (cls == 3) {
                const int dst = K + __popcll(mask & lt_mask);
                if (dst < kWvLds) POS[dst] = pb[u] + lg;
                else if (dst < kWvLds + kWvGcap) gpos[dst - kWvLds] = pb[u] + lg;
              }
              K += __popcll(mask);
            }
          }
          too_long = K > kWvLds + kWvGcap;  // uniform
        }
      }
      if (too_long) {  // dense neighbourhood: the workgroup kernel's global-scratch stage takes it
        if (lane == 0) {
          const unsigned at = atomicAdd(&A.st->n_overflow, 1u);
          A.overflow[at] = t;
        }
        break;
      }
      if (!tighten) acc_k2 += (unsigned long long)wave_sum_i(k2);
      acc_kcrop += (unsigned long long)K;
      if (K == 0) break;  // hand_search.cpp:201 (no neighbours) / no cropped points => no fingers
      wave_mem_sync();    // the list is complete and visible to every lane of the wave

      // ---- pass A for ALL orientations in one sweep over the list (see k_sweep.hip) ----------------
      auto exact_A = [&](int i, float fx, float fy, float fz, unsigned& flg, unsigned& blk_bits) {
        const double cs = hc.cos_t[i], sn = hc.sin_t[i];
        const double p0 = (double)fx, p1 = (double)fy, p2 = (double)fz;
        double c0[3], c1[3];  // columns 0 and 1 of frame_rot = frame * rot, hand_search.cpp:356-357
#pragma unroll
        for (int a = 0; a < 3; a++) {
          c0[a] = (F[a][0] * cs + F[a][1] * sn) + F[a][2] * 0.0;
          c1[a] = (F[a][0] * (-1.0 * sn) + F[a][1] * cs) + F[a][2] * 0.0;
        }
        const double x = (c0[0] * p0 + c0[1] * p1) + c0[2] * p2;
        const double y = (c1[0] * p0 + c1[1] * p1) + c1[2] * p2;
        if (y < top0) {
          flg |= 1u;
          if (y < bottom0) flg |= 2u;
          if (slot_span <= 2) {
#pragma unroll
            for (int blk = 0; blk < 2; blk++) {
              double rel = (x - (blk ? slot_base1 : slot_base0)) * slot_inv_step;
              rel = __builtin_fmin(__builtin_fmax(rel, -4.0), 14.0);
              const int kf = (int)__builtin_floor(rel);
#pragma unroll
              for (int dk = -1; dk <= 1; dk++) {
                const int k = kf + dk;
                const double hk = (double)k * slot_step;
                const double f = blk ? hk : ((hk - hand_od) + finger_w);
                const bool in = (x > f) & (x < f + finger_w) & ((unsigned)k < 10u);
                blk_bits |= in ? (1u << ((blk * 10 + k) & 31)) : 0u;
              }
            }
          } else {
            for (int kk = 0; kk < 20; kk++)
              if (x > hc.fs[kk] && x < hc.fsr[kk]) blk_bits |= (1u << kk);
          }
        }
      };
      unsigned blk_acc[RMAX], flg_acc[RMAX], raw_acc[RMAX];
#pragma unroll
      for (int i = 0; i < RMAX; i++) {
        blk_acc[i] = 0;
        flg_acc[i] = 0;
        raw_acc[i] = 0;
      }
      unsigned s_below = 0, s_behind = 0;
      {
        const float n0 = (float)F[0][0], n1 = (float)F[1][0], n2 = (float)F[2][0];
        const float b0 = (float)F[0][1], b1 = (float)F[1][1], b2 = (float)F[2][1];
        const float top0f = (float)top0, bot0f = (float)bottom0;
        const float mY = 1.0e-6f, eX = 1.0e-4f;
        const float invs = (float)slot_inv_step, rm1 = (float)(slot_ratio - 1.0);
        const bool fast_ok = (slot_span <= 2) && slot_ratio > 1.001 && slot_ratio < 1.999 &&
                             __builtin_fabs((slot_base1 - slot_base0) * slot_inv_step - 9.0) < 1.0e-9;
        unsigned alive = (R >= 32) ? 0xFFFFFFFFu : ((1u << R) - 1u);  // wave-uniform, scalar
        float qx, qy, qz;  // next step's point, requested one step ahead
        ldp(min(lane, K - 1), qx, qy, qz);
        for (int j0 = 0; j0 < K; j0 += kWave) {
          if (alive == 0u) break;
          const int j = j0 + lane;
          const bool valid = j < K;
          const float px = qx, py = qy, pz = qz;
          ldp(min(j + kWave, K - 1), qx, qy, qz);
          const float u = (n0 * px + n1 * py) + n2 * pz;
          const float v = (b0 * px + b1 * py) + b2 * pz;
          unsigned need_exact = 0;
#pragma unroll
          for (int i = 0; i < RMAX; i++) {
            if (i < R && ((alive >> i) & 1u)) {  // wave-uniform
              const float cf = cosf_t[i], sf = sinf_t[i];
              const float ya = __builtin_fmaf(cf, v, -(sf * u));
              const float dy = __builtin_fminf(__builtin_fabsf(ya - top0f), __builtin_fabsf(ya - bot0f));
              // points in front of the fingertips take part in nothing (finger_hand.cpp:27); the list is
              // in cell order, so the 64 points of a step often ALL lie in front: skip the slot arithmetic
              if (fast_ok && __ballot(valid && (ya < top0f || dy < mY)) == 0ull) continue;
              const float xa = __builtin_fmaf(cf, u, sf * v);
              const float rel = xa * invs;
              const float kff = __builtin_floorf(rel);
              const float frac = rel - kff;
              const float dx = __builtin_fminf(__builtin_fminf(frac, 1.f - frac), __builtin_fabsf(frac - rm1));
              const bool below = ya < top0f;
              const bool fast = fast_ok && !(dy < mY) && !((dx < eX) && below);
              const bool hit = fast && below && valid;
              if (hit) {
                int pbit = (int)kff + 11;
                pbit = min(max(pbit, 0), 22);
                const unsigned bit = 1u << pbit;
                raw_acc[i] |= bit | ((frac < rm1) ? (bit >> 1) : 0u);
              }
              if (!fast && valid) need_exact |= 1u << i;
              if (__ballot(hit)) {
                s_below |= 1u << i;
                if (__ballot(hit && ya < bot0f)) {
                  s_behind |= 1u << i;
                  alive &= ~(1u << i);  // blocked from behind: nothing more to learn (finger_hand.cpp:27-38)
                }
              }
            }
          }
          while (need_exact) {  // the rare exact evaluations: one copy of the f64 code
            const int ie = __ffs((int)need_exact) - 1;
            need_exact &= need_exact - 1u;
            unsigned flg = 0, bits = 0;
            exact_A(ie, px, py, pz, flg, bits);
#pragma unroll
            for (int i = 0; i < RMAX; i++) {
              flg_acc[i] |= (i == ie) ? flg : 0u;
              blk_acc[i] |= (i == ie) ? bits : 0u;
            }
          }
        }
      }
      // gates: lane i evaluates orientation i (finger_hand.cpp:35-42, hand_search.cpp:366, :370)
      unsigned hand_l = 0;
      {
        unsigned cb = 0, cf = 0;
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
          if (i < R) {
            const unsigned pm = (raw_acc[i] >> 2) & 0x7FFFFu;  // the 19 positions -9 .. 9
            const unsigned bsum = wave_or_u(blk_acc[i] | (pm & 0x3FFu) | ((pm >> 9) << 10));
            const unsigned fsum = wave_or_u(flg_acc[i]) | ((s_below >> i) & 1u) | (((s_behind >> i) & 1u) ? 3u : 0u);
            cb = (lane == i) ? bsum : cb;
            cf = (lane == i) ? fsum : cf;
          }
        }
        const unsigned free_ = (~cb) & 0xFFFFFu;
        const bool open = (lane < R) && !(cf & 2u) && (cf & 1u) && (__popc(free_) > 2);
        hand_l = open ? (free_ & (free_ >> 10) & 0x3FFu) : 0u;
      }
      unsigned long long todo = __ballot(hand_l != 0u);
      while (todo) {
        const int oi = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const unsigned hand = (unsigned)__builtin_amdgcn_readlane((int)hand_l, oi);
        const int nvalid = __popc(hand);
        const double cs = hc.cos_t[oi], sn = hc.sin_t[oi];
        double Fr[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          Fr[a][0] = (F[a][0] * cs + F[a][1] * sn) + F[a][2] * 0.0;
          Fr[a][1] = (F[a][0] * (-1.0 * sn) + F[a][1] * cs) + F[a][2] * 0.0;
          Fr[a][2] = (F[a][0] * 0.0 + F[a][1] * 0.0) + F[a][2] * 1.0;
        }
        // deepenHand, finger_hand.cpp:96-134: middle valid hand = valid[ceil(n/2) - 1]
        int idx = 0;
        {
          const int want = (nvalid + 1) / 2 - 1;
          int seen = 0;
          for (int k = 0; k < 10; k++)
            if (hand & (1u << k)) {
              if (seen == want) idx = k;
              seen++;
            }
        }
        const double fl0 = hc.fs[idx], fl1 = hc.fsr[idx], fr0 = hc.fs[10 + idx], fr1 = hc.fsr[10 + idx];
        // pass B: first failing depth step and surface = min y over ALL rotated points
        int kfail = n_depths;
        double miny = __builtin_inf();
        {
          float qx, qy, qz;
          ldp(min(lane, K - 1), qx, qy, qz);
          for (int j0 = 0; j0 < K; j0 += kWave) {
            const int j = j0 + lane;
            const float fx_ = qx, fy_ = qy, fz_ = qz;
            ldp(min(j + kWave, K - 1), qx, qy, qz);
            if (j < K) {
              const double p0 = (double)fx_, p1 = (double)fy_, p2 = (double)fz_;
              const double x = (Fr[0][0] * p0 + Fr[1][0] * p1) + Fr[2][0] * p2;
              const double y = (Fr[0][1] * p0 + Fr[1][1] * p1) + Fr[2][1] * p2;
              miny = (y < miny) ? y : miny;
              const bool zone = (x > fl0 && x < fl1) || (x > fr0 && x < fr1);
              for (int di = 0; di < kfail; di++) {
                const double d = s_depths[di];
                if (y < d && (zone || y < d - hand_depth)) {
                  kfail = di;
                  break;
                }
              }
            }
          }
        }
        kfail = wave_min_i(kfail);
        const double surface = wave_min_d(miny);
        double top = top0, bottom = bottom0;
        if (kfail > 0) {  // last successful step, finger_hand.cpp:128-129
          top = s_depths[kfail - 1];
          bottom = top - hand_depth;
        }
        // closing region, finger_hand.cpp:137-180
        const double left = fl0 + hc.finger_width;
        const double right = fr0;
        const double center = 0.5 * (left + right);
        // pass C: in-box points, appended in list order (one wave: one pass)
        int P = 0;
        double mnx = __builtin_inf(), mxx = -__builtin_inf();
        wave_mem_sync();  // (the previous orientation's readers of the index list are done)
        {
          float qx, qy, qz;
          ldp(min(lane, K - 1), qx, qy, qz);
          for (int j0 = 0; j0 < K; j0 += kWave) {
            const int j = j0 + lane;
            bool in = false;
            const float fx_ = qx, fy_ = qy, fz_ = qz;
            ldp(min(j + kWave, K - 1), qx, qy, qz);
            if (j < K) {
              const double p0 = (double)fx_, p1 = (double)fy_, p2 = (double)fz_;
              const double x = (Fr[0][0] * p0 + Fr[1][0] * p1) + Fr[2][0] * p2;
              const double y = (Fr[0][1] * p0 + Fr[1][1] * p1) + Fr[2][1] * p2;
              in = (y < top && x > left && x < right);
              if (in) {
                mnx = (x < mnx) ? x : mnx;
                mxx = (x > mxx) ? x : mxx;
              }
            }
            const unsigned long long mask = __ballot(in);
            if (in) {
              const int dst = P + __popcll(mask & lt_mask);
              if (dst < kWvBox) box16[dst] = (unsigned short)j;
            }
            P += __popcll(mask);
          }
        }
        if (P == 0) continue;                                         // hand_search.cpp:377-381
        mnx = wave_min_d(mnx);
        mxx = wave_max_d(mxx);
        wave_mem_sync();
        long long off = -1;
        if (A.emit_lists) {
          unsigned long long o64 = 0;
          if (lane == 0) o64 = atomicAdd(&A.st->arena_top, (unsigned long long)P);
          const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)o64);
          const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(o64 >> 32));
          off = (long long)(((unsigned long long)hi << 32) | lo);
          if (off + P > A.arena_cap) {
            if (lane == 0) atomicOr(&A.st->err_flags, 1u);
            off = -1;
          }
        }
        // pass D: unit-box scaling (hand_search.cpp:399-409), list emission, antipodal extents
        const double baseline = 0.1;
        const double left_const = left - 0.5 * (baseline - (right - left));
        const double lower[3] = {left_const, bottom, -1.0 * hh};
        const double scales[3] = {1.0 / baseline, 1.0 / (top - bottom), 1.0 / (2.0 * hh)};
        const double lt = scales[0] * (mnx - lower[0]) + 0.003;       // antipodal.cpp:16
        const double rt = scales[0] * (mxx - lower[0]) - 0.003;       // antipodal.cpp:17
        int nl = 0, nr = 0;
        double e[8] = {-__builtin_inf(), __builtin_inf(), -__builtin_inf(), __builtin_inf(),
                       -__builtin_inf(), __builtin_inf(), -__builtin_inf(), __builtin_inf()};
        // e: lmaxy lminy lmaxz lminz rmaxy rminy rmaxz rminz
        auto emit = [&](int bpos, int j) {
          float fx_, fy_, fz_;
          ldp(j, fx_, fy_, fz_);
          const double p0 = (double)fx_, p1 = (double)fy_, p2 = (double)fz_;
          const float4 nn = A.nrm[pos_at(j)];  // hand_search.cpp:211, :394: the point's normal
          const double q0 = (double)nn.x, q1 = (double)nn.y, q2 = (double)nn.z;
          double X[3], Y[3], U[3];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            X[a] = (Fr[0][a] * p0 + Fr[1][a] * p1) + Fr[2][a] * p2;
            Y[a] = (Fr[0][a] * q0 + Fr[1][a] * q1) + Fr[2][a] * q2;
            U[a] = scales[a] * (X[a] - lower[a]);
          }
          if (off >= 0) {
            double* dst = A.arena + (size_t)(off + bpos) * 6;
            dst[0] = U[0]; dst[1] = U[1]; dst[2] = U[2];
            dst[3] = Y[0]; dst[4] = Y[1]; dst[5] = Y[2];
          }
          const double ldot = (-1.0 * Y[0] + 0.0 * Y[1]) + 0.0 * Y[2];  // antipodal.cpp:20-25
          const double rdot = (1.0 * Y[0] + 0.0 * Y[1]) + 0.0 * Y[2];
          if (ldot > hc.cos_fc && U[0] < lt) {
            nl++;
            e[0] = (U[1] > e[0]) ? U[1] : e[0]; e[1] = (U[1] < e[1]) ? U[1] : e[1];
            e[2] = (U[2] > e[2]) ? U[2] : e[2]; e[3] = (U[2] < e[3]) ? U[2] : e[3];
          }
          if (rdot > hc.cos_fc && U[0] > rt) {
            nr++;
            e[4] = (U[1] > e[4]) ? U[1] : e[4]; e[5] = (U[1] < e[5]) ? U[1] : e[5];
            e[6] = (U[2] > e[6]) ? U[2] : e[6]; e[7] = (U[2] < e[7]) ? U[2] : e[7];
          }
        };
        if (P <= kWvBox) {
          for (int bpos = lane; bpos < P; bpos += kWave) emit(bpos, (int)box16[bpos]);
        } else {
          // a closing region larger than the index list: walk the whole list again, in order
          int run = 0;
          for (int j0 = 0; j0 < K; j0 += kWave) {
            const int j = j0 + lane;
            bool in = false;
            if (j < K) {
              float fx_, fy_, fz_;
              ldp(j, fx_, fy_, fz_);
              const double p0 = (double)fx_, p1 = (double)fy_, p2 = (double)fz_;
              const double x = (Fr[0][0] * p0 + Fr[1][0] * p1) + Fr[2][0] * p2;
              const double y = (Fr[0][1] * p0 + Fr[1][1] * p1) + Fr[2][1] * p2;
              in = (y < top && x > left && x < right);
            }
            const unsigned long long mask = __ballot(in);
            if (in) emit(run + __popcll(mask & lt_mask), j);
            run += __popcll(mask);
          }
        }
        nl = wave_sum_i(nl);
        nr = wave_sum_i(nr);
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          e[k] = wave_max_d(e[k]);
          e[k + 1] = wave_min_d(e[k + 1]);
        }
        const int slot = t * R + oi;
        if (lane == 0) {
          int label = 0;
          if (nl > 0 || nr > 0) label = 1;                              // antipodal.cpp:48-51
          if (nl > 0 && nr > 0) {                                       // :54-81
            const double top_y = (e[0] < e[4]) ? e[0] : e[4], bot_y = (e[1] > e[5]) ? e[1] : e[5];
            const double top_z = (e[2] < e[6]) ? e[2] : e[6], bot_z = (e[3] > e[7]) ? e[3] : e[7];
            if (top_y > bot_y && top_z > bot_z) label = 2;
          }
          ag2_hypothesis h;
          const double ys[3] = {surface, bottom, top};
          double* dstv[3] = {h.surface, h.bottom, h.top};
          for (int k = 0; k < 3; k++)                                   // finger_hand.cpp:189-199
            for (int a = 0; a < 3; a++)
              dstv[k][a] = ((Fr[a][0] * center + Fr[a][1] * ys[k]) + Fr[a][2] * 0.0) + smp[a];
          for (int a = 0; a < 3; a++) {                                 // hand_search.cpp:383-385
            h.binormal[a] = Fr[a][0];
            h.approach[a] = Fr[a][1];
            h.axis[a] = Fr[a][2];
          }
          h.width = mxx - mnx;                                          // hand_search.cpp:397
          h.score = 0.0;
          h.sample_slot = slot_base + t;
          h.orientation = oi;
          h.half_antipodal = (label >= 1) ? 1 : 0;                      // hand_search.cpp:417-418
          h.full_antipodal = (label == 2) ? 1 : 0;
          h.reserved = 0;
          h.n_points = P;
          // pruneGraspsOnHandParameters, grasp_detector.cpp:363-395
          bool keep = !(hc.filter_half && !h.half_antipodal);
          if (keep) {
            const double hw = 0.5 * hc.hand_outer_diameter;
            double mn[3], mx[3];
            for (int a = 0; a < 3; a++) {
              const double c5[5] = {h.bottom[a] + hw * h.binormal[a], h.bottom[a] - hw * h.binormal[a],
                                    h.top[a] + hw * h.binormal[a], h.top[a] - hw * h.binormal[a],
                                    h.bottom[a] - 0.10 * h.approach[a]};
              mn[a] = mx[a] = c5[0];
              for (int k = 1; k < 5; k++) {
                mn[a] = (c5[k] < mn[a]) ? c5[k] : mn[a];
                mx[a] = (c5[k] > mx[a]) ? c5[k] : mx[a];
              }
            }
            keep = h.width >= hc.min_aperture && h.width <= hc.max_aperture &&
                   mn[2] >= (double)cloud_min_z && mn[1] >= (double)hc.ws_min_y &&
                   mx[1] <= (double)hc.ws_max_y && mn[0] >= (double)hc.ws_min_x &&
                   mx[0] <= (double)hc.ws_max_x;
          }
          A.table[slot] = h;
          A.tab_off[slot] = off;
          A.tab_keep[slot] = keep ? 1 : 4;  // slot state: 0 empty, 1 survives the prune, 4 pruned away
        }
        acc_hyp++;
        acc_p += (unsigned long long)P;
        acc_maxp = max(acc_maxp, (unsigned)P);
      }
    } while (false);
    t = __builtin_amdgcn_readfirstlane(nxt);
  }
  if (lane == 0) {
    if (acc_k2) atomicAdd(&A.st->sum_k2, acc_k2);
    if (acc_kcrop) atomicAdd(&A.st->sum_kcrop, acc_kcrop);
    if (acc_hyp) {
      atomicAdd(&A.st->n_hyp, acc_hyp);
      atomicAdd(&A.st->sum_p, acc_p);
      atomicMax(&A.st->max_p, acc_maxp);
    }
  }
}

size_t sweep_wave_gpos_ints(int grid) { return (size_t)grid * kWvWaves * kWvGcap; }

int launch_sweep_wave(ag2_ctx* c, const SweepArgs& A, size_t s, int R) {
  typedef void (*Fn)(SweepArgs);
  const Fn fn = (R <= 8) ? k_sweep_wave<8> : (R <= 16 ? k_sweep_wave<16> : k_sweep_wave<32>);
  const size_t lds = (size_t)kWvWaves * kWvSliceBytes;
  static bool attr_set = false;
  if (!attr_set) {
    for (Fn f : {(Fn)k_sweep_wave<8>, (Fn)k_sweep_wave<16>, (Fn)k_sweep_wave<32>})
      AG2_HIP(c, hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int grid = (int)std::min<size_t>((s + kWvWaves - 1) / kWvWaves, 256 * kWvWgPerCu);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kWvWaves * kWave), lds, c->stream, A);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
