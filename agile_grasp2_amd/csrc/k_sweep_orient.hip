// k_sweep_orient.hip -- K3, second half: the per-orientation passes of the hand sweep, one workgroup
// per (sample, orientation) pair that passed the gates.
//
// calculateHand's tail (hand_search.cpp:366-426) for one orientation: deepenHand
// (finger_hand.cpp:96-134), computePointsInClosingRegion (:137-180), calculateGraspParameters
// (:183-214), the unit-box scaling (hand_search.cpp:399-409), Antipodal::evaluateGrasp
// (antipodal.cpp:8-84), the prune predicate (grasp_detector.cpp:363-395) and the record.
//
// k_sweep used to run these passes inside the sample's workgroup, one passing orientation after the
// other.  They are the part of the sweep with the most live state (rotated frame, eight f64 extents, scaling
// constants: the spills of the one-kernel sweep were here) and the least parallelism (0.5 passing
// orientations per sample at configuration 2, 1.4 at configuration 3, a few thousand points each).
// Split off, every pair is its own workgroup: the first kernels (k_sweep<.., SPLIT>) stop at the gates,
// leave the sample's cropped list in a list arena -- one float4 per point: the centred coordinates
// p - q (float: the very value the crop computed) and the sorted position -- and queue the pair; this
// kernel streams the list without any indirection.  Same arithmetic, same results.
#include "ag2_internal.h"
#include "k_sweep_common.h"

namespace ag2 {

constexpr int kOThreads = 256;
constexpr int kONW = kOThreads / kWave;
constexpr int kOMaskWords = 16;    // 64-bit membership ballots of one wave's quarter of a chunk
constexpr int kOMaskChunks = 8;    // chunks whose ballots are kept from pass C for pass D (lists up to 19 456 points)
// lists are staged in LDS in chunks of this many points (the usual list is one chunk)
constexpr int kOStage = 2432;
static_assert((((kOStage + 3) / 4 + 63) & ~63) / 64 <= 16, "mask words per wave");

struct OrientShared {
  double fs[20], fsr[20];
  double depths[kMaxDepths];
  double depths_b[kMaxDepths];     // depths[i] - hand_depth: where the hand's back is at deepen step i
  Red<kONW> red;
  unsigned long long inmask[kONW][kOMaskChunks * kOMaskWords];
  int chunk_cnt[kOMaskChunks][kONW];  // members per (chunk, wave) of a list of several chunks
  int next_w[2];
  long long arena_off;
  double smp[3];                   // the sample (for the record, which one thread writes at the very end)
  // what only the record needs of the passes' results: parked here by thread 0 as soon as it is known, so that
  // 256 lanes do not carry twelve registers of it through pass D (they were the kernel's scratch: 11 spilled
  // VGPRs = 48 B per lane = 26 MB of writes per cfg2 step)
  double rec_surface, rec_top, rec_bottom, rec_center, rec_width;
  struct {
    struct {
      float px[kOStage], py[kOStage], pz[kOStage];  // the staged chunk: centred coordinates ...
      unsigned short box16[kOStage];               // ... and, for a one-chunk list, its members' indices
    } st;
  } u;
};
static_assert(sizeof(OrientShared) * 4 <= 160 * 1024, "k_sweep_orient: four workgroups per CU");

// (four workgroups per CU: 128 VGPRs, 52 B of scratch per lane around pass D; three per CU -- 150 VGPRs,
// no scratch -- is equal at configuration 2 and 7 % slower at configuration 3)
__global__ void __launch_bounds__(kOThreads, 4) k_sweep_orient(SweepArgs A) {
  __shared__ OrientShared S;
  constexpr int NT = kOThreads, NW = kONW;
  const HandConst& hc = *A.hc;
  const float cloud_min_z = A.gp ? A.gp->min_z : A.min_z;
  const int slot_base = A.fa ? (int)A.fa->slot_base : A.slot_base;
  const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int R = hc.R;
  const double hh = hc.hand_height;
  const int n_work = (int)A.st->n_pairs;
  if (n_work == 0) return;
  if (tid < 20) {
    S.fs[tid] = hc.fs[tid];
    S.fsr[tid] = hc.fsr[tid];
  }
  if (tid < kMaxDepths) {
    S.depths[tid] = hc.depths[tid];
    S.depths_b[tid] = hc.depths[tid] - hc.hand_depth;  // the very f64 difference the reference's test forms
  }
  __syncthreads();
  const int n_depths = hc.n_depths;
  const double hand_depth = hc.hand_depth;
  const double top0 = hc.init_bite, bottom0 = hc.init_bite - hc.hand_depth;
  int red_sel = 0;
  unsigned nxt = 0;
  int it = 0;
  auto next_work = [&]() -> int {
    if (tid == 0) S.next_w[it & 1] = (int)nxt;
    __syncthreads();  // also: every reader of S of this iteration is done
    const int w = __builtin_amdgcn_readfirstlane(S.next_w[it & 1]);
    it++;
    return w;
  };
  for (int w = blockIdx.x; w < n_work; w = next_work()) {
    if (tid == 0) nxt = gridDim.x + atomicAdd(&A.st->work_next[2], 1u);
    const SweepPair pr = A.pairs[w];
    const int t = pr.t, oi = pr.oi, K = pr.K;
    if (K <= 0) continue;  // (uniform) a place whose list found no room in the arena: the run is being repeated
    const unsigned hand = pr.hand;
    const float4* plist = A.lists + pr.list_off;
    // (the list's segments: the same in every lane, kept in scalar registers)
    const int ge0 = __builtin_amdgcn_readfirstlane(pr.segs.end[0]), ge1 = __builtin_amdgcn_readfirstlane(pr.segs.end[1]),
              ge2 = __builtin_amdgcn_readfirstlane(pr.segs.end[2]);
    const int gs0 = __builtin_amdgcn_readfirstlane(pr.segs.shift[0]), gs1 = __builtin_amdgcn_readfirstlane(pr.segs.shift[1]),
              gs2 = __builtin_amdgcn_readfirstlane(pr.segs.shift[2]);
    auto lslot = [&](int j) { return list_slot6(ge0, ge1, ge2, gs0, gs1, gs2, j); };
    // The list is worked through in chunks of kOStage points staged in LDS: every thread has five loads
    // of the (contiguous) list in flight, and the arithmetic of a pass reads LDS.  The usual list is
    // ONE chunk, staged once for all passes; a long one is re-staged by each pass.
    const int nchunks = (K + kOStage - 1) / kOStage;
    auto stage = [&](int c) -> int {
      const int c0 = c * kOStage, clen = min(kOStage, K - c0);
      constexpr int kPer = (kOStage + NT - 1) / NT, kHalf = (kPer + 1) / 2;
      __syncthreads();  // the readers of the chunk that is being replaced are done
#pragma unroll
      for (int h = 0; h < 2; h++) {  // five loads in flight per thread, twice
        if (h * kHalf * NT >= clen) break;  // (uniform) a short chunk is all in the first round trip
        float4 v[kHalf];
#pragma unroll
        for (int k = 0; k < kHalf; k++) v[k] = plist[lslot(c0 + min(tid + (h * kHalf + k) * NT, clen - 1))];
#pragma unroll
        for (int k = 0; k < kHalf; k++) {
          const int j = tid + (h * kHalf + k) * NT;
          if (j < clen) {
            S.u.st.px[j] = v[k].x;
            S.u.st.py[j] = v[k].y;
            S.u.st.pz[j] = v[k].z;
          }
        }
      }
      __syncthreads();
      return clen;
    };
    stage(0);  // (here, before the frame and the rotation occupy registers)
    const double* fr = A.frames + (size_t)t * 12;
    // frame = [normal binormal curvature_axis] as columns, hand_search.cpp:325-326
    const double F[3][3] = {{fr[3], fr[6], fr[9]}, {fr[4], fr[7], fr[10]}, {fr[5], fr[8], fr[11]}};
    // (requested with the frame: read at the record it was one more round trip on the pair's critical path;
    // S.smp's reader of the previous pair is behind the barrier of next_work)
    if (tid == 0) {
      S.smp[0] = fr[0];
      S.smp[1] = fr[1];
      S.smp[2] = fr[2];
    }
    const int nvalid = __popc(hand);
    // rot = [c -s 0; s c 0; 0 0 1], frame_rot = frame * rot, hand_search.cpp:356-357
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
    const double fl0 = S.fs[idx], fl1 = S.fsr[idx], fr0 = S.fs[10 + idx], fr1 = S.fsr[10 + idx];
    auto rot_xy = [&](int j, double& x, double& y) {  // staged point j in the rotated frame
      const double p0 = (double)S.u.st.px[j], p1 = (double)S.u.st.py[j], p2 = (double)S.u.st.pz[j];
      x = (Fr[0][0] * p0 + Fr[1][0] * p1) + Fr[2][0] * p2;
      y = (Fr[0][1] * p0 + Fr[1][1] * p1) + Fr[2][1] * p2;
    };
    // pass B: first depth step that fails (some point under the finger pads or behind the hand)
    // (the same pass yields surface = min y over ALL rotated points, finger_hand.cpp:158)
    // The first failing step of ONE point is a search, not a scan: step di fails the point when
    // y < depths[di] && (zone || y < depths[di] - hand_depth), both tables ascend, so it is the first index
    // of the table that decides -- depths[] under a finger pad, depths_b[] elsewhere -- with y below its
    // entry.  The steps are 5 mm apart (finger_hand.cpp:122), which places an estimate; the exact f64
    // comparisons against the table then settle it (the same comparisons the scan made: same result).  The
    // scan cost a thread ten iterations per point until one of ITS points failed a step -- pass B was a
    // third of this kernel's time.
    int kfail = n_depths;
    double miny = __builtin_inf();
    const double depth0 = S.depths[0], depth0_b = S.depths_b[0];
    for (int c = 0; c < nchunks; c++) {
      const int clen = (c == 0) ? min(kOStage, K) : stage(c);
      for (int j = tid; j < clen; j += NT) {
        double x, y;
        rot_xy(j, x, y);
        miny = (y < miny) ? y : miny;
        const bool zone = (x > fl0 && x < fl1) || (x > fr0 && x < fr1);
        const double* tab = zone ? S.depths : S.depths_b;
        const double tq = (y - (zone ? depth0 : depth0_b)) * 200.0;
        int e = (tq < 0.0) ? 0 : ((tq >= (double)n_depths) ? n_depths : (int)tq + 1);
        if (e < kfail + 3) {  // (an estimate three or more above the current minimum cannot lower it: it is off by at most one)
          while (e > 0 && y < tab[e - 1]) e--;
          while (e < n_depths && !(y < tab[e])) e++;
          kfail = min(kfail, e);
        }
      }
    }
    kfail = wave_min_i(kfail);
    miny = wave_min_d(miny);
    red_sel ^= 1;
    if (lane == 0) {
      S.red.i[red_sel][wid][0] = kfail;
      S.red.d[red_sel][wid][0] = miny;
    }
    __syncthreads();
    kfail = n_depths;
    double surface = __builtin_inf();
#pragma unroll
    for (int k = 0; k < NW; k++) {
      kfail = min(kfail, S.red.i[red_sel][k][0]);
      const double v = S.red.d[red_sel][k][0];
      surface = (v < surface) ? v : surface;
    }
    double top = top0, bottom = bottom0;
    if (kfail > 0) {  // last successful step, finger_hand.cpp:128-129
      top = S.depths[kfail - 1];
      bottom = top - hand_depth;
    }
    // closing region, finger_hand.cpp:137-180
    const double left = fl0 + hc.finger_width;
    const double right = fr0;
    if (tid == 0) {  // (for the record; S.rec_*'s reader of the previous pair is behind the barrier of next_work)
      S.rec_surface = surface;
      S.rec_top = top;
      S.rec_bottom = bottom;
      S.rec_center = 0.5 * (left + right);
    }
    // pass C: members of the closing region.  Inside a chunk every wave takes a contiguous quarter, so
    // that (chunk, wave, lane) order is list order; the membership ballots of a chunk are kept in LDS.
    auto seg_of = [&](int clen, int& jb, int& je) {
      const int seg = (((clen + NW - 1) / NW) + 63) & ~63;
      jb = min(wid * seg, clen);
      je = min(jb + seg, clen);
    };
    // (mbase: where in inmask[wid] the chunk's ballots go -- a list of up to kOMaskChunks chunks keeps them all)
    auto members = [&](int clen, double& mn, double& mx, int mbase) -> int {  // this wave's count in the chunk
      int jb, je, cnt = 0;
      seg_of(clen, jb, je);
      for (int j0 = jb; j0 < je; j0 += 64) {
        const int j = j0 + lane;
        bool in = false;
        if (j < je) {
          double x, y;
          rot_xy(j, x, y);
          in = (y < top && x > left && x < right);
          if (in) {
            mn = (x < mn) ? x : mn;
            mx = (x > mx) ? x : mx;
          }
        }
        const unsigned long long mask = __ballot(in);
        if (lane == 0) S.inmask[wid][mbase + ((j0 - jb) >> 6)] = mask;
        cnt += __popcll(mask);
      }
      return cnt;
    };
    int cnt = 0;
    double mnx = __builtin_inf(), mxx = -__builtin_inf();
    const bool keep_masks = nchunks > 1 && nchunks <= kOMaskChunks;  // (uniform)
    for (int c = 0; c < nchunks; c++) {
      const int clen = (nchunks > 1) ? stage(c) : K;  // (a single chunk is still staged from pass B)
      const int mine = members(clen, mnx, mxx, keep_masks ? c * kOMaskWords : 0);
      if (keep_masks && lane == 0) S.chunk_cnt[c][wid] = mine;
      cnt += mine;
    }
    mnx = wave_min_d(mnx);
    mxx = wave_max_d(mxx);
    red_sel ^= 1;
    if (lane == 0) {
      S.red.i[red_sel][wid][0] = cnt;
      S.red.d[red_sel][wid][0] = mnx;
      S.red.d[red_sel][wid][1] = mxx;
    }
    __syncthreads();
    int P = 0, pbase_w = 0;
    mnx = __builtin_inf();
    mxx = -__builtin_inf();
#pragma unroll
    for (int k = 0; k < NW; k++) {
      const int ck = S.red.i[red_sel][k][0];
      if (k < wid) pbase_w += ck;
      P += ck;
      const double a = S.red.d[red_sel][k][0], b = S.red.d[red_sel][k][1];
      mnx = (a < mnx) ? a : mnx;
      mxx = (b > mxx) ? b : mxx;
    }
    if (P == 0) continue;                                         // hand_search.cpp:377-381
    const int slot = t * R + oi;
    if (tid == 0) {
      S.rec_width = mxx - mnx;                                    // hand_search.cpp:397
      long long off = -1;
      if (A.emit_lists) {
        off = (long long)atomicAdd(&A.st->arena_top, (unsigned long long)P);
        if (off + P > A.arena_cap) {
          atomicOr(&A.st->err_flags, 1u);
          off = -1;
        }
      }
      S.arena_off = off;
    }
    __syncthreads();
    const long long off = S.arena_off;
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
    auto emit_p = [&](int bpos, double p0, double p1, double p2, int npos) {  // the member at ordered position bpos
      const float4 nn = A.nrm[npos];  // hand_search.cpp:211, :394: the point's normal
      const double q0 = (double)nn.x, q1 = (double)nn.y, q2 = (double)nn.z;
      double X[3], Y[3], U[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        X[a] = (Fr[0][a] * p0 + Fr[1][a] * p1) + Fr[2][a] * p2;
        Y[a] = (Fr[0][a] * q0 + Fr[1][a] * q1) + Fr[2][a] * q2;
        U[a] = scales[a] * (X[a] - lower[a]);
      }
      if (off >= 0) {  // 48 B per member as three 16-byte stores
        double2* dst = reinterpret_cast<double2*>(A.arena + (size_t)(off + bpos) * 6);
        dst[0] = make_double2(U[0], U[1]);
        dst[1] = make_double2(U[2], Y[0]);
        dst[2] = make_double2(Y[1], Y[2]);
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
    auto emit = [&](int bpos, int jl, int jg) {  // member jl of the staged chunk = list entry jg
      emit_p(bpos, (double)S.u.st.px[jl], (double)S.u.st.py[jl], (double)S.u.st.pz[jl],
             __float_as_int(plist[lslot(jg)].w));
    };
    if (nchunks == 1) {
      // the usual case: the masks of pass C give every member its ordered position; the members are
      // listed (16-bit indices) so that ALL threads share the arithmetic of the emission
      int jb, je;
      seg_of(K, jb, je);
      int run = pbase_w;
      for (int j0 = jb; j0 < je; j0 += 64) {
        const unsigned long long mask = S.inmask[wid][(j0 - jb) >> 6];  // written by this wave
        if ((mask >> lane) & 1ull) S.u.st.box16[run + __popcll(mask & lt_mask)] = (unsigned short)(j0 + lane);
        run += __popcll(mask);
      }
      __syncthreads();
      for (int b = tid; b < P; b += NT) {
        const int j = (int)S.u.st.box16[b];
        emit(b, j, j);
      }
    } else if (keep_masks) {
      // a list of a few chunks: the ballots of pass C are all still there, so this pass touches only the
      // MEMBERS -- gathered from the arena, at their ordered positions (chunk, then wave, then lane = list
      // order) -- instead of staging every chunk again and rotating all of its points a third time (that
      // was 38 % of this kernel at configuration 3)
      int run = 0;  // members in front of this wave's quarter of chunk c
      for (int c = 0; c < nchunks; c++) {
        const int c0 = c * kOStage, clen = min(kOStage, K - c0);
        int mine_base = run;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          const int ck = S.chunk_cnt[c][k];  // (written before the barrier behind pass C's reduction)
          if (k < wid) mine_base += ck;
          run += ck;
        }
        int jb, je;
        seg_of(clen, jb, je);
        int at = mine_base;
        for (int j0 = jb; j0 < je; j0 += 64) {
          const unsigned long long mask = S.inmask[wid][c * kOMaskWords + ((j0 - jb) >> 6)];
          if ((mask >> lane) & 1ull) {
            const float4 v = plist[lslot(c0 + j0 + lane)];
            emit_p(at + __popcll(mask & lt_mask), (double)v.x, (double)v.y, (double)v.z, __float_as_int(v.w));
          }
          at += __popcll(mask);
        }
      }
    } else {
      // a long list: chunk by chunk, every wave emits the members of its quarter at their ordered
      // positions (count per wave -> one LDS hop -> offsets)
      int done = 0;  // members of the chunks before this one
      for (int c = 0; c < nchunks; c++) {
        const int clen = stage(c);
        double d0 = 0.0, d1 = 0.0;
        const int mine = members(clen, d0, d1, 0);
        red_sel ^= 1;
        if (lane == 0) S.red.i[red_sel][wid][0] = mine;
        __syncthreads();
        int run = done;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          const int ck = S.red.i[red_sel][k][0];
          if (k < wid) run += ck;
          done += ck;
        }
        int jb, je;
        seg_of(clen, jb, je);
        for (int j0 = jb; j0 < je; j0 += 64) {
          const unsigned long long mask = S.inmask[wid][(j0 - jb) >> 6];
          if ((mask >> lane) & 1ull) emit(run + __popcll(mask & lt_mask), j0 + lane, c * kOStage + j0 + lane);
          run += __popcll(mask);
        }
      }
    }
    nl = wave_sum_i(nl);
    nr = wave_sum_i(nr);
    // (a side none of the wave's members lies on has its four extents at their initial values in every lane:
    // nothing to reduce -- the usual case, and sixteen of this phase's twenty DPP reductions)
    if (nl > 0) {  // uniform
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        e[k] = wave_max_d(e[k]);
        e[k + 1] = wave_min_d(e[k + 1]);
      }
    }
    if (nr > 0) {  // uniform
#pragma unroll
      for (int k = 4; k < 8; k += 2) {
        e[k] = wave_max_d(e[k]);
        e[k + 1] = wave_min_d(e[k + 1]);
      }
    }
    red_sel ^= 1;
    if (lane == 0) {
      S.red.i[red_sel][wid][0] = nl;
      S.red.i[red_sel][wid][1] = nr;
#pragma unroll
      for (int k = 0; k < 8; k++) S.red.d[red_sel][wid][k] = e[k];
    }
    __syncthreads();
    if (tid == 0) {
      nl = nr = 0;
      for (int k = 0; k < 8; k += 2) {
        e[k] = -__builtin_inf();
        e[k + 1] = __builtin_inf();
      }
      for (int wv = 0; wv < NW; wv++) {
        nl += S.red.i[red_sel][wv][0];
        nr += S.red.i[red_sel][wv][1];
        for (int k = 0; k < 8; k += 2) {
          const double a = S.red.d[red_sel][wv][k], b = S.red.d[red_sel][wv][k + 1];
          e[k] = (a > e[k]) ? a : e[k];
          e[k + 1] = (b < e[k + 1]) ? b : e[k + 1];
        }
      }
      int label = 0;
      if (nl > 0 || nr > 0) label = 1;                              // antipodal.cpp:48-51
      if (nl > 0 && nr > 0) {                                       // :54-81
        const double top_y = (e[0] < e[4]) ? e[0] : e[4], bot_y = (e[1] > e[5]) ? e[1] : e[5];
        const double top_z = (e[2] < e[6]) ? e[2] : e[6], bot_z = (e[3] > e[7]) ? e[3] : e[7];
        if (top_y > bot_y && top_z > bot_z) label = 2;
      }
      ag2_hypothesis h;
      const double smp[3] = {S.smp[0], S.smp[1], S.smp[2]};
      const double ys[3] = {S.rec_surface, S.rec_bottom, S.rec_top};
      const double center = S.rec_center;
      double* dstv[3] = {h.surface, h.bottom, h.top};
      for (int k = 0; k < 3; k++)                                   // finger_hand.cpp:189-199
        for (int a = 0; a < 3; a++)
          dstv[k][a] = ((Fr[a][0] * center + Fr[a][1] * ys[k]) + Fr[a][2] * 0.0) + smp[a];
      for (int a = 0; a < 3; a++) {                                 // hand_search.cpp:383-385
        h.binormal[a] = Fr[a][0];
        h.approach[a] = Fr[a][1];
        h.axis[a] = Fr[a][2];
      }
      h.width = S.rec_width;
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
      // (the run's statistics -- hypotheses, their points, the largest list -- are gathered from the slot
      // table by k_hyp_stats afterwards: three atomics per pair on one cache line, issued by thousands
      // of workgroups at once, cost this kernel a third of its time)
    }
  }
}

// n_hyp, sum_p, max_p of a run from its slot table: 16 slot states per thread and load; the point
// counts of the occupied slots are read from the records.  A few atomics per workgroup.
__global__ void __launch_bounds__(256) k_hyp_stats(const unsigned char* __restrict__ keep,
                                                   const ag2_hypothesis* __restrict__ table, int n_slots,
                                                   DevStats* st) {
  const int s0 = (blockIdx.x * 256 + threadIdx.x) * 16;
  unsigned cnt = 0, mx = 0;
  unsigned long long sum = 0;
  if (s0 < n_slots) {  // (the state buffer is padded past n_slots: DevBuf slack)
    const uint4 v = *reinterpret_cast<const uint4*>(keep + s0);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const unsigned b = (w[k >> 2] >> (8 * (k & 3))) & 255u;
      if (b != 0u && s0 + k < n_slots) {
        const unsigned p = (unsigned)table[s0 + k].n_points;
        cnt++;
        sum += p;
        mx = max(mx, p);
      }
    }
  }
  cnt = (unsigned)wave_sum_i((int)cnt);
  const unsigned slo = (unsigned)wave_sum_i((int)(unsigned)(sum & 0xFFFFFFu));  // < 2^24 per lane: no overflow
  const unsigned shi = (unsigned)wave_sum_i((int)(unsigned)(sum >> 24));
  mx = (unsigned)(-wave_min_i(-(int)mx));
  if (lane_id() == 0 && cnt) {
    atomicAdd(&st->n_hyp, cnt);
    atomicAdd(&st->sum_p, (unsigned long long)slo + ((unsigned long long)shi << 24));
    atomicMax(&st->max_p, mx);
  }
}

int launch_sweep_orient(ag2_ctx* c, const SweepArgs& A, size_t n_slots) {
  // the queue length is read on the device: a fixed launch, the surplus workgroups leave at once
  const int grid = (int)std::min<size_t>(std::max<size_t>(n_slots, 1), 256 * 8);
  hipLaunchKernelGGL(k_sweep_orient, dim3(grid), dim3(kOThreads), 0, c->stream, A);
  // the statistics: along with the slot compaction when the caller has announced one (one launch less)
  if (c->defer_hyp_stats) {
    c->hyp_stats_pending = true;
    AG2_HIP(c, hipGetLastError());
    return 0;
  }
  return launch_hyp_stats(c, n_slots);
}

int launch_hyp_stats(ag2_ctx* c, size_t n_slots) {
  hipLaunchKernelGGL(k_hyp_stats, dim3((unsigned)((n_slots + 4095) / 4096)), dim3(256), 0, c->stream,
                     c->d_tab_keep.as<unsigned char>(), c->d_table.as<ag2_hypothesis>(), (int)n_slots,
                     c->d_stats.as<DevStats>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
