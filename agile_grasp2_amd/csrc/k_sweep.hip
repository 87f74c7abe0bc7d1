// k_sweep.hip -- K2 local frames and K3 the hand sweep.
//
// K2 k_frames replaces HandSearch::calculateLocalFrames (src/agile_grasp2/hand_search.cpp:97-170,
// :238-317) + LocalFrame::findAverageNormalAxis (local_frame.cpp:26-59): one 64-lane wave per
// sample; the <= 50 drawn normals live one per lane, the 50x50 Gram column sums are one column per
// lane, the 3x3 eigen-problem is solved redundantly in every lane.
//
// K3 k_sweep replaces HandSearch::evaluateHands (hand_search.cpp:173-235), calculateHand
// (:319-426), FingerHand (finger_hand.cpp:17-214, :313-325) and Antipodal::evaluateGrasp
// (antipodal.cpp:8-84): persistent 256-thread workgroups, one sample at a time; the radius
// neighbourhood is read as contiguous float4 row spans of the sorted cloud (coalesced), cropped to
// the +-hand_height slab and staged in LDS (24 B / point); every orientation then runs as a few
// passes over the LDS list with wavefront ballots / shuffles + one LDS hop for the block-wide
// reductions (finger-slot occupancy mask, back-of-hand collision, deepen step, closing-region
// extents, antipodal extents).  All geometry is f64 with the op order of ag2_device.h.
#include "ag2_internal.h"
#include "k_sweep_common.h"

namespace ag2 {

// Threads per workgroup of the sweep's stages.  The sweep is bound by the latencies of its ~40
// dependent phases per sample (SQ: waves wait > 50 % of their cycles), not by issue: halving the
// threads of a workgroup costs 3-4 %, doubling the samples in flight per CU gains 12 %.  Stage 0
// therefore runs FOUR 256-thread workgroups per CU, and so does the stage for long lists.
constexpr int kSweepThreads0 = 256;   // stage 0
constexpr int kSweepThreads1 = 256;   // stage 1 (long lists, global scratch)
constexpr int sweep_threads(int stage) { return stage == 0 ? kSweepThreads0 : kSweepThreads1; }

// ---------------------------------------------------------------------------------------------
// sample queries: (x, y, z, valid) per sample
// ---------------------------------------------------------------------------------------------
// With tab_keep the kernel also clears what a hypothesis run starts from -- the slot states of its
// sample and (thread 0) the per-run statistics -- instead of two separate fills.
__global__ void k_sample_queries_idx(const int* __restrict__ idx, int s,
                                     const float4* __restrict__ xyz_in, int n,
                                     float4* __restrict__ q, unsigned char* __restrict__ tab_keep,
                                     int R, DevStats* __restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (tab_keep && blockIdx.x == 0) {  // the per-run statistics: one word per thread
    unsigned* w = reinterpret_cast<unsigned*>(st);
    for (int k = threadIdx.x; k < (int)(offsetof(DevStats, bounds) / 4); k += blockDim.x) w[k] = 0u;
  }
  if (i >= s) return;
  if (tab_keep)
    for (int k = 0; k < R; k++) tab_keep[(size_t)i * R + k] = 0;
  const int id = idx[i];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (id >= 0 && id < n) {
    const float4 p = xyz_in[id];
    if (finite3(p.x, p.y, p.z)) o = make_float4(p.x, p.y, p.z, 1.f);
  }
  q[i] = o;
}

// ---------------------------------------------------------------------------------------------
// K2: local frames, one wave per sample
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_frames(const float4* __restrict__ pts,
                                               const float4* __restrict__ nrm,
                                               const unsigned* __restrict__ cell, GridDesc g_arg,
                                               const GridDesc* __restrict__ gp,
                                               const HandConst* __restrict__ hc,
                                               const float4* __restrict__ sample_q, int s,
                                               unsigned long long slot_base,
                                               unsigned long long seed,
                                               const FrameArgs* __restrict__ fa,
                                               double* __restrict__ frames,
                                               int* __restrict__ frame_ok, DevStats* st) {
  // frame mode: grid description and the per-frame scalars come from memory (see FrameArgs)
  const GridDesc g = gp ? *gp : g_arg;
  if (fa) {
    seed = fa->seed;
    slot_base = fa->slot_base;
  }
  __shared__ int buf[64];
  __shared__ double Nsh[50][3];
  __shared__ double Msh[6];
  const int t = blockIdx.x;
  if (t >= s) return;
  const int lane = lane_id();
  const float4 q = sample_q[t];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  QueryRange qr = query_range(g, q.x, q.y, q.z, hc->rq_taubin);
  const bool usable = (q.w != 0.f) && !qr.empty;
  const float r2 = hc->r2_taubin;
  // Fast path (the usual neighbourhood of <= 256 candidates in <= 64 stencil rows): the kernel is
  // bound by its chain of dependent loads, so the chain is cut to three -- all row bounds at once
  // (one row per lane), all candidate points at once (one per lane), their normals -- and the
  // finite-normal neighbours are kept in LDS in canonical order, which makes the resolution of the
  // drawn ranks a table lookup instead of a second walk.
  __shared__ int nbuf[256];
  __shared__ int row_first[64], row_begin[64];
  int k1f = 0;
  bool fast = false;
  if (usable) {
    const int ny = qr.hi[1] - qr.lo[1] + 1, nz = qr.hi[2] - qr.lo[2] + 1;
    const int nrows = ny * nz;
    if (nrows <= 64) {
      int rb = 0, rl = 0;
      if (lane < nrows) {
        const int cz = qr.lo[2] + lane / ny, cy = qr.lo[1] + lane % ny;  // canonical row order
        const int rowbase = (cz * g.dims[1] + cy) * g.dims[0];
        rb = (int)cell[rowbase + qr.lo[0]];
        rl = (int)cell[rowbase + qr.hi[0] + 1] - rb;
      }
      int inc = rl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      const int total = __builtin_amdgcn_readlane(inc, 63);
      if (total <= 256) {
        fast = true;
        row_first[lane] = inc - rl;
        row_begin[lane] = rb;
        __syncthreads();
        // all (up to four) candidates of a lane are requested together, points and normals alike: one
        // round trip instead of two per group of 64 (a normal is read whether or not its point turns
        // out to lie within the radius -- a few bytes more for a shorter chain)
        int jj[4];
        float4 pp[4], nn4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int cidx = 64 * u + lane;
          jj[u] = -1;
          pp[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          nn4[u] = pp[u];
          if (cidx < total) {
            int row = 0;
            for (int r = 1; r < nrows; r++) row = (row_first[r] <= cidx) ? r : row;
            jj[u] = row_begin[row] + (cidx - row_first[row]);
            pp[u] = pts[jj[u]];
            nn4[u] = nrm[jj[u]];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (64 * u < total) {  // uniform
            bool pred = false;
            if (jj[u] >= 0) {
              const float dx = pp[u].x - q.x, dy = pp[u].y - q.y, dz = pp[u].z - q.z;
              const float d2 = (dx * dx + dy * dy) + dz * dz;
              pred = (d2 < r2) && finite3(nn4[u].x, nn4[u].y, nn4[u].z);
            }
            const unsigned long long mask = __ballot(pred);
            if (pred) nbuf[k1f + __popcll(mask & lt_mask)] = jj[u];
            k1f += __popcll(mask);
          }
        }
        __syncthreads();
      }
    }
  }
  // pass 1: count neighbours with a finite normal (NaN-normal neighbours never enter the draw)
  if (usable && !fast) {
    for (int cz = qr.lo[2]; cz <= qr.hi[2]; cz++)
      for (int cy = qr.lo[1]; cy <= qr.hi[1]; cy++) {
        const int rowbase = (cz * g.dims[1] + cy) * g.dims[0];
        const int b = (int)cell[rowbase + qr.lo[0]], e = (int)cell[rowbase + qr.hi[0] + 1];
        for (int j0 = b; j0 < e; j0 += 64) {
          const int j = j0 + lane;
          bool pred = false;
          if (j < e) {
            const float4 p = pts[j];
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < r2) {
              const float4 nn = nrm[j];
              pred = finite3(nn.x, nn.y, nn.z);
            }
          }
          k1f += __popcll(__ballot(pred));
        }
      }
  }
  if (k1f == 0) {  // hand_search.cpp:122 (no neighbours => no frame)
    if (lane == 0) frame_ok[t] = 0;
    return;
  }
  const int m = min(50, k1f);                                        // :124-125
  unsigned long long rj = 0;
  if (lane < m) rj = draw_u64(seed, slot_base + (unsigned long long)t, (unsigned long long)lane) %
                     (unsigned long long)k1f;                        // :130
  // pass 2: resolve the drawn ranks to sorted positions
  int pick = -1;
  int base = 0;
  if (fast) {
    if (lane < m) pick = nbuf[(int)rj];
  } else
  for (int cz = qr.lo[2]; cz <= qr.hi[2]; cz++)
    for (int cy = qr.lo[1]; cy <= qr.hi[1]; cy++) {
      const int rowbase = (cz * g.dims[1] + cy) * g.dims[0];
      const int b = (int)cell[rowbase + qr.lo[0]], e = (int)cell[rowbase + qr.hi[0] + 1];
      for (int j0 = b; j0 < e; j0 += 64) {
        const int j = j0 + lane;
        bool pred = false;
        if (j < e) {
          const float4 p = pts[j];
          const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
          const float d2 = (dx * dx + dy * dy) + dz * dz;
          if (d2 < r2) {
            const float4 nn = nrm[j];
            pred = finite3(nn.x, nn.y, nn.z);
          }
        }
        const unsigned long long mask = __ballot(pred);
        const int cnt = __popcll(mask);
        if (cnt) {
          __syncthreads();
          if (pred) buf[__popcll(mask & lt_mask)] = j;
          __syncthreads();
          if (lane < m && (long long)rj >= base && (long long)rj < base + cnt)
            pick = buf[(int)rj - base];
          base += cnt;
        }
      }
    }
  // drawn normals, one per lane; camera votes (:137-146)
  double nx = 0.0, ny = 0.0, nz = 0.0;
  int cam_bits = 0;
  if (lane < m) {
    const float4 nn = nrm[pick];
    const double vx = (double)nn.x, vy = (double)nn.y, vz = (double)nn.z;
    const double mag = __builtin_sqrt((vx * vx + vy * vy) + vz * vz);  // :148-149
    nx = vx / mag;
    ny = vy / mag;
    nz = vz / mag;
    cam_bits = __float_as_int(pts[pick].w);
    Nsh[lane][0] = nx;
    Nsh[lane][1] = ny;
    Nsh[lane][2] = nz;
  }
  const int votes0 = __popcll(__ballot(lane < m && (cam_bits & 1)));
  const int votes1 = __popcll(__ballot(lane < m && (cam_bits & 2)));
  const int majority = (hc->n_cams > 1 && votes1 > votes0) ? 1 : 0;
  __syncthreads();
  // local_frame.cpp:29 M = N N^T, each entry summed over the draws in order
  if (lane < 6) {
    const int a = (lane < 3) ? 0 : (lane < 5 ? 1 : 2);
    const int b = (lane < 3) ? lane : (lane < 5 ? lane - 2 : 2);
    double acc = 0.0;
    for (int j = 0; j < m; j++) acc = acc + Nsh[j][a] * Nsh[j][b];
    Msh[lane] = acc;
  }
  // :42 column sums of (N^T N)^6
  double colsum = -2.0;
  if (lane < m) {
    double acc = 0.0;
    for (int i = 0; i < m; i++) {
      const double gij = (Nsh[i][0] * nx + Nsh[i][1] * ny) + Nsh[i][2] * nz;
      const double g2 = gij * gij;
      acc = acc + (g2 * g2) * g2;
    }
    colsum = (acc == acc) ? acc : -2.0;  // NaN never wins (oracle: "acc > best" is false)
  }
  int best_i = (lane < m) ? lane : 0x7fffffff;
  double best_v = colsum;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(best_v, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ov > best_v || (ov == best_v && oi < best_i)) {
      best_v = ov;
      best_i = oi;
    }
  }
  if (!(best_v > -1.0)) best_i = 0;
  __syncthreads();
  // The rest -- a 3x3 eigen-solve and a dozen vector operations -- is serial per sample: done here it
  // would cost a whole wave per sample (the kernel was issue-bound on exactly that), so the
  // intermediate (M, n_max, majority camera) is parked in the sample's frame slot and
  // k_frames_finish completes it with one THREAD per sample.
  if (lane == 0) {
    double* f = frames + (size_t)t * 12;
#pragma unroll
    for (int k = 0; k < 6; k++) f[k] = Msh[k];
    f[6] = Nsh[best_i][0];
    f[7] = Nsh[best_i][1];
    f[8] = Nsh[best_i][2];
    f[9] = (double)majority;
    frame_ok[t] = 1;
  }
}

__global__ void __launch_bounds__(64) k_frames_finish(const HandConst* __restrict__ hc,
                                                      const float4* __restrict__ sample_q, int s,
                                                      double* __restrict__ frames,
                                                      const int* __restrict__ frame_ok, DevStats* st) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = (t < s) && frame_ok[t] != 0;
  const unsigned long long lm = __ballot(live);
  if (lane_id() == 0 && lm) atomicAdd(&st->n_frames, (unsigned)__popcll(lm));
  if (!live) return;
  const float4 q = sample_q[t];
  double* fio = frames + (size_t)t * 12;
  Sym3 M;
  M.a00 = fio[0]; M.a01 = fio[1]; M.a02 = fio[2]; M.a11 = fio[3]; M.a12 = fio[4]; M.a22 = fio[5];
  const double nsel[3] = {fio[6], fio[7], fio[8]};
  const int majority = (int)fio[9];
  const int lane = 0, best_i = 0;
  const double (*Nsh)[3] = &nsel;  // Nsh[best_i] below reads the parked n_max
  (void)lane;
  const Eig3 e = jacobi3(M);
  const int mi = argmin3(e.d);                                        // :36-38
  V3 cv{mi == 0 ? e.v[0][0] : (mi == 1 ? e.v[0][1] : e.v[0][2]),
        mi == 0 ? e.v[1][0] : (mi == 1 ? e.v[1][1] : e.v[1][2]),
        mi == 0 ? e.v[2][0] : (mi == 1 ? e.v[2][1] : e.v[2][2])};
  const double ncv = norm3(cv);
  cv = V3{cv.x / ncv, cv.y / ncv, cv.z / ncv};
  const double nm0 = Nsh[best_i][0], nm1 = Nsh[best_i][1], nm2 = Nsh[best_i][2];
  // :43-45 normal = normalize((I - c c^T) n_max)
  const double cc[3] = {cv.x, cv.y, cv.z};
  double np_[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double p0 = ((a == 0) ? 1.0 : 0.0) - cc[a] * cc[0];
    const double p1 = ((a == 1) ? 1.0 : 0.0) - cc[a] * cc[1];
    const double p2 = ((a == 2) ? 1.0 : 0.0) - cc[a] * cc[2];
    np_[a] = (p0 * nm0 + p1 * nm1) + p2 * nm2;
  }
  const V3 npv{np_[0], np_[1], np_[2]};
  const double nn_ = norm3(npv);
  V3 normal{npv.x / nn_, npv.y / nn_, npv.z / nn_};
  V3 binormal = cross3(cv, normal);                                   // :48
  const V3 sample{(double)q.x, (double)q.y, (double)q.z};
  const V3 v{sample.x - hc->cam_origin[majority][0], sample.y - hc->cam_origin[majority][1],
             sample.z - hc->cam_origin[majority][2]};                 // :51
  if (dot3(normal, v) > 0.0) normal = neg3(normal);                   // :52-53
  if (dot3(binormal, v) > 0.0) binormal = neg3(binormal);             // :54-55
  const V3 curv = cross3(normal, binormal);                           // :58
  fio[0] = sample.x; fio[1] = sample.y; fio[2] = sample.z;
  fio[3] = normal.x; fio[4] = normal.y; fio[5] = normal.z;
  fio[6] = binormal.x; fio[7] = binormal.y; fio[8] = binormal.z;
  fio[9] = curv.x; fio[10] = curv.y; fio[11] = curv.z;
}

// ---------------------------------------------------------------------------------------------
// K3: hand sweep
// ---------------------------------------------------------------------------------------------
// phase stamps for the diagnostic build of the launch (thread 0 of each workgroup)
// AG2_SWEEP_DIAG=1 (compile time, tools/ab_build.sh): pass A also counts its orientation-steps -- a branch per step
// that costs the product kernel 2 us, so it is not compiled in by default
#ifndef AG2_SWEEP_DIAG
#define AG2_SWEEP_DIAG 0
#endif
#define AG2_PROF(i)                                                    \
  do {                                                                 \
    if (A.prof && tid == 0) {                                          \
      const long long _t = clock64();                                  \
      atomicAdd(&A.prof[i], (unsigned long long)(_t - tprev));         \
      tprev = _t;                                                      \
    }                                                                  \
  } while (0)

constexpr int kMaxPieces = 1024;
#ifndef AG2_SWEEP_GROUP
#define AG2_SWEEP_GROUP 8
#endif
constexpr int kGrp = AG2_SWEEP_GROUP;  // lanes that share one piece in the crop passes (tuning knob)
constexpr int kGpw = kWave / kGrp;     // pieces per wave-wide load
static_assert(kGrp >= 4 && kGrp <= 64 && (kGrp & (kGrp - 1)) == 0, "lane group must be a power of two");

template <int NT>
struct SweepShared {
  static constexpr int NW = NT / kWave;
  int row_start[kMaxRows];       // per stencil row (cy, cz): first sorted position, length
  int row_len[kMaxRows];
  int piece_start[kMaxPieces];   // rows cut into pieces of <= PL points
  // Stage 0 (pieces <= 32 points, lists <= kGposCap): piece_lc = length (low 16 bits) | survivor
  // count, then list offset (high 16); piece_mask = which points of the piece survive the crop.
  // Stage 1 (no bound on either): piece_lc = length, piece_mask = survivor count, then list offset.
  unsigned piece_lc[kMaxPieces];
  unsigned piece_mask[kMaxPieces];
  double fs[20], fsr[20];        // finger-slot table (exact path of pass A, deepen)
  double depths[kMaxDepths];
  double cosd[kMaxOrient], sind[kMaxOrient];    // hand angles, f64 (exact path)
  float cosf_t[kMaxOrient], sinf_t[kMaxOrient]; // and rounded to f32 (classification)
  unsigned res_a[NW][kMaxOrient][2];            // pass A per wave: slot mask, flags
  unsigned exact_a[NW][kMaxOrient][2];          // pass A, exact path (a few points per thousand): slot bits, flags,
                                                // ORed here by the lanes that take it instead of living in 2 x RMAX
                                                // registers per lane (RMAX = 16: they were the kernel's scratch)
  Red<NW> red;
  int wave_cnt[NW + 1];
  long long arena_off;
  int wrun[NW];                  // ... survivors each wave wrote
  int flag;
  unsigned dead;                 // pass A: orientations known to have a point behind the hand
  int next_w[2];                 // work item of the next loop iteration (double-buffered)
};

// Stage 0 keeps only the sorted POSITION of every cropped point in LDS (4 B + the 2-B in-box index)
// and re-reads the point from the sorted cloud in L2 where a pass needs it (p - q in float: the very
// value the crop computed).  The list costs a third of what staged coordinates would, so four
// workgroups fit a CU with room for ~4 000 points each.
constexpr int kStage0WgPerCu = 4;  // (five: 96 VGPRs, spills, 0.153 -> 0.172 ms)
constexpr int kStage0PointBytes = 6;
constexpr size_t sweep_ctl_bytes(int stage) {
  return stage == 0 ? ((sizeof(SweepShared<kSweepThreads0>) + 15) & ~size_t(15))
                    : ((sizeof(SweepShared<kSweepThreads1>) + 15) & ~size_t(15));
}
constexpr int kLdsCapRaw = (int)(((163840 / kStage0WgPerCu - sweep_ctl_bytes(0)) / kStage0PointBytes) & ~size_t(31));
constexpr int kLdsCap = kLdsCapRaw > 32752 ? 32752 : kLdsCapRaw;
static_assert(kLdsCap >= 2048, "unexpected LDS stage size");
// A list that is longer than stage 0's LDS but not by much keeps its positions in a per-workgroup
// global slice instead (L2-resident, same code): the few such samples of a tabletop cloud would
// otherwise cost a whole extra launch whose duration is one sample's latency.
// The slice's capacity is a launch argument (SweepArgs::gpos_cap), chosen by the host from what the previous
// run of the context saw (results do not depend on it: a routing decision):
//   kGposCap     clouds on which many samples overflow anyway (configuration 3: 85 % of them; A/B there: 10 k,
//                24 k and 32 k are all slower than ~7.4 k -- the long-list stage crops such lists in one pass)
//   kGposCapBig  clouds on which only a few samples per run are that long: they stay in this stage, among
//                the other samples, instead of costing a launch of their own whose duration is one long
//                sample's latency (55 us per frame at configuration 5)
constexpr int kGposCap = 2 * kLdsCap;
constexpr int kGposCapBig = 16384;
static_assert(kGposCapBig <= 65535, "list offsets inside a piece table entry are 16 bits");

// Two instantiations run back to back; the second reads its queue length on the device (no host
// round trip):
//   STAGE 0: every sample; positions of the cropped list in 40 KiB of LDS (four 256-thread
//            workgroups per CU), or in the workgroup's global slice up to kGposCap points
//   STAGE 1: the samples stage 0 handed on (dense clouds); centred coordinates + positions of the
//            cropped list in a per-workgroup global scratch (four 256-thread workgroups per CU)
// RMAX: compile-time bound on num_orientations (8, 16 or 32) for the per-orientation registers
// The kernel stops at the gates: the sample's list goes to the list arena, its passing orientations to
// the pair queue of k_sweep_orient (k_sweep_orient.hip).  (The one-kernel form of rounds 1-2, which ran
// the per-orientation passes here, lives on in tools/experiments/ only.)
template <int STAGE, int RMAX>
__global__ void __launch_bounds__(sweep_threads(STAGE),
                                  (STAGE == 0 ? kStage0WgPerCu : 4) * sweep_threads(STAGE) / 256)
k_sweep(SweepArgs A) {
  constexpr int NT = sweep_threads(STAGE), NW = NT / kWave;
  constexpr int kRowsPerThread = (kMaxRows + NT - 1) / NT;
  constexpr int kPiecesPerThread = (kMaxPieces + NT - 1) / NT;
  constexpr bool LDS_STORE = STAGE == 0;
  constexpr bool LITE = STAGE == 0;  // positions only: points are re-read from the sorted cloud
  constexpr int kCapL = kLdsCap;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SweepShared<NT>& S = *reinterpret_cast<SweepShared<NT>*>(smem_raw);
  // Split sweep, stage 1: the cropped list is written ONCE, in the form k_sweep_orient reads (one
  // float4 per point in the list arena), and pass A streams it from there -- no per-workgroup scratch,
  // no copy at the gates, and no bound on the list length other than the arena's (which grows).
  constexpr bool ARENA = STAGE == 1;
  const int CAP = LITE ? A.gpos_cap : (LDS_STORE ? kCapL : (ARENA ? 0x7fffffff : A.gcap));
  float4* L = nullptr;  // ARENA: this sample's list
  float* pbase;
  // stage 0: sorted positions of the cropped list in LDS; stage 1: the list arena (L)
  if (LDS_STORE) {
    pbase = reinterpret_cast<float*>(smem_raw + sweep_ctl_bytes(STAGE));
  } else {
    pbase = ARENA ? nullptr : A.gscratch + (size_t)blockIdx.x * 5 * (size_t)A.gcap;
  }
  float* PX = pbase;
  float* PY = pbase + CAP;
  float* PZ = pbase + 2 * (size_t)CAP;
  int* POS = reinterpret_cast<int*>(pbase + (LITE ? 0 : 3) * (size_t)(LITE ? kCapL : CAP));
  int* gpos = nullptr;
  if (LITE) gpos = A.gpos + (size_t)blockIdx.x * A.gpos_cap;

  const HandConst& hc = *A.hc;
  // frame mode: grid description, cloud minimum and slot base come from memory (uniform loads)
  const GridDesc G = A.gp ? *A.gp : A.g;
  const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int R = hc.R;
  const double hh = hc.hand_height;
  const int n_work = (STAGE == 0) ? A.n_samples : (int)A.st->n_overflow;
  if (STAGE != 0 && n_work == 0) return;
  if (tid < 20) {
    S.fs[tid] = hc.fs[tid];
    S.fsr[tid] = hc.fsr[tid];
  }
  if (tid < kMaxDepths) S.depths[tid] = hc.depths[tid];
  if (tid < kMaxOrient) {
    S.cosd[tid] = hc.cos_t[tid];
    S.sind[tid] = hc.sin_t[tid];
    S.cosf_t[tid] = (float)hc.cos_t[tid];
    S.sinf_t[tid] = (float)hc.sin_t[tid];
  }
  const double slot_inv_step = hc.slot_inv_step, slot_step = hc.slot_step;
  const double hand_od = hc.hand_outer_diameter, finger_w = hc.finger_width;
  const double slot_ratio = hc.finger_width * hc.slot_inv_step;  // finger width in slot spacings
  const int slot_span = hc.slot_span;
  const double slot_base0 = hc.fs[0], slot_base1 = hc.fs[10];
  const float r2_hands = hc.r2_hands;
  const bool tighten = (A.flags & 1) == 0;

  // Work distribution: the cost of a sample varies with its neighbourhood, and a workgroup sees only
  // about ten of them, so a static stride leaves a long tail.  The first gridDim.x items are taken
  // by position, every further one from a device counter; thread 0 asks for the next item at the
  // START of the current one, so the atomic's latency is never waited for.  Every workgroup leaves
  // the loop with the first item >= n_work it draws.  Results go to fixed slots: the order in which
  // samples are processed changes nothing.
  unsigned nxt = 0;
  int it = 0;
  auto next_work = [&]() -> int {
    if (tid == 0) S.next_w[it & 1] = (int)nxt;
    __syncthreads();  // also: every reader of S of this iteration is done
    const int w = __builtin_amdgcn_readfirstlane(S.next_w[it & 1]);
    it++;  // the slot is rewritten two barriers from now: no reader can still be pending
    return w;
  };
  for (int w = blockIdx.x; w < n_work; w = next_work()) {
    if (tid == 0) nxt = gridDim.x + atomicAdd(&A.st->work_next[STAGE], 1u);
    const int t = (STAGE == 0) ? w : A.overflow[w];
    long long tprev = A.prof ? clock64() : 0;
    if (!A.frame_ok[t]) continue;  // uniform
    const float4 q = A.sample_q[t];
    bool gmode = false;  // stage 0: this sample's list lives in the global slice (set once K is known)
    // the list's segments (long-list stage: set by the crop; else one segment)
    int ge0 = 0x7fffffff, ge1 = 0x7fffffff, ge2 = 0x7fffffff, gs0 = 0, gs1 = 0, gs2 = 0;
    auto lslot = [&](int j) { return list_slot6(ge0, ge1, ge2, gs0, gs1, gs2, j); };
    auto pos_at = [&](int j) -> int {
      if (ARENA) return __float_as_int(L[lslot(j)].w);
      return (LITE && gmode) ? gpos[j] : POS[j];
    };
    auto ldp = [&](int j, float& x, float& y, float& z) {  // cropped point j, centred on the sample
      if (LITE) {
        const float4 p = A.pts[pos_at(j)];
        x = p.x - q.x;
        y = p.y - q.y;
        z = p.z - q.z;
      } else {
        if (ARENA) {
          const float4 v = L[lslot(j)];
          x = v.x;
          y = v.y;
          z = v.z;
        } else {
          x = PX[j];
          y = PY[j];
          z = PZ[j];
        }
      }
    };
    const double* fr = A.frames + (size_t)t * 12;
    // frame = [normal binormal curvature_axis] as columns, hand_search.cpp:325-326
    const double F[3][3] = {{fr[3], fr[6], fr[9]}, {fr[4], fr[7], fr[10]}, {fr[5], fr[8], fr[11]}};

    // ---- row table: one contiguous span of the sorted cloud per (cy, cz) -------------------
    const QueryRange qr = query_range(G, q.x, q.y, q.z, hc.rq_hands);
    const int ny = qr.empty ? 0 : (qr.hi[1] - qr.lo[1] + 1);
    const int nz = qr.empty ? 0 : (qr.hi[2] - qr.lo[2] + 1);
    const int nrows = ny * nz;  // <= kMaxRows by check_params
    // (the previous sample's readers of S are done: barrier in next_work)
    for (int r = tid; r < nrows; r += NT) {
      const int cz = qr.lo[2] + r / ny, cy = qr.lo[1] + r % ny;
      int cxa = qr.lo[0], cxb = qr.hi[0];
      if (tighten) tighten_row(G, hc, q, F, hh, cy, cz, cxa, cxb);
      const int rowbase = (cz * G.dims[1] + cy) * G.dims[0];
      int b = 0, e = 0;
      if (cxa <= cxb) {
        b = (int)A.cell[rowbase + cxa];
        e = (int)A.cell[rowbase + cxb + 1];
      }
      S.row_start[r] = b;
      S.row_len[r] = e - b;
    }
    __syncthreads();
    // ---- piece table: rows cut into pieces of <= PL consecutive points, in canonical order -----
    // A small group of lanes takes one piece in the crop passes, so every wave has many independent
    // loads in flight (the rows are short after the culling above; a wave per row would idle most
    // lanes).
    int nrows_c = 0, kcand = 0;
    {
      int cnt = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < kRowsPerThread; k++) {
        const int r = tid * kRowsPerThread + k;
        const int l = (r < nrows) ? S.row_len[r] : 0;
        cnt += (l > 0) ? 1 : 0;
        tot += l;
      }
      cnt = wave_sum_i(cnt);
      tot = wave_sum_i(tot);
      if (lane == 0) {
        S.wave_cnt[wid] = cnt;
        S.red.i[0][wid][0] = tot;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NW; k++) {
        nrows_c += S.wave_cnt[k];
        kcand += S.red.i[0][k][0];
      }
    }
    if (STAGE == 0 && kcand > 2 * A.gpos_cap) {
      // Dense neighbourhood: the cropped list is about 0.6 of the candidates, so it will not fit
      // this stage -- hand the sample to the global-scratch stage before the crop pass instead of
      // after it.  Only a routing decision: both stages compute the same result.
      if (tid == 0) {
        const unsigned at = atomicAdd(&A.st->n_overflow, 1u);
        A.overflow[at] = t;
      }
      continue;
    }
    int PL = kGrp, n_pieces = 0, K = 0;
    if constexpr (!ARENA) {
    PL = kGrp;  // piece length; grows only if the table would overflow (dense clouds)
    while (kcand / PL + nrows_c > kMaxPieces) PL <<= 1;
    if (LITE && PL > 32) {  // (cannot happen below the hand-on threshold above; the masks are 32 bits)
      if (tid == 0) {
        const unsigned at = atomicAdd(&A.st->n_overflow, 1u);
        A.overflow[at] = t;
      }
      continue;
    }
    {
      int len[kRowsPerThread], st[kRowsPerThread], np = 0;
#pragma unroll
      for (int k = 0; k < kRowsPerThread; k++) {
        const int r = tid * kRowsPerThread + k;
        len[k] = (r < nrows) ? S.row_len[r] : 0;
        st[k] = (r < nrows) ? S.row_start[r] : 0;
        np += (len[k] + PL - 1) / PL;
      }
      int inc = np;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      __syncthreads();  // totals above have been read by everyone
      if (lane == 63) S.wave_cnt[wid] = inc;
      __syncthreads();
      int poff = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        if (k < wid) poff += S.wave_cnt[k];
        n_pieces += S.wave_cnt[k];
      }
      int pi = poff + inc - np;
#pragma unroll
      for (int k = 0; k < kRowsPerThread; k++)
        for (int o = 0; o < len[k]; o += PL) {
          S.piece_start[pi] = st[k] + o;
          S.piece_lc[pi] = (unsigned)min(PL, len[k] - o);  // low 16: length, high 16: count/offset
          pi++;
        }
      __syncthreads();
    }
    }
    AG2_PROF(0);

    // ---- crop to the +-hand_height slab, ordered compaction --------------------------------
    const float cropn[3] = {(float)F[0][2], (float)F[1][2], (float)F[2][2]};
    const float crop_hh = (float)hh;
    auto classify = [&](const float4& p, float4& d) -> int {  // 0 miss, 1 in radius, 3 + in slab
      d = make_float4(p.x - q.x, p.y - q.y, p.z - q.z, 0.f);
      const float d2 = (d.x * d.x + d.y * d.y) + d.z * d.z;
      if (!(d2 < r2_hands)) return 0;
      {  // decided from an f32 estimate unless it is within 2e-6 m of the slab faces (error < 1e-7 m)
        const float ze = __builtin_fabsf((cropn[0] * d.x + cropn[1] * d.y) + cropn[2] * d.z);
        if (ze < crop_hh - 2.0e-6f) return 3;
        if (ze > crop_hh + 2.0e-6f) return 1;
      }
      // hand_search.cpp:209-210 centred in float then widened; :329-339 crop on row 2 of frame^T p
      const double p0 = (double)d.x, p1 = (double)d.y, p2 = (double)d.z;
      const double zf = (F[0][2] * p0 + F[1][2] * p1) + F[2][2] * p2;
      return (zf > -1.0 * hh && zf < hh) ? 3 : 1;
    };
    if constexpr (!ARENA) {
    // pass 1: survivors per piece.  A GROUP of kGrp lanes takes one piece (rows of a surface-sheet
    // cloud hold ~8-10 candidates after the culling, so wider groups would idle most lanes), one
    // load instruction covers 64 / kGrp pieces with 16-B-per-lane contiguous reads, and four such
    // instructions are issued before the first result is needed (the path is latency-bound).
    const int grp = lane / kGrp, lg = lane % kGrp;
    const unsigned long long grp_mask = ((kGrp == 64) ? ~0ull : ((1ull << kGrp) - 1ull)) << (grp * kGrp);
    const int n_sets = (n_pieces + kGpw - 1) / kGpw;  // one set = the pieces of one wave-wide load
    int my_k2 = 0;
    for (int pp0 = wid; pp0 < n_sets; pp0 += 4 * NW) {
      int pb[4], pl[4];
      float4 pv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int pc = kGpw * (pp0 + u * NW) + grp;
        const bool ok = pc < n_pieces;
        pb[u] = ok ? S.piece_start[pc] : 0;
        pl[u] = ok ? (int)(LITE ? (S.piece_lc[pc] & 0xFFFFu) : S.piece_lc[pc]) : 0;
        // unconditional load from a clamped position (lanes beyond the piece re-read its last point,
        // an empty slot reads point 0): no branch around the load, so the four loads of this loop are
        // issued back to back and waited for once -- the pass is latency-bound
        pv[u] = A.pts[pb[u] + max(min(lg, pl[u] - 1), 0)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int pc = kGpw * (pp0 + u * NW) + grp;
        float4 d;
        int cls = (lg < pl[u]) ? classify(pv[u], d) : 0;
        unsigned long long bal = __ballot(cls == 3) & grp_mask;
        int keep = __popcll(bal);
        unsigned pm = (unsigned)(bal >> (grp * kGrp));  // survivors of the piece, bit = position in it
        my_k2 += (cls != 0) ? 1 : 0;
        for (int o = kGrp; o < pl[u]; o += kGrp) {  // pieces longer than a group (dense clouds)
          const int j = o + lg;
          cls = (j < pl[u]) ? classify(A.pts[pb[u] + j], d) : 0;
          bal = __ballot(cls == 3) & grp_mask;
          keep += __popcll(bal);
          if (LITE) pm |= (unsigned)(bal >> (grp * kGrp)) << (o & 31);
          my_k2 += (cls != 0) ? 1 : 0;
        }
        if (lg == 0 && pc < n_pieces) {
          if (LITE) {
            S.piece_lc[pc] = (unsigned)pl[u] | ((unsigned)keep << 16);
            S.piece_mask[pc] = pm;
          } else {
            S.piece_mask[pc] = (unsigned)keep;
          }
        }
      }
    }
    my_k2 = wave_sum_i(my_k2);
    if (lane == 0) S.red.i[0][wid][0] = my_k2;
    __syncthreads();
    AG2_PROF(1);
    // exclusive scan of the per-piece counts (thread t owns pieces 4t .. 4t+3)
    int k2 = 0;
    {
      int c[kPiecesPerThread], tot = 0;
#pragma unroll
      for (int k = 0; k < kPiecesPerThread; k++) {
        const int pc = tid * kPiecesPerThread + k;
        c[k] = (pc < n_pieces) ? (int)(LITE ? (S.piece_lc[pc] >> 16) : S.piece_mask[pc]) : 0;
        tot += c[k];
      }
      int inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (lane == 63) S.wave_cnt[wid] = inc;
      __syncthreads();
      int woff = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        if (k < wid) woff += S.wave_cnt[k];
        K += S.wave_cnt[k];
        k2 += S.red.i[0][k][0];
      }
      if (K <= CAP) {  // (stage 0: offsets fit 16 bits, kGposCap <= 65536 and an offset is < K)
        int run = woff + inc - tot;
#pragma unroll
        for (int k = 0; k < kPiecesPerThread; k++) {
          const int pc = tid * kPiecesPerThread + k;
          if (pc < n_pieces) {
            if (LITE) S.piece_lc[pc] = (S.piece_lc[pc] & 0xFFFFu) | ((unsigned)run << 16);
            else S.piece_mask[pc] = (unsigned)run;
          }
          run += c[k];
        }
      }
    }
    const bool too_big = K > CAP;
    gmode = LITE && K > kCapL;
    if (too_big) {  // uniform
      if (tid == 0) {
        if (STAGE == 0) {
          const unsigned at = atomicAdd(&A.st->n_overflow, 1u);
          A.overflow[at] = t;
        } else {
          // Longer than this launch's scratch: the host sizes the scratch to the longest list of the
          // run and repeats it (run_hypotheses) -- hand_search.cpp:329-349 crops into a list of any
          // length, so no list length is an error.
          atomicMax(&A.st->max_k_over, (unsigned)K);
          atomicOr(&A.st->err_flags, 8u);
        }
      }
      continue;
    }
    if (tid == 0) {
      // K2 (all radius neighbours) is only visited when row tightening is off (debug_flags bit 0)
      if (!tighten) atomicAdd(&A.st->sum_k2, (unsigned long long)k2);
      atomicAdd(&A.st->sum_kcrop, (unsigned long long)K);
      S.dead = 0u;  // read in pass A, two barriers further down
    }
    if (K == 0) continue;  // hand_search.cpp:201 (no neighbours) / no cropped points => no fingers
    if (ARENA) {  // room for the list in the arena (kept only if an orientation passes the gates; the
                  // arena is a bump allocator, so the others' room is simply not used again this run)
      if (tid == 0) {
        long long loff = (long long)atomicAdd(&A.st->list_top, (unsigned long long)K);
        if (loff + K > A.list_cap) {
          atomicOr(&A.st->err_flags, 2u);  // the host grows the list arena and repeats the run
          loff = -1;
        }
        S.arena_off = loff;
      }
      __syncthreads();
      if (S.arena_off < 0) continue;  // uniform
      L = A.lists + S.arena_off;
    }
    __syncthreads();
    // pass 2: write the survivors of each piece at its offset (order within a piece preserved).
    // Stage 0 stores positions only, and pass 1 left every piece's survivors as a bit mask: the list
    // is written from the masks, without touching the points again.
    if (LITE) {
      for (int pc = tid; pc < n_pieces; pc += NT) {
        unsigned m = S.piece_mask[pc];
        int dst = (int)(S.piece_lc[pc] >> 16);
        const int st = S.piece_start[pc];
        while (m) {
          const int b = __ffs((int)m) - 1;
          m &= m - 1u;
          if (gmode) gpos[dst] = st + b; else POS[dst] = st + b;
          dst++;
        }
      }
    }
    const unsigned long long lt_grp = lt_mask & grp_mask;  // lower lanes of my lane group
    for (int pp0 = wid; !LITE && pp0 < n_sets; pp0 += 4 * NW) {
      int pb[4], pl[4], po[4];
      float4 pv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int pc = kGpw * (pp0 + u * NW) + grp;
        const bool ok = pc < n_pieces;
        pb[u] = ok ? S.piece_start[pc] : 0;
        pl[u] = ok ? (int)S.piece_lc[pc] : 0;
        po[u] = ok ? (int)S.piece_mask[pc] : 0;
        pv[u] = A.pts[pb[u] + max(min(lg, pl[u] - 1), 0)];  // (clamped, unconditional: as in pass 1)
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        int dst0 = po[u];
        float4 d;
        int j = lg;
        int cls = (j < pl[u]) ? classify(pv[u], d) : 0;
        for (int o = 0;;) {
          const unsigned long long mask = __ballot(cls == 3);
          if (cls == 3) {
            const int dst = dst0 + __popcll(mask & lt_grp);
            if (ARENA) {
              L[dst] = make_float4(d.x, d.y, d.z, __int_as_float(pb[u] + j));
            } else {
              PX[dst] = d.x; PY[dst] = d.y; PZ[dst] = d.z;
              POS[dst] = pb[u] + j;
            }
          }
          dst0 += __popcll(mask & grp_mask);
          o += kGrp;
          if (o >= pl[u]) break;
          j = o + lg;
          cls = (j < pl[u]) ? classify(A.pts[pb[u] + j], d) : 0;
        }
      }
    }
    }
    if constexpr (ARENA) {
      // ---- long lists: the crop in ONE pass -------------------------------------------------------
      // This stage is bound by bytes (a list of a dense cloud is tens of thousands of points), and
      // counting the survivors before writing them reads every candidate twice.  Here every wave
      // takes a contiguous quarter of the candidates (units of one row's 64 consecutive candidates: fully
      // coalesced loads, four in flight), and writes its survivors at once, in order,
      // into its own part of a reservation as long as the CANDIDATE list.  The list is then up to four
      // dense segments in canonical order (ListSegs); no piece table, no scan, no barrier in the crop.
      // Unit table (the piece tables' arrays, unused in this stage): a unit is up to UL consecutive
      // candidates of one row, UL = 64 unless the table would overflow; start, length and the number of
      // candidates in front of it (= its place in the reservation).
      int UL = 64;
      while (kcand / UL + nrows_c > kMaxPieces) UL <<= 1;
      int n_units = 0;
      {
        int len[kRowsPerThread], st[kRowsPerThread], ctot = 0, nu = 0;
#pragma unroll
        for (int k = 0; k < kRowsPerThread; k++) {
          const int r = tid * kRowsPerThread + k;
          len[k] = (r < nrows) ? S.row_len[r] : 0;
          st[k] = (r < nrows) ? S.row_start[r] : 0;
          ctot += len[k];
          nu += (len[k] + UL - 1) / UL;
        }
        int cinc = ctot, uinc = nu;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(cinc, o, 64), w2 = __shfl_up(uinc, o, 64);
          if (lane >= o) {
            cinc += v;
            uinc += w2;
          }
        }
        __syncthreads();  // the totals above have been read by everyone
        if (lane == 63) {
          S.wave_cnt[wid] = cinc;
          S.wrun[wid] = uinc;
        }
        if (tid == 0) {
          long long loff = (long long)atomicAdd(&A.st->list_top, (unsigned long long)kcand);
          if (loff + kcand > A.list_cap) {
            atomicOr(&A.st->err_flags, 2u);  // the host grows the list arena and repeats the run
            loff = -1;
          }
          S.arena_off = loff;
        }
        __syncthreads();
        int coff = 0, uoff = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          if (k < wid) {
            coff += S.wave_cnt[k];
            uoff += S.wrun[k];
          }
          n_units += S.wrun[k];
        }
        int cp = coff + cinc - ctot, ui = uoff + uinc - nu;
#pragma unroll
        for (int k = 0; k < kRowsPerThread; k++) {
          for (int o = 0; o < len[k]; o += UL) {
            S.piece_start[ui] = st[k] + o;
            S.piece_lc[ui] = (unsigned)min(UL, len[k] - o);
            S.piece_mask[ui] = (unsigned)(cp + o);
            ui++;
          }
          cp += len[k];
        }
        __syncthreads();
      }
      if (S.arena_off < 0) continue;  // uniform
      if (kcand == 0) continue;
      // wave w: units [w, w + 1) * n_units / NW, written from the place of its first unit on
      const int ubeg = (int)(((long long)wid * n_units) / NW), uend = (int)(((long long)(wid + 1) * n_units) / NW);
      const int seg0 = (ubeg < n_units) ? (int)S.piece_mask[ubeg] : kcand;
      float4* Lw = A.lists + S.arena_off + seg0;
      int run = 0, my_k2 = 0;
      for (int u0 = ubeg; u0 < uend; u0 += 4) {  // (wave-uniform)
        int ub[4], ul[4];
        float4 pv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool ok = u0 + u < uend;
          ub[u] = ok ? S.piece_start[u0 + u] : 0;
          ul[u] = ok ? (int)S.piece_lc[u0 + u] : 0;
          // unconditional load from a clamped position: the four loads are issued back to back
          pv[u] = A.pts[ub[u] + max(min(lane, ul[u] - 1), 0)];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          float4 d;
          int j = lane;
          int cls = (j < ul[u]) ? classify(pv[u], d) : 0;
          for (int o = 0;;) {
            const unsigned long long mask = __ballot(cls == 3);
            if (cls == 3) Lw[run + __popcll(mask & lt_mask)] = make_float4(d.x, d.y, d.z, __int_as_float(ub[u] + j));
            run += __popcll(mask);
            my_k2 += (cls != 0) ? 1 : 0;
            o += 64;
            if (o >= ul[u]) break;  // (units longer than a wave: very dense clouds only)
            j = o + lane;
            cls = (j < ul[u]) ? classify(A.pts[ub[u] + j], d) : 0;
          }
        }
      }
      __syncthreads();  // (the unit counts in S.wrun have been read by everyone)
      my_k2 = wave_sum_i(my_k2);
      if (lane == 0) {
        S.red.i[0][wid][0] = my_k2;
        S.wrun[wid] = run;
        S.wave_cnt[wid] = seg0;
      }
      __syncthreads();
      int k2 = 0;
      {
        int cum[NW + 1];
        cum[0] = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          cum[k + 1] = cum[k] + S.wrun[k];
          k2 += S.red.i[0][k][0];
        }
        K = cum[NW];
        static_assert(NW == 4, "ListSegs describes four segments");
        // (the same in every lane: scalar registers)
        ge0 = __builtin_amdgcn_readfirstlane(cum[1]);
        ge1 = __builtin_amdgcn_readfirstlane(cum[2]);
        ge2 = __builtin_amdgcn_readfirstlane(cum[3]);
        gs0 = __builtin_amdgcn_readfirstlane(S.wave_cnt[1] - cum[1]);
        gs1 = __builtin_amdgcn_readfirstlane(S.wave_cnt[2] - cum[2]);
        gs2 = __builtin_amdgcn_readfirstlane(S.wave_cnt[3] - cum[3]);
      }
      if (tid == 0) {
        if (!tighten) atomicAdd(&A.st->sum_k2, (unsigned long long)k2);
        atomicAdd(&A.st->sum_kcrop, (unsigned long long)K);
        S.dead = 0u;  // read in pass A, two barriers further down
      }
      if (K == 0) continue;
      L = A.lists + S.arena_off;
    }
    for (int e = tid; e < NW * kMaxOrient * 2; e += NT) (&S.exact_a[0][0][0])[e] = 0u;  // (read behind the barrier below)
    __syncthreads();
    AG2_PROF(2);

    // ---- pass A for ALL orientations in one sweep over the LDS list -------------------------
    // evaluateFingers(points_rot, init_bite) (finger_hand.cpp:17-72) needs, per orientation, only
    // three facts about the cropped points: is any below the fingertips (y < top), is any behind
    // the hand (y < bottom), and which of the 20 finger slots hold a point.  All are threshold
    // tests, so every point is CLASSIFIED from a float32 estimate of its rotated coordinates
    //   x ~ c (n.p) + s (b.p),  y ~ c (b.p) - s (n.p)      (good to ~3e-7 m)
    // whenever that estimate is more than 2e-6 m away from every threshold; a point nearer than
    // that to any threshold (a few per thousand) is redone with the reference's exact f64
    // expressions.  The decisions are therefore exactly those of the f64 arithmetic, at a fraction
    // of the f64 work, and the points are read from LDS once instead of once per orientation.
    const double top0 = hc.init_bite, bottom0 = hc.init_bite - hc.hand_depth;
    auto exact_A = [&](int i, float fx, float fy, float fz, unsigned& flg, unsigned& blk_bits) {
      const double cs = S.cosd[i], sn = S.sind[i];
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
        // strict f64 compares against the slot table (finger_hand.cpp:63).  Usual geometry: only
        // slots kf-1, kf, kf+1 of each half can hold x; their bounds are re-derived in registers
        // with the host's own expressions (fs[10+k] = k*step, fs[k] = (k*step - od) + fw,
        // fsr = fs + fw; ag2_context.hip), bit-identical to the table, branch-free.
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
            if (x > S.fs[kk] && x < S.fsr[kk]) blk_bits |= (1u << kk);
        }
      }
    };
    // Exact path: blk_acc (final slot bits), flg_acc (bit0: some point has y < top, bit1: y < bottom).
    // Fast path: raw_acc holds bit (position + 2) of the clamped slot lattice, converted once at the
    // end; its two flags are wave-uniform (ballots), kept per orientation in the scalars s_below /
    // s_behind -- the loop body below is what the kernel's VALU time goes to, so it is kept short.
    unsigned raw_acc[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; i++) raw_acc[i] = 0;
    unsigned s_below = 0, s_behind = 0;
    {
      const float n0 = (float)F[0][0], n1 = (float)F[1][0], n2 = (float)F[2][0];
      const float b0 = (float)F[0][1], b1 = (float)F[1][1], b2 = (float)F[2][1];
      const float top0f = (float)top0, bot0f = (float)bottom0;
      // Error budget of the estimates: |x|,|y| <= 0.1 m, a dozen f32 roundings => < 1.5e-7 m, i.e.
      // < 2e-5 slot spacings.  Margins: 1e-6 m on y, 1e-4 spacings (~9e-7 m) on x.
      const float mY = 1.0e-6f, eX = 1.0e-4f;
      const float invs = (float)slot_inv_step, rm1 = (float)(slot_ratio - 1.0);
      // The two finger blocks share one lattice: left slot k starts at (k - 9) spacings, right slot k
      // at k spacings (finger_hand.cpp:9-12: fs_half = LinSpaced(10, 0, od - fw), left = fs_half -
      // (od - fw)), so ONE position p = floor(x / spacing) serves both: left slot p + 9 for p in
      // [-9, 0], right slot p for p in [0, 9].  (The f64 thresholds of the two blocks differ in the
      // last ulp; a point that close to an edge takes the exact path anyway.)
      // Only set up for the usual geometry: 1 < finger width / spacing < 2.
      const bool fast_ok = (slot_span <= 2) && slot_ratio > 1.001 && slot_ratio < 1.999 &&
                           __builtin_fabs((slot_base1 - slot_base0) * slot_inv_step - 9.0) < 1.0e-9;
      // Orientations for which some point is already known to lie behind the hand (y < bottom:
      // evaluateFingers returns with every finger blocked, finger_hand.cpp:27-38) need no further
      // work; the set is shared across the workgroup through S.dead (an optimisation only: a late
      // reader just does redundant work).
      unsigned alive = (R >= 32) ? 0xFFFFFFFFu : ((1u << R) - 1u);
      // The point of the NEXT iteration is requested before this iteration's arithmetic (the list
      // position comes from LDS, the point from L2: two dependent latencies that would otherwise be
      // exposed once per iteration); lanes beyond the list re-read its last point and are masked.
      float qx, qy, qz;
      const int n_it = (K + NT - 1) / NT;
      auto jof = [&](int sidx) { return sidx * NT + tid; };
      ldp(min(jof(0), K - 1), qx, qy, qz);
#if AG2_SWEEP_DIAG
      if (A.prof && lane == 0) atomicAdd(&A.prof[6], (unsigned long long)(n_it * R));
#endif
      for (int sidx = 0; sidx < n_it; sidx++) {
        alive = (unsigned)__builtin_amdgcn_readfirstlane((int)(alive & ~S.dead));
#if AG2_SWEEP_DIAG  // (orientation-steps done / possible, summed over waves: how early orientations retire)
        if (A.prof && lane == 0) atomicAdd(&A.prof[4], (unsigned long long)__popc(alive));
#endif
        if (alive == 0u) break;
        const int j = jof(sidx);
        const bool valid = j < K;
        const float px = qx, py = qy, pz = qz;
        ldp(min(jof(sidx + 1), K - 1), qx, qy, qz);
        const float u = (n0 * px + n1 * py) + n2 * pz;
        const float v = (b0 * px + b1 * py) + b2 * pz;
        unsigned need_exact = 0;  // orientations whose estimate is too close to a threshold
        unsigned newly = 0;       // (uniform) orientations found blocked from behind in this step
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
          if (i < R && ((alive >> i) & 1u)) {  // wave-uniform
            const float cf = S.cosf_t[i], sf = S.sinf_t[i];
            const float ya = __builtin_fmaf(cf, v, -(sf * u));
            // distance of the estimates to the nearest threshold, in y and in slot spacings
            const float dy = __builtin_fminf(__builtin_fabsf(ya - top0f), __builtin_fabsf(ya - bot0f));
            // Points in front of the fingertips (y >= top) take part in nothing (finger_hand.cpp:27:
            // only points with y < top are looked at).  The list is in cell order, so the 64 points of
            // a wave are neighbours in space and often ALL lie in front: the slot arithmetic of this
            // orientation is then skipped for the whole wave.
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
              // position p = floor(x / spacing) holds x (0 < frac < 1 < width); position p - 1 holds
              // it iff frac + 1 < width.  Bit (p + 11), clamped: positions outside -9 .. 9 land on
              // bits the conversion below drops.
              int pb = (int)kff + 11;
              pb = min(max(pb, 0), 22);
              const unsigned bit = 1u << pb;
              raw_acc[i] |= bit | ((frac < rm1) ? (bit >> 1) : 0u);
            }
            if (!fast && valid) need_exact |= 1u << i;
            if (__ballot(hit)) {
              s_below |= 1u << i;
              if (__ballot(hit && ya < bot0f)) {
                s_behind |= 1u << i;
                newly |= 1u << i;
              }
            }
          }
        }
        // the rare exact evaluations sit outside the unrolled loop: one copy of the f64 code
        while (need_exact) {
          const int ie = __ffs((int)need_exact) - 1;
          need_exact &= need_exact - 1u;
          unsigned flg = 0, bits = 0;
          exact_A(ie, px, py, pz, flg, bits);
          if (bits) atomicOr(&S.exact_a[wid][ie][0], bits);
          if (flg) atomicOr(&S.exact_a[wid][ie][1], flg);
        }
        // Long lists: every eighth step, an orientation whose finger slots are already so occupied --
        // by the points THIS wave has seen -- that no hand placement is left (the gates below:
        // finger_hand.cpp:313-325, hand_search.cpp:366) is retired as well.  Occupancy only grows with
        // more points, so it cannot come back; it is marked like one with a point behind the hand
        // (its slot mask stays incomplete, and the gates must not read it as open).
        if (STAGE == 1 && (sidx & 7) == 7) {  // (compiled into the long-list stage only: in the other it
                                               //  cost more -- 3 % of the kernel at cfg2 -- than the few lists
                                               //  long enough to profit gave back)
          unsigned full = 0;
#pragma unroll
          for (int i = 0; i < RMAX; i++) {
            if (i < R && ((alive >> i) & 1u)) {  // wave-uniform
              const unsigned pm = (raw_acc[i] >> 2) & 0x7FFFFu;
              const unsigned cbw = wave_or_u((pm & 0x3FFu) | ((pm >> 9) << 10)) | S.exact_a[wid][i][0];
              const unsigned fr = (~cbw) & 0xFFFFFu;
              if ((fr & (fr >> 10) & 0x3FFu) == 0u || __popc(fr) <= 2) full |= 1u << i;
            }
          }
          full = (unsigned)__builtin_amdgcn_readfirstlane((int)full);
          s_behind |= full;
          newly |= full;
        }
        if (newly) {
          alive &= ~newly;
          if (lane == 0) atomicOr(&S.dead, newly);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RMAX; i++) {
      if (i < R) {
        const unsigned pm = (raw_acc[i] >> 2) & 0x7FFFFu;  // the 19 positions -9 .. 9
        // (the exact path's bits of this wave: its lanes' LDS atomics are complete -- same wave, program order)
        const unsigned bsum = wave_or_u((pm & 0x3FFu) | ((pm >> 9) << 10)) | S.exact_a[wid][i][0];
        const unsigned fsum = S.exact_a[wid][i][1] | ((s_below >> i) & 1u) | (((s_behind >> i) & 1u) ? 3u : 0u);
        if (lane == 0) {
          S.res_a[wid][i][0] = bsum;
          S.res_a[wid][i][1] = fsum;
        }
      }
    }
    __syncthreads();
    AG2_PROF(3);

    // ---- orientations -----------------------------------------------------------------------
    // Lane i of every wave combines the waves' pass-A results for orientation i and applies the
    // gates; the loop below then visits only the orientations that pass (usually none or one).
    unsigned hand_l = 0;
    {
      unsigned cb = 0, cf = 0;
      if (lane < R) {
#pragma unroll
        for (int k = 0; k < NW; k++) {
          cb |= S.res_a[k][lane][0];
          cf |= S.res_a[k][lane][1];
        }
      }
      const unsigned free_ = (~cb) & 0xFFFFFu;
      const bool open = (lane < R) && !(cf & 2u) && (cf & 1u)       // finger_hand.cpp:35-36, :41-42
                        && (__popc(free_) > 2);                      // hand_search.cpp:366
      hand_l = open ? (free_ & (free_ >> 10) & 0x3FFu) : 0u;        // finger_hand.cpp:313-325
    }
    unsigned long long todo = __ballot(hand_l != 0u);               // hand_search.cpp:370
    {
      if (todo) {  // (uniform: every wave holds the same gates)
        // the two reservations -- room for the list, places in the pair queue -- go out together: one round
        // trip instead of two on the critical path of a sample that passes
        if (tid == 0) {
          const unsigned pbase = atomicAdd(&A.st->n_pairs, (unsigned)__popcll(todo));
          if (!ARENA) {
            long long loff = (long long)atomicAdd(&A.st->list_top, (unsigned long long)K);
            if (loff + K > A.list_cap) {
              atomicOr(&A.st->err_flags, 2u);  // the host grows the list arena and repeats the run
              loff = -1;
            }
            S.arena_off = loff;
          }
          S.flag = (int)pbase;
        }
        __syncthreads();
        const long long loff = S.arena_off;  // (ARENA: where the crop put the list)
        const unsigned base = (unsigned)S.flag;
        if (loff >= 0) {
          if constexpr (!ARENA) {
            // four points requested before the first is stored (the position comes from LDS, the point from
            // L2: one point per iteration was a chain of seven exposed round trips per thread)
            for (int j0 = tid; j0 < K; j0 += 4 * NT) {
              float4 v4[4];
              int ps[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                ps[u] = pos_at(min(j0 + u * NT, K - 1));
                v4[u] = A.pts[ps[u]];
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int j = j0 + u * NT;
                if (j < K)  // (p - q in float: the value ldp() hands to the passes)
                  A.lists[loff + j] = make_float4(v4[u].x - q.x, v4[u].y - q.y, v4[u].z - q.z, __int_as_float(ps[u]));
              }
            }
          }
        }
        // (the queue places were taken together with the list's room: when that room was refused -- the run is
        // then repeated with a larger arena -- the places are filled with pairs k_sweep_orient skips: K = 0)
        if (wid == 0 && hand_l != 0u) {
          SweepPair pr;
          pr.t = t;
          pr.oi = lane;
          pr.hand = hand_l;
          pr.K = (loff >= 0) ? K : 0;
          pr.list_off = (loff >= 0) ? loff : 0;
          pr.segs.end[0] = ge0; pr.segs.end[1] = ge1; pr.segs.end[2] = ge2;
          pr.segs.shift[0] = gs0; pr.segs.shift[1] = gs1; pr.segs.shift[2] = gs2;
          pr.pad[0] = pr.pad[1] = 0;
          A.pairs[base + (unsigned)__popcll(todo & lt_mask)] = pr;
        }
      }
    }
    AG2_PROF(7);
  }
}

// Which capacity the next run's stage 0 gets (see kGposCap): the big slice while the long-list stage sees only
// a few samples per run, the small one when a sizeable part of them overflows anyway.
void sweep_adapt_gpos(ag2_ctx* c, size_t n_samples, size_t n_overflow) {
  if (c->sweep_gpos_cap <= 0) c->sweep_gpos_cap = kGposCap;
  if (n_overflow * 4 > n_samples) c->sweep_gpos_cap = kGposCap;
  else if (n_overflow > 0 && n_overflow * 16 <= n_samples) c->sweep_gpos_cap = kGposCapBig;
  c->sweep_no_overflow_runs = n_overflow ? 0 : std::min(c->sweep_no_overflow_runs + 1, 1000);
}

static size_t sweep_lds_bytes(int stage) {
  size_t b = sweep_ctl_bytes(stage);
  if (stage == 0) b += (size_t)kLdsCap * (size_t)kStage0PointBytes;
  return b;
}

int launch_frames(ag2_ctx* c, size_t s, uint64_t slot_base, uint64_t seed) {
  if (s == 0) return 0;
  hipLaunchKernelGGL(k_frames, dim3((unsigned)s), dim3(64), 0, c->stream, c->d_sorted.as<float4>(),
                     c->d_nrm.as<float4>(), c->d_cell.as<unsigned>(), c->grid,
                     c->fm_on ? c->d_griddesc.as<GridDesc>() : (const GridDesc*)nullptr,
                     c->d_hc.as<HandConst>(), c->d_sample_q.as<float4>(), (int)s,
                     (unsigned long long)slot_base, (unsigned long long)seed,
                     c->fm_on ? c->fm_args_dev : (const FrameArgs*)nullptr,
                     c->d_frames.as<double>(), c->d_frame_ok.as<int>(), c->d_stats.as<DevStats>());
  hipLaunchKernelGGL(k_frames_finish, dim3(((unsigned)s + 63) / 64), dim3(64), 0, c->stream,
                     c->d_hc.as<HandConst>(), c->d_sample_q.as<float4>(), (int)s, c->d_frames.as<double>(),
                     c->d_frame_ok.as<int>(), c->d_stats.as<DevStats>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// d_idx: device-visible sample indices; clear_run as in upload_samples
int launch_sample_queries(ag2_ctx* c, const int* d_idx, size_t s, bool clear_run) {
  hipLaunchKernelGGL(k_sample_queries_idx, dim3(((unsigned)s + 255) / 256), dim3(256), 0, c->stream,
                     d_idx, (int)s, c->d_xyz_in.as<float4>(), (int)(c->fm_on ? c->fm_n_max : c->n),
                     c->d_sample_q.as<float4>(),
                     clear_run ? c->d_tab_keep.as<unsigned char>() : (unsigned char*)nullptr,
                     c->p.num_orientations, c->d_stats.as<DevStats>());
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// clear_run: the index kernel also clears tab_keep (reserved by the caller) and the per-run statistics
int upload_samples(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                   bool clear_run) {
  AG2_HIP(c, c->d_sample_q.reserve(std::max<size_t>(s, 1) * 16));
  AG2_HIP(c, c->d_frames.reserve(std::max<size_t>(s, 1) * 12 * 8));
  AG2_HIP(c, c->d_frame_ok.reserve(std::max<size_t>(s, 1) * 4));
  if (s == 0) return 0;
  // (frame_ok needs no clearing: k_frames writes it for every sample)
  if (sample_idx || !sample_xyz) {
    const int* d_idx = c->d_samples.as<int>();  // left by ag2_subsample_uniformly
    if (sample_idx) {
      // The kernel reads the indices straight from page-locked host memory (20 KB over PCIe:
      // cheaper than a DMA operation of its own in front of the kernel); the staging area is not
      // written again before this call has synchronised.
      const int rc = pin_reserve(c, s * 4);
      if (rc) return rc;
      __builtin_memcpy(pin_bulk(c), sample_idx, s * 4);
      void* dev_view = nullptr;
      AG2_HIP(c, hipHostGetDevicePointer(&dev_view, pin_bulk(c), 0));
      d_idx = (const int*)dev_view;
    }
    const int rc = launch_sample_queries(c, d_idx, s, clear_run);
    if (rc) return rc;
  } else {
    std::vector<float> q(s * 4);
    for (size_t i = 0; i < s; i++) {
      // hand_search.cpp:261-263: sample.x = samples(0,i)  (double -> float)
      const float x = (float)sample_xyz[3 * i], y = (float)sample_xyz[3 * i + 1],
                  z = (float)sample_xyz[3 * i + 2];
      const bool ok = std::isfinite(x) && std::isfinite(y) && std::isfinite(z);
      q[4 * i] = x; q[4 * i + 1] = y; q[4 * i + 2] = z; q[4 * i + 3] = ok ? 1.f : 0.f;
    }
    AG2_HIP(c, hipMemcpyAsync(c->d_sample_q.p, q.data(), s * 16, hipMemcpyHostToDevice, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));  // q goes out of scope
  }
  return 0;
}

// run_cleared: tab_keep was cleared by k_sample_queries_idx already
int launch_sweep(ag2_ctx* c, size_t s, uint64_t slot_base, bool emit_lists, bool run_cleared) {
  const int R = c->p.num_orientations;
  const size_t n_slots = s * (size_t)R;
  AG2_HIP(c, c->d_table.reserve(std::max<size_t>(n_slots, 1) * sizeof(ag2_hypothesis)));
  AG2_HIP(c, c->d_tab_off.reserve(std::max<size_t>(n_slots, 1) * 8));
  AG2_HIP(c, c->d_tab_keep.reserve(std::max<size_t>(n_slots, 1)));
  AG2_HIP(c, c->d_overflow.reserve(std::max<size_t>(s, 1) * 4));  // queue of sample ids for stage 1
  if (s == 0) return 0;
  if (emit_lists && c->arena_points == 0) {
    c->arena_points = (size_t)16 << 20;  // 16 Mi points = 768 MiB; grown on AG2_ERR_CAPACITY
  }
  if (emit_lists) AG2_HIP(c, c->d_arena.reserve(c->arena_points * 48));
  // (the records of empty slots are never read: every consumer goes through the slot states, and
  // ag2_export_candidates_device writes zeros for them)
  if (!run_cleared) AG2_HIP(c, hipMemsetAsync(c->d_tab_keep.p, 0, n_slots, c->stream));
  SweepArgs A{};
  A.pts = c->d_sorted.as<float4>();
  A.nrm = c->d_nrm.as<float4>();
  A.cell = c->d_cell.as<unsigned>();
  A.g = c->grid;
  A.gp = c->fm_on ? c->d_griddesc.as<GridDesc>() : nullptr;
  A.fa = c->fm_on ? c->fm_args_dev : nullptr;
  A.hc = c->d_hc.as<HandConst>();
  A.sample_q = c->d_sample_q.as<float4>();
  A.frames = c->d_frames.as<double>();
  A.frame_ok = c->d_frame_ok.as<int>();
  A.n_samples = (int)s;
  A.slot_base = (int)slot_base;
  A.table = c->d_table.as<ag2_hypothesis>();
  A.tab_off = c->d_tab_off.as<long long>();
  A.tab_keep = c->d_tab_keep.as<unsigned char>();
  A.arena = c->d_arena.as<double>();
  A.arena_cap = emit_lists ? (long long)c->arena_points : 0;
  A.emit_lists = emit_lists ? 1 : 0;
  A.st = c->d_stats.as<DevStats>();
  A.overflow = c->d_overflow.as<int>();
  A.min_z = c->min_z;
  A.flags = c->p.debug_flags & 1;
  DevBuf& prof_buf = c->d_sweep_prof;  // diagnostic only: per-phase cycle sums when AG2_SWEEP_PROF is set
  const bool want_prof = !c->fm_on && getenv("AG2_SWEEP_PROF") != nullptr;
  if (want_prof) {
    AG2_HIP(c, prof_buf.reserve(16 * 8));
    AG2_HIP(c, hipMemsetAsync(prof_buf.p, 0, 16 * 8, c->stream));
    A.prof = prof_buf.as<unsigned long long>();
  }
  const size_t lds = sweep_lds_bytes(0);
  typedef void (*SweepFn)(SweepArgs);
  const SweepFn fn_lds = (R <= 8) ? k_sweep<0, 8> : (R <= 16 ? k_sweep<0, 16> : k_sweep<0, 32>);
  const SweepFn fn_glb = (R <= 8) ? k_sweep<1, 8> : (R <= 16 ? k_sweep<1, 16> : k_sweep<1, 32>);
  if (c->list_ints == 0)  // 16 Mi points = 256 MiB, grown on demand (debug_flags bit 1: start tiny, to exercise that)
    c->list_ints = (c->p.debug_flags & 2) ? (size_t)4096 : (size_t)16 << 20;
  AG2_HIP(c, c->d_lists.reserve(c->list_ints * 16));
  AG2_HIP(c, c->d_pairs.reserve(std::max<size_t>(n_slots, 1) * sizeof(SweepPair)));
  A.lists = c->d_lists.as<float4>();
  A.list_cap = (long long)c->list_ints;
  A.pairs = c->d_pairs.as<SweepPair>();
  const int grid = (int)std::min<size_t>(s, 256 * kStage0WgPerCu);
  // first stage: one workgroup per sample
  if (c->sweep_gpos_cap <= 0) c->sweep_gpos_cap = kGposCap;
  A.gpos_cap = c->sweep_gpos_cap;
  AG2_HIP(c, c->d_gpos.reserve((size_t)256 * kStage0WgPerCu * (size_t)A.gpos_cap * 4));
  A.gpos = c->d_gpos.as<int>();
  hipLaunchKernelGGL(fn_lds, dim3(grid), dim3(kSweepThreads0), lds, c->stream, A);
  AG2_HIP(c, hipGetLastError());
  AG2_HIP(c, stage_event(c, 2));
  if (want_prof) {
    unsigned long long h[8];
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    AG2_HIP(c, hipMemcpy(h, prof_buf.p, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[ag2 sweep prof, stage 0, cycles summed over workgroups] rows %llu crop1 %llu "
            "crop2 %llu passA %llu other %llu | pass A orientation-steps done %llu of %llu\n",
            h[0], h[1], h[2], h[3], h[7], h[4], h[6]);
    AG2_HIP(c, hipMemsetAsync(prof_buf.p, 0, 16 * 8, c->stream));
  }
  // Samples whose cropped list exceeds stage 0 were queued on the device; the second stage is always
  // launched and reads the queue length itself (st->n_overflow), so no host round trip sits between
  // the launches.
  // (default: four workgroups per CU x 5 x 64 Ki words = 1.3 GB of scratch; run_hypotheses resizes
  // it when a list of the run is longer)
  const int gcap = c->sweep_gcap, g2 = c->sweep_g2;
  A.gscratch = c->d_gscratch.as<float>();
  A.gcap = gcap;
  // (The one-round-trip detect leaves the launch out when the context's last runs queued nothing for this stage --
  // an empty launch is 5 us of a 0.68 ms step -- and repeats the call step by step if this run did.)
  c->sweep_stage1_skipped = c->fm_on ? c->fm_skip_stage1
                                     : (c->sweep_may_skip_stage1 && c->sweep_no_overflow_runs >= 2);
  if (!c->sweep_stage1_skipped)
    hipLaunchKernelGGL(fn_glb, dim3(g2), dim3(kSweepThreads1), sweep_lds_bytes(1), c->stream, A);
  AG2_HIP(c, hipGetLastError());
  {
    const int rc = launch_sweep_orient(c, A, n_slots);
    if (rc) return rc;
  }
  AG2_HIP(c, stage_event(c, 11));
  if (want_prof) {
    unsigned long long h[8];
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    AG2_HIP(c, hipMemcpy(h, prof_buf.p, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[ag2 sweep prof, stage 1 (global scratch)] rows %llu crop1 %llu crop2 %llu passA %llu "
            "deepen %llu passC %llu passD %llu other %llu\n",
            h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  return 0;
}

}  // namespace ag2
