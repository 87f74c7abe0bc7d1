// k_image.hip -- K4: grasp image renderer.
//
// Replaces Learning::createGraspImages / convertToImageRGB (src/agile_grasp2/learning.cpp:4-33,
// :143-209) and the convertTo(CV_8UC3, 255.0) at :16.  The reference scans all P points for each of
// the 3600 cells; here the per-cell f64 sums are formed in the SAME order (list order within a cell)
// without atomics, so the bytes are deterministic and equal to the oracle's.
//
// k_render_sparse (P <= 1024, the usual case: P ~ 250 << 3600 cells): every point finds, among the
// earlier points, how many share its cell (its rank) and the first that does (the cell's leader).
// Round r then adds the normals of all rank-r points to their leader's accumulator -- distinct
// cells within a round, list order across rounds -- and only the leaders quantise a pixel.  34 KB of
// LDS: four workgroups per CU.
// k_render_counted (1024 < P <= 4096): cell-major order by a stable counting sort in LDS (ranks from
// wave ballots), then one run per cell, added in a counted loop.
// k_render_sorted (4096 < P <= 16384): the same order from a bitonic network over (cell, position) keys.
// The images of these two are listed per renderer first (k_render_classify) and handed out one at a time.
// k_render (any P): every cell is owned by one thread (cell % 256) which adds the normals of its
// points in list order; two passes of 30 image rows, 76 KB of LDS.
// Quantisation to u8 happens BEFORE the 3x3 dilate: v -> sat(rint(255 v)) is monotone, so
// max-then-quantise == quantise-then-max, bit for bit.
#include "ag2_internal.h"

namespace ag2 {

constexpr int kImgThreads = 256;
constexpr int kCidChunk = 4096;
constexpr int kCells = kImg * kImg;
constexpr int kHalfCells = kCells / 2;
constexpr int kSparseMax = 1024;  // points per image the sparse kernel takes

struct ImgShared {
  double acc[kHalfCells * 3];
  unsigned pix[kCells];
  short cid[kCidChunk];
  double red[kImgThreads / kWave];
  unsigned char obuf[kCells * 3];  // output image staged for coalesced dword stores
};
static_assert(sizeof(ImgShared) * 2 <= 160 * 1024, "k_render: two workgroups per CU");

// learning.cpp:152-156: cell = floor(x / cellsize) + floor((y - min y) / cellsize) * 60; ids outside
// 0..3599 are dropped, x-cells >= 60 alias into the next row (replicated)
__device__ __forceinline__ short cell_id(double ux, double uy, double miny) {
  const double cellsize = 1.0 / (double)kImg;
  const double fx = __builtin_floor(ux / cellsize);
  const double fy = __builtin_floor((uy - miny) / cellsize);
  if (__builtin_fabs(fx) < 1.0e9 && __builtin_fabs(fy) < 1.0e9) {
    const long long cl = (long long)fx + (long long)fy * kImg;
    if (cl >= 0 && cl < kCells) return (short)cl;
  }
  return (short)-1;
}

// :181-190 avg <- |avg / ||avg|||, then convertTo(.., 255.0) (:16) with cvRound (half to even),
// channels packed b0 | b1 << 8 | b2 << 16.  An all-zero sum gives the 0/0 -> NaN -> 0 pixel.
__device__ __forceinline__ unsigned quantise(double ax, double ay, double az) {
  unsigned packed = 0;
  if (ax != 0.0 || ay != 0.0 || az != 0.0) {
    const double s = 1.0 / __builtin_sqrt((ax * ax + ay * ay) + az * az);
    const double v[3] = {__builtin_fabs(s * ax), __builtin_fabs(s * ay), __builtin_fabs(s * az)};
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float f = (float)v[ch];
      const float tq = f * 255.0f;
      unsigned u = 0;
      if (tq == tq) {
        const float r = __builtin_rintf(tq);
        u = (unsigned)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
      }
      packed |= u << (8 * ch);
    }
  }
  return packed;
}

// two u16 lanes of a dword at once (v_pk_max_u16 / v_pk_sub_u16 / v_pk_min_u16 / v_pk_add_u16)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(unsigned v) {
  us2 r;
  __builtin_memcpy(&r, &v, 4);
  return r;
}
__device__ __forceinline__ unsigned as_u32(us2 v) {
  unsigned r;
  __builtin_memcpy(&r, &v, 4);
  return r;
}

// :202-203 3x3 rect dilate (border taps ignored), :206 BGR2RGB swap; staged so that the global
// store is coalesced dwords.  A pixel is b0 | b1 << 8 | b2 << 16; split once into (b0, b2) as two u16
// lanes and b1, a per-channel max of two pixels is one packed and one plain max.  A thread owns a
// column and 15 consecutive rows of it: the horizontal maxima of its 17 rows (three reads each) stay
// in registers for the vertical pass -- 51 LDS reads and ~130 instructions for 15 pixels, where nine
// taps with three masked maxima each cost ~100 instructions per pixel (this pass was the largest part
// of the renderer for the typical image).
template <int NT = kImgThreads>
__device__ __forceinline__ void dilate_store(const unsigned* __restrict__ pix,
                                             unsigned char* __restrict__ obuf,
                                             unsigned char* __restrict__ out_img, int tid) {
  constexpr int kStrip = 15, kStrips = kImg / kStrip;
  static_assert(kStrips * kStrip == kImg, "rows per strip");
  for (int s = tid; s < kImg * kStrips; s += NT) {
    const int cc = s % kImg, r0 = (s / kImg) * kStrip;
    us2 he[kStrip + 2];    // horizontal max of rows r0 - 1 .. r0 + kStrip: channels 0 and 2
    unsigned ho[kStrip + 2];  // ... channel 1
#pragma unroll
    for (int k = 0; k < kStrip + 2; k++) {
      const int rr = r0 - 1 + k;
      us2 e = as_us2(0u);
      unsigned o = 0u;
      if (rr >= 0 && rr < kImg) {
        const unsigned* row = pix + rr * kImg;
        const unsigned v1 = row[cc];
        const unsigned v0 = (cc > 0) ? row[cc - 1] : 0u;       // (0 never wins a max: border taps ignored)
        const unsigned v2 = (cc < kImg - 1) ? row[cc + 1] : 0u;
        e = __builtin_elementwise_max(__builtin_elementwise_max(as_us2(v0 & 0x00FF00FFu), as_us2(v1 & 0x00FF00FFu)),
                                      as_us2(v2 & 0x00FF00FFu));
        o = max(max(v0 & 0xFF00u, v1 & 0xFF00u), v2 & 0xFF00u);
      }
      he[k] = e;
      ho[k] = o;
    }
#pragma unroll
    for (int k = 0; k < kStrip; k++) {
      const unsigned e = as_u32(__builtin_elementwise_max(__builtin_elementwise_max(he[k], he[k + 1]), he[k + 2]));
      const unsigned o = max(max(ho[k], ho[k + 1]), ho[k + 2]);
      const int p = (r0 + k) * kImg + cc;
      obuf[p * 3 + 2] = (unsigned char)(e & 255u);
      obuf[p * 3 + 1] = (unsigned char)(o >> 8);
      obuf[p * 3 + 0] = (unsigned char)(e >> 16);
    }
  }
  __syncthreads();
  unsigned* dst = reinterpret_cast<unsigned*>(out_img);
  const unsigned* src = reinterpret_cast<const unsigned*>(obuf);
  for (int i = tid; i < kCells * 3 / 4; i += NT) dst[i] = src[i];
}

template <int NT = kImgThreads>
__device__ __forceinline__ double block_min_y(const double* __restrict__ pts, int P, double* red, int tid) {
  double miny = __builtin_inf();  // learning.cpp:148-149  y <- y - min y
  for (int b = tid; b < P; b += NT) {
    const double y = pts[(size_t)b * 6 + 1];
    miny = (y < miny) ? y : miny;
  }
  miny = wave_min_d(miny);
  if (lane_id() == 0) red[wave_id()] = miny;
  __syncthreads();
  miny = red[0];
#pragma unroll
  for (int k = 1; k < NT / kWave; k++) miny = (red[k] < miny) ? red[k] : miny;
  return miny;
}

// The images of one renderer as a dense list, handed out through a counter: a renderer's images differ by a
// factor of four in their point counts, and a static deal of a few of them to every workgroup ends with most
// of the GPU waiting for the unluckiest one.  list == nullptr: all images in turn, each kernel skipping the
// others' by the point count (when only one renderer runs there is nothing to balance).
struct ImgQueue {
  const int* list;        // image indices of this renderer
  const unsigned* count;  // how many
  unsigned* next;         // zero at launch; a workgroup takes gridDim.x + (this++) when it is done with an image
};
__device__ __forceinline__ int img_queue_next(const ImgQueue& q, int w, int* s_next, int& it) {
  if (!q.list) return w + (int)gridDim.x;
  if (threadIdx.x == 0) s_next[it & 1] = (int)gridDim.x + (int)atomicAdd(q.next, 1u);
  __syncthreads();
  const int v = __builtin_amdgcn_readfirstlane(s_next[it & 1]);
  it++;
  return v;
}

// kinds: 0 = (kSparseMax, kSortedMax], 1 = (kSortedMax, kSortedMaxBig], 2 = beyond; ctr[kind] = list length
__global__ void __launch_bounds__(256) k_render_classify(const long long* __restrict__ desc_off,
                                                         const int* __restrict__ desc_cnt, int n_img,
                                                         const unsigned* __restrict__ d_n, int cap,
                                                         unsigned* __restrict__ ctr, int* __restrict__ lists) {
  if (d_n) n_img = min(n_img, (int)*d_n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  int kind = -1;
  if (i < n_img) {
    const int P = (desc_off[i] >= 0) ? desc_cnt[i] : 0;
    kind = (P <= kSparseMax) ? -1 : ((P <= 4096) ? 0 : ((P <= 16384) ? 1 : 2));
  }
  const unsigned long long lt = (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const unsigned long long m = __ballot(kind == k);
    if (m == 0ull) continue;
    unsigned base = 0;
    if (lane_id() == 0) base = atomicAdd(&ctr[k], (unsigned)__popcll(m));
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (kind == k) lists[(size_t)k * cap + base + (unsigned)__popcll(m & lt)] = i;
  }
}

__global__ void __launch_bounds__(kImgThreads) k_render(const double* __restrict__ arena,
                                                        const long long* __restrict__ desc_off,
                                                        const int* __restrict__ desc_cnt, int n_img,
                                                        const unsigned* __restrict__ d_n,
                                                        int p_min, unsigned char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ImgShared& S = *reinterpret_cast<ImgShared*>(smem_raw);
  if (d_n) n_img = min(n_img, (int)*d_n);  // frame mode: the list length is read on the device
  const int tid = threadIdx.x;
  {  // nothing to do for this workgroup (the usual case: the sparse kernel took every image)?
    bool any = false;
    for (int im = blockIdx.x; im < n_img; im += gridDim.x)
      any = any || (desc_off[im] >= 0 && desc_cnt[im] >= p_min);
    if (!any) return;  // uniform
  }
  // accumulators are zeroed once; every quantisation pass re-zeroes the cells it consumed
  for (int i = tid; i < kHalfCells * 3; i += kImgThreads) S.acc[i] = 0.0;
  for (int im = blockIdx.x; im < n_img; im += gridDim.x) {
    const long long off = desc_off[im];
    const int P = (off >= 0) ? desc_cnt[im] : 0;
    if (P < p_min) continue;  // taken by k_render_sparse (uniform)
    const double* pts = arena + (size_t)(off >= 0 ? off : 0) * 6;
    __syncthreads();
    const double miny = block_min_y(pts, P, S.red, tid);
    const bool one_chunk = P <= kCidChunk;       // usual case: cell ids computed once for both halves
    for (int half = 0; half < 2; half++) {
      const int cell_lo = half * kHalfCells;
      for (int c0 = 0; c0 < P; c0 += kCidChunk) {
        const int cn = min(kCidChunk, P - c0);
        if (!(one_chunk && half == 1)) {
          __syncthreads();
          for (int b = tid; b < ((cn + 7) & ~7); b += kImgThreads)  // pads the chunk to 8 ids
            S.cid[b] = (b < cn) ? cell_id(pts[(size_t)(c0 + b) * 6], pts[(size_t)(c0 + b) * 6 + 1], miny)
                                : (short)-1;
          __syncthreads();
        }
        // every thread scans the chunk, 8 ids per LDS read, and adds only to cells it owns
        for (int b0 = 0; b0 < cn; b0 += 8) {
          const uint4 w = *reinterpret_cast<const uint4*>(&S.cid[b0]);
          const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int id = (int)(short)((ww[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
            const int cell = id - cell_lo;
            if ((unsigned)cell < (unsigned)kHalfCells && (cell & (kImgThreads - 1)) == tid) {
              const double* y = pts + (size_t)(c0 + b0 + k) * 6 + 3;     // :166-179
              S.acc[3 * cell + 0] = S.acc[3 * cell + 0] + y[0];
              S.acc[3 * cell + 1] = S.acc[3 * cell + 1] + y[1];
              S.acc[3 * cell + 2] = S.acc[3 * cell + 2] + y[2];
            }
          }
        }
      }
      // written at (59 - row, col).  Only the owner of a cell ever touches its accumulator, so no
      // barrier is needed between accumulation and this pass.
      for (int hc = tid; hc < kHalfCells; hc += kImgThreads) {
        const double ax = S.acc[3 * hc], ay = S.acc[3 * hc + 1], az = S.acc[3 * hc + 2];
        if (ax != 0.0 || ay != 0.0 || az != 0.0) {
          S.acc[3 * hc] = 0.0;
          S.acc[3 * hc + 1] = 0.0;
          S.acc[3 * hc + 2] = 0.0;
        }
        const int cell = hc + cell_lo;
        const int row = kImg - 1 - cell / kImg, col = cell % kImg;
        S.pix[row * kImg + col] = quantise(ax, ay, az);
      }
    }
    __syncthreads();
    dilate_store(S.pix, S.obuf, out + (size_t)im * (kCells * 3), tid);
  }
}

// The accumulators (one f64 triple per leader point) are dead once the leaders have quantised their
// cell, and the pixel map + the staged output image are needed only from then on: the two share one
// area.  With the cell -> first-point table a workgroup takes 34 KB: four per CU, 1 024 images in
// flight, so the ~900 images of a configuration-2 step run in one round (the kernel is bound by the
// latency of its largest image: the rank loop, ~15 barriers and a dependent read of its point list).
struct SparseShared {
  union {
    unsigned short hist[kImgThreads / kWave][kCells];  // rank phase: points of (wave, cell) so far
    double acc[kSparseMax * 3];
    struct {
      unsigned pix[kCells];
      unsigned char obuf[kCells * 3];
    } img;
  } u;
  unsigned short lead[kCells];  // list position of each occupied cell's first point
  double red[kImgThreads / kWave];
  int max_rank;
};
static_assert(sizeof(SparseShared) * 4 <= 160 * 1024, "k_render_sparse: four workgroups per CU");

__global__ void __launch_bounds__(kImgThreads, 4) k_render_sparse(const double* __restrict__ arena,
                                                               const long long* __restrict__ desc_off,
                                                               const int* __restrict__ desc_cnt,
                                                               int n_img, const unsigned* __restrict__ d_n,
                                                               unsigned char* __restrict__ out) {
  __shared__ SparseShared S;
  if (d_n) n_img = min(n_img, (int)*d_n);  // frame mode: the list length is read on the device
  const int tid = threadIdx.x;
  constexpr int kPer = kSparseMax / kImgThreads;  // points per thread
  for (int im = blockIdx.x; im < n_img; im += gridDim.x) {
    const long long off = desc_off[im];
    const int P = (off >= 0) ? desc_cnt[im] : 0;
    if (P > kSparseMax) continue;  // taken by k_render (uniform)
    const double* pts = arena + (size_t)(off >= 0 ? off : 0) * 6;
    __syncthreads();  // previous image's readers of S are done
    const double miny = block_min_y(pts, P, S.red, tid);
    // Every wave owns a contiguous quarter of the list (seg positions, a multiple of 64), in chunks of 64
    // consecutive positions: (wave, chunk, lane) order is list order.
    const int seg = ((P + kImgThreads - 1) / kImgThreads) * kWave, nc = seg / kWave;
    const int b_first = wave_id() * seg + lane_id();
    double yv[kPer][3];
    short myc[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int b = b_first + k * kWave;
      myc[k] = -1;
      yv[k][0] = yv[k][1] = yv[k][2] = 0.0;
      if (k < nc && b < P) {
        const double* p = pts + (size_t)b * 6;
        myc[k] = cell_id(p[0], p[1], miny);
        yv[k][0] = p[3]; yv[k][1] = p[4]; yv[k][2] = p[5];
      }
    }
    for (int i = tid; i < (kImgThreads / kWave) * kCells / 2; i += kImgThreads)
      reinterpret_cast<unsigned*>(&S.u.hist[0][0])[i] = 0u;
    if (tid == 0) S.max_rank = 0;
    __syncthreads();
    // Rank of every point among the earlier points of its cell, by COUNTING (round 3; before: every point
    // compared its cell id with those of all earlier points -- P^2 / 2 compares for the largest image, and the
    // kernel lasts as long as its largest image).  In a chunk the lanes that share a cell find each other with
    // twelve ballots (after bit k a lane's mask keeps the lanes whose bit k of the cell id equals its own): the
    // rank inside the chunk; + the wave's count of the cell so far (u16 per wave and cell) = the rank inside
    // the wave's quarter; + the totals of the waves in front = the rank.  Data-independent: ~100 instructions
    // per chunk whatever the cells hold.
    const unsigned long long lt_mask = (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
    int my_rank[kPer], my_lead[kPer], mx = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      my_rank[k] = -1;  // beyond the list, or a dropped point (:156)
      my_lead[k] = b_first + k * kWave;
      if (k < nc) {  // uniform
        const bool valid = myc[k] >= 0;
        const int cell = valid ? (int)myc[k] : 0;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 12; bit++) {
          const bool mine = (cell >> bit) & 1;
          const unsigned long long bb = __ballot(mine);
          m &= mine ? bb : ~bb;
        }
        if (valid) {
          const int before = (int)S.u.hist[wave_id()][cell];
          my_rank[k] = before + __popcll(m & lt_mask);
          if ((m >> lane_id()) == 1ull) S.u.hist[wave_id()][cell] = (unsigned short)(before + __popcll(m));
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++)
      if (my_rank[k] >= 0) {
        for (int w = 0; w < wave_id(); w++) my_rank[k] += (int)S.u.hist[w][myc[k]];
        mx = max(mx, my_rank[k]);
        if (my_rank[k] == 0) S.lead[myc[k]] = (unsigned short)(b_first + k * kWave);  // one writer per cell
      }
    if (mx > 0) atomicMax(&S.max_rank, mx);
    __syncthreads();  // the counts are dead: their area becomes the accumulators
    for (int i = tid; i < kSparseMax * 3; i += kImgThreads) S.u.acc[i] = 0.0;  // every leader starts from 0.0
#pragma unroll
    for (int k = 0; k < kPer; k++)
      if (my_rank[k] > 0) my_lead[k] = S.lead[myc[k]];
    __syncthreads();
    const int rounds = S.max_rank;
    for (int r = 0; r <= rounds; r++) {  // :166-179, list order within each cell
#pragma unroll
      for (int k = 0; k < kPer; k++)
        if (my_rank[k] == r) {
          const int l = my_lead[k];
          S.u.acc[3 * l] = S.u.acc[3 * l] + yv[k][0];
          S.u.acc[3 * l + 1] = S.u.acc[3 * l + 1] + yv[k][1];
          S.u.acc[3 * l + 2] = S.u.acc[3 * l + 2] + yv[k][2];
        }
      __syncthreads();
    }
    // leaders quantise their cell into a register; only when every accumulator has been read is the
    // area reused for the pixel map
    unsigned qv[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int b = b_first + k * kWave;
      qv[k] = (my_rank[k] == 0) ? quantise(S.u.acc[3 * b], S.u.acc[3 * b + 1], S.u.acc[3 * b + 2]) : 0u;
    }
    __syncthreads();
    for (int i = tid; i < kCells; i += kImgThreads) S.u.img.pix[i] = 0u;  // image.setTo(0)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++)
      if (my_rank[k] == 0) {  // written at (59 - row, col)
        const int cell = myc[k];
        const int row = kImg - 1 - cell / kImg, col = cell % kImg;
        S.u.img.pix[row * kImg + col] = qv[k];
      }
    __syncthreads();
    dilate_store(S.u.img.pix, S.u.img.obuf, out + (size_t)im * (kCells * 3), tid);
  }
}

// k_render_sorted (kSortedMax < P <= kSortedMaxBig, dense clouds; AG2_RENDER_BITONIC=1: also the range of
// k_render_counted below, for A/B): the points' (cell, list position)
// keys are sorted in LDS (bitonic, the position in the low bits keeps list order inside a cell), so
// every cell's points sit together; the thread that finds the head of a run adds the run's normals
// in list order -- the same f64 sums as the reference's scan over all points for each cell, at
// O(P log^2 P) instead of O(3600 P).
// Two instantiations: up to 16384 points (96 KB of LDS, one workgroup per CU) and up to 4096 (48 KB, three per
// CU: the A/B twin of k_render_counted); NBITS = bits of the list position inside a key.
constexpr int kSortedMax = 4096, kSortedMaxBig = 16384;
constexpr int kWalk = 768;         // sorted positions whose normals are staged in LDS at a time
template <int NMAX, int NT>
struct SortedShared {
  unsigned key[NMAX];              // cell << NBITS | position; 0xFFFFFFFF = dropped point / padding
  unsigned pix[kCells];
  double nbuf[kWalk * 3];          // normals in sorted order; the staged output image aliases it afterwards
  double red[NT / kWave];
  double carry[3];                 // sum of a run that continues into the next chunk
};
static_assert(kWalk * 3 * 8 >= kCells * 3, "output staging must fit the normal stage");
static_assert(kCells <= 4096, "12 bits for the cell");
static_assert(sizeof(SortedShared<kSortedMax, kImgThreads>) * 3 <= 160 * 1024, "k_render_sorted: three workgroups per CU");
static_assert(sizeof(SortedShared<kSortedMaxBig, 1024>) <= 160 * 1024, "k_render_sorted, big: one workgroup per CU");

// Bitonic sort (ascending) of PER * NT keys held PER per thread: element index = tid * PER + r.
// Compare-exchanges between registers of one thread need no memory at all (v_min / v_max), those
// between threads of a wave go through a lane shuffle, only partners in different waves exchange
// through LDS.  A phase that runs descending for this thread sorts the complemented keys ascending
// instead, so every compare-exchange is a plain (min, max).
template <int PER, int NT>
__device__ __forceinline__ void bitonic_sort_regs(unsigned (&x)[PER], unsigned* __restrict__ lds, int tid) {
  constexpr int N = PER * NT;
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
    unsigned flip = 0u;
    if (k >= PER) {
      flip = (((unsigned)tid * PER) & (unsigned)k) ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int r = 0; r < PER; r++) x[r] ^= flip;
    }
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < PER) {
#pragma unroll
        for (int r = 0; r < PER; r++) {
          if ((r & j) == 0) {
            const unsigned a = x[r], b = x[r | j];
            const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
            const bool up = (k >= PER) || ((r & k) == 0);  // known at compile time
            x[r] = up ? lo : hi;
            x[r | j] = up ? hi : lo;
          }
        }
      } else {
        const int m = j / PER;  // the partner thread holds the same r
        const bool lower = (tid & m) == 0;
        if (m < kWave) {
#pragma unroll
          for (int r = 0; r < PER; r++) {
            const unsigned p = (unsigned)__shfl_xor((int)x[r], m, kWave);
            const unsigned lo = x[r] < p ? x[r] : p, hi = x[r] < p ? p : x[r];
            x[r] = lower ? lo : hi;
          }
        } else {
          __syncthreads();  // earlier readers of the exchange area are done
#pragma unroll
          for (int r = 0; r < PER; r++) lds[r * NT + tid] = x[r];
          __syncthreads();
#pragma unroll
          for (int r = 0; r < PER; r++) {
            const unsigned p = lds[r * NT + (tid ^ m)];
            const unsigned lo = x[r] < p ? x[r] : p, hi = x[r] < p ? p : x[r];
            x[r] = lower ? lo : hi;
          }
        }
      }
    }
    if (k >= PER) {
#pragma unroll
      for (int r = 0; r < PER; r++) x[r] ^= flip;
    }
  }
  __syncthreads();  // the exchange area is free again: the caller stores the result there
#pragma unroll
  for (int r = 0; r < PER; r++) lds[tid * PER + r] = x[r];
}

template <int NMAX, int NBITS, int NT>
__global__ void __launch_bounds__(NT) k_render_sorted(const double* __restrict__ arena,
                                                               const long long* __restrict__ desc_off,
                                                               const int* __restrict__ desc_cnt,
                                                               int n_img, const unsigned* __restrict__ d_n,
                                                               int p_min, unsigned char* __restrict__ out, ImgQueue q) {
  static_assert((1 << NBITS) >= NMAX && NBITS + 12 < 32, "key layout");
  if (d_n) n_img = min(n_img, (int)*d_n);  // frame mode: the list length is read on the device
  __shared__ int s_qnext[2];
  int qit = 0;
  const int n_work = q.list ? (int)*q.count : n_img;
  constexpr unsigned kPosMask = (1u << NBITS) - 1u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_sorted[];
  SortedShared<NMAX, NT>& S = *reinterpret_cast<SortedShared<NMAX, NT>*>(smem_sorted);
  const int tid = threadIdx.x;
  for (int w = blockIdx.x; w < n_work; w = img_queue_next(q, w, s_qnext, qit)) {
    const int im = q.list ? q.list[w] : w;
    const long long off = desc_off[im];
    const int P = (off >= 0) ? desc_cnt[im] : 0;
    if (P < p_min || P > NMAX) continue;  // the other renderers' images (uniform)
    const double* pts = arena + (size_t)off * 6;
    __syncthreads();  // previous image's readers of S are done
    // x and y of this thread's points are read ONCE, all loads in flight together (the kernel is
    // bound by memory latency, not bandwidth), and kept in registers for the cell ids
    constexpr int kPer = NMAX / NT;
    double px[kPer], py[kPer];
    double miny = __builtin_inf();  // learning.cpp:148-149  y <- y - min y
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int b = tid + k * NT;
      px[k] = 0.0;
      py[k] = __builtin_inf();
      if (b < P) {
        px[k] = pts[(size_t)b * 6];
        py[k] = pts[(size_t)b * 6 + 1];
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) miny = (py[k] < miny) ? py[k] : miny;
    miny = wave_min_d(miny);
    if (lane_id() == 0) S.red[wave_id()] = miny;
    __syncthreads();
    miny = S.red[0];
#pragma unroll
    for (int k = 1; k < NT / kWave; k++) miny = (S.red[k] < miny) ? S.red[k] : miny;
    unsigned key[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int b = tid + k * NT;
      key[k] = 0xFFFFFFFFu;
      if (b < P) {
        const short c = cell_id(px[k], py[k], miny);
        if (c >= 0) key[k] = ((unsigned)c << NBITS) | (unsigned)b;
      }
    }
    for (int i = tid; i < kCells; i += NT) S.pix[i] = 0u;  // image.setTo(0)
    if (P <= NMAX / 2) {  // uniform: half the keys would be padding -- sort half as many
      unsigned half[kPer / 2];
#pragma unroll
      for (int k = 0; k < kPer / 2; k++) half[k] = key[k];
      bitonic_sort_regs<kPer / 2, NT>(half, S.key, tid);
    } else {
      bitonic_sort_regs<kPer, NT>(key, S.key, tid);
    }
    // Runs of equal cells, kWalk sorted positions at a time.  The normals of a chunk are gathered
    // into LDS by all threads first (independent loads), so the thread that owns a run -- the one at
    // its head -- adds them in list order (:166-179) without waiting for memory; a run that crosses
    // the chunk boundary hands its partial sum on through S.carry.
    for (int c0 = 0; c0 < P; c0 += kWalk) {
      const int cn = min(kWalk, P - c0);
      __syncthreads();  // previous chunk's readers of nbuf are done, its carry is written
      {
        constexpr int kG = (kWalk + NT - 1) / NT;  // gathers per thread: issued together, then stored
        double g[kG][3];
#pragma unroll
        for (int u = 0; u < kG; u++) {
          const int t = tid + u * NT;
          const unsigned k = (t < cn) ? S.key[c0 + t] : 0xFFFFFFFFu;
          g[u][0] = g[u][1] = g[u][2] = 0.0;
          if (k != 0xFFFFFFFFu) {
            const double* y = pts + (size_t)(k & kPosMask) * 6 + 3;
            g[u][0] = y[0];
            g[u][1] = y[1];
            g[u][2] = y[2];
          }
        }
#pragma unroll
        for (int u = 0; u < kG; u++) {
          const int t = tid + u * NT;
          if (t < cn) {
            S.nbuf[3 * t] = g[u][0];
            S.nbuf[3 * t + 1] = g[u][1];
            S.nbuf[3 * t + 2] = g[u][2];
          }
        }
      }
      __syncthreads();
      for (int t = tid; t < cn; t += NT) {
        const unsigned k = S.key[c0 + t];
        if (k == 0xFFFFFFFFu) continue;
        const unsigned cell = k >> NBITS;
        const bool cont = (c0 + t > 0) && (S.key[c0 + t - 1] >> NBITS) == cell;
        if (cont && t != 0) continue;  // inside a run: its head does the work
        double ax = 0.0, ay = 0.0, az = 0.0;
        if (cont) {  // first position of the chunk, run started in the previous one
          ax = S.carry[0];
          ay = S.carry[1];
          az = S.carry[2];
        }
        int j = t;
        unsigned kk;
        do {
          ax = ax + S.nbuf[3 * j];
          ay = ay + S.nbuf[3 * j + 1];
          az = az + S.nbuf[3 * j + 2];
          j++;
          kk = (c0 + j < P) ? S.key[c0 + j] : 0xFFFFFFFFu;
        } while (j < cn && (kk >> NBITS) == cell);
        if (j == cn && (kk >> NBITS) == cell) {  // continues in the next chunk (at most one such run)
          S.carry[0] = ax;
          S.carry[1] = ay;
          S.carry[2] = az;
        } else {  // quantise, write at (59 - row, col)
          const int row = kImg - 1 - (int)cell / kImg, col = (int)cell % kImg;
          S.pix[row * kImg + col] = quantise(ax, ay, az);
        }
      }
    }
    __syncthreads();
    dilate_store<NT>(S.pix, reinterpret_cast<unsigned char*>(S.nbuf), out + (size_t)im * (kCells * 3), tid);
  }
}

// k_render_counted (kSparseMax < P <= kSortedMax): the same cell-major order by COUNTING instead of comparing.
// Each wave owns a contiguous quarter of the list and takes it in chunks of 64 consecutive positions.  In
// a chunk the lanes that share a cell find each other with twelve ballots (one per bit of the cell id: after
// bit k a lane's mask keeps the lanes whose bit k equals its own), which gives a point its rank among the
// chunk's points of its cell -- list order, data-independent cost: a cell with hundreds of points (a surface
// seen edge-on) costs what sixty-four different cells cost.  Rank in the chunk + the wave's running count
// of the cell (one u16 per wave and cell in LDS) + the counts of the waves before it + the cell's first
// slot (a scan over the 3 600 cell totals) = the point's slot in (cell, list position) order: a stable
// counting sort, ~1 400 instructions per wave for 2 300 points where the bitonic network takes ~4 000.
// The walk then has every cell's run as a pair of slots, so a thread adds a run's normals in a COUNTED
// loop (the loads of the next terms run ahead of the chain of additions; the network version found a
// run's end by reading keys as it went -- one LDS round trip per term, the longest run of a chunk on the
// critical path).  Cells are dealt to threads modulo 256; the sums are the reference's, term by term.
constexpr int kCWaves = kImgThreads / kWave;
struct CountedShared {
  unsigned short pos[kSortedMax];       // list positions in (cell, list position) order
  unsigned short start[kCells + 8];     // first slot of every cell; start[kCells] = points kept
  union {
    unsigned short hist[kCWaves][kCells];  // sort: running count / first slot of (wave, cell)
    struct {
      unsigned pix[kCells];
      double nbuf[kWalk * 3];              // normals of kWalk slots; the staged output image afterwards
    } w;
  } u;
  double red[kCWaves];
  int part[kCWaves];
};
static_assert(sizeof(CountedShared) * 3 <= 160 * 1024, "k_render_counted: three workgroups per CU");
static_assert(kCells % 15 == 0 && kCells / 15 <= kImgThreads, "scan: 15 cells per thread");
static_assert(kSortedMax / kCWaves / kWave == 16, "sixteen chunks per wave at most");

__global__ void __launch_bounds__(kImgThreads, 3) k_render_counted(const double* __restrict__ arena,
                                                                const long long* __restrict__ desc_off,
                                                                const int* __restrict__ desc_cnt, int n_img,
                                                                const unsigned* __restrict__ d_n, int p_min,
                                                                unsigned char* __restrict__ out, ImgQueue q) {
  if (d_n) n_img = min(n_img, (int)*d_n);  // frame mode: the list length is read on the device
  __shared__ CountedShared S;
  __shared__ int s_qnext[2];
  int qit = 0;
  const int n_work = q.list ? (int)*q.count : n_img;
  constexpr int NT = kImgThreads, NW = kCWaves, kChunks = 16;
  const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int w = blockIdx.x; w < n_work; w = img_queue_next(q, w, s_qnext, qit)) {
    const int im = q.list ? q.list[w] : w;
    const long long off = desc_off[im];
    const int P = (off >= 0) ? desc_cnt[im] : 0;
    if (P < p_min || P > kSortedMax) continue;  // the other renderers' images (uniform)
    const double* pts = arena + (size_t)off * 6;
    __syncthreads();  // previous image's readers of S are done
    // this wave's quarter of the list: seg positions (a multiple of 64) from wid * seg on
    const int seg = ((P + NT - 1) / NT) * kWave, nc = seg / kWave;
    const int b0 = wid * seg + lane;
    double px[kChunks], py[kChunks];
    double miny = __builtin_inf();  // learning.cpp:148-149  y <- y - min y
#pragma unroll
    for (int c = 0; c < kChunks; c++) {
      const int b = b0 + c * kWave;
      px[c] = 0.0;
      py[c] = __builtin_inf();
      if (c < nc && b < P) {
        px[c] = pts[(size_t)b * 6];
        py[c] = pts[(size_t)b * 6 + 1];
      }
    }
    for (int i = tid; i < NW * kCells / 2; i += NT) reinterpret_cast<unsigned*>(&S.u.hist[0][0])[i] = 0u;
#pragma unroll
    for (int c = 0; c < kChunks; c++) miny = (py[c] < miny) ? py[c] : miny;
    miny = wave_min_d(miny);
    if (lane == 0) S.red[wid] = miny;
    __syncthreads();  // (also: the counts are zero)
    miny = S.red[0];
#pragma unroll
    for (int k = 1; k < NW; k++) miny = (S.red[k] < miny) ? S.red[k] : miny;
    // cell and rank of this thread's points (rank: among the points of its cell in this wave's quarter)
    short cid[kChunks];
    unsigned short rank[kChunks];
#pragma unroll
    for (int c = 0; c < kChunks; c++) {
      cid[c] = -1;
      rank[c] = 0;
      if (c < nc) {  // uniform
        const int b = b0 + c * kWave;
        if (b < P) cid[c] = cell_id(px[c], py[c], miny);
        const bool valid = cid[c] >= 0;
        const int cell = valid ? (int)cid[c] : 0;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 12; bit++) {
          const bool mine = (cell >> bit) & 1;
          const unsigned long long bb = __ballot(mine);
          m &= mine ? bb : ~bb;
        }
        if (valid) {
          const int before = (int)S.u.hist[wid][cell];
          rank[c] = (unsigned short)(before + __popcll(m & lt_mask));
          if ((m >> lane) == 1ull) S.u.hist[wid][cell] = (unsigned short)(before + __popcll(m));  // the group's last lane
        }
      }
    }
    __syncthreads();
    // first slot of every cell (scan over the cell totals), then of every (wave, cell)
    {
      constexpr int kPerT = 15;
      const int c0 = tid * kPerT;
      int tot[kPerT], sum = 0;
      if (c0 < kCells) {
#pragma unroll
        for (int q = 0; q < kPerT; q++) {
          int t = 0;
#pragma unroll
          for (int w = 0; w < NW; w++) t += (int)S.u.hist[w][c0 + q];
          tot[q] = t;
          sum += t;
        }
      }
      int incl = sum;  // inclusive scan over the wave
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
      }
      if (lane == kWave - 1) S.part[wid] = incl;
      __syncthreads();
      int base = incl - sum;
#pragma unroll
      for (int w = 0; w < NW; w++)
        if (w < wid) base += S.part[w];
      if (c0 < kCells) {
#pragma unroll
        for (int q = 0; q < kPerT; q++) {
          S.start[c0 + q] = (unsigned short)base;
          int run = base;
#pragma unroll
          for (int w = 0; w < NW; w++) {
            const int h = (int)S.u.hist[w][c0 + q];
            S.u.hist[w][c0 + q] = (unsigned short)run;
            run += h;
          }
          base += tot[q];
        }
        if (c0 + kPerT == kCells) S.start[kCells] = (unsigned short)base;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kChunks; c++)
      if (c < nc && cid[c] >= 0) S.pos[(int)S.u.hist[wid][cid[c]] + (int)rank[c]] = (unsigned short)(b0 + c * kWave);
    __syncthreads();  // the counts are dead: their area becomes the pixel map and the normal stage
    const int kept = (int)S.start[kCells];
    for (int i = tid; i < kCells; i += NT) S.u.w.pix[i] = 0u;  // image.setTo(0)
    // Runs of equal cells, kWalk slots at a time.  The normals of a window are gathered into LDS by all
    // threads first (independent loads); a thread then adds the runs of ITS cells (cell % 256 == tid) that lie
    // in the window, in slot order (:166-179).  A run that crosses the window's end stays open in the thread's
    // accumulators: it is the last of the window, and the first of the next one for the same thread.
    double ax = 0.0, ay = 0.0, az = 0.0;
    constexpr int kG = (kWalk + NT - 1) / NT;  // gathers per thread: issued together, stored a window later
    double g[kG][3];
    auto gather = [&](int c0) {  // requests the normals of the window that starts at slot c0
      const int cn = min(kWalk, kept - c0);
#pragma unroll
      for (int u = 0; u < kG; u++) {
        const int t = tid + u * NT;
        g[u][0] = g[u][1] = g[u][2] = 0.0;
        if (t < cn) {
          const double* y = pts + (size_t)S.pos[c0 + t] * 6 + 3;
          g[u][0] = y[0];
          g[u][1] = y[1];
          g[u][2] = y[2];
        }
      }
    };
    if (kept > 0) gather(0);
    for (int c0 = 0; c0 < kept; c0 += kWalk) {
      const int cn = min(kWalk, kept - c0);
      __syncthreads();  // previous window's readers of nbuf are done (first window: the pixel map is clear)
#pragma unroll
      for (int u = 0; u < kG; u++) {
        const int t = tid + u * NT;
        if (t < cn) {
          S.u.w.nbuf[3 * t] = g[u][0];
          S.u.w.nbuf[3 * t + 1] = g[u][1];
          S.u.w.nbuf[3 * t + 2] = g[u][2];
        }
      }
      __syncthreads();
      if (c0 + kWalk < kept) gather(c0 + kWalk);  // the next window's normals travel while this one is added up
      for (int cell = tid; cell < kCells; cell += NT) {
        const int s0 = (int)S.start[cell], s1 = (int)S.start[cell + 1];
        const int a = max(s0, c0), b = min(s1, c0 + cn);
        if (a >= b) continue;
        if (a == s0) ax = ay = az = 0.0;
        const double* nb = S.u.w.nbuf + 3 * (a - c0);
        const int len = b - a;
        for (int j = 0; j < len; j++) {
          ax = ax + nb[3 * j];
          ay = ay + nb[3 * j + 1];
          az = az + nb[3 * j + 2];
        }
        if (b == s1) {  // the run is complete: quantise, write at (59 - row, col)
          const int row = kImg - 1 - cell / kImg, col = cell % kImg;
          S.u.w.pix[row * kImg + col] = quantise(ax, ay, az);
        }
      }
    }
    __syncthreads();
    dilate_store<NT>(S.u.w.pix, reinterpret_cast<unsigned char*>(S.u.w.nbuf), out + (size_t)im * (kCells * 3), tid);
  }
}

// max_p: an upper bound of the images' point counts (the sweep's statistics have it): renderers
// none of whose images can occur are not launched.
// largest point count the renderers launched for an upper bound of max_p can take
int render_capacity_for(int max_p) {
  if (max_p <= kSparseMax) return kSparseMax;
  if (max_p <= kSortedMax) return kSortedMax;
  if (max_p <= kSortedMaxBig) return kSortedMaxBig;
  return 0x7fffffff;
}

// d_n (frame mode): n_img is the capacity of the list, its length is read from *d_n on the device.
int launch_render(ag2_ctx* c, const double* d_arena, const long long* d_off, const int* d_cnt,
                  size_t n_img, uint8_t* d_out, int max_p, const unsigned* d_n) {
  if (n_img == 0) return 0;
  const size_t lds = sizeof(ImgShared);
  if (!(c->func_attr_done & kAttrRender)) {  // (per context = per device)
    AG2_HIP(c, hipFuncSetAttribute((const void*)k_render, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    AG2_HIP(c, hipFuncSetAttribute((const void*)k_render_sorted<kSortedMaxBig, 14, 1024>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(SortedShared<kSortedMaxBig, 1024>)));
    c->func_attr_done |= kAttrRender;
  }
  // images with at most kSparseMax points (nearly all) ...
  hipLaunchKernelGGL(k_render_sparse, dim3((int)std::min<size_t>(n_img, 256 * 6)), dim3(kImgThreads), 0,
                     c->stream, d_arena, d_off, d_cnt, (int)n_img, d_n, d_out);
  // ... the rest: one list per renderer, handed out image by image (ImgQueue)
  static const bool bitonic = getenv("AG2_RENDER_BITONIC") != nullptr;  // (A/B: the sorting-network renderer)
  static const bool no_queue = getenv("AG2_RENDER_STATIC") != nullptr;  // (A/B: images dealt statically)
  ImgQueue q[3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  if (max_p > kSparseMax && !no_queue) {
    AG2_HIP(c, c->d_rlist.reserve(32 + 3 * n_img * 4));
    unsigned* ctr = c->d_rlist.as<unsigned>();
    int* lists = reinterpret_cast<int*>(ctr + 8);
    AG2_HIP(c, hipMemsetAsync(ctr, 0, 32, c->stream));
    hipLaunchKernelGGL(k_render_classify, dim3((unsigned)((n_img + 255) / 256)), dim3(256), 0, c->stream, d_off, d_cnt,
                       (int)n_img, d_n, (int)n_img, ctr, lists);
    for (int k = 0; k < 3; k++) q[k] = ImgQueue{lists + (size_t)k * n_img, ctr + k, ctr + 4 + k};
  }
  if (max_p > kSparseMax && !bitonic)
    hipLaunchKernelGGL(k_render_counted, dim3((int)std::min<size_t>(n_img, 256 * 3)), dim3(kImgThreads), 0, c->stream,
                       d_arena, d_off, d_cnt, (int)n_img, d_n, kSparseMax + 1, d_out, q[0]);
  if (max_p > kSparseMax && bitonic)
    hipLaunchKernelGGL((k_render_sorted<kSortedMax, 12, kImgThreads>), dim3((int)std::min<size_t>(n_img, 256 * 3)),
                       dim3(kImgThreads), sizeof(SortedShared<kSortedMax, kImgThreads>), c->stream, d_arena, d_off, d_cnt,
                       (int)n_img, d_n, kSparseMax + 1, d_out, q[0]);
  if (max_p > kSortedMax)
    hipLaunchKernelGGL((k_render_sorted<kSortedMaxBig, 14, 1024>), dim3((int)std::min<size_t>(n_img, 256)),
                       dim3(1024), sizeof(SortedShared<kSortedMaxBig, 1024>), c->stream, d_arena, d_off,
                       d_cnt, (int)n_img, d_n, kSortedMax + 1, d_out, q[1]);
  if (max_p > kSortedMaxBig)
    hipLaunchKernelGGL(k_render, dim3((int)std::min<size_t>(n_img, 256 * 2)), dim3(kImgThreads), lds, c->stream,
                       d_arena, d_off, d_cnt, (int)n_img, d_n, kSortedMaxBig + 1, d_out);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
