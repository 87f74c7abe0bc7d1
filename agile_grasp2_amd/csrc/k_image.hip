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
// cells within a round, list order across rounds -- and only the leaders quantise a pixel.  43 KB of
// LDS: three workgroups per CU.
// k_render_sorted (1024 < P <= 4096): cell-major sort of the points in LDS, then one run per cell.
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

// :202-203 3x3 rect dilate (border taps ignored), :206 BGR2RGB swap; staged so that the global
// store is coalesced dwords
__device__ __forceinline__ void dilate_store(const unsigned* __restrict__ pix,
                                             unsigned char* __restrict__ obuf,
                                             unsigned char* __restrict__ out_img, int tid) {
  for (int p = tid; p < kCells; p += kImgThreads) {
    const int r = p / kImg, cc = p % kImg;
    unsigned m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
    for (int dr = -1; dr <= 1; dr++)
#pragma unroll
      for (int dc = -1; dc <= 1; dc++) {
        const int rr = r + dr, c2 = cc + dc;
        if (rr >= 0 && rr < kImg && c2 >= 0 && c2 < kImg) {
          const unsigned v = pix[rr * kImg + c2];
          m0 = max(m0, v & 255u);
          m1 = max(m1, (v >> 8) & 255u);
          m2 = max(m2, (v >> 16) & 255u);
        }
      }
    obuf[p * 3 + 2] = (unsigned char)m0;
    obuf[p * 3 + 1] = (unsigned char)m1;
    obuf[p * 3 + 0] = (unsigned char)m2;
  }
  __syncthreads();
  unsigned* dst = reinterpret_cast<unsigned*>(out_img);
  const unsigned* src = reinterpret_cast<const unsigned*>(obuf);
  for (int i = tid; i < kCells * 3 / 4; i += kImgThreads) dst[i] = src[i];
}

__device__ __forceinline__ double block_min_y(const double* __restrict__ pts, int P, double* red, int tid) {
  double miny = __builtin_inf();  // learning.cpp:148-149  y <- y - min y
  for (int b = tid; b < P; b += kImgThreads) {
    const double y = pts[(size_t)b * 6 + 1];
    miny = (y < miny) ? y : miny;
  }
  miny = wave_min_d(miny);
  if (lane_id() == 0) red[wave_id()] = miny;
  __syncthreads();
  miny = red[0];
#pragma unroll
  for (int k = 1; k < kImgThreads / kWave; k++) miny = (red[k] < miny) ? red[k] : miny;
  return miny;
}

__global__ void __launch_bounds__(kImgThreads) k_render(const double* __restrict__ arena,
                                                        const long long* __restrict__ desc_off,
                                                        const int* __restrict__ desc_cnt, int n_img,
                                                        int p_min, unsigned char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ImgShared& S = *reinterpret_cast<ImgShared*>(smem_raw);
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

struct SparseShared {
  double acc[kSparseMax * 3];     // per leader point; the staged output image aliases it afterwards
  unsigned pix[kCells];
  short cid[kSparseMax];
  short lead[kSparseMax];
  unsigned char rank[kSparseMax];
  double red[kImgThreads / kWave];
  int max_rank;
};
static_assert(sizeof(SparseShared) * 3 <= 160 * 1024, "k_render_sparse: three workgroups per CU");
static_assert(kSparseMax * 3 * 8 >= kCells * 3, "output staging must fit the accumulator area");

__global__ void __launch_bounds__(kImgThreads) k_render_sparse(const double* __restrict__ arena,
                                                               const long long* __restrict__ desc_off,
                                                               const int* __restrict__ desc_cnt,
                                                               int n_img, unsigned char* __restrict__ out) {
  __shared__ SparseShared S;
  const int tid = threadIdx.x;
  constexpr int kPer = kSparseMax / kImgThreads;  // points per thread
  for (int im = blockIdx.x; im < n_img; im += gridDim.x) {
    const long long off = desc_off[im];
    const int P = (off >= 0) ? desc_cnt[im] : 0;
    if (P > kSparseMax) continue;  // taken by k_render (uniform)
    const double* pts = arena + (size_t)(off >= 0 ? off : 0) * 6;
    __syncthreads();  // previous image's readers of S are done
    const double miny = block_min_y(pts, P, S.red, tid);
    double yv[kPer][3];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int b = tid + k * kImgThreads;
      short c = -1;
      yv[k][0] = yv[k][1] = yv[k][2] = 0.0;
      if (b < P) {
        const double* p = pts + (size_t)b * 6;
        c = cell_id(p[0], p[1], miny);
        yv[k][0] = p[3]; yv[k][1] = p[4]; yv[k][2] = p[5];
      }
      S.cid[b] = c;
    }
    for (int i = tid; i < kCells; i += kImgThreads) S.pix[i] = 0u;  // image.setTo(0)
    if (tid == 0) S.max_rank = 0;
    __syncthreads();
    // rank of every point among the earlier points of its cell, and the cell's first point
    int my_rank[kPer], my_lead[kPer], mx = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int b = tid + k * kImgThreads;
      my_rank[k] = 0;
      my_lead[k] = b;
      if (b < P) {
        const int c = S.cid[b];
        if (c >= 0) {
          int r = 0, first = b;
          for (int b0 = 0; b0 < b; b0 += 8) {  // 8 ids per LDS read; entries >= b are masked off
            const uint4 w = *reinterpret_cast<const uint4*>(&S.cid[b0]);
            const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int id = (int)(short)((ww[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
              if (id == c && b0 + j < b) {
                first = min(first, b0 + j);
                r++;
              }
            }
          }
          my_rank[k] = r;
          my_lead[k] = first;
          mx = max(mx, r);
        } else {
          my_rank[k] = -1;  // dropped point (:156)
        }
      } else {
        my_rank[k] = -1;
      }
      if (b < kSparseMax) {  // every leader starts from 0.0 like the reference's running sum
        S.acc[3 * b] = 0.0;
        S.acc[3 * b + 1] = 0.0;
        S.acc[3 * b + 2] = 0.0;
      }
    }
    if (mx > 0) atomicMax(&S.max_rank, mx);
    __syncthreads();
    const int rounds = S.max_rank;
    for (int r = 0; r <= rounds; r++) {  // :166-179, list order within each cell
#pragma unroll
      for (int k = 0; k < kPer; k++)
        if (my_rank[k] == r) {
          const int l = my_lead[k];
          S.acc[3 * l] = S.acc[3 * l] + yv[k][0];
          S.acc[3 * l + 1] = S.acc[3 * l + 1] + yv[k][1];
          S.acc[3 * l + 2] = S.acc[3 * l + 2] + yv[k][2];
        }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kPer; k++)
      if (my_rank[k] == 0) {  // leaders: written at (59 - row, col)
        const int b = tid + k * kImgThreads;
        const int cell = S.cid[b];
        const int row = kImg - 1 - cell / kImg, col = cell % kImg;
        S.pix[row * kImg + col] = quantise(S.acc[3 * b], S.acc[3 * b + 1], S.acc[3 * b + 2]);
      }
    __syncthreads();
    dilate_store(S.pix, reinterpret_cast<unsigned char*>(S.acc), out + (size_t)im * (kCells * 3), tid);
  }
}

// k_render_sorted (kSparseMax < P <= kSortedMax, dense clouds): the points' (cell, list position)
// keys are sorted in LDS (bitonic, the position in the low bits keeps list order inside a cell), so
// every cell's points sit together; the thread that finds the head of a run adds the run's normals
// in list order -- the same f64 sums as the reference's scan over all points for each cell, at
// O(P log^2 P) instead of O(3600 P).
constexpr int kSortedMax = 4096;
struct SortedShared {
  unsigned key[kSortedMax];        // cell << 12 | position; 0xFFFFFFFF = dropped point / padding
  unsigned pix[kCells];
  unsigned char obuf[kCells * 3];
  double red[kImgThreads / kWave];
};
static_assert(kCells <= 4096 && kSortedMax <= 4096, "12 bits each for cell and position");
static_assert(sizeof(SortedShared) * 3 <= 160 * 1024, "k_render_sorted: three workgroups per CU");

__global__ void __launch_bounds__(kImgThreads) k_render_sorted(const double* __restrict__ arena,
                                                               const long long* __restrict__ desc_off,
                                                               const int* __restrict__ desc_cnt,
                                                               int n_img, int p_min,
                                                               unsigned char* __restrict__ out) {
  __shared__ SortedShared S;
  const int tid = threadIdx.x;
  for (int im = blockIdx.x; im < n_img; im += gridDim.x) {
    const long long off = desc_off[im];
    const int P = (off >= 0) ? desc_cnt[im] : 0;
    if (P < p_min || P > kSortedMax) continue;  // the other renderers' images (uniform)
    const double* pts = arena + (size_t)off * 6;
    __syncthreads();  // previous image's readers of S are done
    const double miny = block_min_y(pts, P, S.red, tid);
    int N = 2048;
    while (N < P) N <<= 1;
    for (int b = tid; b < N; b += kImgThreads) {
      unsigned k = 0xFFFFFFFFu;
      if (b < P) {
        const short c = cell_id(pts[(size_t)b * 6], pts[(size_t)b * 6 + 1], miny);
        if (c >= 0) k = ((unsigned)c << 12) | (unsigned)b;
      }
      S.key[b] = k;
    }
    for (int i = tid; i < kCells; i += kImgThreads) S.pix[i] = 0u;  // image.setTo(0)
    __syncthreads();
    for (int k = 2; k <= N; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < N / 2; t += kImgThreads) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int l = i | j;
          const unsigned a = S.key[i], b = S.key[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            S.key[i] = b;
            S.key[l] = a;
          }
        }
        __syncthreads();
      }
    // heads of runs: sum the run in list order (:166-179), quantise, write at (59 - row, col)
    for (int i = tid; i < P; i += kImgThreads) {
      const unsigned k = S.key[i];
      if (k == 0xFFFFFFFFu) continue;
      const unsigned cell = k >> 12;
      if (i > 0 && (S.key[i - 1] >> 12) == cell) continue;
      double ax = 0.0, ay = 0.0, az = 0.0;
      unsigned kk = k;
      int j = i;
      do {
        const double* y = pts + (size_t)(kk & 4095u) * 6 + 3;
        ax = ax + y[0];
        ay = ay + y[1];
        az = az + y[2];
        j++;
        kk = (j < P) ? S.key[j] : 0xFFFFFFFFu;
      } while ((kk >> 12) == cell);
      const int row = kImg - 1 - (int)cell / kImg, col = (int)cell % kImg;
      S.pix[row * kImg + col] = quantise(ax, ay, az);
    }
    __syncthreads();
    dilate_store(S.pix, S.obuf, out + (size_t)im * (kCells * 3), tid);
  }
}

// max_p: an upper bound of the images' point counts (the sweep's statistics have it): renderers
// none of whose images can occur are not launched.
int launch_render(ag2_ctx* c, const double* d_arena, const long long* d_off, const int* d_cnt,
                  size_t n_img, uint8_t* d_out, int max_p) {
  if (n_img == 0) return 0;
  const size_t lds = sizeof(ImgShared);
  static bool attr_set = false;
  if (!attr_set) {
    AG2_HIP(c, hipFuncSetAttribute((const void*)k_render, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    attr_set = true;
  }
  // images with at most kSparseMax points (nearly all) ...
  hipLaunchKernelGGL(k_render_sparse, dim3((int)std::min<size_t>(n_img, 256 * 12)), dim3(kImgThreads), 0,
                     c->stream, d_arena, d_off, d_cnt, (int)n_img, d_out);
  // ... the rest; each kernel skips the others' images by the point count alone
  if (max_p > kSparseMax)
    hipLaunchKernelGGL(k_render_sorted, dim3((int)std::min<size_t>(n_img, 256 * 3)), dim3(kImgThreads), 0,
                     c->stream, d_arena, d_off, d_cnt, (int)n_img, kSparseMax + 1, d_out);
  if (max_p > kSortedMax)
    hipLaunchKernelGGL(k_render, dim3((int)std::min<size_t>(n_img, 256 * 2)), dim3(kImgThreads), lds, c->stream,
                     d_arena, d_off, d_cnt, (int)n_img, kSortedMax + 1, d_out);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
