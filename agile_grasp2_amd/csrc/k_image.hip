// k_image.hip -- K4: grasp image renderer.
//
// Replaces Learning::createGraspImages / convertToImageRGB (src/agile_grasp2/learning.cpp:4-33,
// :143-209) and the convertTo(CV_8UC3, 255.0) at :16.  The reference scans all P points for each of
// the 3600 cells; here every cell is owned by one thread (cell % 256), which adds the normals of
// its points in list order -- the same per-cell summation order as the reference's inner loop, with
// no atomics, so the f64 sums (and therefore the bytes) are deterministic.
//
// One 256-thread workgroup per image.  The image is accumulated in two halves of 30 rows so that the
// f64 accumulators take 43 KB instead of 86 KB and two workgroups fit a CU (76 KB each): 1800 x 3
// f64 accumulators, a 4096-entry chunk of cell ids, the packed pre-dilation image and the staged
// output.  Quantisation to u8 happens BEFORE the 3x3 dilate: v -> sat(rint(255 v)) is monotone, so
// max-then-quantise == quantise-then-max, bit for bit.
#include "ag2_internal.h"

namespace ag2 {

constexpr int kImgThreads = 256;
constexpr int kCidChunk = 4096;
constexpr int kCells = kImg * kImg;
constexpr int kHalfCells = kCells / 2;

struct ImgShared {
  double acc[kHalfCells * 3];
  unsigned pix[kCells];
  short cid[kCidChunk];
  double red[kImgThreads / kWave];
  unsigned char obuf[kCells * 3];  // output image staged for coalesced dword stores
};
static_assert(sizeof(ImgShared) * 2 <= 160 * 1024, "k_render: two workgroups per CU");

__global__ void __launch_bounds__(kImgThreads) k_render(const double* __restrict__ arena,
                                                        const long long* __restrict__ desc_off,
                                                        const int* __restrict__ desc_cnt, int n_img,
                                                        unsigned char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ImgShared& S = *reinterpret_cast<ImgShared*>(smem_raw);
  const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
  // accumulators are zeroed once; every quantisation pass re-zeroes the cells it consumed
  for (int i = tid; i < kHalfCells * 3; i += kImgThreads) S.acc[i] = 0.0;
  for (int im = blockIdx.x; im < n_img; im += gridDim.x) {
    const long long off = desc_off[im];
    const int P = (off >= 0) ? desc_cnt[im] : 0;
    const double* pts = arena + (size_t)(off >= 0 ? off : 0) * 6;
    __syncthreads();
    // learning.cpp:148-149  y <- y - min y
    double miny = __builtin_inf();
    for (int b = tid; b < P; b += kImgThreads) {
      const double y = pts[(size_t)b * 6 + 1];
      miny = (y < miny) ? y : miny;
    }
    miny = wave_min_d(miny);
    if (lane == 0) S.red[wid] = miny;
    __syncthreads();
    miny = S.red[0];
#pragma unroll
    for (int k = 1; k < kImgThreads / kWave; k++) miny = (S.red[k] < miny) ? S.red[k] : miny;
    const double cellsize = 1.0 / (double)kImg;  // :152
    const bool one_chunk = P <= kCidChunk;       // usual case: cell ids computed once for both halves
    for (int half = 0; half < 2; half++) {
      const int cell_lo = half * kHalfCells;
      for (int c0 = 0; c0 < P; c0 += kCidChunk) {
        const int cn = min(kCidChunk, P - c0);
        if (!(one_chunk && half == 1)) {
          __syncthreads();
          for (int b = tid; b < ((cn + 7) & ~7); b += kImgThreads) {
            short cell = -1;  // also pads the chunk to a multiple of 8 ids
            if (b < cn) {
              const double ux = pts[(size_t)(c0 + b) * 6], uy = pts[(size_t)(c0 + b) * 6 + 1];
              const double fx = __builtin_floor(ux / cellsize);           // :153-156
              const double fy = __builtin_floor((uy - miny) / cellsize);
              if (__builtin_fabs(fx) < 1.0e9 && __builtin_fabs(fy) < 1.0e9) {
                const long long cl = (long long)fx + (long long)fy * kImg;  // x-cells >= 60 alias
                if (cl >= 0 && cl < kCells) cell = (short)cl;
              }
            }
            S.cid[b] = cell;
          }
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
      // :181-190 avg <- |avg / ||avg|||, written at (59 - row, col).  Only the owner of a cell ever
      // touches its accumulator, so no barrier is needed between accumulation and this pass.
      // An all-zero sum is either an empty cell (image.setTo(0)) or the 0/0 -> NaN -> 0 case.
      for (int hc = tid; hc < kHalfCells; hc += kImgThreads) {
        const double ax = S.acc[3 * hc], ay = S.acc[3 * hc + 1], az = S.acc[3 * hc + 2];
        unsigned packed = 0;
        if (ax != 0.0 || ay != 0.0 || az != 0.0) {
          S.acc[3 * hc] = 0.0;
          S.acc[3 * hc + 1] = 0.0;
          S.acc[3 * hc + 2] = 0.0;
          const double s = 1.0 / __builtin_sqrt((ax * ax + ay * ay) + az * az);
          const double v[3] = {__builtin_fabs(s * ax), __builtin_fabs(s * ay), __builtin_fabs(s * az)};
#pragma unroll
          for (int ch = 0; ch < 3; ch++) {
            const float f = (float)v[ch];
            const float tq = f * 255.0f;                               // convertTo(.., 255.0), :16
            unsigned u = 0;
            if (tq == tq) {
              const float r = __builtin_rintf(tq);                     // cvRound: half to even
              u = (unsigned)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
            }
            packed |= u << (8 * ch);
          }
        }
        const int cell = hc + cell_lo;
        const int row = kImg - 1 - cell / kImg, col = cell % kImg;
        S.pix[row * kImg + col] = packed;
      }
    }
    __syncthreads();
    // :202-203 3x3 rect dilate (border taps ignored), :206 BGR2RGB swap; staged in LDS so the
    // global store is coalesced dwords
    for (int p = tid; p < kCells; p += kImgThreads) {
      const int r = p / kImg, cc = p % kImg;
      unsigned m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
      for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++) {
          const int rr = r + dr, c2 = cc + dc;
          if (rr >= 0 && rr < kImg && c2 >= 0 && c2 < kImg) {
            const unsigned v = S.pix[rr * kImg + c2];
            m0 = max(m0, v & 255u);
            m1 = max(m1, (v >> 8) & 255u);
            m2 = max(m2, (v >> 16) & 255u);
          }
        }
      S.obuf[p * 3 + 2] = (unsigned char)m0;
      S.obuf[p * 3 + 1] = (unsigned char)m1;
      S.obuf[p * 3 + 0] = (unsigned char)m2;
    }
    __syncthreads();
    unsigned* dst = reinterpret_cast<unsigned*>(out + (size_t)im * (kCells * 3));
    const unsigned* src = reinterpret_cast<const unsigned*>(S.obuf);
    for (int i = tid; i < kCells * 3 / 4; i += kImgThreads) dst[i] = src[i];
  }
}

int launch_render(ag2_ctx* c, const double* d_arena, const long long* d_off, const int* d_cnt,
                  size_t n_img, uint8_t* d_out) {
  if (n_img == 0) return 0;
  const size_t lds = sizeof(ImgShared);
  static bool attr_set = false;
  if (!attr_set) {
    AG2_HIP(c, hipFuncSetAttribute((const void*)k_render, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    attr_set = true;
  }
  const int grid = (int)std::min<size_t>(n_img, 256 * 8);
  hipLaunchKernelGGL(k_render, dim3(grid), dim3(kImgThreads), lds, c->stream, d_arena, d_off, d_cnt,
                     (int)n_img, d_out);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
