// k_lenet_x3.hip -- K5 convolutions on the bf16 matrix cores with fp32-accurate split operands.
//
// Same layers as k_lenet.hip's k_lenet_conv (conv1 20@5x5 -> max 2/2 -> conv2 50@5x5 -> max 2/2 of
// caffe/test_1batch2.prototxt; Classifier::PredictBatch, caffe_classifier.cpp:94-127), but every
// fp32 operand is written as the EXACT sum of three bf16 terms (x = h + m + l, each the truncation
// of the running remainder to 8 significant bits) and the products are formed on
// v_mfma_f32_32x32x16_bf16 (32 cycles for 16 k, vs 64 cycles for 2 k on the f32-input form) with
// fp32 accumulation:
//   * conv1: the inputs are u8 pixels, exact in bf16, so  x * w = x*wh + x*wm + x*wl  holds exactly
//     (8 bit x 8 bit products are exact in fp32): 3 MFMAs per 16 k, no approximation at all;
//   * conv2: x * w ~ xh*wh + xh*wm + xm*wh + xm*wm + xh*wl + xl*wh; the three dropped terms are
//     below 2^-23 |x w|, i.e. under the rounding of a single fp32 product: 6 MFMAs per 16 k.
// Only the order of the fp32 additions differs from k_lenet_conv; both are held to the same
// tolerance against the oracle (tests/test_gpu_lenet_detect.py).
//
// Unit of work: a third of an image per 256-thread workgroup, two workgroups per CU (k_lenet_conv_x3b,
// the default) or a whole image per 512-thread workgroup (k_lenet_conv_x3, AG2_LENET_WHOLE=1: A/B).
// The image (band) is staged as bf16 in the HWC order it was rendered in; the pooled conv1 map stays
// in LDS, stored already split (one X3Term per bf16 term) and channel-interleaved, so a lane's A
// fragment needs no VALU work:
//   pa[group][y * 40 + x][8]  channels 0-7 / 8-15: 8 k-values = ONE ds_read_b128
//   pc[y * 48 + x][4]         channels 16-19:      4 k-values = one ds_read_b64
// The row pitches (40 = 8 mod 16 slots of 16 B, 48 = 16 mod 32 slots of 8 B) put the two pixel
// rows of a tile's pooling windows on disjoint LDS banks.  K order of conv1: 5 kernel rows x (15
// consecutive (kx, channel) values + one zero-weight tap).  K order of conv2: 25 taps x channels
// 0-15 (lanes 0-31 take channels 0-7, lanes 32-63 channels 8-15), then 7 blocks that cover four
// taps each for channels 16-19 (half h of the wave takes taps 4i + 2h and 4i + 2h + 1).  Weights are
// pre-split and pre-packed on the host in B-fragment order.
#include <string.h>

#include <type_traits>

#include "ag2_internal.h"

namespace ag2 {

// ablation switches for tools/ab_build.sh (timing only, wrong results): 1 no conv1, 2 no conv2, 4 / 8 no A / B requests in conv2
// Wave priority during conv1 of the banded kernel (s_setprio; 0 = leave it alone, for A/B).  The two workgroups of a
// CU share every SIMD; conv2 of one saturates the matrix pipe by itself (round 4: conv2 alone is as fast with one
// workgroup per CU as with two), so the other's conv1 -- few MFMAs, much vector work -- is the phase that should
// win the arbitration: 0.1915 -> 0.188 ms at 934 images (same bits: scheduling only).
#ifndef AG2_CONV1_PRIO
#define AG2_CONV1_PRIO 2
#endif
#ifndef AG2_EXP_ABL
#define AG2_EXP_ABL 0
#endif
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kXThreads = 512;
constexpr int kXWaves = kXThreads / 64;
constexpr int kXImgRow = 180;        // bf16 elements per staged image row (60 pixels x 3 channels, HWC)
constexpr int kXPA = 40;             // row pitch of pa
constexpr int kXPC = 48;             // row pitch of pc
constexpr int kXC1Blocks = 5;        // one ky row per block: its 5 x 3 (kx, channel) values are 15 consecutive HWC elements (+1 zero tap)
constexpr int kXC2Main = 25;
constexpr int kXC2Blocks = kXC2Main + 7;

// one bf16 term of the pooled conv1 map (ROWS rows of it); the three terms of a value sit one
// sizeof(X3Term) apart, in pa and in pc alike, so a term is an immediate offset of a store
template <int ROWS>
struct X3Term {
  unsigned short pa[2][ROWS * kXPA][8];
  unsigned short pc[ROWS * kXPC][4];
};

struct X3Shared {
  X3Term<28> t[3];
  unsigned short imgb[60 * kXImgRow + 16];  // the image as bf16 (u8 values are exact), HWC as handed over
};
static_assert(sizeof(X3Shared) <= 160 * 1024, "k_lenet_conv_x3: LDS");

__host__ __device__ __forceinline__ void split3(float v, unsigned short t[3]) {
  unsigned b;
  __builtin_memcpy(&b, &v, 4);
  const unsigned hb = b & 0xFFFF0000u;
  float h;
  __builtin_memcpy(&h, &hb, 4);
  const float r1 = v - h;  // exact
  unsigned b1;
  __builtin_memcpy(&b1, &r1, 4);
  const unsigned mb = b1 & 0xFFFF0000u;
  float m;
  __builtin_memcpy(&m, &mb, 4);
  const float r2 = r1 - m;  // exact
  unsigned b2;
  __builtin_memcpy(&b2, &r2, 4);
  t[0] = (unsigned short)(hb >> 16);
  t[1] = (unsigned short)(mb >> 16);
  t[2] = (unsigned short)(b2 >> 16);
}

// max of four accumulator values (no NaNs by construction: the canonicalising form fmaxf compiles to
// costs two more instructions per call)
__device__ __forceinline__ float x3_max4(float a, float b, float c, float d) {
  float m;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(d));
  return m;
}

// requests in front of it stay in front, MFMAs behind it stay behind: a compiler-level memory fence
// (the loads are plain reads the instruction selector may otherwise place anywhere) plus a
// scheduling barrier for the machine scheduler
__device__ __forceinline__ void x3_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ bf16x8 as_frag(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }

// four u8 pixels values (one dword of the HWC image) -> four bf16 at elements 4 i .. 4 i + 3
__device__ __forceinline__ void x3_stage4(unsigned short* imgb, int i, unsigned v) {
  const unsigned f0 = __float_as_uint((float)(v & 255u)), f1 = __float_as_uint((float)((v >> 8) & 255u)),
                 f2 = __float_as_uint((float)((v >> 16) & 255u)), f3 = __float_as_uint((float)(v >> 24));
  reinterpret_cast<uint2*>(imgb)[i] = make_uint2((f1 & 0xFFFF0000u) | (f0 >> 16), (f3 & 0xFFFF0000u) | (f2 >> 16));
}

// conv2 + bias + max-pool for the NT tiles mgrp, mgrp + 4, ... of this wave, one channel half
// SH: the shared-memory layout (whole map or one band of it); TS: tile stride between a wave's tiles
template <class SH, int NT, int TS>
__device__ __forceinline__ void x3_conv2(const SH& S, const uint4* __restrict__ w2x,
                                         float* __restrict__ dst, float bias2, int nh, int mgrp,
                                         int lane) {
  const int h = lane >> 5, r = lane & 31;
  const int g = r >> 2, q = r & 3;
  v16f acc[NT];
  int pa0[NT], pc0[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    acc[t] = (v16f){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int w = 8 * (mgrp + TS * t) + g;
    const int wy = w / 12, wx = w - wy * 12;
    const int y = 2 * wy + (q >> 1), x = 2 * wx + (q & 1);
    pa0[t] = y * kXPA + x;
    pc0[t] = y * kXPC + x;
  }
  const uint4* wl = w2x + (size_t)nh * 3 * 64 + lane;
  // Operand fragments are requested ahead of the MFMAs that use them: the B fragments (weights, from
  // L2: several hundred cycles) three blocks ahead into a ring of four register sets, the A fragments
  // (LDS) one block ahead into two.  One block's 6 NT MFMAs are 576 cycles of a wave that has the
  // matrix pipe to itself -- less than an L2 round trip.  The fences keep the requests in front of
  // the MFMAs: left alone the compiler gives consecutive blocks the same registers and so sinks
  // every request behind the last use of the current block, i.e. in front of its own first use.
  // (The whole-image kernel's 4 or 5 tiles per wave leave no room for two sets of A fragments:
  // there only the B fragments run ahead.)
  constexpr bool kAhead = NT <= 3;
  constexpr int kLast = kXC2Blocks - 1;
  uint4 A[2][NT][3], B[4][3];
  auto load_b = [&](int b, uint4(&d)[3]) {
    const uint4* wn = wl + (size_t)b * (2 * 3 * 64);
    d[0] = wn[0];
    d[1] = wn[64];
    d[2] = wn[128];
  };
  auto load_a_main = [&](int b, uint4(&d)[NT][3]) {  // one tap, channels 8 h .. 8 h + 7
    const int ky = b / 5, kx = b - 5 * ky;
    const int off = ky * kXPA + kx;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int s = 0; s < 3; s++) d[t][s] = *reinterpret_cast<const uint4*>(&S.t[s].pa[h][pa0[t] + off][0]);
  };
  // taps 4i + 2h and 4i + 2h + 1, channels 16-19; a tap past the 25th carries zero weights and re-reads tap 24
  auto load_a_tail = [&](int b, uint4(&d)[NT][3]) {
    const int i = b - kXC2Main;
    const int ta = min(4 * i + 2 * h, 24), tb = min(4 * i + 2 * h + 1, 24);
    const int offa = (ta / 5) * kXPC + ta % 5, offb = (tb / 5) * kXPC + tb % 5;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const uint2 lo = *reinterpret_cast<const uint2*>(&S.t[s].pc[pc0[t] + offa][0]);
        const uint2 hi = *reinterpret_cast<const uint2*>(&S.t[s].pc[pc0[t] + offb][0]);
        d[t][s] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
  };
  // per accumulator the six terms in the order hl, lh, mm, hm, mh, hh (smallest first); the tiles
  // interleaved so that consecutive MFMAs are independent
  auto mma = [&](const uint4(&a)[NT][3], const uint4(&bb)[3]) {
    constexpr int ia[6] = {0, 2, 1, 0, 1, 0}, ib[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
      for (int t = 0; t < NT; t++)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[t][ia[k]]), as_frag(bb[ib[k]]), acc[t], 0, 0, 0);
  };
  // block b = 4 i + K: K picks the register sets at compile time; TC / TN: block b / b + 1 is one of
  // the seven channel-16-19 blocks
  auto step = [&](int b, auto K_, auto TC_, auto TN_) {
    constexpr int K = decltype(K_)::value;
    constexpr bool TC = decltype(TC_)::value, TN = decltype(TN_)::value;
#if !(AG2_EXP_ABL & 8)
    load_b(min(b + 3, kLast), B[(K + 3) & 3]);  // (the last requests are repeats nobody uses)
#endif
#if !(AG2_EXP_ABL & 4)
    if constexpr (kAhead) {
      if constexpr (TN) load_a_tail(min(b + 1, kLast), A[(K + 1) & 1]);
      else load_a_main(b + 1, A[(K + 1) & 1]);
    } else {
      if constexpr (TC) load_a_tail(b, A[0]);
      else load_a_main(b, A[0]);
    }
#endif
    if constexpr (!kAhead) x3_fence();
    mma(A[kAhead ? (K & 1) : 0], B[K & 3]);
    if constexpr (kAhead) {
      // issue order inside the step: one request behind each of the first MFMAs (an MFMA keeps the
      // pipe busy for 32 cycles; a request issued in its shadow costs nothing, a block of twelve
      // requests in front of the MFMAs drains the pipe)
#pragma unroll
      for (int i = 0; i < 3 * NT; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);   // LDS reads of the next block's A
      }
#pragma unroll
      for (int i = 0; i < 3; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            // one B fragment of block b + 3
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 6 * NT - 3 * NT - 3, 0);
    }
    x3_fence();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  constexpr std::false_type M{};
  constexpr std::true_type T{};
  static_assert(kXC2Main == 25 && kXC2Blocks == 32, "x3_conv2: the block schedule below");
  load_b(0, B[0]);
  load_b(1, B[1]);
  load_b(2, B[2]);
  if constexpr (kAhead) load_a_main(0, A[0]);
#pragma unroll 1
  for (int b = 0; b < 24; b += 4) {
    step(b, I0{}, M, M);
    step(b + 1, I1{}, M, M);
    step(b + 2, I2{}, M, M);
    step(b + 3, I3{}, M, M);
  }
  step(24, I0{}, M, T);
  step(25, I1{}, T, T);
  step(26, I2{}, T, T);
  step(27, I3{}, T, T);
  step(28, I0{}, T, T);
  step(29, I1{}, T, T);
  step(30, I2{}, T, T);
  step(31, I3{}, T, T);
  const int oc = nh * 32 + r;
  if (oc < 50) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float m = x3_max4(acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]);
        const int wdw = 8 * (mgrp + TS * t) + 2 * j + h;
        dst[wdw * 50 + oc] = m + bias2;  // K' order of ip1: window-major, channel-minor
      }
    }
  }
}

// conv1 + bias + max-pool for NT tiles (T0, T0 + TS, ...): 8 windows x 32 channels each; the pooled
// values are split into three bf16 terms on the way into LDS.
// TS: tile stride between a wave's tiles (= waves per workgroup).  imgb: the staged image (or band of
// it) as bf16 in the HWC order it was rendered in, so the 15 (kx, channel) values a window takes from
// one image row are CONSECUTIVE: K = 5 rows x 16 (one zero-weight tap each) = 5 k-blocks, where a
// planar image needs 8 (two (channel, ky) rows of 5 + 3 zero taps per block).
template <class SH, int NT, int TS>
__device__ __forceinline__ void x3_conv1(SH& S, const uint4 (&W)[kXC1Blocks][3], float bias1,
                                         int T0, int lane, const unsigned short* imgb) {
  const int h = lane >> 5, r = lane & 31;
  const int g = r >> 2, q = r & 3;
  const int T0u = __builtin_amdgcn_readfirstlane(T0);  // tiles are per wave: their arithmetic is scalar
  v16f acc[NT];
  // Eight consecutive bf16 values = five aligned dwords and a funnel shift by 0 or 16 bits; the 16th
  // value of the row (half 1, element 7) meets a zero weight: any finite pixel will do.  Two dword
  // pointers per tile (image rows 0-1 and 2-4 of the window) keep every read an immediate offset.
  const unsigned* pw[NT][2];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    acc[t] = (v16f){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int w = 8 * (T0u + t * TS) + g;
    const int wy = w / 28, wx = w - wy * 28;
    // first element of this lane's half of the window's top row; its parity is that of q (the row
    // pitch and 8 h are even, 3 * (2 wx + (q & 1)) has the parity of q): a lane-constant shift
    const int a0 = (2 * wy + (q >> 1)) * kXImgRow + 3 * (2 * wx + (q & 1)) + 8 * h;
    pw[t][0] = reinterpret_cast<const unsigned*>(imgb) + (a0 >> 1);
    pw[t][1] = pw[t][0] + 2 * (kXImgRow / 2);
  }
  const unsigned sh = (unsigned)(q & 1) * 16u;
  // Issue order of a block: the funnel shifts of tile t directly in front of its first MFMA (only
  // tile 0's are not in the shadow of an MFMA), then the other 2 NT MFMAs with the requests for block
  // b + 1's dwords behind them, one or two per MFMA.
  unsigned raw[2][NT][5];
  auto request = [&](int b, unsigned(&d)[NT][5]) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const unsigned* p = (b < 2 ? pw[t][0] : pw[t][1]) + (b < 2 ? b : b - 2) * (kXImgRow / 2);
#pragma unroll
      for (int k = 0; k < 5; k++) d[t][k] = p[k];
    }
  };
  request(0, raw[0]);
#pragma unroll
  for (int b = 0; b < kXC1Blocks; b++) {
    bf16x8 Af[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const unsigned(&d)[5] = raw[b & 1][t];
      uint4 au;
      au.x = __builtin_amdgcn_alignbit(d[1], d[0], sh);
      au.y = __builtin_amdgcn_alignbit(d[2], d[1], sh);
      au.z = __builtin_amdgcn_alignbit(d[3], d[2], sh);
      au.w = __builtin_amdgcn_alignbit(d[4], d[3], sh);
      Af[t] = as_frag(au);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[t], as_frag(W[b][2]), acc[t], 0, 0, 0);  // low term
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    x3_fence();
    if (b + 1 < kXC1Blocks) request(b + 1, raw[(b + 1) & 1]);
#pragma unroll
    for (int k = 1; k >= 0; k--)  // middle, high term of the weights
#pragma unroll
      for (int t = 0; t < NT; t++)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[t], as_frag(W[b][k]), acc[t], 0, 0, 0);
    if (b + 1 < kXC1Blocks) {
#pragma unroll
      for (int i = 0; i < 2 * NT; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
    x3_fence();
  }
  if (r < 20) {
    // Channels 0-15 go to pa[r >> 3][pos][r & 7], 16-19 to pc[pos][r & 3] of each term: one address
    // expression with lane-constant base, row pitch and position stride.  The eight positions of a
    // tile are consecutive on the 28-wide map and tiles start at multiples of 8, so the row, the
    // column and whether the tile's last four positions wrap into the next row are scalars.
    const bool main = r < 16;
    const int spos = main ? 8 : 4;                   // in bf16 units
    const int srow = main ? kXPA * 8 : kXPC * 4;
    unsigned short* base = (main ? &S.t[0].pa[r >> 3][0][r & 7] : &S.t[0].pc[0][r & 3]) + h * spos;
    constexpr int sterm = (int)(sizeof(S.t[0]) / 2);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int p0 = 8 * (T0u + t * TS);             // first pooled position of the tile
      const int y0 = p0 / 28, x0 = p0 - 28 * y0;
      unsigned short* oa = base + y0 * srow + x0 * spos;
      unsigned short* ob = oa + (srow - 28 * spos);  // the same column count, one row on
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float m = x3_max4(acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]) + bias1;
        unsigned short s3[3];
        split3(m, s3);
        unsigned short* o = (x0 + 2 * j >= 28 ? ob : oa) + 2 * j * spos;  // position p0 + 2 j + h
#pragma unroll
        for (int s = 0; s < 3; s++) o[s * sterm] = s3[s];
      }
    }
  }
}

// the conv1 weights of this lane: 5 k-blocks x 3 terms, kept in registers for a whole unit of work
__device__ __forceinline__ void x3_conv1_weights(const uint4* __restrict__ w1x, int lane, uint4 (&W)[kXC1Blocks][3]) {
#pragma unroll
  for (int b = 0; b < kXC1Blocks; b++)
#pragma unroll
    for (int k = 0; k < 3; k++) W[b][k] = w1x[(b * 3 + k) * 64 + lane];
}

__global__ void __launch_bounds__(kXThreads, 2)
k_lenet_conv_x3(const unsigned char* __restrict__ images, int n_img, const unsigned* __restrict__ d_n,
                const uint4* __restrict__ w1x,
                const float* __restrict__ b1, const uint4* __restrict__ w2x,
                const float* __restrict__ b2, float* __restrict__ pooled2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  X3Shared& S = *reinterpret_cast<X3Shared*>(smem_raw);
  if (d_n) n_img = min(n_img, (int)*d_n);  // frame mode: the list length is read on the device
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31;
  const float bias1 = b1[r];
  const int nh = wid & 1;     // conv2: which 32 output channels
  const int mgrp = wid >> 1;  // conv2: tiles mgrp, mgrp + 4, ...
  const float bias2 = b2[nh * 32 + r];

  for (int im = blockIdx.x; im < n_img; im += gridDim.x) {
    __syncthreads();  // previous image's conv2 readers of the pooled map are done
    {  // stage the image: u8 -> bf16 (exact: the high half of the float), order unchanged
      const unsigned* src = reinterpret_cast<const unsigned*>(images + (size_t)im * 10800);
      for (int i = tid; i < 2700; i += kXThreads) x3_stage4(S.imgb, i, src[i]);
      if (tid < 8) reinterpret_cast<unsigned*>(S.imgb)[5400 + tid] = 0u;
    }
    __syncthreads();
    // conv1: 98 tiles; wave w takes tiles w, w + 8, ... (12 each, waves 0 and 1 a 13th)
    {
      int ln = lane;  // opaque once per image, see k_lenet_conv_x3b
      asm volatile("" : "+v"(ln));
      uint4 W[kXC1Blocks][3];
      x3_conv1_weights(w1x, ln, W);
#pragma unroll 1
      for (int k0 = 0; k0 < 12; k0 += 4) x3_conv1<X3Shared, 4, kXWaves>(S, W, bias1, wid + 8 * k0, ln, S.imgb);
      if (wid < 2) x3_conv1<X3Shared, 1, kXWaves>(S, W, bias1, wid + 96, ln, S.imgb);
    }
    __syncthreads();
    // conv2: 18 tiles x 2 channel halves over 8 waves
    {
      float* dst = pooled2 + (size_t)im * 7200;
      int ln = lane;
      asm volatile("" : "+v"(ln));
      if (mgrp < 2) x3_conv2<X3Shared, 5, 4>(S, w2x, dst, bias2, nh, mgrp, ln);   // tiles mgrp, +4, .., +16
      else x3_conv2<X3Shared, 4, 4>(S, w2x, dst, bias2, nh, mgrp, ln);            // tiles mgrp, +4, .., +12
    }
  }
}

// ---- the same layers, one BAND of an image per workgroup ------------------------------------------
// k_lenet_conv_x3 keeps the whole three-term pooled conv1 map of an image in LDS (148 KB): one
// workgroup per CU, whose staging, conv1 (issue-bound: the u8 -> bf16 A path), the two barriers and
// the pooling / store epilogue all leave the matrix pipes idle (MFMA busy 0.54).  Here the unit of work
// is a third of an image: the conv2 output rows 8k .. 8k+7 (pooled-output rows 4k .. 4k+3, 48 of the
// 144 windows) need the pooled conv1 rows 8k .. 8k+11, i.e. image rows 16k .. 16k+27.  A band's map is
// 12 rows (70 KB with the band of the image), so TWO 256-thread workgroups share a CU and one's
// staging / conv1 / epilogue overlaps the other's conv2; 3 x n units instead of n also cut the last,
// partly empty round of workgroups to a third.  Cost: the four pooled rows two neighbouring bands share
// are computed twice (conv1 + 29 %, 5 % of all MFMA work).  The band of the image is staged as bf16 (a
// u8 value is exact), so a conv1 A fragment is aligned dword reads and funnel shifts instead of byte
// reads, conversions and the packing.  Every output is the same chain of MFMAs in
// the same k order as in the whole-image kernel: bit-identical results.  Measured (934 images): whole
// image 0.276 ms, bands 0.245 ms at the time; 0.194 ms with the operand requests in the shadow of the
// MFMAs (x3_conv2 / x3_conv1) -- ablations of the current kernel (AG2_EXP_ABL): conv2 alone 0.132 ms
// (0.114 with no operand requests at all = the matrix pipe at the ~1.9 GHz this kernel sustains), conv1
// alone 0.058 ms (pipe: 0.029), neither 0.008 ms.
constexpr int kBThreads = 256;
constexpr int kBWaves = kBThreads / 64;
constexpr int kBRows = 12;                   // pooled conv1 rows of a band
constexpr int kBImgRows = 28;                // image rows of a band

struct X3Band {
  X3Term<kBRows> t[3];
  unsigned short imgb[kBImgRows * kXImgRow + 16];  // the band of the image as bf16 (u8 values are exact), HWC
};
static_assert(2 * sizeof(X3Band) <= 160 * 1024, "k_lenet_conv_x3b: two workgroups per CU");

__global__ void __launch_bounds__(kBThreads, 2)
k_lenet_conv_x3b(const unsigned char* __restrict__ images, int n_img, const unsigned* __restrict__ d_n,
                 const uint4* __restrict__ w1x, const float* __restrict__ b1,
                 const uint4* __restrict__ w2x, const float* __restrict__ b2,
                 float* __restrict__ pooled2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  X3Band& S = *reinterpret_cast<X3Band*>(smem_raw);
  if (d_n) n_img = min(n_img, (int)*d_n);  // frame mode: the list length is read on the device
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31;
  const float bias1 = b1[r];
  const int nh = wid & 1;     // conv2: which 32 output channels
  const int mgrp = wid >> 1;  // conv2: tiles mgrp, mgrp + 2, mgrp + 4
  const float bias2 = b2[nh * 32 + r];
  const int units = 3 * n_img;
  // The band of the NEXT unit is requested from global memory while this unit's conv1 runs and kept in
  // registers (5 dwords per thread) through conv2; it goes into LDS at the top of the next round, when
  // conv1 -- the only reader of the staged band -- is long done.  No global round trip and one barrier
  // less on a unit's critical path.
  constexpr int kBandDw = kBImgRows * 45;                       // 28 rows x 180 bytes
  constexpr int kBandPer = (kBandDw + kBThreads - 1) / kBThreads;
  unsigned nxt[kBandPer];
  auto fetch = [&](int u) {
    const int im = u / 3, band = u - 3 * im;
    const unsigned* src = reinterpret_cast<const unsigned*>(images + (size_t)im * 10800 + band * (16 * 180));
#pragma unroll
    for (int k = 0; k < kBandPer; k++) {
      const int i = tid + k * kBThreads;
      nxt[k] = src[min(i, kBandDw - 1)];
    }
  };
  if ((int)blockIdx.x < units) fetch(blockIdx.x);
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int im = u / 3, band = u - 3 * im;
    {  // image rows 16 band .. 16 band + 27 as bf16, order unchanged
#pragma unroll
      for (int k = 0; k < kBandPer; k++) {
        const int i = tid + k * kBThreads;
        if (i < kBandDw) x3_stage4(S.imgb, i, nxt[k]);
      }
      if (tid < 8) reinterpret_cast<unsigned*>(S.imgb)[kBImgRows * 90 + tid] = 0u;
    }
    __syncthreads();  // the band is staged; the previous unit's conv2 readers of the pooled map are done
    // conv1: 42 tiles; wave w takes tiles w, w + 4, ... (10 each, waves 0 and 1 an 11th)
    {
      // (the lane index is made opaque once per unit: otherwise every address of the unrolled tile
      // code is hoisted out of the unit loop as a loop invariant and spilled)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      uint4 W[kXC1Blocks][3];
      x3_conv1_weights(w1x, ln, W);
      if (u + (int)gridDim.x < units) fetch(u + gridDim.x);  // (after the weights: waiting for those does not wait for these)
#if !(AG2_EXP_ABL & 1)
#if AG2_CONV1_PRIO
      __builtin_amdgcn_s_setprio(AG2_CONV1_PRIO);
#endif
      x3_conv1<X3Band, 4, kBWaves>(S, W, bias1, wid, ln, S.imgb);
      x3_conv1<X3Band, 4, kBWaves>(S, W, bias1, wid + 16, ln, S.imgb);
      // (tiles wid + 32, + 36 and, for waves 0 and 1, + 40: three tiles at once rather than a single
      // one with its chain of dependent MFMAs)
      if (wid < 2) x3_conv1<X3Band, 3, kBWaves>(S, W, bias1, wid + 32, ln, S.imgb);
      else x3_conv1<X3Band, 2, kBWaves>(S, W, bias1, wid + 32, ln, S.imgb);
#endif
    }
    __syncthreads();
    // conv2: 6 tiles x 2 channel halves over 4 waves; the band's 48 windows follow the 48 band
    // windows before them in the K' order of ip1
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
#if AG2_CONV1_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
#if !(AG2_EXP_ABL & 2)
      x3_conv2<X3Band, 3, 2>(S, w2x, pooled2 + (size_t)im * 7200 + band * (48 * 50), bias2, nh, mgrp, ln);
#endif
    }
  }
}

// host: split and pack the conv weights in B-fragment order
int lenet_pack_weights_x3(ag2_ctx* c, const float* c1w, const float* c2w) {
  std::vector<unsigned short> w1x((size_t)kXC1Blocks * 3 * 64 * 8, 0), w2x((size_t)kXC2Blocks * 2 * 3 * 64 * 8, 0);
  unsigned short s3[3];
  for (int b = 0; b < kXC1Blocks; b++)  // block b = kernel row ky; k = 8 h + j = 3 kx + channel
    for (int l = 0; l < 64; l++) {
      const int h = l >> 5, oc = l & 31;
      if (oc >= 20) continue;
      for (int j = 0; j < 8; j++) {
        const int e = 8 * h + j;
        if (e >= 15) continue;
        const int kx = e / 3, ch = e % 3;
        split3(c1w[((oc * 3 + ch) * 5 + b) * 5 + kx], s3);
        for (int t = 0; t < 3; t++) w1x[(((size_t)b * 3 + t) * 64 + l) * 8 + j] = s3[t];
      }
    }
  for (int b = 0; b < kXC2Blocks; b++)
    for (int nh = 0; nh < 2; nh++)
      for (int l = 0; l < 64; l++) {
        const int h = l >> 5, oc = nh * 32 + (l & 31);
        if (oc >= 50) continue;
        for (int j = 0; j < 8; j++) {
          int ch, tap;
          if (b < kXC2Main) {
            ch = 8 * h + j;
            tap = b;
          } else {
            tap = 4 * (b - kXC2Main) + 2 * h + (j >> 2);
            if (tap >= 25) continue;
            ch = 16 + (j & 3);
          }
          const int ky = tap / 5, kx = tap % 5;
          split3(c2w[((oc * 20 + ch) * 5 + ky) * 5 + kx], s3);
          for (int t = 0; t < 3; t++) w2x[((((size_t)b * 2 + nh) * 3 + t) * 64 + l) * 8 + j] = s3[t];
        }
      }
  LeNetDev& d = c->net;
  AG2_HIP(c, d.w1x.reserve(w1x.size() * 2));
  AG2_HIP(c, d.w2x.reserve(w2x.size() * 2));
  AG2_HIP(c, hipMemcpyAsync(d.w1x.p, w1x.data(), w1x.size() * 2, hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, hipMemcpyAsync(d.w2x.p, w2x.data(), w2x.size() * 2, hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---- ip1 on the bf16 matrix cores, same three-term split ------------------------------------------
// One 128-image x 128-output tile per workgroup over a K range (split-K; k_lenet_fc_finish adds the
// partial sums in split order).  The fp32 activations are split into three bf16 terms while they are
// staged into LDS (row pitch 40 bf16 = 80 bytes: the eight rows of one quarter of a b128 read fall on
// disjoint banks), the weights are pre-split and pre-packed in B-fragment order.  Twice the tile
// height of k_lenet_fc1: the weight stream from L2 is read half as often.
// The stage is DOUBLE-BUFFERED in chunks of 32 k: while the MFMAs of chunk c read one buffer, the same
// waves split chunk c + 1 (in registers since the chunk before) into the other and request chunk
// c + 2 -- the conversion, the LDS writes and every request are issued in the shadow of the MFMAs,
// and a chunk costs one barrier.  (Staging, then MFMAs, two barriers per chunk: matrix pipes 49 %
// busy.)
// (AG2_EXP_FCABL: timing-only ablations of k_lenet_fc1_x3 -- wrong results -- for tools/ab_build.sh: 1 half of the
//  A-fragment LDS reads, 2 no B-fragment loads in the loop, 4 no conversion / LDS writes, 8 no activation
//  loads in the loop, 16 no MFMAs)
#ifndef AG2_EXP_FCABL
#define AG2_EXP_FCABL 0
#endif
constexpr int kFxBM = 128;
constexpr int kFxKC = 32;              // k per chunk: 7200 = 225 chunks of two 16-k blocks
constexpr int kFxPitch = 40;           // bf16 per staged row
constexpr int kFxN = 512;
static_assert(kFc1X3Chunks * kFxKC == 7200, "ip1: K");

struct FxShared {
  unsigned short a[2][3][kFxBM][kFxPitch];  // [buffer][term]
};
static_assert(2 * sizeof(FxShared) <= 160 * 1024, "k_lenet_fc1_x3: two workgroups per CU");

__global__ void __launch_bounds__(256, 2)
k_lenet_fc1_x3(const float* __restrict__ x, int n_img, const unsigned* __restrict__ d_n, int n_pad,
               const uint4* __restrict__ w3x, int chunks_per_split, float* __restrict__ part) {
  __shared__ FxShared S;
  if (d_n) {
    // frame mode: a launch as large as any batch up to the list's capacity can need; the batch size
    // is read here and the split chosen by the SAME rule the host applies to a known batch size, so
    // the partial sums (and with them every logit) are bit-identical to those of an exact-size launch
    n_img = min(n_img, (int)*d_n);
    chunks_per_split = kFc1X3Chunks / fc1_x3_ksplit((n_img + kFxBM - 1) / kFxBM);
  }
  // Work item -> (image tile bx, column group by, K split bz), K split slowest.  Workgroups go to the
  // 8 XCDs round-robin by their index and every XCD has its own L2, so the items are dealt such that
  // one XCD gets a CONTIGUOUS eighth of them, i.e. one or two K splits: it then fetches an eighth of
  // the weights (2.8 of 22 MB: they stay in its L2 for all image tiles) instead of all of them.
  const int mtiles = (n_img + kFxBM - 1) / kFxBM;
  const int items = mtiles * 4 * (kFc1X3Chunks / chunks_per_split);
  const int per_xcd = (items + 7) >> 3;
  const int slot = blockIdx.x >> 3;
  const int item = (int)(blockIdx.x & 7) * per_xcd + slot;
  if (slot >= per_xcd || item >= items) return;  // uniform
  const int bx = item % mtiles, by = (item / mtiles) & 3, bz = item / (mtiles * 4);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int img0 = bx * kFxBM;
  const int nt = by * 4 + wid;  // 32-column tile of this wave
  const int chunk0 = bz * chunks_per_split;
  v16f acc[4];
#pragma unroll
  for (int t = 0; t < 4; t++) acc[t] = (v16f){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // staging: thread t owns image row t / 2 and 16 consecutive k of the chunk
  const int srow = tid >> 1, scol = (tid & 1) * 16;
  const bool srow_ok = (img0 + srow) < n_img;
  const float* sx = x + (size_t)(img0 + srow) * 7200 + scol + (size_t)chunk0 * kFxKC;
  auto gload = [&](int ci, float4(&v)[4]) {
    const int cc = min(ci, chunks_per_split - 1);  // (requests past the split repeat its last chunk)
#pragma unroll
    for (int i = 0; i < 4; i++)
      v[i] = srow_ok ? *reinterpret_cast<const float4*>(sx + (size_t)cc * kFxKC + 4 * i) : make_float4(0, 0, 0, 0);
  };
  // one float4 (four k) of the thread's row: three terms, 8 bytes each
  auto lstore4 = [&](int buf, int i, const float4& v) {
    unsigned short t3[4][3];
    split3(v.x, t3[0]);
    split3(v.y, t3[1]);
    split3(v.z, t3[2]);
    split3(v.w, t3[3]);
#pragma unroll
    for (int s = 0; s < 3; s++) {
      uint2 u;
      u.x = (unsigned)t3[0][s] | ((unsigned)t3[1][s] << 16);
      u.y = (unsigned)t3[2][s] | ((unsigned)t3[3][s] << 16);
      *reinterpret_cast<uint2*>(&S.a[buf][s][srow][scol + 4 * i]) = u;
    }
  };
  // B fragments: k-block kb of this split, ring of four sets, requested two k-blocks ahead
  const uint4* wl = w3x + ((size_t)(chunk0 * 2) * 16 + nt) * 3 * 64 + lane;  // k-block stride: 16*3*64
  const int nkb = chunks_per_split * 2;
  uint4 B[4][3];
  auto load_b = [&](int kb, uint4(&d)[3]) {
#if AG2_EXP_FCABL & 2
    if (kb > 1) return;
#endif
    const uint4* wn = wl + (size_t)min(kb, nkb - 1) * (16 * 3 * 64);
    d[0] = wn[0];
    d[1] = wn[64];
    d[2] = wn[128];
  };
  // A fragments of two image tiles (t0, t0 + 1) for k-block kb of the chunk in buffer buf
  auto load_a = [&](int buf, int kbl, int t0, uint4(&d)[2][3]) {
#pragma unroll
    for (int t = 0; t < ((AG2_EXP_FCABL & 1) ? 1 : 2); t++)
#pragma unroll
      for (int s = 0; s < 3; s++)
        d[t][s] = *reinterpret_cast<const uint4*>(&S.a[buf][s][32 * (t0 + t) + r][16 * kbl + 8 * h]);
#if AG2_EXP_FCABL & 1
    for (int s = 0; s < 3; s++) d[1][s] = d[0][s];
#endif
  };
  auto mma2 = [&](int t0, const uint4(&a)[2][3], const uint4(&bb)[3]) {
    constexpr int ia[6] = {0, 2, 1, 0, 1, 0}, ib[6] = {2, 0, 1, 1, 0, 0};  // hl, lh, mm, hm, mh, hh
#if AG2_EXP_FCABL & 16
    acc[t0][0] += __uint_as_float(a[0][0].x ^ a[1][1].y ^ a[0][2].z ^ a[1][0].w ^ bb[0].x ^ bb[1].y ^ bb[2].z);
    return;
#endif
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
      for (int t = 0; t < 2; t++)
        acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[t][ia[k]]), as_frag(bb[ib[k]]), acc[t0 + t], 0, 0, 0);
  };
  float4 RA[4], RB[4];
  gload(0, RA);
#pragma unroll
  for (int i = 0; i < 4; i++) lstore4(0, i, RA[i]);
  gload(1, RA);
  gload(2, RB);
  load_b(0, B[0]);
  load_b(1, B[1]);
  uint4 A[2][2][3];
  __syncthreads();
  load_a(0, 0, 0, A[0]);
  // One chunk: four half-steps of 12 MFMAs (k-block 0 tiles 0-1, tiles 2-3, k-block 1 tiles 0-1, tiles 2-3).
  // Behind the MFMAs of a half-step: the A fragments of the next one, one quarter of the next chunk's
  // conversion + LDS writes, and (per k-block) the B fragments two k-blocks on.
  auto chunk = [&](int ci, auto PAR_, float4(&R)[4]) {
    constexpr int PAR = decltype(PAR_)::value;  // ci & 1: buffer of this chunk, B ring phase
    const int buf = PAR, nbuf = PAR ^ 1;
#pragma unroll
    for (int hs = 0; hs < 4; hs++) {
      const int kbl = hs >> 1, t0 = 2 * (hs & 1);
      if (hs == 0) load_b(2 * ci + 2, B[(2 * PAR + 2) & 3]);
      if (hs == 2) load_b(2 * ci + 3, B[(2 * PAR + 3) & 3]);
      // next half-step's A fragments (the first of the next chunk come after the barrier)
      if (hs < 3) load_a(buf, (hs + 1) >> 1, 2 * ((hs + 1) & 1), A[(hs + 1) & 1]);
#if !(AG2_EXP_FCABL & 4)
      lstore4(nbuf, hs, R[hs]);
#endif
      mma2(t0, A[hs & 1], B[(2 * PAR + kbl) & 3]);
#pragma unroll
      for (int i = 0; i < 12; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // an A-fragment read of the next half-step
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // conversion of the next chunk
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // its LDS writes
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // B fragments two k-blocks on
      }
      x3_fence();
    }
#if !(AG2_EXP_FCABL & 8)
    gload(ci + 3, R);  // (in flight for a chunk and a half)
#endif
    __syncthreads();   // chunk ci + 1 is staged; every reader of this chunk's buffer is done
    load_a(nbuf, 0, 0, A[0]);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  for (int ci = 0; ci < chunks_per_split; ci += 2) {
    chunk(ci, P0{}, RA);
    if (ci + 1 < chunks_per_split) chunk(ci + 1, P1{}, RB);
  }
  float* dst = part + ((size_t)bz * n_pad + img0) * kFxN + nt * 32 + r;
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int row = 32 * t + (q & 3) + 8 * (q >> 2) + 4 * h;
      dst[(size_t)row * kFxN] = acc[t][q];
    }
}

// w3p: the [7200][512] K'-ordered ip1 matrix of lenet_pack_weights (k_lenet.hip), host memory
int lenet_pack_fc_x3(ag2_ctx* c, const float* w3p) {
  std::vector<unsigned short> w3x((size_t)450 * 16 * 3 * 64 * 8);
  unsigned short s3[3];
  for (int kbk = 0; kbk < 450; kbk++)
    for (int nt = 0; nt < 16; nt++)
      for (int l = 0; l < 64; l++) {
        const int h = l >> 5, col = nt * 32 + (l & 31);
        for (int j = 0; j < 8; j++) {
          split3(w3p[(size_t)(16 * kbk + 8 * h + j) * kFxN + col], s3);
          for (int t = 0; t < 3; t++) w3x[((((size_t)kbk * 16 + nt) * 3 + t) * 64 + l) * 8 + j] = s3[t];
        }
      }
  LeNetDev& d = c->net;
  AG2_HIP(c, d.w3x.reserve(w3x.size() * 2));
  AG2_HIP(c, hipMemcpyAsync(d.w3x.p, w3x.data(), w3x.size() * 2, hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

// partial sums into d_fcpart; *n_pad_out, *ksplit_out describe them for k_lenet_fc_finish.
// d_n (frame mode): n is the capacity, the batch size is read on the device (grid: capacity x the
// finest split, the surplus workgroups leave at once).
int launch_lenet_fc1_x3(ag2_ctx* c, size_t n, int* n_pad_out, int* ksplit_out, const unsigned* d_n) {
  LeNetDev& d = c->net;
  const int mtiles = (int)((n + kFxBM - 1) / kFxBM);
  const int n_pad = mtiles * kFxBM;
  const int ksplit = d_n ? kFc1X3MaxSplit : fc1_x3_ksplit(mtiles);
  AG2_HIP(c, c->d_fcpart.reserve((size_t)ksplit * n_pad * kFxN * 4));
  int items = mtiles * 4 * ksplit;
  if (d_n) {  // the largest number of work items any batch of 1 .. mtiles tiles needs
    items = 0;
    for (int m = 1; m <= mtiles; m++) items = std::max(items, m * 4 * fc1_x3_ksplit(m));
  }
  const dim3 grid((items + 7) / 8 * 8, 1, 1);
  hipLaunchKernelGGL(k_lenet_fc1_x3, grid, dim3(256), 0, c->stream,
                     c->d_act1.as<float>(), (int)n, d_n, n_pad, d.w3x.as<uint4>(), kFc1X3Chunks / ksplit,
                     c->d_fcpart.as<float>());
  AG2_HIP(c, hipGetLastError());
  *n_pad_out = n_pad;
  *ksplit_out = ksplit;
  return 0;
}

int launch_lenet_conv_x3(ag2_ctx* c, const uint8_t* d_images, size_t n, float* d_pooled2,
                         const unsigned* d_n) {
  LeNetDev& d = c->net;
  const size_t lds = sizeof(X3Shared);
  if (!(c->func_attr_done & kAttrLenetX3)) {
    AG2_HIP(c, hipFuncSetAttribute((const void*)k_lenet_conv_x3,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    c->func_attr_done |= kAttrLenetX3;
  }
  if (d.use_bands) {  // (AG2_LENET_WHOLE=1 at weight load: one workgroup per image, for A/B)
    if (!(c->func_attr_done & kAttrLenetX3b)) {
      AG2_HIP(c, hipFuncSetAttribute((const void*)k_lenet_conv_x3b,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(X3Band)));
      c->func_attr_done |= kAttrLenetX3b;
    }
    // (AG2_EXP_CONV_GRID: experiment -- another number of workgroups, e.g. 256 = one per CU)
    static const int exp_grid = [] { const char* e = getenv("AG2_EXP_CONV_GRID"); return e ? atoi(e) : 0; }();
    const int gridb = (int)std::min<size_t>(3 * n, exp_grid > 0 ? (size_t)exp_grid : 512);
    hipLaunchKernelGGL(k_lenet_conv_x3b, dim3(gridb), dim3(kBThreads), sizeof(X3Band), c->stream, d_images,
                       (int)n, d_n, d.w1x.as<uint4>(), d.b1.as<float>(), d.w2x.as<uint4>(), d.b2.as<float>(),
                       d_pooled2);
    AG2_HIP(c, hipGetLastError());
    return 0;
  }
  const int grid = (int)std::min<size_t>(n, 256);
  hipLaunchKernelGGL(k_lenet_conv_x3, dim3(grid), dim3(kXThreads), lds, c->stream, d_images, (int)n,
                     d_n, d.w1x.as<uint4>(), d.b1.as<float>(), d.w2x.as<uint4>(), d.b2.as<float>(),
                     d_pooled2);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
