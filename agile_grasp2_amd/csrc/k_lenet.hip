// k_lenet.hip -- K5: batched LeNet forward on the f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces Classifier::PredictBatch (src/agile_grasp2/caffe_classifier.cpp:94-127; Caffe's
// im2col + SGEMM) for the network of caffe/test_1batch2.prototxt:1-92:
//   3x60x60 u8 (no mean, no scale; PreprocessBatch :158-198) -> conv1 20@5x5 -> max 2/2 ->
//   conv2 50@5x5 -> max 2/2 -> ip1 500 -> ReLU -> ip2 2 (raw logits, blob "ip2", :121).
// f32 in / f32 accumulate MFMA is bit-for-bit a k-ordered fmaf chain, i.e. genuine fp32 -- the
// parity mode.  This is the only GEMM-shaped work on the path.
//
// k_lenet_conv: one 256-thread workgroup per image (persistent).  The image (u8 planar) and the
// pooled conv1 map (20 x 28 x 28 f32) stay in LDS, so conv1 -> pool -> conv2 -> pool never touches
// HBM.  Implicit GEMM with M = output pixels, N = output channels:
//   * the 32 rows of an MFMA tile are 8 pooling windows x their 4 pixels, ordered so that the four
//     pixels of a window land in four consecutive accumulator registers of ONE lane: the 2x2
//     max-pool is three v_max on registers, no shuffles, no LDS;
//   * K is ordered so the two k of a 32x32x2 step differ by a CONSTANT LDS offset (conv1: image
//     rows ky / ky+1, a zero-weight sixth row pads 5 -> 6; conv2: channels c / c+1), hence every
//     A-operand read is one ds_read with an immediate offset off a per-tile base;
//   * weights are pre-packed on the host in exactly the lane order of the B operand.
// k_lenet_fc1: ip1 partial sums, 64-image x 128-output tiles with split K (K = 7200), A staged
// through LDS, B streamed from L2 / Infinity Cache in lane order; k_lenet_fc_finish adds the partial
// sums in split order, applies bias + ReLU and ip2.
// These f32-input kernels are the AG2_LENET_F32=1 path; the default convolutions and ip1 run on the
// bf16 matrix cores with three-term operand splits (k_lenet_x3.hip) and share k_lenet_fc_finish.
#include "ag2_internal.h"

namespace ag2 {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kConvThreads = 256;
constexpr int kP1Stride = 785;            // 28*28 + 1: conflict-free pooled-conv1 stores
constexpr int kImgRow = 60;
constexpr int kImgPlane = 61 * 60;        // one zero row appended per channel (ky = 5 pad tap)
constexpr int kC1Steps = 45;              // 3 channels x 3 ky-pairs x 5 kx
constexpr int kC2Steps = 250;             // 10 channel pairs x 5 ky x 5 kx
constexpr int kFcK = 7200;
constexpr int kFcN = 512;                 // 500 padded
constexpr int kFcBM = 64;
constexpr int kFcKC = 96;                 // K chunk staged in LDS (7200 = 75 * 96)
constexpr int kFcKCP = 97;                // padded row: A reads hit 32 distinct banks

struct ConvShared {
  float p1[20 * kP1Stride];
  unsigned char img[3 * kImgPlane + 12];
};

// conv2 for NT of this wave's M-tiles (tile index mgrp + 2 * (t0 + t)), all 250 k-steps.
template <int NT>
__device__ __forceinline__ void conv2_pass(const float* __restrict__ p1, const float* __restrict__ w2p,
                                           float* __restrict__ dst, float bias2, int nh, int mgrp,
                                           int t0, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int g = l31 >> 2, q = l31 & 3;
  v16f acc[NT];
  int base[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    acc[t] = (v16f){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int w = 8 * (mgrp + 2 * (t0 + t)) + g;
    const int wy = w / 12, wx = w - wy * 12;
    base[t] = half * kP1Stride + (2 * wy + (q >> 1)) * 28 + 2 * wx + (q & 1);
  }
  const float* wp = w2p + nh * 64 + lane;
  for (int cp = 0; cp < 10; cp++) {
    const float* pc = p1 + 2 * cp * kP1Stride;
    const float* wc = wp + (size_t)cp * 25 * 128;
#pragma unroll
    for (int ky = 0; ky < 5; ky++) {
      float bw[5];
#pragma unroll
      for (int kx = 0; kx < 5; kx++) bw[kx] = wc[(ky * 5 + kx) * 128];
#pragma unroll
      for (int kx = 0; kx < 5; kx++) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const float a = pc[base[t] + ky * 28 + kx];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[kx], acc[t], 0, 0, 0);
        }
      }
    }
  }
  const int oc = nh * 32 + l31;
  if (oc < 50) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float m = fmaxf(fmaxf(acc[t][4 * j], acc[t][4 * j + 1]),
                              fmaxf(acc[t][4 * j + 2], acc[t][4 * j + 3]));
        const int wdw = 8 * (mgrp + 2 * (t0 + t)) + 2 * j + half;
        dst[wdw * 50 + oc] = m + bias2;  // K' order of ip1: window-major, channel-minor
      }
    }
  }
}

__global__ void __launch_bounds__(kConvThreads, 2)
k_lenet_conv(const unsigned char* __restrict__ images, int n_img, const float* __restrict__ w1p,
             const float* __restrict__ b1, const float* __restrict__ w2p,
             const float* __restrict__ b2, float* __restrict__ pooled2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ConvShared& S = *reinterpret_cast<ConvShared*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int g = l31 >> 2, q = l31 & 3;  // pooling window within the tile, pixel within the window

  const float bias1 = b1[l31];
  const int nh = wid & 1;       // conv2: which 32 output channels
  const int mgrp = wid >> 1;    // conv2: tiles mgrp, mgrp + 2, ...
  const float bias2 = b2[nh * 32 + l31];

  for (int im = blockIdx.x; im < n_img; im += gridDim.x) {
    __syncthreads();  // previous image's conv2 readers of p1 are done
    // ---- stage the image: HWC u8 -> planar u8 (+ one zero row per channel) -------------------
    {
      const unsigned* src = reinterpret_cast<const unsigned*>(images + (size_t)im * 10800);
      for (int i = tid; i < 2700; i += kConvThreads) {
        const unsigned v = src[i];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const int e = i * 4 + b;  // byte index = pixel * 3 + ch
          const int pix = e / 3, ch = e - pix * 3;
          S.img[ch * kImgPlane + pix] = (unsigned char)((v >> (8 * b)) & 255u);
        }
      }
      if (tid < 180) S.img[(tid / 60) * kImgPlane + 3600 + (tid % 60)] = 0;
    }
    __syncthreads();
    // ---- conv1 + bias + max-pool: 98 tiles of 8 windows x 32 channels -----------------------
    // conv1 weights for this lane: one VGPR per k-step, (re)loaded per image from L2 so they do
    // not stay live across conv2's 144 accumulator registers
    float wreg[kC1Steps];
#pragma unroll
    for (int s = 0; s < kC1Steps; s++) wreg[s] = w1p[s * 64 + lane];
    for (int T = wid; T < 98; T += 4) {
      const int w = 8 * T + g;
      const int wy = w / 28, wx = w - wy * 28;
      const int oy = 2 * wy + (q >> 1), ox = 2 * wx + (q & 1);
      const unsigned char* a0 = &S.img[(oy + half) * kImgRow + ox];
      v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int kp = 0; kp < 3; kp++)
#pragma unroll
          for (int kx = 0; kx < 5; kx++) {
            const float a = (float)a0[c * kImgPlane + 2 * kp * kImgRow + kx];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[(c * 3 + kp) * 5 + kx], acc, 0, 0, 0);
          }
      if (l31 < 20) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float m = fmaxf(fmaxf(acc[4 * j], acc[4 * j + 1]), fmaxf(acc[4 * j + 2], acc[4 * j + 3]));
          S.p1[l31 * kP1Stride + 8 * T + 2 * j + half] = m + bias1;
        }
      }
    }
    __syncthreads();
    // ---- conv2 + bias + max-pool: 18 tiles x 2 channel halves; 9 tiles per wave, done as two
    // passes (5 + 4 tiles) so the accumulators (80 / 64 VGPRs) leave room for two waves per SIMD
    {
      float* dst = pooled2 + (size_t)im * kFcK;
      conv2_pass<5>(S.p1, w2p, dst, bias2, nh, mgrp, 0, lane);
      conv2_pass<4>(S.p1, w2p, dst, bias2, nh, mgrp, 5, lane);
    }
  }
}

struct FcShared {
  float a[2][kFcBM * kFcKCP];
};

// ip1 partial products: one 64-image x 128-output tile per workgroup over a K range (split-K so
// that small batches still fill 256 CUs).  grid = (image tiles, 4 column tiles, K splits).
// part[ks][image][512] receives the partial sums; k_lenet_fc_finish adds them in ks order.
__global__ void __launch_bounds__(256)
k_lenet_fc1(const float* __restrict__ x, int n_img, int n_pad, const float* __restrict__ w3p,
            int chunks_per_split, float* __restrict__ part) {
  __shared__ FcShared S;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int img0 = blockIdx.x * kFcBM;
  const int col0 = blockIdx.y * 128 + wid * 32;
  const int kbeg = blockIdx.z * chunks_per_split * kFcKC;
  v16f acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v16f acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // staging: 64 images x 96 k per chunk; thread t owns image t/4, 24 consecutive floats.  Global
  // loads for chunk i+1 are issued before the MFMAs of chunk i and written to LDS after them.
  const int srow = tid >> 2, scol = (tid & 3) * 24;
  const bool srow_ok = (img0 + srow) < n_img;
  const float* sx = x + (size_t)(img0 + srow) * kFcK + scol + kbeg;
  float4 v[6];
  auto gload = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 6; i++)
      v[i] = srow_ok ? *reinterpret_cast<const float4*>(sx + kc + 4 * i) : make_float4(0, 0, 0, 0);
  };
  auto lstore = [&](int buf) {
    float* d = &S.a[buf][srow * kFcKCP + scol];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      d[4 * i] = v[i].x; d[4 * i + 1] = v[i].y; d[4 * i + 2] = v[i].z; d[4 * i + 3] = v[i].w;
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();
  const float* wb = w3p + (size_t)(kbeg + half) * kFcN + col0 + l31;
  for (int it = 0; it < chunks_per_split; it++) {
    const int buf = it & 1;
    const bool more = (it + 1) < chunks_per_split;
    if (more) gload((it + 1) * kFcKC);
    const float* a0 = &S.a[buf][l31 * kFcKCP + half];
    const float* a1 = a0 + 32 * kFcKCP;
    const float* wk = wb + (size_t)it * kFcKC * kFcN;
#pragma unroll 12
    for (int s = 0; s < kFcKC / 2; s++) {
      const float bv = wk[(size_t)(2 * s) * kFcN];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[2 * s], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[2 * s], bv, acc1, 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }
  float* dst = part + ((size_t)blockIdx.z * n_pad + img0) * kFcN + col0 + l31;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    dst[(size_t)row * kFcN] = acc0[r];
    dst[(size_t)(row + 32) * kFcN] = acc1[r];
  }
}

// ip1 finish (+ bias, ReLU in place: prototxt relu1) and ip2: one wave per image, partial sums
// added in split order, the 512-wide dot products reduced with a fixed shuffle tree.
__global__ void __launch_bounds__(256)
k_lenet_fc_finish(const float* __restrict__ part, int n_img, const unsigned* __restrict__ d_n,
                  int n_pad, int ksplit,
                  const float* __restrict__ b3, const float* __restrict__ w4,
                  const float* __restrict__ b4, float* __restrict__ logits) {
  if (d_n) {  // frame mode: batch size and split as k_lenet_fc1_x3 derived them
    n_img = min(n_img, (int)*d_n);
    ksplit = fc1_x3_ksplit((n_img + 127) / 128);
  }
  const int lane = threadIdx.x & 63;
  const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (img >= n_img) return;
  float s0 = 0.f, s1 = 0.f;
  // the eight outputs of a lane are summed side by side: eight independent loads per split in
  // flight instead of one chain of 8 * ksplit dependent ones (each sum still runs in split order)
  float h[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* pr = part + (size_t)img * kFcN + lane;
  for (int ks = 0; ks < ksplit; ks++) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = pr[j * 64];
#pragma unroll
    for (int j = 0; j < 8; j++) h[j] += v[j];
    pr += (size_t)n_pad * kFcN;
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int n = j * 64 + lane;
    const float r = fmaxf(h[j] + b3[n], 0.f);
    s0 = __builtin_fmaf(r, w4[n], s0);
    s1 = __builtin_fmaf(r, w4[kFcN + n], s1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  if (lane == 0) {
    logits[(size_t)img * 2] = s0 + b4[0];
    logits[(size_t)img * 2 + 1] = s1 + b4[1];
  }
}

int lenet_pack_weights(ag2_ctx* c, const float* c1w, const float* c1b, const float* c2w,
                       const float* c2b, const float* f1w, const float* f1b, const float* f2w,
                       const float* f2b) {
  std::vector<float> w1p((size_t)kC1Steps * 64, 0.f), b1(32, 0.f);
  for (int ch = 0; ch < 3; ch++)
    for (int kp = 0; kp < 3; kp++)
      for (int kx = 0; kx < 5; kx++) {
        const int s = (ch * 3 + kp) * 5 + kx;
        for (int l = 0; l < 64; l++) {
          const int oc = l & 31, ky = 2 * kp + (l >> 5);
          if (oc < 20 && ky < 5) w1p[(size_t)s * 64 + l] = c1w[((oc * 3 + ch) * 5 + ky) * 5 + kx];
        }
      }
  for (int i = 0; i < 20; i++) b1[i] = c1b[i];
  std::vector<float> w2p((size_t)kC2Steps * 128, 0.f), b2(64, 0.f);
  for (int cp = 0; cp < 10; cp++)
    for (int ky = 0; ky < 5; ky++)
      for (int kx = 0; kx < 5; kx++) {
        const int s = (cp * 5 + ky) * 5 + kx;
        for (int nh = 0; nh < 2; nh++)
          for (int l = 0; l < 64; l++) {
            const int oc = nh * 32 + (l & 31), ch = 2 * cp + (l >> 5);
            if (oc < 50) w2p[(size_t)s * 128 + nh * 64 + l] = c2w[((oc * 20 + ch) * 5 + ky) * 5 + kx];
          }
      }
  for (int i = 0; i < 50; i++) b2[i] = c2b[i];
  // ip1: K' = window * 50 + channel  <->  Caffe's flattened CHW index channel * 144 + window
  std::vector<float> w3p((size_t)kFcK * kFcN, 0.f), b3(kFcN, 0.f), w4(2 * kFcN, 0.f);
  for (int wdw = 0; wdw < 144; wdw++)
    for (int oc = 0; oc < 50; oc++) {
      const size_t kp = (size_t)wdw * 50 + oc, k = (size_t)oc * 144 + wdw;
      float* dst = &w3p[kp * kFcN];
      for (int n = 0; n < 500; n++) dst[n] = f1w[(size_t)n * kFcK + k];
    }
  for (int n = 0; n < 500; n++) {
    b3[n] = f1b[n];
    w4[n] = f2w[n];
    w4[kFcN + n] = f2w[500 + n];
  }
  LeNetDev& d = c->net;
  struct Up { DevBuf* b; const float* p; size_t n; };
  const Up ups[] = {{&d.w1p, w1p.data(), w1p.size()}, {&d.b1, b1.data(), b1.size()},
                    {&d.w2p, w2p.data(), w2p.size()}, {&d.b2, b2.data(), b2.size()},
                    {&d.w3p, w3p.data(), w3p.size()}, {&d.b3, b3.data(), b3.size()},
                    {&d.w4, w4.data(), w4.size()},    {&d.b4, f2b, 2}};
  for (const Up& u : ups) {
    AG2_HIP(c, u.b->reserve(u.n * 4));
    AG2_HIP(c, hipMemcpyAsync(u.b->p, u.p, u.n * 4, hipMemcpyHostToDevice, c->stream));
  }
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  int rc = lenet_pack_weights_x3(c, c1w, c2w);
  if (rc) return rc;
  rc = lenet_pack_fc_x3(c, w3p.data());
  if (rc) return rc;
  d.use_x3 = getenv("AG2_LENET_F32") == nullptr;
  d.use_bands = getenv("AG2_LENET_WHOLE") == nullptr;
  d.loaded = true;
  return 0;
}

// d_n (frame mode, three-term bf16 path only): n is the capacity of the image list, its length
// is read from *d_n on the device.
int launch_lenet(ag2_ctx* c, const uint8_t* d_images, size_t n, float* d_logits, int ev_mid,
                 const unsigned* d_n) {
  if (d_n && !c->net.use_x3) return set_err(c, AG2_ERR_STATE, "frame mode needs the default LeNet path");
  if (n == 0) {
    if (ev_mid >= 0) AG2_HIP(c, stage_event(c, ev_mid));
    return 0;
  }
  LeNetDev& d = c->net;
  AG2_HIP(c, c->d_act1.reserve(n * (size_t)kFcK * 4));
  const size_t lds = sizeof(ConvShared);
  if (!(c->func_attr_done & kAttrLenetConv)) {
    AG2_HIP(c, hipFuncSetAttribute((const void*)k_lenet_conv,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    c->func_attr_done |= kAttrLenetConv;
  }
  if (d.use_x3) {  // bf16 matrix cores, operands split into three exact bf16 terms (k_lenet_x3.hip)
    const int rc = launch_lenet_conv_x3(c, d_images, n, c->d_act1.as<float>(), d_n);
    if (rc) return rc;
  } else {
    const int grid = (int)std::min<size_t>(n, 512);
    hipLaunchKernelGGL(k_lenet_conv, dim3(grid), dim3(kConvThreads), lds, c->stream, d_images, (int)n,
                       d.w1p.as<float>(), d.b1.as<float>(), d.w2p.as<float>(), d.b2.as<float>(),
                       c->d_act1.as<float>());
  }
  if (ev_mid >= 0) AG2_HIP(c, stage_event(c, ev_mid));
  int n_pad = 0, ksplit = 0;
  if (d.use_x3) {  // ip1 with the same three-term split on the bf16 matrix cores
    const int rc = launch_lenet_fc1_x3(c, n, &n_pad, &ksplit, d_n);
    if (rc) return rc;
  } else {
    const int mtiles = (int)((n + kFcBM - 1) / kFcBM);
    n_pad = mtiles * kFcBM;
    // split K (75 chunks of 96) so that small batches still put >= 2 workgroups on every CU
    static const int kSplits[] = {1, 3, 5, 15, 25};
    ksplit = 25;
    for (int ks : kSplits)
      if ((long long)mtiles * 4 * ks >= 512) {
        ksplit = ks;
        break;
      }
    AG2_HIP(c, c->d_fcpart.reserve((size_t)ksplit * n_pad * kFcN * 4));
    hipLaunchKernelGGL(k_lenet_fc1, dim3(mtiles, 4, ksplit), dim3(256), 0, c->stream,
                       c->d_act1.as<float>(), (int)n, n_pad, d.w3p.as<float>(), 75 / ksplit,
                       c->d_fcpart.as<float>());
  }
  hipLaunchKernelGGL(k_lenet_fc_finish, dim3(((int)n + 3) / 4), dim3(256), 0, c->stream,
                     c->d_fcpart.as<float>(), (int)n, d_n, n_pad, ksplit, d.b3.as<float>(),
                     d.w4.as<float>(), d.b4.as<float>(), d_logits);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

}  // namespace ag2
