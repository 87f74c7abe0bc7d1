// Which workgroups of a 512-workgroup launch (256 threads, 70 KB of LDS: two per CU) share a CU?
// The convolution kernel's last round is 242 of 512 workgroups: whether those sit two to a CU or one to a
// CU decides how long that round lasts.  Prints blockIdx -> (XCC, SE, CU) and the pairing statistics.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wg_placement tools/experiments/wg_placement.hip && /tmp/wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256, 2) k_where(unsigned* out, int spin) {
  extern __shared__ unsigned char lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) {
    lds[0] = 1;
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay resident long enough for the whole grid to be placed
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

int main() {
  const int G = 512;
  unsigned* d;
  hipMalloc(&d, G * 8);
  hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
  hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 70 * 1024, 0, d, 2000000);
  std::vector<unsigned> h(2 * G);
  hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < G; b++) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu_id;
    cu[key].push_back(b);
    if (b < 40) printf("block %3d: xcc %u se %u sh %u cu %2u\n", b, xcc, se, sh, cu_id);
  }
  printf("distinct CUs: %zu\n", cu.size());
  int shown = 0, both_low = 0, one_low = 0;
  for (auto& kv : cu) {
    if (shown++ < 24) {
      printf("cu %05x:", kv.first);
      for (int b : kv.second) printf(" %d", b);
      printf("\n");
    }
    int low = 0;
    for (int b : kv.second) low += b < 242;
    both_low += low >= 2;
    one_low += low == 1;
  }
  printf("CUs holding two of blocks 0..241: %d, exactly one: %d\n", both_low, one_low);
  return 0;
}
