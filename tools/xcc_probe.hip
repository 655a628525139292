// Which XCD does workgroup b run on?  Reads HW_REG_XCC_ID (s_getreg) per block and compares it with b & 7, the
// round-robin rule the tile maps rely on for L2 locality (development probe, not shipped).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
  // hwreg id 20 = HW_REG_XCC_ID on gfx940+, bits [3:0]
  const int x = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  const int n = 4096;
  int* d;
  hipMalloc(&d, n * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(n), dim3(512), 0, 0, d);
  std::vector<int> h(n);
  hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
  int match = 0, hist[16] = {0};
  for (int b = 0; b < n; ++b) {
    match += (h[b] == (b & 7));
    hist[h[b] & 15]++;
  }
  printf("blocks %d, xcc_id == (block & 7) for %d; first 16:", n, match);
  for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
  printf("\nhistogram:");
  for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
  printf("\n");
  return 0;
}
