// Which XCD / SE / CU does work-group b land on?  (s_getreg HW_REG_XCC_ID and HW_ID.)  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__global__ void k(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  // keep the work-group resident for a while so that the grid spreads over the chip
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
}
int main() {
  const int nb = 1024;
  unsigned* d; CK(hipMalloc(&d, 8 * nb));
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 65536, 0, d, 200000);
  CK(hipDeviceSynchronize());
  unsigned h[2 * nb]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("block : xcc_id (low 4 bits) se cu   [HW_ID: cu_id bits 8-11, sh 12, se 13-15]\n");
  for (int b = 0; b < 40; ++b) printf("%4d : xcc %u  se %u  cu %u\n", b, h[2 * b] & 0xf, (h[2 * b + 1] >> 13) & 7, (h[2 * b + 1] >> 8) & 0xf);
  int hist[16] = {0};
  int rr = 0;
  for (int b = 0; b < nb; ++b) { hist[h[2 * b] & 0xf]++; if ((int)(h[2 * b] & 0xf) == b % 8) ++rr; }
  printf("blocks per xcc:"); for (int i = 0; i < 8; ++i) printf(" %d", hist[i]); printf("\n%d of %d blocks have xcc == block %% 8\n", rr, nb);
  return 0;
}
