// Does a grid of 2 work-groups per CU become fully RESIDENT on a CU-masked stream whose shader engines have UNEQUAL CU counts?
// (The static-schedule executor needs every work-group of its launch resident; with 32 panel CUs — one per shader engine — the
// mask is uniform.  Could the panel's reservation shrink to 8 or 16 CUs?)  Each work-group takes 74 KB of LDS (two per CU),
// checks in on a counter and spins until everybody has, or 0.5 s.  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int* count, int total) {
  extern __shared__ double sm[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  sm[threadIdx.x] = 1.0;
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = xcc; out[3 * blockIdx.x + 1] = hwid;
    __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = (long long)wall_clock64();
    int seen = 0;
    while ((seen = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < total &&
           (long long)wall_clock64() - t0 < 50000000LL) __builtin_amdgcn_s_sleep(8);
    out[3 * blockIdx.x + 2] = (unsigned)seen;
  }
  __syncthreads();
}
int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int lds = 74 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int reserve : {32, 16, 8, 1}) {
    uint32_t mask[32] = {0};
    for (int c = 0; c < ncu; ++c) if (c >= reserve) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (ncu + 31) / 32, mask));
    for (int extra : {0}) {
      const int total = 2 * (ncu - reserve) + extra;
      unsigned* d; int* cnt; CK(hipMalloc(&d, 12 * total)); CK(hipMalloc(&cnt, 4)); CK(hipMemset(cnt, 0, 4)); CK(hipMemset(d, 0, 12 * total));
      hipLaunchKernelGGL(k, dim3(total), dim3(256), lds, s, d, cnt, total);
      CK(hipStreamSynchronize(s));
      std::vector<unsigned> h(3 * total); CK(hipMemcpy(h.data(), d, 12 * total, hipMemcpyDeviceToHost));
      int ok = 0; int percu[8][64] = {{0}};
      for (int b = 0; b < total; ++b) {
        if ((int)h[3 * b + 2] >= total) ++ok;
        const int x = h[3 * b] & 7, se = (h[3 * b + 1] >> 13) & 7, cu = (h[3 * b + 1] >> 8) & 0xf;
        percu[x][(se * 16 + cu) & 63]++;
      }
      int used = 0, two = 0;
      for (int x = 0; x < 8; ++x) for (int q = 0; q < 64; ++q) { if (percu[x][q]) ++used; if (percu[x][q] == 2) ++two; }
      printf("reserve %2d CUs (mask of the first bits): grid %d -> %d work-groups saw everybody resident; %d CUs used, %d with two work-groups\n",
             reserve, total, ok, used, two);
      CK(hipFree(d)); CK(hipFree(cnt));
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
