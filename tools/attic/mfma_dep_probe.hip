// Back-to-back issue interval of v_mfma_f64_4x4x4_4b_f64 as a function of how many independent accumulators rotate
// (1 = fully dependent chain).  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned long long now() { unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }
template <int NACC>
__device__ __forceinline__ unsigned long long run(double a, double b, double* sink) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  unsigned long long t0 = now();
#pragma unroll
  for (int r = 0; r < 64 / NACC; ++r)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  asm volatile("" : "+v"(s));
  unsigned long long t1 = now();
  *sink += s;
  return t1 - t0;
}
__global__ void probe(double* p, unsigned long long* out) {
  double a = p[threadIdx.x], b = p[threadIdx.x + 64], sink = 0;
  out[0] = run<1>(a, b, &sink); out[1] = run<2>(a, b, &sink); out[2] = run<4>(a, b, &sink); out[3] = run<8>(a, b, &sink);
  out[4] = run<16>(a, b, &sink);
  p[threadIdx.x] = sink;
}
int main() {
  double* p; unsigned long long* o; CK(hipMalloc(&p, 8 * 256)); CK(hipMalloc(&o, 64)); CK(hipMemset(p, 0, 8 * 256));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, o); CK(hipDeviceSynchronize()); }
  unsigned long long r[5]; CK(hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost));
  const int n[] = {1, 2, 4, 8, 16};
  for (int i = 0; i < 5; ++i) printf("64 mfma_f64_4x4x4, %2d rotating accumulators: %5llu cycles  %.1f / mfma\n", n[i], r[i], (double)(r[i] - 40) / 64);
  return 0;
}
