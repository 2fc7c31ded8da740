// Latency / issue probe for the scalar-ish fp64 chains of the leaf kernel (one wave, s_memtime around unrolled
// sequences).  Dev tool: prints cycles per instruction for dependent and independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int J> __device__ __forceinline__ double bcast16(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, false); }
template <int J> __device__ __forceinline__ void fnma_b(double& acc, double l, double m) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(l), "v"(m), "n"(J));
}
__device__ __forceinline__ unsigned long long now() { unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }

__global__ void probe(double* p, unsigned long long* out) {
  __shared__ double sh[512];
  const int l = threadIdx.x;
  sh[l] = p[l]; sh[l + 64] = p[l + 64]; sh[l + 128] = 1.0; sh[l + 192] = 0.5;
  __syncthreads();
  double x = p[l], y = p[l + 64], z = 1.000001;
  unsigned long long t0, t1;
  // 1: 64 dependent fma
  t0 = now();
#pragma unroll
  for (int i = 0; i < 64; ++i) x = fma(x, z, y);
  asm volatile("" : "+v"(x)); t1 = now(); out[0] = t1 - t0;
  // 2: 16 dependent rsq
  t0 = now();
#pragma unroll
  for (int i = 0; i < 16; ++i) x = __builtin_amdgcn_rsq(x + 2.0);
  asm volatile("" : "+v"(x)); t1 = now(); out[1] = t1 - t0;
  // 3: 64 independent fma (8 chains)
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = x + i;
  t0 = now();
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = fma(a[i], z, y);
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
  t1 = now(); out[2] = t1 - t0;
  // 4: 64 fmac_dpp on 8 independent accumulators
  t0 = now();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    fnma_b<1>(a[0], y, y); fnma_b<2>(a[1], y, y); fnma_b<3>(a[2], y, y); fnma_b<4>(a[3], y, y);
    fnma_b<5>(a[4], y, y); fnma_b<6>(a[5], y, y); fnma_b<7>(a[6], y, y); fnma_b<8>(a[7], y, y);
  }
  t1 = now(); out[3] = t1 - t0;
  // 5: 32 dependent (mov_dpp -> fma)
  t0 = now();
#pragma unroll
  for (int i = 0; i < 32; ++i) x = fma(bcast16<3>(x), z, y);
  asm volatile("" : "+v"(x)); t1 = now(); out[4] = t1 - t0;
  // 6: 32 dependent LDS reads (address depends on the value read)
  int idx = l;
  t0 = now();
#pragma unroll
  for (int i = 0; i < 32; ++i) { double v = sh[idx & 255]; idx = (int)v + l; }
  asm volatile("" : "+v"(idx)); t1 = now(); out[5] = t1 - t0;
  // 7: 32 independent broadcast LDS reads then sum
  t0 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += sh[128 + 2 * i];
  asm volatile("" : "+v"(s)); t1 = now(); out[6] = t1 - t0;
  // 8: 32 dependent readlane -> fma
  t0 = now();
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
    x = fma(__hiloint2double(hi, lo), z, y);
  }
  asm volatile("" : "+v"(x)); t1 = now(); out[7] = t1 - t0;
  // 9: 32 dependent v_mul_f64
  t0 = now();
#pragma unroll
  for (int i = 0; i < 32; ++i) x = x * z;
  asm volatile("" : "+v"(x)); t1 = now(); out[8] = t1 - t0;
  // 10: 16 x (cmp + 2 cndmask) dependent
  t0 = now();
#pragma unroll
  for (int i = 0; i < 16; ++i) { x = (x > 0.0) ? x : 1.0; x = x * z; }
  asm volatile("" : "+v"(x)); t1 = now(); out[9] = t1 - t0;
  // 11: empty
  t0 = now(); t1 = now(); out[10] = t1 - t0;
  p[l] = x + s + idx + a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
}
int main() {
  double* p; unsigned long long* o;
  CK(hipMalloc(&p, 8 * 256)); CK(hipMalloc(&o, 8 * 16));
  double h[256]; for (int i = 0; i < 256; ++i) h[i] = 1.0 + 0.001 * i;
  CK(hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice));
  unsigned long long r[16];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, o); CK(hipDeviceSynchronize()); }
  CK(hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost));
  const char* nm[] = {"64 dependent fma", "16 dependent rsq(+add)", "64 fma, 8 chains", "64 fmac_dpp(+nop), 8 chains", "32 dep mov_dpp+fma",
                      "32 dependent LDS reads", "32 indep LDS bcast reads + adds", "32 dep readlane+fma", "32 dependent mul", "16 dep cmp+sel+mul", "empty"};
  const int cnt[] = {64, 16, 64, 64, 32, 32, 32, 32, 32, 16, 1};
  for (int i = 0; i < 11; ++i) printf("%-36s %6llu cycles  %.1f / op\n", nm[i], r[i], (double)(r[i] - r[10]) / cnt[i]);
  return 0;
}
