// Micro-benchmark (dev tool): throughput of v_mfma_f64_16x16x4_f64 vs v_mfma_f64_4x4x4_4b_f64 with DISTINCT operand
// registers per instruction (8 A fragments x 2 B fragments -> 16 accumulators, the register pattern of a 64x64 wave tile),
// optionally interleaved with LDS fragment reads.  Reports cycles per MFMA per wave (s_memtime) and chip TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, rt; };

template <bool LDS>
__global__ __launch_bounds__(256) void probe16(double* out, Stamp* st, int iters) {
  __shared__ double sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = 1e-3 * i;
  __syncthreads();
  v4d acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a[8], b[2];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  b[0] = blockIdx.x * 1e-4 + 0.5; b[1] = b[0] + 1.0;
  const double* sp = sm + (threadIdx.x & 63);
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = sp[((it & 3) * 8 + i) * 64];
      b[0] = sp[2048 + (it & 7) * 64]; b[1] = sp[2048 + 512 + (it & 7) * 64];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[0], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[8 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[7 - i], b[1], acc[8 + i], 0, 0, 0);
    if (!LDS) { a[it & 7] += 1e-9; }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}

template <bool LDS>
__global__ __launch_bounds__(256) void probe4(double* out, Stamp* st, int iters) {
  __shared__ double sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = 1e-3 * i;
  __syncthreads();
  double acc[64];
  for (int i = 0; i < 64; ++i) acc[i] = 0;
  double a[16], b[4];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3 + i;
  for (int i = 0; i < 4; ++i) b[i] = blockIdx.x * 1e-4 + 0.5 + i;
  const double* sp = sm + (threadIdx.x & 63);
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = sp[((it & 1) * 16 + i) * 64];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = sp[2048 + ((it & 3) * 4 + i) * 64];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    if (!LDS) { a[it & 15] += 1e-9; }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 64; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}

template <typename K>
void run(const char* name, K kern, int nb, int iters, int mfma_per_iter, double flop_per_mfma) {
  double* out; Stamp* st;
  hipMalloc(&out, sizeof(double) * nb * 256); hipMalloc(&st, sizeof(Stamp) * nb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<nb, 256>>>(out, st, 50); hipDeviceSynchronize();
  hipEventRecord(e0); kern<<<nb, 256>>>(out, st, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  Stamp* h = new Stamp[nb]; hipMemcpy(h, st, sizeof(Stamp) * nb, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0; for (int i = 0; i < nb; ++i) { cyc += h[i].cyc; rt += h[i].rt; } cyc /= nb; rt /= nb;
  double flops = (double)nb * 4 * iters * mfma_per_iter * flop_per_mfma;
  printf("%-44s %8.3f ms %7.2f TFLOP/s  clock %.3f GHz  %.1f cyc/MFMA/wave\n", name, ms, flops / ms / 1e9, cyc / (rt * 10.0),
         cyc / ((double)iters * mfma_per_iter));
  delete[] h; hipFree(out); hipFree(st);
}
int main() {
  const int CU = 256;
  run("16x16x4 distinct regs          1 wg/cu", probe16<false>, CU * 1, 4000, 16, 2048.0);
  run("16x16x4 distinct regs          2 wg/cu", probe16<false>, CU * 2, 4000, 16, 2048.0);
  run("16x16x4 distinct regs + LDS    1 wg/cu", probe16<true>, CU * 1, 4000, 16, 2048.0);
  run("16x16x4 distinct regs + LDS    2 wg/cu", probe16<true>, CU * 2, 4000, 16, 2048.0);
  run("4x4x4   distinct regs          1 wg/cu", probe4<false>, CU * 1, 4000, 64, 512.0);
  run("4x4x4   distinct regs          2 wg/cu", probe4<false>, CU * 2, 4000, 64, 512.0);
  run("4x4x4   distinct regs + LDS    1 wg/cu", probe4<true>, CU * 1, 4000, 64, 512.0);
  run("4x4x4   distinct regs + LDS    2 wg/cu", probe4<true>, CU * 2, 4000, 64, 512.0);
  return 0;
}
