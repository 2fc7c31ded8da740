// Micro-benchmark: v_mfma_f64_16x16x4_f64 and v_fma_f64 issue rates on this GPU with the in-kernel clock
// (s_memtime / s_memrealtime), for zero and random operands.  Dev tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, rt; };
template <int NACC>
__global__ __launch_bounds__(256) void probe_mfma(double* out, Stamp* st, int iters, double scale) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = (threadIdx.x * 1e-3 + 1.0) * scale, b = (blockIdx.x * 1e-4 + 0.5) * scale;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void probe_mfma4(double* out, Stamp* st, int iters, double scale) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = (threadIdx.x * 1e-3 + 1.0) * scale, b = (blockIdx.x * 1e-4 + 0.5) * scale;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void probe_fma(double* out, Stamp* st, int iters, double scale) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = (threadIdx.x * 1e-3 + 1.0) * scale, b = (blockIdx.x * 1e-4 + 0.5) * scale;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}
template <typename F>
void run(const char* name, F launch, int nb, double flop_per_wave_iter, int iters, int inst_per_iter) {
  double* out; Stamp* st;
  CK(hipMalloc(&out, sizeof(double) * nb * 256));
  CK(hipMalloc(&st, sizeof(Stamp) * nb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(out, st, 100, nb);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch(out, st, iters, nb);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  Stamp* h = new Stamp[nb];
  CK(hipMemcpy(h, st, sizeof(Stamp) * nb, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < nb; ++i) { cyc += h[i].cyc; rt += h[i].rt; }
  cyc /= nb; rt /= nb;
  double ghz = cyc / (rt * 10.0);  // memrealtime ticks at 100 MHz -> ns = rt*10
  double flops = (double)nb * 4 * iters * flop_per_wave_iter;
  printf("%-34s %8.3f ms %7.2f TFLOP/s  clock %.3f GHz  %.1f cyc/inst/wave\n", name, ms, flops / ms / 1e9, ghz,
         cyc / ((double)iters * inst_per_iter));
  delete[] h; (void)hipFree(out); (void)hipFree(st);
}
int main() {
  const int CU = 256;
#define MF(N, BPC, SC, IT) run("mfma acc=" #N " wg/cu=" #BPC " scale=" #SC, [](double* o, Stamp* s, int it, int nb) { probe_mfma<N><<<nb, 256>>>(o, s, it, SC); }, CU * BPC, N * 2048.0, IT, N)
  MF(4, 1, 1.0, 20000); MF(4, 1, 0.0, 20000); MF(16, 1, 1.0, 5000); MF(16, 1, 0.0, 5000); MF(4, 2, 1.0, 20000); MF(16, 2, 1.0, 5000);
  MF(1, 1, 1.0, 40000); MF(2, 1, 1.0, 40000); MF(8, 2, 1.0, 10000);
  MF(4, 4, 1.0, 10000); MF(8, 4, 1.0, 5000); MF(4, 8, 1.0, 5000); MF(2, 8, 1.0, 10000); MF(8, 3, 1.0, 5000); MF(16, 3, 1.0, 2500);
#define M4(N, BPC, SC, IT) run("mfma4x4x4 acc=" #N " wg/cu=" #BPC, [](double* o, Stamp* s, int it, int nb) { probe_mfma4<N><<<nb, 256>>>(o, s, it, SC); }, CU * BPC, N * 512.0, IT, N)
  M4(4, 1, 1.0, 20000); M4(8, 2, 1.0, 10000); M4(8, 4, 1.0, 10000);
  // long unrolled bodies: is the 4x4x4 shortfall (68 of 78.6 TF) loop overhead?  and does the operand scale matter (power)?
  M4(64, 1, 1.0, 4000); M4(64, 2, 1.0, 2000); M4(64, 2, 0.0, 2000); M4(32, 2, 1.0, 4000); M4(16, 2, 1.0, 8000);
#define FM(N, BPC, SC, IT) run("fma acc=" #N " wg/cu=" #BPC " scale=" #SC, [](double* o, Stamp* s, int it, int nb) { probe_fma<N><<<nb, 256>>>(o, s, it, SC); }, CU * BPC, N * 128.0, IT, N)
  FM(8, 1, 1.0, 100000); FM(8, 2, 1.0, 100000); FM(8, 4, 1.0, 100000); FM(8, 4, 0.0, 100000);
  return 0;
}
