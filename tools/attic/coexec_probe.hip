// Dev tool: do v_mfma_f64_16x16x4_f64 and v_fma_f64 execute concurrently on gfx950 (MI355X lists 78.6 TFLOP/s for BOTH the fp64
// vector and the fp64 matrix rate)?  One wave per SIMD issues, per loop iteration, 8 independent MFMAs and NF independent FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NF, bool MF>
__global__ __launch_bounds__(256, 2) void probe(double* out, int iters) {
  v4d acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double f[NF > 0 ? NF : 1];
  for (int i = 0; i < NF; ++i) f[i] = threadIdx.x * 1e-3 + i;
  double a = threadIdx.x * 1e-3 + 1.0, b = blockIdx.x * 1e-4 + 0.5, c = 1.0000001;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MF) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NF / 8; ++j) f[i * (NF / 8) + j] = __builtin_fma(f[i * (NF / 8) + j], c, b);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NF; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NF, bool MF>
void run(const char* name) {
  const int nb = 256 * 2, iters = 4000;
  double* out; hipMalloc(&out, sizeof(double) * nb * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NF, MF><<<nb, 256>>>(out, 50); hipDeviceSynchronize();
  hipEventRecord(e0); probe<NF, MF><<<nb, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)nb * 4;
  const double fl_m = MF ? waves * iters * 8 * 2048.0 : 0, fl_f = waves * iters * NF * 128.0;
  printf("%-28s %8.3f ms   mfma %6.2f TF + fma %6.2f TF = %6.2f TFLOP/s\n", name, ms, fl_m / ms / 1e9, fl_f / ms / 1e9, (fl_m + fl_f) / ms / 1e9);
  hipFree(out);
}
int main() {
  run<0, true>("8 mfma");
  run<8, true>("8 mfma + 8 fma");
  run<32, true>("8 mfma + 32 fma");
  run<64, true>("8 mfma + 64 fma");
  run<96, true>("8 mfma + 96 fma");
  run<64, false>("64 fma");
  return 0;
}
