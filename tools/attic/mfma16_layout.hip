// Dev tool: register layout of v_mfma_f64_16x16x4_f64 on gfx950.  Hypothesis (checked here): lane l supplies A[m=l%16][k=l/16],
// B[k=l/16][n=l%16] and receives D[m = 4*(l/16) + v][n = l%16] in element v = 0..3 of its accumulator.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  v4d acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[l * 4 + v] = acc[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = sin(1.0 + i); hB[i] = cos(0.3 * i); }
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[m * 4 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    e1 = fmax(e1, fabs(hD[l * 4 + v] - ref[(4 * (l / 16) + v) * 16 + l % 16]));   // hypothesis 1
    e2 = fmax(e2, fabs(hD[l * 4 + v] - ref[((l / 16) + 4 * v) * 16 + l % 16]));   // hypothesis 2: row = l/16 + 4 v
  }
  printf("layout check: row = 4*(l/16)+v : max err %.3e ; row = (l/16)+4v : max err %.3e\n", e1, e2);
  return 0;
}
