// Layout discovery for v_mfma_f64_4x4x4_4b_f64 on gfx950: which (A lane, B lane) pairs feed which output lane,
// for cbsz/abid/blgp variants.  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
template <int CBSZ, int ABID, int BLGP>
__global__ void k(double* out) {  // grid = 64*64 waves: (la, lb)
  const int lane = threadIdx.x, la = blockIdx.x / 64, lb = blockIdx.x % 64;
  double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, BLGP);
  out[(size_t)blockIdx.x * 64 + lane] = d;
}
template <int CBSZ, int ABID, int BLGP>
int run(const char* name) {
  double* d; CK(hipMalloc(&d, sizeof(double) * 4096 * 64));
  k<CBSZ, ABID, BLGP><<<4096, 64>>>(d);
  CK(hipDeviceSynchronize());
  std::vector<double> h(4096 * 64);
  CK(hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  printf("== %s: for each output lane: list of (la,lb) contributing\n", name);
  for (int lo = 0; lo < 64; ++lo) {
    printf("out %2d:", lo);
    for (int p = 0; p < 4096; ++p) if (h[(size_t)p * 64 + lo] != 0.0) printf(" (%d,%d)%s", p / 64, p % 64, h[(size_t)p*64+lo] == 1.0 ? "" : "*");
    printf("\n");
  }
  CK(hipFree(d));
  return 0;
}
int main() {
  if (run<0, 0, 0>("cbsz0 abid0")) return 1;
  if (run<2, 1, 0>("cbsz2 abid1")) return 1;
  if (run<1, 1, 0>("cbsz1 abid1")) return 1;
  return 0;
}
