// What core clock does the chip hold while the dominant GEMM launch (LAUUM at N = 20000) runs?  A one-wave sampler kernel
// reads s_memtime (core clock) and s_memrealtime (100 MHz) every ~50 us on its own stream while gpp_lauum runs on another;
// prints the clock per millisecond.  Dev tool: hipcc --offload-arch=gfx950 -O2 tools/attic/clock_probe.hip -Iinclude
//   -Lgp-plus_amd -lgpp_hip -Wl,-rpath,$PWD/gp-plus_amd -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gpp.h"
struct Stamp { unsigned long long cyc, rt; };
__global__ void sampler(Stamp* st, int n) {
  for (int i = 0; i < n; ++i) {
    st[i].cyc = __builtin_amdgcn_s_memtime();
    st[i].rt = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 5000) __builtin_amdgcn_s_sleep(32);  // 50 us at 100 MHz
  }
}
__global__ void fill(double* A, long n, long ld) {
  long i = blockIdx.x, j = threadIdx.x + (long)blockIdx.y * blockDim.x;
  if (j < n) A[i * ld + j] = (j <= i ? 1e-3 * ((i * 31 + j * 17) % 97) : 0.0) + (i == j ? 1.0 : 0.0);
}
int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 20000, ld = N;
  double *Li, *Ki;
  Stamp* st;
  const int ns = 3000;  // 150 ms
  if (hipMalloc(&Li, sizeof(double) * N * ld) || hipMalloc(&Ki, sizeof(double) * N * ld) || hipMalloc(&st, sizeof(Stamp) * ns)) return 1;
  fill<<<dim3(N, (N + 255) / 256), 256>>>(Li, N, ld);
  gpp_handle_t h;
  if (gpp_create(&h, 0)) return 2;
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  (void)gpp_set_stream(h, s2);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    sampler<<<1, 64, 0, s1>>>(st, ns);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // 20 ms idle, then two LAUUM launches back to back
    (void)hipStreamSynchronize(s2);
    struct timespec ts = {0, 20000000}; nanosleep(&ts, nullptr);
    (void)hipEventRecord(e0, s2);
    int rc = gpp_lauum(h, Li, N, ld, Ki, ld);
    rc |= gpp_lauum(h, Li, N, ld, Ki, ld);
    (void)hipEventRecord(e1, s2);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) continue;
    std::vector<Stamp> v(ns);
    (void)hipMemcpy(v.data(), st, sizeof(Stamp) * ns, hipMemcpyDeviceToHost);
    printf("rc %d; two LAUUM launches: %.2f ms (%.1f TFLOP/s)\n", rc, ms, 2.0 * N * N * N / 3.0 / ms / 1e9);
    for (int i = 0; i + 20 < ns; i += 20) {
      double dc = (double)(v[i + 20].cyc - v[i].cyc), dr = (double)(v[i + 20].rt - v[i].rt);
      printf("t=%6.1f ms  clock %.3f GHz\n", (double)(v[i].rt - v[0].rt) / 1e5, dc / dr * 0.1);
    }
  }
  return 0;
}
