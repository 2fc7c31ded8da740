// What does a hand-off between work-groups INSIDE one kernel cost on this GPU?  A token (a payload of P doubles) is relayed H times
// round-robin over G work-groups through flags in device memory; every hop reads the previous hop's payload, adds one and
// publishes it.  Variants: (0) release / acquire fences at agent scope around a plain payload (what the memory model asks for when
// the work-groups may sit on different XCDs: buffer_wbl2 / buffer_inv), (1) the payload moved with agent-scope relaxed atomics
// (sc1 loads / stores that bypass the per-XCD L2) and only a waitcnt before the flag.  Each alone and beside a kernel that keeps
// dirtying the L2s from another stream.  Dev tool for the cooperative panel kernel:
//   hipcc --offload-arch=gfx950 -O2 tools/attic/flag_probe.hip -o tools/flag_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void relay(int* flags, double* buf, int P, int H, unsigned long long* cycles, int* bad) {
  const int G = gridDim.x, tid = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int h = blockIdx.x; h < H; h += G) {
    if (h > 0) {
      if (tid == 0) {
        long spins = 0;
        while (__hip_atomic_load(&flags[h - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 50000000L) { *bad = 1; break; }  // never hang the box
        }
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    const double* src = buf + (size_t)((h + 1) & 1) * P;
    double* dst = buf + (size_t)(h & 1) * P;
    for (int i = tid; i < P; i += 256) {
      double v;
      if (MODE == 0) v = src[i];
      else v = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = (h == 0) ? 1.0 : v + 1.0;
      if (MODE == 0) dst[i] = v;
      else __hip_atomic_store(&dst[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have left the CU
    __syncthreads();
    if (tid == 0) {
      if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(&flags[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (h == H - 1 && tid == 0) cycles[0] = __builtin_amdgcn_s_memtime() - t0;
  }
}

__global__ void dirty(double* p, size_t n_per_wg, int rounds) {
  double* q = p + (size_t)blockIdx.x * n_per_wg;
  for (int r = 0; r < rounds; ++r)
    for (size_t i = threadIdx.x; i < n_per_wg; i += blockDim.x) q[i] = (double)(r + i);
}
__global__ void tiny(int* x) { if (threadIdx.x == 0) x[0] += 1; }

int main() {
  const int H = 2000, PMAX = 16384;
  int* flags; double* buf; unsigned long long* cyc; int* bad; double* big;
  CK(hipMalloc(&flags, H * sizeof(int))); CK(hipMalloc(&buf, 2 * PMAX * sizeof(double)));
  CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  const size_t per_wg = 1 << 17; const int dwgs = 1024;  // 1 MiB per work-group, 1 GiB in all
  CK(hipMalloc(&big, per_wg * dwgs * sizeof(double)));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("relay of H=%d hops; us per hop (kernel time / H), payload checked\n", H);
  for (int busy = 0; busy < 2; ++busy)
    for (int mode = 0; mode < 2; ++mode)
      for (int G : {2, 8, 32})
        for (int P : {0, 2048, 16384}) {
          CK(hipMemsetAsync(flags, 0, H * sizeof(int), s1));
          CK(hipMemsetAsync(buf, 0, 2 * PMAX * sizeof(double), s1));
          CK(hipStreamSynchronize(s1));
          if (busy) hipLaunchKernelGGL(dirty, dim3(dwgs), dim3(256), 0, s2, big, per_wg, 40);
          CK(hipEventRecord(e0, s1));
          if (mode == 0) hipLaunchKernelGGL(relay<0>, dim3(G), dim3(256), 0, s1, flags, buf, P, H, cyc, bad);
          else hipLaunchKernelGGL(relay<1>, dim3(G), dim3(256), 0, s1, flags, buf, P, H, cyc, bad);
          CK(hipEventRecord(e1, s1));
          CK(hipStreamSynchronize(s1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          CK(hipDeviceSynchronize());
          std::vector<double> hb(2 * PMAX); int hbad;
          CK(hipMemcpy(hb.data(), buf, 2 * PMAX * sizeof(double), hipMemcpyDeviceToHost));
          CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
          int wrong = 0;
          const double* last = hb.data() + (size_t)((H - 1) & 1) * P;
          for (int i = 0; i < P; ++i) wrong += (last[i] != (double)H);
          printf("%s mode %d (%s)  G=%2d  P=%5d doubles : %7.2f us/hop   wrong=%d timeout=%d\n", busy ? "BUSY" : "idle", mode,
                 mode ? "sc1 atomics, no fence" : "fences", G, P, 1e3 * ms / H, wrong, hbad);
        }
  // reference: a chain of H tiny launches on one stream
  int* x; CK(hipMalloc(&x, 4)); CK(hipMemset(x, 0, 4));
  CK(hipEventRecord(e0, s1));
  for (int i = 0; i < H; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, x);
  CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("chain of %d tiny launches: %.2f us per launch\n", H, 1e3 * ms / H);
  return 0;
}
