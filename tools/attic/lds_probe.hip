// LDS read issue-rate probe (one wave or four): N back-to-back ds_read instructions of one kind, one s_waitcnt at the
// end.  Dev tool behind the fragment-read choices in gpp_leaf.hip / gpp_gemm.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long now() { unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }

template <int MODE>
__device__ __forceinline__ unsigned long long run(unsigned addr) {
  double r[32]; v2d q[16];
  unsigned long long t0 = now();
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 136));
  } else if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(addr), "n"(i * 272));
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(q[i]) : "v"(addr), "n"(i * 2), "n"(i * 2 + 64));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = now();
  if (MODE == 0) { for (int i = 0; i < 32; ++i) asm volatile("" ::"v"(r[i])); }
  else { for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(q[i])); }
  return t1 - t0;
}
__global__ void probe(unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) double sh[4096];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)sh;
  unsigned long long r[16];
  r[0] = run<0>(base + 8 * l);                 // b64, lane-contiguous (conflict-free)
  r[1] = run<0>(base + 8 * (l & 15));          // b64, 16 distinct addresses (4-way broadcast)
  r[2] = run<0>(base);                         // b64, full broadcast
  r[3] = run<1>(base + 16 * l);                // b128 contiguous
  r[4] = run<1>(base + 16 * (l & 15));         // b128, 16 distinct
  r[5] = run<1>(base);                         // b128 broadcast
  r[6] = run<2>(base + 8 * l);                 // read2_b64 contiguous
  r[7] = run<2>(base);                         // read2_b64 broadcast
  r[8] = run<0>(base + 128 * (l & 15) + 8 * (l >> 4));  // b64, 16 rows same column group (stride 128 B): conflicts
  {  // leaf-kernel fragment patterns: old swizzle col ^ (row>>1), new swizzle col ^ (4*(m&3) + (m>>2)), m = row>>1
    const int n = l & 15, k = l >> 4, m = n >> 1, i4 = l & 3;
    r[9] = run<0>(base + 8 * (n * 16 + (k ^ m)));                                   // B^T fragment, old
    r[10] = run<0>(base + 8 * (n * 16 + (k ^ (4 * (m & 3) + (m >> 2)))));           // B^T fragment, new
    r[11] = run<0>(base + 8 * (i4 * 16 + (k ^ (i4 >> 1))));                         // A 4x4 fragment, old
    r[12] = run<0>(base + 8 * (i4 * 16 + (k ^ (4 * ((i4 >> 1) & 3)))));             // A 4x4 fragment, new
    r[13] = run<0>(base + 8 * (k * 16 + (n ^ (k >> 1))));                           // C layout rows, old
    r[14] = run<0>(base + 8 * (n * 16 + ((l >> 4) * 0 + (3 ^ m))) );                // column read (16 rows), old
    r[15] = run<0>(base + 8 * (n * 16 + (3 ^ (4 * (m & 3) + (m >> 2)))));           // column read, new
  }
  if (l == 0) for (int i = 0; i < 16; ++i) out[w * 16 + i] = r[i];
}
int main() {
  unsigned long long* o; CK(hipMalloc(&o, 8 * 64));
  const char* nm[] = {"32 ds_read_b64 contiguous", "32 ds_read_b64 16 distinct addr", "32 ds_read_b64 broadcast", "16 ds_read_b128 contiguous",
                      "16 ds_read_b128 16 distinct", "16 ds_read_b128 broadcast", "16 ds_read2_b64 contiguous", "16 ds_read2_b64 broadcast",
                      "32 ds_read_b64 stride-128B rows", "B^T frag old swizzle", "B^T frag new swizzle", "A4x4 frag old", "A4x4 frag new",
                      "C-layout rows", "column read old", "column read new"};
  const int cnt[] = {32, 32, 32, 16, 16, 16, 16, 16, 32, 32, 32, 32, 32, 32, 32, 32};
  for (int waves = 1; waves <= 4; waves *= 4) {
    unsigned long long r[64];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 0, 0, o); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost));
    printf("-- %d wave(s) in the work-group (wave 0 shown)\n", waves);
    for (int i = 0; i < 16; ++i) printf("%-36s %6llu cycles  %.1f / instr\n", nm[i], r[i], (double)(r[i] - 40) / cnt[i]);
  }
  return 0;
}
