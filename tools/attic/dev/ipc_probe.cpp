// Round-6 probe for the push transport: can two PROCESSES on one device share (a) an ordinary hipMalloc buffer and (b) an uncached /
// fine-grained flag page through hipIpcMemHandle on this pool (dmabuf IPC only), write into the peer's with hipMemcpy2DAsync and a
// kernel store, and see the flag from a spinning kernel on the other side?   hipcc --offload-arch=gfx950 -O2 ipc_probe.cpp -o ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s -> %s\n", (int)getpid(), #x, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)

__global__ void spin_until(volatile int* flag, int want, long long budget, int* out) {
  const long long t0 = wall_clock64();
  int v;
  while ((v = __hip_atomic_load((int*)flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < want)
    if (wall_clock64() - t0 > budget) { *out = -1; return; }
  *out = v;
}
__global__ void store_flag(int* flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0: flag page uncached, 1: fine-grained, 2: plain hipMalloc
  int a2b[2], b2a[2];
  if (pipe(a2b) || pipe(b2a)) return 2;
  const size_t rows = 256, cols = 4096, ld = 8192;
  pid_t pid = fork();  // before any HIP call
  if (pid != 0) {  // A: owns buffer + flag page, waits for B's push
    double* buf; int* flags; int* out;
    CK(hipMalloc(&buf, rows * cols * sizeof(double)));
    CK(hipMemset(buf, 0, rows * cols * sizeof(double)));
    if (mode == 0) CK(hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocUncached));
    else if (mode == 1) CK(hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocFinegrained));
    else CK(hipMalloc(&flags, 4096));
    CK(hipMemset(flags, 0, 4096));
    CK(hipMalloc(&out, 4));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h[2];
    CK(hipIpcGetMemHandle(&h[0], buf));
    CK(hipIpcGetMemHandle(&h[1], flags));
    if (write(a2b[1], h, sizeof(h)) != (ssize_t)sizeof(h)) return 2;
    hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, 0, flags, 7, 20LL * 100000000LL, out);
    CK(hipDeviceSynchronize());
    int got; CK(hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost));
    double* hostb = (double*)malloc(rows * cols * sizeof(double));
    CK(hipMemcpy(hostb, buf, rows * cols * sizeof(double), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t r = 0; r < rows; ++r) for (size_t c = 0; c < cols; ++c) bad += hostb[r * cols + c] != (double)(r * ld + c);
    printf("A: mode %d flag seen %d (want 7), payload mismatches %zu of %zu\n", mode, got, bad, rows * cols);
    char done = 1; if (write(a2b[1], &done, 1) != 1) return 2;
    int st; waitpid(pid, &st, 0);
    printf("A: child exit %d\n", WEXITSTATUS(st));
    return (got == 7 && bad == 0) ? 0 : 1;
  }
  // B: opens A's handles, pushes a strided block + raises the flag
  hipIpcMemHandle_t h[2];
  if (read(a2b[0], h, sizeof(h)) != (ssize_t)sizeof(h)) _exit(2);
  double* src; CK(hipMalloc(&src, rows * ld * sizeof(double)));
  double* hs = (double*)malloc(rows * ld * sizeof(double));
  for (size_t i = 0; i < rows * ld; ++i) hs[i] = (double)i;
  CK(hipMemcpy(src, hs, rows * ld * sizeof(double), hipMemcpyHostToDevice));
  void *pbuf, *pflags;
  CK(hipIpcOpenMemHandle(&pbuf, h[0], hipIpcMemLazyEnablePeerAccess));
  CK(hipIpcOpenMemHandle(&pflags, h[1], hipIpcMemLazyEnablePeerAccess));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipMemcpy2DAsync(pbuf, cols * sizeof(double), src, ld * sizeof(double), cols * sizeof(double), rows, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(store_flag, dim3(1), dim3(1), 0, s, (int*)pflags, 7);
  CK(hipStreamSynchronize(s));
  printf("B: pushed\n"); fflush(stdout);
  char done; if (read(a2b[0], &done, 1) != 1) _exit(2);
  CK(hipIpcCloseMemHandle(pbuf)); CK(hipIpcCloseMemHandle(pflags));
  _exit(0);
}
