// gpp_shard.hip — the sharded evaluation behind the C ABI ("gpp_set_comm(handle, comm, rank, nranks)").
//
// gpp_shard_eval runs what gp-plus_amd/sharded.py::ShardedMLLFunction runs — build of the owned block rows, the rank's ticket list
// (factorisation + forward sweep) with the block rows' messages around it, mirror, z / alpha, the back-substitution's list, the
// gradient reduction — from C, so that a C / Fortran / MPI caller of gpp.h can use several GPUs without the Python host.  The
// collectives are the caller's: two callbacks (gpp_set_comm), or RCCL itself (gpp_comm_init_rccl: librccl.so is opened at run time,
// the library has no link-time dependency on it).  Written against the PUBLIC entry points of gpp.h only (plus the handle's fields
// for its own communication stream).  Reference counterpart of the whole call: `mll(output, y)` + `loss.backward()`,
// optim/mll_torch.py:114-117; the reference has no multi-GPU evaluation.
#include "../../include/gpp.h"
#include "gpp_internal.h"

#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

typedef double v2d_t __attribute__((ext_vector_type(2)));

inline int rc(hipError_t e) { return e == hipSuccess ? 0 : 1000 + (int)e; }
#define SH_HIP(expr)                      \
  do {                                    \
    hipError_t _e = (expr);               \
    if (_e != hipSuccess) return rc(_e);  \
  } while (0)
#define SH_TRY(expr)        \
  do {                      \
    int _r = (expr);        \
    if (_r != 0) return _r; \
  } while (0)

// Kc's diagonal block of an owned column block: the lower triangle of D[c] (zeros above)
__global__ void gpp_copy_lower(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j < n) dst[(int64_t)i * ldd + j] = j <= i ? src[(int64_t)i * lds + j] : 0.0;
}

// ---- virtual-rank replay (tools/replay_rank.py; debug entry points, not part of gpp.h) -----------------------------------------------
// A block row that would arrive from another GPU is PLAYED into this rank's buffers by a copy kernel with a collective kernel's
// footprint (a few work-groups of 256-512 threads on whatever CUs have room) that (a) does not start before a given time on the
// device's 100 MHz constant clock — the earliest moment its owner could have sent it — and (b) moves its bytes no faster than a given
// rate — the wire.  Two time stamps (start of the first piece, end of the last) come back relative to the caller's epoch.
__global__ __launch_bounds__(512) void gpp_replay_copy(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src, int64_t lds,
                                                       int rows, int cols, double bytes_per_tick, const unsigned long long* epoch,
                                                       long long not_before, unsigned long long* stamps) {
  __shared__ long long s_start;
  const int tid = threadIdx.x, nth = blockDim.x;
  constexpr int PIECE = 4096;  // 16-byte vectors per piece (64 KiB)
  if (tid == 0) {
    long long now = (long long)wall_clock64();
    if (epoch && not_before >= 0) {
      const long long target = (long long)*epoch + not_before;
      while (now < target) {
        __builtin_amdgcn_s_sleep(16);
        now = (long long)wall_clock64();
      }
    }
    s_start = now;
    if (stamps && blockIdx.x == 0) stamps[0] = (unsigned long long)(now - (epoch ? (long long)*epoch : 0));
  }
  __syncthreads();
  const long long start = s_start;
  const int vpr = cols >> 1;  // (cols even: the callers' column ranges are multiples of the block height or end at an even N)
  const long long nvec = (long long)rows * vpr;
  const long long npieces = (nvec + PIECE - 1) / PIECE;
  for (long long q = blockIdx.x; q < npieces; q += gridDim.x) {
    if (bytes_per_tick > 0.0) {  // pacing: piece q does not start before the wire would have delivered the q pieces in front of it
      if (tid == 0) {
        const long long target = start + (long long)((double)q * (PIECE * 16.0) / bytes_per_tick);
        while ((long long)wall_clock64() < target) __builtin_amdgcn_s_sleep(4);
      }
      __syncthreads();
    }
    const long long v0 = q * PIECE, v1 = v0 + PIECE < nvec ? v0 + PIECE : nvec;
    for (long long v = v0 + tid; v < v1; v += nth) {
      const int r = (int)(v / vpr), c = (int)(v - (long long)r * vpr) << 1;
      const v2d_t x = *reinterpret_cast<const v2d_t*>(src + (int64_t)r * lds + c);
      *reinterpret_cast<v2d_t*>(dst + (int64_t)r * ldd + c) = x;
    }
  }
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      long long now = (long long)wall_clock64();
      if (bytes_per_tick > 0.0) {  // the last byte is not "there" before the wire says so
        const long long target = start + (long long)((double)nvec * 16.0 / bytes_per_tick);
        while (now < target) {
          __builtin_amdgcn_s_sleep(4);
          now = (long long)wall_clock64();
        }
      }
      atomicMax(stamps + 1, (unsigned long long)(now - (epoch ? (long long)*epoch : 0)));
    }
  }
}
__global__ __launch_bounds__(64) void gpp_replay_stamp(unsigned long long* slot, const unsigned long long* epoch) {
  if (threadIdx.x == 0) *slot = wall_clock64() - (epoch ? *epoch : 0ull);
}

// ---- RCCL, opened at run time ------------------------------------------------------------------------------------------------------
struct Uid {
  char internal[128];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value: 128 bytes */ Uid, int) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
};
Rccl g_rccl;
bool rccl_open() {
  if (g_rccl.lib) return true;
  void* l = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!l) l = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!l) l = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!l) return false;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(l, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(l, "ncclCommInitRank"));
  g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(dlsym(l, "ncclBroadcast"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(l, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(l, "ncclCommDestroy"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.Broadcast || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    dlclose(l);
    return false;
  }
  g_rccl.lib = l;
  return true;
}
// (rccl.h: ncclInt8 = 0, ncclInt32 = 2, ncclFloat64 = 8; ncclSum = 0, ncclMax = 2)
int rccl_bcast(void* user, void* buf, size_t bytes, int root, void* stream) {
  return g_rccl.Broadcast(buf, buf, bytes, 0, root, user, reinterpret_cast<hipStream_t>(stream));
}
int rccl_allreduce(void* user, void* buf, size_t count, int kind, void* stream) {
  return g_rccl.AllReduce(buf, buf, count, kind == GPP_COMM_MAX_I32 ? 2 : 8, kind == GPP_COMM_MAX_I32 ? 2 : 0, user,
                          reinterpret_cast<hipStream_t>(stream));
}

int64_t owned_blocks(int64_t nblk, int rank, int nranks) { return rank < nblk ? (nblk - rank + nranks - 1) / nranks : 0; }

}  // namespace

extern "C" {

/* (debug, tools/replay_rank.py) rows x cols doubles (cols even, both pointers 16-byte aligned, ld even) from src to dst on `stream` by
 * `wgs` work-groups of `threads` threads, not before *epoch + not_before ticks of the 100 MHz clock (not_before < 0: at once) and no
 * faster than `gbps` GB/s (<= 0: unthrottled); stamps2 (device, 2 x uint64, zeroed by the caller): start and end relative to *epoch.
 * With a pacing rate the kernel ends when the LAST byte is due, whatever the copy itself took. */
int gpp_debug_replay_copy(void* stream, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols, int wgs,
                          int threads, double gbps, const void* epoch, long long not_before, void* stamps2) {
  if (rows < 0 || cols < 0 || (cols & 1) || (ldd & 1) || (lds & 1) || rows * (cols / 2) >= ((int64_t)1 << 31)) return -1;
  if (wgs < 1 || threads < 64 || threads > 512 || threads % 64 != 0) return -2;
  hipLaunchKernelGGL(gpp_replay_copy, dim3((unsigned)wgs), dim3((unsigned)threads), 0, reinterpret_cast<hipStream_t>(stream), dst, ldd, src,
                     lds, (int)rows, (int)cols, gbps > 0.0 ? gbps * 10.0 : 0.0, reinterpret_cast<const unsigned long long*>(epoch),
                     not_before, reinterpret_cast<unsigned long long*>(stamps2));
  return rc(hipGetLastError());
}
/* (debug) *slot = the 100 MHz clock now, minus *epoch when given */
int gpp_debug_replay_stamp(void* stream, void* slot, const void* epoch) {
  hipLaunchKernelGGL(gpp_replay_stamp, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<unsigned long long*>(slot),
                     reinterpret_cast<const unsigned long long*>(epoch));
  return rc(hipGetLastError());
}

int gpp_set_comm(gpp_handle_t h, const gpp_comm_t* comm, int rank, int nranks) {
  if (!h) return -1;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -3;
  if (nranks > 1 && (!comm || !comm->bcast || !comm->allreduce)) return -2;
  if (comm) h->comm = *comm;
  else memset(&h->comm, 0, sizeof h->comm);
  h->comm_rank = rank;
  h->comm_nranks = nranks;
  return 0;
}

int gpp_comm_unique_id(void* out128) {
  if (!out128) return -1;
  if (!rccl_open()) return 1;
  return g_rccl.GetUniqueId(out128) == 0 ? 0 : 2;
}

int gpp_comm_init_rccl(gpp_handle_t h, const void* unique_id128, int rank, int nranks) {
  if (!h) return -1;
  if (!unique_id128) return -2;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -3;
  if (!rccl_open()) return 1;
  SH_HIP(hipSetDevice(h->device));
  Uid id;
  memcpy(&id, unique_id128, sizeof id);
  void* c = nullptr;
  if (g_rccl.CommInitRank(&c, nranks, id, rank) != 0 || !c) return 2;
  if (h->rccl_comm) (void)g_rccl.CommDestroy(h->rccl_comm);
  h->rccl_comm = c;
  gpp_comm_t cm;
  cm.user = c;
  cm.bcast = rccl_bcast;
  cm.allreduce = rccl_allreduce;
  return gpp_set_comm(h, &cm, rank, nranks);
}

void gpp_shard_release_comm(gpp_handle_t h) {  // (gpp_destroy)
  if (!h) return;
  if (h->rccl_comm && g_rccl.lib) (void)g_rccl.CommDestroy(h->rccl_comm);
  h->rccl_comm = nullptr;
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  h->comm_stream = nullptr;
  if (h->comm_event) (void)hipEventDestroy(h->comm_event);
  h->comm_event = nullptr;
}

size_t gpp_shard_buffer_doubles(int64_t N, int64_t nb, int rank, int nranks, int which) {
  const int64_t nblk = (N + nb - 1) / nb, ld = (N + 15) / 16 * 16;
  const int64_t wc = std::max<int64_t>(owned_blocks(nblk, rank, nranks), 1) * nb;
  switch (which) {
    /* (+ GPP_TILE: the tile kernels READ — never write — up to the next multiple of 128 columns past N along a row; in the last
       row that is past the matrix: a buffer that ends exactly on a page would fault, as hipMalloc'ed ones of N = 20 000 do) */
    case 0: return (size_t)(N * ld + GPP_TILE); /* A (ld = N rounded up to 16) */
    case 1: return (size_t)(N * wc + GPP_TILE); /* Kc, Lc each (ldc = owned blocks x nb) */
    case 2: return (size_t)(nblk * nb * nb);    /* D */
    case 3: return (size_t)(nb * ld + GPP_TILE); /* each scratch row W0..W2 (ldw = ld) */
    case 4: return (size_t)(nb * (N + 2 * nb)); /* msg */
    default: return 0;
  }
}

int gpp_shard_eval(gpp_handle_t h, int64_t N, int64_t nb, const double* U, int D, const double* w, const double* sf2, const double* tau,
                   const int32_t* grp, int S, int kind, int d_split, double jitter, int dU, int need_grad, const gpp_shard_buffers_t* b,
                   int32_t* info_host) {
  if (!h) return -1;
  if (!b || !info_host) return -16;
  if (N < 2 || nb < 128 || nb % 128 != 0) return -2;
  if (!U || !w || !sf2 || !tau || D < 1 || S < 1) return -4;
  if (!b->A || !b->Kc || !b->Lc || !b->D || !b->W0 || !b->W1 || !b->W2 || !b->msg || !b->z || !b->alpha || !b->r || !b->flat || !b->out3 ||
      !b->info)
    return -16;
  const int P = std::max(h->comm_nranks, 1), me = h->comm_rank;
  const bool travel = P > 1 || (h->comm.bcast && getenv("GPP_SHARDED_FORCE_COLLECTIVES") && atoi(getenv("GPP_SHARDED_FORCE_COLLECTIVES")) != 0);
  if (P > 1 && (!h->comm.bcast || !h->comm.allreduce)) return -1;
  *info_host = 0;
  SH_HIP(hipSetDevice(h->device));
  const int64_t nblk = (N + nb - 1) / nb;
  // (decided from the sizes alone, i.e. the same on every rank, BEFORE anything is enqueued: a rank without a block has no list, and
  //  a rank that leaves while the others enter the first broadcast would hang them)
  if (P > nblk) return GPP_SHARD_UNSUPPORTED;
  auto off = [&](int64_t k) { return std::min(k * nb, N); };
  hipStream_t main = h->stream;
  if (!h->comm_stream) SH_HIP(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  if (!h->comm_event) SH_HIP(hipEventCreateWithFlags(&h->comm_event, hipEventDisableTiming));
  hipStream_t cs = h->comm_stream;
  // ---- build of the owned block rows; the list ---------------------------------------------------------------------------------------
  SH_HIP(hipMemsetAsync(b->info, 0, 2 * sizeof(int32_t), main));
  for (int64_t k = me; k < nblk; k += P)
    SH_TRY(gpp_kernel_build(h, U, N, D, w, sf2, tau, grp, S, jitter, kind, d_split, /* full */ 0, b->A, b->ld, off(k), off(k + 1) - off(k)));
  // (the communication stream is ordered behind the caller's BEFORE the list starts: see gpp_shard_list_begin in gpp.h)
  SH_HIP(hipEventRecord(h->comm_event, main));
  SH_HIP(hipStreamWaitEvent(cs, h->comm_event, 0));
  int used = 0;
  SH_TRY(gpp_shard_list_begin(h, N, nb, me, P, b->A, b->ld, b->Kc, b->Lc, b->ldc, b->D, b->W0, b->W1, b->W2, b->ldw, b->info, 0, &used));
  if (!used) return GPP_SHARD_UNSUPPORTED;
  // (from here to gpp_shard_list_end no early return: an open list would leave the handle unusable)
  int comm_rc = 0, err = 0;
  auto hip_ok = [&](hipError_t e) {
    if (e != hipSuccess && !err) err = rc(e);
    return e == hipSuccess;
  };
  auto api_ok = [&](int r) {
    if (r != 0 && !err) err = r;
    return r == 0;
  };
  if (travel) {
    for (int64_t k = 0; k < nblk && !comm_rc && !err; ++k) {
      const int64_t o = off(k), o1 = off(k + 1), o2 = off(k + 2), nbk = o1 - o;
      const bool own = k % P == me;
      double* Dk = b->D + k * nb * nb;
      // messages of block row k (gpp.h): 0 = the head — columns [o, o2) (diagonal block + the next block's columns), then D[k] —,
      // 1 + g = piece g of the tail: W columns from o2 + g W on
      const int64_t W = gpp_shard_piece_cols() > 0 ? gpp_shard_piece_cols() : N;
      const int nmsg = 1 + (N > o2 ? (int)((N - o2 + W - 1) / W) : 0);
      for (int tail = 0; tail < nmsg && !comm_rc && !err; ++tail) {
        const int64_t c0 = tail ? o2 + (tail - 1) * W : o, wcols = (tail ? std::min(c0 + W, N) : o2) - c0;
        if (wcols <= 0) continue;
        const size_t count = (size_t)(nbk * wcols + (tail ? 0 : nbk * nbk));
        if (own) {
          if (!api_ok(gpp_shard_list_gate(h, cs, tail, (int)k))) break;
          if (!hip_ok(hipMemcpy2DAsync(b->msg, wcols * 8, b->A + o * b->ld + c0, b->ld * 8, wcols * 8, nbk, hipMemcpyDeviceToDevice, cs))) break;
          if (!tail && !hip_ok(hipMemcpy2DAsync(b->msg + nbk * wcols, nbk * 8, Dk, nb * 8, nbk * 8, nbk, hipMemcpyDeviceToDevice, cs))) break;
        }
        comm_rc = h->comm.bcast(h->comm.user, b->msg, count * sizeof(double), (int)(k % P), cs);
        if (comm_rc) break;
        if (!own) {
          if (!hip_ok(hipMemcpy2DAsync(b->A + o * b->ld + c0, b->ld * 8, b->msg, wcols * 8, wcols * 8, nbk, hipMemcpyDeviceToDevice, cs))) break;
          if (!tail && !hip_ok(hipMemcpy2DAsync(Dk, nb * 8, b->msg + nbk * wcols, nbk * 8, nbk * 8, nbk, hipMemcpyDeviceToDevice, cs))) break;
          if (!api_ok(gpp_shard_list_signal(h, cs, tail, (int)k))) break;
        }
      }
    }
  }
  SH_TRY(gpp_shard_list_end(h));  // (a list whose messages stopped runs into its time-out: the status says so)
  if (err) return err;
  if (comm_rc) return 3000 + comm_rc;
  SH_HIP(hipEventRecord(h->comm_event, cs));
  SH_HIP(hipStreamWaitEvent(main, h->comm_event, 0));
  // ---- behind the list: the mirror L = U^T into A's strict lower triangle, the owned diagonal blocks of L^-1 into Kc --------------------
  for (int64_t k = 0; k + 1 < nblk; ++k)
    SH_TRY(gpp_transpose(h, b->A + off(k) * b->ld + off(k + 1), b->ld, off(k + 1) - off(k), N - off(k + 1), b->A + off(k + 1) * b->ld + off(k),
                         b->ld));
  for (int64_t c = me, q = 0; c < nblk; c += P, ++q) {
    const int n = (int)(off(c + 1) - off(c));
    hipLaunchKernelGGL(gpp_copy_lower, dim3((n + 255) / 256, n), dim3(256), 0, main, b->D + c * nb * nb, nb, b->Kc + off(c) * b->ldc + q * nb,
                       b->ldc, n);
  }
  SH_HIP(hipGetLastError());
  // ---- the status, agreed on by all ranks (LAPACK-style info, or a time-out's bits) --------------------------------------------------------
  if (travel) {
    if (int r = h->comm.allreduce(h->comm.user, b->info, 1, GPP_COMM_MAX_I32, main)) return 3000 + r;
  }
  SH_HIP(hipMemcpyAsync(info_host, b->info, sizeof(int32_t), hipMemcpyDeviceToHost, main));
  SH_HIP(hipStreamSynchronize(main));
  if (*info_host != 0) return 0;  // (not positive definite at this jitter, or a time-out: the caller decides)
  // ---- z = L^-1 r, the scalars, alpha = L^-T z ----------------------------------------------------------------------------------------
  SH_TRY(gpp_trmv_lower_cols(h, b->Kc, b->ldc, N, b->r, b->z, nb, me, P, 0, 1));
  if (travel) {
    if (int r = h->comm.allreduce(h->comm.user, b->z, (size_t)N, GPP_COMM_SUM_F64, main)) return 3000 + r;
  }
  SH_TRY(gpp_mll_scalars(h, b->A, b->ld, N, b->z, b->out3));
  if (!need_grad) return 0;
  SH_TRY(gpp_trmv_lower_cols(h, b->Kc, b->ldc, N, b->z, b->alpha, nb, me, P, 1, 1));
  if (travel) {
    if (int r = h->comm.allreduce(h->comm.user, b->alpha, (size_t)N, GPP_COMM_SUM_F64, main)) return 3000 + r;
  }
  // ---- the owned column blocks of Ky^-1 by back-substitution (one list, nothing on the wire) ------------------------------------------------
  for (int64_t c = me, q = 0; c < nblk; c += P, ++q)  // (the diagonal blocks are written as lower triangles: clear the sums above them)
    SH_HIP(hipMemset2DAsync(b->Lc + off(c) * b->ldc + q * nb, b->ldc * 8, 0, (off(c + 1) - off(c)) * 8, off(c + 1) - off(c), main));
  SH_TRY(gpp_shard_back_list(h, N, nb, me, P, b->A, b->ld, b->Kc, b->Lc, b->ldc, b->D, b->info, 0, &used));
  if (!used) return GPP_SHARD_UNSUPPORTED;
  // ---- gradient: partial sums over the owned column blocks, one all-reduce -------------------------------------------------------------------
  const size_t nflat = (size_t)(D + 1 + S) + (dU > 0 ? (size_t)N * dU : 0);
  SH_HIP(hipMemsetAsync(b->flat, 0, nflat * sizeof(double), main));
  SH_TRY(gpp_grad_reduce_cols(h, U, N, D, w, sf2, grp, S, kind, d_split, b->alpha, b->Lc, b->ldc, dU, nb, me, P, b->flat, b->flat + D,
                              b->flat + D + 1, dU > 0 ? b->flat + D + 1 + S : nullptr, 1));
  if (travel) {
    if (int r = h->comm.allreduce(h->comm.user, b->flat, nflat, GPP_COMM_SUM_F64, main)) return 3000 + r;
    // (a time-out inside ONE rank's back-substitution list must be every rank's status: the caller's reaction — switch the executor
    //  off, evaluate again — involves collectives, so all ranks have to take it together)
    if (int r = h->comm.allreduce(h->comm.user, b->info, 1, GPP_COMM_MAX_I32, main)) return 3000 + r;
  }
  SH_HIP(hipMemcpyAsync(info_host, b->info, sizeof(int32_t), hipMemcpyDeviceToHost, main));  // (a time-out inside the back-substitution)
  SH_HIP(hipStreamSynchronize(main));
  return 0;
}

}  // extern "C"
