// gpp_push.hip — a direct one-to-all PUSH of the sharded evaluation's messages (round 6; SURVEY.md:204, :423-425: "owner pushes the
// same panel on all 7 links concurrently"), the alternative to a broadcast collective.  Opt-in (GPP_SHARD_PUSH=1 in
// gp-plus_amd/sharded.py); RCCL's broadcast stays the default until the two have been measured on a multi-GPU node.
//
// Every rank owns two message SLOTS (ordinary device memory) and one FLAG page (uncached device memory), both exported with
// hipIpcGetMemHandle and mapped by every other rank.  Messages are numbered 1, 2, ... in the order every rank moves them (the order
// of gp-plus_amd/sharded.py's communication stream); message s uses slot s & 1.
//   owner of s:     on one stream PER PEER, behind the caller's gate: wait until that peer has consumed message s - 2 (the slot's
//                   previous tenant: acks[peer] in the OWNER's flag page) -> hipMemcpy2DAsync of each part straight from where it
//                   lies (the strided block row of the factor: no packing) into the peer's slot -> arrived[s & 1] = s in the PEER's
//                   flag page.  P - 1 streams = P - 1 links busy at once, one hop, no collective kernel.
//   receiver of s:  on its communication stream: wait until arrived[s & 1] == s in its OWN flag page -> the parts from its slot
//                   into place (local 2-D copies) -> (the caller's signal) -> acks[me] = s in EVERY peer's flag page.
// Coherence on real hardware: payload and flags are separate allocations; a pusher's flag store is a system-scope release issued by a
// kernel that starts after the copies into that peer have completed; the receiver's wait is a system-scope acquire in a kernel of
// its own, and the copies that read the slot are LATER kernels (a kernel boundary invalidates what the receiver's L2s may still
// hold of the slot's previous tenant).  The payload never goes straight into the factor: the persistent executor reads the factor
// through its L2 without a kernel boundary, and a peer's stores over xGMI do not pass through that L2.
// Every wait is bounded (GPP_SHARD_TIMEOUT_MS, as the lists' waits): on expiry the kernel ORs `code` into the caller's status word
// and returns, so a lost peer ends in an error, not in a hung GPU.
// Same-device IPC (the ranks of the tests share one GPU) exercises the protocol, not the links: unmeasurable on a one-GPU box.
#include "../../include/gpp.h"
#include "gpp_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline int prc(hipError_t e) { return e == hipSuccess ? 0 : 1000 + (int)e; }
#define PU_HIP(expr)                       \
  do {                                     \
    hipError_t _e = (expr);                \
    if (_e != hipSuccess) return prc(_e);  \
  } while (0)

constexpr int FLAG_ARRIVED = 0;  // [2]: the number of the message that is complete in slot 0 / 1 of this rank
constexpr int FLAG_ACKS = 8;     // [nranks]: acks[q] = the last message rank q has consumed (written by q into every page)
constexpr size_t FLAG_BYTES = 4096;

__global__ void gpp_push_wait_ge(const long long* flag, long long want, long long budget_ticks, int* status, int code) {
  const long long t0 = (long long)wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    __builtin_amdgcn_s_sleep(8);
    if ((long long)wall_clock64() - t0 > budget_ticks) {
      if (status) atomicOr(status, code);
      return;
    }
  }
}

__global__ void gpp_push_store(long long* flag, long long v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// acks[me] = v in every peer's flag page (thread q: peer q)
__global__ void gpp_push_ack_all(long long* const* pages, int me, int nranks, long long v) {
  const int q = threadIdx.x;
  if (q < nranks && q != me && pages[q]) __hip_atomic_store(pages[q] + FLAG_ACKS + me, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct gpp_push {
  int device = 0, rank = 0, nranks = 1;
  int64_t slot_bytes = 0;
  char* slots = nullptr;         // 2 x slot_bytes, this rank's
  long long* flags = nullptr;    // this rank's flag page
  int flag_kind = 0;             // 0 uncached, 1 fine-grained, 2 ordinary
  std::vector<char*> peer_slots;
  std::vector<long long*> peer_flags;
  long long** peer_flags_dev = nullptr;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> done;
  hipEvent_t gate = nullptr;
  long long budget_ticks = 0;
  bool connected = false;
};

extern "C" {

int gpp_push_create(int device, int rank, int nranks, int64_t slot_bytes, gpp_push_t* out, void* handles_out) {
  if (!out || !handles_out) return -5;
  if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return -2;
  if (slot_bytes <= 0) return -4;
  static_assert(2 * sizeof(hipIpcMemHandle_t) <= GPP_PUSH_HANDLE_BYTES, "handle record too small");
  PU_HIP(hipSetDevice(device));
  gpp_push* p = new gpp_push();
  p->device = device; p->rank = rank; p->nranks = nranks;
  p->slot_bytes = (slot_bytes + 255) / 256 * 256;
  const char* ms = getenv("GPP_SHARD_TIMEOUT_MS");
  const long long ms_v = ms && atoll(ms) > 0 ? atoll(ms) : 60000;
  p->budget_ticks = ms_v * 100000LL;  // wall_clock64: 100 MHz
  hipError_t e = hipMalloc((void**)&p->slots, 2 * (size_t)p->slot_bytes);
  if (e != hipSuccess) { delete p; return prc(e); }
  e = hipExtMallocWithFlags((void**)&p->flags, FLAG_BYTES, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    p->flag_kind = 1;
    e = hipExtMallocWithFlags((void**)&p->flags, FLAG_BYTES, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    p->flag_kind = 2;
    e = hipMalloc((void**)&p->flags, FLAG_BYTES);
  }
  if (e != hipSuccess) { (void)hipFree(p->slots); delete p; return prc(e); }
  e = hipMemset(p->flags, 0, FLAG_BYTES);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t hd[2];
  if (e == hipSuccess) e = hipIpcGetMemHandle(&hd[0], p->slots);
  if (e == hipSuccess) e = hipIpcGetMemHandle(&hd[1], p->flags);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->gate, hipEventDisableTiming);
  if (e != hipSuccess) { (void)hipFree(p->slots); (void)hipFree(p->flags); delete p; return prc(e); }
  memset(handles_out, 0, GPP_PUSH_HANDLE_BYTES);
  memcpy(handles_out, hd, sizeof(hd));
  p->peer_slots.assign(nranks, nullptr);
  p->peer_flags.assign(nranks, nullptr);
  p->streams.assign(nranks, nullptr);
  p->done.assign(nranks, nullptr);
  *out = p;
  return 0;
}

int gpp_push_connect(gpp_push_t p, const void* handles) {
  if (!p || !handles) return -1;
  if (p->connected) return -1;
  PU_HIP(hipSetDevice(p->device));
  for (int q = 0; q < p->nranks; ++q) {
    if (q == p->rank) continue;
    hipIpcMemHandle_t hd[2];
    memcpy(hd, (const char*)handles + (size_t)q * GPP_PUSH_HANDLE_BYTES, sizeof(hd));
    void *s = nullptr, *f = nullptr;
    PU_HIP(hipIpcOpenMemHandle(&s, hd[0], hipIpcMemLazyEnablePeerAccess));
    PU_HIP(hipIpcOpenMemHandle(&f, hd[1], hipIpcMemLazyEnablePeerAccess));
    p->peer_slots[q] = (char*)s;
    p->peer_flags[q] = (long long*)f;
    PU_HIP(hipStreamCreateWithFlags(&p->streams[q], hipStreamNonBlocking));
    PU_HIP(hipEventCreateWithFlags(&p->done[q], hipEventDisableTiming));
  }
  PU_HIP(hipMalloc((void**)&p->peer_flags_dev, sizeof(long long*) * (size_t)p->nranks));
  PU_HIP(hipMemcpy(p->peer_flags_dev, p->peer_flags.data(), sizeof(long long*) * (size_t)p->nranks, hipMemcpyHostToDevice));
  p->connected = true;
  return 0;
}

int gpp_push_send(gpp_push_t p, void* after_stream, int64_t seq, int32_t* status, int code, int nparts, const void* const* src,
                  const int64_t* spitch, const int64_t* offset, const int64_t* dpitch, const int64_t* width, const int64_t* height) {
  if (!p || !p->connected) return -1;
  if (seq < 1) return -3;
  if (nparts < 0 || (nparts > 0 && (!src || !spitch || !offset || !dpitch || !width || !height))) return -6;
  for (int i = 0; i < nparts; ++i)
    if (width[i] < 0 || height[i] < 0 || offset[i] < 0 || dpitch[i] < width[i] || spitch[i] < width[i] ||
        (height[i] > 0 && offset[i] + (height[i] - 1) * dpitch[i] + width[i] > p->slot_bytes))
      return -7;
  hipStream_t after = (hipStream_t)after_stream;
  const size_t slot_off = (size_t)(seq & 1) * (size_t)p->slot_bytes;
  PU_HIP(hipEventRecord(p->gate, after));
  for (int q = 0; q < p->nranks; ++q) {
    if (q == p->rank) continue;
    hipStream_t s = p->streams[q];
    PU_HIP(hipStreamWaitEvent(s, p->gate, 0));
    // the slot's previous tenant (message seq - 2) has been unpacked by this peer
    hipLaunchKernelGGL(gpp_push_wait_ge, dim3(1), dim3(1), 0, s, p->flags + FLAG_ACKS + q, (long long)seq - 2, p->budget_ticks, status, code);
    for (int i = 0; i < nparts; ++i)
      if (width[i] > 0 && height[i] > 0)
        PU_HIP(hipMemcpy2DAsync(p->peer_slots[q] + slot_off + offset[i], (size_t)dpitch[i], src[i], (size_t)spitch[i], (size_t)width[i],
                                (size_t)height[i], hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(gpp_push_store, dim3(1), dim3(1), 0, s, p->peer_flags[q] + FLAG_ARRIVED + (int)(seq & 1), (long long)seq);
    PU_HIP(hipEventRecord(p->done[q], s));
    // the caller's stream goes on when the message has left: what it does next may overwrite the parts' sources (a scratch row)
    PU_HIP(hipStreamWaitEvent(after, p->done[q], 0));
  }
  PU_HIP(hipGetLastError());
  return 0;
}

int gpp_push_recv(gpp_push_t p, void* stream, int64_t seq, int32_t* status, int code, int nparts, void* const* dst, const int64_t* dpitch,
                  const int64_t* offset, const int64_t* spitch, const int64_t* width, const int64_t* height) {
  if (!p || !p->connected) return -1;
  if (seq < 1) return -3;
  if (nparts < 0 || (nparts > 0 && (!dst || !dpitch || !offset || !spitch || !width || !height))) return -6;
  for (int i = 0; i < nparts; ++i)
    if (width[i] < 0 || height[i] < 0 || offset[i] < 0 || dpitch[i] < width[i] || spitch[i] < width[i] ||
        (height[i] > 0 && offset[i] + (height[i] - 1) * spitch[i] + width[i] > p->slot_bytes))
      return -7;
  hipStream_t s = (hipStream_t)stream;
  const char* slot = p->slots + (size_t)(seq & 1) * (size_t)p->slot_bytes;
  hipLaunchKernelGGL(gpp_push_wait_ge, dim3(1), dim3(1), 0, s, p->flags + FLAG_ARRIVED + (int)(seq & 1), (long long)seq, p->budget_ticks, status, code);
  for (int i = 0; i < nparts; ++i)
    if (width[i] > 0 && height[i] > 0)
      PU_HIP(hipMemcpy2DAsync(dst[i], (size_t)dpitch[i], slot + offset[i], (size_t)spitch[i], (size_t)width[i], (size_t)height[i],
                              hipMemcpyDeviceToDevice, s));
  PU_HIP(hipGetLastError());
  return 0;
}

int gpp_push_ack(gpp_push_t p, void* stream, int64_t seq) {
  if (!p || !p->connected) return -1;
  if (seq < 1) return -3;
  if (p->nranks > 1)
    hipLaunchKernelGGL(gpp_push_ack_all, dim3(1), dim3(64), 0, (hipStream_t)stream, p->peer_flags_dev, p->rank, p->nranks, (long long)seq);
  PU_HIP(hipGetLastError());
  return 0;
}

int gpp_push_info(gpp_push_t p, int64_t* slot_bytes, int* flag_kind) {
  if (!p) return -1;
  if (slot_bytes) *slot_bytes = p->slot_bytes;
  if (flag_kind) *flag_kind = p->flag_kind;
  return 0;
}

/* stage 0: drain this rank's streams and unmap the peers' memory; stage 1 (after EVERY rank has done stage 0 — the caller's barrier):
 * free this rank's own.  One call with stage 2 does both (one rank, or a process that is going away anyway). */
int gpp_push_destroy(gpp_push_t p, int stage) {
  if (!p) return -1;
  (void)hipSetDevice(p->device);
  if (stage == 0 || stage == 2) {
    (void)hipDeviceSynchronize();
    for (int q = 0; q < p->nranks; ++q) {
      if (p->peer_slots[q]) (void)hipIpcCloseMemHandle(p->peer_slots[q]);
      if (p->peer_flags[q]) (void)hipIpcCloseMemHandle(p->peer_flags[q]);
      if (p->streams[q]) (void)hipStreamDestroy(p->streams[q]);
      if (p->done[q]) (void)hipEventDestroy(p->done[q]);
      p->peer_slots[q] = nullptr; p->peer_flags[q] = nullptr; p->streams[q] = nullptr; p->done[q] = nullptr;
    }
    if (p->peer_flags_dev) (void)hipFree(p->peer_flags_dev);
    p->peer_flags_dev = nullptr;
    p->connected = false;
  }
  if (stage == 1 || stage == 2) {
    if (p->gate) (void)hipEventDestroy(p->gate);
    if (p->slots) (void)hipFree(p->slots);
    if (p->flags) (void)hipFree(p->flags);
    delete p;
  }
  return 0;
}

}  // extern "C"
