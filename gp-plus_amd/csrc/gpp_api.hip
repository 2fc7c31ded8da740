// gpp_api.hip — C ABI of libgpp_hip.so (declared in include/gpp.h) and the host-side orchestration of the blocked
// dense algorithms.  No reference code corresponds to this file: the reference delegates the whole path to
// gpytorch/ATen (optim/mll_torch.py:112-117); the algorithms here are the MI355X-native replacement.
//
// Storage (see gpp.h): the factored matrix keeps its UPPER triangle (A = U^T U, U = L^T), the inverse factor buffer
// keeps L^-1 lower and its mirror L^-T upper.  With that mix EVERY product below is a "TN" GEMM (both operands
// row-contiguous along the non-contracted index), the fastest variant of the kernel.
// Cholesky (gpp_potrf): recursive upper factorisation
//     potrf(A)  = potrf(A11); U12 <- U11^-T A12 (trsm); A22 -= U12^T U12 (syrk); potrf(A22)
//     trsm(B,U) = trsm(B1,U11); B2 -= U12^T X1 (gemm); trsm(B2,U22)        -- splits at multiples of 128
// so that every flop above the 128x128 leaves is a call of the MFMA GEMM kernel with a large K; the leaves (factor +
// inverse of a 128 block in LDS) also leave inv(L_bb) (mirrored) on the diagonal of Linv, and the leaf trsm is the
// in-place TN product  U12 <- (inv(L_bb)^T)^T A12.
// Inverse (gpp_trtri): bottom-up pair merging, for s = 128, 256, ...:
//     Linv21 = -Linv22 (U12^T Linv11) = -(W22)^T (U12^T Linv11),  W = Linv^T (the mirror), batched over all pairs;
//     each Linv21 is also written transposed into the mirror.
// gpp_lauum: Kinv = Linv^T Linv in one lower-triangular TN launch (k >= max(i,j) tile ranges).
#include "../../include/gpp.h"
#include "gpp_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr int NBLK = GPP_TILE;
constexpr int64_t LOOKAHEAD_NB = 1024;  // block-row height of the two-stream outer level (N >= 4 NB)

inline int rc(hipError_t e) { return e == hipSuccess ? 0 : 1000 + (int)e; }
#define GPP_TRY(expr)                   \
  do {                                  \
    hipError_t _e = (expr);             \
    if (_e != hipSuccess) return rc(_e); \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

GemmArgs mk(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int64_t M, int64_t N,
            int64_t K, double alpha, double beta) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.alpha = alpha; g.beta = beta;
  return g;
}

// split point: multiple of 128 closest to n/2 (from above)
inline int64_t split(int64_t n) {
  int64_t h = ((n / 2 + NBLK - 1) / NBLK) * NBLK;
  if (h >= n) h -= NBLK;
  return h;
}

struct Ctx {
  hipStream_t s;
  double* A; int64_t ld;       // matrix being factored (upper; U on exit)
  double* Li; int64_t ldi;     // Linv buffer (mirrored diagonal leaves written here)
  int32_t* info;
};

// Solve U_oo^T X = B for the n x n upper block U at (o,o); B is n x m at rows o.., columns c0..; in place.
hipError_t trsm_rec(const Ctx& c, int64_t c0, int64_t m, int64_t o, int64_t n) {
  if (m <= 0 || n <= 0) return hipSuccess;
  if (n <= NBLK) {
    // X[k][j] = sum_i W[i][k] B[i][j] with W = inv(L_leaf)^T = upper part of the mirrored leaf block (keep i <= k).
    // In place: one row tile (n <= 128), every work-group reads and writes only its own column block.
    double* Bp = c.A + o * c.ld + c0;
    GemmArgs g = mk(c.Li + o * c.ldi + o, c.ldi, Bp, c.ld, Bp, c.ld, n, m, n, 1.0, 0.0);
    g.a_mask = 1;
    // all 128 rows in ONE row tile (in place); narrow column tiles so a short panel still covers the chip
    return gpp_launch_gemm(c.s, 2, g, 1, NBLK, (m + NBLK - 1) / NBLK >= 512 ? NBLK : 32);
  }
  const int64_t n1 = split(n), n2 = n - n1;
  hipError_t e = trsm_rec(c, c0, m, o, n1);
  if (e != hipSuccess) return e;
  // B2 -= U12^T X1 : U12 = A[o.., o+n1..] (n1 x n2), X1 = A[o.., c0..] (n1 x m), B2 = A[o+n1.., c0..] (n2 x m)
  GemmArgs g = mk(c.A + o * c.ld + (o + n1), c.ld, c.A + o * c.ld + c0, c.ld, c.A + (o + n1) * c.ld + c0, c.ld, n2, m, n1,
                  -1.0, 1.0);
  e = gpp_launch_gemm(c.s, 2, g, 1);
  if (e != hipSuccess) return e;
  return trsm_rec(c, c0, m, o + n1, n2);
}

// Right-looking factorisation of a block of up to BLK_MAX rows (every N below the look-ahead threshold, and the look-ahead's diagonal blocks) in steps of one leaf: leaf, ONE in-place panel solve over
// all remaining columns, ONE rank-128 update of the remaining upper triangle (which stays in L2 at this size).  3 launches
// per 128 rows where the recursion needs 4, and none of them narrower than the block: 0.70 -> 0.57 ms for 1024 rows, 3.4 -> 3.0 ms for 3968.
constexpr int64_t BLK_MAX = 6144;
hipError_t potrf_blk(const Ctx& c, int64_t o, int64_t n) {
  for (int64_t j0 = 0; j0 < n; j0 += NBLK) {
    const int64_t nb = std::min<int64_t>(NBLK, n - j0), oo = o + j0, rem = n - j0 - nb;
    hipError_t e = gpp_launch_leaf(c.s, c.A + oo * c.ld + oo, c.ld, c.Li + oo * c.ldi + oo, c.ldi, (int)nb, c.info, (int)oo);
    if (e != hipSuccess) return e;
    if (rem == 0) break;
    double* Bp = c.A + oo * c.ld + (oo + nb);  // block row oo, columns to the right of the leaf
    GemmArgs g = mk(c.Li + oo * c.ldi + oo, c.ldi, Bp, c.ld, Bp, c.ld, nb, rem, nb, 1.0, 0.0);
    g.a_mask = 1;
    e = gpp_launch_gemm(c.s, 2, g, 1, NBLK, 32);
    if (e != hipSuccess) return e;
    GemmArgs u = mk(Bp, c.ld, Bp, c.ld, c.A + (oo + nb) * c.ld + (oo + nb), c.ld, rem, rem, nb, -1.0, 1.0);
    u.c_lower = 2;
    static const int ut = getenv("GPP_BLK_UPD_TILE") ? atoi(getenv("GPP_BLK_UPD_TILE")) : 32;  // K = 128: small tiles win (measured)
    e = gpp_launch_gemm(c.s, 2, u, 1, ut, ut);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// The same leaf-step factorisation for `batch` independent matrices at A + b*sA (inverse leaves at Li + b*sLi, info + b):
// every launch covers the whole batch, so 64 restarts of a 500-point model cost the latency chain of ONE.
hipError_t potrf_blk_batched(const Ctx& c, int64_t n, int batch, int64_t sA, int64_t sLi) {
  for (int64_t j0 = 0; j0 < n; j0 += NBLK) {
    const int64_t nb = std::min<int64_t>(NBLK, n - j0), oo = j0, rem = n - j0 - nb;
    hipError_t e = gpp_launch_leaf(c.s, c.A + oo * c.ld + oo, c.ld, c.Li + oo * c.ldi + oo, c.ldi, (int)nb, c.info, (int)oo,
                                   batch, sA, sLi);
    if (e != hipSuccess) return e;
    if (rem == 0) break;
    double* Bp = c.A + oo * c.ld + (oo + nb);
    GemmArgs g = mk(c.Li + oo * c.ldi + oo, c.ldi, Bp, c.ld, Bp, c.ld, nb, rem, nb, 1.0, 0.0);
    g.a_mask = 1;
    g.sA = sLi; g.sB = sA; g.sC = sA;
    e = gpp_launch_gemm(c.s, 2, g, batch, NBLK, 32);
    if (e != hipSuccess) return e;
    GemmArgs u = mk(Bp, c.ld, Bp, c.ld, c.A + (oo + nb) * c.ld + (oo + nb), c.ld, rem, rem, nb, -1.0, 1.0);
    u.c_lower = 2;
    u.sA = sA; u.sB = sA; u.sC = sA;
    e = gpp_launch_gemm(c.s, 2, u, batch, 32, 32);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t potrf_rec(const Ctx& c, int64_t o, int64_t n) {
  if (n <= 0) return hipSuccess;
  if (n <= NBLK)
    return gpp_launch_leaf(c.s, c.A + o * c.ld + o, c.ld, c.Li + o * c.ldi + o, c.ldi, (int)n, c.info, (int)o);
  static const int64_t blk_max = getenv("GPP_BLK_MAX") ? atol(getenv("GPP_BLK_MAX")) : BLK_MAX;  // experiment knob
  if (n <= blk_max) return potrf_blk(c, o, n);
  const int64_t n1 = split(n), n2 = n - n1;
  hipError_t e = potrf_rec(c, o, n1);
  if (e != hipSuccess) return e;
  e = trsm_rec(c, o + n1, n2, o, n1);
  if (e != hipSuccess) return e;
  // A22 -= U12^T U12 (upper triangle only)
  const double* U12 = c.A + o * c.ld + (o + n1);
  GemmArgs g = mk(U12, c.ld, U12, c.ld, c.A + (o + n1) * c.ld + (o + n1), c.ld, n2, n2, n1, -1.0, 1.0);
  g.c_lower = 2;
  e = gpp_launch_gemm(c.s, 2, g, 1);
  if (e != hipSuccess) return e;
  return potrf_rec(c, o + n1, n2);
}

// Cooperative panel (gpp_leaf.hip): the block [o, o+n) factored AND inverted by one launch — used for the look-ahead's diagonal
// blocks and for a whole small matrix.  GPP_COOP_PANEL=0 restores the chain of leaf-step launches + pair merges (experiment knob).
constexpr int64_t PANEL_MAX_N = 2048;
inline bool panel_enabled() {
  static const bool on = !(getenv("GPP_COOP_PANEL") && atoi(getenv("GPP_COOP_PANEL")) == 0);
  return on;
}
inline bool panel_fits(const gpp_handle_s* h, int64_t n) {
  static const int64_t nmax = getenv("GPP_PANEL_MAX_N") ? atol(getenv("GPP_PANEL_MAX_N")) : PANEL_MAX_N;  // experiment knob
  return h->coop_panel && h->panel_flags && h->ncu >= 2 && n > 2 * NBLK && n <= nmax && (n + NBLK - 1) / NBLK <= gpp_panel_max_leaves();
}
// The cooperative panel of an n x n diagonal block whose own addresses are c.A / c.Li (o: its first row, for the status word).
hipError_t launch_panel_at(gpp_handle_s* h, const Ctx& c, int64_t o, int64_t n, int max_wgs) {
  if (h->panel_fault) {  // test hook: behave like a launch whose wait timed out
    h->panel_fault = 0;
    return gpp_launch_fill_i32(c.s, c.info, 1, GPP_INFO_PANEL_TIMEOUT);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c.s, &cap) != hipSuccess) {
    (void)hipGetLastError();
    cap = hipStreamCaptureStatusNone;
  }
  int slot;
  if (cap == hipStreamCaptureStatusActive) slot = GPP_PANEL_RING + (h->cap_next++ % GPP_PANEL_CAP_RING);  // (see gpp_internal.h)
  else slot = h->panel_next++ % GPP_PANEL_RING;
  int* fl = reinterpret_cast<int*>(h->panel_flags + (size_t)slot * gpp_panel_flag_bytes());
  return gpp_launch_panel(c.s, c.A, c.ld, c.Li, c.ldi, (int)n, c.info, (int)o, fl, max_wgs, h->panel_timeout_ms);
}

// ... of the block at (o, o) of c.A, its inverse into the same place of c.Li
hipError_t launch_panel(gpp_handle_s* h, const Ctx& c, int64_t o, int64_t n, int max_wgs) {
  Ctx b = c;
  b.A = c.A + o * c.ld + o;
  b.Li = c.Li + o * c.ldi + o;
  return launch_panel_at(h, b, o, n, max_wgs);
}

// ---- triangular inverse by pair merging -------------------------------------------------------------------------
// One level: for the pairs of s-blocks [b, b+s), [b+s, min(b+2s, base+n)) of the range [base, base+n):
//     T21 = U12^T Linv11 ;  Linv21 = -W22^T T21 (+ mirror).  ``skip(b)`` drops pairs that are already merged.
template <typename Skip>
hipError_t trtri_level(hipStream_t st, const double* U, int64_t ld, double* Linv, int64_t ldi, double* T, int64_t ldt,
                       int64_t base, int64_t n, int64_t s, Skip skip, int mbatch = 1, int64_t msU = 0, int64_t msLi = 0,
                       int64_t msT = 0, int g1_tile = 0, hipEvent_t before_g2 = nullptr) {
  const int64_t npairs_full = n / (2 * s);
  const int64_t rem = n - npairs_full * 2 * s;  // leftover rows after the full pairs
  auto launch = [&](int64_t o, int64_t m2, int batch) -> hipError_t {
    const int64_t pstride_U = 2 * s * (ld + 1), pstride_I = 2 * s * (ldi + 1), pstride_T = 2 * s * (ldt + 1);
    // T21 = U12^T * Linv11   (TN; U12 = U[o.., o+s..] is s x m2, Linv11 lower: keep k >= n)
    GemmArgs g1 = mk(U + o * ld + (o + s), ld, Linv + o * ldi + o, ldi, T + (o + s) * ldt + o, ldt, m2, s, s, 1.0, 0.0);
    g1.b_mask = 2; g1.klo_mode = 2; g1.col_major = 1;  // K range depends on the column tile: keep columns together
    g1.sA = pstride_U; g1.sB = pstride_I; g1.sC = pstride_T;
    g1.batch2 = mbatch; g1.zA = msU; g1.zB = msLi; g1.zC = msT;  // independent matrices (batched evaluation)
    static const bool bf = !(getenv("GPP_BATCH_FAST") && atoi(getenv("GPP_BATCH_FAST")) == 0);  // experiment knob
    g1.batch_fast = bf;  // the pairs' tiles of equal K run together: the launch ends on every pair's short tiles
    hipError_t e = gpp_launch_gemm(st, 2, g1, batch, g1_tile, g1_tile);
    if (e != hipSuccess) return e;
    if (before_g2) {  // W22 comes from another stream (the look-ahead's bordering: the first product does not need it)
      e = hipStreamWaitEvent(st, before_g2, 0);
      if (e != hipSuccess) return e;
    }
    // Linv21 = -W22^T * T21  (TN; W22 = mirrored upper part of the (o+s) block: keep k <= m), plus its mirror
    GemmArgs g2 = mk(Linv + (o + s) * ldi + (o + s), ldi, T + (o + s) * ldt + o, ldt, Linv + (o + s) * ldi + o, ldi, m2, s,
                     m2, -1.0, 0.0);
    g2.a_mask = 1; g2.khi_mode = 1; g2.row_reverse = 1;
    g2.sA = pstride_I; g2.sB = pstride_T; g2.sC = pstride_I;
    g2.C2 = Linv + o * ldi + (o + s); g2.ldc2 = ldi; g2.sC2 = pstride_I;
    g2.batch2 = mbatch; g2.zA = msLi; g2.zB = msT; g2.zC = msLi; g2.zC2 = msLi;
    g2.batch_fast = bf;
    return gpp_launch_gemm(st, 2, g2, batch);
  };
  // full pairs: batched over maximal runs of pairs that still need merging
  int64_t p = 0;
  while (p < npairs_full) {
    if (skip(base + 2 * s * p)) { ++p; continue; }
    int64_t q = p;
    while (q < npairs_full && !skip(base + 2 * s * q)) ++q;
    hipError_t e = launch(base + 2 * s * p, s, (int)(q - p));
    if (e != hipSuccess) return e;
    p = q;
  }
  if (rem > s && !skip(base + npairs_full * 2 * s)) {  // ragged last pair (second block shorter than s)
    hipError_t e = launch(base + npairs_full * 2 * s, rem - s, 1);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---- look-ahead (right-looking) driver on two streams ----------------------------------------------------------
// A factorisation made of leaf steps spends most of its time in latency-bound launches (128-block leaves, the leaf trsm,
// rank-128 updates) during which most of the 256 CUs idle.  For N >= 6144 the outer level is therefore right-looking over
// block rows of 1024 / 512 with ONE step of look-ahead on two internal streams:
//   latency stream   : wait S(k-1); U_kk = potrf(A_kk) in leaf steps; invert the block completely; record D(k)
//   throughput stream: wait D(k); U_k,k+1: = inv(U_kk)^T A_k,k+1: (one GEMM); strip (next block row) -= ...; record S(k);
//                      rest of the trailing matrix -= ...
// Stream priorities alone do not help: the leaf needs a CU to itself (its register allocation does not fit beside a GEMM
// work-group), so while an update saturates every CU it is not placed until the update drains (measured: 5.8 ms).  The
// two streams therefore get DISJOINT CU sets through CU masks (see the comment at the mask below for what that costs).
constexpr int PANEL_CUS_DEFAULT = 32;
constexpr int64_t BORDER_MAX_N = 11264;  // largest N whose inverse is built by bordering inside the look-ahead
constexpr int64_t BORDER_MIN_N = 3840;   // with bordering the look-ahead already wins from here (4.16 vs 4.32 ms per evaluation; 3712: 4.10 vs 4.03)
hipError_t ensure_streams(gpp_handle_s* h) {
  if (h->cu_split < 0) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, h->device);
    if (e != hipSuccess) return e;
    const int ncu = prop.multiProcessorCount;
    h->cu_split = 0;
    int PANEL_CUS = PANEL_CUS_DEFAULT;
    if (const char* e = getenv("GPP_PANEL_CUS")) PANEL_CUS = atoi(e);  // experiment knob
    if (ncu >= 4 * PANEL_CUS && ncu <= 1024 && PANEL_CUS > 0 && !getenv("GPP_NO_CU_SPLIT")) {
      uint32_t mp[32] = {0}, mu[32] = {0};
      const int words = (ncu + 31) / 32;
      // The FIRST mask bits.  Measured (tools/attic/gemm_update_probe.py with GPP_GEMM_ON_UPD=1): a K = 1024 trailing update runs
      // at 62.3 TFLOP/s on all 256 CUs and at 54.3 / 54.8 / 54.7 with 2 / 16 / 32 CUs masked out — the loss is a step of
      // 12.5 %, not proportional.  Consecutive mask bits fall in different XCDs and then in different shader engines of an
      // XCD; work-groups are dealt round-robin to XCDs and to their 4 shader engines regardless of the CUs each has left, so
      // throughput follows the SMALLEST engine (7 of 8 CUs) as soon as any CU is taken, and stays there until every engine
      // of every XCD has given one: 32 CUs cost the update stream exactly what 2 do.  (Taking the CUs from one XCD, mask bits
      // 0, 8, 16, ..., is far worse: 59 -> 95 ms.)  Without masks the leaf needs a CU to itself (its register allocation
      // does not fit beside a GEMM work-group) and starves; capped at 256 registers it is placed, but every panel launch
      // then queues behind 300-us GEMM work-groups and the update stream idles 0.55 ms per step: 62 ms vs 59.
      for (int c = 0; c < ncu; ++c) (c < PANEL_CUS ? mp : mu)[c >> 5] |= 1u << (c & 31);
      hipStream_t sp = nullptr, su = nullptr, sf = nullptr;
      if (hipExtStreamCreateWithCUMask(&sp, words, mp) == hipSuccess &&
          hipExtStreamCreateWithCUMask(&su, words, mu) == hipSuccess &&
          hipExtStreamCreateWithCUMask(&sf, words, mu) == hipSuccess) {
        h->panel_stream = sp;
        h->upd_stream = su;
        h->fill_stream = sf;
        h->cu_split = 1;
        h->panel_cus = PANEL_CUS;
      } else {
        (void)hipGetLastError();
        if (sp) (void)hipStreamDestroy(sp);
        if (su) (void)hipStreamDestroy(su);
        if (sf) (void)hipStreamDestroy(sf);
      }
    }
    if (!h->cu_split) {  // fallback: priority streams sharing all CUs
      // NOT silent: without disjoint CU sets a leaf waits for a CU until the co-running update drains (measured: 5.8 ms per
      // diagonal block, 90 instead of ~55 ms per factorisation at N = 20000)
      static bool warned = false;
      if (!warned && !getenv("GPP_NO_CU_SPLIT")) {
        warned = true;
        fprintf(stderr, "libgpp_hip: hipExtStreamCreateWithCUMask unavailable on device %d (%d CUs): the look-ahead factorisation "
                        "falls back to priority streams and will be markedly slower\n", h->device, ncu);
      }
      int lo = 0, hi = 0;
      e = hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (e != hipSuccess) return e;
      e = hipStreamCreateWithPriority(&h->panel_stream, hipStreamNonBlocking, hi);
      if (e != hipSuccess) return e;
      e = hipStreamCreateWithFlags(&h->upd_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return e;
      e = hipStreamCreateWithFlags(&h->fill_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return e;
    }
  }
  if (!h->full_stream) {
    // Experiment knob GPP_FULL_PRIO=1: LOWEST priority for the stream without a CU mask.  The bulk of a trailing update overlaps
    // the start of the panel after next, and its work-groups also run on the panel's CUs; with equal priorities the panel's small
    // launches queue for slots behind the bulk's thousands of tiles (traced: a 28-work-group panel solve took 1.57 ms).  Measured
    // A/B, twice each: potrf 52.26 / 52.07 -> 51.73 / 51.48 ms at N = 20000; no change at 30000.  NOT the default: with the
    // priority stream `rocprofv3 --kernel-trace --stats -- python3 bench.py` never returned on this stack (ROCm 7.2.0; the plain
    // run is fine) — a 0.4 % gain is not worth a benchmark that cannot be profiled.
    static const int prio = getenv("GPP_FULL_PRIO") ? atoi(getenv("GPP_FULL_PRIO")) : 0;
    hipError_t e;
    if (prio) {
      int lo = 0, hi = 0;
      e = hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (e != hipSuccess) return e;
      e = hipStreamCreateWithPriority(&h->full_stream, hipStreamNonBlocking, lo);
    } else {
      e = hipStreamCreateWithFlags(&h->full_stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) return e;
  }
  while (h->n_events < 16) {
    hipError_t e = hipEventCreateWithFlags(&h->events[h->n_events], hipEventDisableTiming);
    if (e != hipSuccess) return e;
    ++h->n_events;
  }
  return hipSuccess;
}
inline hipEvent_t next_event(gpp_handle_s* h) {
  hipEvent_t ev = h->events[h->ev_next];
  h->ev_next = (h->ev_next + 1) % 16;
  return ev;
}
#define HIP_TRY(expr)                \
  do {                               \
    hipError_t _e = (expr);          \
    if (_e != hipSuccess) return _e; \
  } while (0)

// Step k (block row o, height nb) of the look-ahead, stream by stream (events in capitals):
//   panel (32 CUs)    : wait S(k-1); U_oo = potrf(A_oo) in leaf steps; invert the block completely; record D(k)
//   throughput (224)  : wait D(k); solve the columns of the NEXT diagonal block, update that block; record S(k);
//                       solve the rest of block row o (R(k)); first rows of the trailing update
//   unmasked (256)    : wait D(k+1); the bulk of step k's trailing update (N > 11264 only)
//   fill (224, shared): bordering step of the inverse, Linv[o.., :o) (4096 <= N <= 11264 only): first product after
//                       R(k-1), second after D(k)
// The calling stream waits for all of them at the end; nothing else synchronises with the host.
hipError_t potrf_lookahead(gpp_handle_s* h, const Ctx& cm, int64_t N, int64_t NB, double* T, int64_t ldt) {
  HIP_TRY(ensure_streams(h));
  Ctx cp = cm, cu = cm;
  cp.s = h->panel_stream;
  cu.s = h->upd_stream;
  const char* env_nb = getenv("GPP_LOOKAHEAD_NB");  // experiment knob: "big,small,threshold"
  static const int64_t border_max_x = getenv("GPP_BORDER_MAX") ? atol(getenv("GPP_BORDER_MAX")) : BORDER_MAX_N;
  hipEvent_t ev = next_event(h);
  HIP_TRY(hipEventRecord(ev, cm.s));  // inputs (kernel build) are ready
  HIP_TRY(hipStreamWaitEvent(cp.s, ev, 0));
  HIP_TRY(hipStreamWaitEvent(cu.s, ev, 0));
  const int64_t o_begin = 0;
  // Round 1 measured 1024 above / 512 below 6144 remaining rows best (70.6 vs 71.7 ms for 1024 flat at N = 20000); with the panel
  // kernel a 1024-row block costs 0.58 ms where two 512-row blocks cost 2 x (0.29 + 0.07 ms hand-off), and 1024 flat wins (means
  // of 3: potrf 15.50 -> 15.22 ms at N = 12288, 25.27 -> 24.87 at 15000, 51.84 -> 51.46 at 20000, 155.3 -> 155.3 at 30000).
  // (The bordering range chooses its own height below.)
  long nb_big = NB, nb_small = NB, nb_thresh = 0;
  if (env_nb) sscanf(env_nb, "%ld,%ld,%ld", &nb_big, &nb_small, &nb_thresh);
  // bordering pays while the factorisation is bound by its chain of diagonal blocks (measured: 9.6 -> 7.8 ms per
  // evaluation at N = 6144, 16.3 -> 14.0 at 8192, 26.5 -> 24.4 at 10000, a tie at 12288, 147 -> 156 at 20000 where the
  // throughput CUs have no idle time to give and the long-K bordering products are slower than batched pair merges)
  const int64_t border_max = border_max_x;  // knob GPP_BORDER_MAX
  const bool border = T != nullptr && N <= border_max;
  // the first product of a bordering step needs neither this block's factor nor its inverse: it is issued before them
  // and only the second waits for D (measured: 4.54 -> 4.44 ms per evaluation at 4096, 13.7 -> 13.4 at 8192, even above)
  static const bool border_early = !(getenv("GPP_BORDER_EARLY") && atoi(getenv("GPP_BORDER_EARLY")) == 0);  // experiment knob
  // (measured: potrf 58.2 -> 56.4 ms at N = 20000, 28.4 -> 27.3 at 15000, 17.0 -> 16.4 at 10000)
  static const bool merge_upd_on = !(getenv("GPP_MERGE_UPD") && atoi(getenv("GPP_MERGE_UPD")) == 0);  // experiment knob
  static const bool split_chain = !(getenv("GPP_SPLIT_CHAIN") && atoi(getenv("GPP_SPLIT_CHAIN")) == 0);  // experiment knob
  // Round 4: from N = 9216 the inverse's bordering products run on the stream WITHOUT a CU mask, i.e. also on the panel's 32 CUs
  // while those idle (60 % of the time at N = 10 000, profiles/r04_timeline_n10000.txt).  Measured A/B on one box, potrf + inverse:
  // 16.08 -> 15.68 ms at N = 10 000, 21.03 -> 20.54 at 11 264, even at 8192, a LOSS at 6144 (4.99 -> 5.23: the chain matters more
  // there and a long bordering tile on a panel CU holds the next diagonal block up).  GPP_BORDER_FULL_MIN moves the threshold.
  static const int64_t border_full_min = getenv("GPP_BORDER_FULL_MIN") ? atol(getenv("GPP_BORDER_FULL_MIN")) : 9216;
  hipStream_t cf = (T != nullptr && N <= border_max_x && N >= border_full_min) ? h->full_stream : h->fill_stream;
  static const int64_t border_t128 = getenv("GPP_BORDER_T128") ? atol(getenv("GPP_BORDER_T128")) : 640;  // experiment knob
  hipEvent_t R_prev = nullptr;  // row solves of the steps before the current one are complete
  // The CU mask costs the throughput stream 12.5 % (see ensure_streams) although the panel needs its CUs only while it
  // factors the next diagonal block (~1 ms of a 6.7 ms step at N = 20000).  The trailing update is therefore split: the
  // first rows run on the masked stream beside the panel, the bulk waits for the panel (event D) and runs on a stream
  // WITHOUT a mask.  Measured: potrf 56.6 -> 55.5 ms at N = 20000, 173.3 -> 165 ms at 30000; the masked part is sized in
  // entries of the upper triangle (4.5e7 .. 9e7 equal; 2e8 loses the gain, 1.5e7 half of it).
  static const bool split_upd_on = !(getenv("GPP_SPLIT_UPD") && atoi(getenv("GPP_SPLIT_UPD")) == 0);  // experiment knob
  // (round 3, with the panel kernel — the next diagonal block now takes 0.6 instead of ~1 ms: 3e7 against 6e7 entries, twice each:
  //  potrf 15.28 vs 15.76 ms at N = 12288, 51.2-51.5 vs 51.5-51.8 at 20000, 154.0-154.5 vs 156.0-157.6 at 30000, 1139 vs 1144-1147 at 60000)
  static const int64_t split_elems = getenv("GPP_SPLIT_ELEMS") ? atol(getenv("GPP_SPLIT_ELEMS")) : 30000000;
  const bool split_upd = split_upd_on && !border && h->cu_split == 1;
  hipStream_t cx = h->full_stream;
  struct { bool on; GemmArgs g; hipEvent_t rows_ready; int64_t rows_masked; } pend{false, GemmArgs{}, nullptr, 0};
  hipEvent_t be_wait = nullptr;  // end of the previous step's unmasked part, not yet waited for by the throughput stream
  // (measured: potrf 55.3 -> 54.0 ms at N = 20000, 162.8 -> 161.1 at 30000)
  static const bool defer_be = !(getenv("GPP_DEFER_BE") && atoi(getenv("GPP_DEFER_BE")) == 0);  // experiment knob
  // with bordering the block height matters little; one height per N measured best (512 up to ~7000 rows: 7.5 vs 8.3 ms per
  // evaluation at 6144; 1024 above: 23.3 vs 24.2 ms at 10000)
  if (border && !env_nb) {
    nb_small = NB / 2;
    nb_thresh = (N <= 7168) ? N + 1 : 0;
  }
  for (int64_t o = o_begin, nb = 0; o < N; o += nb) {
    // tall block rows while the trailing update is long enough to hide their diagonal factorisation, shorter after
    const int64_t want = (N - o >= nb_thresh) ? nb_big : nb_small;
    nb = std::min(want, N - o);
    const int64_t rem = N - o - nb;
    const bool coop = T != nullptr && panel_fits(h, nb);
    if (coop) {
      // factor + complete inverse of the diagonal block in ONE cooperative launch on the panel's CUs
      HIP_TRY(launch_panel(h, cp, o, nb, h->cu_split == 1 ? h->panel_cus : std::min(64, h->ncu)));
      if (h->inv_nblocks < 128) {
        h->inv_o[h->inv_nblocks] = o;
        h->inv_n[h->inv_nblocks] = nb;
        ++h->inv_nblocks;
      }
    } else {
      HIP_TRY(potrf_rec(cp, o, nb));
    }
    if (T && !coop) {
      // complete inverse of this diagonal block, still on the panel stream (hidden behind the trailing update): it turns
      // the wide trsm below into ONE GEMM and is exactly the low levels of gpp_trtri, which will skip them
      for (int64_t s = NBLK; s < nb; s *= 2)
        HIP_TRY(trtri_level(cp.s, cm.A, cm.ld, cm.Li, cm.ldi, T, ldt, o, nb, s, [](int64_t) { return false; }));
      if (h->inv_nblocks < 128) {
        h->inv_o[h->inv_nblocks] = o;
        h->inv_n[h->inv_nblocks] = nb;
        ++h->inv_nblocks;
      }
    }
    hipEvent_t D = next_event(h);
    HIP_TRY(hipEventRecord(D, cp.s));
    HIP_TRY(hipStreamWaitEvent(cu.s, D, 0));
    if (pend.on) {
      // the bulk of the previous step's trailing update: the diagonal block it ran beside is done (D), so no leaf is
      // waiting for a CU and this part may take all of them (62 instead of 55 TFLOP/s)
      HIP_TRY(hipStreamWaitEvent(cx, pend.rows_ready, 0));
      HIP_TRY(hipStreamWaitEvent(cx, D, 0));
      HIP_TRY(gpp_launch_gemm(cx, 2, pend.g, 1, NBLK, NBLK));
      hipEvent_t BE = next_event(h);
      HIP_TRY(hipEventRecord(BE, cx));
      pend.on = false;
      // This step's chain (next diagonal block's columns and update) and row solve touch only rows the MASKED part of that
      // update produced, when it was tall enough: then they run beside the bulk, and only this step's own trailing update
      // waits for it.
      if (defer_be && pend.rows_masked >= nb + std::min<int64_t>((rem >= nb_thresh) ? nb_big : nb_small, rem)) be_wait = BE;
      else HIP_TRY(hipStreamWaitEvent(cu.s, BE, 0));
    }
    auto border_step = [&]() -> hipError_t {
      // bordering step of the inverse: Linv[o.., 0..o) = -W_oo^T (U[0..o, o..)^T Linv[0..o, 0..o)) — the ragged pair merge
      // of [0, o) with [o, o+nb).  Needs block rows < o of U (row solves of the earlier steps: event R) and this block's
      // inverse (D).  Its work grows as the trailing update shrinks, so the two together keep the throughput CUs busy.
      if (!border_early) HIP_TRY(hipStreamWaitEvent(cf, D, 0));
      HIP_TRY(hipStreamWaitEvent(cf, R_prev, 0));
      // few, long tiles (K up to o): 64-wide tiles balance better until there are several waves of 128-wide ones
      const int64_t t128 = ((nb + 127) / 128) * ((o + 127) / 128);
      return trtri_level(cf, cm.A, cm.ld, cm.Li, cm.ldi, T, ldt, 0, o + nb, o, [](int64_t) { return false; }, 1, 0, 0, 0,
                         t128 < border_t128 ? 64 : 0, border_early ? D : nullptr);
    };
    // Experiment knob GPP_BORDER_RL=1 (round 4, measured and NOT adopted): the bordered inverse RIGHT-looking, as rank-nb updates
    // of running sums kept in the scratch T (what the sharded forward sweep does, gp-plus_amd/sharded.py): with X = L^-1, step k
    //   A(k): X[k, :k) = -X_kk S_k                 S_k = T[o:o+nb, 0:o): sum over j < k of L[k, j] X[j, :]   (K <= nb, + mirror)
    //   B(k): S_m += L[m, k] X[k, :k+1), m > k     ONE product with K = nb over (N - o - nb) x (o + nb) entries
    // instead of the left-looking X[k, :k) = -X_kk (U[:k, k]^T X[:k, :k)) whose K grows to N and whose few long tiles run
    // 64 wide.  Correct (the kernel tests pass with it) and no faster: potrf + inverse 16.46 vs 16.10 ms at N = 10 000, 9.45 vs 9.21
    // at 8192, 5.28 vs 5.07 at 6144, 20.7 vs 21.1 at 11 264 — at these sizes the throughput CUs are the shared bottleneck of the
    // factor's updates and the inverse's products whatever the products' shape (profiles/EXPERIMENTS.md, round 4).
    static const bool border_rl = getenv("GPP_BORDER_RL") && atoi(getenv("GPP_BORDER_RL")) != 0;
    auto border_A = [&]() -> hipError_t {
      HIP_TRY(hipStreamWaitEvent(cf, D, 0));  // this block's inverse (its sums are complete in stream order: B(k-1))
      GemmArgs g2 = mk(cm.Li + o * cm.ldi + o, cm.ldi, T + o * ldt, ldt, cm.Li + o * cm.ldi, cm.ldi, nb, o, nb, -1.0, 0.0);
      g2.a_mask = 1; g2.khi_mode = 1; g2.row_reverse = 1;
      g2.C2 = cm.Li + o; g2.ldc2 = cm.ldi;
      return gpp_launch_gemm(cf, 2, g2, 1);
    };
    auto border_B = [&](hipEvent_t row_solved) -> hipError_t {
      HIP_TRY(hipStreamWaitEvent(cf, row_solved, 0));  // block row o of U is final
      const int64_t o1 = o + nb, M = N - o1;
      if (o > 0) {  // the columns left of this block: X[k, :k) from A(k) above
        GemmArgs b1 = mk(cm.A + o * cm.ld + o1, cm.ld, cm.Li + o * cm.ldi, cm.ldi, T + o1 * ldt, ldt, M, o, nb, 1.0, 1.0);
        HIP_TRY(gpp_launch_gemm(cf, 2, b1, 1));
      }
      // this block's own columns start the sums (beta = 0: the scratch is never cleared): X_kk, lower triangular
      GemmArgs b2 = mk(cm.A + o * cm.ld + o1, cm.ld, cm.Li + o * cm.ldi + o, cm.ldi, T + o1 * ldt + o, ldt, M, nb, nb, 1.0, 0.0);
      b2.b_mask = 2; b2.klo_mode = 2;
      return gpp_launch_gemm(cf, 2, b2, 1);
    };
    if (border && border_rl && o > 0) HIP_TRY(border_A());
    if (border && !border_rl && o > 0 && border_early) HIP_TRY(border_step());
    if (rem == 0) {
      if (border && !border_rl && o > 0 && !border_early) HIP_TRY(border_step());
      if (be_wait) HIP_TRY(hipStreamWaitEvent(cu.s, be_wait, 0));
      break;
    }
    const int64_t want2 = (rem >= nb_thresh) ? nb_big : nb_small;
    const int64_t nb2 = std::min(want2, rem), rest = rem - nb2;
    // U[o.., c0..c0+nc) = W_oo^T A[o.., c0..c0+nc) : one TN GEMM (W_oo = mirror of the block's inverse, keep i <= k) into
    // the scratch, then copied over A (a GEMM with several row tiles cannot run in place)
    auto row_solve = [&](int64_t c0, int64_t nc) -> hipError_t {
      if (!T) return trsm_rec(cu, c0, nc, o, nb);
      GemmArgs gt = mk(cm.Li + o * cm.ldi + o, cm.ldi, cm.A + o * cm.ld + c0, cm.ld, T + o * ldt + c0, ldt, nb, nc, nb, 1.0,
                       0.0);
      gt.a_mask = 1; gt.khi_mode = 1;
      gt.row_reverse = 1;  // K grows with the row tile: longest tiles first, so the launch does not end on them
      HIP_TRY(gpp_launch_gemm(cu.s, 2, gt, 1));
      return hipMemcpy2DAsync(cm.A + o * cm.ld + c0, cm.ld * sizeof(double), T + o * ldt + c0, ldt * sizeof(double),
                              nc * sizeof(double), nb, hipMemcpyDeviceToDevice, cu.s);
    };
    const double* Urow = cm.A + o * cm.ld;  // block row o: U[o.., :]
    const bool merge_upd = merge_upd_on && split_chain && nb2 % NBLK == 0;
    // the chain first: the columns of the NEXT diagonal block, its update, and the panel stream may go on (event S)
    HIP_TRY(row_solve(o + nb, split_chain ? nb2 : rem));
    GemmArgs g = mk(Urow + (o + nb), cm.ld, Urow + (o + nb), cm.ld, cm.A + (o + nb) * cm.ld + (o + nb), cm.ld, nb2, nb2, nb,
                    -1.0, 1.0);
    g.c_lower = 2;
    HIP_TRY(gpp_launch_gemm(cu.s, 2, g, 1));
    hipEvent_t S = next_event(h);
    if (split_chain) {
      HIP_TRY(hipEventRecord(S, cu.s));
      if (rest > 0) HIP_TRY(row_solve(o + nb + nb2, rest));
    }
    hipEvent_t R = nullptr;
    if (border) {
      R = next_event(h);
      HIP_TRY(hipEventRecord(R, cu.s));
      if (border_rl) HIP_TRY(border_B(R));
    }
    if (rest > 0 && !merge_upd) {  // the part of the next block row to the right of its diagonal block
      GemmArgs g2 = mk(Urow + (o + nb), cm.ld, Urow + (o + nb + nb2), cm.ld, cm.A + (o + nb) * cm.ld + (o + nb + nb2), cm.ld,
                       nb2, rest, nb, -1.0, 1.0);
      HIP_TRY(gpp_launch_gemm(cu.s, 2, g2, 1));
    }
    if (!split_chain) HIP_TRY(hipEventRecord(S, cu.s));
    HIP_TRY(hipStreamWaitEvent(cp.s, S, 0));  // the next diagonal block may be factored
    if (border && !border_rl && o > 0 && !border_early) {
      HIP_TRY(hipStreamWaitEvent(cf, S, 0));  // behind the strip: the chain's own launches get the CUs first
      HIP_TRY(border_step());
    }
    if (be_wait) {
      HIP_TRY(hipStreamWaitEvent(cu.s, be_wait, 0));
      be_wait = nullptr;
    }
    if (rest > 0 && merge_upd) {
      // everything but the next diagonal block (done above) in ONE launch: the next block row's part to the right of its
      // diagonal block no longer runs as a launch of its own with a nearly empty last wave of work-groups
      GemmArgs g3 = mk(Urow + (o + nb), cm.ld, Urow + (o + nb), cm.ld, cm.A + (o + nb) * cm.ld + (o + nb), cm.ld, rem, rem, nb,
                       -1.0, 1.0);
      g3.c_lower = 2;
      g3.skip_lead = (int)nb2;
      // rows [0, rA) of the trailing matrix run beside the next diagonal block on the CU-masked stream; rA is sized so
      // that this takes about as long as that block (split_elems entries of the upper triangle), the rest follows unmasked
      int64_t rA = ((split_elems / rem + NBLK - 1) / NBLK) * NBLK;
      rA = std::max(rA, nb2);
      if (split_upd && rem - rA >= 2048) {
        pend.rows_ready = next_event(h);
        HIP_TRY(hipEventRecord(pend.rows_ready, cu.s));  // block row o is final (the unmasked part needs nothing else)
        GemmArgs a2 = mk(Urow + (o + nb), cm.ld, Urow + (o + nb + rA), cm.ld, cm.A + (o + nb) * cm.ld + (o + nb + rA), cm.ld,
                         rA, rem - rA, nb, -1.0, 1.0);  // rows [0, rA) x columns [rA, rem)
        HIP_TRY(gpp_launch_gemm(cu.s, 2, a2, 1, NBLK, NBLK));
        GemmArgs a1 = g3;  // upper triangle of the leading rA x rA block (minus the next diagonal block): a few hundred
        a1.M = a1.N = (int)rA;  // tiles, issued last so that they run beside the unmasked part instead of on an emptying chip
        // (measured A/B twice: potrf 26.8 -> 25.9 ms at N = 15000, 52.9 -> 52.3 at 20000, 16.2 -> 16.0 at 12288, even at 30000)
        static const bool a1_fill = !(getenv("GPP_A1_FILL") && atoi(getenv("GPP_A1_FILL")) == 0);  // experiment knob
        if (a1_fill) {
          // on the second masked stream, beside the rectangle above: the two launches share ONE ragged last wave of work-groups
          // instead of each ending on its own (a K = 1024 tile runs ~240 us; 990 tiles on 448 slots are 2.2 waves)
          HIP_TRY(hipStreamWaitEvent(cf, pend.rows_ready, 0));
          HIP_TRY(gpp_launch_gemm(cf, 2, a1, 1, NBLK, NBLK));
          hipEvent_t A1 = next_event(h);
          HIP_TRY(hipEventRecord(A1, cf));
          HIP_TRY(hipStreamWaitEvent(cu.s, A1, 0));
        } else {
          HIP_TRY(gpp_launch_gemm(cu.s, 2, a1, 1, NBLK, NBLK));
        }
        pend.g = mk(Urow + (o + nb + rA), cm.ld, Urow + (o + nb + rA), cm.ld, cm.A + (o + nb + rA) * cm.ld + (o + nb + rA),
                    cm.ld, rem - rA, rem - rA, nb, -1.0, 1.0);
        pend.g.c_lower = 2;
        pend.rows_masked = rA;
        pend.on = true;
      } else {
        HIP_TRY(gpp_launch_gemm(cu.s, 2, g3, 1, NBLK, NBLK));
      }
    } else if (rest > 0) {
      GemmArgs g3 = mk(Urow + (o + nb + nb2), cm.ld, Urow + (o + nb + nb2), cm.ld,
                       cm.A + (o + nb + nb2) * cm.ld + (o + nb + nb2), cm.ld, rest, rest, nb, -1.0, 1.0);
      g3.c_lower = 2;
      HIP_TRY(gpp_launch_gemm(cu.s, 2, g3, 1));
    }
    R_prev = R;
  }
  if (pend.on) return hipErrorUnknown;  // (cannot happen: the last step has no trailing update)
  hipEvent_t E = next_event(h);
  HIP_TRY(hipEventRecord(E, cu.s));
  HIP_TRY(hipStreamWaitEvent(cm.s, E, 0));  // (the update stream already waited for the last D)
  if (border) {
    hipEvent_t F = next_event(h);
    HIP_TRY(hipEventRecord(F, cf));
    HIP_TRY(hipStreamWaitEvent(cm.s, F, 0));
    h->inv_nblocks = 1;  // the inverse is complete: gpp_trtri has nothing left to merge
    h->inv_o[0] = 0;
    h->inv_n[0] = N;
  }
  return hipSuccess;
}

// ---- DAG executor (round 5, gpp_dag.hip / gpp_dag_f64) ---------------------------------------------------------------------------
// The factorisation — and for N <= GPP_DAG_INV_MAX the whole inverse beside it, right-looking — as ONE ticket list of tile tasks on
// the throughput CUs, the diagonal blocks as cooperative panel launches on the panel CUs behind one-wave gate kernels.  Replaces
// the launch-per-product look-ahead with bordering in 3840 <= N <= 11264, where the factorisation is bound by its chain of
// diagonal blocks: there every launch boundary (head solve -> diagonal update -> rest solve -> trailing update, in stream order)
// and every bordering product's few long tiles cost idle CUs (profiles/r04_timeline_n10000.txt: 16.3 ms for 10 ms of tile work).
// Plans depend on (N, nb, leading dimensions, flags) only — the operands' addresses are kernel arguments — and live in a small LRU
// (GPP_DAG_PLANS) per handle; a plan that is HIT costs no synchronisation, an eviction does (gpp_dag_free: hipFree).  *used = false: not applicable, the caller continues with the older paths.
// what gpp_trtri will find inverted: the leading block the list built (all of the matrix: nothing left to merge), and every
// diagonal block behind it
void dag_note_inverted(gpp_handle_s* h, const DagPlan* P, int64_t N) {
  h->inv_nblocks = 0;
  if (P->inv_rows > 0) {
    h->inv_o[0] = 0;
    h->inv_n[0] = std::min(P->inv_rows, N);
    h->inv_nblocks = 1;
  }
  for (int b = 0; b < P->B && h->inv_nblocks < 128; ++b) {
    const int64_t o = (int64_t)P->tb[b] * NBLK, rows = std::min<int64_t>((int64_t)P->tb[b + 1] * NBLK, N) - o;
    if (o < P->inv_rows) continue;
    h->inv_o[h->inv_nblocks] = o;
    h->inv_n[h->inv_nblocks] = rows;
    ++h->inv_nblocks;
  }
}

// Raise the abort word of a running list (counters[0]: every wait of the executor, its fillers and its gate kernels sees it and
// leaves) from the handle's stream WITHOUT a CU mask, on which nothing of a list is ever queued — so the store is not ordered behind
// the launches it is meant to end.
hipError_t dag_abort(gpp_handle_s* h, const DagPlan* P) {
  if (!h->full_stream || !P || !P->d_counters) return hipErrorInvalidValue;
  HIP_TRY(gpp_launch_fill_i32(h->full_stream, P->d_counters, 1, 1));
  return hipStreamSynchronize(h->full_stream);  // (an error path: the host may wait for one store)
}

hipError_t potrf_dag(gpp_handle_s* h, const Ctx& cm, int64_t N, double* T, int64_t ldt, bool* used) {
  *used = false;
  static const bool dag_env = !(getenv("GPP_DAG_SCHED") && atoi(getenv("GPP_DAG_SCHED")) == 0);
  // Measured on one box, potrf + inverse (tools/attic/sweep_dag.sh, profiles/r05_dag_sweep.txt): launches win below ~6900 rows, where the
  // chain of diagonal blocks is all there is (4.99 vs 5.39 ms at 6144), the ticket list from there (6.35 vs 6.82 at 7168, 8.05 vs
  // 9.16 at 8192, 13.45 vs 15.68 at 10 000, 18.14 vs 20.46 at 11 264).  With the inverse inside the list up to ~19 000 rows (22.64 vs
  // 24.73 at 12 288, 40.19 vs 41.97 at 15 000); above, the inverse's K = 1024 tiles on 503 slots lose to gpp_trtri's long-K launches on
  // 256 CUs and only the factorisation (+ the leading block of the inverse) runs here (88.23 vs 89.51 at 20 000 with the round-4
  // executor, 284.8 vs 287.1 at 30 000; with the fused steps 86.3 and 278.2).
  static const int64_t dag_min = getenv("GPP_DAG_MIN_N") ? atol(getenv("GPP_DAG_MIN_N")) : 6912;  // (6656: 5.98 vs 5.78 ms for launches; 7168: 6.32 vs 6.82)
  // (up to 65 536 rows since the fused steps: 645.9 vs 663.3 ms at 40 000, 1253 vs 1286 at 50 000, 2165 vs 2213 at 60 000 against
  //  the launch path; without them the list lost at 60 000.  Planning a 60 000-row list takes ~1.2 s of host time, once per size.)
  static const int64_t dag_max = getenv("GPP_DAG_MAX_N") ? atol(getenv("GPP_DAG_MAX_N")) : 65536;
  // (with the fused steps the whole inverse inside the list pays up to ~19 000 rows: 54.7 vs 57.0 ms at 17 000, 69.8 vs 71.5 at 18 500,
  //  87.3 vs 86.7 at 20 000)
  static const int64_t dag_inv_max = getenv("GPP_DAG_INV_MAX") ? atol(getenv("GPP_DAG_INV_MAX")) : 19456;
  static const int64_t dag_nb_env = getenv("GPP_DAG_NB") ? atol(getenv("GPP_DAG_NB")) : 0;
  static const int64_t dag_nb_small = getenv("GPP_DAG_NB_SMALL_N") ? atol(getenv("GPP_DAG_NB_SMALL_N")) : 0;
  if (!dag_env || !h->dag_sched || !h->coop_panel || !T || N < dag_min || N > dag_max) return hipSuccess;
  HIP_TRY(ensure_streams(h));
  if (h->cu_split != 1) return hipSuccess;
  const int64_t nb = dag_nb_env ? dag_nb_env : (N <= dag_nb_small ? 512 : 1024);
  if (nb % NBLK != 0 || !panel_fits(h, nb) || !panel_fits(h, nb + 256)) return hipSuccess;
  // Above GPP_DAG_INV_MAX only the LEADING block of the inverse is built inside the list: its low-priority tasks fill the slots the
  // factorisation leaves idle (start-up, the dips beside each panel, the chain-bound tail), and gpp_trtri, which treats a
  // complete leading block like the diagonal blocks it finds inverted, merges the rest around it.  The block is a power of two
  // times 1024 rows (the pair merges' alignment), by default the largest one <= 0.42 N (8192 of 20 000).
  static const int64_t dag_inv_lead = getenv("GPP_DAG_INV_LEAD") ? atol(getenv("GPP_DAG_INV_LEAD")) : -1;
  int64_t inv_rows = N;
  if (N > dag_inv_max) {
    inv_rows = 0;
    if (dag_inv_lead < 0) {
      for (int64_t r = 2 * nb; r <= (int64_t)(0.42 * (double)N); r *= 2) inv_rows = r;
    } else if (dag_inv_lead >= 2 * nb && dag_inv_lead < N && dag_inv_lead % nb == 0) {
      inv_rows = dag_inv_lead;
    }
  }
  const int flags = inv_rows > 0 ? DAG_INV : 0;
  // the plan: a hit in the handle's LRU, or planned now (host only) and uploaded (a few MB, once per shape)
  DagPlan* P = nullptr;
  int slot = -1, lru = 0;
  for (int i = 0; i < GPP_DAG_PLANS; ++i) {
    DagPlan* q = h->dag_plans[i];
    if (q && q->N == N && q->nb == nb && q->ld == cm.ld && q->ldi == cm.ldi && q->ldt == ldt && q->flags == flags && q->inv_rows == inv_rows) slot = i;
    if (!q) lru = i;
    else if (h->dag_plans[lru] && q->stamp < h->dag_plans[lru]->stamp) lru = i;
  }
  if (slot >= 0) P = h->dag_plans[slot];
  else {
    DagTuning tune = gpp_dag_default_tuning();
    tune.workers = 2 * (h->ncu - h->panel_cus);
    if (tune.fill > 0) tune.fill = 2 * h->panel_cus;
    tune.inv_rows = inv_rows;
    // fused steps (gpp_dag.hip): far tiles take 2 / 4 consecutive steps' updates in one task.  Measured, factor + inverse, same box
    // (profiles/r05_dag_sweep.txt): 26.85 -> 25.97 ms at 13 000 with 2; 40.15 -> 38.98 -> 38.75 at 15 000, 87.9 -> 86.4 -> 86.3 at
    // 20 000, 284.4 -> 278.7 -> 278.2 at 30 000 with 2 -> 4; below ~11 000 rows the longer tasks cost the chain more than the saved
    // epilogues return (8.03 -> 8.19 ms at 8192 with 2; 13.42 -> 13.58 at 10 000 with 4)
    if (!getenv("GPP_DAG_FUSE")) tune.fuse = N >= 14336 ? 4 : N >= 11264 ? 2 : 1;
    P = gpp_dag_plan(N, nb, cm.ld, cm.ldi, ldt, 0, flags, tune);
    if (!P) return hipSuccess;
    for (int b = 0; b < P->B; ++b) {
      const int64_t rows = std::min<int64_t>((int64_t)P->tb[b + 1] * NBLK, N) - (int64_t)P->tb[b] * NBLK;
      if (!panel_fits(h, rows)) {
        gpp_dag_free(P);
        return hipSuccess;
      }
    }
    if (gpp_dag_upload(P) != hipSuccess) {  // no memory for the device copy: not an error of the factorisation
      (void)hipGetLastError();
      gpp_dag_free(P);
      h->dag_sched = 0;
      return hipSuccess;
    }
    if (h->dag_plans[lru]) gpp_dag_free(h->dag_plans[lru]);  // (its own last launch, then hipFree: a device-wide wait, see gpp_dag_free)
    h->dag_plans[lru] = P;
  }
  P->stamp = ++h->dag_clock;
  Ctx cp = cm, cu = cm;
  cp.s = h->panel_stream;
  cu.s = h->upd_stream;
  HIP_TRY(gpp_launch_fill_i32(cm.s, P->d_counters, P->ncounters, 0));
  DagBases bases{{reinterpret_cast<char*>(cm.A), reinterpret_cast<char*>(cm.Li), reinterpret_cast<char*>(T), nullptr}};
  HIP_TRY(gpp_launch_dag_bind(cm.s, P->d_groups, P->d_groups_abs, (int)P->groups.size(), bases));
  hipEvent_t ev = next_event(h);
  HIP_TRY(hipEventRecord(ev, cm.s));  // inputs (kernel build), cleared counters and bound groups are ready
  HIP_TRY(hipStreamWaitEvent(cp.s, ev, 0));
  HIP_TRY(hipStreamWaitEvent(cu.s, ev, 0));
  const long long budget = (long long)(h->panel_timeout_ms > 0 ? h->panel_timeout_ms : 500) * 100000 * 4;
  DagLaunch dl{};
  dl.groups = P->d_groups_abs; dl.tasks = P->d_tasks; dl.ntasks = (int)P->tasks.size();
  dl.counters = P->d_counters; dl.info = cm.info; dl.budget = budget;
  dl.max_tasks = 0; dl.quit_id = -1; dl.quit_val = 0; dl.ticket_limit = 0;
  dl.trace = P->d_trace; dl.tag = 0;
  static const int dag_workers = getenv("GPP_DAG_WORKERS") ? atoi(getenv("GPP_DAG_WORKERS")) : 0;
  // GPP_DAG_PHASED=1 (profiling): the SAME ticket list executed as a sequence of launches on the caller's stream that never wait for
  // each other — panel b, then one launch of the executor's kernel that stops in front of the first task that needs panel b + 1
  // (the fillers' ticket limit, for every work-group), and so on.  Counter collection (rocprofv3 --pmc) serialises dispatches,
  // under which the concurrent form cannot run; this form runs the same kernel over the same tasks in the same order, so its
  // FETCH_SIZE / WRITE_SIZE describe the timed path's tiles (profiles/r05_potrf_pmc.json).
  static const bool phased = getenv("GPP_DAG_PHASED") && atoi(getenv("GPP_DAG_PHASED")) != 0;
  if (phased) {
    const int wgs = 2 * h->ncu;
    for (int b = 0; b < P->B; ++b) {
      const int64_t o = (int64_t)P->tb[b] * NBLK, rows = std::min<int64_t>((int64_t)P->tb[b + 1] * NBLK, N) - o;
      HIP_TRY(launch_panel(h, cm, o, rows, std::min(64, h->ncu)));
      HIP_TRY(gpp_launch_exec_signal(cm.s, P->d_counters, P->c_pd + b));
      DagLaunch pl = dl;
      pl.ticket_limit = b + 1 < P->B ? P->level_first[b + 1] : 0;
      pl.tag = b;
      if (b + 1 == P->B || pl.ticket_limit > 0) HIP_TRY(gpp_launch_dag(cm.s, wgs, pl));
    }
    HIP_TRY(hipEventRecord(P->last_use, cm.s));
    dag_note_inverted(h, P, N);
    *used = true;
    return hipSuccess;
  }
  HIP_TRY(gpp_launch_dag(cu.s, dag_workers > 0 ? dag_workers : 2 * (h->ncu - h->panel_cus), dl));
  // (the executor is running from here on: a failure below raises the abort word and still joins the streams — its work-groups
  //  would otherwise spin for the whole budget behind a call that has already returned an error)
  auto enqueue_panel_stream = [&]() -> hipError_t {
    for (const DagPlan::Op& op : P->stream_ops) {
      const int64_t o = op.kind < 3 ? (int64_t)P->tb[op.arg] * NBLK : 0;
      if (op.kind == 0) {
        HIP_TRY(gpp_launch_exec_gate(cp.s, P->d_counters, P->c_g1d + op.arg, P->gate_target[op.arg], cm.info, budget));
      } else if (op.kind == 1) {
        const int64_t rows = std::min<int64_t>((int64_t)P->tb[op.arg + 1] * NBLK, N) - o;
        HIP_TRY(launch_panel(h, cp, o, rows, h->panel_cus));
      } else if (op.kind == 2) {
        HIP_TRY(gpp_launch_exec_signal(cp.s, P->d_counters, P->c_pd + op.arg));
      } else {
        // filler: two work-groups per panel CU take tasks from the same list until the next diagonal block's update has begun
        DagLaunch fl = dl;
        fl.max_tasks = op.arg;  // (0 behind the last panel: until the list is exhausted)
        fl.quit_id = op.n >= 0 ? P->c_g1d + op.n : -1;
        fl.quit_val = 1;
        fl.ticket_limit = op.lim;
        fl.tag = op.n >= 0 ? op.n : P->B;
        HIP_TRY(gpp_launch_dag(cp.s, 2 * h->panel_cus, fl));
      }
    }
    return hipSuccess;
  };
  const hipError_t perr = enqueue_panel_stream();
  if (perr != hipSuccess) {
    (void)hipGetLastError();
    (void)dag_abort(h, P);
  }
  hipEvent_t E = next_event(h), F = next_event(h);
  HIP_TRY(hipEventRecord(E, cu.s));
  HIP_TRY(hipEventRecord(F, cp.s));
  HIP_TRY(hipStreamWaitEvent(cm.s, E, 0));
  HIP_TRY(hipStreamWaitEvent(cm.s, F, 0));
  HIP_TRY(hipEventRecord(P->last_use, cm.s));
  if (perr != hipSuccess) return perr;
  dag_note_inverted(h, P, N);
  *used = true;
  return hipSuccess;
}

// columns of a buffer that holds only the block-cyclically owned column blocks of an N x N matrix, side by side
inline int64_t owned_cols(int64_t N, int64_t nb, int rank, int nranks) {
  const int64_t nblk = (N + nb - 1) / nb;
  int64_t cols = 0;
  for (int64_t b = rank; b < nblk; b += nranks) cols += std::min(nb, N - b * nb);
  return cols;
}

int check_mat(const void* p, int64_t ld, int64_t n, int argi) {
  if (!p) return -argi;
  if (!aligned16(p) || (ld & 1) || ld < n) return -(argi + 1);
  return 0;
}

}  // namespace

extern "C" {

#ifndef GPP_SRC_HASH
#define GPP_SRC_HASH "unknown"
#endif
const char* gpp_version(void) { return "gpp_hip 0.3 (gfx950, fp64 MFMA) src " GPP_SRC_HASH; }

#ifdef GPP_PANEL_STAMP
// probe build only: the whole ring of panel flag blocks, and the block the next launch will use
int gpp_debug_panel_flags(gpp_handle_t h, int* out, int nints, int* next_slot) {
  if (next_slot) *next_slot = h->panel_next % GPP_PANEL_RING;
  return (int)hipMemcpy(out, h->panel_flags, (size_t)nints * sizeof(int), hipMemcpyDeviceToHost);
}
#endif

// Debug access to the DAG executor's most recently used plan (tools/dag_trace.py): not part of gpp.h.
static DagPlan* dag_mru(gpp_handle_t h) {
  DagPlan* P = nullptr;
  if (h)
    for (int i = 0; i < GPP_DAG_PLANS; ++i)
      if (h->dag_plans[i] && (!P || h->dag_plans[i]->stamp > P->stamp)) P = h->dag_plans[i];
  return P;
}
// info[0..9]: tasks, groups, blocks, tiles per side, counters, simulated us, simulated busy per mille, flags, rows of the leading block
// of the inverse built inside the list (N: all of it, 0: none), N
int gpp_debug_dag_info(gpp_handle_t h, int64_t* info8) {
  const DagPlan* P = dag_mru(h);
  if (!P) return -1;
  info8[8] = P->inv_rows;
  info8[9] = P->N;
  info8[0] = (int64_t)P->tasks.size(); info8[1] = (int64_t)P->groups.size(); info8[2] = P->B; info8[3] = P->nt;
  info8[4] = P->ncounters; info8[5] = (int64_t)(P->sim_ms * 1000.0); info8[6] = (int64_t)(P->sim_busy * 1000.0); info8[7] = P->flags;
  return 0;
}
// (dev) progress of the most recent plan's list after a device synchronisation: out[0] = abort flag, out[1] = tickets taken, out[2] = B,
// then B values each of PD, G1D and — sharded lists — CPH, CPT, ART
int gpp_debug_dag_counters(gpp_handle_t h, int* out, int nmax) {
  DagPlan* P = dag_mru(h);
  if (!P || !P->d_counters) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  std::vector<int> c(P->ncounters);
  if (hipMemcpy(c.data(), P->d_counters, c.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int n = 0;
  auto put = [&](int v) { if (n < nmax) out[n++] = v; };
  put(c[0]); put(c[1]); put(P->B);
  for (int b = 0; b < P->B; ++b) put(c[P->c_pd + b]);
  for (int b = 0; b < P->B; ++b) put(c[P->c_g1d + b]);
  if (P->flags & DAG_SHARD) {  // (of the tails' pieces: the first one's counters)
    for (int b = 0; b < P->B; ++b) put(c[P->c_cph + b]);
    for (int b = 0; b < P->B; ++b) put(c[P->c_cpt + b * P->GP]);
    for (int b = 0; b < P->B; ++b) put(c[P->c_art + b * P->GP]);
  }
  return n;
}
int gpp_debug_dag_trace(gpp_handle_t h, int on) {
  DagPlan* P = dag_mru(h);
  if (!P) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (on && !P->d_trace) {
    if (hipMalloc(&P->d_trace, 4 * P->tasks.size() * sizeof(unsigned long long)) != hipSuccess) return 2;
    (void)hipMemset(P->d_trace, 0, 4 * P->tasks.size() * sizeof(unsigned long long));
  } else if (!on && P->d_trace) {
    (void)hipFree(P->d_trace);
    P->d_trace = nullptr;
  }
  return 0;
}
int gpp_debug_dag_fetch(gpp_handle_t h, void* tasks, void* trace) {
  const DagPlan* P = dag_mru(h);
  if (!P) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (tasks) memcpy(tasks, P->tasks.data(), P->tasks.size() * sizeof(DagTask));
  if (trace) {
    if (!P->d_trace) return 2;
    if (hipMemcpy(trace, P->d_trace, 4 * P->tasks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 3;
  }
  return 0;
}

int gpp_create(gpp_handle_t* out, int device) {
  if (!out) return -1;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess) return rc(e);
  if (device < 0 || device >= ndev) return -2;
  e = hipSetDevice(device);
  if (e != hipSuccess) return rc(e);
  gpp_handle_s* h = new (std::nothrow) gpp_handle_s();
  if (!h) return rc(hipErrorOutOfMemory);
  h->device = device;
  h->stream = nullptr;
  h->ws = nullptr;
  h->ws_bytes = 0;
  h->panel_stream = nullptr;
  h->upd_stream = nullptr;
  h->fill_stream = nullptr;
  h->full_stream = nullptr;
  h->cu_split = -1;
  h->panel_cus = 0;
  h->n_events = 0;
  h->ev_next = 0;
  h->inv_N = 0;
  h->inv_nblocks = 0;
  h->panel_flags = nullptr;
  h->panel_next = 0;
  h->coop_panel = panel_enabled() ? 1 : 0;
  h->panel_fault = 0;
  h->panel_timeout_ms = 500;
  for (int i = 0; i < GPP_DAG_PLANS; ++i) h->dag_plans[i] = nullptr;
  h->shard_cur = nullptr;
  h->shard_info = nullptr;
  h->shard_ready = nullptr;
  h->dag_clock = 0;
  h->dag_sched = 1;
  h->ncu = 0;
  if (hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || h->ncu < 2) {
    (void)hipGetLastError();
    h->ncu = 0;
  }
  // (allocated here, not lazily: a first use inside a stream capture could not allocate)
  h->cap_next = 0;
  h->handoff = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&h->panel_flags), (GPP_PANEL_RING + GPP_PANEL_CAP_RING) * gpp_panel_flag_bytes()) != hipSuccess ||
      hipMemset(h->panel_flags, 0, (GPP_PANEL_RING + GPP_PANEL_CAP_RING) * gpp_panel_flag_bytes()) != hipSuccess) {
    (void)hipGetLastError();
    if (h->panel_flags) (void)hipFree(h->panel_flags);
    h->panel_flags = nullptr;  // the leaf-step chain is used instead
  }
  *out = h;
  return 0;
}

int gpp_destroy(gpp_handle_t h) {
  if (!h) return -1;
  // (every internal stream drained first, then destroyed in the reverse order of creation: fill_stream and upd_stream carry the SAME
  //  CU mask, and destroying upd_stream first left hipStreamDestroy(fill_stream) hanging in ~3 of 20 runs of examples/shard_eval_c —
  //  on an idle device)
  for (hipStream_t s : {h->panel_stream, h->upd_stream, h->fill_stream, h->full_stream})
    if (s) (void)hipStreamSynchronize(s);
  if (h->full_stream) (void)hipStreamDestroy(h->full_stream);
  if (h->fill_stream) (void)hipStreamDestroy(h->fill_stream);
  if (h->upd_stream) (void)hipStreamDestroy(h->upd_stream);
  if (h->panel_stream) (void)hipStreamDestroy(h->panel_stream);
  for (int i = 0; i < h->n_events; ++i) (void)hipEventDestroy(h->events[i]);
  if (h->handoff) (void)hipEventDestroy(h->handoff);
  if (h->shard_ready) (void)hipEventDestroy(h->shard_ready);
  if (h->panel_flags) (void)hipFree(h->panel_flags);
  for (int i = 0; i < GPP_DAG_PLANS; ++i)
    if (h->dag_plans[i]) gpp_dag_free(h->dag_plans[i]);
  gpp_shard_release_comm(h);
  delete h;
  return 0;
}

int gpp_set_stream(gpp_handle_t h, void* stream) {
  if (!h) return -1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (s != h->stream) {
    // A handle works on ONE stream at a time (its scratch workspace, the panel flag ring and the executor's counters are shared by
    // consecutive calls).  Moving it to another stream therefore orders the new stream behind everything enqueued on the old one —
    // enforced here, not assumed.  (Not while either stream is capturing: a capture may not depend on work outside it, and the
    // caller of a capture synchronises around it anyway.)
    hipStreamCaptureStatus a = hipStreamCaptureStatusNone, b = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &a) != hipSuccess || hipStreamIsCapturing(s, &b) != hipSuccess) {
      (void)hipGetLastError();
      a = b = hipStreamCaptureStatusActive;  // unknown: leave the streams alone
    }
    // (nor for the handle's own CU-masked streams, which a host-side driver that overlaps its own launches — the sharded
    //  evaluation — obtains through gpp_internal_stream and orders with its own events)
    auto internal = [&](hipStream_t q) { return q && (q == h->panel_stream || q == h->upd_stream || q == h->fill_stream || q == h->full_stream); };
    if (a == hipStreamCaptureStatusNone && b == hipStreamCaptureStatusNone && !internal(s) && !internal(h->stream)) {
      if (!h->handoff) GPP_TRY(hipEventCreateWithFlags(&h->handoff, hipEventDisableTiming));
      GPP_TRY(hipEventRecord(h->handoff, h->stream));
      GPP_TRY(hipStreamWaitEvent(s, h->handoff, 0));
    }
    h->stream = s;
  }
  return 0;
}

int gpp_set_option(gpp_handle_t h, int option, int value) {
  if (!h) return -1;
  if (option == GPP_OPT_COOP_PANEL) h->coop_panel = value ? 1 : 0;
  else if (option == GPP_OPT_PANEL_FAULT) h->panel_fault = value ? 1 : 0;
  else if (option == GPP_OPT_DAG_SCHED || option == GPP_OPT_EXEC_SCHED) h->dag_sched = value ? 1 : 0;
  else if (option == GPP_OPT_PANEL_TIMEOUT_MS) {
    if (value < 1 || value > 60000) return -3;
    h->panel_timeout_ms = value;
  } else return -2;
  return 0;
}

int gpp_internal_stream(gpp_handle_t h, int which, void** out) {
  if (!h) return -1;
  if (which < 0 || which > 2) return -2;
  if (!out) return -3;
  GPP_TRY(ensure_streams(h));
  *out = reinterpret_cast<void*>(which == 0 ? h->panel_stream : which == 1 ? h->upd_stream : h->full_stream);
  return 0;
}

// ---- the sharded evaluation's factorisation + forward sweep as ONE ticket list per rank (gpp_dag.hip, DAG_SHARD) -------------------
static long long shard_budget(const gpp_handle_s* h) {
  // the waits of a sharded list include the other ranks' progress (and, the first time, their planning): generous by default
  static const long long ms = getenv("GPP_SHARD_TIMEOUT_MS") ? atoll(getenv("GPP_SHARD_TIMEOUT_MS")) : 60000;
  (void)h;
  return ms * 100000;
}

int gpp_shard_list_begin(gpp_handle_t h, int64_t N, int64_t nb, int rank, int nranks, double* A, int64_t ld, double* Kc, double* Lc,
                         int64_t ldc, double* D, double* W0, double* W1, double* W2, int64_t ldw, int32_t* info, int workers,
                         int* used) {
  if (!h) return -1;
  if (!used) return -18;
  *used = 0;
  if (N < 2 || nb < 128 || nb % 128 != 0) return -2;
  if (rank < 0 || nranks < 1 || rank >= nranks) return -4;
  if (int e = check_mat(A, ld, N, 6)) return e;
  if (!Kc || !aligned16(Kc)) return -8;
  if (!Lc || !aligned16(Lc)) return -9;
  const int64_t wc = ((N + nb - 1) / nb - rank + nranks - 1) / nranks * nb;  // owned blocks, whole
  if ((ldc & 1) || ldc < wc) return -10;
  if (!D || !aligned16(D)) return -11;
  if (!W0 || !W1 || !W2 || !aligned16(W0) || !aligned16(W1) || !aligned16(W2)) return -12;
  if ((ldw & 1) || ldw < N) return -15;
  if (!info) return -16;
  if (h->shard_cur) return -1;  // a list is open on this handle
  if (workers <= 0 && getenv("GPP_SHARD_WORKERS")) workers = atoi(getenv("GPP_SHARD_WORKERS"));  // (tests: ranks sharing one GPU)
  static const bool list_env = !(getenv("GPP_SHARD_LIST") && atoi(getenv("GPP_SHARD_LIST")) == 0);
  static const int64_t list_min = getenv("GPP_SHARD_LIST_MIN_N") ? atol(getenv("GPP_SHARD_LIST_MIN_N")) : 4096;
  if (!list_env || !h->dag_sched || !h->coop_panel || N < list_min || N > 65536) return 0;
  GPP_TRY(ensure_streams(h));
  if (h->cu_split != 1) return 0;
  if (!panel_fits(h, nb)) return 0;
  const int flags = DAG_INV | DAG_SHARD;
  DagPlan* P = nullptr;
  int slot = -1, lru = 0;
  for (int i = 0; i < GPP_DAG_PLANS; ++i) {
    DagPlan* q = h->dag_plans[i];
    if (q && q->N == N && q->nb == nb && q->ld == ld && q->ldi == ldc && q->ldt == ldw && q->flags == flags && q->rank == rank &&
        q->nranks == nranks && q->workers == workers)
      slot = i;
    if (!q) lru = i;
    else if (h->dag_plans[lru] && q->stamp < h->dag_plans[lru]->stamp) lru = i;
  }
  const int nworkers = workers > 0 ? workers : 2 * (h->ncu - h->panel_cus);
  if (slot >= 0) P = h->dag_plans[slot];
  else {
    DagTuning tune = gpp_dag_default_tuning();
    tune.workers = nworkers;
    // filler launches on the panel's CUs between this rank's panels (as in gpp_potrf_ws); not where ranks share one GPU (tests pass
    // `workers`): another rank's panel needs those CUs while a filler here may be waiting for that very rank's message
    static const int fill_env = getenv("GPP_SHARD_FILL") ? atoi(getenv("GPP_SHARD_FILL")) : -1;
    // With messages on the wire (nranks > 1) ONE filler work-group per panel CU: the executor's work-groups fill their CUs to the last
    // register, so the collectives' kernels, the packing copies and the gate / signal kernels live on the panel CUs — and a filler
    // that waits for a message must never be what keeps that message's kernel from being scheduled.
    tune.fill = fill_env >= 0 ? fill_env : (workers > 0 || tune.fill <= 0) ? 0 : (nranks > 1 ? h->panel_cus : 2 * h->panel_cus);
    tune.inv_rows = 0;
    if (!getenv("GPP_DAG_FUSE")) tune.fuse = N >= 14336 ? 4 : N >= 11264 ? 2 : 1;
    P = gpp_dag_plan(N, nb, ld, ldc, ldw, 0, flags, tune, rank, nranks);
    if (!P) return 0;
    P->workers = workers;
    P->fill = tune.fill;
    if (gpp_dag_upload(P) != hipSuccess) {
      (void)hipGetLastError();
      gpp_dag_free(P);
      return 0;
    }
    if (h->dag_plans[lru]) gpp_dag_free(h->dag_plans[lru]);
    h->dag_plans[lru] = P;
  }
  P->stamp = ++h->dag_clock;
  hipStream_t sm = h->stream, sp = h->panel_stream, su = h->upd_stream;
  GPP_TRY(gpp_launch_fill_i32(sm, P->d_counters, P->ncounters, 0));
  DagBases bases{{reinterpret_cast<char*>(A), reinterpret_cast<char*>(Kc), reinterpret_cast<char*>(Lc), reinterpret_cast<char*>(D),
                  reinterpret_cast<char*>(W0), reinterpret_cast<char*>(W1), reinterpret_cast<char*>(W2), nullptr}};
  GPP_TRY(gpp_launch_dag_bind(sm, P->d_groups, P->d_groups_abs, (int)P->groups.size(), bases));
  // (recorded BEFORE the executor is launched: the caller's stream may be the legacy default stream, whose markers wait for every
  //  blocking stream's earlier work — an event recorded on it behind the executor's launch would complete with the list)
  if (!h->shard_ready) GPP_TRY(hipEventCreateWithFlags(&h->shard_ready, hipEventDisableTiming));
  hipEvent_t ev = h->shard_ready;
  GPP_TRY(hipEventRecord(ev, sm));
  GPP_TRY(hipStreamWaitEvent(sp, ev, 0));
  GPP_TRY(hipStreamWaitEvent(su, ev, 0));
  const long long budget = shard_budget(h);
  DagLaunch dl{};
  dl.groups = P->d_groups_abs; dl.tasks = P->d_tasks; dl.ntasks = (int)P->tasks.size();
  dl.counters = P->d_counters; dl.info = info; dl.budget = budget;
  dl.max_tasks = 0; dl.quit_id = -1; dl.quit_val = 0; dl.ticket_limit = 0;
  dl.trace = P->d_trace; dl.tag = 0;
  if (dl.ntasks > 0) GPP_TRY(gpp_launch_dag(su, nworkers, dl));
  // From here on the persistent executor is RUNNING: a failure below must not leave its work-groups spinning for the whole time-out
  // with the handle unable to join them.  The list stays open (the caller's gpp_shard_list_end joins the streams and records the
  // plan's last use) and the abort word is raised from a stream nothing of the list is queued on.
  Ctx cp{sp, A, ld, D, nb, info};
  auto enqueue_panel_stream = [&]() -> int {
    for (const DagPlan::Op& op : P->stream_ops) {
      const int b = op.arg;
      const int64_t o = (int64_t)b * nb, rows = std::min(nb, N - o);
      if (op.kind == 0) {
        GPP_TRY(gpp_launch_exec_gate(sp, P->d_counters, P->c_g1d + b, P->gate_target[b], info, budget));
      } else if (op.kind == 1) {
        // the diagonal block's factor in place, its inverse (+ mirror) into D[b]
        Ctx cb = cp;
        cb.A = A + o * ld + o;
        cb.Li = D + (int64_t)b * nb * nb;
        GPP_TRY(launch_panel_at(h, cb, o, rows, h->panel_cus));
      } else if (op.kind == 2) {
        GPP_TRY(gpp_launch_exec_signal(sp, P->d_counters, P->c_pd + b));
      } else {
        DagLaunch fl = dl;
        fl.max_tasks = op.arg;
        fl.quit_id = op.n >= 0 ? P->c_g1d + op.n : -1;
        fl.quit_val = 1;
        fl.ticket_limit = op.lim;
        fl.tag = op.n >= 0 ? op.n : P->B;
        GPP_TRY(gpp_launch_dag(sp, std::max(P->fill, 1), fl));
      }
    }
    return 0;
  };
  const int erc = enqueue_panel_stream();
  h->shard_cur = P;
  h->shard_info = info;
  if (erc != 0) {
    (void)hipGetLastError();
    (void)dag_abort(h, P);
    return erc;  // (*used stays 0; the list is open: gpp_shard_list_end must still be called)
  }
  *used = 1;
  return 0;
}

/* tail == 0: `stream` waits until this rank's panel of block row k is done and its copies of the head's strips (the columns of block
 * k + 1) are in A — the head message (diagonal block, D[k], head) can be packed; tail = 1 + g: the same for piece g of the columns
 * behind (gpp_shard_piece_cols() columns each). */
static int shard_pieces(const DagPlan* P, int k) {  // pieces of block row k's tail (gpp_dag.hip: Planner::npieces)
  const int hi = P->tb[std::min(k + 2, P->B)];
  return hi >= P->nt ? 0 : (P->nt - hi + P->piece_tiles - 1) / P->piece_tiles;
}
int gpp_shard_list_gate(gpp_handle_t h, void* stream, int tail, int k) {
  if (!h || !h->shard_cur) return -1;
  const DagPlan* P = h->shard_cur;
  if (k < 0 || k >= P->B || k % P->nranks != P->rank) return -4;
  if (tail < 0 || tail > shard_pieces(P, k)) return -3;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long budget = shard_budget(h);
  GPP_TRY(hipStreamWaitEvent(s, h->shard_ready, 0));  // the counters are cleared
  GPP_TRY(gpp_launch_exec_gate(s, P->d_counters, P->c_pd + k, 1, h->shard_info, budget));
  const int piece = k * P->GP + tail - 1;
  const int target = tail ? P->cpt_target[piece] : P->cph_target[k];
  if (target > 0) GPP_TRY(gpp_launch_exec_gate(s, P->d_counters, tail ? P->c_cpt + piece : P->c_cph + k, target, h->shard_info, budget));
  return 0;
}

/* Behind the unpacked head (tail == 0) / piece tail - 1 of the tail of another rank's block row k on `stream`: its tasks may run. */
int gpp_shard_list_signal(gpp_handle_t h, void* stream, int tail, int k) {
  if (!h || !h->shard_cur) return -1;
  const DagPlan* P = h->shard_cur;
  if (k < 0 || k >= P->B || k % P->nranks == P->rank) return -4;
  if (tail < 0 || tail > shard_pieces(P, k)) return -3;
  GPP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), h->shard_ready, 0));
  return rc(gpp_launch_exec_signal(reinterpret_cast<hipStream_t>(stream), P->d_counters, tail ? P->c_art + k * P->GP + tail - 1 : P->c_pd + k));
}

/* The handle's stream waits for the list (its executor and its panels). */
int gpp_shard_list_end(gpp_handle_t h) {
  if (!h || !h->shard_cur) return -1;
  DagPlan* P = h->shard_cur;
  h->shard_cur = nullptr;
  hipEvent_t E = next_event(h), F = next_event(h);
  GPP_TRY(hipEventRecord(E, h->upd_stream));
  GPP_TRY(hipEventRecord(F, h->panel_stream));
  GPP_TRY(hipStreamWaitEvent(h->stream, E, 0));
  GPP_TRY(hipStreamWaitEvent(h->stream, F, 0));
  GPP_TRY(hipEventRecord(P->last_use, h->stream));
  return 0;
}

int gpp_shard_back_list(gpp_handle_t h, int64_t N, int64_t nb, int rank, int nranks, const double* A, int64_t ld, double* Kc, double* Lc,
                        int64_t ldc, const double* D, int32_t* info, int workers, int* used) {
  if (!h) return -1;
  if (!used) return -14;
  *used = 0;
  if (N < 2 || nb < 128 || nb % 128 != 0) return -2;
  if (rank < 0 || nranks < 1 || rank >= nranks) return -4;
  if (int e = check_mat(A, ld, N, 6)) return e;
  if (!Kc || !aligned16(Kc)) return -8;
  if (!Lc || !aligned16(Lc)) return -9;
  const int64_t wc = ((N + nb - 1) / nb - rank + nranks - 1) / nranks * nb;
  if ((ldc & 1) || ldc < wc) return -10;
  if (!D || !aligned16(D)) return -11;
  if (!info) return -12;
  static const bool list_env = !(getenv("GPP_SHARD_LIST") && atoi(getenv("GPP_SHARD_LIST")) == 0) &&
                               !(getenv("GPP_SHARD_BACK_LIST") && atoi(getenv("GPP_SHARD_BACK_LIST")) == 0);
  static const int64_t list_min = getenv("GPP_SHARD_LIST_MIN_N") ? atol(getenv("GPP_SHARD_LIST_MIN_N")) : 4096;
  if (!list_env || !h->dag_sched || N < list_min || N > 65536 || h->shard_cur) return 0;
  if (workers <= 0 && getenv("GPP_SHARD_WORKERS")) workers = 2 * atoi(getenv("GPP_SHARD_WORKERS"));
  GPP_TRY(ensure_streams(h));
  const int flags = DAG_SHARD | DAG_BACK;
  DagPlan* P = nullptr;
  int slot = -1, lru = 0;
  for (int i = 0; i < GPP_DAG_PLANS; ++i) {
    DagPlan* q = h->dag_plans[i];
    if (q && q->N == N && q->nb == nb && q->ld == ld && q->ldi == ldc && q->flags == flags && q->rank == rank && q->nranks == nranks &&
        q->workers == workers)
      slot = i;
    if (!q) lru = i;
    else if (h->dag_plans[lru] && q->stamp < h->dag_plans[lru]->stamp) lru = i;
  }
  const int nworkers = workers > 0 ? workers : 2 * h->ncu;
  if (slot >= 0) P = h->dag_plans[slot];
  else {
    DagTuning tune = gpp_dag_default_tuning();
    tune.workers = nworkers;
    tune.fill = 0;
    tune.inv_rows = 0;
    if (!getenv("GPP_DAG_FUSE")) tune.fuse = N >= 14336 ? 4 : N >= 11264 ? 2 : 1;
    P = gpp_dag_plan(N, nb, ld, ldc, ld, 0, flags, tune, rank, nranks);
    if (!P) return 0;
    P->workers = workers;
    if (P->tasks.empty() || gpp_dag_upload(P) != hipSuccess) {
      (void)hipGetLastError();
      gpp_dag_free(P);
      return 0;
    }
    if (h->dag_plans[lru]) gpp_dag_free(h->dag_plans[lru]);
    h->dag_plans[lru] = P;
  }
  P->stamp = ++h->dag_clock;
  hipStream_t sm = h->stream;
  GPP_TRY(gpp_launch_fill_i32(sm, P->d_counters, P->ncounters, 0));
  DagBases bases{{const_cast<char*>(reinterpret_cast<const char*>(A)), reinterpret_cast<char*>(Kc), reinterpret_cast<char*>(Lc),
                  const_cast<char*>(reinterpret_cast<const char*>(D)), nullptr, nullptr, nullptr, nullptr}};
  GPP_TRY(gpp_launch_dag_bind(sm, P->d_groups, P->d_groups_abs, (int)P->groups.size(), bases));
  DagLaunch dl{};
  dl.groups = P->d_groups_abs; dl.tasks = P->d_tasks; dl.ntasks = (int)P->tasks.size();
  dl.counters = P->d_counters; dl.info = info;
  dl.budget = shard_budget(h);  // (every wait is for this rank's own tasks; generous all the same: ranks that share a GPU in tests)
  dl.max_tasks = 0; dl.quit_id = -1; dl.quit_val = 0; dl.ticket_limit = 0;
  dl.trace = P->d_trace; dl.tag = 0;
  GPP_TRY(gpp_launch_dag(sm, nworkers, dl));
  GPP_TRY(hipEventRecord(P->last_use, sm));
  *used = 1;
  return 0;
}

size_t gpp_workspace_bytes(gpp_handle_t h, int op, int64_t N, int64_t M, int D, int S) {
  (void)h;
  (void)M;
  if (op == GPP_OP_MLL_EVAL) {
    return std::max(gpp_grad_ws_bytes(N, D, S, D), gpp_trmv_t_ws_bytes(N)) + 256;
  }
  if (op == GPP_OP_PREDICT) return 256;
  return 0;
}

int gpp_set_workspace(gpp_handle_t h, void* ws, size_t bytes) {
  if (!h) return -1;
  if (ws && !aligned16(ws)) return -2;
  h->ws = ws;
  h->ws_bytes = bytes;
  return 0;
}

int gpp_kernel_build(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                     const double* tau, const int32_t* grp, int S, double jitter, int kind, int d_split, int uplo,
                     double* Ky, int64_t ld, int64_t row0, int64_t nrows) {
  if (!h) return -1;
  if (!U) return -2;
  if (N < 0) return -3;
  if (D < 1 || D > 64) return -4;
  if (!w) return -5;
  if (!sf2) return -6;
  if (tau && S < 1) return -9;
  if (kind < 0 || kind > 2) return -11;
  if (d_split < 0 || d_split > D) return -12;
  if (uplo != GPP_UPLO_FULL && uplo != GPP_UPLO_LOWER && uplo != GPP_UPLO_UPPER) return -13;
  if (!Ky) return -14;
  if (ld < N) return -15;
  if (row0 < 0 || nrows < 0 || row0 + nrows > N || (row0 % 64) != 0) return -16;
  if (uplo == GPP_UPLO_UPPER && (row0 != 0 || nrows != N)) return -16;  // row shards: full or lower mode only
  GPP_TRY(gpp_launch_kernel_build(h->stream, U, N, D, w, sf2, tau, grp, S, jitter, kind, d_split, uplo, Ky, ld, row0, nrows));
  return 0;
}

int gpp_cross_kernel(gpp_handle_t h, const double* Ua, int64_t Ma, const double* Ub, int64_t Nb, int D, const double* w,
                     const double* sf2, int kind, int d_split, double* Kab, int64_t ld) {
  if (!h) return -1;
  if (!Ua) return -2;
  if (Ma < 0) return -3;
  if (!Ub) return -4;
  if (Nb < 0) return -5;
  if (D < 1 || D > 64) return -6;
  if (!w) return -7;
  if (!sf2) return -8;
  if (kind < 0 || kind > 2) return -9;
  if (d_split < 0 || d_split > D) return -10;
  if (!Kab) return -11;
  if (ld < Nb) return -12;
  GPP_TRY(gpp_launch_cross_kernel(h->stream, Ua, Ma, Ub, Nb, D, w, sf2, kind, d_split, Kab, ld));
  return 0;
}

int gpp_potrf_ws(gpp_handle_t h, double* A, int64_t N, int64_t ld, double* Linv, int64_t ldi, double* T, int64_t ldt,
                 int32_t* info_dev) {
  if (!h) return -1;
  if (N < 0) return -3;
  if (int r = check_mat(A, ld, N, 2)) return r;
  if (int r = check_mat(Linv, ldi, N, 5)) return r;
  if (T) {
    if (int r = check_mat(T, ldt, N, 7)) return r;
  }
  if (!info_dev) return -9;
  GPP_TRY(gpp_launch_fill_i32(h->stream, info_dev, 1, 0));  // (a kernel, not a memset: see gpp_launch_fill_i32)
  Ctx c{h->stream, A, ld, Linv, ldi, info_dev};
  h->inv_N = N;
  h->inv_nblocks = 0;
  static const int64_t la_min = getenv("GPP_LOOKAHEAD_MIN") ? atol(getenv("GPP_LOOKAHEAD_MIN")) : 6 * LOOKAHEAD_NB;  // knob
  // (measured: the leaf-step factorisation on one stream wins up to ~6000 rows — 2.99 vs 3.46 ms at 4096, 4.25 vs 4.53 at
  //  5120, a tie at 6144; the look-ahead wins from there: 8.6 vs 10.1 ms at 8192)
  static const int64_t la_min_b = getenv("GPP_BORDER_MIN") ? atol(getenv("GPP_BORDER_MIN")) : BORDER_MIN_N;  // knob
  bool dag_used = false;
  GPP_TRY(potrf_dag(h, c, N, T, ldt, &dag_used));
  if (dag_used) return 0;
  if (N >= (T ? std::min(la_min, la_min_b) : la_min)) {
    GPP_TRY(potrf_lookahead(h, c, N, LOOKAHEAD_NB, T, ldt));
  } else if (panel_fits(h, N)) {
    // a small matrix: factor and inverse by one cooperative launch on the caller's stream; gpp_trtri finds the inverse complete
    // (every work-group of the launch must fit on the stream's CUs at once: on the handle's own CU-masked streams that is fewer)
    int wgs = h->ncu;
    if (h->cu_split == 1 && h->stream == h->panel_stream) wgs = h->panel_cus;
    else if (h->cu_split == 1 && (h->stream == h->upd_stream || h->stream == h->fill_stream)) wgs = h->ncu - h->panel_cus;
    else if (h->stream) {
      // a caller's stream may carry a CU mask of its own: the launch must not have more work-groups than that mask has CUs
      uint32_t mask[32] = {0};
      if (hipExtStreamGetCUMask(h->stream, 32, mask) == hipSuccess) {
        int cus = 0;
        for (int c = 0; c < h->ncu && c < 1024; ++c) cus += (mask[c >> 5] >> (c & 31)) & 1u;
        if (cus >= 2 && cus < wgs) wgs = cus;
      } else {
        (void)hipGetLastError();
      }
    }
    GPP_TRY(launch_panel(h, c, 0, N, wgs));
    h->inv_nblocks = 1;
    h->inv_o[0] = 0;
    h->inv_n[0] = N;
  } else {
    GPP_TRY(potrf_rec(c, 0, N));
  }
  return 0;
}

int gpp_potrf(gpp_handle_t h, double* A, int64_t N, int64_t ld, double* Linv, int64_t ldi, int32_t* info_dev) {
  const int r = gpp_potrf_ws(h, A, N, ld, Linv, ldi, nullptr, 0, info_dev);
  return r == -9 ? -7 : r;
}

int gpp_trtri(gpp_handle_t h, const double* U, int64_t N, int64_t ld, double* Linv, int64_t ldi, double* T, int64_t ldt) {
  if (!h) return -1;
  if (N < 0) return -3;
  if (int r = check_mat(U, ld, N, 2)) return r;
  if (int r = check_mat(Linv, ldi, N, 5)) return r;
  if (int r = check_mat(T, ldt, N, 7)) return r;
  // diagonal blocks already inverted by the preceding gpp_potrf_ws (same N): a pair inside one of them is done
  const bool have = (h->inv_N == N && h->inv_nblocks > 0);
  for (int64_t s = NBLK; s < N; s *= 2) {
    auto skip = [&](int64_t b) {
      if (!have) return false;
      for (int i = 0; i < h->inv_nblocks; ++i)
        if (b >= h->inv_o[i] && b + 2 * s <= h->inv_o[i] + h->inv_n[i]) return true;
      // a ragged pair [b, N) inside the last block
      for (int i = 0; i < h->inv_nblocks; ++i)
        if (b >= h->inv_o[i] && h->inv_o[i] + h->inv_n[i] == N && b + s < N) return true;
      return false;
    };
    GPP_TRY(trtri_level(h->stream, U, ld, Linv, ldi, T, ldt, 0, N, s, skip));
  }
  h->inv_N = 0;  // consumed
  return 0;
}

int gpp_lauum(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, double* Kinv, int64_t ldk) {
  if (!h) return -1;
  if (N < 0) return -3;
  if (int r = check_mat(Linv, ldi, N, 2)) return r;
  if (int r = check_mat(Kinv, ldk, N, 5)) return r;
  GemmArgs g = mk(Linv, ldi, Linv, ldi, Kinv, ldk, N, N, N, 1.0, 0.0);
  g.a_mask = 2; g.b_mask = 2; g.klo_mode = 3; g.c_lower = 1; g.tag = 1;
  // experiment knob: walk every tile's K range from the top (k = N) down, so that the tile rows in flight — whose ranges start at
  // different k but all END at N — sweep the shared operand columns in lockstep
  static const int krev = getenv("GPP_LAUUM_KREV") ? atoi(getenv("GPP_LAUUM_KREV")) : 0;
  g.k_reverse = krev;
  // one launch of long-K triangular tiles: small tiles balance it until there are several waves of big ones (measured:
  // N = 1024 0.167 / 0.061 / 0.034 ms with 128 / 64 / 32-wide tiles, 2048 0.312 / 0.130 / 0.098, 3072 0.490 / 0.231 /
  // 0.278, 4096 0.742 / 0.502 / 0.610, 6144 1.38 / 1.55 / 1.94)
  static const int lt = getenv("GPP_LAUUM_TILE") ? atoi(getenv("GPP_LAUUM_TILE")) : 0;  // experiment knob (0: by size)
  const int t = lt ? lt : (N <= 2560 ? 32 : N <= 5120 ? 64 : NBLK);
  GPP_TRY(gpp_launch_gemm(h->stream, 2, g, 1, t, t == 256 ? 128 : t));  // (256: the tall 256 x 128 tile, 8 waves — experiment)
  return 0;
}

int gpp_lauum_rows(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, double* Kinv, int64_t ldk, int rank, int nranks) {
  if (!h) return -1;
  if (N < 0) return -3;
  if (int r = check_mat(Linv, ldi, N, 2)) return r;
  if (int r = check_mat(Kinv, ldk, N, 5)) return r;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -7;
  GemmArgs g = mk(Linv, ldi, Linv, ldi, Kinv, ldk, N, N, N, 1.0, 0.0);
  g.a_mask = 2; g.b_mask = 2; g.klo_mode = 3; g.c_lower = 1; g.tag = 1;
  g.row_mod = nranks; g.row_off = rank;
  GPP_TRY(gpp_launch_gemm(h->stream, 2, g, 1, NBLK, NBLK));
  return 0;
}

int gpp_lauum_rows_range(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, double* Kinv, int64_t ldk, int rank,
                         int nranks, int64_t row0, int64_t row1) {
  if (!h) return -1;
  if (N < 0) return -3;
  if (int r = check_mat(Linv, ldi, N, 2)) return r;
  if (int r = check_mat(Kinv, ldk, N, 5)) return r;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -7;
  if (row0 < 0 || row1 < row0 || row1 > N || (row0 % NBLK) != 0) return -9;
  if (row1 == row0) return 0;
  GemmArgs g = mk(Linv, ldi, Linv, ldi, Kinv, ldk, N, N, N, 1.0, 0.0);
  g.a_mask = 2; g.b_mask = 2; g.klo_mode = 3; g.c_lower = 1; g.tag = 1;
  g.row_mod = nranks; g.row_off = rank;
  // owned tile rows are t = rank + nranks * i: those inside [row0, row1) have ceil((t0 - rank) / nranks) <= i < ceil((t1 - rank) / nranks)
  const int64_t t0 = row0 / NBLK, t1 = (row1 + NBLK - 1) / NBLK;
  auto first_i = [&](int64_t t) { return t <= rank ? (int64_t)0 : (t - rank + nranks - 1) / nranks; };
  g.row_i0 = (int)first_i(t0);
  g.row_i1 = (int)first_i(t1);
  if (g.row_i1 <= g.row_i0) return 0;
  GPP_TRY(gpp_launch_gemm(h->stream, 2, g, 1, NBLK, NBLK));
  return 0;
}

int gpp_transpose(gpp_handle_t h, const double* src, int64_t lds, int64_t rows, int64_t cols, double* dst, int64_t ldd) {
  if (!h) return -1;
  if (!src || lds < cols) return -2;
  if (rows < 0 || cols < 0) return -4;
  if (!dst || ldd < rows) return -6;
  GPP_TRY(gpp_launch_transpose(h->stream, src, lds, rows, cols, dst, ldd));
  return 0;
}

int gpp_syrk_rows(gpp_handle_t h, const double* Urow, int64_t ldu, double* C, int64_t ldc, int64_t Nt, int64_t K, int64_t nb,
                  int64_t first_block, int rank, int nranks) {
  if (!h) return -1;
  if (!Urow || !aligned16(Urow) || (ldu & 1)) return -2;
  if (!C || !aligned16(C) || (ldc & 1)) return -4;
  if (Nt < 0 || K < 0) return -6;
  if (nb < NBLK || nb % NBLK != 0) return -8;
  if (first_block < 0) return -9;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -10;
  if (Nt == 0 || K == 0) return 0;
  GemmArgs g = mk(Urow, ldu, Urow, ldu, C, ldc, Nt, Nt, K, -1.0, 1.0);
  g.c_lower = 2;
  g.own_mod = nranks;
  g.own_bt = (int)(nb / NBLK);
  g.own_off = (int)(((first_block - rank) % nranks + nranks) % nranks);
  GPP_TRY(gpp_launch_gemm(h->stream, 2, g, 1, NBLK, NBLK));
  return 0;
}

int gpp_gemm_lower_cols(gpp_handle_t h, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                        int64_t M, int64_t K, double alpha, double beta, int64_t nb, int64_t first_block, int rank, int nranks,
                        int64_t row0, int64_t row1, int compact) {
  if (!h) return -1;
  if (!A || !aligned16(A) || (lda & 1)) return -2;
  if (!B || !aligned16(B) || (ldb & 1)) return -4;
  if (!C || !aligned16(C) || (ldc & 1)) return -6;
  if (M < 0 || K < 0) return -8;
  if (nb < NBLK || nb % NBLK != 0) return -12;
  if (first_block < 0) return -13;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -14;
  if (row0 < 0 || row1 < row0 || row1 > M || row0 % NBLK != 0) return -16;
  if (M == 0 || K == 0 || row1 == row0) return 0;
  GemmArgs g = mk(A, lda, B, ldb, C, ldc, M, M, K, alpha, beta);
  g.c_lower = 1;
  g.row_t0 = (int)(row0 / NBLK);
  g.row_t1 = (int)((row1 + NBLK - 1) / NBLK);
  g.own_mod = nranks;
  g.own_bt = (int)(nb / NBLK);
  g.own_off = (int)(((first_block - rank) % nranks + nranks) % nranks);
  g.compact_bc = compact ? 1 : 0;
  GPP_TRY(gpp_launch_gemm(h->stream, 2, g, 1, NBLK, NBLK));
  return 0;
}

int gpp_trmv_lower_cols(gpp_handle_t h, const double* T, int64_t ldt, int64_t N, const double* x, double* y, int64_t nb, int rank,
                        int nranks, int trans, int compact) {
  if (!h) return -1;
  if (N < 0) return -4;
  if (int q = check_mat(T, ldt, compact ? std::min<int64_t>(N, ldt) : N, 2)) return q;
  if (!x || !aligned16(x)) return -5;
  if (!y) return -6;
  if (nb < 64 || nb % 64 != 0) return -7;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -8;
  if (trans != 0 && trans != 1) return -10;
  if (trans && (!h->ws || h->ws_bytes < gpp_trmv_t_ws_bytes(N))) return -1;  // workspace of GPP_OP_MLL_EVAL (gpp_set_workspace)
  if (compact && nranks >= 1 && rank >= 0 && ldt < owned_cols(N, nb, rank, nranks)) return -3;
  GPP_TRY(gpp_launch_trmv_lower_cols(h->stream, T, ldt, N, x, y, nb, rank, nranks, trans, h->ws, h->ws_bytes, compact));
  return 0;
}

int gpp_mll_scalars(gpp_handle_t h, const double* U, int64_t ld, int64_t N, const double* z, double* out3) {
  if (!h) return -1;
  if (N < 0) return -4;
  if (int q = check_mat(U, ld, N, 2)) return q;
  if (!z) return -5;
  if (!out3) return -6;
  GPP_TRY(gpp_launch_mll_scalars(h->stream, U, ld, N, z, out3));
  return 0;
}

int gpp_mll_reduce(gpp_handle_t h, const double* U, int64_t ld, const double* Linv, int64_t ldi, int64_t N,
                   const double* r, double* z, double* out3) {
  if (!h) return -1;
  if (N < 0) return -6;
  if (int q = check_mat(U, ld, N, 2)) return q;
  if (int q = check_mat(Linv, ldi, N, 4)) return q;
  if (!r || !aligned16(r)) return -7;
  if (!z) return -8;
  if (!out3) return -9;
  GPP_TRY(gpp_launch_trmv_lower(h->stream, Linv, ldi, N, r, z));
  GPP_TRY(gpp_launch_mll_scalars(h->stream, U, ld, N, z, out3));
  return 0;
}

int gpp_alpha(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t N, const double* z, double* alpha) {
  if (!h) return -1;
  if (N < 0) return -4;
  if (int q = check_mat(Linv, ldi, N, 2)) return q;
  if (!z || !aligned16(z)) return -5;
  if (!alpha) return -6;
  // alpha_j = sum_{i>=j} Linv[i][j] z_i = row j of the mirror (upper triangle of the buffer) times z
  GPP_TRY(gpp_launch_trmv_upper(h->stream, Linv, ldi, N, z, alpha));
  return 0;
}

int gpp_grad_reduce(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                    const int32_t* grp, int S, int kind, int d_split, const double* alpha, const double* Kinv,
                    int64_t ldk, int dU, double* g_w, double* g_sf2, double* g_tau, double* g_U) {
  if (!h) return -1;
  if (!U) return -2;
  if (N < 0) return -3;
  if (D < 1 || D > 64) return -4;
  if (!w) return -5;
  if (!sf2) return -6;
  if (S < 1 || S > 64) return -8;
  if (kind < 0 || kind > 2) return -9;
  if (d_split < 0 || d_split > D) return -10;
  if (!alpha) return -11;
  if (int q = check_mat(Kinv, ldk, N, 12)) return q;
  if (dU < 0 || dU > D) return -14;
  if (!g_w) return -15;
  if (!g_sf2) return -16;
  if (!g_tau) return -17;
  if (dU > 0 && !g_U) return -18;
  if (!h->ws || h->ws_bytes < gpp_grad_ws_bytes(N, D, S, dU)) return -1;
  GPP_TRY(gpp_launch_grad_reduce(h->stream, U, N, D, w, sf2, grp, S, kind, d_split, alpha, Kinv, ldk, dU, g_w, g_sf2, g_tau,
                                 g_U, h->ws, h->ws_bytes));
  return 0;
}

int gpp_grad_reduce_rows(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                         const int32_t* grp, int S, int kind, int d_split, const double* alpha, const double* Kinv,
                         int64_t ldk, int dU, int64_t nb, int rank, int nranks, double* g_w, double* g_sf2, double* g_tau,
                         double* g_U) {
  if (!h) return -1;
  if (!U) return -2;
  if (N < 0) return -3;
  if (D < 1 || D > 64) return -4;
  if (!w) return -5;
  if (!sf2) return -6;
  if (S < 1 || S > 64) return -8;
  if (kind < 0 || kind > 2) return -9;
  if (d_split < 0 || d_split > D) return -10;
  if (!alpha) return -11;
  if (int q = check_mat(Kinv, ldk, N, 12)) return q;
  if (dU < 0 || dU > D) return -14;
  if (nb < 64 || nb % 64 != 0 || nb > (1 << 30)) return -15;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -16;
  if (!g_w) return -18;
  if (!g_sf2) return -19;
  if (!g_tau) return -20;
  if (dU > 0 && !g_U) return -21;
  if (!h->ws || h->ws_bytes < gpp_grad_ws_bytes(N, D, S, dU)) return -1;
  GPP_TRY(gpp_launch_grad_reduce(h->stream, U, N, D, w, sf2, grp, S, kind, d_split, alpha, Kinv, ldk, dU, g_w, g_sf2, g_tau,
                                 g_U, h->ws, h->ws_bytes, (int)nb, rank, nranks));
  return 0;
}

int gpp_grad_reduce_cols(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                         const int32_t* grp, int S, int kind, int d_split, const double* alpha, const double* Kinv,
                         int64_t ldk, int dU, int64_t nb, int rank, int nranks, double* g_w, double* g_sf2, double* g_tau,
                         double* g_U, int compact) {
  if (!h) return -1;
  if (!U) return -2;
  if (N < 0) return -3;
  if (D < 1 || D > 64) return -4;
  if (!w) return -5;
  if (!sf2) return -6;
  if (S < 1 || S > 64) return -8;
  if (kind < 0 || kind > 2) return -9;
  if (d_split < 0 || d_split > D) return -10;
  if (!alpha) return -11;
  if (int q = check_mat(Kinv, ldk, compact ? std::min<int64_t>(N, ldk) : N, 12)) return q;
  if (dU < 0 || dU > D) return -14;
  if (nb < 64 || nb % 64 != 0 || nb > (1 << 30)) return -15;
  if (nranks < 1 || rank < 0 || rank >= nranks) return -16;
  if (compact && ldk < owned_cols(N, nb, rank, nranks)) return -13;
  if (!g_w) return -18;
  if (!g_sf2) return -19;
  if (!g_tau) return -20;
  if (dU > 0 && !g_U) return -21;
  if (!h->ws || h->ws_bytes < gpp_grad_ws_bytes(N, D, S, dU)) return -1;
  GPP_TRY(gpp_launch_grad_reduce(h->stream, U, N, D, w, sf2, grp, S, kind, d_split, alpha, Kinv, ldk, dU, g_w, g_sf2, g_tau,
                                 g_U, h->ws, h->ws_bytes, (int)nb, rank, nranks, 1, 0, 0, 0, /*shard_cols=*/compact ? 2 : 1));
  return 0;
}

int gpp_predict(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t N, const double* alpha, const double* Ksn,
                int64_t lds, int64_t M, const double* kss, double* V, int64_t ldv, double* mean_out, double* var_out) {
  if (!h) return -1;
  if (N < 0) return -4;
  if (int q = check_mat(Linv, ldi, N, 2)) return q;
  if (!alpha) return -5;
  if (!Ksn || !aligned16(Ksn) || (lds & 1) || lds < N) return -6;
  if (M < 0) return -8;
  if (!mean_out) return -12;
  if (V) {
    if (!aligned16(V) || (ldv & 1) || ldv < N) return -10;
    if (!kss) return -9;
    if (!var_out) return -13;
    // V = Ksn * Linv^T  (NT; Linv lower: k <= n)
    GemmArgs g = mk(Ksn, lds, Linv, ldi, V, ldv, M, N, N, 1.0, 0.0);
    g.b_mask = 1; g.khi_mode = 2;
    GPP_TRY(gpp_launch_gemm(h->stream, 0, g, 1));
  }
  GPP_TRY(gpp_launch_predict_reduce(h->stream, Ksn, lds, V, ldv, M, N, alpha, kss, mean_out, var_out));
  return 0;
}

int gpp_predict_tn(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t N, const double* z, const double* Kns, int64_t ldk,
                   int64_t M, const double* kss, double* V, int64_t ldv, double* mean_out, double* var_out) {
  if (!h) return -1;
  if (N < 0) return -4;
  if (int q = check_mat(Linv, ldi, N, 2)) return q;
  if (!z || !aligned16(z)) return -5;
  if (!Kns || !aligned16(Kns) || (ldk & 1) || ldk < M) return -6;
  if (M < 0) return -8;
  if (!kss) return -9;
  if (!V || !aligned16(V) || (ldv & 1) || ldv < N) return -10;
  if (!mean_out) return -12;
  if (!var_out) return -13;
  // V = Kns^T W with W = L^-T, the mirror in the upper triangle of the Linv buffer (keep k <= n): the row-contiguous TN product
  // of every other O(N^3) stage, where the [test][train] layout of gpp_predict needs the k-contiguous NT variant (1 work-group
  // per CU, no lean staging: 34 TFLOP/s on M N^2 against ~60 here)
  GemmArgs g = mk(Kns, ldk, Linv, ldi, V, ldv, M, N, N, 1.0, 0.0);
  g.b_mask = 1; g.khi_mode = 2;
  GPP_TRY(gpp_launch_gemm(h->stream, 2, g, 1));
  // mean_a = sum_j V[a][j] z_j (= K_*N alpha, alpha = L^-T z), var_a = kss_a - sum_j V[a][j]^2: one pass over V
  GPP_TRY(gpp_launch_predict_reduce(h->stream, V, ldv, V, ldv, M, N, z, kss, mean_out, var_out));
  return 0;
}

int gpp_gemm(gpp_handle_t h, int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
             int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int a_mask, int b_mask,
             int klo_mode, int khi_mode, int c_tri) {
  if (!h) return -1;
  int variant;
  if (transA == 0 && transB == 1) variant = 0;
  else if (transA == 0 && transB == 0) variant = 1;
  else if (transA == 1 && transB == 0) variant = 2;
  else return -2;
  if (M < 0 || N < 0 || K < 0) return -4;
  if (!A || !aligned16(A) || (lda & 1)) return -8;
  if (!B || !aligned16(B) || (ldb & 1)) return -10;
  if (!C || !aligned16(C) || (ldc & 1)) return -13;
  if (a_mask < 0 || a_mask > 2 || b_mask < 0 || b_mask > 2) return -15;
  if (klo_mode < 0 || klo_mode > 3 || khi_mode < 0 || khi_mode > 2) return -17;
  if (c_tri < 0 || c_tri > 2 || (c_tri && M != N)) return -19;
  GemmArgs g = mk(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta);
  g.a_mask = a_mask; g.b_mask = b_mask; g.klo_mode = klo_mode; g.khi_mode = khi_mode; g.c_lower = c_tri;
  int ftm = 0, ftn = 0;
  if (const char* e = getenv("GPP_GEMM_TILE")) sscanf(e, "%d,%d", &ftm, &ftn);  // dev knob (tools/attic/gemm_small_probe.py)
  if (getenv("GPP_GEMM_ON_UPD")) {  // dev knob (tools/attic/gemm_update_probe.py): run on the CU-masked update stream
    GPP_TRY(ensure_streams(h));
    hipEvent_t a = next_event(h), b = next_event(h);
    GPP_TRY(hipEventRecord(a, h->stream));
    GPP_TRY(hipStreamWaitEvent(h->upd_stream, a, 0));
    GPP_TRY(gpp_launch_gemm(h->upd_stream, variant, g, 1, ftm, ftn));
    GPP_TRY(hipEventRecord(b, h->upd_stream));
    GPP_TRY(hipStreamWaitEvent(h->stream, b, 0));
    return 0;
  }
  if (h->cu_split == 1 && h->stream == h->panel_stream) g.cu_hint = h->panel_cus;
  GPP_TRY(gpp_launch_gemm(h->stream, variant, g, 1, ftm, ftn));
  return 0;
}

int gpp_gemm_batched(gpp_handle_t h, int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                     int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB, double beta, double* C, int64_t ldc,
                     int64_t sC, int batch, int a_mask, int b_mask, int klo_mode, int khi_mode, int c_tri) {
  if (!h) return -1;
  int variant;
  if (transA == 0 && transB == 1) variant = 0;
  else if (transA == 0 && transB == 0) variant = 1;
  else if (transA == 1 && transB == 0) variant = 2;
  else return -2;
  if (M < 0 || N < 0 || K < 0) return -4;
  if (!A || !aligned16(A) || (lda & 1) || (sA & 1)) return -8;
  if (!B || !aligned16(B) || (ldb & 1) || (sB & 1)) return -11;
  if (!C || !aligned16(C) || (ldc & 1) || (sC & 1)) return -15;
  if (batch < 0 || batch > 65535) return -18;
  if (a_mask < 0 || a_mask > 2 || b_mask < 0 || b_mask > 2) return -19;
  if (klo_mode < 0 || klo_mode > 3 || khi_mode < 0 || khi_mode > 2) return -21;
  if (c_tri < 0 || c_tri > 2 || (c_tri && M != N)) return -23;
  if (batch == 0) return 0;
  GemmArgs g = mk(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta);
  g.a_mask = a_mask; g.b_mask = b_mask; g.klo_mode = klo_mode; g.khi_mode = khi_mode; g.c_lower = c_tri;
  g.sA = sA; g.sB = sB; g.sC = sC;
  if (h->cu_split == 1 && h->stream == h->panel_stream) g.cu_hint = h->panel_cus;
  GPP_TRY(gpp_launch_gemm(h->stream, variant, g, batch));
  return 0;
}

// ---- batched evaluation: `batch` independent problems of the same size in every launch ----------------------------
static int check_batch(int batch, int argi) { return (batch < 1 || batch > 65535) ? -argi : 0; }

int gpp_kernel_build_batched(gpp_handle_t h, const double* U, int64_t sU, int64_t N, int D, const double* w, const double* sf2,
                             const double* tau, const int32_t* grp, int S, double jitter, int kind, int d_split, int uplo,
                             double* Ky, int64_t ld, int64_t sK, int batch) {
  if (!h) return -1;
  if (!U) return -2;
  if (N < 0) return -4;
  if (D < 1 || D > 64) return -5;
  if (!w) return -6;
  if (!sf2) return -7;
  if (tau && S < 1) return -10;
  if (kind < 0 || kind > 2) return -12;
  if (d_split < 0 || d_split > D) return -13;
  if (uplo != GPP_UPLO_FULL && uplo != GPP_UPLO_LOWER && uplo != GPP_UPLO_UPPER) return -14;
  if (!Ky) return -15;
  if (ld < N || (sK & 1) || (sU != 0 && sU < N * D)) return -16;
  if (int r = check_batch(batch, 18)) return r;
  GPP_TRY(gpp_launch_kernel_build(h->stream, U, N, D, w, sf2, tau, grp, S, jitter, kind, d_split, uplo, Ky, ld, 0, N, batch, sU, sK));
  return 0;
}

int gpp_potrf_batched(gpp_handle_t h, double* A, int64_t N, int64_t ld, int64_t sA, double* Linv, int64_t ldi, int64_t sLi,
                      int32_t* info_dev, int batch) {
  if (!h) return -1;
  if (N < 0 || N > BLK_MAX) return -3;  // the batched path is the leaf-step factorisation: small and medium N only
  if (int r = check_mat(A, ld, N, 2)) return r;
  if (int r = check_mat(Linv, ldi, N, 6)) return r;
  if ((sA & 1) || (sLi & 1)) return -5;
  if (!info_dev) return -9;
  if (int r = check_batch(batch, 10)) return r;
  GPP_TRY(gpp_launch_fill_i32(h->stream, info_dev, batch, 0));
  Ctx c{h->stream, A, ld, Linv, ldi, info_dev};
  GPP_TRY(potrf_blk_batched(c, N, batch, sA, sLi));
  return 0;
}

int gpp_trtri_batched(gpp_handle_t h, const double* U, int64_t N, int64_t ld, int64_t sA, double* Linv, int64_t ldi,
                      int64_t sLi, double* T, int64_t ldt, int64_t sT, int batch) {
  if (!h) return -1;
  if (N < 0) return -3;
  if (int r = check_mat(U, ld, N, 2)) return r;
  if (int r = check_mat(Linv, ldi, N, 6)) return r;
  if (int r = check_mat(T, ldt, N, 9)) return r;
  if ((sA & 1) || (sLi & 1) || (sT & 1)) return -5;
  if (int r = check_batch(batch, 12)) return r;
  for (int64_t s = NBLK; s < N; s *= 2)
    GPP_TRY(trtri_level(h->stream, U, ld, Linv, ldi, T, ldt, 0, N, s, [](int64_t) { return false; }, batch, sA, sLi, sT));
  return 0;
}

int gpp_lauum_batched(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, int64_t sLi, double* Kinv, int64_t ldk,
                      int64_t sK, int batch) {
  if (!h) return -1;
  if (N < 0) return -3;
  if (int r = check_mat(Linv, ldi, N, 2)) return r;
  if (int r = check_mat(Kinv, ldk, N, 6)) return r;
  if ((sLi & 1) || (sK & 1)) return -5;
  if (int r = check_batch(batch, 9)) return r;
  GemmArgs g = mk(Linv, ldi, Linv, ldi, Kinv, ldk, N, N, N, 1.0, 0.0);
  g.a_mask = 2; g.b_mask = 2; g.klo_mode = 3; g.c_lower = 1;
  g.sA = sLi; g.sB = sLi; g.sC = sK;
  GPP_TRY(gpp_launch_gemm(h->stream, 2, g, batch));
  return 0;
}

int gpp_mll_reduce_batched(gpp_handle_t h, const double* U, int64_t ld, int64_t sA, const double* Linv, int64_t ldi,
                           int64_t sLi, int64_t N, const double* r, double* z, int64_t sv, double* out3, int batch) {
  if (!h) return -1;
  if (N < 0) return -8;
  if (int q = check_mat(U, ld, N, 2)) return q;
  if (int q = check_mat(Linv, ldi, N, 5)) return q;
  if (!r || !aligned16(r)) return -9;
  if (!z || !aligned16(z)) return -10;
  if (sv < N || (sv & 1)) return -11;  // rows of r / z / alpha start at b*sv: 16-byte aligned
  if (!out3) return -12;
  if (int q = check_batch(batch, 13)) return q;
  GPP_TRY(gpp_launch_trmv_lower(h->stream, Linv, ldi, N, r, z, batch, sLi, sv));
  GPP_TRY(gpp_launch_mll_scalars(h->stream, U, ld, N, z, out3, batch, sA, sv));
  return 0;
}

int gpp_alpha_batched(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t sLi, int64_t N, const double* z, double* alpha,
                      int64_t sv, int batch) {
  if (!h) return -1;
  if (N < 0) return -5;
  if (int q = check_mat(Linv, ldi, N, 2)) return q;
  if (!z || !aligned16(z)) return -6;
  if (!alpha || !aligned16(alpha)) return -7;
  if (sv < N || (sv & 1)) return -8;
  if (int q = check_batch(batch, 9)) return q;
  GPP_TRY(gpp_launch_trmv_upper(h->stream, Linv, ldi, N, z, alpha, batch, sLi, sv));
  return 0;
}

int gpp_grad_reduce_batched(gpp_handle_t h, const double* U, int64_t sU, int64_t N, int D, const double* w, const double* sf2,
                            const int32_t* grp, int S, int kind, int d_split, const double* alpha, int64_t sv,
                            const double* Kinv, int64_t ldk, int64_t sK, int dU, double* g_w, double* g_sf2, double* g_tau,
                            double* g_U, int batch) {
  if (!h) return -1;
  if (!U) return -2;
  if (N < 0) return -4;
  if (D < 1 || D > 64) return -5;
  if (!w) return -6;
  if (!sf2) return -7;
  if (S < 1 || S > 64) return -9;
  if (kind < 0 || kind > 2) return -10;
  if (d_split < 0 || d_split > D) return -11;
  if (!alpha) return -12;
  if (int q = check_mat(Kinv, ldk, N, 13)) return q;
  if (dU < 0 || dU > D) return -16;
  if (!g_w) return -17;
  if (!g_sf2) return -18;
  if (!g_tau) return -19;
  if (dU > 0 && !g_U) return -20;
  if (int q = check_batch(batch, 22)) return q;
  if (sv < N) return -13;
  if (!h->ws || h->ws_bytes < (size_t)batch * gpp_grad_ws_bytes(N, D, S, dU)) return -1;
  GPP_TRY(gpp_launch_grad_reduce(h->stream, U, N, D, w, sf2, grp, S, kind, d_split, alpha, Kinv, ldk, dU, g_w, g_sf2, g_tau,
                                 g_U, h->ws, h->ws_bytes, 0, 0, 1, batch, sU, sK, sv));
  return 0;
}

}  // extern "C"
