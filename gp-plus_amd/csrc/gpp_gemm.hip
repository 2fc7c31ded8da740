// gpp_gemm.hip — fp64 MFMA GEMM for gfx950 (MI355X): C = beta*C + alpha*op(A)*op(B) with triangular operand
// masks, per-tile K ranges and lower-only output.  It carries every O(N^3) flop of the exact-GP path:
//   - TRSM / SYRK / GEMM updates of the blocked Cholesky  (replaces torch.linalg.cholesky_ex reached from
//     gpytorch psd_safe_cholesky, reference call site optim/mll_torch.py:116)
//   - TRMM pair products of the bottom-up triangular inverse and the LAUUM product Linv^T Linv
//     (replace ATen cholesky_backward, reference call site optim/mll_torch.py:117)
//   - V = K_*N Linv^T of the prediction path (models/gpregression.py:122-149).
//
// Design (CDNA4, measured on MI355X): v_mfma_f64_16x16x4_f64 — lane l supplies A[row l%16][k l/16] and B[k l/16][col l%16] and
// holds D[row (l/16) + 4v][col l%16] in element v of its 4-double accumulator (tools/attic/mfma16_layout.hip).  A bare loop of the
// instruction sustains 69.6 TFLOP/s chip-wide, the rate rocBLAS's MI16x16x4 DGEMM kernels reach too (72.9); the round-1 kernel
// used the 4x4x4_4b form on the strength of a probe that had measured hipcc's AGPR copies (profiles/r02_mfma_f64_16x16x4.txt).
// A (2*WT)^2 output tile per 256-thread work-group, 2x2 waves; each wave owns WT x WT = (WT/16)^2 blocks of 16 x 16
// (WT=64: 64 accumulator doubles per lane, in VGPRs: __launch_bounds__(256, 2)).  K is consumed in chunks of 16;
// both operand chunks are staged in LDS in [k][row] order (padded strides, see ldt_*), double-buffered: the global
// loads of chunk c+1 are issued before the MFMAs of chunk c and written to the other buffer afterwards, one barrier
// per chunk.  WT = 32 / 16 variants (64^2 / 32^2 tiles) and a 128 x 32 tile serve the small sub-problems, where the
// grid of 128^2 tiles would leave most of the 256 CUs idle; they stage K in chunks of 64 with one LDS buffer (template
// parameters BK, NBUF below).  Work-groups are independent along two batch dimensions (grid.y, grid.z).
#include "../../include/gpp.h"
#include "gpp_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK16 = 16;  // K chunk of the big tile and of every k-contiguous ("KC") operand
// LDS layouts.  Row-contiguous operands ("MC", stored [k][row] in memory) are staged as [k][T+16]: 16-byte aligned
// rows for ds_write_b128, and the two k-rows a 32-lane ds_read_b64 group touches fall in different bank halves.
// k-contiguous operands ("KC", stored [row][k]) are staged UNtransposed as [row][18]: every thread writes its 16-byte
// vector with one ds_write_b128 (8 lanes = one 128-byte row), and with the 144-byte row stride both fragment reads
// are conflict free: A lanes (i=0..3, k, k+1) hit slots {18i + k}, B lanes (c=0..15, k, k+1) hit 18c + k mod 32,
// which enumerates all 32 8-byte slots.
constexpr int ldt_mc(int T, int BK = 16) { return BK == 16 ? T + 16 : T + 4; }
constexpr int LDK = BK16 + 2;  // [row][k] chunks of k-contiguous operands: 144-byte rows, see below

// Staging is split in two so that the global loads of chunk c+1 stay in flight across the MFMAs of chunk c:
//   load_*  : computes the keep-predicates (range + triangular mask; no loaded data involved) and issues one
//             branch-free 16-byte load per vector (from P itself when the element is out of range).  Reading one
//             double past kend/R stays inside the allocation: ld is even and >= the extent (gpp.h).
//   store_* : after the MFMAs, zeroes the dropped elements by select and writes the chunk to LDS.
__device__ __forceinline__ bool keep_elem(int mask, int k, int row) {
  // bitwise on purpose: no short-circuit branches between the loads of a chunk (a branch ends the basic block, and
  // hipcc then waits for the loads issued so far: the 20 vectors of a 128-row chunk arrived one L2 trip at a time)
  return (mask == 0) | ((mask == 1) & (k <= row)) | ((mask == 2) & (k >= row));
}

// Operand stored [row][k] (k contiguous): T rows x 16 k per chunk, T/32 16-byte vectors per thread.
template <int T>
__device__ __forceinline__ unsigned load_kc(const double* __restrict__ P, int64_t ld, int R, int r0, int kb, int kend,
                                            int mask, int tid, v2d (&reg)[T / 32]) {
  unsigned keep = 0;
  const double* ptr[T / 32];
#pragma unroll
  for (int i = 0; i < T / 32; ++i) {
    const int v = tid + 256 * i;
    const int gr = r0 + (v >> 3);
    const int gk = kb + ((v & 7) << 1);
    const bool v0 = (gr < R) & (gk < kend);
    const bool k0 = v0 & keep_elem(mask, gk, gr);
    const bool k1 = v0 & (gk + 1 < kend) & keep_elem(mask, gk + 1, gr);
    keep |= (k0 ? 1u : 0u) << (2 * i);
    keep |= (k1 ? 1u : 0u) << (2 * i + 1);
    ptr[i] = v0 ? P + (int64_t)gr * ld + gk : P;
  }
#pragma unroll
  for (int i = 0; i < T / 32; ++i) reg[i] = *reinterpret_cast<const v2d*>(ptr[i]);
  return keep;
}
template <int T, bool SEL>
__device__ __forceinline__ void store_kc(double* __restrict__ s, int tid, const v2d (&reg)[T / 32], unsigned keep) {
#pragma unroll
  for (int i = 0; i < T / 32; ++i) {
    const int v = tid + 256 * i;
    v2d t = reg[i];
    if (SEL) {
      t.x = ((keep >> (2 * i)) & 1u) ? t.x : 0.0;
      t.y = ((keep >> (2 * i + 1)) & 1u) ? t.y : 0.0;
    }
    *reinterpret_cast<v2d*>(s + (v >> 3) * LDK + ((v & 7) << 1)) = t;
  }
}

// Operand stored [k][row] (row contiguous): BK k x T rows per chunk, BK * T / 512 16-byte vectors per thread.
// (`dummy`: where the lanes of out-of-range elements read instead — any readable address; P itself unless P is a biased base,
//  see GemmArgs::compact_bc)
template <int T, int BK, int NTH = 256>
__device__ __forceinline__ unsigned load_mc(const double* __restrict__ P, int64_t ld, int R, int r0, int kb, int kend,
                                            int mask, int tid, v2d (&reg)[BK * T / (2 * NTH)], const double* __restrict__ dummy) {
  constexpr int NV = BK * T / (2 * NTH);
  unsigned keep = 0;
  const double* ptr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + NTH * i;
    const int gk = kb + v / (T / 2);
    const int gr = r0 + ((v % (T / 2)) << 1);
    const bool v0 = (gk < kend) & (gr < R);
    const bool k0 = v0 & keep_elem(mask, gk, gr);
    const bool k1 = v0 & (gr + 1 < R) & keep_elem(mask, gk, gr + 1);
    keep |= (k0 ? 1u : 0u) << (2 * i);
    keep |= (k1 ? 1u : 0u) << (2 * i + 1);
    ptr[i] = v0 ? P + (int64_t)gk * ld + gr : dummy;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) reg[i] = *reinterpret_cast<const v2d*>(ptr[i]);  // all requests back to back
  return keep;
}
template <int T, int BK, bool SEL, int NTH = 256>
__device__ __forceinline__ void store_mc(double* __restrict__ s, int tid, const v2d (&reg)[BK * T / (2 * NTH)], unsigned keep) {
#pragma unroll
  for (int i = 0; i < BK * T / (2 * NTH); ++i) {
    const int v = tid + NTH * i;
    v2d t = reg[i];
    if (SEL) {
      t.x = ((keep >> (2 * i)) & 1u) ? t.x : 0.0;
      t.y = ((keep >> (2 * i + 1)) & 1u) ? t.y : 0.0;
    }
    *reinterpret_cast<v2d*>(s + (v / (T / 2)) * ldt_mc(T, BK) + ((v % (T / 2)) << 1)) = t;
  }
}

// ---- interior fast path --------------------------------------------------------------------------------
// For a chunk whose 16 k's and T rows are all in range and untouched by the triangular mask (a work-group-uniform
// test), staging needs no predicates at all: one load per vector from  uniform_base + per-thread 32-bit byte offset
// (SGPR-base addressing), and plain LDS stores.  This removes ~3/4 of the VALU instructions that otherwise compete
// with the MFMAs for the SIMD's issue port.
template <int T, int BK>
__device__ __forceinline__ bool chunk_is_interior(int r0, int R, int kb, int kend, int mask) {
  if (r0 + T > R || kb + BK > kend) return false;
  if (mask == 1) return kb + BK - 1 <= r0;      // keep k <= row holds for every row >= r0
  if (mask == 2) return kb >= r0 + T - 1;       // keep k >= row holds for every row <  r0 + T
  return true;
}
// In-range chunk of a row-contiguous operand that the triangular mask cuts: same lean loads as an interior chunk, the
// keep bits come from the indices alone.
template <int T, int BK>
__device__ __forceinline__ bool chunk_in_range(int r0, int R, int kb, int kend) {
  return r0 + T <= R && kb + BK <= kend;
}
template <int T, int BK, int NTH = 256>
__device__ __forceinline__ unsigned mask_bits_mc(int r0, int kb, int mask, int tid) {
  unsigned keep = 0;
#pragma unroll
  for (int i = 0; i < BK * T / (2 * NTH); ++i) {
    const int v = tid + NTH * i;
    const int gk = kb + v / (T / 2);
    const int gr = r0 + ((v % (T / 2)) << 1);
    keep |= (keep_elem(mask, gk, gr) ? 1u : 0u) << (2 * i);
    keep |= (keep_elem(mask, gk, gr + 1) ? 1u : 0u) << (2 * i + 1);
  }
  return keep;
}
template <int T, bool KC, int BK, int NTH = 256>
__device__ __forceinline__ void thread_offsets(int64_t ld, int tid, unsigned (&off)[BK * T / (2 * NTH)]) {
#pragma unroll
  for (int i = 0; i < BK * T / (2 * NTH); ++i) {
    const int v = tid + NTH * i;
    if (KC) off[i] = (unsigned)(((int64_t)(v >> 3) * ld + ((v & 7) << 1)) * 8);
    else off[i] = (unsigned)(((int64_t)(v / (T / 2)) * ld + ((v % (T / 2)) << 1)) * 8);
  }
}
template <int NV>
__device__ __forceinline__ void load_fast(const double* __restrict__ ubase, const unsigned (&off)[NV], v2d (&reg)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i)
    reg[i] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(ubase) + off[i]);
}

// The row-contiguous TN variant (every hot product of the evaluation) keeps 2 work-groups per CU (<= 256 VGPRs);
// variants with a k-contiguous operand are off the hot path (prediction, tests) and take the registers they need.
// TAG only changes the kernel's NAME: the single N^3/3-flop LAUUM launch runs as <2,64,64,1> so that profilers report
// it on its own line (the roofline entry of bench.py), apart from the ~1300 launches of the recursions.
// BK = K chunk, NBUF = LDS buffers per operand.  The big tile runs (16, 2): its MFMA phase (64 accumulators per lane,
// 2 work-groups per CU) covers a load round trip and one barrier per chunk suffices.  The small tiles of the recursions'
// leaves have almost no MFMA work per 16-wide chunk, so every chunk cost one full L2 round trip (~0.6 us: a K = 128
// product took 8-20 us); they run (64, 1): four times fewer round trips, the next chunk's loads in flight in
// registers during the MFMAs, two barriers per chunk.  (k-contiguous operands, VAR != 2, only exist with BK = 16.)
// WR = rows of waves in the work-group: 2 (256 threads, 2 x 2 waves: every instantiation of the evaluation) or 4 (512 threads, 4 x 2
// waves, a 256 x 128 tile on ONE work-group per CU: the experiment of DESIGN.md section 3.5 — 25 % fewer operand bytes per flop).
// One output tile (tm, tn) of the product described by ``p``: the body shared by the launch-per-product kernel (gpp_gemm_f64, p in
// the kernel arguments) and the DAG executor (gpp_dag_f64, p in a device array read through the constant address
// space: scalar loads, re-materialisable like kernel arguments).  A, B, C: the batch element's operands,
// C2: its mirrored output (or null).  Ends with a work-group barrier after the last LDS read, so the caller may stage another tile at once.
struct GemmNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// (`before_epilogue`: called by every thread after the last MFMA and before the first access to C — the DAG executor issues its
//  next ticket's atomic there, so that its latency runs under the epilogue)
template <int VAR, int WTM, int WTN, int TAG, int BK, int NBUF, int WR, class P, class H = GemmNoHook>
__device__ __forceinline__ void gemm_tile(const P& p, const int tm, const int tn, const double* __restrict__ A,
                                          const double* __restrict__ B, double* __restrict__ C, double* C2,
                                          double* __restrict__ smem, H before_epilogue = H()) {
  static_assert(BK == 16 || VAR == 2, "wide K chunks are implemented for row-contiguous (TN) operands only");
  static_assert(BK % 16 == 0 && (NBUF == 1 || NBUF == 2), "bad staging parameters");
  static_assert(WR == 2 || (WR == 4 && VAR == 2 && BK == 16 && NBUF == 2 && WTM == 64 && WTN == 64), "the tall tile is TN only");
  constexpr int NTH = 128 * WR;
  constexpr bool A_KC = (VAR != 2);
  constexpr bool B_KC = (VAR == 0);
  constexpr int TM = WR * WTM, TN = 2 * WTN;  // work-group tile: TM rows x TN columns (WR x 2 waves)
  constexpr int LDA = ldt_mc(TM, BK), LDB = ldt_mc(TN, BK);  // strides of [k][row] chunks (MC operands)
  constexpr int TX = TM > TN ? TM : TN;
  // doubles per staged chunk: one size for both operands of the square tiles, the two true sizes for the tall one
  constexpr int OPSZ = (BK * ldt_mc(TX, BK) > TX * LDK) ? BK * ldt_mc(TX, BK) : TX * LDK;
  constexpr int OPA = (WR == 2) ? OPSZ : BK * LDA, OPB = (WR == 2) ? OPSZ : BK * LDB;
  constexpr int NVA = BK * TM / (2 * NTH), NVB = BK * TN / (2 * NTH);  // 16-byte vectors per thread and chunk
  constexpr int RB = WTM / 4, CB = WTN / 16;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WTN;
  const int li = lane & 15, lk = lane >> 4;
  const int row0 = tm * TM, col0 = tn * TN;
  const double* __restrict__ Bd = B;  // readable addresses for the lanes that must not touch memory (the biased bases below may
  double* __restrict__ Cd = C;        // lie in front of their buffers)
  if (p.compact_bc) {
    // B and C store only the owned column blocks, side by side: shift their bases by (physical - logical) first column of this
    // tile's block (a tile never straddles two blocks: the block width is a multiple of the tile)
    const int b = tn / p.own_bt, q = (b + p.own_off) / p.own_mod;
    const int64_t bias = (int64_t)(q - b) * p.own_bt * TN;
    B += bias;
    C += bias;
  }

  int klo = 0;
  if (p.klo_mode == 1) klo = row0;
  else if (p.klo_mode == 2) klo = col0;
  else if (p.klo_mode == 3) klo = row0 > col0 ? row0 : col0;
  int khi = p.K;
  if (p.khi_mode == 1) khi = min(p.K, row0 + TM);
  else if (p.khi_mode == 2) khi = min(p.K, col0 + TN);
  const int nch = khi > klo ? (khi - klo + BK - 1) / BK : 0;

  // accumulators: acc4[a4][b] element v is C[row0 + wm + 16 a4 + 4 v + (l>>4)][col0 + wn + 16 b + (l&15)] — the 16 x 16 result
  // block of v_mfma_f64_16x16x4_f64 (lane l, element v: row (l>>4) + 4 v, column l&15; tools/attic/mfma16_layout.hip)
  static_assert(RB % 4 == 0, "wave tile rows must be a multiple of 16");
  v4d acc4[RB / 4][CB];
#pragma unroll
  for (int a = 0; a < RB / 4; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b) acc4[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  v2d ra[NVA], rb[NVB];
  unsigned ka = 0, kb_ = 0;
  unsigned offa[NVA], offb[NVB];
  thread_offsets<TM, A_KC, BK, NTH>(p.lda, tid, offa);
  thread_offsets<TN, B_KC, BK, NTH>(p.ldb, tid, offb);
  // uniform bases of the tile's first chunk row/column block; advanced by a scalar per chunk
  const double* __restrict__ ubaseA = A_KC ? A + (int64_t)row0 * p.lda : A + row0;
  const double* __restrict__ ubaseB = B_KC ? B + (int64_t)col0 * p.ldb : B + col0;
  const int64_t stepA = A_KC ? 1 : p.lda, stepB = B_KC ? 1 : p.ldb;  // elements per unit of k

  // extents for the LOADS (see GemmArgs::pad_ok); the stores below always use the true M and N
  const int Mld = (VAR == 2 && p.pad_ok) ? ((p.M + TM - 1) / TM) * TM : p.M;
  const int Nld = (VAR == 2 && p.pad_ok) ? ((p.N + TN - 1) / TN) * TN : p.N;
  // Per operand and chunk, work-group uniform: 0 = edge (predicated loads, select on store), 1 = interior (lean loads,
  // plain stores), 2 = in range but cut by the triangular mask (lean loads, select on store).  The lean paths exist for
  // the row-contiguous TN variant only: with a k-contiguous operand the extra live registers push hipcc over the
  // 256-VGPR budget of 2 waves/SIMD and the spills cost more than the VALU saved.
  int pa = 0, pb = 0;
  auto stage_load = [&](int kb) {
    pa = pb = 0;
    if constexpr (VAR == 2) {
      if (chunk_in_range<TM, BK>(row0, Mld, kb, khi)) pa = chunk_is_interior<TM, BK>(row0, Mld, kb, khi, p.a_mask) ? 1 : 2;
      if (chunk_in_range<TN, BK>(col0, Nld, kb, khi)) pb = chunk_is_interior<TN, BK>(col0, Nld, kb, khi, p.b_mask) ? 1 : 2;
    }
    if (pa) {
      load_fast<NVA>(ubaseA + (int64_t)kb * stepA, offa, ra);
      if (pa == 2) ka = mask_bits_mc<TM, BK, NTH>(row0, kb, p.a_mask, tid);
    } else {
      if constexpr (A_KC) ka = load_kc<TM>(A, p.lda, p.M, row0, kb, khi, p.a_mask, tid, ra);
      else ka = load_mc<TM, BK, NTH>(A, p.lda, p.M, row0, kb, khi, p.a_mask, tid, ra, A);
    }
    if (pb) {
      load_fast<NVB>(ubaseB + (int64_t)kb * stepB, offb, rb);
      if (pb == 2) kb_ = mask_bits_mc<TN, BK, NTH>(col0, kb, p.b_mask, tid);
    } else {
      if constexpr (B_KC) kb_ = load_kc<TN>(B, p.ldb, p.N, col0, kb, khi, p.b_mask, tid, rb);
      else kb_ = load_mc<TN, BK, NTH>(B, p.ldb, p.N, col0, kb, khi, p.b_mask, tid, rb, Bd);
    }
  };
  auto stage_store = [&](double* da, double* db) {
    if (pa == 1) {
      if constexpr (A_KC) store_kc<TM, false>(da, tid, ra, 0); else store_mc<TM, BK, false, NTH>(da, tid, ra, 0);
    } else {
      if constexpr (A_KC) store_kc<TM, true>(da, tid, ra, ka); else store_mc<TM, BK, true, NTH>(da, tid, ra, ka);
    }
    if (pb == 1) {
      if constexpr (B_KC) store_kc<TN, false>(db, tid, rb, 0); else store_mc<TN, BK, false, NTH>(db, tid, rb, 0);
    } else {
      if constexpr (B_KC) store_kc<TN, true>(db, tid, rb, kb_); else store_mc<TN, BK, true, NTH>(db, tid, rb, kb_);
    }
  };

  // chunk c covers k in [kpos(c), kpos(c)+BK).  With k_reverse the chunks run from the top of the range down, so that
  // tiles whose ranges END together (klo differs per column tile, e.g. X^T * lower-triangular) sweep the shared
  // operand in lockstep and hit in L2 instead of each streaming its own k rows.
  auto kpos = [&](int c) { return p.k_reverse ? klo + (nch - 1 - c) * BK : klo + c * BK; };
  if (nch > 0) {
    stage_load(kpos(0));
    stage_store(smem, smem + OPA);
  }
  __syncthreads();

  auto compute = [&](int cur) {
    // v_mfma_f64_16x16x4_f64: lane (i = l&15, k = l>>4) supplies A[row i][k] and B[k][col i]; one instruction yields a
    // 16 x 16 block of C.  Per k-step of 4 a 64 x 64 wave tile takes 4 + 4 fragment reads for 16 MFMAs.
    const double* sa = smem + cur * (OPA + OPB) + (A_KC ? (wm + li) * LDK + lk : wm + li + lk * LDA);
    const double* sb = smem + cur * (OPA + OPB) + OPA + (B_KC ? (wn + li) * LDK + lk : wn + li + lk * LDB);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double bf[CB], af[RB / 4];
#pragma unroll
      for (int b = 0; b < CB; ++b) bf[b] = B_KC ? sb[16 * b * LDK + kk * 4] : sb[kk * 4 * LDB + 16 * b];
#pragma unroll
      for (int a = 0; a < RB / 4; ++a) af[a] = A_KC ? sa[16 * a * LDK + kk * 4] : sa[kk * 4 * LDA + 16 * a];
#pragma unroll
      for (int a = 0; a < RB / 4; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b)
          acc4[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc4[a][b], 0, 0, 0);
    }
  };

  // Chunks whose loads are interior for BOTH operands form one contiguous range of k (each operand's condition is an
  // interval in kb).  Iterations that prefetch such a chunk run a branch-free body (lean loads, plain stores): one basic
  // block, so the compiler can spread the memory instructions over the MFMAs and the wave spends no issue slots on
  // the path selection (~100 scalar instructions per chunk).
  int ca = 0, cb = 0;  // clean iterations: ca <= c < cb
  // (big tile only: the small tiles run a handful of chunks per launch and lose more to the second loop's code)
  constexpr bool SPLIT = (VAR == 2 && RB == 16 && CB == 4 && BK == 16);
  if constexpr (SPLIT) {
    if (row0 + TM <= Mld && col0 + TN <= Nld && nch > 1) {
      int k_lo = klo, k_hi = khi;  // chunk [kb, kb+BK) is interior iff k_lo <= kb and kb + BK <= k_hi
      if (p.a_mask == 1) k_hi = min(k_hi, row0 + 1);
      if (p.a_mask == 2) k_lo = max(k_lo, row0 + TM - 1);
      if (p.b_mask == 1) k_hi = min(k_hi, col0 + 1);
      if (p.b_mask == 2) k_lo = max(k_lo, col0 + TN - 1);
      // chunk positions are klo + j*BK, j = c (forward) or nch-1-c (reverse)
      int j_lo = (k_lo - klo + BK - 1) / BK, j_hi = (k_hi - klo) / BK;  // interior positions j_lo <= j < j_hi
      j_lo = max(j_lo, 0);
      j_hi = min(j_hi, nch);
      if (j_hi > j_lo) {
        const int ci_lo = p.k_reverse ? nch - j_hi : j_lo, ci_hi = p.k_reverse ? nch - j_lo : j_hi;  // chunk indices
        ca = max(ci_lo - 1, 0);          // iteration c prefetches chunk c + 1
        cb = min(ci_hi - 1, nch - 1);
        if (cb < ca) ca = cb = 0;
      }
    }
  }
  auto general_iter = [&](int c) {
    const int cur = (NBUF == 2) ? (c & 1) : 0;
    const bool more = (c + 1 < nch);
    if (more) stage_load(kpos(c + 1));
    compute(cur);
    if (NBUF == 1) __syncthreads();  // everyone has read this chunk: the single buffer may take the next one
    if (more) {
      const int nxt = (NBUF == 2) ? (cur ^ 1) : 0;
      stage_store(smem + nxt * (OPA + OPB), smem + nxt * (OPA + OPB) + OPA);
    }
    __syncthreads();
  };
  int c = 0;
  for (; c < ca; ++c) general_iter(c);
  if constexpr (SPLIT) {
    for (; c < cb; ++c) {
      const int cur = (NBUF == 2) ? (c & 1) : 0, nxt = (NBUF == 2) ? (cur ^ 1) : 0;
      const int kb = kpos(c + 1);
      load_fast<NVA>(ubaseA + (int64_t)kb * stepA, offa, ra);
      load_fast<NVB>(ubaseB + (int64_t)kb * stepB, offb, rb);
      // raised wave priority around the MFMA phase: +2.3 % for LAUUM (TAG 1: 63.95 -> 65.4 TFLOP/s, same box), nothing
      // for gpp_trtri's masked merges, -3 % for the look-ahead factorisation's concurrent streams (both TAG 0)
      if constexpr (TAG == 1) __builtin_amdgcn_s_setprio(1);
      compute(cur);
      if constexpr (TAG == 1) __builtin_amdgcn_s_setprio(0);
      if (NBUF == 1) __syncthreads();
      store_mc<TM, BK, false, NTH>(smem + nxt * (OPA + OPB), tid, ra, 0);
      store_mc<TN, BK, false, NTH>(smem + nxt * (OPA + OPB) + OPA, tid, rb, 0);
      __syncthreads();
    }
  }
  for (; c < nch; ++c) general_iter(c);
  before_epilogue();

  // epilogue: slab (a,b) holds C[row0+wm+4a+(l>>4)][col0+wn+16b+(l&15)].  The beta path first issues all C loads of a
  // group of slabs (clamped addresses, no branches around loads) and only then combines and stores.
  const double alpha = p.alpha, beta = p.beta;
  constexpr int GA = RB < 4 ? RB : 4;  // slab rows per group
#pragma unroll
  for (int a0 = 0; a0 < RB; a0 += GA) {
    double cold[GA][CB];
    if (beta != 0.0) {
#pragma unroll
      for (int a = 0; a < GA; ++a) {
        const int m = row0 + wm + 4 * (a0 + a) + lk;
#pragma unroll
        for (int b = 0; b < CB; ++b) {
          const int n = col0 + wn + 16 * b + li;
          const bool ok = (m < p.M) && (n < p.N) && (p.c_lower == 0 || (p.c_lower == 1 ? n <= m : n >= m));
          const double* src = ok ? C + (int64_t)m * p.ldc + n : Cd;
          cold[a][b] = *src;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < GA; ++a) {
      const int m = row0 + wm + 4 * (a0 + a) + lk;
#pragma unroll
      for (int b = 0; b < CB; ++b) {
        const int n = col0 + wn + 16 * b + li;
        const bool ok = (m < p.M) && (n < p.N) && (p.c_lower == 0 || (p.c_lower == 1 ? n <= m : n >= m));
        double v = alpha * acc4[(a0 + a) >> 2][b][(a0 + a) & 3];
        if (beta != 0.0) v = fma(beta, cold[a][b], v);
        if (ok) {
          C[(int64_t)m * p.ldc + n] = v;
          if (C2)  // mirrored (transposed) copy
            C2[(int64_t)n * p.ldc2 + m] = v;
        }
      }
    }
  }
}

template <int VAR, int WTM, int WTN, int TAG = 0, int BK = 16, int NBUF = 2, int WR = 2>
__global__ __launch_bounds__(128 * WR, ((VAR == 2 && !(BK > 16 && WTM >= 64)) || WTM < 64) ? 2 : 1) void gpp_gemm_f64(GemmArgs p) {
  constexpr int TM = WR * WTM, TN = 2 * WTN;  // work-group tile: TM rows x TN columns (WR x 2 waves)
  extern __shared__ __attribute__((aligned(16))) double smem[];  // NBUF * 2 * OPSZ doubles (see gemm_lds_bytes)
  int tm, tn;
  if (p.swz) {
    // XCD-aware mapping (blocks are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2): the 64 work-groups
    // an XCD runs concurrently (32 CUs x 2) form ONE 8 x 8 super-tile of output tiles, so every staged A chunk is
    // shared by 8 and every B chunk by 8 work-groups of that L2.  Tiles outside the matrix / triangle exit at once.
    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
    const int S = (i >> 6) * 8 + xcd, w = i & 63;
    const int super_n = (p.tiles_n + 7) >> 3;
    const int sm = S / super_n, sn = S - sm * super_n;
    tm = sm * 8 + (w >> 3);
    tn = sn * 8 + (w & 7);
    if (tm >= p.tiles_m || tn >= p.tiles_n) return;
    if ((p.c_lower == 1 && tn > tm) || (p.c_lower == 2 && tn < tm)) return;
  } else {
    const int t = (p.batch_fast ? (int)(blockIdx.x / (unsigned)p.nbatch) : (int)blockIdx.x) + (int)p.tile_base;
    if (p.c_lower == 1 && p.row_mod >= 1) {
      // owned tile rows tm = row_off + row_mod * i, row i has tm + 1 tiles: S(i) = i (row_off + 1) + row_mod i (i-1) / 2
      const float a = 0.5f * (float)p.row_mod, b = (float)p.row_off + 1.f - a;
      int i = (int)((-b + sqrtf(b * b + 4.f * a * (float)t)) / (2.f * a));
      auto S = [&](int q) { return q * (p.row_off + 1) + p.row_mod * (q * (q - 1) / 2); };
      while (S(i + 1) <= t) ++i;
      while (S(i) > t) --i;
      tm = p.row_off + p.row_mod * i;
      tn = t - S(i);
    } else if (p.c_lower == 1 && TM == 2 * TN) {
      // tall tiles, lower triangle: tile row tm (TM rows) has the column tiles 0 .. 2 tm + 1; S(tm) = tm (tm + 1) tiles precede it
      tm = (int)((sqrtf(4.f * (float)t + 1.f) - 1.f) * 0.5f);
      while ((tm + 1) * (tm + 2) <= t) ++tm;
      while (tm * (tm + 1) > t) --tm;
      tn = t - tm * (tm + 1);
      if (tn >= p.tiles_n) return;
    } else if (p.c_lower) {
      tm = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
      while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
      while (tm * (tm + 1) / 2 > t) --tm;
      tn = t - tm * (tm + 1) / 2;
      // lower triangle, column blocks owned block-cyclically (the sharded back-substitution): another rank's column block
      if (p.c_lower == 1 && p.own_mod > 1 && (tn / p.own_bt + p.own_off) % p.own_mod != 0) return;
      if (p.c_lower == 2) {  // upper triangle: same enumeration, mirrored tile
        const int q = tm;
        tm = tn;
        tn = q;
        if (p.own_mod > 1 && (tm / p.own_bt + p.own_off) % p.own_mod != 0) return;  // another rank's block row
        if ((tn + 1) * TN <= p.skip_lead) return;  // (tm <= tn) inside the leading block another launch has updated
        if (p.row_limit > 0 && tm * TM >= p.row_limit) return;  // below the trapezoid this launch produces
      }
    } else if (p.col_major) {
      // column-major tile order: consecutive work-groups share the column tile (hence the K range when it depends on
      // the column, and the B chunks) and run in lockstep
      tn = t / p.tiles_m;
      tm = t - tn * p.tiles_m;
    } else {
      tm = t / p.tiles_n;
      tn = t - tm * p.tiles_n;
      if (p.row_reverse) tm = p.tiles_m - 1 - tm;  // longest K ranges (khi grows with the row) first: short tail
    }
  }
  const int64_t bi = p.batch_fast ? (int64_t)(blockIdx.x % (unsigned)p.nbatch) : (int64_t)blockIdx.y;  // batch element
  gemm_tile<VAR, WTM, WTN, TAG, BK, NBUF, WR>(p, tm, tn, p.A + bi * p.sA + (int64_t)blockIdx.z * p.zA, p.B + bi * p.sB + (int64_t)blockIdx.z * p.zB,
                                             p.C + bi * p.sC + (int64_t)blockIdx.z * p.zC,
                                             p.C2 ? p.C2 + bi * p.sC2 + (int64_t)blockIdx.z * p.zC2 : nullptr, smem);

}

// LDS bytes of an instantiation (dynamic: the wide-chunk variants exceed the 64 KiB static limit)
constexpr size_t gemm_lds_bytes(int var, int tm, int tn, int bk, int nbuf) {
  const int tx = tm > tn ? tm : tn;
  const int mc = bk * ldt_mc(tx, bk), kc = tx * LDK;
  return (size_t)nbuf * 2 * ((var != 2 && kc > mc) ? kc : mc) * sizeof(double);
}
template <int VAR, int WTM, int WTN, int TAG, int BK, int NBUF, int WR = 2>
hipError_t launch_inst(hipStream_t s, dim3 grid, const GemmArgs& a) {
  constexpr size_t bytes = (WR == 2) ? gemm_lds_bytes(VAR, 2 * WTM, 2 * WTN, BK, NBUF)
                                     : (size_t)NBUF * BK * (ldt_mc(WR * WTM, BK) + ldt_mc(2 * WTN, BK)) * sizeof(double);
  auto* fn = gpp_gemm_f64<VAR, WTM, WTN, TAG, BK, NBUF, WR>;
  if (bytes > 48 * 1024) {
    static std::atomic<bool> attr_set[64];  // per instantiation and device (function attributes are per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(128 * WR), bytes, s, a);
  return hipGetLastError();
}
// small tiles: wide K chunks for the row-contiguous variant, the classic staging otherwise
#ifndef GPP_SMALL_BK
#define GPP_SMALL_BK 64
#endif
template <int VAR>
hipError_t launch_var(hipStream_t s, int tm, int tn, dim3 grid, const GemmArgs& a) {
  constexpr int SBK = (VAR == 2) ? GPP_SMALL_BK : 16;
  constexpr int SNB = (VAR == 2 && GPP_SMALL_BK > 16) ? 1 : 2;
  if constexpr (VAR == 2) {
    if (tm == 256 && tn == 128) return launch_inst<2, 64, 64, 1, 16, 2, 4>(s, grid, a);  // (TAG 1: its own name in profiles)
  }
  if (tm == 128 && tn == 128 && a.tag == 1 && VAR == 2) return launch_inst<2, 64, 64, 1, 16, 2>(s, grid, a);
  if (tm == 128 && tn == 128) return launch_inst<VAR, 64, 64, 0, 16, 2>(s, grid, a);
  if (tm == 64 && tn == 64) return launch_inst<VAR, 32, 32, 0, SBK, SNB>(s, grid, a);
  if (tm == 32 && tn == 32) return launch_inst<VAR, 16, 16, 0, SBK, SNB>(s, grid, a);
  if (tm == 128 && tn == 32) return launch_inst<VAR, 64, 16, 0, SBK, SNB>(s, grid, a);
  return hipErrorInvalidValue;
}


// ---- DAG executor: device side ----------------------------------------------------------------------------------------------
// (see gpp_internal.h.)  The task list and the product descriptors are read through the CONSTANT address space: scalar loads the
// compiler may repeat at will, exactly like kernel arguments — the tile body compiles to the code of the launch-per-product kernel.
#define GPP_AS4 __attribute__((address_space(4)))
typedef const GemmArgs GPP_AS4 CGemmArgs;

__device__ __forceinline__ int exec_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// thread 0 polls until both counters have reached their values; false: the launch is being abandoned (time-out or abort word)
__device__ __forceinline__ bool exec_poll(int* counters, int w0, int v0, int w1, int v1, long long budget, int32_t* info,
                                          int status = GPP_INFO_EXEC_TIMEOUT) {
  long long t0 = 0;
  bool timed = false;
  for (;;) {
    const bool r0 = w0 < 0 || exec_load(counters + w0) >= v0;
    const bool r1 = w1 < 0 || exec_load(counters + w1) >= v1;
    if (r0 & r1) return true;
    if (!timed) {
      t0 = (long long)wall_clock64();
      timed = true;
    }
    const long long waited = (long long)wall_clock64() - t0;
    if (exec_load(counters) != 0) return false;
    if (waited > budget) {
      const long long ms = waited / 100000;
      __hip_atomic_store(counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicCAS(info, 0, status | (int)(ms > 0xFFFFF ? 0xFFFFF : ms));
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

typedef const DagTask GPP_AS4 CDagTask;
// Plans are independent of the operands' addresses (GemmArgs::buf + byte offsets); this one-wave-per-64-groups kernel writes the
// absolute copy the executor reads, stream-ordered in front of it (the tile body then finds its operands as re-loadable scalars of a
// constant-address-space struct, as kernel arguments would be — computing them per task cost spilled VGPRs).
__global__ __launch_bounds__(64) void gpp_dag_bind(const GemmArgs* rel, GemmArgs* abs, int n, DagBases bases) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= n) return;
  GemmArgs a = rel[g];
  auto base = [&](int b) {
    return b == 0 ? bases.p[0] : b == 1 ? bases.p[1] : b == 2 ? bases.p[2] : b == 3 ? bases.p[3] : b == 4 ? bases.p[4] : b == 5 ? bases.p[5]
                                                                                                     : b == 6 ? bases.p[6] : bases.p[7];
  };
  a.A = reinterpret_cast<const double*>(base(a.buf[0]) + reinterpret_cast<uintptr_t>(a.A));
  a.B = reinterpret_cast<const double*>(base(a.buf[1]) + reinterpret_cast<uintptr_t>(a.B));
  a.C = reinterpret_cast<double*>(base(a.buf[2]) + reinterpret_cast<uintptr_t>(a.C));
  a.C2 = a.buf[3] >= 0 ? reinterpret_cast<double*>(base(a.buf[3]) + reinterpret_cast<uintptr_t>(a.C2)) : nullptr;
  abs[g] = a;
}
__global__ __launch_bounds__(256, 2) void gpp_dag_f64(DagLaunch e) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_idx, s_ok;
  const int tid = threadIdx.x;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  CDagTask* tasks = (CDagTask*)e.tasks;
  CGemmArgs* groups = (CGemmArgs*)e.groups;
#pragma clang diagnostic pop
  // The ticket of the NEXT task is taken just before the current tile's epilogue (thread 0; the value is consumed after the task's
  // increments), so the atomic's round trip runs under the epilogue's own memory traffic.  A work-group then holds two tickets for
  // those few microseconds, its current one always the smaller: the earliest unfinished task is still some work-group's CURRENT
  // task, so progress is as before.  (Earlier than the epilogue the next task — possibly a chain task — would sit behind a whole tile.)
  int pre = -2;  // thread 0: -2 nothing taken ahead, -1 stop, >= 0 ticket
  auto take = [&](int done_after) -> int {  // thread 0 only
    // (main workers: no budget, no quit flag, no limit — nothing but the atomic; the abort word is seen by the waits)
    if (e.max_tasks > 0 && done_after >= e.max_tasks) return -1;
    if (e.quit_id >= 0 && (exec_load(e.counters) != 0 || exec_load(e.counters + e.quit_id) >= e.quit_val)) return -1;
    if (e.ticket_limit <= 0) return __hip_atomic_fetch_add(e.counters + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a filler launch in front of a panel never takes a task that needs that panel: the launch could not end (gpp_dag.hip)
    int old = exec_load(e.counters + 1);
    while (old < e.ticket_limit &&
           !__hip_atomic_compare_exchange_strong(e.counters + 1, &old, old + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    }
    return old < e.ticket_limit ? old : -1;
  };
  for (int done = 0;; ++done) {
    if (tid == 0) {
      s_idx = pre != -2 ? pre : take(done);
      pre = -2;
    }
    __syncthreads();
    const int idx = __builtin_amdgcn_readfirstlane(s_idx);
    __syncthreads();  // (thread 0 rewrites s_idx at the top of the next iteration; the copy tasks have no barrier of their own)
    if (idx < 0 || idx >= e.ntasks) break;
    if (e.trace && tid == 0) {
      e.trace[4 * (size_t)idx] = wall_clock64();
      e.trace[4 * (size_t)idx + 3] = (unsigned long long)blockIdx.x | ((unsigned long long)(unsigned)e.tag << 32);
    }
    const int w0 = tasks[idx].wait_id[0], w1 = tasks[idx].wait_id[1], w2 = tasks[idx].wait_id[2];
    if (w0 >= 0 || w1 >= 0 || w2 >= 0) {
      if (tid == 0) {
        // (fast path: all three counters' loads in flight together; the polling loop only when one of them is not there yet)
        const int v0 = tasks[idx].wait_val[0], v1 = tasks[idx].wait_val[1], v2 = tasks[idx].wait_val[2];
        const int c0 = w0 >= 0 ? exec_load(e.counters + w0) : v0, c1 = w1 >= 0 ? exec_load(e.counters + w1) : v1,
                  c2 = w2 >= 0 ? exec_load(e.counters + w2) : v2;
        bool ok = (c0 >= v0) & (c1 >= v1) & (c2 >= v2);
        if (!ok) {
          ok = exec_poll(e.counters, w0, v0, w1, v1, e.budget, e.info, GPP_INFO_EXEC_TIMEOUT);
          if (ok && w2 >= 0) ok = exec_poll(e.counters, w2, v2, -1, 0, e.budget, e.info, GPP_INFO_EXEC_TIMEOUT);
        }
        s_ok = ok ? 1 : 0;
      }
      __syncthreads();
      const int ok = s_ok;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // what the publishers wrote before their increments is visible from here
      __syncthreads();
      if (!ok) break;
    }
    if (e.trace && tid == 0) e.trace[4 * (size_t)idx + 1] = wall_clock64();
    const CGemmArgs& p = groups[tasks[idx].group];
    const int tm = tasks[idx].tm, tn = tasks[idx].tn;
    if (p.op == 0) {
      auto ahead = [&]() {
        if (tid == 0) pre = take(done + 1);
      };
      if (p.etile == 64) gemm_tile<2, 32, 32, 0, 64, 1, 2>(p, tm, tn, p.A, p.B, p.C, p.C2, smem);
      else gemm_tile<2, 64, 64, 0, 16, 2, 2>(p, tm, tn, p.A, p.B, p.C, p.C2, smem, ahead);
    } else {
      // copy the M x 128 strip tn of B into C: 64 16-byte vectors per row, 4 rows per pass, 8 passes in flight
      const int c = tn * 128 + ((tid & 63) << 1), r4 = tid >> 6;
      if (c < p.N) {
        const bool pair = c + 1 < p.N;
        for (int r0 = 0; r0 < p.M; r0 += 32) {
          v2d v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = r0 + 4 * q + r4;
            const double* src = p.B + (int64_t)(r < p.M ? r : 0) * p.ldb + c;
            if (pair) v[q] = *reinterpret_cast<const v2d*>(src);
            else v[q] = (v2d){*src, 0.0};
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = r0 + 4 * q + r4;
            if (r < p.M) {
              double* dst = p.C + (int64_t)r * p.ldc + c;
              if (pair) *reinterpret_cast<v2d*>(dst) = v[q];
              else *dst = v[q].x;
            }
          }
        }
      }
    }
    const int i0 = tasks[idx].inc_id[0], i1 = tasks[idx].inc_id[1];
    if (i0 >= 0 || i1 >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left the CU
      __syncthreads();
      if (tid == 0) {
        // (the explicit wait between the write-back and the increments is REQUIRED: see panel_publish in gpp_leaf.hip)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (i0 >= 0) __hip_atomic_fetch_add(e.counters + i0, (int)tasks[idx].inc_val[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i1 >= 0) __hip_atomic_fetch_add(e.counters + i1, (int)tasks[idx].inc_val[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (e.trace) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) e.trace[4 * (size_t)idx + 2] = wall_clock64();
    }
  }
}

__global__ __launch_bounds__(64) void gpp_exec_gate(int* counters, int id, int target, int32_t* info, long long budget) {
  if (threadIdx.x == 0) (void)exec_poll(counters, id, target, -1, 0, budget, info);
}
__global__ __launch_bounds__(64) void gpp_exec_signal(int* counters, int id) {
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counters + id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

// tile_m/tile_n: 0 = pick a square tile by grid size; else the work-group tile (128x128, 64x64, 32x32 or 128x32).
hipError_t gpp_launch_gemm(hipStream_t s, int variant, const GemmArgs& a_in, int batch, int tile_m, int tile_n) {
  GemmArgs a = a_in;
  if (a.M <= 0 || a.N <= 0 || batch <= 0) return hipSuccess;
  auto ntiles = [&](int T) -> int64_t {
    const int64_t tm = (a.M + T - 1) / T, tn = (a.N + T - 1) / T;
    return (a.c_lower ? tm * (tm + 1) / 2 : tm * tn) * batch * (a.batch2 > 1 ? a.batch2 : 1);  // triangular: M == N
  };
  if (tile_m == 0) {
    // enough 128^2 tiles to give every CU a work-group -> big tile; otherwise shrink until the chip is covered
    static const int64_t t128 = getenv("GPP_TILE_T128") ? atol(getenv("GPP_TILE_T128")) : 256;  // experiment knobs
    static const int64_t t64 = getenv("GPP_TILE_T64") ? atol(getenv("GPP_TILE_T64")) : 192;
    // (a launch on the panel stream has 32 CUs to fill, not 256: the thresholds scale with the stream's CU count)
    const int64_t cus = a.cu_hint > 0 ? a.cu_hint : 256;
    if (ntiles(128) * 256 >= t128 * cus) tile_m = 128;
    else if (ntiles(64) * 256 >= t64 * cus) tile_m = 64;
    else tile_m = 32;
    tile_n = tile_m;
  }
  const bool tall = (tile_m == 256 && tile_n == 128);
  if (a.c_lower && tile_m != tile_n && !(tall && a.c_lower == 1 && a.row_mod <= 1 && a.row_i1 == 0 && a.row_t1 == 0)) return hipErrorInvalidValue;
  a.tiles_m = (a.M + tile_m - 1) / tile_m;
  a.tiles_n = (a.N + tile_n - 1) / tile_n;
  int64_t nt = a.c_lower ? (int64_t)a.tiles_m * (a.tiles_m + 1) / 2 : (int64_t)a.tiles_m * a.tiles_n;
  if (tall && a.c_lower == 1) nt = (int64_t)a.tiles_m * (a.tiles_m + 1);  // row tm: column tiles 0 .. 2 tm + 1 (those past N exit)
  a.tile_base = 0;
  if (a.c_lower == 1 && (a.row_mod > 1 || a.row_i1 > 0)) {
    if (a.row_mod < 1) { a.row_mod = 1; a.row_off = 0; }
    if (a.row_off < 0 || a.row_off >= a.row_mod) return hipErrorInvalidValue;
    const int64_t rows = a.tiles_m > a.row_off ? (a.tiles_m - a.row_off + a.row_mod - 1) / a.row_mod : 0;
    // tiles before owned row i: S(i) = i (row_off + 1) + row_mod i (i - 1) / 2
    auto S = [&](int64_t i) { return i * (a.row_off + 1) + (int64_t)a.row_mod * (i * (i - 1) / 2); };
    const int64_t i0 = std::min<int64_t>(std::max(a.row_i0, 0), rows), i1 = a.row_i1 > 0 ? std::min<int64_t>(a.row_i1, rows) : rows;
    if (i1 <= i0) return hipSuccess;
    a.tile_base = S(i0);
    nt = S(i1) - S(i0);
  } else {
    a.row_mod = 0;  // plain enumeration
    if (a.c_lower == 1 && a.row_t1 > 0) {
      // a band of tile rows [row_t0, row_t1): the row-by-row enumeration of the lower triangle makes it a contiguous range
      const int64_t t0 = std::min<int64_t>(std::max(a.row_t0, 0), a.tiles_m), t1 = std::min<int64_t>(a.row_t1, a.tiles_m);
      if (t1 <= t0) return hipSuccess;
      a.tile_base = t0 * (t0 + 1) / 2;
      nt = t1 * (t1 + 1) / 2 - a.tile_base;
    }
  }
  a.swz = 0;
  // Measured on MI355X (N = 20000): the super-tile mapping LOSES 10-20 % against plain row-major order (row-major
  // already shares each A chunk among 16 and each B chunk among 4 work-groups of an XCD, and the triangular launches
  // waste whole super-tiles), so it stays off; kept for experiments with GPP_SWZ-style builds.
  if (false && nt >= 1024) {
    const int64_t n_super = (int64_t)((a.tiles_m + 7) / 8) * ((a.tiles_n + 7) / 8);
    nt = ((n_super + 7) / 8) * 8 * 64;
    a.swz = 1;
  }
  const int batch2 = a.batch2 > 1 ? a.batch2 : 1;
  dim3 grid((unsigned)nt, (unsigned)batch, (unsigned)batch2);
  if (a.batch_fast && batch > 1 && !a.swz && nt * batch < (int64_t)1 << 31) {
    a.nbatch = batch;
    grid = dim3((unsigned)(nt * batch), 1u, (unsigned)batch2);
  } else {
    a.batch_fast = 0;
  }
  switch (variant) {
    case 0: return launch_var<0>(s, tile_m, tile_n, grid, a);
    case 1: return launch_var<1>(s, tile_m, tile_n, grid, a);
    case 2: return launch_var<2>(s, tile_m, tile_n, grid, a);
    default: return hipErrorInvalidValue;
  }
}

hipError_t gpp_launch_dag_bind(hipStream_t s, const GemmArgs* rel, GemmArgs* abs, int n, const DagBases& bases) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gpp_dag_bind, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, rel, abs, n, bases);
  return hipGetLastError();
}
hipError_t gpp_launch_dag(hipStream_t s, int nworkers, const DagLaunch& e) {
  if (nworkers <= 0) return hipSuccess;
  constexpr size_t b128 = gemm_lds_bytes(2, 128, 128, 16, 2), b64 = gemm_lds_bytes(2, 64, 64, 64, 1);
  constexpr size_t bytes = b128 > b64 ? b128 : b64;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) return err;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(gpp_dag_f64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err != hipSuccess) return err;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(gpp_dag_f64, dim3((unsigned)nworkers), dim3(256), bytes, s, e);
  return hipGetLastError();
}
hipError_t gpp_launch_exec_gate(hipStream_t s, int* counters, int id, int target, int32_t* info, long long budget) {
  hipLaunchKernelGGL(gpp_exec_gate, dim3(1), dim3(64), 0, s, counters, id, target, info, budget);
  return hipGetLastError();
}
hipError_t gpp_launch_exec_signal(hipStream_t s, int* counters, int id) {
  hipLaunchKernelGGL(gpp_exec_signal, dim3(1), dim3(64), 0, s, counters, id);
  return hipGetLastError();
}
