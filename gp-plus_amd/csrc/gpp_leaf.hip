// gpp_leaf.hip — diagonal leaf of the recursive Cholesky: factor an n x n (n <= 128) SPD block and invert its
// factor inside ONE work-group, with the block held in MFMA accumulator registers and LDS.  Replaces the unblocked
// LAPACK potf2/trti2 steps inside torch.linalg.cholesky_ex (reference call site: gpytorch psd_safe_cholesky reached
// from optim/mll_torch.py:116).  It runs N/128 times per factorisation on a latency-bound chain, so it is latency-tuned
// (tools/attic/leaf_probe.hip prints its per-phase cycle counts).
//
//   in : A[n x n], UPPER triangle read (A = U^T U with U stored row-major = L stored column-major; the strict lower
//        triangle is never touched).  Internally the kernel works on L = U^T: every access to A swaps its indices.
//   out: A    <- U (upper triangle),  Linv block <- inv(L) in the lower triangle AND inv(L)^T mirrored in the strict
//        upper triangle (the mirror lets the inverse enter later products as a row-contiguous "TN" operand)
//        *info <- row_offset + k + 1 for the first non-positive / NaN pivot (kept if already non-zero)
//
// Layout: the 128 x 128 block is an 8 x 8 grid of 16 x 16 tiles; the 36 lower tiles are dealt to the 4 waves
// (9 accumulator tiles = 72 VGPRs each, C/D layout row = (l>>4) + 4r, col = l&15).  Right-looking over tile columns s:
//   (1) owners park column s in LDS;
//   (2) wave 0 factors the 16 x 16 diagonal tile with lane-per-row registers and v_readlane broadcasts and leaves it in
//       LDS with its diagonal replaced by the reciprocals 1/L_cc;
//   (3) panel rows are solved by forward substitution, one lane per row, against broadcast LDS reads of that tile
//       (no 16 x 16 inverse on the critical path);
//   (4) trailing tiles acc(i,j) -= L(i,s) L(j,s)^T on the MFMA (one v_mfma_f64_16x16x4_f64 per 16x16x4 step).
// Then the 8 diagonal tiles are inverted in parallel (2 per wave, lane per column) and the inverse is assembled by
// pair merging at tile level (sizes 16, 32, 64): X21 = -X22 (L21 X11), MFMA again.
#include "../../include/gpp.h"
#include "gpp_internal.h"
#include <algorithm>
#include <atomic>
#include <type_traits>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// Phase stamps for tools/attic/leaf_probe.hip only (the product build defines nothing and the macro vanishes).
#ifdef GPP_LEAF_STAMP
__device__ unsigned long long g_leaf_stamps[64];
#define STAMP(i)                                                                   \
  do {                                                                             \
    if (threadIdx.x == 0) g_leaf_stamps[(i)] = __builtin_amdgcn_s_memtime();       \
  } while (0)
#else
#define STAMP(i) \
  do {           \
  } while (0)
#endif

namespace {

constexpr int NB = GPP_TILE;
constexpr int TSZ = 16 * 16;  // doubles per LDS tile (unpadded: the whole image must fit ONE GEMM LDS slot)
constexpr int NT = 36;        // lower tiles of an 8x8 grid
constexpr int SLOTS = 9;      // tiles per wave

__device__ __forceinline__ int toff(int i, int j) { return (i * (i + 1) / 2 + j) * TSZ; }

// Element (row, col) of a 16 x 16 LDS tile.  Rows are 128 bytes; the XOR swizzle (col ^ row>>1) spreads a column
// read (16 rows, same col) over 16 distinct 8-byte bank slots without padding the tile.
__device__ __forceinline__ int tix(int row, int col) { return row * 16 + (col ^ (row >> 1)); }

// Broadcast lane J of every 16-lane row (gfx90a+ DPP row_newbcast, the only DPP mode of the fp64 ALU).
template <int J>
__device__ __forceinline__ double bcast16(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, false);
}
// acc -= bcast16<J>(l) * m in ONE instruction (v_fmac_f64 with a DPP source).  The s_nop covers the two wait states
// a DPP read needs after a VALU write of its source, which the compiler does not track through inline asm.
template <int J>
__device__ __forceinline__ void fnma_bcast16(double& acc, double l, double m) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(acc)
      : "v"(l), "v"(m), "n"(J));
}

// Per-lane element offsets inside a tile for the three MFMA fragment shapes, swizzle folded in so that every LDS read
// is "tile base + lane offset + immediate".  With lr = lane>>4, lc = lane&15, i = lane&3:
//  (a[]: offsets of the round-1 4x4x4 A fragments, rows 4rb+i, k = 4kk+lr: tix = a[rb&1] + 64 rb + (4kk ^ 4(rb>>1)); kept for
//   the probes, the tile products read A through the "tile^T" pattern below now)
//  B operand "tile^T" (n = lc, k = 4kk+lr reads tile[n][k]):      tix = rk[kk&1] + 8 (kk>>1)
//  B operand "tile"   (k = 4kk+lr, n = lc reads tile[k][n]) and the C/D layout (row lr+4r, col lc): tix = kn[kk]
struct LaneOff {
  int a[2], rk[2], kn[4];
};
__device__ __forceinline__ LaneOff lane_offsets(int lane) {
  const int lr = lane >> 4, lc = lane & 15, i = lane & 3;
  LaneOff o;
  o.a[0] = 16 * i + (lr ^ (i >> 1));
  o.a[1] = 16 * i + (lr ^ (2 + (i >> 1)));
  const int m = (lc >> 1) & 4, base = 16 * lc + (lr ^ ((lc >> 1) & 3));
  o.rk[0] = base + m;
  o.rk[1] = base + (4 ^ m);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) o.kn[kk] = 64 * kk + 16 * lr + ((lc ^ (lr >> 1)) ^ (2 * kk));
  return o;
}

// Operand fragments of one 16 x 16 x 16 tile product acc += At * B on the MFMA.  All 20 LDS reads of a product are
// requested together, and the callers request the NEXT product's fragments before issuing this one's 16 MFMAs, so the
// LDS latency hides behind the matrix pipe instead of preceding every instruction.
struct Frag {
  double a[4];     // [kk]: At[row = l&15][k = 4kk + (l>>4)]  (v_mfma_f64_16x16x4_f64 A operand; the "tile^T" read pattern)
  double b[4];     // [kk]: B[k = 4kk + (l>>4)][n = l&15]
};
// KN: B is stored [k][n] (merge products);  !KN: B is stored [n][k], i.e. the product is At * Bt^T (trailing update)
template <bool NEG, bool KN>
__device__ __forceinline__ void load_frag(Frag& f, const double* At, const double* Bt, const LaneOff& o) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double v = KN ? Bt[o.kn[kk]] : Bt[o.rk[kk & 1] + 8 * (kk >> 1)];
    f.b[kk] = NEG ? -v : v;
  }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f.a[kk] = At[o.rk[kk & 1] + 8 * (kk >> 1)];
}
// one v_mfma_f64_16x16x4_f64 per k-step of 4 (round 1 issued four 4x4x4 forms with broadcast A fragments: 16 + 4 LDS reads and
// 16 MFMAs per tile product where 4 + 4 and 4 do; the accumulator layout — row (l>>4) + 4v, col l&15 — is the same)
__device__ __forceinline__ void mma_frag(v4d& acc, const Frag& f) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[kk], f.b[kk], acc, 0, 0, 0);
}
// t += sgn * sum_{k=k0}^{k1} tile(i,k) * tile(k,j)   (k1 >= k0), double-buffered over k
template <bool NEG>
__device__ __forceinline__ void tile_chain(v4d& t, const double* img, int i, int j, int k0, int k1, const LaneOff& o) {
  Frag f0, f1;
  int k = k0;
  load_frag<NEG, true>(f0, img + toff(i, k), img + toff(k, j), o);
  // every "request next, multiply current" pair sits in ONE basic block: the compiler's s_waitcnt bookkeeping is only
  // exact inside a block, and a conservative wait at a join would serialise the LDS reads and the MFMAs again
  while (true) {
    if (k == k1) {
      mma_frag(t, f0);
      break;
    }
    load_frag<NEG, true>(f1, img + toff(i, k + 1), img + toff(k + 1, j), o);
    mma_frag(t, f0);
    ++k;
    if (k == k1) {
      mma_frag(t, f1);
      break;
    }
    load_frag<NEG, true>(f0, img + toff(i, k + 1), img + toff(k + 1, j), o);
    mma_frag(t, f1);
    ++k;
  }
}

__device__ __forceinline__ void store_acc(double* tile, const v4d& a, const LaneOff& o) {
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[o.kn[r]] = a[r];
}

// Right-looking Cholesky of the 16 x 16 tile held one row per lane (a[c] = row (lane&15), column c), pivot K.
template <int K, int J>
__device__ __forceinline__ void chol_update(double (&a)[16], double l) {
  if constexpr (J < 16) {
    fnma_bcast16<J>(a[J], l, l);  // a[J] -= L[J][K] * L[i][K]
    chol_update<K, J + 1>(a, l);
  }
}
template <int K>
__device__ __forceinline__ void chol_pivots(double (&a)[16], double& rd, int i, int& bad) {
  if constexpr (K < 16) {
    const double akk = bcast16<K>(a[K]);
    // Off the dependency chain: a non-positive or NaN pivot only raises the flag; the NaNs it breeds are never used
    // because the caller retries with jitter when *info != 0 (gp-plus_amd/linalg.py::_factor).
    if (!(akk > 0.0) && !bad) bad = K + 1;
    // 1/sqrt(pivot) by hardware rsq + two Newton steps, L_KK = pivot * r: a far shorter dependency chain than sqrt
    // followed by a division, and this chain runs 128 times per leaf
    double r = __builtin_amdgcn_rsq(akk);
    r = fma(r * 0.5, fma(-akk * r, r, 1.0), r);
    r = fma(r * 0.5, fma(-akk * r, r, 1.0), r);
    const double l = (i == K) ? akk * r : a[K] * r;
    a[K] = l;
    rd = (i == K) ? r : rd;
    chol_update<K, K + 1>(a, l);
    chol_pivots<K + 1>(a, rd, i, bad);
  }
}

// One level of the pair-merge inversion: diagonal blocks of H tiles are already inverted; for every pair (based at
// tile b = 2H p) the off-diagonal block becomes X21 = -X22 (L21 X11).  4H output tiles, dealt to the 4 waves so that
// each wave gets the same number of tile products in both phases.
template <int H>
__device__ __forceinline__ void merge_level(double* img, int wave, const LaneOff& o) {
  constexpr int NTILE = 4 * H, PER = (NTILE + 3) / 4;
  int ti[PER], tj[PER], tb[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int e = wave + 4 * q, p = e / (H * H), rem = e % (H * H), r = rem / H;
    tb[q] = 2 * H * p;
    ti[q] = tb[q] + H + r;
    tj[q] = tb[q] + (rem % H + r) % H;  // rotate the columns so a wave's tiles have different product counts
  }
  v4d t[PER];
  // phase a: T(i,j) = sum_{k=j}^{b+H-1} L(i,k) X(k,j)
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    t[q] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (wave + 4 * q < NTILE) tile_chain<false>(t[q], img, ti[q], tj[q], tj[q], tb[q] + H - 1, o);
  }
  __syncthreads();  // every L(i,k) of this level has been read: the slots may now take T
#pragma unroll
  for (int q = 0; q < PER; ++q)
    if (wave + 4 * q < NTILE) store_acc(img + toff(ti[q], tj[q]), t[q], o);
  __syncthreads();
  // phase b: X(i,j) = -sum_{k=b+H}^{i} X(i,k) T(k,j)
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    t[q] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (wave + 4 * q < NTILE) tile_chain<true>(t[q], img, ti[q], tj[q], tb[q] + H, ti[q], o);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; ++q)
    if (wave + 4 * q < NTILE) store_acc(img + toff(ti[q], tj[q]), t[q], o);
  __syncthreads();
}

// Rows 16s .. 16s+15 of U (= columns of L, strictly below the diagonal; the diagonal itself is stored by the factoring
// wave) from the LDS image to memory: U[16s+c][j] = L[j][16s+c] for j > 16s+c.  Work item e = (row c, 64-column half),
// dealt round-robin to `nw` waves; a lane owns one column j, so every store is a contiguous run of a U row.
__device__ __forceinline__ void write_u_rows(double* __restrict__ A, int64_t lda, int n, int s, int w, int nw, int lane) {
  extern __shared__ __attribute__((aligned(16))) double img[];
  const int halves = (s < 4) ? 2 : 1;  // columns 16s .. 127: more than 64 of them only while s < 4
  for (int e = w; e < 16 * halves; e += nw) {
    const int c = e & 15, j = 16 * s + 64 * (e >> 4) + lane;
    if (j < n && j > 16 * s + c) A[(int64_t)(16 * s + c) * lda + j] = img[toff(j >> 4, s) + tix(j & 15, c)];
  }
}

// 64 x 64 block (rows r0.., columns c0..) of the result from the LDS image to memory: inv(L) where col <= row, its
// transpose above the diagonal.  A lane owns two consecutive columns (one 16-byte store), half a wave one row, and
// each wave keeps four row pairs in flight.
__device__ __forceinline__ void write_linv_block(double* __restrict__ Linv, int64_t ldi, int n, int r0, int c0, int wave,
                                                 int lane) {
  extern __shared__ __attribute__((aligned(16))) double img[];
  const int col = c0 + 2 * (lane & 31), J = col >> 4, cc = col & 15;  // column tile, (even) column inside it
  const int triJ = J * (J + 1) / 2;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    double v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = r0 + 32 * pass + 8 * u + 2 * wave + (lane >> 5);
      const int I = row >> 4, rr = row & 15;
      const int lo_base = (I * (I + 1) / 2 + J) * TSZ + rr * 16, hi_base = (triJ + I) * TSZ + (rr ^ (cc >> 1));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = cc + e;
        const bool lower = (J < I) || (J == I && c <= rr);
        v[u][e] = img[lower ? lo_base + (c ^ (rr >> 1)) : hi_base + c * 16];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = r0 + 32 * pass + 8 * u + 2 * wave + (lane >> 5);
      double* dst = Linv + (int64_t)row * ldi + col;
      if (row < n && col + 1 < n) *reinterpret_cast<v2d*>(dst) = (v2d){v[u][0], v[u][1]};
      else if (row < n && col < n) dst[0] = v[u][0];
    }
  }
}

// The whole leaf, called by all 256 threads of a work-group (the leaf kernel below, and the chain work-group of gpp_panel_potrf_inv).
__device__ __forceinline__ void leaf_body(double* __restrict__ A, int64_t lda, double* __restrict__ Linv, int64_t ldi, int n,
                                          int32_t* info, int row_offset) {
  // ONE 36-tile image of exactly 72 KiB: LDS is allocated contiguously, so on the look-ahead stream the leaf can only
  // start beside a running GEMM work-group if it fits the 72.5 KiB slot a finished GEMM work-group leaves behind.
  // Slot (i,j) holds, in turn: the parked raw tile; L(i,j) (off-diagonal) or L_jj with its diagonal replaced by the
  // reciprocals 1/L_cc (diagonal); and finally inv(L)(i,j): L(i,j) is consumed exactly at the merge level that
  // overwrites it.
  extern __shared__ __attribute__((aligned(16))) double img[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const LaneOff off = lane_offsets(lane);

  // tile ownership: lower tiles enumerated column-major (j outer), dealt round-robin to the 4 waves
  int ti[SLOTS], tj[SLOTS];
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    int t = 4 * q + wave, j = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c)
      if (t >= 8 - j) { t -= 8 - j; ++j; }
    tj[q] = j;
    ti[q] = j + t;
  }

  v4d acc[SLOTS];
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ti[q] + lr + 4 * r, col = 16 * tj[q] + lc;
      double v = (row == col) ? 1.0 : 0.0;
      if (row < n && col <= row) v = A[(int64_t)col * lda + row];  // L[row][col] = U[col][row]
      acc[q][r] = v;
    }
  }

  STAMP(0);
  for (int s = 0; s < 8; ++s) {
    STAMP(1 + 5 * s);
    // (1) park column s
#pragma unroll
    for (int q = 0; q < SLOTS; ++q)
      if (tj[q] == s) store_acc(img + toff(ti[q], s), acc[q], off);
    __syncthreads();

    STAMP(2 + 5 * s);
    double* D = img + toff(s, s);
    // (2) wave 0: factor the diagonal tile.  lane (l & 15) owns row i; the four 16-lane rows of the wave carry
    // identical copies, so the row_newbcast broadcasts need no cross-row traffic.
    if (wave == 0) {
      const int i = lc;
      double a[16], rd = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = D[tix(i, c)];
      int bad = 0;
      chol_pivots<0>(a, rd, i, bad);
      if (bad && lane == 0) atomicCAS(info, 0, row_offset + 16 * s + bad);
      if (lane < 16) {
        // LDS copy: strict lower L, 1/L_ii on the diagonal (what the substitutions multiply by); the upper part is
        // never read.  Only the diagonal of U goes to memory from here (one store): the rest of the tile is written
        // from LDS by the waves that idle during the next factorisation.
#pragma unroll
        for (int c = 0; c < 16; ++c) D[tix(i, c)] = (c == i) ? rd : a[c];
        double lii = a[0];
#pragma unroll
        for (int c = 1; c < 16; ++c) lii = (c == i) ? a[c] : lii;
        const int g = 16 * s + i;
        if (g < n) A[(int64_t)g * lda + g] = lii;
      }
    } else if (s > 0) {
      write_u_rows(A, lda, n, s - 1, wave - 1, 3, lane);
    }
    STAMP(3 + 5 * s);
    __syncthreads();

    STAMP(4 + 5 * s);
    // (3) panel: row R of the block (one lane per row) solves x L_ss^T = b by forward substitution, right-looking so
    // that the updates of one column are independent instructions
    if (tid < NB && tid >= 16 * (s + 1)) {
      const int R = tid;
      double* P = img + toff(R >> 4, s);
      double x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = P[tix(R & 15, c)];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        x[c] *= D[tix(c, c)];
#pragma unroll
        for (int q = c + 1; q < 16; ++q) x[q] = fma(-x[c], D[tix(q, c)], x[q]);
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) P[tix(R & 15, c)] = x[c];
    }
    __syncthreads();

    STAMP(5 + 5 * s);
    // (4) trailing update acc(i,j) -= L(i,s) L(j,s)^T; a wave's tiles of columns > s are its slots q0 .. 8 (column-major
    // deal), fragments of slot q+1 are requested before the MFMAs of slot q
    // (slots are sorted by column, so "slot q is active" implies "slot q+1 is active")
    {
      Frag f[2];
      if (tj[0] > s) load_frag<true, false>(f[0], img + toff(ti[0], s), img + toff(tj[0], s), off);
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) {
        constexpr int LAST = SLOTS - 1;
        const int qn = q < LAST ? q + 1 : LAST;
        if (tj[q] > s) {  // request slot q+1, multiply slot q: one basic block (see tile_chain)
          if (q < LAST) load_frag<true, false>(f[qn & 1], img + toff(ti[qn], s), img + toff(tj[qn], s), off);
          mma_frag(acc[q], f[q & 1]);
        } else if (q < LAST && tj[qn] > s) {
          load_frag<true, false>(f[qn & 1], img + toff(ti[qn], s), img + toff(tj[qn], s), off);
        }
      }
    }
  }
  write_u_rows(A, lda, n, 7, wave, 4, lane);
  __syncthreads();

  STAMP(41);
  // ---- inverse of the 8 diagonal tiles, two per wave (lane groups 0 and 1), lane per column ---------------------
  if (lr < 2) {
    double* Dt = img + toff(wave + 4 * lr, wave + 4 * lr);
    const int j = lc;
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (r == j) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      x[r] *= Dt[tix(r, r)];
#pragma unroll
      for (int k = r + 1; k < 16; ++k) x[k] = fma(-Dt[tix(k, r)], x[r], x[k]);
    }
    // every lane of the wave has finished reading its tile before any of them overwrites it (one wave, in order)
#pragma unroll
    for (int r = 0; r < 16; ++r) Dt[tix(r, j)] = x[r];
  }
  __syncthreads();
  STAMP(44);

  // ---- pair merging at tile level: 16 -> 32 -> 64 -> 128 ------------------------------------------------------------
  merge_level<1>(img, wave, off);
  merge_level<2>(img, wave, off);
  // the two 64 x 64 diagonal blocks of the result are final: their stores drain behind the last level's MFMAs (a
  // single CU moves the 128 KiB result at only ~16 B/clk, so unhidden it is 8% of the kernel)
  write_linv_block(Linv, ldi, n, 0, 0, wave, lane);
  write_linv_block(Linv, ldi, n, 64, 64, wave, lane);
  merge_level<4>(img, wave, off);

  STAMP(42);
  write_linv_block(Linv, ldi, n, 64, 0, wave, lane);
  write_linv_block(Linv, ldi, n, 0, 64, wave, lane);
  STAMP(43);
}

__global__ __launch_bounds__(256) void gpp_leaf_potrf_inv(double* __restrict__ A, int64_t lda, double* __restrict__ Linv,
                                                          int64_t ldi, int n, int32_t* info, int row_offset, int64_t sA,
                                                          int64_t sLi) {
  // batch element: an independent block
  leaf_body(A + (int64_t)blockIdx.x * sA, lda, Linv + (int64_t)blockIdx.x * sLi, ldi, n, info + blockIdx.x, row_offset);
}

// ---- cooperative panel: a whole diagonal block (C leaves of 128 rows) factored AND inverted by ONE launch -----------------------
// The leaf-step factorisation of a diagonal block (potrf_blk in gpp_api.hip) is a chain of 3 launches per 128 rows — leaf, panel
// solve, rank-128 update — followed by the pair merges of the block's inverse: at 1024 rows 8 x (45 + ~32) us + ~200 us, of which
// only the leaves are inherently serial.  Here ONE work-group runs the leaves back to back and the others do everything else in
// 128 x 32 strips, handing results over through flags in device memory (release / acquire at agent scope: tools/attic/flag_probe.hip
// measures 0.8 us per hand-off on an idle chip, ~4 us beside a kernel that saturates the memory system — a launch gap is 3-12):
//   chain work-group : for j = 0 .. C-1:  wait until tile (j,j) has its update from row j-1;  leaf(j) -> U_jj, inv(L_jj);  publish
//   factor strip (c,q), c = 1 .. C-1, q = 0 .. 3 (columns 32 q .. 32 q + 31 of tile column c), for j = 0 .. c-1:
//         S : U[j,c]_q = inv(L_jj) A[j,c]_q            (needs leaf j)
//         U : A[r,c]_q -= U[j,r]^T U[j,c]_q, r = j+1..c (needs all four strips of U[j,r]); after (r = c, j = c-1) tile (c,c) is ready
//   inverse strip (i,q), i = 0 .. C-2, for j = i+1 .. C-1 (bordering, one leaf row at a time, one step BEHIND the factor strips so
//         that it never delays them):  Linv[j,i]_q = -inv(L_jj) sum_{m=i}^{j-1} U[m,j]^T Linv[m,i]_q   (+ its mirror)
// Every strip belongs to one work-group for the whole launch, so each entry is produced by the same sequence of operations
// whatever the timing (bitwise repeatable), a strip's read-modify-write needs no lock, and the only waits are on EARLIER tasks
// of a fixed global order (step j: S, then U, then the inverse row j-1) — no cycles, provided every work-group of the launch
// gets onto the chip at some point: the grid never exceeds what the stream's CUs hold at one work-group each (72 KiB of LDS, ~350
// registers per lane).
// A wait gives up after `budget` ticks of the 100 MHz constant clock (default 0.5 s; then *info = 2^30 + the milliseconds waited and
// every work-group leaves): a logic error or a co-tenant must not hang the GPU.
// The flag block is all zero between launches: the LAST work-group to leave clears it (a counter of finished work-groups), so
// no memset precedes the launch — which also keeps memset nodes out of captured graphs (on this stack a replayed
// hipMemsetAsync node of a few KiB came to write an address-like 8-byte pattern instead of zeros once the process had
// synchronised the device: tools/_dbg notes in profiles/r03_potrf_experiments.txt).
constexpr int PMAXC = 32;                         // leaves per launch the flag block is laid out for
constexpr int PF_ABORT = 0, PF_DONE = 1, PF_LEAF = 2, PF_DIAG = 2 + PMAXC, PF_SOLVED = 2 + 2 * PMAXC;  // offsets (ints) in a flag block
constexpr int PF_INTS = PF_SOLVED + PMAXC * PMAXC;
constexpr int PBK = 16;                           // k rows per staged chunk
constexpr int PLDA = 128 + 16, PLDB = 32 + 16;    // LDS row strides (doubles): two consecutive k rows fall in different bank halves
constexpr int P_BUF = PBK * (PLDA + PLDB);        // one staged chunk of both operands; two of them: 48 KiB, less than the leaf's image

// Phase stamps of the chain and of the strip that feeds the next leaf (tools/attic/panel_stamps.py; a probe build only).
#ifdef GPP_PANEL_STAMP
__device__ unsigned long long g_panel_stamps[512];
#define PSTAMP(i)                                                                   \
  do {                                                                              \
    if (threadIdx.x == 0) g_panel_stamps[(i)] = __builtin_amdgcn_s_memtime();       \
  } while (0)
#else
#define PSTAMP(i) \
  do {            \
  } while (0)
#endif

// One 128 x 32 strip product  acc += A^T B  over K = 128 on a work-group of four waves.  Both operands are fetched WHOLE into
// registers first (A: 128 k x 128 m, 32 16-byte vectors per thread; B: 128 k x 32 n, 8 vectors) — one memory round trip per
// product instead of one per chunk, and the fetch can be issued before the flag the product waits for when the operand is the
// work-group's own data — and then fed through two LDS chunk buffers of 16 k, one barrier per chunk.
// Wave w owns the 16-row blocks w and 7 - w: with the triangular A of a leaf inverse (AMASK: keep k <= m) chunk c only matters for
// row blocks >= c, and this deal gives every wave the same 72 of 128 MFMAs.
struct StripAcc {
  v4d c[2][2];  // [a][b] element v: row 16 rb(a) + 4 v + (lane >> 4), rb(0) = wave, rb(1) = 7 - wave; column 16 b + (lane & 15)
};
struct StripA { v2d r[32]; };
struct StripB { v2d r[8]; };
// (addresses = work-group-uniform row base + ONE per-thread 32-bit byte offset: no address registers per load)
// klim / mlim / nlim: valid rows and columns of the operand (the last tile of a ragged block is smaller); the rest reads as zero.
// A 16-byte vector that starts on the last valid column reads one double past it: still inside the row (ld is even, gpp.h).
__device__ __forceinline__ void strip_load_a(StripA& o, const double* __restrict__ Ag, int64_t lda, int tid, int klim = 128,
                                             int mlim = 128) {
  const unsigned off = (unsigned)(((int64_t)(tid >> 6) * lda + ((tid & 63) << 1)) * 8);
  if (klim == 128 && mlim == 128) {
#pragma unroll
    for (int i = 0; i < 32; ++i)  // vector i: row (tid >> 6) + 4 i, columns 2 (tid & 63) ..
      o.r[i] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Ag + (int64_t)(4 * i) * lda) + off);
  } else {
    const int c2 = (tid & 63) << 1;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const bool ok = ((tid >> 6) + 4 * i < klim) & (c2 < mlim);
      // (uniform row base, per-thread offset: a dropped vector reads the first element of a valid row instead)
      const v2d t = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Ag + (int64_t)(4 * i < klim ? 4 * i : 0) * lda) + (ok ? off : 0u));
      o.r[i].x = ok ? t.x : 0.0;
      o.r[i].y = (ok & (c2 + 1 < mlim)) ? t.y : 0.0;
    }
  }
}
__device__ __forceinline__ void strip_load_b(StripB& o, const double* __restrict__ Bg, int64_t ldb, int tid, int klim = 128,
                                             int nlim = 32) {
  const unsigned off = (unsigned)(((int64_t)(tid >> 4) * ldb + ((tid & 15) << 1)) * 8);
  if (klim == 128 && nlim >= 32) {
#pragma unroll
    for (int i = 0; i < 8; ++i)  // vector i: row (tid >> 4) + 16 i, columns 2 (tid & 15) ..
      o.r[i] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Bg + (int64_t)(16 * i) * ldb) + off);
  } else {
    const int c2 = (tid & 15) << 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = ((tid >> 4) + 16 * i < klim) & (c2 < nlim);
      const v2d t = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Bg + (int64_t)(16 * i < klim ? 16 * i : 0) * ldb) + (ok ? off : 0u));
      o.r[i].x = ok ? t.x : 0.0;
      o.r[i].y = (ok & (c2 + 1 < nlim)) ? t.y : 0.0;
    }
  }
}
// AMASK: A keeps k <= m.  BSRC 0: B from `ob`; 1: B from `ob`, keeping k >= bcol0 + n (the lower-triangular inverse of a leaf);
// 2: B is the strip held in the accumulators `tb` (the result of a previous product, rows = k).
template <bool AMASK, int BSRC>
__device__ __forceinline__ void strip_mma(StripAcc& acc, const StripA& oa, const StripB& ob, const StripAcc& tb, int bcol0,
                                          double* lds, int tid) {
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int rb0 = wave, rb1 = 7 - wave;
  auto stage = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    double* sA = lds + (c & 1) * P_BUF;
    double* sB = sA + PBK * PLDA;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int row = wave + 4 * ii, c2 = (tid & 63) << 1, k = PBK * c + row;
      v2d t = oa.r[4 * c + ii];
      if (AMASK) {
        t.x = (k <= c2) ? t.x : 0.0;
        t.y = (k <= c2 + 1) ? t.y : 0.0;
      }
      *reinterpret_cast<v2d*>(sA + row * PLDA + c2) = t;
    }
    if (BSRC < 2) {
      const int row = tid >> 4, c2 = (tid & 15) << 1, k = PBK * c + row;
      v2d t = ob.r[c];
      if (BSRC == 1) {
        t.x = (k >= bcol0 + c2) ? t.x : 0.0;
        t.y = (k >= bcol0 + c2 + 1) ? t.y : 0.0;
      }
      *reinterpret_cast<v2d*>(sB + row * PLDB + c2) = t;
    } else {
      // rows 16 c .. 16 c + 15 of the strip in `tb` are row block c: wave c (its block 0) or wave 7 - c (its block 1)
      constexpr int a = c < 4 ? 0 : 1;
      if (wave == (c < 4 ? c : 7 - c)) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) sB[(4 * v + lk) * PLDB + 16 * b + li] = tb.c[a][b][v];
      }
    }
  };
  auto compute = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const double* sA = lds + (c & 1) * P_BUF;
    const double* sB = sA + PBK * PLDA;
    const bool do0 = !AMASK || c <= rb0, do1 = !AMASK || c <= rb1;  // (wave-uniform)
    const double* pa = sA + lk * PLDA + li;
    const double* pb = sB + lk * PLDB + li;
    if (do1) {
#pragma unroll
      for (int kk = 0; kk < PBK / 4; ++kk) {
        const double b0 = pb[4 * kk * PLDB], b1 = pb[4 * kk * PLDB + 16];
        const double a1 = pa[4 * kk * PLDA + 16 * rb1];
        if (do0) {
          const double a0 = pa[4 * kk * PLDA + 16 * rb0];
          acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.c[0][0], 0, 0, 0);
          acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.c[0][1], 0, 0, 0);
        }
        acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.c[1][0], 0, 0, 0);
        acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.c[1][1], 0, 0, 0);
      }
    }
  };
  auto step = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (c + 1 < 128 / PBK) stage(std::integral_constant<int, c + 1>{});
    compute(cc);
    __syncthreads();
  };
  stage(std::integral_constant<int, 0>{});
  __syncthreads();
  step(std::integral_constant<int, 0>{});
  step(std::integral_constant<int, 1>{});
  step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 3>{});
  step(std::integral_constant<int, 4>{});
  step(std::integral_constant<int, 5>{});
  step(std::integral_constant<int, 6>{});
  step(std::integral_constant<int, 7>{});
}
__device__ __forceinline__ void strip_zero(StripAcc& acc) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc.c[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// Wait until *flag >= target (thread 0 polls; everybody learns the outcome).  false: the launch is being abandoned.
// The wait is bounded by the 100 MHz constant clock (s_memrealtime), not by a poll count: `budget` ticks of 10 ns, whatever the
// core clock and however slow a poll is beside a kernel that saturates memory.  The wait that gives up leaves 1 + the milliseconds
// it waited in PF_ABORT; the chain work-group reports them in the low bits of the status word.
__device__ __forceinline__ bool panel_wait(int* flag, int target, int* flags, int tid, long long budget) {
  __shared__ int s_ok;
  if (tid == 0) {
    int ok = 1;
    long long t0 = 0;
    bool timed = false;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (!timed) {  // the clock is read only by waits that actually wait
        t0 = (long long)wall_clock64();
        timed = true;
      }
      const long long waited = (long long)wall_clock64() - t0;
      if (__hip_atomic_load(flags + PF_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || waited > budget) {
        ok = 0;
#ifdef GPP_PANEL_STAMP
        if (waited > budget) {  // the wait that gave up
          flags[PF_INTS + 8] = (int)blockIdx.x;
          flags[PF_INTS + 9] = (int)(flag - flags);
          flags[PF_INTS + 10] = target;
          flags[PF_INTS + 11] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        if (waited > budget) {
          const long long ms = waited / 100000;
          __hip_atomic_store(flags + PF_ABORT, 1 + (int)(ms > 0xFFFFF ? 0xFFFFF : ms), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_ok = ok;
  }
  __syncthreads();
  const int ok = s_ok;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // what the publisher wrote before raising the flag is visible from here
  __syncthreads();
  return ok != 0;
}
// Everything this work-group has written becomes visible to whoever sees the flag change.
__device__ __forceinline__ void panel_publish(int* flag, int tid) {
  __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have left the CU
  __syncthreads();
  if (tid == 0) {
    // The explicit wait behind the fence is REQUIRED: for `fence release` (or a release atomic) hipcc 7.2 emits buffer_wbl2 here
    // WITHOUT the s_waitcnt vmcnt(0) that has to separate the write-back from the atomic (its counter bookkeeping does not see
    // the write-back as outstanding), so a flag could overtake the data it publishes — seen as one wrong factor in ~50 000
    // launches, only beside a stream that saturates memory (tools/stress_panel.py; the ISA: buffer_wbl2 sc1, global_atomic_add)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0);
    __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A work-group is done with the flag block; the last one of the launch to say so zeroes it for the next launch.
__device__ __forceinline__ void panel_leave(int* flags, int tid) {
  __shared__ int s_last;
  __syncthreads();
  if (tid == 0) {
    // release + relaxed add + acquire written out, with the same explicit wait as panel_publish: an ACQ_REL atomic is lowered to
    // buffer_wbl2 + atomic and is exposed to the same missing s_waitcnt (tests/test_host_cpu.py checks every write-back of the file)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0);
    s_last = __hip_atomic_fetch_add(flags + PF_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (s_last)
    for (int i = tid; i < PF_INTS; i += 256) __hip_atomic_store(flags + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void gpp_panel_potrf_inv(double* __restrict__ A, int64_t lda, double* __restrict__ Linv,
                                                           int64_t ldi, int n, int32_t* info, int row_offset, int* flags,
                                                           long long budget) {
  extern __shared__ __attribute__((aligned(16))) double img[];
  const int tid = threadIdx.x;
  const int C = (n + 127) >> 7, nl = n - 128 * (C - 1);  // leaves; rows of the last one (1 .. 128)
  auto ext = [&](int t) { return t == C - 1 ? nl : 128; };
#ifdef GPP_PANEL_STAMP
  if (blockIdx.x == 0 && tid == 0) {  // what did the launch find in its flag block?
    flags[PF_INTS + 0] = __hip_atomic_load(flags + PF_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flags[PF_INTS + 1] = __hip_atomic_load(flags + PF_LEAF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flags[PF_INTS + 2] = __hip_atomic_load(flags + PF_LEAF + C - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flags[PF_INTS + 3] += 1;  // launches on this block since the last memset that reached it
  }
#endif
  if (blockIdx.x == 0) {  // the chain
    for (int j = 0; j < C; ++j) {
      PSTAMP(4 * j);
      if (j > 0 && !panel_wait(flags + PF_DIAG + j, 4, flags, tid, budget)) break;
      PSTAMP(4 * j + 1);
      leaf_body(A + (int64_t)(128 * j) * lda + 128 * j, lda, Linv + (int64_t)(128 * j) * ldi + 128 * j, ldi, ext(j), info,
                row_offset + 128 * j);
      PSTAMP(4 * j + 2);
      panel_publish(flags + PF_LEAF + j, tid);
      PSTAMP(4 * j + 3);
    }
    if (tid == 0) {  // status: 2^30 + the milliseconds the abandoned wait had waited (capped at 2^20 - 1)
      const int ab = __hip_atomic_load(flags + PF_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ab != 0) atomicCAS(info, 0, GPP_INFO_PANEL_TIMEOUT | ((ab - 1) & 0xFFFFF));
    }
    panel_leave(flags, tid);
    return;
  }
  const int W = (int)gridDim.x - 1, w = (int)blockIdx.x - 1, nf = 4 * (C - 1), ns = 2 * nf;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int rbase[2] = {16 * wave, 16 * (7 - wave)};  // first rows of this wave's two 16-row blocks (see StripAcc)
  auto tileA = [&](int r, int c) { return A + (int64_t)(128 * r) * lda + 128 * c; };
  auto tileI = [&](int r, int c) { return Linv + (int64_t)(128 * r) * ldi + 128 * c; };
  StripA oa;
  StripB ob;
  StripAcc none;
  strip_zero(none);
  // row j of the inverse, strip (i, q)
  auto inverse_row = [&](int j, int i, int q) {
    StripAcc t;
    strip_zero(t);
    const int ej = ext(j);  // (i < j: tile column i is never the ragged one)
    for (int m = i; m < j; ++m) {
      strip_load_a(oa, tileA(m, j), lda, tid, 128, ej);
      strip_load_b(ob, tileI(m, i) + 32 * q, ldi, tid);
      if (m == i) strip_mma<false, 1>(t, oa, ob, none, 32 * q, img, tid);
      else strip_mma<false, 0>(t, oa, ob, none, 0, img, tid);
    }
    strip_load_a(oa, tileI(j, j), ldi, tid, ej, ej);
    StripAcc x;
    strip_zero(x);
    strip_mma<true, 2>(x, oa, ob, t, 0, img, tid);
    double* lo = tileI(j, i) + 32 * q;                   // Linv[j,i] strip: rows m, columns n
    double* up = tileI(i, j) + (int64_t)(32 * q) * ldi;  // its mirror: rows n, columns m
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int m = rbase[a] + 4 * v + lk, nn = 16 * b + li;
          const double val = -x.c[a][b][v];
          if (m < ej) {
            lo[(int64_t)m * ldi + nn] = val;
            up[(int64_t)nn * ldi + m] = val;
          }
        }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();  // (the next row of this strip reads what this one wrote)
  };
  bool ok = true;
  for (int j = 0; j < C && ok; ++j) {
    // S: the strips of block row j
    for (int s = w; s < nf && ok; s += W) {
      const int c = 1 + (s >> 2), q = s & 3;
      if (c <= j) continue;
      const bool stamp = (c == j + 1 && q == 0);
      double* Bp = tileA(j, c) + 32 * q;
      const int nlim = ext(c) - 32 * q;  // valid columns of this strip (<= 0: nothing to do but to be counted)
      strip_load_b(ob, Bp, lda, tid, 128, nlim);  // this work-group's own data (its update of the previous step): fetched while the leaf runs
      if (stamp) PSTAMP(128 + 8 * j);
      ok = panel_wait(flags + PF_LEAF + j, 1, flags, tid, budget);
      if (!ok) break;
      if (stamp) PSTAMP(128 + 8 * j + 1);
      strip_load_a(oa, tileI(j, j), ldi, tid);
      StripAcc x;
      strip_zero(x);
      strip_mma<true, 0>(x, oa, ob, none, 0, img, tid);
      if (stamp) PSTAMP(128 + 8 * j + 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (16 * b + li < nlim) Bp[(int64_t)(rbase[a] + 4 * v + lk) * lda + 16 * b + li] = x.c[a][b][v];
      panel_publish(flags + PF_SOLVED + j * PMAXC + c, tid);
      if (stamp) PSTAMP(128 + 8 * j + 3);
    }
    // U: rank-128 update of the strips below block row j
    for (int s = w; s < nf && ok; s += W) {
      const int c = 1 + (s >> 2), q = s & 3;
      if (c <= j) continue;
      const bool stamp = (c == j + 1 && q == 0);
      const int nlim = ext(c) - 32 * q;
      strip_load_b(ob, tileA(j, c) + 32 * q, lda, tid, 128, nlim);  // own S result
      for (int r = j + 1; r <= c && ok; ++r) {
        double* Cp = tileA(r, c) + 32 * q;
        const int er = ext(r);
        double cold[2][2][4];  // own data as well: fetched before the wait
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int m = rbase[a] + 4 * v + lk, nn = 16 * b + li;
              cold[a][b][v] = (m < er && nn < nlim) ? Cp[(int64_t)m * lda + nn] : 0.0;
            }
        ok = panel_wait(flags + PF_SOLVED + j * PMAXC + r, 4, flags, tid, budget);
        if (!ok) break;
        if (stamp) PSTAMP(128 + 8 * j + 4);
        strip_load_a(oa, tileA(j, r), lda, tid, 128, er);
        StripAcc x;
        strip_zero(x);
        strip_mma<false, 0>(x, oa, ob, none, 0, img, tid);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int m = rbase[a] + 4 * v + lk, nn = 16 * b + li;
              // (diagonal tile: upper triangle only)
              if (m < er && nn < nlim && (r < c || m <= 32 * q + nn)) Cp[(int64_t)m * lda + nn] = cold[a][b][v] - x.c[a][b][v];
            }
      }
      if (stamp) PSTAMP(128 + 8 * j + 5);
      if (ok && c == j + 1) panel_publish(flags + PF_DIAG + c, tid);
      else { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
      if (stamp) PSTAMP(128 + 8 * j + 6);
    }
    // inverse: row j-1 (row C-1 after the loop)
    for (int s = w; s < ns && ok; s += W) {
      if (s < nf) continue;
      const int i = (s - nf) >> 2, q = s & 3, jj = j - 1;
      if (jj < 1 || i >= jj) continue;
      ok = panel_wait(flags + PF_LEAF + jj, 1, flags, tid, budget);
      if (ok) inverse_row(jj, i, q);
    }
  }
  for (int s = w; s < ns && ok; s += W) {
    if (s < nf) continue;
    const int i = (s - nf) >> 2, q = s & 3, jj = C - 1;
    if (jj < 1 || i >= jj) continue;
    ok = panel_wait(flags + PF_LEAF + jj, 1, flags, tid, budget);
    if (ok) inverse_row(jj, i, q);
  }
  panel_leave(flags, tid);
}

}  // namespace

#ifdef GPP_PANEL_STAMP
extern "C" int gpp_debug_panel_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_panel_stamps), sizeof(unsigned long long) * 512);
}
#endif
namespace {
__global__ void gpp_fill_i32(int32_t* p, int n, int32_t value) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = value;
}
}  // namespace
hipError_t gpp_launch_fill_i32(hipStream_t s, int32_t* p, int n, int32_t value) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gpp_fill_i32, dim3((unsigned)std::min(64, (n + 255) / 256)), dim3(256), 0, s, p, n, value);
  return hipGetLastError();
}

size_t gpp_panel_flag_bytes() { return (((size_t)PF_INTS * sizeof(int) + 255) / 256) * 256; }
int gpp_panel_max_leaves() { return PMAXC; }

// n <= 128 * 32 rows (the last leaf may be ragged); `flags` = gpp_panel_flag_bytes() bytes of ZEROED device memory that no other launch
// in flight uses (the launch leaves them zeroed again);
// `max_wgs` = work-groups the stream's CUs hold at one each (the grid never exceeds it: see the kernel's comment).
hipError_t gpp_launch_panel(hipStream_t s, double* A, int64_t lda, double* Linv, int64_t ldi, int n, int32_t* info,
                            int row_offset, int* flags, int max_wgs, int timeout_ms) {
  if (n <= 0) return hipSuccess;
  const int C = (n + NB - 1) / NB;
  if (C > PMAXC || max_wgs < 2 || !flags) return hipErrorInvalidValue;
  static std::atomic<bool> attr_set[64];
  // the leaf's 72 KiB image is the larger tenant: like the leaf alone, a panel work-group fits the slot a finished GEMM work-group
  // leaves behind on a CU (see gpp_leaf_potrf_inv)
  const size_t shmem = (size_t)NT * TSZ * sizeof(double);
  static_assert((size_t)2 * P_BUF <= (size_t)NT * TSZ, "the strip buffers must fit the leaf image");
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gpp_panel_potrf_inv), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int workers = C > 1 ? std::min(8 * (C - 1), max_wgs - 1) : 0;
  const long long budget = (long long)(timeout_ms > 0 ? timeout_ms : 500) * 100000;  // ticks of the 100 MHz constant clock
  hipLaunchKernelGGL(gpp_panel_potrf_inv, dim3((unsigned)(1 + workers)), dim3(256), shmem, s, A, lda, Linv, ldi, n, info, row_offset,
                     flags, budget);
  return hipGetLastError();
}

hipError_t gpp_launch_leaf(hipStream_t s, double* A, int64_t lda, double* Linv, int64_t ldi, int n, int32_t* info,
                           int row_offset, int batch, int64_t sA, int64_t sLi) {
  if (n <= 0 || batch <= 0) return hipSuccess;
  if (n > NB) return hipErrorInvalidValue;
  static std::atomic<bool> attr_set[64];  // per device (function attributes are per device)
  const size_t shmem = (size_t)NT * TSZ * sizeof(double);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gpp_leaf_potrf_inv), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(gpp_leaf_potrf_inv, dim3((unsigned)batch), dim3(256), shmem, s, A, lda, Linv, ldi, n, info, row_offset, sA,
                     sLi);
  return hipGetLastError();
}
