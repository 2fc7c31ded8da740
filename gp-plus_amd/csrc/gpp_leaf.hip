// gpp_leaf.hip — diagonal leaf of the recursive Cholesky: factor an n x n (n <= 128) SPD block and invert its
// factor inside ONE work-group, with the block held in MFMA accumulator registers and LDS.  Replaces the unblocked
// LAPACK potf2/trti2 steps inside torch.linalg.cholesky_ex (reference call site: gpytorch psd_safe_cholesky reached
// from optim/mll_torch.py:116).  It sits on the critical path N/128 times per factorisation, so it is latency-tuned.
//
//   in : A[n x n], UPPER triangle read (A = U^T U with U stored row-major = L stored column-major; the strict lower
//        triangle is never touched).  Internally the kernel works on L = U^T: every access to A swaps its indices.
//   out: A    <- U (upper triangle),  Linv block <- inv(L) in the lower triangle AND inv(L)^T mirrored in the strict
//        upper triangle (the mirror lets the inverse enter later products as a row-contiguous "TN" operand)
//        *info <- row_offset + k + 1 for the first non-positive / NaN pivot (kept if already non-zero)
//
// Layout: the 128 x 128 block is an 8 x 8 grid of 16 x 16 tiles; the 36 lower tiles are dealt to the 4 waves
// (9 accumulator tiles = 72 VGPRs each, v_mfma_f64_16x16x4_f64 C/D layout).  Right-looking over tile columns s:
//   (1) owners park column s in LDS;  (2) wave 0 factors the 16 x 16 diagonal tile and inverts it with lane-per-row
//   registers and v_readlane broadcasts (no LDS round trips inside the 16 sequential pivots);  (3) panel tiles
//   become L(i,s) = raw(i,s) * inv(L_ss)^T on the MFMA;  (4) trailing tiles acc(i,j) -= L(i,s) L(j,s)^T on the MFMA.
// The inverse is then assembled by pair merging at tile level (sizes 16, 32, 64): X21 = -X22 (L21 X11), MFMA again.
#include "gpp_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

constexpr int NB = GPP_TILE;
constexpr int TSZ = 16 * 16;       // doubles per LDS tile (unpadded: the whole image must fit ONE GEMM LDS slot)
constexpr int NT = 36;             // lower tiles of an 8x8 grid
constexpr int SLOTS = 9;           // tiles per wave

__device__ __forceinline__ int toff(int i, int j) { return (i * (i + 1) / 2 + j) * TSZ; }

__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// Element (row, col) of a 16 x 16 LDS tile.  Rows are 128 bytes; the XOR swizzle (col ^ row>>1) spreads a column
// read (16 rows, same col) over 16 distinct 8-byte bank slots without padding the tile.
__device__ __forceinline__ int tix(int row, int col) { return row * 16 + (col ^ (row >> 1)); }

// A-operand fragment of tile[m][k] (also the B operand of an "X * tile^T" product): lane (m = l&15, k = k0 + l>>4)
__device__ __forceinline__ double frag_rk(const double* tile, int k0, int lane) {
  return tile[tix(lane & 15, k0 + (lane >> 4))];
}
// B-operand fragment of tile[k][n]: lane (n = l&15, k = k0 + l>>4)
__device__ __forceinline__ double frag_kn(const double* tile, int k0, int lane) {
  return tile[tix(k0 + (lane >> 4), lane & 15)];
}
__device__ __forceinline__ void store_acc(double* tile, const v4d& a, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[tix((lane >> 4) + 4 * r, lane & 15)] = a[r];
}

__global__ __launch_bounds__(256) void gpp_leaf_potrf_inv(double* __restrict__ A, int64_t lda, double* __restrict__ Linv,
                                                          int64_t ldi, int n, int32_t* info, int row_offset) {
  // ONE 36-tile image of exactly 72 KiB: LDS is allocated contiguously, so on the look-ahead stream the leaf can only
  // start beside a running GEMM work-group if it fits the 72.5 KiB slot a finished GEMM work-group leaves behind.  Slot (i,j) holds, in turn: the parked raw tile, L(i,j) (off-diagonal) or inv(L_jj) (diagonal),
  // and finally inv(L)(i,j): L(i,j) is consumed exactly at the merge level that overwrites it.
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Limg = lds;
  double* Ximg = lds;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;

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

  for (int s = 0; s < 8; ++s) {
    // (1) park column s
#pragma unroll
    for (int q = 0; q < SLOTS; ++q)
      if (tj[q] == s) store_acc(Limg + toff(ti[q], s), acc[q], lane);
    __syncthreads();

    // (2) wave 0: factor + invert the diagonal tile.  lane (l & 15) owns row i of L and column i of X.
    if (wave == 0) {
      double* D = Limg + toff(s, s);
      const int i = lc;
      double a[16], x[16], rd[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = D[tix(i, c)];
      int bad = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        double akk = readlane_d(a[k], k);
        if (!(akk > 0.0)) {
          if (!bad) bad = k + 1;
          akk = 1.0;
        }
        const double d = sqrt(akk);
        const double r = 1.0 / d;
        rd[k] = r;
        const double l = (i == k) ? d : a[k] * r;
        a[k] = l;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) {
          const double ljk = readlane_d(l, j);
          a[j] = fma(-l, ljk, a[j]);
        }
      }
      if (bad && lane == 0) atomicCAS(info, 0, row_offset + 16 * s + bad);
      // X = inv(L): lane owns column i;  x[r] = (delta - sum_{k<r} L[r][k] x[k]) / L[r][r]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double sacc = (r == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) {
          const double lrk = readlane_d(a[k], r);
          sacc = fma(-lrk, x[k], sacc);
        }
        x[r] = sacc * rd[r];
      }
      if (lane < 16) {
        const int grow = 16 * s + i;
        double* Xd = Ximg + toff(s, s);  // == D: the raw diagonal tile was read into registers above
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const double lv = (c <= i) ? a[c] : 0.0;
          Xd[tix(c, i)] = x[c];  // X[c][i]
          if (grow < n && c <= i) A[(int64_t)(16 * s + c) * lda + grow] = lv;  // U[col][row] = L[row][col]
        }
      }
    }
    __syncthreads();

    // (3) panel: L(i,s) = raw(i,s) * X_ss^T
    {
      const double* Xd = Ximg + toff(s, s);
      double bx[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) bx[kk] = frag_rk(Xd, 4 * kk, lane);
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) {
        if (tj[q] == s && ti[q] > s) {
          double* P = Limg + toff(ti[q], s);
          v4d o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) o = __builtin_amdgcn_mfma_f64_16x16x4f64(frag_rk(P, 4 * kk, lane), bx[kk], o, 0, 0, 0);
          store_acc(P, o, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti[q] + lr + 4 * r;
            if (row < n) A[(int64_t)(16 * s + lc) * lda + row] = o[r];  // U[col][row] = L[row][col]
          }
        }
      }
    }
    __syncthreads();

    // (4) trailing update
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      if (tj[q] > s) {
        const double* Pi = Limg + toff(ti[q], s);
        const double* Pj = Limg + toff(tj[q], s);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag_rk(Pi, 4 * kk, lane), frag_rk(Pj, 4 * kk, lane), acc[q], 0, 0, 0);
      }
    }
  }
  __syncthreads();

  // ---- inverse by pair merging: half-size h tiles, pairs based at b = 2h*p -------------------------------
  for (int h = 1; h <= 4; h <<= 1) {
    const int ntile = 4 * h;  // output tiles at this level
    v4d t[4];
    // phase a: T(i,j) = sum_{k=j}^{b+h-1} L(i,k) X(k,j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = wave + 4 * q;
      t[q] = (v4d){0.0, 0.0, 0.0, 0.0};
      if (e < ntile) {
        const int p = e / (h * h), rem = e - p * h * h;
        const int b = 2 * h * p, i = b + h + rem / h, j = b + rem % h;
        for (int k = j; k < b + h; ++k) {
          const double* Lt = Limg + toff(i, k);
          const double* Xt = Ximg + toff(k, j);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            t[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag_rk(Lt, 4 * kk, lane), frag_kn(Xt, 4 * kk, lane), t[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // every L(i,k) of this level has been read: the slots may now take T
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = wave + 4 * q;
      if (e < ntile) {
        const int p = e / (h * h), rem = e - p * h * h;
        const int b = 2 * h * p, i = b + h + rem / h, j = b + rem % h;
        store_acc(Ximg + toff(i, j), t[q], lane);
      }
    }
    __syncthreads();
    // phase b: X(i,j) = -sum_{k=b+h}^{i} X(i,k) T(k,j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = wave + 4 * q;
      t[q] = (v4d){0.0, 0.0, 0.0, 0.0};
      if (e < ntile) {
        const int p = e / (h * h), rem = e - p * h * h;
        const int b = 2 * h * p, i = b + h + rem / h, j = b + rem % h;
        for (int k = b + h; k <= i; ++k) {
          const double* Xa = Ximg + toff(i, k);
          const double* Tt = Ximg + toff(k, j);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            t[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag_rk(Xa, 4 * kk, lane), frag_kn(Tt, 4 * kk, lane), t[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = wave + 4 * q;
      if (e < ntile) {
        const int p = e / (h * h), rem = e - p * h * h;
        const int b = 2 * h * p, i = b + h + rem / h, j = b + rem % h;
        store_acc(Ximg + toff(i, j), t[q], lane);
      }
    }
    __syncthreads();
  }

  // write inv(L) into the lower triangle of the block and its transpose into the strict upper triangle
  for (int e = tid; e < n * n; e += 256) {
    const int row = e / n, col = e - row * n;
    const int hi = row > col ? row : col, lo = row > col ? col : row;
    Linv[(int64_t)row * ldi + col] = Ximg[toff(hi >> 4, lo >> 4) + tix(hi & 15, lo & 15)];
  }
}

}  // namespace

hipError_t gpp_launch_leaf(hipStream_t s, double* A, int64_t lda, double* Linv, int64_t ldi, int n, int32_t* info,
                           int row_offset) {
  if (n <= 0) return hipSuccess;
  if (n > NB) return hipErrorInvalidValue;
  static bool attr_set = false;
  const size_t shmem = (size_t)NT * TSZ * sizeof(double);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gpp_leaf_potrf_inv),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(gpp_leaf_potrf_inv, dim3(1), dim3(256), shmem, s, A, lda, Linv, ldi, n, info, row_offset);
  return hipGetLastError();
}
