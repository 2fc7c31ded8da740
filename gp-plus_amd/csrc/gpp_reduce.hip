// gpp_reduce.hip — the O(N^2), HBM-bound reductions of the exact-GP path.
//
// Reference call sites replaced:
//   optim/mll_torch.py:116  MultivariateNormal.log_prob -> gpytorch inv_quad_logdet (triangular solve, diag log-sum)
//   optim/mll_torch.py:117  loss.backward(): the backward of exp / distance / outputscale / noise-add, i.e.
//                           dMLL/dtheta = sum_ij W_ij dKy_ij/dtheta with W = 0.5 (alpha alpha' - Ky^-1)
//   models/gpregression.py:142-147  predictive mean and variance
// Every matrix is streamed exactly once; all reductions are two-stage and deterministic (no float atomics).
#include "gpp_internal.h"

typedef double v2d __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// y_i = sum_{k<=i} T[i][k] x_k : one wave per row, rows interleaved over waves so long and short rows mix.
__global__ __launch_bounds__(256) void gpp_trmv_lower(const double* __restrict__ T, int64_t ldt, int64_t N,
                                                      const double* __restrict__ x, double* __restrict__ y, int64_t sT,
                                                      int64_t sv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= N) return;
  T += (int64_t)blockIdx.y * sT;  // batch element
  x += (int64_t)blockIdx.y * sv;
  y += (int64_t)blockIdx.y * sv;
  const double* row = T + i * ldt;
  double acc = 0.0;
  const int64_t len = i + 1;
  const int64_t len2 = len & ~(int64_t)1;
  for (int64_t k = 2 * lane; k < len2; k += 128) {
    const v2d t = *reinterpret_cast<const v2d*>(row + k);
    const v2d xx = *reinterpret_cast<const v2d*>(x + k);
    acc = fma(t.x, xx.x, acc);
    acc = fma(t.y, xx.y, acc);
  }
  if (lane == 0 && (len & 1)) acc = fma(row[len - 1], x[len - 1], acc);
  acc = wave_sum(acc);
  if (lane == 0) y[i] = acc;
}

// y_j = sum_{i>=j} T[j][i] x_i for the UPPER triangle of T: one wave per row (the mirror of gpp_trmv_lower).
__global__ __launch_bounds__(256) void gpp_trmv_upper(const double* __restrict__ T, int64_t ldt, int64_t N,
                                                      const double* __restrict__ x, double* __restrict__ y, int64_t sT,
                                                      int64_t sv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + wave;
  if (j >= N) return;
  T += (int64_t)blockIdx.y * sT;  // batch element
  x += (int64_t)blockIdx.y * sv;
  y += (int64_t)blockIdx.y * sv;
  const double* row = T + j * ldt;
  double acc = 0.0;
  int64_t k0 = j;
  if (k0 & 1) {  // align the vector loop to an even column
    if (lane == 0) acc = fma(row[k0], x[k0], acc);
    ++k0;
  }
  const int64_t nvec = (N - k0) >> 1;
  for (int64_t v = lane; v < nvec; v += 64) {
    const v2d t = *reinterpret_cast<const v2d*>(row + k0 + 2 * v);
    const v2d xx = *reinterpret_cast<const v2d*>(x + k0 + 2 * v);
    acc = fma(t.x, xx.x, acc);
    acc = fma(t.y, xx.y, acc);
  }
  if (lane == 0 && ((N - k0) & 1)) acc = fma(row[N - 1], x[N - 1], acc);
  acc = wave_sum(acc);
  if (lane == 0) y[j] = acc;
}

// Column-sharded products with a lower-triangular T of which this rank holds only the column blocks it owns (block-cyclic:
// columns [b*nb, (b+1)*nb) with b % nranks == rank): the sharded evaluation's z = L^-1 r and alpha = L^-T z from the owned
// column blocks of L^-1 (the caller all-reduces the partial results).
//   y_i = sum over owned columns k <= i of T[i][k] x_k        (one wave per row; other columns are never read)
// compact: T holds ONLY the owned column blocks, side by side (the q-th owned block, global block rank + q nranks, in columns
// [q nb, (q+1) nb)): N x (N / nranks) doubles per rank instead of N x N.
__global__ __launch_bounds__(256) void gpp_trmv_lower_cols(const double* __restrict__ T, int64_t ldt, int64_t N,
                                                           const double* __restrict__ x, double* __restrict__ y, int64_t nb,
                                                           int rank, int nranks, int compact) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= N) return;
  const double* row = T + i * ldt;
  double acc = 0.0;
  int64_t q = 0;
  for (int64_t c0 = (int64_t)rank * nb; c0 <= i; c0 += (int64_t)nranks * nb, ++q) {
    const int64_t c1 = (c0 + nb < i + 1) ? c0 + nb : i + 1;  // nb is even: c0 is, so the 16-byte loads are aligned
    const int64_t len = c1 - c0, len2 = len & ~(int64_t)1;
    const double* seg = row + (compact ? q * nb : c0);
    for (int64_t k = 2 * lane; k < len2; k += 128) {
      const v2d t = *reinterpret_cast<const v2d*>(seg + k);
      const v2d xx = *reinterpret_cast<const v2d*>(x + c0 + k);
      acc = fma(t.x, xx.x, acc);
      acc = fma(t.y, xx.y, acc);
    }
    if (lane == 0 && (len & 1)) acc = fma(seg[len - 1], x[c1 - 1], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) y[i] = acc;
}

//   y_k = sum_{i >= k} T[i][k] x_i   for the owned columns k, 0 for the others.  Work-group (g, r): 64 columns (lane = column,
//   coalesced along the row) x the r-th chunk of rows, the four waves striding over the rows with eight loads in flight each;
//   the chunks' partial sums are added in a fixed order by the second kernel (deterministic, no float atomics).
constexpr int TRT_CHUNKS = 16;
__global__ __launch_bounds__(256) void gpp_trmv_lower_t_cols(const double* __restrict__ T, int64_t ldt, int64_t N,
                                                             const double* __restrict__ x, double* __restrict__ part, int64_t nb,
                                                             int rank, int nranks, int64_t chunk, int compact) {
  __shared__ double red[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t k0 = (int64_t)blockIdx.x * 64, k = k0 + lane;
  const bool owned = (int)((k0 / nb) % nranks) == rank;  // nb is a multiple of 64: a group of 64 columns has one owner
  double acc = 0.0;
  const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = (r0 + chunk < N) ? r0 + chunk : N;
  if (owned && k < N && r1 > k0) {
    int64_t i = (r0 > k0 ? r0 : k0) + wave;
    const double* p = T + (compact ? ((k0 / nb) / nranks) * nb + (k0 % nb) + lane : k);
    for (; i + 28 < r1; i += 32) {  // rows i, i+4, ..., i+28 of this wave: eight independent loads
      double t[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        t[u] = p[(i + 4 * u) * ldt];
        xv[u] = x[i + 4 * u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = (i + 4 * u >= k) ? fma(t[u], xv[u], acc) : acc;
    }
    for (; i < r1; i += 4)
      if (i >= k) acc = fma(p[i * ldt], x[i], acc);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && k < N) part[(int64_t)blockIdx.y * N + k] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ __launch_bounds__(256) void gpp_trmv_lower_t_finish(const double* __restrict__ part, int64_t N, int chunks,
                                                               double* __restrict__ y) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  double s = 0.0;
  for (int r = 0; r < chunks; ++r) s += part[(int64_t)r * N + k];
  y[k] = s;
}

// out3 = { quad = z'z, logdet = 2 sum log L_ii, mll = -0.5 (quad + logdet + N log 2pi) } : one work-group.
__global__ __launch_bounds__(1024) void gpp_mll_scalars(const double* __restrict__ L, int64_t ld, int64_t N,
                                                        const double* __restrict__ z, double* __restrict__ out3, int64_t sL,
                                                        int64_t sv) {
  __shared__ double sq[16], sl[16];
  L += (int64_t)blockIdx.x * sL;  // batch element
  z += (int64_t)blockIdx.x * sv;
  out3 += 3 * (int64_t)blockIdx.x;
  double q = 0.0, l = 0.0;
  for (int64_t i = threadIdx.x; i < N; i += 1024) {
    const double zi = z[i];
    q = fma(zi, zi, q);
    l += log(L[i * ld + i]);
  }
  q = wave_sum(q);
  l = wave_sum(l);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sq[wave] = q;
    sl[wave] = l;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double qq = 0.0, ll = 0.0;
    for (int w = 0; w < 16; ++w) {
      qq += sq[w];
      ll += sl[w];
    }
    ll *= 2.0;
    out3[0] = qq;
    out3[1] = ll;
    out3[2] = -0.5 * (qq + ll + (double)N * 1.8378770664093454835606594728112);  // log(2 pi)
  }
}

// ------------------------------------------------------------------------------------------------
// gradient reduction.  Tiles of 64x64 over the lower triangle of Kinv; 256 threads, 4x4 per thread.
// Per tile the kernel forms G_ij = mult * W_ij * sf2*k_ij  (W = 0.5(alpha_i alpha_j - Kinv_ij), mult = 2 for the
// strictly-lower entries, which stand for (i,j) and (j,i), 1 on the diagonal) and accumulates
//    g_w[d]  += G_ij * (-(u_id-u_jd)^2)          g_sf2 += mult * W_ij * k_ij          wdiag[i] = W_ii
//    g_U[i,d] += G_ij (-2 w_d)(u_id-u_jd)   and   g_U[j,d] += G_ij (-2 w_d)(u_jd-u_id)      (d < dU)
// Work-groups grid-stride over tiles; scalar sums go to one record per work-group, g_U row/column sums to the
// slot [partner tile][row] of a partial buffer (every slot is written exactly once).  Finish kernels sum both.
constexpr int GT = 64;
constexpr int GD_MAX = 64;
constexpr int GS_MAX = 64;
constexpr int G_WGS = 2048;

__device__ __forceinline__ void tile_from_index(int64_t t, int64_t& ti, int64_t& tj) {
  int64_t r = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  while (r * (r + 1) / 2 > t) --r;
  ti = r;
  tj = t - r * (r + 1) / 2;
}

// MAT = false: pure RBF product (kind 0): no Matern distance / derivative registers, two waves per SIMD
// HASU = false: no gradient w.r.t. feature columns (dU = 0): the 4 x 4 blocks of G are never materialised
template <int DT, bool MAT, bool HASU>
__global__ __launch_bounds__(256) void gpp_grad_tiles(const double* __restrict__ U, int64_t N, int D,
                                                      const double* __restrict__ w, const double* __restrict__ sf2p,
                                                      int kind, int d_split, const double* __restrict__ alpha, const double* __restrict__ Kinv,
                                                      int64_t ldk, int dU, int64_t ntiles, int shard_nb, int shard_rank,
                                                      int shard_nranks, int shard_cols, int64_t sU, int64_t sK, int64_t sv, int64_t ws_stride,
                                                      double* __restrict__ rec /* [gridDim.x][D+1] */,
                                                      double* __restrict__ wdiag /* [N] */,
                                                      double* __restrict__ gUpart /* [T][N][dU] */) {
  {  // batch element blockIdx.y: its own parameters, matrices and slice of the workspace
    const int64_t b = blockIdx.y;
    U += b * sU;
    w += b * D;
    sf2p += b;
    alpha += b * sv;
    Kinv += b * sK;
    rec += b * ws_stride;
    wdiag += b * ws_stride;
    gUpart += b * ws_stride;
  }
  __shared__ __attribute__((aligned(16))) double sa[DT * GT];  // raw U rows of tile-row ti, [d][r]
  __shared__ __attribute__((aligned(16))) double sb[DT * GT];
  __shared__ double sal_a[GT], sal_b[GT];
  __shared__ double sw[DT];
  __shared__ double red[256];
  __shared__ double rowpart[GT * 16];
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  const double sf2 = *sf2p;
  const GppExpConsts ec = gpp_exp_consts();
  if (tid < DT) sw[tid] = (tid < D) ? w[tid] : 0.0;

  // per-feature sums of this thread over all its tiles — in registers up to 16 features; beyond that (2 x DT registers, on top of
  // the 4 x 4 blocks of the feature gradients: <32,true,true> spilled 122 registers, <64,true,true> 645) every half-tile's
  // per-feature sum is reduced over the wave at once and kept in LDS, one accumulator per wave and feature
  constexpr bool WACC = (DT >= 32);
  __shared__ double wacc[WACC ? 4 * DT : 1];
  double my_sf2 = 0.0;
  double my_w[WACC ? 1 : DT];
#pragma unroll
  for (int d = 0; d < (WACC ? 1 : DT); ++d) my_w[d] = 0.0;
  if constexpr (WACC) {
    for (int e = tid; e < 4 * DT; e += 256) wacc[e] = 0.0;
  }

  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int64_t ti, tj;
    tile_from_index(t, ti, tj);
    const int64_t i0 = ti * GT, j0 = tj * GT;
    // row-sharded evaluation: this rank only holds (and reduces) the block rows of Kinv it owns (block-cyclic)
    // (shard_cols: the rank holds column blocks of Kinv instead — the back-substituted sharded evaluation)
    if (shard_nranks > 1 && (int)(((shard_cols ? j0 : i0) / shard_nb) % shard_nranks) != shard_rank) continue;
    const bool diag_tile = (ti == tj);
    __syncthreads();
    for (int e = tid; e < DT * GT; e += 256) {
      const int d = e / GT, r = e - d * GT;
      sa[e] = (d < D && i0 + r < N) ? U[(i0 + r) * D + d] : 0.0;
      sb[e] = (d < D && j0 + r < N) ? U[(j0 + r) * D + d] : 0.0;
    }
    if (tid < GT) {
      sal_a[tid] = (i0 + tid < N) ? alpha[i0 + tid] : 0.0;
      sal_b[tid] = (j0 + tid < N) ? alpha[j0 + tid] : 0.0;
    }
    __syncthreads();

    // G  = mult * W_ij * dK_ij/d(-r2_rbf)  (= mult W K for the RBF dims)
    // GM = mult * W_ij * dK_ij/d(-r2_mat)  for the Matern dims (d >= d_split), with K = sf2 e^{-r2_rbf} m(a):
    //      nu = 3/2: a = sqrt(6 r2_mat), m = (1+a)e^{-a},          dm/d(-r2_mat) = 3 e^{-a}
    //      nu = 5/2: a = sqrt(10 r2_mat), m = (1+a+a^2/3)e^{-a},   dm/d(-r2_mat) = (5/3)(1+a) e^{-a}
    const int dsp = MAT ? d_split : DT;
    typedef double v2d __attribute__((ext_vector_type(2)));
    double G[4][4], GM[4][4];  // (GM only lives in the Matern instantiation; G / GM feed the g_U part below)
    // Two half-tiles of 2 rows x 4 columns per thread, each with its own two passes over the features: the live set stays
    // at 8 distances + 8 Kinv entries + 8 G values (two waves per SIMD), and every LDS access is a 16-byte read of
    // consecutive rows / columns of one feature.  Round 4: a thread's columns are the PAIRS {2 tx, 2 tx + 1} and {32 + 2 tx, ...}, so
    // the 16 lanes of a ds_read_b128 cover 256 contiguous bytes (every bank once); with 4 consecutive columns per thread lanes
    // k and k + 8 met in the same banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.36, profiles/r03_sq_counters.txt).
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const int ra = 4 * ty + 2 * h;  // first of this half's two rows inside the tile
      double kin[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        // the thread's columns: the pairs {2 tx, 2 tx + 1} and {32 + 2 tx, 32 + 2 tx + 1} of the tile (see the LDS reads below)
        const int64_t i = i0 + ra + a, jb = j0 + 2 * tx;
        // (shard_cols == 2: Kinv holds only the owned column blocks, side by side — see gpp_trmv_lower_cols)
        const int64_t pjb = (shard_cols == 2) ? ((j0 / shard_nb) / shard_nranks) * shard_nb + (j0 % shard_nb) + 2 * tx : jb;
        const double* src = Kinv + (i < N ? i : N - 1) * ldk + pjb;
        if (pjb + 34 <= ldk) {  // two 16-byte loads, issued before the arithmetic (entries above the diagonal are never used)
          const v2d p0 = *reinterpret_cast<const v2d*>(src), p1 = *reinterpret_cast<const v2d*>(src + 32);
          kin[a][0] = p0.x; kin[a][1] = p0.y; kin[a][2] = p1.x; kin[a][3] = p1.y;
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b) kin[a][b] = (jb + (b & 1) + ((b >> 1) << 5) < N) ? src[(b & 1) + ((b >> 1) << 5)] : 0.0;
        }
      }
      double r2[2][4], r2m[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r2[a][b] = r2m[a][b] = 0.0;
#pragma unroll 4
      for (int d = 0; d < DT; ++d) {
        const v2d ua = *reinterpret_cast<const v2d*>(sa + d * GT + ra);
        const v2d b01 = *reinterpret_cast<const v2d*>(sb + d * GT + 2 * tx), b23 = *reinterpret_cast<const v2d*>(sb + d * GT + 32 + 2 * tx);
        const double ub[4] = {b01.x, b01.y, b23.x, b23.y};
        const double wd = sw[d];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double df = (a ? ua.y : ua.x) - ub[b];
            if (!MAT || d < dsp) r2[a][b] = fma(wd * df, df, r2[a][b]);
            else r2m[a][b] = fma(wd * df, df, r2m[a][b]);
          }
      }
      // compiler barrier: without it the second pass below re-uses the first pass's LDS reads (same addresses), i.e. keeps
      // 12 registers per feature alive across the exp section (DT = 16: 398 VGPRs, DT >= 32: spills)
      __asm__ volatile("" ::: "memory");
      double g2[2][4], gm2[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int64_t i = i0 + ra + a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int64_t j = j0 + 2 * tx + (b & 1) + ((b >> 1) << 5);
          double g = 0.0, gm = 0.0;
          if (i < N && j <= i) {
            const double er = gpp_exp_nonpos(-r2[a][b], ec);
            double kv = er, kd = 0.0;
            if (MAT && kind == 1) {
              const double aa = sqrt(6.0 * r2m[a][b]), ea = gpp_exp_nonpos(-aa, ec);
              kv = er * (1.0 + aa) * ea;
              kd = er * 3.0 * ea;
            } else if (MAT && kind == 2) {
              const double aa = sqrt(10.0 * r2m[a][b]), ea = gpp_exp_nonpos(-aa, ec);
              kv = er * (1.0 + aa + aa * aa * (1.0 / 3.0)) * ea;
              kd = er * (5.0 / 3.0) * (1.0 + aa) * ea;
            }
            const double Wij = 0.5 * (sal_a[ra + a] * sal_b[2 * tx + (b & 1) + ((b >> 1) << 5)] - kin[a][b]);
            const double mult = (i == j) ? 1.0 : 2.0;
            my_sf2 = fma(mult * Wij, kv, my_sf2);
            g = mult * Wij * sf2 * kv;
            gm = mult * Wij * sf2 * kd;
            if (i == j) wdiag[i] = Wij;
          }
          g2[a][b] = g;
          gm2[a][b] = gm;
        }
      }
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const v2d ua = *reinterpret_cast<const v2d*>(sa + d * GT + ra);
        const v2d b01 = *reinterpret_cast<const v2d*>(sb + d * GT + 2 * tx), b23 = *reinterpret_cast<const v2d*>(sb + d * GT + 32 + 2 * tx);
        const double ub[4] = {b01.x, b01.y, b23.x, b23.y};
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double df = (a ? ua.y : ua.x) - ub[b];
            s = fma(-((!MAT || d < dsp) ? g2[a][b] : gm2[a][b]) * df, df, s);
          }
        if constexpr (WACC) {
          const double t = wave_sum(s);
          if ((tid & 63) == 0) wacc[(tid >> 6) * DT + d] += t;
        } else {
          my_w[d] += s;
        }
      }
      if constexpr (HASU) {  // the manifold-gradient part below wants the whole 4 x 4 block
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if (h == 0) { G[a][b] = g2[a][b]; GM[a][b] = gm2[a][b]; }
            else { G[2 + a][b] = g2[a][b]; GM[2 + a][b] = gm2[a][b]; }
          }
      }
    }
    for (int d = 0; HASU && d < dU; ++d) {
      const double m2w = -2.0 * sw[d];
      const bool rbf_dim = !MAT || d < dsp;
      double rs[4], cs[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          s = fma(rbf_dim ? G[a][b] : GM[a][b], sa[d * GT + 4 * ty + a] - sb[d * GT + 2 * tx + (b & 1) + ((b >> 1) << 5)], s);
        rs[a] = s;
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
          s = fma(rbf_dim ? G[a][b] : GM[a][b], sb[d * GT + 2 * tx + (b & 1) + ((b >> 1) << 5)] - sa[d * GT + 4 * ty + a], s);
        cs[b] = s;
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a < 4; ++a) rowpart[(4 * ty + a) * 16 + tx] = rs[a];
      __syncthreads();
      double rsum = 0.0;
      if (tid < GT)
        for (int q = 0; q < 16; ++q) rsum += rowpart[tid * 16 + q];
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 4; ++b) rowpart[(2 * tx + (b & 1) + ((b >> 1) << 5)) * 16 + ty] = cs[b];
      __syncthreads();
      if (tid < GT) {
        double csum = 0.0;
        for (int q = 0; q < 16; ++q) csum += rowpart[tid * 16 + q];
        if (diag_tile) {
          if (i0 + tid < N) gUpart[((int64_t)ti * N + (i0 + tid)) * dU + d] = m2w * (rsum + csum);
        } else {
          if (i0 + tid < N) gUpart[((int64_t)tj * N + (i0 + tid)) * dU + d] = m2w * rsum;
          if (j0 + tid < N) gUpart[((int64_t)ti * N + (j0 + tid)) * dU + d] = m2w * csum;
        }
      }
    }
  }

  const int nrec = D + 1;
  for (int q = 0; q < nrec; ++q) {
    double v = my_sf2;
    if (q < D) {
      if constexpr (WACC) {
        v = ((tid & 63) == 0) ? wacc[(tid >> 6) * DT + q] : 0.0;
      } else {
#pragma unroll
        for (int d = 0; d < DT; ++d)
          if (q == d) v = my_w[d];
      }
    }
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) rec[(int64_t)blockIdx.x * nrec + q] = red[0];
  }
}

// q < D: g_w[q]; q == D: g_sf2; q > D: g_tau[q-D-1] = sum_{i in group} wdiag[i]
__global__ __launch_bounds__(256) void gpp_grad_finish(const double* __restrict__ rec, int nwg, int D, int S,
                                                       const double* __restrict__ wdiag, const int32_t* __restrict__ grp,
                                                       int64_t N, double* __restrict__ g_w, double* __restrict__ g_sf2,
                                                       double* __restrict__ g_tau, int64_t ws_stride) {
  __shared__ double red[256];
  rec += (int64_t)blockIdx.y * ws_stride;  // batch element
  wdiag += (int64_t)blockIdx.y * ws_stride;
  g_w += (int64_t)blockIdx.y * D;
  g_sf2 += blockIdx.y;
  g_tau += (int64_t)blockIdx.y * S;
  const int q = blockIdx.x, nrec = D + 1, tid = threadIdx.x;
  double v = 0.0;
  if (q <= D) {
    for (int b = tid; b < nwg; b += 256) v += rec[(int64_t)b * nrec + q];
  } else {
    const int s = q - D - 1;
    for (int64_t i = tid; i < N; i += 256)
      if ((grp ? grp[i] : 0) == s) v += wdiag[i];
  }
  red[tid] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    if (q < D) g_w[q] = red[0];
    else if (q == D) g_sf2[0] = red[0];
    else g_tau[q - D - 1] = red[0];
  }
}

__global__ __launch_bounds__(256) void gpp_gU_finish(const double* __restrict__ gUpart, int64_t N, int dU, int T,
                                                     double* __restrict__ g_U, int64_t ws_stride) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * dU) return;
  gUpart += (int64_t)blockIdx.y * ws_stride;  // batch element
  g_U += (int64_t)blockIdx.y * N * dU;
  double s = 0.0;
  for (int t = 0; t < T; ++t) s += gUpart[(int64_t)t * N * dU + e];
  g_U[e] = s;
}

// mean_a = sum_j Ksn[a][j] alpha_j ; var_a = kss_a - sum_j V[a][j]^2 : one wave per test row.
__global__ __launch_bounds__(256) void gpp_predict_rows(const double* __restrict__ Ksn, int64_t lds, const double* __restrict__ V,
                                                        int64_t ldv, int64_t M, int64_t N, const double* __restrict__ alpha,
                                                        const double* __restrict__ kss, double* __restrict__ mean_out,
                                                        double* __restrict__ var_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t a = (int64_t)blockIdx.x * 4 + wave;
  if (a >= M) return;
  double m = 0.0, q = 0.0;
  const double* kr = Ksn + a * lds;
  for (int64_t j = lane; j < N; j += 64) m = fma(kr[j], alpha[j], m);
  if (V) {
    const double* vr = V + a * ldv;
    for (int64_t j = lane; j < N; j += 64) q = fma(vr[j], vr[j], q);
  }
  m = wave_sum(m);
  q = wave_sum(q);
  if (lane == 0) {
    mean_out[a] = m;
    if (V && var_out) var_out[a] = kss[a] - q;
  }
}

}  // namespace

hipError_t gpp_launch_trmv_lower(hipStream_t s, const double* T, int64_t ldt, int64_t N, const double* x, double* y,
                                 int batch, int64_t sT, int64_t sv) {
  if (N <= 0 || batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(gpp_trmv_lower, dim3((unsigned)((N + 3) / 4), (unsigned)batch), dim3(256), 0, s, T, ldt, N, x, y, sT, sv);
  return hipGetLastError();
}

hipError_t gpp_launch_trmv_upper(hipStream_t s, const double* T, int64_t ldt, int64_t N, const double* x, double* y,
                                 int batch, int64_t sT, int64_t sv) {
  if (N <= 0 || batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(gpp_trmv_upper, dim3((unsigned)((N + 3) / 4), (unsigned)batch), dim3(256), 0, s, T, ldt, N, x, y, sT, sv);
  return hipGetLastError();
}

size_t gpp_trmv_t_ws_bytes(int64_t N) { return (size_t)TRT_CHUNKS * (size_t)N * sizeof(double); }

hipError_t gpp_launch_trmv_lower_cols(hipStream_t s, const double* T, int64_t ldt, int64_t N, const double* x, double* y,
                                      int64_t nb, int rank, int nranks, int trans, void* ws, size_t ws_bytes, int compact) {
  if (N <= 0) return hipSuccess;
  if (nb < 64 || nb % 64 != 0 || nranks < 1 || rank < 0 || rank >= nranks) return hipErrorInvalidValue;
  if (trans) {
    if (!ws || ws_bytes < gpp_trmv_t_ws_bytes(N)) return hipErrorInvalidValue;
    const int64_t chunk = (((N + TRT_CHUNKS - 1) / TRT_CHUNKS) + 31) / 32 * 32;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(gpp_trmv_lower_t_cols, dim3((unsigned)((N + 63) / 64), TRT_CHUNKS), dim3(256), 0, s, T, ldt, N, x, part, nb,
                       rank, nranks, chunk, compact);
    hipLaunchKernelGGL(gpp_trmv_lower_t_finish, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, part, N, TRT_CHUNKS, y);
  } else
    hipLaunchKernelGGL(gpp_trmv_lower_cols, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, T, ldt, N, x, y, nb, rank, nranks, compact);
  return hipGetLastError();
}

hipError_t gpp_launch_mll_scalars(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* z, double* out3,
                                  int batch, int64_t sL, int64_t sv) {
  if (batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(gpp_mll_scalars, dim3((unsigned)batch), dim3(1024), 0, s, L, ld, N, z, out3, sL, sv);
  return hipGetLastError();
}

size_t gpp_grad_ws_bytes(int64_t N, int D, int S, int dU) {
  (void)S;
  const int64_t T = (N + GT - 1) / GT;
  return (size_t)G_WGS * (D + 1) * sizeof(double) + (size_t)N * sizeof(double) +
         (size_t)T * N * (dU > 0 ? dU : 0) * sizeof(double) + 256;
}

hipError_t gpp_launch_grad_reduce(hipStream_t s, const double* U, int64_t N, int D, const double* w, const double* sf2,
                                  const int32_t* grp, int S, int kind, int d_split, const double* alpha,
                                  const double* Kinv, int64_t ldk, int dU, double* g_w, double* g_sf2, double* g_tau,
                                  double* g_U, void* ws, size_t ws_bytes, int shard_nb, int shard_rank, int shard_nranks,
                                  int batch, int64_t sU, int64_t sK, int64_t sv, int shard_cols) {
  if (kind < 0 || kind > 2) return hipErrorInvalidValue;
  if (batch < 1 || batch > 65535) return hipErrorInvalidValue;
  if (shard_nranks > 1 && (shard_nb < GT || shard_nb % GT != 0 || shard_rank < 0 || shard_rank >= shard_nranks))
    return hipErrorInvalidValue;
  if (D > GD_MAX || D < 1 || S > GS_MAX || S < 1 || dU > D || dU < 0) return hipErrorInvalidValue;
  if (ws_bytes < (size_t)batch * gpp_grad_ws_bytes(N, D, S, dU)) return hipErrorInvalidValue;
  const int64_t ws_stride = (int64_t)(gpp_grad_ws_bytes(N, D, S, dU) / sizeof(double));  // doubles per batch element
  const int T = (int)((N + GT - 1) / GT);
  const int64_t ntiles = (int64_t)T * (T + 1) / 2;
  const int nwg = (int)(ntiles < G_WGS ? ntiles : G_WGS);
  double* rec = reinterpret_cast<double*>(ws);
  double* wdiag = rec + (size_t)G_WGS * (D + 1);
  double* gUpart = wdiag + N;
  if (shard_nranks > 1) {  // skipped tiles leave their slots unwritten: start from zero
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)batch * ws_stride * sizeof(double), s);
    if (e != hipSuccess) return e;
  }
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(nwg, batch), dim3(256), 0, s, U, N, D, w, sf2, kind, d_split, alpha, Kinv, ldk, dU, ntiles,
                       shard_nb, shard_rank, shard_nranks, shard_cols, sU, sK, sv, ws_stride, rec, wdiag, gUpart);
  };
  const bool mat = kind != 0, hasu = dU > 0;
#define GPP_GT(DT)                                                                                                   \
  (mat ? (hasu ? launch(gpp_grad_tiles<DT, true, true>) : launch(gpp_grad_tiles<DT, true, false>))                   \
       : (hasu ? launch(gpp_grad_tiles<DT, false, true>) : launch(gpp_grad_tiles<DT, false, false>)))
  if (D <= 8) GPP_GT(8);
  else if (D <= 16) GPP_GT(16);
  else if (D <= 32) GPP_GT(32);
  else GPP_GT(64);
#undef GPP_GT
  hipLaunchKernelGGL(gpp_grad_finish, dim3(D + 1 + S, batch), dim3(256), 0, s, rec, nwg, D, S, wdiag, grp, N, g_w, g_sf2, g_tau,
                     ws_stride);
  if (dU > 0)
    hipLaunchKernelGGL(gpp_gU_finish, dim3((unsigned)((N * dU + 255) / 256), batch), dim3(256), 0, s, gUpart, N, dU, T, g_U,
                       ws_stride);
  return hipGetLastError();
}

hipError_t gpp_launch_predict_reduce(hipStream_t s, const double* Ksn, int64_t lds, const double* V, int64_t ldv,
                                     int64_t M, int64_t N, const double* alpha, const double* kss, double* mean_out,
                                     double* var_out) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(gpp_predict_rows, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, Ksn, lds, V, ldv, M, N, alpha, kss,
                     mean_out, var_out);
  return hipGetLastError();
}

// ---- out-of-place transpose (the mirror L^-T of the sharded evaluation's column blocks) --------------------------------
namespace {
__global__ __launch_bounds__(256) void gpp_transpose_tile(const double* __restrict__ src, int64_t lds, int64_t rows, int64_t cols,
                                                          double* __restrict__ dst, int64_t ldd) {
  __shared__ double tile[64][65];  // padded: the transposed read walks a column
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4 threads, 16 passes
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * lds + c] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;  // dst row = source column
    if (c < cols && r < rows) dst[c * ldd + r] = tile[tx][i];
  }
}
}  // namespace

hipError_t gpp_launch_transpose(hipStream_t s, const double* src, int64_t lds, int64_t rows, int64_t cols, double* dst, int64_t ldd) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  hipLaunchKernelGGL(gpp_transpose_tile, grid, dim3(256), 0, s, src, lds, rows, cols, dst, ldd);
  return hipGetLastError();
}
