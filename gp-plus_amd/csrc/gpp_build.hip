// gpp_build.hip — fused covariance-tile kernels.
//
// Reference call sites replaced:
//   models/gp_plus.py:472-474   covar_module(x_new).evaluate()   -> gpytorch ScaleKernel(ProductKernel(RBF,RBF)):
//                               covar_dist skinny GEMM, clamp, div, exp, elementwise product, outputscale multiply
//   likelihoods_noise/multifidelity.py:63-67 (and gpytorch GaussianLikelihood)   K + diag(noise[fidelity_i])
//   kernels/Rough_RBF.py:27-32  exp(-|| sqrt(l) (x1-x2) ||^2)
// All of these are  Ky[i,j] = sf2 * exp(-sum_d w_d (u_id-u_jd)^2) (+ Matern factor) + (i==j)(tau[grp_i]+jitter):
// ONE pass that writes each output element exactly once (HBM-write bound: 8 B per element, 4 N^2 B for the lower
// triangle), instead of the reference's ~6 N^2-sized fp64 passes.
//
// Tile: 64x64 outputs per 256-thread work-group, 4x4 per thread.  The two 64-row slabs of U are staged in LDS
// pre-multiplied by sqrt(w_d), stored [d][row].  A thread's 4 rows are one 32-byte LDS read (a broadcast: 16 lanes share it); its 4
// COLUMNS are two pairs, {2 tx, 2 tx + 1} and {32 + 2 tx, 32 + 2 tx + 1}: each ds_read_b128 of 16 lanes then covers 256
// contiguous bytes = every bank once.  (Round 3 read 4 consecutive columns per thread: lanes k and k + 8 of a 16-lane group hit
// the same banks, and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.31, profiles/r03_sq_counters.txt.)  Each thread stores two
// 16-byte pieces per row: 16 lanes x 16 B = 256 B contiguous per piece.
#include "gpp_internal.h"

#include <atomic>

namespace {

constexpr int TB = 64;
constexpr int DMAX = 64;

__device__ __forceinline__ double kfun(double r2_rbf, double r2_mat, int kind, const GppExpConsts& ec) {
  double v = gpp_exp_nonpos(-r2_rbf, ec);
  if (kind == 1) {  // Matern 3/2 in the scaled distance r = sqrt(2 * r2_mat)  (gpytorch MaternKernel nu=1.5)
    const double r = sqrt(3.0 * 2.0 * r2_mat);
    v *= (1.0 + r) * gpp_exp_nonpos(-r, ec);
  } else if (kind == 2) {  // Matern 5/2
    const double r = sqrt(5.0 * 2.0 * r2_mat);
    v *= (1.0 + r + r * r * (1.0 / 3.0)) * gpp_exp_nonpos(-r, ec);
  }
  return v;
}

// grid.x = tile id.  lower = 1 / 2: only the tiles of the lower / upper triangle of a square problem are written.
// MAT = false: pure RBF product (kind 0) — no second set of distance accumulators (32 VGPRs: four instead of three waves per
// SIMD behind the loads and the exp chains) and no Matern code.
template <bool MAT>
__global__ __launch_bounds__(256) void gpp_cov_tile(const double* __restrict__ Ua, int64_t Ma, const double* __restrict__ Ub,
                                                    int64_t Nb, int D, const double* __restrict__ w,
                                                    const double* __restrict__ sf2p, const double* __restrict__ tau,
                                                    const int32_t* __restrict__ grp, double jitter, int kind, int d_split,
                                                    int lower, int add_diag, double* __restrict__ K, int64_t ld,
                                                    int64_t row0, int tiles_n, int64_t tile_row0, int64_t sU, int64_t sK,
                                                    int S) {
  // the two feature slabs, D x 64 doubles each, sized by the launch (dynamic LDS): with the static DMAX-sized arrays (64 KiB)
  // only two work-groups fitted a CU and the exp chains had two waves per SIMD to hide behind
  extern __shared__ __attribute__((aligned(16))) double cov_smem[];
  double* sa = cov_smem;
  double* sb = cov_smem + (size_t)D * TB;
  {  // batch element blockIdx.y: its own features (sU = 0: shared), weights, scale, noise levels and output matrix
    const int64_t b = blockIdx.y;
    Ua += b * sU;
    Ub += b * sU;
    w += b * D;
    sf2p += b;
    if (tau) tau += b * S;
    K += b * sK;
  }
  int64_t ti, tj;
  {
    const int64_t t = blockIdx.x;
    if (lower) {
      // rows start at tile_row0 (in tiles); row r has r+1 tiles.  offset(r) = r(r+1)/2 - tile_row0(tile_row0+1)/2
      const double base = 0.5 * (double)tile_row0 * (double)(tile_row0 + 1);
      int64_t r = (int64_t)((sqrt(8.0 * ((double)t + base) + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 - (int64_t)base <= t) ++r;
      while (r * (r + 1) / 2 - (int64_t)base > t) --r;
      ti = r;
      tj = t - (r * (r + 1) / 2 - (int64_t)base);
      if (lower == 2) {  // upper triangle: mirrored tile of the same enumeration
        const int64_t q = ti;
        ti = tj;
        tj = q;
      }
    } else {
      ti = tile_row0 + t / tiles_n;
      tj = t % tiles_n;
    }
  }
  const int tid = threadIdx.x;
  const int64_t i0 = ti * TB, j0 = tj * TB;

  for (int e = tid; e < D * TB; e += 256) {
    const int d = e / TB, r = e - d * TB;
    const double sw = sqrt(w[d]);
    sa[d * TB + r] = (i0 + r < Ma) ? Ua[(i0 + r) * D + d] * sw : 0.0;
    sb[d * TB + r] = (j0 + r < Nb) ? Ub[(j0 + r) * D + d] * sw : 0.0;
  }
  __syncthreads();

  const int ty = tid >> 4, tx = tid & 15;
  double r2a[4][4], r2b[MAT ? 4 : 1][MAT ? 4 : 1];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      r2a[a][b] = 0.0;
      if (MAT) r2b[a][b] = 0.0;
    }
  const int dsp = (!MAT || kind == 0) ? D : d_split;
  for (int d = 0; d < D; ++d) {
    // the thread's 4 rows (one 32-byte broadcast read) and 2 + 2 columns (two conflict-free ds_read_b128) of feature d
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d a01 = reinterpret_cast<const v2d*>(sa + d * TB + 4 * ty)[0], a23 = reinterpret_cast<const v2d*>(sa + d * TB + 4 * ty)[1];
    const v2d b01 = *reinterpret_cast<const v2d*>(sb + d * TB + 2 * tx), b23 = *reinterpret_cast<const v2d*>(sb + d * TB + 32 + 2 * tx);
    const double ua[4] = {a01.x, a01.y, a23.x, a23.y}, ub[4] = {b01.x, b01.y, b23.x, b23.y};
    if (!MAT || d < dsp) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = ua[a] - ub[b];
          r2a[a][b] = fma(df, df, r2a[a][b]);
        }
    } else if constexpr (MAT) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = ua[a] - ub[b];
          r2b[a][b] = fma(df, df, r2b[a][b]);
        }
    }
  }
  const double sf2 = *sf2p;
  const GppExpConsts ec = gpp_exp_consts();
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = i0 + 4 * ty + a;
    if (i >= Ma || i < row0) continue;
    double v[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t j = j0 + 2 * tx + (b & 1) + ((b >> 1) << 5);
      double x = sf2 * (MAT ? kfun(r2a[a][b], r2b[MAT ? a : 0][MAT ? b : 0], kind, ec) : gpp_exp_nonpos(-r2a[a][b], ec));
      if (add_diag && i == j) x += (tau ? tau[grp ? grp[i] : 0] : 0.0) + jitter;
      v[b] = x;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // the two column pairs of this thread
      double* out = K + i * ld + j0 + 2 * tx + 32 * h;
      const int64_t jrem = Nb - (j0 + 2 * tx + 32 * h);
      if (jrem >= 2 && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d p0 = {v[2 * h], v[2 * h + 1]};
        *reinterpret_cast<v2d*>(out) = p0;
      } else {
        if (jrem >= 1) out[0] = v[2 * h];
        if (jrem >= 2) out[1] = v[2 * h + 1];
      }
    }
  }
}

}  // namespace

// more than 48 KiB of dynamic LDS (D > 48 features) needs the opt-in, once per device
static hipError_t cov_lds_optin(int D) {
  if ((size_t)2 * D * TB * sizeof(double) <= 48 * 1024) return hipSuccess;
  static std::atomic<bool> done[64];  // (per device; handles of different host threads may get here together)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gpp_cov_tile<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          2 * DMAX * TB * (int)sizeof(double));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gpp_cov_tile<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * DMAX * TB * (int)sizeof(double));
  if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
  return e;
}

hipError_t gpp_launch_kernel_build(hipStream_t s, const double* U, int64_t N, int D, const double* w, const double* sf2,
                                   const double* tau, const int32_t* grp, int S, double jitter, int kind, int d_split,
                                   int uplo, double* Ky, int64_t ld, int64_t row0, int64_t nrows, int batch, int64_t sU,
                                   int64_t sK) {
  if (N <= 0 || nrows <= 0 || batch <= 0) return hipSuccess;
  if (D > DMAX) return hipErrorInvalidValue;
  if (hipError_t e = cov_lds_optin(D); e != hipSuccess) return e;
  const int64_t tr0 = row0 / TB;
  const int64_t tr1 = (row0 + nrows + TB - 1) / TB;  // exclusive
  const int tiles_n = (int)((N + TB - 1) / TB);
  int64_t nt;
  if (uplo) nt = tr1 * (tr1 + 1) / 2 - tr0 * (tr0 + 1) / 2;
  else nt = (tr1 - tr0) * tiles_n;
  // rows of the last tile row beyond row0+nrows are cut by passing Ma = row0+nrows
  // (same box, A/B twice at N = 20000: 0.496 / 0.540 ms with the RBF instantiation against 0.575 / 0.594 with the general one)
  auto* fn = kind == 0 ? gpp_cov_tile<false> : gpp_cov_tile<true>;
  hipLaunchKernelGGL(fn, dim3((unsigned)nt, (unsigned)batch), dim3(256), (size_t)2 * D * TB * sizeof(double), s, U, row0 + nrows, U, N, D, w, sf2, tau,
                     grp, jitter, kind, d_split, uplo, 1, Ky, ld, row0, tiles_n, tr0, sU, sK, S);
  return hipGetLastError();
}

hipError_t gpp_launch_cross_kernel(hipStream_t s, const double* Ua, int64_t Ma, const double* Ub, int64_t Nb, int D,
                                   const double* w, const double* sf2, int kind, int d_split, double* Kab, int64_t ld) {
  if (Ma <= 0 || Nb <= 0) return hipSuccess;
  if (D > DMAX) return hipErrorInvalidValue;
  if (hipError_t e = cov_lds_optin(D); e != hipSuccess) return e;
  const int tiles_n = (int)((Nb + TB - 1) / TB);
  const int64_t tiles_m = (Ma + TB - 1) / TB;
  auto* fn = kind == 0 ? gpp_cov_tile<false> : gpp_cov_tile<true>;
  hipLaunchKernelGGL(fn, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), (size_t)2 * D * TB * sizeof(double), s, Ua, Ma, Ub, Nb, D, w, sf2,
                     (const double*)nullptr, (const int32_t*)nullptr, 0.0, kind, d_split, 0, 0, Kab, ld, (int64_t)0,
                     tiles_n, (int64_t)0, (int64_t)0, (int64_t)0, 0);
  return hipGetLastError();
}
