// rp_linalg.hip — the linear solve inside one LM step (gfx950):
//   time_eliminate : damp + factor every 7x7 time block, Y = H_rt s_t L^-T
//   schur_syrk     : P = H_rr - Y Y^T on the fp64 matrix cores
//                    (v_mfma_f64_16x16x4_f64), split-K with fixed-order slab reduce
//   finalize       : S = s_r P s_r + D_r^2
//   cholesky_solve : blocked dense Cholesky of S (MFMA trailing update) + solves
//   backsub        : time blocks, candidate point, model cost change
// Together they replace Ceres' SchurEliminator + dense LLT behind
// options.linear_solver_type = DENSE_SCHUR (/root/reference/src/solving-structure.cpp:975)
// and LevenbergMarquardtStrategy::ComputeStep (Ceres, not in the reference tree).
#include "rp_internal.hpp"

namespace rp {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---------------------------------------------------------------- time elimination
// one thread per segment: scale, damp, factor, z
__global__ __launch_bounds__(64) void time_factor_kernel(ElimArgs e) {
  const int sgi = blockIdx.x * 64 + threadIdx.x;
  if (sgi >= e.n_seg) return;
  const int v7 = 7 * e.seg_var[sgi];
  const double* E = e.Ett + (int64_t)sgi * 49;
  double sc[7], A[28];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const double d = E[8 * k];
    if (e.compute_scale) e.scale[v7 + k] = e.use_scale ? 1.0 / (1.0 + sqrt(d)) : 1.0;
    sc[k] = e.scale[v7 + k];
    if (!e.reuse_diag) {
      double dd = sc[k] * sc[k] * d;
      dd = fmin(fmax(dd, e.min_diag), e.max_diag);
      e.diag[v7 + k] = dd;
    }
    e.gscaled[v7 + k] = sc[k] * e.gt[(int64_t)sgi * 7 + k];
  }
  // lower triangle of s E s + diag/radius, packed row-wise
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double a = sc[i] * sc[j] * E[7 * i + j];
      if (i == j) a += e.diag[v7 + i] / e.radius;
      A[i * (i + 1) / 2 + j] = a;
    }
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double d = A[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j * (j + 1) / 2 + k] * A[j * (j + 1) / 2 + k];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    d = sqrt(d);
    A[j * (j + 1) / 2 + j] = d;
    const double inv = 1.0 / d;
#pragma unroll
    for (int i = j + 1; i < 7; ++i) {
      double s = A[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[i * (i + 1) / 2 + k] * A[j * (j + 1) / 2 + k];
      A[i * (i + 1) / 2 + j] = s * inv;
    }
  }
  if (!ok) atomicExch(e.fail, 1);
  double* L = e.Lt + (int64_t)sgi * 49;
  double zz[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j < 7; ++j) L[7 * i + j] = (j <= i) ? A[i * (i + 1) / 2 + j] : 0.0;
    double s = e.gscaled[v7 + i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= A[i * (i + 1) / 2 + k] * zz[k];
    zz[i] = s / A[i * (i + 1) / 2 + i];
    e.z[(int64_t)sgi * 7 + i] = zz[i];
    e.Y[(int64_t)(7 * sgi + i) * e.ldn + e.n] = zz[i];  // augmented column
  }
}

// one thread per (segment, row): y^T L^T = (H_rt row) s_t  <=>  L y = s_t .* h
__global__ __launch_bounds__(256) void time_y_kernel(ElimArgs e, int row_blocks) {
  const int sgi = blockIdx.x / row_blocks;
  const int row = (blockIdx.x - sgi * row_blocks) * 256 + threadIdx.x;
  if (row >= e.n) return;
  const int v7 = 7 * e.seg_var[sgi];
  const double* L = e.Lt + (int64_t)sgi * 49;
  double y[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    double s = e.scale[v7 + i] * e.Hrt[(int64_t)(7 * sgi + i) * e.ldn + row];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[7 * i + k] * y[k];
    y[i] = s / L[7 * i + i];
    e.Y[(int64_t)(7 * sgi + i) * e.ldn + row] = y[i];
  }
}

void launch_time_eliminate(const ElimArgs& e, hipStream_t s) {
  if (e.n_seg <= 0) return;
  hipLaunchKernelGGL(time_factor_kernel, dim3((e.n_seg + 63) / 64), dim3(64), 0, s, e);
  if (e.n > 0) {
    const int row_blocks = (e.n + 255) / 256;
    hipLaunchKernelGGL(time_y_kernel, dim3((unsigned)row_blocks * (unsigned)e.n_seg), dim3(256), 0, s, e,
                       row_blocks);
  }
}

// ---------------------------------------------------------------- Schur SYRK (MFMA f64)
// C tile 128x128 per workgroup (4 waves, each 64x64 = 4x4 MFMA tiles of 16x16),
// K advanced 16 rows of Y at a time through LDS.  v_mfma_f64_16x16x4_f64 lane map:
//   A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
//   D reg r -> row (lane>>4) + 4r, col lane&15.
constexpr int SY_T = 128, SY_KC = 16, SY_LD = 144;  // LD % 32 == 16: the two k rows a 32-lane
                                                    // group reads fall on disjoint banks

int syrk_tile_pairs(int ldn) {
  const int nt = ldn / SY_T;
  return nt * (nt + 1) / 2;
}

__global__ __launch_bounds__(256) void syrk_kernel(SyrkArgs a, int n_pairs, int k_per_split) {
  __shared__ __attribute__((aligned(16))) double sA[SY_KC * SY_LD];
  __shared__ __attribute__((aligned(16))) double sB[SY_KC * SY_LD];
  const int pairi = blockIdx.x, split = blockIdx.y;
  // decode lower-triangular tile pair (ti >= tj)
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= pairi) ++ti;
  const int tj = pairi - ti * (ti + 1) / 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  const int k_begin = split * k_per_split;
  int k_end = k_begin + k_per_split;
  if (k_end > a.n_krows) k_end = a.n_krows;

  f64x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = (f64x4){0.0, 0.0, 0.0, 0.0};

  const int lrow = tid >> 6;        // 0..3
  const int lcol = 2 * (tid & 63);  // 0..126
  const bool diag_tile = (ti == tj);
  for (int k0 = k_begin; k0 < k_end; k0 += SY_KC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = lrow + 4 * i;
      const double2 va = *reinterpret_cast<const double2*>(a.Y + (int64_t)(k0 + kr) * a.ldn + ti * SY_T + lcol);
      *reinterpret_cast<double2*>(sA + kr * SY_LD + lcol) = va;
      if (!diag_tile) {
        const double2 vb = *reinterpret_cast<const double2*>(a.Y + (int64_t)(k0 + kr) * a.ldn + tj * SY_T + lcol);
        *reinterpret_cast<double2*>(sB + kr * SY_LD + lcol) = vb;
      }
    }
    __syncthreads();
    const double* pB = diag_tile ? sA : sB;
#pragma unroll
    for (int kk = 0; kk < SY_KC; kk += 4) {
      const int kr = kk + (lane >> 4);
      double av[4], bv[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        av[m] = sA[kr * SY_LD + wi + 16 * m + (lane & 15)];
        bv[m] = pB[kr * SY_LD + wj + 16 * m + (lane & 15)];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
          acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[nn], acc[m][nn], 0, 0, 0);
    }
    __syncthreads();
  }
  double* out = a.partial + ((int64_t)split * n_pairs + pairi) * (SY_T * SY_T);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wi + 16 * m + (lane >> 4) + 4 * r;
        const int col = wj + 16 * nn + (lane & 15);
        out[row * SY_T + col] = acc[m][nn][r];
      }
}

// P = Hrr - sum_split partial (lower tiles), mirrored to the upper triangle.
__global__ __launch_bounds__(256) void syrk_reduce_kernel(SyrkArgs a, int n_pairs) {
  const int pairi = blockIdx.x;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= pairi) ++ti;
  const int tj = pairi - ti * (ti + 1) / 2;
  for (int e = blockIdx.y * 256 + threadIdx.x; e < SY_T * SY_T; e += 256 * gridDim.y) {
    const int row = e / SY_T, col = e - row * SY_T;
    const int gi = ti * SY_T + row, gj = tj * SY_T + col;
    if (gi < gj) continue;  // diagonal tile: lower part only
    double s = 0.0;
    for (int sp = 0; sp < a.ksplit; ++sp) s += a.partial[((int64_t)sp * n_pairs + pairi) * (SY_T * SY_T) + e];
    const double v = a.Hrr[(int64_t)gi * a.ldn + gj] - s;
    a.P[(int64_t)gi * a.ldn + gj] = v;
    a.P[(int64_t)gj * a.ldn + gi] = v;
  }
}
__global__ __launch_bounds__(256) void schur_rhs_kernel(SyrkArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n) a.q[i] = a.gr[i] + a.P[(int64_t)a.n * a.ldn + i];
}

void launch_schur_syrk(const SyrkArgs& a, hipStream_t s) {
  const int n_pairs = syrk_tile_pairs(a.ldn);
  int k_per_split = (a.n_krows + a.ksplit - 1) / a.ksplit;
  k_per_split = (k_per_split + SY_KC - 1) / SY_KC * SY_KC;
  hipLaunchKernelGGL(syrk_kernel, dim3(n_pairs, a.ksplit), dim3(256), 0, s, a, n_pairs, k_per_split);
  hipLaunchKernelGGL(syrk_reduce_kernel, dim3(n_pairs, 8), dim3(256), 0, s, a, n_pairs);
  hipLaunchKernelGGL(schur_rhs_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------- reduced system
__global__ __launch_bounds__(256) void reduced_scale_kernel(ReducedArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int v = 7 * a.rvar[i / 7] + (i % 7);
  const double d = a.hdiag[i];
  if (a.compute_scale) a.scale[v] = a.use_scale ? 1.0 / (1.0 + sqrt(d)) : 1.0;
  const double sc = a.scale[v];
  if (!a.reuse_diag) a.diag[v] = fmin(fmax(sc * sc * d, a.min_diag), a.max_diag);
  a.gscaled[v] = sc * a.gr[i];
  a.rhs[i] = sc * a.q[i];
}
__global__ __launch_bounds__(256) void reduced_matrix_kernel(ReducedArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= a.n) return;
  const int vi = 7 * a.rvar[i / 7] + (i % 7), vj = 7 * a.rvar[j / 7] + (j % 7);
  double v = a.scale[vi] * a.scale[vj] * a.P[(int64_t)i * a.ldn + j];
  if (i == j) v += a.diag[vi] / a.radius;
  a.S[(int64_t)i * a.ldn + j] = v;
}
void launch_finalize_reduced(const ReducedArgs& a, hipStream_t s) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(reduced_scale_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(reduced_matrix_kernel, dim3((a.n + 255) / 256, a.n), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------- dense Cholesky
constexpr int CH_NB = 32;

// factor the diagonal block [j0, j0+nb) and store its inverse (lower) in invL
__global__ __launch_bounds__(64) void chol_diag_kernel(int n, int ldn, double* S, double* invL, int j0, int* fail) {
  __shared__ double A[CH_NB][CH_NB + 1];
  __shared__ double Vi[CH_NB][CH_NB + 1];
  const int t = threadIdx.x;
  const int nb = (n - j0 < CH_NB) ? (n - j0) : CH_NB;
  for (int e = t; e < CH_NB * CH_NB; e += 64) {
    const int i = e / CH_NB, j = e % CH_NB;
    A[i][j] = (i < nb && j < nb) ? S[(int64_t)(j0 + i) * ldn + j0 + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int j = 0; j < CH_NB; ++j) {
    if (t == 0) {
      const double d = A[j][j];
      if (!(d > 0.0) || !isfinite(d)) atomicExch(fail, 1);
      A[j][j] = sqrt(d);
    }
    __syncthreads();
    const double dj = A[j][j];
    if (t > j && t < CH_NB) A[t][j] /= dj;
    __syncthreads();
    // trailing update of the lower triangle: thread t owns row t (t < 32) / helps with columns
    for (int e = t; e < CH_NB * CH_NB; e += 64) {
      const int i = e / CH_NB, k = e % CH_NB;
      if (k > j && i >= k) A[i][k] -= A[i][j] * A[k][j];
    }
    __syncthreads();
  }
  // inverse of the lower-triangular block: column c by thread c
  if (t < CH_NB) {
    const int c = t;
    for (int i = 0; i < CH_NB; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= A[i][k] * Vi[k][c];
      Vi[i][c] = (i >= c) ? s / A[i][i] : 0.0;
    }
  }
  __syncthreads();
  double* inv = invL + (int64_t)(j0 / CH_NB) * CH_NB * CH_NB;
  for (int e = t; e < CH_NB * CH_NB; e += 64) {
    const int i = e / CH_NB, j = e % CH_NB;
    inv[e] = Vi[i][j];
    if (i < nb && j < nb) S[(int64_t)(j0 + i) * ldn + j0 + j] = (j <= i) ? A[i][j] : 0.0;
  }
}

// L21 = A21 * inv(L11)^T for rows below the diagonal block; one row per thread
__global__ __launch_bounds__(64) void chol_panel_kernel(int n, int ldn, double* S, const double* invL, int j0) {
  __shared__ double Vi[CH_NB][CH_NB + 1];
  const double* inv = invL + (int64_t)(j0 / CH_NB) * CH_NB * CH_NB;
  for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 64) Vi[e / CH_NB][e % CH_NB] = inv[e];
  __syncthreads();
  const int row = j0 + CH_NB + blockIdx.x * 64 + threadIdx.x;
  if (row >= n) return;
  double a[CH_NB];
  double* p = S + (int64_t)row * ldn + j0;
#pragma unroll
  for (int k = 0; k < CH_NB; ++k) a[k] = p[k];
#pragma unroll
  for (int c = 0; c < CH_NB; ++c) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k <= c; ++k) s += a[k] * Vi[c][k];  // inv(L)^T[k][c] = inv(L)[c][k]
    p[c] = s;
  }
}

// trailing update A22 -= L21 L21^T on 64x64 tiles (lower tiles only), MFMA f64.
// 4 waves: wave w owns the 32x32 quadrant (w>>1, w&1) = 2x2 MFMA tiles; K = 32.
__global__ __launch_bounds__(256) void chol_update_kernel(int n, int ldn, double* S, int j0) {
  __shared__ double sA[CH_NB][64 + 16];  // [k][row]  (stride 80: 80 % 32 == 16)
  __shared__ double sB[CH_NB][64 + 16];
  const int base = j0 + CH_NB;
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (tj > ti) return;
  const int r0 = base + ti * 64, c0 = base + tj * 64;
  if (r0 >= n) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int e = tid; e < 64 * CH_NB; e += 256) {
    const int row = e / CH_NB, k = e % CH_NB;
    sA[k][row] = (r0 + row < n) ? S[(int64_t)(r0 + row) * ldn + j0 + k] : 0.0;
    sB[k][row] = (c0 + row < n) ? S[(int64_t)(c0 + row) * ldn + j0 + k] : 0.0;
  }
  __syncthreads();
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  f64x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) acc[m][nn] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < CH_NB; kk += 4) {
    const int kr = kk + (lane >> 4);
    double av[2], bv[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      av[m] = sA[kr][wi + 16 * m + (lane & 15)];
      bv[m] = sB[kr][wj + 16 * m + (lane & 15)];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
        acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[nn], acc[m][nn], 0, 0, 0);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = r0 + wi + 16 * m + (lane >> 4) + 4 * r;
        const int gj = c0 + wj + 16 * nn + (lane & 15);
        if (gi < n && gj <= gi) S[(int64_t)gi * ldn + gj] -= acc[m][nn][r];
      }
}

// forward and backward substitution with the stored block inverses; single workgroup.
__global__ __launch_bounds__(1024) void chol_solve_kernel(int n, int ldn, const double* S, const double* invL,
                                                          double* b) {
  __shared__ double xs[CH_NB];
  __shared__ double red[32][CH_NB + 1];
  const int t = threadIdx.x;
  const int nblk = (n + CH_NB - 1) / CH_NB;
  // ---- L w = b
  for (int jb = 0; jb < nblk; ++jb) {
    const int j0 = jb * CH_NB;
    const int nb = (n - j0 < CH_NB) ? (n - j0) : CH_NB;
    const double* inv = invL + (int64_t)jb * CH_NB * CH_NB;
    if (t < CH_NB) {
      double s = 0.0;
      if (t < nb)
        for (int k = 0; k <= t; ++k) s += inv[t * CH_NB + k] * b[j0 + k];
      xs[t] = s;
    }
    __syncthreads();
    if (t < nb) b[j0 + t] = xs[t];
    for (int row = j0 + CH_NB + t; row < n; row += 1024) {
      const double* p = S + (int64_t)row * ldn + j0;
      double s = 0.0;
#pragma unroll 8
      for (int k = 0; k < CH_NB; ++k) s += p[k] * xs[k];
      b[row] -= s;
    }
    __syncthreads();
  }
  // ---- L^T y = w
  for (int jb = nblk - 1; jb >= 0; --jb) {
    const int j0 = jb * CH_NB;
    const int nb = (n - j0 < CH_NB) ? (n - j0) : CH_NB;
    const double* inv = invL + (int64_t)jb * CH_NB * CH_NB;
    // partial[col] = sum_{rows below the block} L[row][j0+col] * y[row]
    const int col = t & 31, chunk = t >> 5;
    double s = 0.0;
    for (int row = j0 + CH_NB + chunk; row < n; row += 32) s += S[(int64_t)row * ldn + j0 + col] * b[row];
    red[chunk][col] = s;
    __syncthreads();
    if (t < CH_NB) {
      double tot = 0.0;
      for (int c = 0; c < 32; ++c) tot += red[c][t];
      xs[t] = (t < nb) ? b[j0 + t] - tot : 0.0;
    }
    __syncthreads();
    if (t < nb) {
      double y = 0.0;
      for (int k = t; k < nb; ++k) y += inv[k * CH_NB + t] * xs[k];  // inv(L)^T
      b[j0 + t] = y;
    }
    __syncthreads();
  }
}

void launch_cholesky_solve(int n, int ldn, double* S, double* invL, double* rhs, int* fail, hipStream_t s) {
  if (n <= 0) return;
  for (int j0 = 0; j0 < n; j0 += CH_NB) {
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, s, n, ldn, S, invL, j0, fail);
    const int below = n - (j0 + CH_NB);
    if (below > 0) {
      hipLaunchKernelGGL(chol_panel_kernel, dim3((below + 63) / 64), dim3(64), 0, s, n, ldn, S, invL, j0);
      const int nt = (below + 63) / 64;
      hipLaunchKernelGGL(chol_update_kernel, dim3(nt, nt), dim3(256), 0, s, n, ldn, S, j0);
    }
  }
  hipLaunchKernelGGL(chol_solve_kernel, dim3(1), dim3(1024), 0, s, n, ldn, S, invL, rhs);
}

// ---------------------------------------------------------------- back-substitution
// reduced part: w = s_r y_r, delta = -w, candidate, partial scalars (single block)
__global__ __launch_bounds__(256) void backsub_reduced_kernel(BacksubArgs a) {
  __shared__ double sh[3][4];
  double m = 0.0, d2 = 0.0, x2 = 0.0;
  for (int i = threadIdx.x; i < a.ldn; i += 256) {
    if (i < a.n) {
      const int v = 7 * a.rvar[i / 7] + (i % 7);
      const double y = a.yr[i];
      const double w = a.scale[v] * y;
      a.w[i] = w;
      a.delta[v] = -w;
      a.cand[v] = a.x[v] - w;
      m += 0.5 * y * a.gscaled[v] + 0.5 * (a.diag[v] / a.radius) * y * y;
      d2 += w * w;
      x2 += a.x[v] * a.x[v];
    } else {
      a.w[i] = 0.0;
    }
  }
  m = wsum(m); d2 = wsum(d2); x2 = wsum(x2);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = m; sh[1][threadIdx.x >> 6] = d2; sh[2][threadIdx.x >> 6] = x2;
  }
  __syncthreads();
  if (threadIdx.x < 3) a.scalars[threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// one workgroup (7 waves) per segment: v_i = z_i - <Y row 7s+i, w>, then L^T y = v
__global__ __launch_bounds__(448) void backsub_time_kernel(BacksubArgs a) {
  __shared__ double vv[7];
  const int sgi = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* yrow = a.Y + (int64_t)(7 * sgi + wave) * a.ldn;
  double s = 0.0;
  for (int i = lane; i < a.n; i += 64) s += yrow[i] * a.w[i];
  s = wsum(s);
  if (lane == 0) vv[wave] = a.z[(int64_t)sgi * 7 + wave] - s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double* L = a.Lt + (int64_t)sgi * 49;
    const int v7 = 7 * a.seg_var[sgi];
    double y[7];
    double m = 0.0, d2 = 0.0, x2 = 0.0;
    for (int i = 6; i >= 0; --i) {
      double t = vv[i];
      for (int k = i + 1; k < 7; ++k) t -= L[7 * k + i] * y[k];
      y[i] = t / L[7 * i + i];
    }
    for (int i = 0; i < 7; ++i) {
      const double w = a.scale[v7 + i] * y[i];
      a.delta[v7 + i] = -w;
      a.cand[v7 + i] = a.x[v7 + i] - w;
      m += 0.5 * y[i] * a.gscaled[v7 + i] + 0.5 * (a.diag[v7 + i] / a.radius) * y[i] * y[i];
      d2 += w * w;
      x2 += a.x[v7 + i] * a.x[v7 + i];
    }
    a.seg_partial[(int64_t)sgi * 4 + 0] = m;
    a.seg_partial[(int64_t)sgi * 4 + 1] = d2;
    a.seg_partial[(int64_t)sgi * 4 + 2] = x2;
  }
}
// scalars[3+k] = sum over segments (fixed order)
__global__ __launch_bounds__(256) void backsub_sum_kernel(BacksubArgs a) {
  __shared__ double sh[3][4];
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < a.n_seg; i += 256)
    for (int k = 0; k < 3; ++k) acc[k] += a.seg_partial[(int64_t)i * 4 + k];
  for (int k = 0; k < 3; ++k) {
    const double v = wsum(acc[k]);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) a.scalars[3 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

void launch_backsub(const BacksubArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(backsub_reduced_kernel, dim3(1), dim3(256), 0, s, a);
  if (a.n_seg > 0) {
    hipLaunchKernelGGL(backsub_time_kernel, dim3(a.n_seg), dim3(448), 0, s, a);
    hipLaunchKernelGGL(backsub_sum_kernel, dim3(1), dim3(256), 0, s, a);
  }
}

}  // namespace rp
