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
#include <cstdlib>
#include <utility>
#include "rp_internal.hpp"
#include "rp_model.hpp"

namespace rp {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---------------------------------------------------------------- time elimination
// one thread per segment: scale, damp, factor, z; with rotation-vector records also the thin QR of
// the six-row form (the same thread went through a launch of its own until round 3)
__global__ __launch_bounds__(64) void time_factor_kernel(ElimArgs e) {
  const int sgi = blockIdx.x * 64 + threadIdx.x;
  if (sgi >= e.n_seg) return;
  const int v7 = 7 * e.seg_var[sgi];
  const double* E = e.Ett + (int64_t)sgi * 49;
  double sc[7], A[28];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    // (a diagonal entry of J^T J; the rotation-vector records give it as a sandwich T^T S T, which
    // rounds to a few -1e-15 where a column of J is exactly zero: a corner at the board's origin
    // does not move with the block's rotation)
    const double d = fmax(E[8 * k], 0.0);
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
    if (!e.rows6) e.Y[(int64_t)(7 * sgi + i) * e.ldn + e.n] = zz[i];  // augmented column (six-row form: z6 below)
  }
  if (!e.rows6) return;
  // ---- Six-row form (rotation-vector records): W = L^-1 diag(s_t) T_t^T (7 x 6) = Q R (thin QR:
  // W^T W = R^T R, R upper, row-major), z6 = Q^T z.  Q | R | z6 go to e.segw (unit_y reads R, the
  // back-substitution Q and z6) and z6 into the augmented column of the segment's six rows of Y.
  double Tt[42], W[42];
  block_map(e.x, e.seg_var[sgi], true, Tt);
#pragma unroll
  for (int m = 0; m < 6; ++m)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      double t = sc[i] * Tt[7 * m + i];
#pragma unroll
      for (int k = 0; k < i; ++k) t -= A[i * (i + 1) / 2 + k] * W[7 * m + k];
      W[7 * m + i] = t / A[i * (i + 1) / 2 + i];
    }
  // W = Q R by modified Gram-Schmidt (columns of W are [7 m + i]): R to the accuracy of cond(W),
  // where the Cholesky factor of W^T W would have cond(W)^2 against it
  double R[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) R[k] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double nn = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) nn += W[7 * j + i] * W[7 * j + i];
    const double r = sqrt(nn), inv = 1.0 / r;
    R[6 * j + j] = r;
#pragma unroll
    for (int i = 0; i < 7; ++i) W[7 * j + i] *= inv;
#pragma unroll
    for (int k = j + 1; k < 6; ++k) {
      double d = 0.0;
#pragma unroll
      for (int i = 0; i < 7; ++i) d += W[7 * j + i] * W[7 * k + i];
      R[6 * j + k] = d;
#pragma unroll
      for (int i = 0; i < 7; ++i) W[7 * k + i] -= d * W[7 * j + i];
    }
  }
  double* out = e.segw + (int64_t)sgi * SEGW;
#pragma unroll
  for (int k = 0; k < 42; ++k) out[k] = W[k];  // Q
#pragma unroll
  for (int k = 0; k < 36; ++k) out[SEGW_R + k] = R[k];
  // z6 = R^-T W^T z = Q^T z
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) t += W[7 * a + i] * zz[i];
    out[SEGW_Z + a] = t;
    e.Y[(int64_t)(6 * sgi + a) * e.ldn + e.n] = t;  // augmented column
  }
}

void launch_time_eliminate(const ElimArgs& e, const GatherArgs& g, hipStream_t s) {
  if (e.n_seg <= 0) return;
  hipLaunchKernelGGL(time_factor_kernel, dim3((e.n_seg + 63) / 64), dim3(64), 0, s, e);
  if (e.n > 0) launch_unit_y(g, e, s);
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

// Work split inside a 128 x 128 tile = 8 x 8 MFMA tiles, of which only those that hold an entry
// (row <= n, column <= row) are computed, spread evenly over the four waves:
//   off-diagonal tile  wave w: all active tile rows x tile columns {2w, 2w + 1}     (<= 16 MFMA tiles)
//   diagonal tile      wave 0: rows 0..3 x columns <= row (10), waves 1 / 2: rows 4..7 x columns
//                      {0, 1} / {2, 3} (8 each), wave 3: rows 4..7 x columns 4..row (10)
// (a fixed 64 x 64 quadrant per wave left the upper quadrant of the 7 diagonal tiles and the
// padding rows of the last tile row idle: 448 instead of 342 tile products on the critical path
// of config D's 7 x 7 tiles.)  active rows: ra = tile rows that reach index n (the augmented row).
template <bool DIAG>
__device__ __forceinline__ void syrk_tile(const SyrkArgs& a, int n_pairs, int k_per_split, int pairi, int split, int ti, int tj,
                                          double* sA, double* sB) {
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int k_begin = split * k_per_split;
  int k_end = k_begin + k_per_split;
  if (k_end > a.n_krows) k_end = a.n_krows;
  int ra = (a.n + 1 - ti * SY_T + 15) / 16;  // active MFMA tile rows of this tile
  ra = ra > 8 ? 8 : ra;
  // this wave's block of MFMA tiles: rows rbase + i, columns cbase + j (j < ncol)
  const int rbase = DIAG ? (wave == 0 ? 0 : 4) : 0;
  const int cbase = DIAG ? (wave == 0 ? 0 : wave == 1 ? 0 : wave == 2 ? 2 : 4) : 2 * wave;
  const int ncol = DIAG ? ((wave == 1 || wave == 2) ? 2 : 4) : 2;

  f64x4 acc[16];  // DIAG: [i][j] at 4 i + j (4 x 4); else [m][j] at 2 m + j (8 x 2)
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[m] = (f64x4){0.0, 0.0, 0.0, 0.0};

  const int lrow = tid >> 6;        // 0..3
  const int lcol = 2 * (tid & 63);  // 0..126
  // the next 16 rows of Y travel to registers while the MFMAs of the current 16 run out of LDS
  double2 va0, va1, va2, va3, vb0, vb1, vb2, vb3;
  const double* pa = a.Y + (int64_t)lrow * a.ldn + ti * SY_T + lcol;
  const double* pb = a.Y + (int64_t)lrow * a.ldn + tj * SY_T + lcol;
  const int64_t r4 = 4 * (int64_t)a.ldn;
#define SY_FETCH(K0)                                                                  \
  do {                                                                                \
    const int64_t o_ = (int64_t)(K0) * a.ldn;                                         \
    va0 = *reinterpret_cast<const double2*>(pa + o_);                                 \
    va1 = *reinterpret_cast<const double2*>(pa + o_ + r4);                            \
    va2 = *reinterpret_cast<const double2*>(pa + o_ + 2 * r4);                        \
    va3 = *reinterpret_cast<const double2*>(pa + o_ + 3 * r4);                        \
    if (!DIAG) {                                                                      \
      vb0 = *reinterpret_cast<const double2*>(pb + o_);                               \
      vb1 = *reinterpret_cast<const double2*>(pb + o_ + r4);                          \
      vb2 = *reinterpret_cast<const double2*>(pb + o_ + 2 * r4);                      \
      vb3 = *reinterpret_cast<const double2*>(pb + o_ + 3 * r4);                      \
    }                                                                                 \
  } while (0)
  if (k_begin < k_end) SY_FETCH(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += SY_KC) {
    double* wa = sA + lrow * SY_LD + lcol;
    *reinterpret_cast<double2*>(wa) = va0;
    *reinterpret_cast<double2*>(wa + 4 * SY_LD) = va1;
    *reinterpret_cast<double2*>(wa + 8 * SY_LD) = va2;
    *reinterpret_cast<double2*>(wa + 12 * SY_LD) = va3;
    if (!DIAG) {
      double* wb = sB + lrow * SY_LD + lcol;
      *reinterpret_cast<double2*>(wb) = vb0;
      *reinterpret_cast<double2*>(wb + 4 * SY_LD) = vb1;
      *reinterpret_cast<double2*>(wb + 8 * SY_LD) = vb2;
      *reinterpret_cast<double2*>(wb + 12 * SY_LD) = vb3;
    }
    __syncthreads();
    SY_FETCH(k0 + SY_KC < k_end ? k0 + SY_KC : k0);  // the last step fetches its own rows again (unused)
    const double* pB = DIAG ? sA : sB;
#pragma unroll
    for (int kk = 0; kk < SY_KC; kk += 4) {
      const int kr = kk + (lane >> 4);
      if (DIAG) {
        double av[4], bv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          av[m] = sA[kr * SY_LD + 16 * (rbase + m) + (lane & 15)];
          bv[m] = pB[kr * SY_LD + 16 * (cbase + (m < ncol ? m : 0)) + (lane & 15)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < ncol && cbase + j <= rbase + i && rbase + i < ra)  // wave-uniform
              acc[4 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[4 * i + j], 0, 0, 0);
      } else {
        double av[8], bv[2];
#pragma unroll
        for (int m = 0; m < 8; ++m) av[m] = sA[kr * SY_LD + 16 * m + (lane & 15)];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = pB[kr * SY_LD + 16 * (cbase + j) + (lane & 15)];
#pragma unroll
        for (int m = 0; m < 8; ++m)
          if (m < ra) {  // wave-uniform
            acc[2 * m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[0], acc[2 * m], 0, 0, 0);
            acc[2 * m + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[1], acc[2 * m + 1], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
#undef SY_FETCH
  double* out = a.partial + ((int64_t)split * n_pairs + pairi) * (SY_T * SY_T);
  if (DIAG) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < ncol && cbase + j <= rbase + i && rbase + i < ra)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[(16 * (rbase + i) + (lane >> 4) + 4 * r) * SY_T + 16 * (cbase + j) + (lane & 15)] = acc[4 * i + j][r];
  } else {
#pragma unroll
    for (int m = 0; m < 8; ++m)
      if (m < ra)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[(16 * m + (lane >> 4) + 4 * r) * SY_T + 16 * (cbase + j) + (lane & 15)] = acc[2 * m + j][r];
  }
}

// One-dimensional grid of n_pairs * ksplit workgroups.  Consecutive workgroup ids go round-robin
// over the 8 XCDs, each with its own L2, and the tile pairs of one K slab all read the same rows
// of Y (every element serves 8 tiles): with ksplit a multiple of 8, slab s runs on XCD s % 8 with
// all its pairs next to each other in that XCD's dispatch order, so Y comes from HBM about once
// instead of once per XCD (3.2 GB fetched for a 0.5 GB Y before).  Within a slab the
// off-diagonal pairs come first (they take longest), then the diagonal ones.
__global__ __launch_bounds__(256, 2) void syrk_kernel(SyrkArgs a, int n_pairs, int k_per_split) {
  __shared__ __attribute__((aligned(16))) double sA[SY_KC * SY_LD];
  __shared__ __attribute__((aligned(16))) double sB[SY_KC * SY_LD];
  const int nt = a.ldn / SY_T, n_off = nt * (nt - 1) / 2;
  int split, pr;
  if ((a.ksplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    split = (q / n_pairs) * 8 + xcd;
    pr = q % n_pairs;
  } else {
    split = blockIdx.x / n_pairs;
    pr = blockIdx.x % n_pairs;
  }
  if (pr < n_off) {
    // off-diagonal pair number pr: ti > tj
    int ti = 1;
    while (ti * (ti + 1) / 2 <= pr) ++ti;
    const int tj = pr - ti * (ti - 1) / 2;
    syrk_tile<false>(a, n_pairs, k_per_split, ti * (ti + 1) / 2 + tj, split, ti, tj, sA, sB);
  } else {
    const int ti = pr - n_off;
    syrk_tile<true>(a, n_pairs, k_per_split, ti * (ti + 1) / 2 + ti, split, ti, ti, sA, sB);
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
    if (gi < gj || gi > a.n) continue;  // lower part only; rows beyond n are padding (P stays 0 there)
    double s = 0.0;
    const double* src = a.partial + (int64_t)pairi * (SY_T * SY_T) + e;
    const int64_t sstride = (int64_t)n_pairs * (SY_T * SY_T);
    int sp = 0;
    for (; sp + 8 <= a.ksplit; sp += 8) {  // eight slabs in flight, added in slab order
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[(sp + k) * sstride];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; sp < a.ksplit; ++sp) s += src[sp * sstride];
    const double v = (a.Hrr ? a.Hrr[(int64_t)gi * a.ldn + gj] : 0.0) - s;
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
  hipLaunchKernelGGL(syrk_kernel, dim3(n_pairs * a.ksplit), dim3(256), 0, s, a, n_pairs, k_per_split);
  hipLaunchKernelGGL(syrk_reduce_kernel, dim3(n_pairs, 32), dim3(256), 0, s, a, n_pairs);
  hipLaunchKernelGGL(schur_rhs_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}

// P = Hrr + T^T Su T for the block pair (blockIdx.x, blockIdx.y), thread = (a, b) of the 7 x 7
// block; T maps a block's six columns [w (3) | translation (3)] to its seven parameters:
// quaternion column a < 4 = sum_i E[i][a] w_i, translation column as is.  Row blockIdx.y ==
// n_rblocks is the augmented row (right-hand side): P[n][.] and q = gr + P[n][.].
__global__ __launch_bounds__(64) void syrk_convert_kernel(SyrkConvertArgs c) {
  const int rb = blockIdx.x, rb2 = blockIdx.y;
  const int a = threadIdx.x >> 3, b = threadIdx.x & 7;
  if (a >= 7 || b >= 7) return;
  // lower triangle only, mirrored: P stays exactly symmetric
  if (rb2 < rb || (rb2 == rb && a < b)) return;
  double E[12];
  // column side (rb, parameter b)
  double cb[3];
  int nb, jb;
  if (b < 4) {
    quat_rotvec_map(c.x + 7 * (int64_t)c.rvar[rb], E);
    cb[0] = E[b]; cb[1] = E[4 + b]; cb[2] = E[8 + b];
    nb = 3; jb = 0;
  } else {
    cb[0] = 1.0; cb[1] = cb[2] = 0.0;
    nb = 1; jb = 3 + (b - 4);
  }
  if (rb2 == c.n_rblocks) {
    if (a != 0) return;
    double acc = 0.0;
    for (int j = 0; j < nb; ++j) acc += c.Su[(int64_t)c.nu * c.ldu + 6 * rb + jb + j] * cb[j];
    c.P[(int64_t)c.n * c.ldn + 7 * rb + b] = acc;
    c.q[7 * rb + b] = c.gr[7 * rb + b] + acc;
    return;
  }
  double ca[3];
  int na, ia;
  if (a < 4) {
    quat_rotvec_map(c.x + 7 * (int64_t)c.rvar[rb2], E);
    ca[0] = E[a]; ca[1] = E[4 + a]; ca[2] = E[8 + a];
    na = 3; ia = 0;
  } else {
    ca[0] = 1.0; ca[1] = ca[2] = 0.0;
    na = 1; ia = 3 + (a - 4);
  }
  double acc = 0.0;
  for (int i = 0; i < na; ++i) {
    const double* row = c.Su + (int64_t)(6 * rb2 + ia + i) * c.ldu + 6 * rb + jb;
    double t = 0.0;
    for (int j = 0; j < nb; ++j) t += row[j] * cb[j];
    acc += ca[i] * t;
  }
  const int64_t o = (int64_t)(7 * rb2 + a) * c.ldn + 7 * rb + b;
  const double v = c.Hrr[o] + acc;
  c.P[o] = v;
  c.P[(int64_t)(7 * rb + b) * c.ldn + 7 * rb2 + a] = v;
}

// The same for the rows of the camera-parameter blocks (reference row n7 + blockIdx.x; their
// 12 columns are not mixed: T = I) against every column j <= row, and their entry of the
// augmented row.
__global__ __launch_bounds__(256) void syrk_convert_k_kernel(SyrkConvertArgs c) {
  const int i = c.n7 + blockIdx.x, ui = c.nu6 + blockIdx.x;
  const double* row = c.Su + (int64_t)ui * c.ldu;
  for (int j = threadIdx.x; j <= i; j += 256) {
    double acc;
    if (j < c.n7) {
      const int rb = j / 7, b = j % 7;
      if (b < 4) {
        double E[12];
        quat_rotvec_map(c.x + 7 * (int64_t)c.rvar[rb], E);
        acc = row[6 * rb] * E[b] + row[6 * rb + 1] * E[4 + b] + row[6 * rb + 2] * E[8 + b];
      } else {
        acc = row[6 * rb + 3 + (b - 4)];
      }
    } else {
      acc = row[c.nu6 + (j - c.n7)];
    }
    const double v = c.Hrr[(int64_t)i * c.ldn + j] + acc;
    c.P[(int64_t)i * c.ldn + j] = v;
    c.P[(int64_t)j * c.ldn + i] = v;
  }
  if (threadIdx.x == 0) {
    const double acc = c.Su[(int64_t)c.nu * c.ldu + ui];
    c.P[(int64_t)c.n * c.ldn + i] = acc;
    c.q[i] = c.gr[i] + acc;
  }
}

void launch_schur_syrk_w_product(const SyrkArgs& a, hipStream_t s) {
  const int n_pairs = syrk_tile_pairs(a.ldn);
  int k_per_split = (a.n_krows + a.ksplit - 1) / a.ksplit;
  k_per_split = (k_per_split + SY_KC - 1) / SY_KC * SY_KC;
  hipLaunchKernelGGL(syrk_kernel, dim3(n_pairs * a.ksplit), dim3(256), 0, s, a, n_pairs, k_per_split);
  hipLaunchKernelGGL(syrk_reduce_kernel, dim3(n_pairs, 32), dim3(256), 0, s, a, n_pairs);
}
void launch_schur_syrk_w_convert(const SyrkConvertArgs& c, hipStream_t s) {
  if (c.n_rblocks > 0)
    hipLaunchKernelGGL(syrk_convert_kernel, dim3(c.n_rblocks, c.n_rblocks + 1), dim3(64), 0, s, c);
  if (c.n > c.n7) hipLaunchKernelGGL(syrk_convert_k_kernel, dim3(c.n - c.n7), dim3(256), 0, s, c);
}

// ---------------------------------------------------------------- exchange buffer
// What the ranks all-reduce per linear solve: the lower triangle of P (row i at
// i (i + 1) / 2), then q, g_r and diag(H_rr): n (n + 1) / 2 + 3 n doubles instead of the
// ldn x ldn + 3 ldn of the working layout (3.6 MB instead of 8.4 MB on config D).
// where entry (i, j <= i) of P and entry j of vector `which` (0 q, 1 g_r, 2 diag) live in the
// buffer: one definition for the device kernels and the host versions below
__host__ __device__ __forceinline__ int64_t exchange_tri_offset(int i, int j) { return (int64_t)i * (i + 1) / 2 + j; }
__host__ __device__ __forceinline__ int64_t exchange_vec_offset(int n, int which, int j) {
  return (int64_t)n * (n + 1) / 2 + (int64_t)which * n + j;
}
__global__ __launch_bounds__(256) void exchange_pack_kernel(int n, int ldn, const double* __restrict__ pack,
                                                            double* __restrict__ buf) {
  const int i = blockIdx.y;
  if (i < n) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j <= i; j += 256 * gridDim.x)
      buf[exchange_tri_offset(i, j)] = pack[(int64_t)i * ldn + j];
  } else {  // i = n, n + 1, n + 2: q, g_r, diag
    const double* src = pack + (int64_t)ldn * ldn + (int64_t)(i - n) * ldn;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += 256 * gridDim.x) buf[exchange_vec_offset(n, i - n, j)] = src[j];
  }
}
__global__ __launch_bounds__(256) void exchange_unpack_kernel(int n, int ldn, const double* __restrict__ buf,
                                                              double* __restrict__ pack) {
  const int i = blockIdx.y;
  if (i < n) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j <= i; j += 256 * gridDim.x) {
      const double v = buf[exchange_tri_offset(i, j)];
      pack[(int64_t)i * ldn + j] = v;
      pack[(int64_t)j * ldn + i] = v;
    }
  } else {
    double* dst = pack + (int64_t)ldn * ldn + (int64_t)(i - n) * ldn;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += 256 * gridDim.x) dst[j] = buf[exchange_vec_offset(n, i - n, j)];
  }
}
// the same on host arrays (calico_rp_exchange_pack_host / _unpack_host: multi-process tests
// without a GPU drive the layout the ranks really exchange)
void exchange_pack_host(int n, int ldn, const double* pack, double* buf) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) buf[exchange_tri_offset(i, j)] = pack[(int64_t)i * ldn + j];
  for (int w = 0; w < 3; ++w)
    for (int j = 0; j < n; ++j) buf[exchange_vec_offset(n, w, j)] = pack[(int64_t)ldn * ldn + (int64_t)w * ldn + j];
}
void exchange_unpack_host(int n, int ldn, const double* buf, double* pack) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      const double v = buf[exchange_tri_offset(i, j)];
      pack[(int64_t)i * ldn + j] = v;
      pack[(int64_t)j * ldn + i] = v;
    }
  for (int w = 0; w < 3; ++w)
    for (int j = 0; j < n; ++j) pack[(int64_t)ldn * ldn + (int64_t)w * ldn + j] = buf[exchange_vec_offset(n, w, j)];
}
size_t exchange_count(int n) { return (size_t)n * (n + 1) / 2 + 3 * (size_t)n; }
void launch_exchange_pack(int n, int ldn, const double* pack, double* buf, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(exchange_pack_kernel, dim3((n + 255) / 256, n + 3), dim3(256), 0, s, n, ldn, pack, buf);
}
void launch_exchange_unpack(int n, int ldn, const double* buf, double* pack, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(exchange_unpack_kernel, dim3((n + 255) / 256, n + 3), dim3(256), 0, s, n, ldn, buf, pack);
}

// ---------------------------------------------------------------- reduced system
// One launch: block rows i < n fill row i of S, block row n the vectors (scale, diag, gscaled, rhs
// and the augmented row).  The matrix rows work out the scale and damping of their row and columns
// themselves wherever this launch is the one that (re)computes them, so nothing is read that the
// vector row writes.
__device__ __forceinline__ double reduced_scale_of(const ReducedArgs& a, int i, int v) {
  return a.compute_scale ? (a.use_scale ? 1.0 / (1.0 + sqrt(fmax(a.hdiag[i], 0.0))) : 1.0) : a.scale[v];
}
__global__ __launch_bounds__(256) void reduced_system_kernel(ReducedArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= a.n) return;
  const int vj = reduced_slot(j, a.n7, a.vn7, a.rvar);
  const double sj = reduced_scale_of(a, j, vj);
  if (i == a.n) {
    const double d = a.hdiag[j];
    if (a.compute_scale) a.scale[vj] = sj;
    if (!a.reuse_diag) a.diag[vj] = fmin(fmax(sj * sj * d, a.min_diag), a.max_diag);
    a.gscaled[vj] = sj * a.gr[j];
    a.rhs[j] = sj * a.q[j];
    a.S[(int64_t)a.n * a.ldn + j] = sj * a.q[j];  // augmented row: forward substitution rides the panels
    return;
  }
  const int vi = reduced_slot(i, a.n7, a.vn7, a.rvar);
  const double si = reduced_scale_of(a, i, vi);
  double v = si * sj * a.P[(int64_t)i * a.ldn + j];
  if (i == j) v += (a.reuse_diag ? a.diag[vi] : fmin(fmax(si * si * a.hdiag[i], a.min_diag), a.max_diag)) / a.radius;
  a.S[(int64_t)i * a.ldn + j] = v;
}
void launch_finalize_reduced(const ReducedArgs& a, hipStream_t s) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(reduced_system_kernel, dim3((a.n + 255) / 256, a.n + 1), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------- dense Cholesky
// Right-looking blocked Cholesky, 64-column panels, on the AUGMENTED matrix
//     [ S  . ]      row n holds rhs^T, so the panel products that form L21 also carry
//     [ b' . ]      out the forward substitution w = L^-1 b for free (row n of L).
// Per panel:
//   chol_diag   one wave, row per lane: Cholesky-Crout of the 64x64 block with the
//               finished rows broadcast from LDS, then its inverse V = L11^-1
//               (column per lane) so that everything after is a matrix product;
//   chol_panel  L21 = A21 V^T        (64x64x64 tiles, fp64 MFMA)
//   chol_update A22 -= L21 L21^T     (lower 64x64 tiles, fp64 MFMA)
// and finally one workgroup does the backward substitution L^T y = w block by block
// with the stored inverses.
constexpr int CH_NB = 64;
constexpr int CH_LD = CH_NB + 2;  // LDS row stride (doubles): rows 528 B apart -> the 16 rows an
                                  // MFMA operand read touches land on distinct bank pairs

// 64x64 diagonal block: L11 (in LDS only) and V = L11^-1 (to invL), one workgroup of 4 waves,
// block resident in LDS, 16-column sub-panels:
//   (1) wave 0: Cholesky-Crout of the 16x16 sub-block, row per lane, fused with its
//       inverse W (column per lane): both dot products of step j run over the same
//       broadcast row j of L, so W costs no extra latency;
//   (2) L21 = A21 W^T and (3) A22 -= L21 L21^T on 16x16x4 fp64 MFMA tiles;
// then V off-diagonal blocks by rows: V_ij = -W_i (sum_k L_ik V_kj), where the MFMA result
// layout of the inner sum is exactly the B-operand layout of the outer product; rows 1 and 2
// are formed beside the Crout of sub-blocks 2 and 3 by waves that would otherwise wait.
// Rows/columns >= nb (last, partial block) are padded with the identity.
constexpr int CH_SB = 16;

// value of lane `src` (compile-time constant) as a wave-uniform double
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, src);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ f64x4 mfma4(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// dev probe (tools/chol_probe.hip defines CH_PROBE): shader-clock stamps at the phase boundaries
#ifdef CH_PROBE
__device__ long long ch_probe_buf[64];
#define CH_TICK(i)                                                             \
  do {                                                                         \
    if (threadIdx.x == 0) ch_probe_buf[i] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define CH_TICK(i) ((void)0)
#endif

// value of lane J of every 16-lane row, in all lanes of that row
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true);  // row_newbcast:J
}

// Step J of the Cholesky of a 16 x 16 block, row r of L per lane (x) fused with column r of its
// inverse W (v; v[k] starts as the unit matrix's entry), right-looking: the pivot column is
// scaled, then columns c > J of both take their update from it, with lane c's entry of the pivot
// column broadcast over its 16-lane row.  Only the update of column J + 1 lies on the chain from
// one pivot to the next; the others fill the issue slots under the next pivot's square root
// (the left-looking form, whose dot products all sit between two pivots, took 4.5 k cycles per
// block against 16 x 140 for the bare chain; the square-root-free form was 9 % slower still).
template <int C>
__device__ __forceinline__ void crout_update(double (&x)[CH_SB], double (&v)[CH_SB], double xj, double vj) {
  const double l_ = row_bcast<C>(xj);
  x[C] = fma(-xj, l_, x[C]);
  v[C] = fma(-vj, l_, v[C]);
}
template <int J, int... Ds>
__device__ __forceinline__ void crout_updates(double (&x)[CH_SB], double (&v)[CH_SB], double xj, double vj,
                                              std::integer_sequence<int, Ds...>) {
  (crout_update<J + 1 + Ds>(x, v, xj, vj), ...);
}
template <int J>
__device__ __forceinline__ void crout_step(double (&x)[CH_SB], double (&v)[CH_SB], bool& bad) {
  const double d = row_bcast<J>(x[J]);
  if (!__builtin_amdgcn_class(d, 0x180)) bad = true;  // anything but a positive normal or denormal number
  // 1/sqrt(d): hardware estimate + two Newton steps (y += y (1/2 - d y^2 / 2))
  double rinv = __builtin_amdgcn_rsq(d);
  const double hd = -0.5 * d;
  rinv = fma(rinv, fma(hd * rinv, rinv, 0.5), rinv);
  rinv = fma(rinv, fma(hd * rinv, rinv, 0.5), rinv);
  // No masks: lanes r < J carry the unused upper triangle in x[J] (finite, never read as part of
  // L: every later use is of columns <= row), and v[J] is an exact zero for r > J by itself (its
  // updates were products with exact zeros).  On the diagonal x[J] = d, so d / sqrt(d) comes out
  // of the same product.
  const double xj = x[J] * rinv, vj = v[J] * rinv;
  x[J] = xj;
  v[J] = vj;
  crout_updates<J>(x, v, xj, vj, std::make_integer_sequence<int, CH_SB - 1 - J>{});
}
template <int... Js>
__device__ __forceinline__ void crout_steps(double (&x)[CH_SB], double (&v)[CH_SB], bool& bad,
                                            std::integer_sequence<int, Js...>) {
  (crout_step<Js>(x, v, bad), ...);
}

// Factor of the 64 x 64 block held in Ls (lower triangle, identity padding beyond the block's
// size) and its inverse into Vs (zero on entry), both LDS with row stride CH_LD.  Called by
// every thread of a workgroup of NWAVES waves (waves >= 4 only keep the barriers company).
template <int NWAVES>
__device__ __forceinline__ bool chol_diag_lds(double* Ls, double* Vs, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lg = lane >> 4;
  bool bad = false;
  // off-diagonal block (bi > bj) of V = L^-1:  V_ij = -W_i sum_{k=j}^{i-1} L_ik V_kj.  Needs W_i, the
  // finished column blocks k < i of L and rows < i of V: row i - 1 is formed by the idle waves
  // while wave 0 runs the Crout of sub-block i + 1, the last row's product with W after the loop.
  auto v_sum = [&](int bi, int bj) {
    f64x4 T = {0.0, 0.0, 0.0, 0.0};
    for (int k = bj; k < bi; ++k)
#pragma unroll
      for (int st = 0; st < 4; ++st)
        T = mfma4(Ls[(CH_SB * bi + li) * CH_LD + CH_SB * k + 4 * st + lg],
                  Vs[(CH_SB * k + 4 * st + lg) * CH_LD + CH_SB * bj + li], T);
    return T;
  };
  auto v_finish = [&](int bi, int bj, f64x4 T) {
    f64x4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 4; ++st) R = mfma4(Vs[(CH_SB * bi + li) * CH_LD + CH_SB * bi + 4 * st + lg], T[st], R);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Vs[(CH_SB * bi + lg + 4 * rr) * CH_LD + CH_SB * bj + li] = -R[rr];
  };
  // the inner sums of the last row do not need W of the last sub-block: they are taken beside its
  // Crout too (by the wave that has just formed the entry of row 2 they end on)
  f64x4 t_last = {0.0, 0.0, 0.0, 0.0};
  constexpr int LAST = CH_NB / CH_SB - 1;
#pragma unroll 1
  for (int kb = 0; kb < CH_NB / CH_SB; ++kb) {
    const int c0 = CH_SB * kb;
    CH_TICK(4 + 4 * kb);
    if (wave == 0) {
      // the four 16-lane rows compute identical copies; r = row of L / column of W.  Row j of L
      // reaches the other lanes of its 16-lane row by DPP row broadcast straight out of lane j's
      // registers: no LDS round trip and no scalar read on the chain from one column to the next.
      const int r = li;
      double x[CH_SB], v[CH_SB];
      const double* myrow = Ls + (c0 + r) * CH_LD + c0;
#pragma unroll
      for (int k = 0; k < CH_SB; ++k) {
        x[k] = myrow[k];  // the whole row: the part right of the diagonal is carried along unused
        v[k] = (k == r) ? 1.0 : 0.0;
      }
      crout_steps(x, v, bad, std::make_integer_sequence<int, CH_SB>{});
      double* outrow = Ls + (c0 + r) * CH_LD + c0;
#pragma unroll
      for (int k = 0; k < CH_SB; ++k) {
        outrow[k] = x[k];  // right of the diagonal: not part of L, never read
        Vs[(c0 + k) * CH_LD + c0 + r] = v[k];  // W[k][r]
      }
    } else {
      // kb >= 2: row kb - 1 of V (blocks disjoint from what wave 0 touches)
      if (wave < kb) v_finish(kb - 1, wave - 1, v_sum(kb - 1, wave - 1));
      if (kb == LAST && wave <= LAST) t_last = v_sum(LAST, wave - 1);
    }
    __syncthreads();
    CH_TICK(5 + 4 * kb);
    const int m = CH_NB / CH_SB - 1 - kb;  // 16-row tiles below the sub-block
    // (2) L21 = A21 W^T, one tile per wave (m <= 3)
    if (wave < m) {
      const int rb = c0 + CH_SB * (wave + 1);
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = mfma4(Ls[(rb + li) * CH_LD + c0 + 4 * st + lg], Vs[(c0 + li) * CH_LD + c0 + 4 * st + lg], acc);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Ls[(rb + lg + 4 * rr) * CH_LD + c0 + li] = acc[rr];
    }
    __syncthreads();
    CH_TICK(6 + 4 * kb);
    // (3) trailing tiles (ti >= tj).  Wave 0 takes only tile (0, 0), the next sub-block, and goes
    // straight on to its Crout (no barrier: nobody else touches that tile); the other waves share
    // the rest and meet wave 0 at the barrier behind the Crout, which is also what the next
    // sub-panel product waits for.
    for (int t = (wave == 0) ? 0 : wave; t < m * (m + 1) / 2; t += (wave == 0) ? m * (m + 1) / 2 : NWAVES - 1) {
      int ti = 0, rem = t;
      while (rem > ti) {
        rem -= ti + 1;
        ++ti;
      }
      const int tj = rem;
      const int ra = c0 + CH_SB * (ti + 1), rbm = c0 + CH_SB * (tj + 1);
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = mfma4(Ls[(ra + li) * CH_LD + c0 + 4 * st + lg], Ls[(rbm + li) * CH_LD + c0 + 4 * st + lg], acc);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Ls[(ra + lg + 4 * rr) * CH_LD + rbm + li] -= acc[rr];
    }
  }
  // last row of V
  CH_TICK(20);
  if (wave >= 1 && wave <= LAST) v_finish(LAST, wave - 1, t_last);
  __syncthreads();
  CH_TICK(21);
  return bad;
}

__global__ __launch_bounds__(256) void chol_diag_kernel(int n, int ldn, double* S, double* invL, int j0, int* fail) {
  __shared__ __attribute__((aligned(16))) double Ls[CH_NB * CH_LD];
  __shared__ __attribute__((aligned(16))) double Vs[CH_NB * CH_LD];
  const int tid = threadIdx.x;
  CH_TICK(0);
  const int nb = (n - j0 < CH_NB) ? (n - j0) : CH_NB;
  {
    // all 16 loads of a thread in flight together: clamped addresses, no branch around a load
    double reg[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = tid + 256 * q, i = e >> 6, k = e & 63;
      const bool ok = i < nb && k < nb && k <= i;
      reg[q] = S[(int64_t)(j0 + (ok ? i : 0)) * ldn + j0 + (ok ? k : 0)];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = tid + 256 * q, i = e >> 6, k = e & 63;
      Ls[i * CH_LD + k] = (i < nb && k < nb) ? ((k <= i) ? reg[q] : 0.0) : ((k == i) ? 1.0 : 0.0);
      Vs[i * CH_LD + k] = 0.0;
    }
  }
  __syncthreads();
  CH_TICK(1);
  const bool bad = chol_diag_lds<4>(Ls, Vs, tid);
  CH_TICK(30);
  if (bad && (tid & 63) == 0) atomicExch(fail, 1);
  double* inv = invL + (int64_t)(j0 / CH_NB) * CH_NB * CH_NB;
  {
    // only the inverse leaves: nothing after this reads the factor's diagonal blocks themselves
    double rv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = tid + 256 * q;
      rv[q] = Vs[(e >> 6) * CH_LD + (e & 63)];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) inv[tid + 256 * q] = rv[q];  // row-major V
  }
  CH_TICK(31);
}

// C(64 x 64) = A(64 x 64) * B(64 x 64)^T on MFMA from row-major LDS tiles of stride CH_LD.
// 8 waves: wave w owns rows (w>>2)*32..+32 (two MFMA tiles) x columns (w&3)*16..+16.
__device__ __forceinline__ void tile_abt_64(const double* sA, const double* sB, int wave, int lane, f64x4 acc[2]) {
  const int wi = (wave >> 2) * 32, wj = (wave & 3) * 16;
  acc[0] = (f64x4){0.0, 0.0, 0.0, 0.0};
  acc[1] = acc[0];
#pragma unroll
  for (int kk = 0; kk < CH_NB; kk += 4) {
    const int k = kk + (lane >> 4);
    const double a0 = sA[(wi + (lane & 15)) * CH_LD + k], a1 = sA[(wi + 16 + (lane & 15)) * CH_LD + k];
    const double bv = sB[(wj + (lane & 15)) * CH_LD + k];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1], 0, 0, 0);
  }
}
// the same, added to acc
__device__ __forceinline__ void tile_abt_64_acc(const double* sA, const double* sB, int wave, int lane, f64x4 acc[2]) {
  const int wi = (wave >> 2) * 32, wj = (wave & 3) * 16;
#pragma unroll
  for (int kk = 0; kk < CH_NB; kk += 4) {
    const int k = kk + (lane >> 4);
    const double a0 = sA[(wi + (lane & 15)) * CH_LD + k], a1 = sA[(wi + 16 + (lane & 15)) * CH_LD + k];
    const double bv = sB[(wj + (lane & 15)) * CH_LD + k];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1], 0, 0, 0);
  }
}
// acc (layout of tile_abt_64) -> row-major LDS tile
__device__ __forceinline__ void tile_store_64(double* sT, int wave, int lane, const f64x4 acc[2]) {
  const int wi = (wave >> 2) * 32, wj = (wave & 3) * 16;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) sT[(wi + 16 * m + (lane >> 4) + 4 * r) * CH_LD + wj + (lane & 15)] = acc[m][r];
}

// The inverse of the factor, Z = L^-1, formed beside the panel updates by workgroups the launches
// would not otherwise use, so that the backward substitution becomes one product x = Z^T w
// (chol_apply_inverse_kernel) instead of a chain of 15 dependent steps in one workgroup (80 us of
// the 0.41 ms on config D).  Z is kept TRANSPOSED in the part of Lout the factor does not use:
// tile (j, p) of Lout, j <= p, holds Z_pj^T (the diagonal tiles V_p^T), and
//     Z_pj^T = -(sum_{k = j}^{p-1} Z_kj^T L_pk^T) V_p^T
// has every operand as a row-major tile as stored.  The sum is taken as the rows appear, so that
// no launch carries a chain: launch p (V_p and row p - 1 of Z are known)
//   * extra grid row 0, workgroup j <= p:  Z_pj^T from the partial sum P_pj^T (terms k <= p - 2) and
//     the term k = p - 1; workgroup j == p copies V_p^T;
//   * extra grid row 1, one workgroup per (p' > p, j < p):  P_p'j^T (+)= Z_(p-1)j^T L_p'(p-1)^T.
// P uses tile (j, p') of its own buffer.  Rows >= n of L (row n is the right-hand side) count as
// zero; nothing is written beyond column n - 1.
struct InvTileIndex {
  int tid, wave, lane, wi, wj;
  __device__ InvTileIndex() {
    tid = threadIdx.x;
    wave = tid >> 6;
    lane = tid & 63;
    wi = (wave >> 2) * 32;
    wj = (wave & 3) * 16;
  }
};
// acc = Z_kj^T L_rk^T (+ the tile (j, r) of P if with_p), in the layout of tile_abt_64; rv: V of block r
// fetched alongside when wanted
template <bool WITH_V>
__device__ __forceinline__ void inv_term(int n, int ldn, const double* Lout, const double* P, const double* inv, int j,
                                         int k, int r, bool with_p, const InvTileIndex& ix, double* sA, double* sB,
                                         double* sV, f64x4 acc[2]) {
  const int jj = j * CH_NB, jr = r * CH_NB;
  double ra[8], rb[8], rv[8], pt[2][4];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = ix.tid + 512 * q, row = e >> 6, col = e & 63;
    ra[q] = Lout[(int64_t)(jj + row) * ldn + CH_NB * k + col];
    const int rr = (jr + row < n) ? jr + row : n - 1;
    rb[q] = Lout[(int64_t)rr * ldn + CH_NB * k + col];
    if (WITH_V) rv[q] = inv[e];
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = ix.wi + 16 * m + (ix.lane >> 4) + 4 * q, rr = jr + ix.wj + (ix.lane & 15);
      const double v = P[(int64_t)(jj + cc) * ldn + (rr < n ? rr : n - 1)];  // clamped: no branch around the load
      pt[m][q] = (with_p && rr < n) ? v : 0.0;
    }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = ix.tid + 512 * q, row = e >> 6, col = e & 63;
    sA[row * CH_LD + col] = ra[q];
    sB[row * CH_LD + col] = (jr + row < n) ? rb[q] : 0.0;
    if (WITH_V) sV[row * CH_LD + col] = rv[q];
  }
  __syncthreads();
  tile_abt_64(sA, sB, ix.wave, ix.lane, acc);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[m][q] += pt[m][q];
}
__device__ __forceinline__ void inv_tile_store(int n, int ldn, double* dst, int j, int r, const InvTileIndex& ix,
                                               const f64x4 acc[2], double sign) {
  const int jj = j * CH_NB, jr = r * CH_NB;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = ix.wi + 16 * m + (ix.lane >> 4) + 4 * q, rr = jr + ix.wj + (ix.lane & 15);
      if (rr < n) dst[(int64_t)(jj + cc) * ldn + rr] = sign * acc[m][q];
    }
}
__device__ __forceinline__ void chol_inverse_row(int n, int ldn, double* Lout, const double* P,
                                                 const double* __restrict__ invL, int p, int j, double* sA, double* sB,
                                                 double* sV) {
  if (j > p) return;
  const InvTileIndex ix;
  const int jp = p * CH_NB;
  const double* inv = invL + (int64_t)p * CH_NB * CH_NB;
  if (j == p) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = ix.tid + 512 * q, cc = e >> 6, r = e & 63;
      const double v = inv[r * CH_NB + cc];
      if (jp + cc < n && jp + r < n) Lout[(int64_t)(jp + cc) * ldn + jp + r] = v;
    }
    return;
  }
  f64x4 acc[2];
  inv_term<true>(n, ldn, Lout, P, inv, j, p - 1, p, j < p - 1, ix, sA, sB, sV, acc);
  __syncthreads();
  tile_store_64(sA, ix.wave, ix.lane, acc);  // the whole sum, transposed: [column of Z][row of Z]
  __syncthreads();
  f64x4 res[2];
  tile_abt_64(sA, sV, ix.wave, ix.lane, res);
  inv_tile_store(n, ldn, Lout, j, p, ix, res, -1.0);
}
__device__ __forceinline__ void chol_inverse_ahead(int n, int ldn, const double* Lout, double* P, int p, int idx,
                                                   double* sA, double* sB) {
  if (p == 0) return;
  const int r = p + 1 + idx / p, j = idx % p;
  if (r >= (n + CH_NB - 1) / CH_NB) return;
  const InvTileIndex ix;
  f64x4 acc[2];
  inv_term<false>(n, ldn, Lout, P, nullptr, j, p - 1, r, j < p - 1, ix, sA, sB, nullptr, acc);
  inv_tile_store(n, ldn, P, j, r, ix, acc, 1.0);
}

// x = L^-T w = Z^T w from the transposed inverse in the upper part of Lout (chol_inverse_row),
// w = row n of Lout: one wave per unknown c, x_c = sum_{r >= 64 (c / 64)} Lout[c][r] w_r (the entries
// left of the diagonal inside the block are the zeros of V^T), terms in a fixed order.
__global__ __launch_bounds__(256) void chol_apply_inverse_kernel(int n, int ldn, const double* __restrict__ Lz,
                                                                 double* __restrict__ out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= n) return;  // wave-uniform
  const double* zr = Lz + (int64_t)c * ldn;
  const double* w = Lz + (int64_t)n * ldn;
  double s0 = 0.0, s1 = 0.0;
  for (int rb = c & ~63; rb < n; rb += 512) {
    double z[8], ww[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = rb + lane + 64 * q;
      const int rc = r < n ? r : c;  // clamped: no branch around a load
      z[q] = zr[rc];
      ww[q] = w[rc];
    }
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      s0 = fma((rb + lane + 64 * q < n) ? z[q] : 0.0, ww[q], s0);
      s1 = fma((rb + lane + 64 * (q + 1) < n) ? z[q + 1] : 0.0, ww[q + 1], s1);
    }
  }
  double s = s0 + s1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[c] = s;
}

// Panel solve and trailing update in one launch.  Workgroup (ti >= tj) over the 64-row
// tiles below the diagonal block (rows j0+nb .. n, row n = rhs):
//     L_i = A21[ti] V^T,  L_j = A21[tj] V^T,  S[ti][tj] -= L_i L_j^T
// A21 is only read here (the update writes columns >= j0+64), so every workgroup may
// recompute the two panel tiles it needs; the diagonal workgroups (ti == tj) write L_i to
// Lout, the factor's home below the diagonal blocks (S keeps A21 until overwritten never:
// the backward substitution reads Lout).
// Workgroup (0, 0) owns the tile that is the NEXT diagonal block: once updated it is factored
// on the spot (chol_diag_lds) and its inverse published for the next launch, so a panel costs
// one launch instead of two and the block never makes the round trip through HBM.
__global__ __launch_bounds__(512) void chol_panel_update_kernel(int n, int ldn, double* S, double* Lout, double* P,
                                                                double* invL, int j0, int* fail) {
  __shared__ __attribute__((aligned(16))) double sI[CH_NB * CH_LD];
  __shared__ __attribute__((aligned(16))) double sJ[CH_NB * CH_LD];
  __shared__ __attribute__((aligned(16))) double sV[CH_NB * CH_LD];
  const int ti = blockIdx.x, tj = blockIdx.y;
  const int nb = (n - j0 < CH_NB) ? (n - j0) : CH_NB;
  const int base = j0 + nb;
  const int nt = (n + 1 - base + 63) / 64;  // 64-row tiles below the block (rows base .. n)
  if (tj >= nt) {  // the two extra grid rows (launch_cholesky_solve): the inverse of the factor
    if (tj == nt)
      chol_inverse_row(n, ldn, Lout, P, invL, j0 / CH_NB, ti, sI, sJ, sV);
    else
      chol_inverse_ahead(n, ldn, Lout, P, j0 / CH_NB, ti, sI, sJ);
    return;
  }
  if (tj > ti || ti >= nt) return;
  const int r0 = base + ti * 64, c0 = base + tj * 64;
  if (r0 > n) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool diag_tile = (ti == tj);
  const double* inv = invL + (int64_t)(j0 / CH_NB) * CH_NB * CH_NB;
  // the C tile entries this lane will update, fetched ahead of the products
  const int wi = (wave >> 2) * 32, wj = (wave & 3) * 16;
  double cpre[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = r0 + wi + 16 * m + (lane >> 4) + 4 * r, gj = c0 + wj + (lane & 15);
      const bool ok = gi <= n && gj <= gi && gj < n;
      const double c = S[ok ? (int64_t)gi * ldn + gj : (int64_t)r0 * ldn + c0];  // clamped: no branch around the load
      cpre[m][r] = ok ? c : 0.0;
    }
  {
    // the panel tiles and V: all loads of a thread in flight together (clamped addresses)
    double ra[8], rb[8], rv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 512 * q, row = e >> 6, k = e & 63;
      const int kc = k < nb ? k : nb - 1;
      const int ri = r0 + row <= n ? r0 + row : n, rj = c0 + row <= n ? c0 + row : n;
      ra[q] = S[(int64_t)ri * ldn + j0 + kc];
      rb[q] = diag_tile ? 0.0 : S[(int64_t)rj * ldn + j0 + kc];
      rv[q] = inv[e];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 512 * q, row = e >> 6, k = e & 63;
      sI[row * CH_LD + k] = (r0 + row <= n && k < nb) ? ra[q] : 0.0;
      if (!diag_tile) sJ[row * CH_LD + k] = (c0 + row <= n && k < nb) ? rb[q] : 0.0;
      sV[row * CH_LD + k] = rv[q];  // V[c=row][k]
    }
  }
  __syncthreads();
  f64x4 li[2], lj[2];
  tile_abt_64(sI, sV, wave, lane, li);
  if (!diag_tile) tile_abt_64(sJ, sV, wave, lane, lj);
  __syncthreads();
  tile_store_64(sI, wave, lane, li);
  if (!diag_tile) tile_store_64(sJ, wave, lane, lj);
  if (diag_tile) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = r0 + wi + 16 * m + (lane >> 4) + 4 * r, cj = wj + (lane & 15);
        if (gi <= n && cj < nb) Lout[(int64_t)gi * ldn + j0 + cj] = li[m][r];
      }
  }
  __syncthreads();
  f64x4 acc[2];
  tile_abt_64(sI, diag_tile ? sI : sJ, wave, lane, acc);
  const bool next_diag = (ti == 0 && tj == 0 && base < n);  // workgroup-uniform
  const int nb2 = (n - base < CH_NB) ? (n - base) : CH_NB;  // size of the next diagonal block
  if (next_diag) __syncthreads();                            // sI is about to become the block
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int li = wi + 16 * m + (lane >> 4) + 4 * r, lj = wj + (lane & 15);
      const int gi = r0 + li, gj = c0 + lj;
      const double val = cpre[m][r] - acc[m][r];
      if (gi <= n && gj <= gi && gj < n) S[(int64_t)gi * ldn + gj] = val;
      if (next_diag) {
        sI[li * CH_LD + lj] = (li < nb2 && lj < nb2) ? ((lj <= li) ? val : 0.0) : ((li == lj) ? 1.0 : 0.0);
        sJ[li * CH_LD + lj] = 0.0;
      }
    }
  if (!next_diag) return;
  __syncthreads();
  const bool bad = chol_diag_lds<8>(sI, sJ, tid);
  if (bad && lane == 0) atomicExch(fail, 1);
  double* inv2 = invL + (int64_t)(base / CH_NB) * CH_NB * CH_NB;
  double rv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = tid + 512 * q;
    rv[q] = sJ[(e >> 6) * CH_LD + (e & 63)];
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) inv2[tid + 512 * q] = rv[q];
}

void launch_cholesky_solve(int n, int ldn, double* S, double* Lout, double* invL, double* rhs, double* P,
                           int* fail, hipStream_t s) {
  if (n <= 0) return;
  // the first diagonal block on its own; every later one is factored by workgroup (0, 0) of the
  // launch that finishes its update
  hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s, n, ldn, S, invL, 0, fail);
  for (int j0 = 0; j0 < n; j0 += CH_NB) {
    const int nb = (n - j0 < CH_NB) ? (n - j0) : CH_NB;
    const int below = n + 1 - (j0 + nb);  // rows below the block, rhs row n included: >= 1
    const int nt = (below + 63) / 64;
    // two more grid rows: block row p of the inverse (workgroups 0 .. p) and the partial sums of
    // the rows after it (one workgroup per (row, column < p))
    const int p = j0 / CH_NB, nblk = (n + CH_NB - 1) / CH_NB;
    int width = nt > p + 1 ? nt : p + 1;
    if ((nblk - 1 - p) * p > width) width = (nblk - 1 - p) * p;
    hipLaunchKernelGGL(chol_panel_update_kernel, dim3(width, nt + 2), dim3(512), 0, s, n, ldn, S, Lout, P, invL, j0, fail);
  }
  hipLaunchKernelGGL(chol_apply_inverse_kernel, dim3((n + 3) / 4), dim3(256), 0, s, n, ldn, Lout, rhs);
}

// ---------------------------------------------------------------- back-substitution
// reduced part: w = s_r y_r, delta = -w, candidate, partial scalars (single block)
constexpr int BSR_THREADS = 1024;  // one workgroup (three global sums); ldn = 1024 on config D: one pass
__global__ __launch_bounds__(BSR_THREADS) void backsub_reduced_kernel(BacksubArgs a) {
  __shared__ double sh[3][BSR_THREADS / 64];
  double m = 0.0, d2 = 0.0, x2 = 0.0;
  for (int i = threadIdx.x; i < a.ldn; i += BSR_THREADS) {
    if (i < a.n) {
      const int v = reduced_slot(i, a.n7, a.vn7, a.rvar);
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
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int k = 0; k < BSR_THREADS / 64; ++k) t += sh[threadIdx.x][k];
    a.scalars[threadIdx.x] = t;
  }
  if (a.wspace) {
    // the step of each reduced block in the six columns Y is stored in: wu = T w  (w: this
    // workgroup's own stores, visible behind the barrier above)
    for (int i = threadIdx.x; i < a.ldy; i += BSR_THREADS) {
      double v = 0.0;
      if (i >= 6 * a.n_rblocks) {  // camera-parameter blocks keep their columns
        if (i < a.ny) v = a.w[a.n7 + (i - 6 * a.n_rblocks)];
      } else if (i < a.ny) {
        const int rb = i / 6, m = i % 6;
        const double* wb = a.w + 7 * rb;
        if (m < 3) {
          double E[12];
          quat_rotvec_map(a.x + 7 * (int64_t)a.rvar[rb], E);
          v = E[4 * m] * wb[0] + E[4 * m + 1] * wb[1] + E[4 * m + 2] * wb[2] + E[4 * m + 3] * wb[3];
        } else {
          v = wb[4 + (m - 3)];
        }
      }
      a.wu[i] = v;
    }
  }
}

// Back-substitution of the time blocks in two launches.  backsub_dots_kernel: one wave per row of Y,
// v_i = z_i - <Y row, w> (a workgroup = the rows of one segment), nothing else: the launch streams Y
// once.  backsub_tail_kernel: one THREAD per segment for the serial part (rhs from the six-row
// form, L^T y = v, step, candidate, partial scalars).  As one kernel (a workgroup per segment whose
// thread 0 ran the serial part behind a barrier) every workgroup held its 447 other threads
// resident through a few microseconds of dependent fp64 work: 0.115 ms for 0.39 GB on config D.
__global__ __launch_bounds__(448) void backsub_dots_kernel(BacksubArgs a) {
  const int sgi = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ldy = a.wspace ? a.ldy : a.ldn, ny = a.wspace ? a.ny : a.n;
  const double* wv = a.wspace ? a.wu : a.w;
  const bool six = a.wspace;
  const int rows = six ? 6 : 7;  // = waves of the workgroup
  const double* yrow = a.Y + (int64_t)(rows * sgi + wave) * ldy;
  // sixteen entries of the row and of w per lane requested together (clamped: no branch around a
  // load); the terms are added in column order per lane
  double s = 0.0;
  for (int base = 0; base < ny; base += 128 * 8) {  // rows and w are 16-byte aligned, their padding is readable
    double2 yv[8], ww[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = base + 2 * lane + 128 * q;
      const int ic = i < ny ? i : 0;
      yv[q] = *reinterpret_cast<const double2*>(yrow + ic);
      ww[q] = *reinterpret_cast<const double2*>(wv + ic);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = base + 2 * lane + 128 * q;
      s += (i < ny) ? yv[q].x * ww[q].x : 0.0;
      s += (i + 1 < ny) ? yv[q].y * ww[q].y : 0.0;
    }
  }
  s = wsum(s);
  if (lane == 0) a.seg_partial[(int64_t)sgi * SEGP + SEGP_V + wave] = (six ? 0.0 : a.z[(int64_t)sgi * 7 + wave]) - s;
}
__global__ __launch_bounds__(64) void backsub_tail_kernel(BacksubArgs a) {
  const int sgi = blockIdx.x * 64 + threadIdx.x;
  if (sgi >= a.n_seg) return;
  const bool six = a.wspace;
  const int v7 = 7 * a.seg_var[sgi];
  const double* L = a.Lt + (int64_t)sgi * 49;
  double* part = a.seg_partial + (int64_t)sgi * SEGP;
  double Lr[28], vv[7], sc[7], xv[7], gs[7], dg[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    vv[i] = part[SEGP_V + (i < (six ? 6 : 7) ? i : 0)];
    sc[i] = a.scale[v7 + i];
    xv[i] = a.x[v7 + i];
    gs[i] = a.gscaled[v7 + i];
    dg[i] = a.diag[v7 + i];
#pragma unroll
    for (int k = 0; k <= i; ++k) Lr[i * (i + 1) / 2 + k] = L[7 * i + k];
  }
  double y[7], rhs[7];
  if (six) {
    // t = z6 - Y6 w, rhs = Q t  (= z - Y w of the seven-row form: W = Q R, Y = Q Y6, z = Q z6)
    const double* sW = a.segw + (int64_t)sgi * SEGW;
    double t6[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) t6[i] = sW[SEGW_Z + i] + vv[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) t += sW[7 * m + i] * t6[m];
      rhs[i] = t;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 7; ++i) rhs[i] = vv[i];
  }
  double m = 0.0, d2 = 0.0, x2 = 0.0;
#pragma unroll
  for (int i = 6; i >= 0; --i) {
    double t = rhs[i];
#pragma unroll
    for (int k = i + 1; k < 7; ++k) t -= Lr[k * (k + 1) / 2 + i] * y[k];
    y[i] = t / Lr[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const double w = sc[i] * y[i];
    a.delta[v7 + i] = -w;
    a.cand[v7 + i] = xv[i] - w;
    m += 0.5 * y[i] * gs[i] + 0.5 * (dg[i] / a.radius) * y[i] * y[i];
    d2 += w * w;
    x2 += xv[i] * xv[i];
  }
  part[0] = m;
  part[1] = d2;
  part[2] = x2;
}
// scalars[3+k] = sum over segments (fixed order)
__global__ __launch_bounds__(256) void backsub_sum_kernel(BacksubArgs a) {
  __shared__ double sh[3][4];
  double acc[3] = {0.0, 0.0, 0.0};
  // eight segments' partials in flight per thread (one workgroup: the loop is all latency), added
  // in segment order
  for (int i0 = threadIdx.x; i0 < a.n_seg; i0 += 256 * 8) {
    double v[8][3];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + 256 * q;
      const double* p = a.seg_partial + (int64_t)(i < a.n_seg ? i : i0) * SEGP;
#pragma unroll
      for (int k = 0; k < 3; ++k) v[q][k] = p[k];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (i0 + 256 * q < a.n_seg)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += v[q][k];
  }
  for (int k = 0; k < 3; ++k) {
    const double v = wsum(acc[k]);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) a.scalars[3 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

void launch_backsub(const BacksubArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(backsub_reduced_kernel, dim3(1), dim3(BSR_THREADS), 0, s, a);
  if (a.n_seg > 0) {
    const int rows = a.wspace ? 6 : 7;
    hipLaunchKernelGGL(backsub_dots_kernel, dim3(a.n_seg), dim3(64 * rows), 0, s, a);
    hipLaunchKernelGGL(backsub_tail_kernel, dim3((a.n_seg + 63) / 64), dim3(64), 0, s, a);
    if (!a.defer_sum) hipLaunchKernelGGL(backsub_sum_kernel, dim3(1), dim3(256), 0, s, a);
  }
}

}  // namespace rp
