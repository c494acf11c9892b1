// rp_kernels.hip — observation-sweep and block-accumulation kernels (gfx950).
//
//  jacobian_sweep : residual + analytic Jacobian per corner observation
//                   (replaces AutoDiffCostFunction::Evaluate of the functors at
//                   /root/reference/src/solving-structure.hpp:643-1310), SoA planes,
//                   HBM-bound: 24 B read + 352 B written per observation.
//  cost_sweep     : residual only, block-reduced (Ceres' cost-only evaluation of the
//                   candidate point inside ceres::Solve, solving-structure.cpp:982).
//  accumulate     : per-FR Gram blocks [J r]^T [J r]; the J tile is staged in LDS,
//                   rows are split over 5 lane groups, partials are combined through
//                   LDS (replaces Ceres' block-sparse J^T J assembly).
//  gathers        : fixed-order (deterministic) sums of FR records into the
//                   camera/pattern blocks and the per-time blocks.
#include "rp_internal.hpp"
#include "rp_model.hpp"

namespace rp {

// ---------------------------------------------------------------- Jacobian sweep
__global__ __launch_bounds__(SWEEP_BLOCK) void jacobian_sweep_kernel(SweepArgs a) {
  const int64_t stride = (int64_t)gridDim.x * SWEEP_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * SWEEP_BLOCK + threadIdx.x; i < a.n_obs; i += stride) {
    const int f = a.obs_fr[i];
    const int c = a.fr_cam[f], p = a.fr_pat[f], t = a.fr_tim[f];
    const int j = a.obs_corner[i];
    const double X[3] = {(double)a.pts[3 * j], (double)a.pts[3 * j + 1], (double)a.pts[3 * j + 2]};
    double r[2], J[42];
    eval_observation<true>(a.x + 7 * (int64_t)c, p >= 0 ? a.x + 7 * (int64_t)p : nullptr,
                           t >= 0 ? a.x + 7 * (int64_t)t : nullptr, a.intr + 12 * (int64_t)c, X,
                           a.obs_u[i], a.obs_v[i], r, J);
    a.r[i] = r[0];
    a.r[a.n_obs + i] = r[1];
#pragma unroll
    for (int k = 0; k < 42; ++k) a.J[(int64_t)k * a.n_obs + i] = J[k];
  }
}

void launch_jacobian_sweep(const SweepArgs& a, hipStream_t s) {
  if (a.n_obs <= 0) return;
  int64_t blocks = (a.n_obs + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
  if (blocks > SWEEP_MAX_GRID) blocks = SWEEP_MAX_GRID;
  hipLaunchKernelGGL(jacobian_sweep_kernel, dim3((unsigned)blocks), dim3(SWEEP_BLOCK), 0, s, a);
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
// sum over a 256-thread block, result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sh /*4*/) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double tot = 0.0;
  if (threadIdx.x == 0) tot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return tot;
}

// ---------------------------------------------------------------- cost sweep
__global__ __launch_bounds__(SWEEP_BLOCK) void cost_sweep_kernel(SweepArgs a, double* partial) {
  __shared__ double sh[4];
  const int64_t stride = (int64_t)gridDim.x * SWEEP_BLOCK;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SWEEP_BLOCK + threadIdx.x; i < a.n_obs; i += stride) {
    const int f = a.obs_fr[i];
    const int c = a.fr_cam[f], p = a.fr_pat[f], t = a.fr_tim[f];
    const int j = a.obs_corner[i];
    const double X[3] = {(double)a.pts[3 * j], (double)a.pts[3 * j + 1], (double)a.pts[3 * j + 2]};
    double r[2];
    eval_observation<false>(a.x + 7 * (int64_t)c, p >= 0 ? a.x + 7 * (int64_t)p : nullptr,
                            t >= 0 ? a.x + 7 * (int64_t)t : nullptr, a.intr + 12 * (int64_t)c, X,
                            a.obs_u[i], a.obs_v[i], r, nullptr);
    acc += r[0] * r[0] + r[1] * r[1];
  }
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single block: out = scale * sum(partial[0..n)) in a fixed tree order
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* partial, int n, double scale,
                                                           double* out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) *out = scale * tot;
}

void launch_cost_sweep(const SweepArgs& a, double* partial, double* cost_out, hipStream_t s) {
  int64_t blocks = (a.n_obs + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
  if (blocks > SWEEP_MAX_GRID) blocks = SWEEP_MAX_GRID;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cost_sweep_kernel, dim3((unsigned)blocks), dim3(SWEEP_BLOCK), 0, s, a, partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, (int)blocks, 0.5, cost_out);
}

// ---------------------------------------------------------------- per-FR accumulation
// One wave per FR.  LDS tile: up to 32 observations = 64 rows of 26 doubles
//   [c0..c6 r | p0..p6 0 | t0..t6 0 | pad pad]   (row stride 208 B keeps 16-B alignment)
// Lane = kg*12 + unit: unit = pair*2 + half computes rows a in [4*half, 4*half+4) x 8
// columns of one block pair over the tile rows congruent to kg mod 5; lanes 60..63
// only help with staging.  The 5 partial sums per output are combined through LDS.
constexpr int ACC_LD = 26;
constexpr int ACC_WAVE_LDS = 1920;  // doubles: max(64*26 tile, 60*32 partials)

__global__ __launch_bounds__(256) void accumulate_kernel(int n_fr, const int64_t* __restrict__ fr_obs_begin,
                                                         const double* __restrict__ J,
                                                         const double* __restrict__ r, int64_t n_obs,
                                                         double* __restrict__ rec) {
  __shared__ __attribute__((aligned(16))) double lds[4 * ACC_WAVE_LDS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + wave;
  if (f >= n_fr) return;  // whole wave; no block-level barrier is used below
  double* T = lds + wave * ACC_WAVE_LDS;
  const int64_t b = fr_obs_begin[f], e = fr_obs_begin[f + 1];

  const int kg = lane / 12, unit = lane - kg * 12;
  const int pair = unit >> 1, half = unit & 1;
  // A block / B block column offsets in the tile row per pair
  const int a_blk = (pair == 0) ? 0 : (pair == 1 || pair == 3) ? 8 : 16;
  const int b_blk = (pair <= 2) ? 0 : (pair <= 4) ? 8 : 16;
  const int a_off = a_blk + 4 * half, b_off = b_blk;

  double acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) acc[k] = 0.0;

  // zero the pad columns once (15, 23, 24, 25)
  {
    const int row = lane;
    T[row * ACC_LD + 15] = 0.0;
    T[row * ACC_LD + 23] = 0.0;
    T[row * ACC_LD + 24] = 0.0;
    T[row * ACC_LD + 25] = 0.0;
  }

  const int o = lane & 31, h = lane >> 5;
  for (int64_t c0 = b; c0 < e; c0 += 32) {
    const int nobs = (int)((e - c0) < 32 ? (e - c0) : 32);
    // ---- stage: plane k = 2*i + h for observation o
    if (o < nobs) {
      const int64_t gi = c0 + o;
#pragma unroll
      for (int i = 0; i < 22; ++i) {
        const int k = 2 * i + h;  // 0..43
        double val;
        int row, col;
        if (k < 42) {
          val = J[(int64_t)k * n_obs + gi];
          const int rr = (k >= 21) ? 1 : 0;
          const int j = k - 21 * rr;
          const int blk = j / 7;
          row = 2 * o + rr;
          col = blk * 8 + (j - 7 * blk);
        } else {
          val = r[(int64_t)(k - 42) * n_obs + gi];
          row = 2 * o + (k - 42);
          col = 7;
        }
        T[row * ACC_LD + col] = val;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- accumulate rows kg, kg+5, ...
    if (kg < 5) {
      const int nrows = 2 * nobs;
      for (int row = kg; row < nrows; row += 5) {
        const double2* ap = reinterpret_cast<const double2*>(T + row * ACC_LD + a_off);
        const double2* bp = reinterpret_cast<const double2*>(T + row * ACC_LD + b_off);
        const double2 a01 = ap[0], a23 = ap[1];
        const double2 b01 = bp[0], b23 = bp[1], b45 = bp[2], b67 = bp[3];
        const double av[4] = {a01.x, a01.y, a23.x, a23.y};
        const double bv[8] = {b01.x, b01.y, b23.x, b23.y, b45.x, b45.y, b67.x, b67.y};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i * 8 + j] = fma(av[i], bv[j], acc[i * 8 + j]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- combine the 5 row groups through LDS
  if (kg < 5) {
#pragma unroll
    for (int k = 0; k < 32; ++k) T[lane * 32 + k] = acc[k];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double* out = rec + (int64_t)f * FR_REC;
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int idx = lane + 64 * m;  // = unit*32 + k
    double s = T[idx];
#pragma unroll
    for (int g = 1; g < 5; ++g) s += T[g * 12 * 32 + idx];
    out[idx] = s;
  }
}

void launch_accumulate(int n_fr, const int64_t* fr_obs_begin, const double* J, const double* r,
                       int64_t n_obs, double* rec, hipStream_t s) {
  if (n_fr <= 0) return;
  hipLaunchKernelGGL(accumulate_kernel, dim3((n_fr + 3) / 4), dim3(256), 0, s, n_fr, fr_obs_begin, J, r,
                     n_obs, rec);
}

// ---------------------------------------------------------------- gathers
// record offsets
constexpr int REC_CC = 0, REC_PC = 64, REC_TC = 128, REC_PP = 192, REC_TP = 256, REC_TT = 320;

// chunk_sum[chunk][0..63]=sum cc~, [64..127]=sum pc~, [128..191]=sum pp
__global__ __launch_bounds__(192) void pair_chunk_kernel(GatherArgs g) {
  const int ch = blockIdx.x, t = threadIdx.x;
  const int src = (t < 64) ? (REC_CC + t) : (t < 128) ? (REC_PC + t - 64) : (REC_PP + t - 128);
  double acc = 0.0;
  for (int i = g.chunk_begin[ch]; i < g.chunk_begin[ch + 1]; ++i)
    acc += g.rec[(int64_t)g.pair_fr[i] * FR_REC + src];
  g.chunk_sum[(int64_t)ch * 192 + t] = acc;
}
__global__ __launch_bounds__(192) void pair_group_kernel(GatherArgs g) {
  const int gr = blockIdx.x, t = threadIdx.x;
  double acc = 0.0;
  for (int ch = g.group_chunk_begin[gr]; ch < g.group_chunk_begin[gr + 1]; ++ch)
    acc += g.chunk_sum[(int64_t)ch * 192 + t];
  g.group_sum[(int64_t)gr * 192 + t] = acc;
}
// one block per reduced block: diagonal 7x7, gradient, and (for cameras) the
// pattern-camera off-diagonal blocks of its groups.
__global__ __launch_bounds__(64) void assemble_reduced_kernel(GatherArgs g) {
  const int rb = blockIdx.x, t = threadIdx.x;
  const int a = t >> 3, b = t & 7;  // 8x8 slot
  double acc = 0.0;
  bool is_cam = false;
  for (int k = g.rblock_group_begin[rb]; k < g.rblock_group_begin[rb + 1]; ++k) {
    const int gr = g.rblock_groups[k];
    const double* gs = g.group_sum + (int64_t)gr * 192;
    if (g.group_rc[gr] == rb) {
      is_cam = true;
      acc += gs[t];  // cc~
      const int rp = g.group_rp[gr];
      if (rp >= 0 && a < 7 && b < 7) {
        const double v = gs[64 + t];  // pc~[a][b]: row = pattern param a, col = camera param b
        g.Hrr[(int64_t)(7 * rp + a) * g.ldn + 7 * rb + b] = v;
        g.Hrr[(int64_t)(7 * rb + b) * g.ldn + 7 * rp + a] = v;
      }
    } else {
      // pattern block: pp at [128..], gradient from pc~[a][7]
      acc += (b < 7) ? gs[128 + t] : gs[64 + a * 8 + 7];
    }
  }
  if (a < 7) {
    if (b < 7) {
      g.Hrr[(int64_t)(7 * rb + a) * g.ldn + 7 * rb + b] = acc;
      if (a == b) g.hdiag[7 * rb + a] = acc;
    } else
      g.gr[7 * rb + a] = acc;
  }
  (void)is_cam;
}
// cost: sum over groups of cc~[7][7]
__global__ __launch_bounds__(256) void cost_from_groups_kernel(GatherArgs g) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < g.n_groups; i += 256) acc += g.group_sum[(int64_t)i * 192 + 63];
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) *g.cost_sum = tot;
}
// per eliminated segment: E = sum tt, g_t = sum tc~[.][7]
__global__ __launch_bounds__(64) void seg_diag_kernel(GatherArgs g) {
  const int sgi = blockIdx.x, t = threadIdx.x;
  const int a = t >> 3, b = t & 7;
  if (a >= 7) return;
  const int src = (b < 7) ? (REC_TT + a * 8 + b) : (REC_TC + a * 8 + 7);
  double acc = 0.0;
  for (int f = g.seg_fr_begin[sgi]; f < g.seg_fr_begin[sgi + 1]; ++f) acc += g.rec[(int64_t)f * FR_REC + src];
  if (b < 7)
    g.Ett[(int64_t)sgi * 49 + a * 7 + b] = acc;
  else
    g.gt[(int64_t)sgi * 7 + a] = acc;
}
// per (segment, reduced block): coupling block into Hrt[7*seg+i][7*rb+j]
__global__ __launch_bounds__(256) void unit_gather_kernel(GatherArgs g) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t u = gid / 49;
  if (u >= g.n_units) return;
  const int e = (int)(gid - u * 49);
  const int i = e / 7, j = e - 7 * i;  // i: time parameter, j: reduced-block parameter
  double acc = 0.0;
  for (int k = g.unit_begin[u]; k < g.unit_begin[u + 1]; ++k) {
    const int c = g.contrib[k];
    const int f = c >> 1;
    const int base = (c & 1) ? REC_TP : REC_TC;
    acc += g.rec[(int64_t)f * FR_REC + base + i * 8 + j];
  }
  g.Hrt[(int64_t)(7 * g.unit_seg[u] + i) * g.ldn + 7 * g.unit_rblock[u] + j] = acc;
}

void launch_gathers(const GatherArgs& g, hipStream_t s) {
  if (g.n_chunks > 0) {
    hipLaunchKernelGGL(pair_chunk_kernel, dim3(g.n_chunks), dim3(192), 0, s, g);
    hipLaunchKernelGGL(pair_group_kernel, dim3(g.n_groups), dim3(192), 0, s, g);
    hipLaunchKernelGGL(assemble_reduced_kernel, dim3(g.n_rblocks), dim3(64), 0, s, g);
    hipLaunchKernelGGL(cost_from_groups_kernel, dim3(1), dim3(256), 0, s, g);
  }
  if (g.n_seg > 0) {
    hipLaunchKernelGGL(seg_diag_kernel, dim3(g.n_seg), dim3(64), 0, s, g);
    const int64_t threads = (int64_t)g.n_units * 49;
    hipLaunchKernelGGL(unit_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, g);
  }
}

// ---------------------------------------------------------------- misc
__global__ __launch_bounds__(256) void gradient_max_kernel(const double* gr, int n, const double* gt,
                                                           int nt7, double* out) {
  __shared__ double sh[4];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, fabs(gr[i]));
  for (int i = threadIdx.x; i < nt7; i += 256) m = fmax(m, fabs(gt[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
void launch_gradient_max(const double* gr, int n, const double* gt, int nt7, double* out, hipStream_t s) {
  hipLaunchKernelGGL(gradient_max_kernel, dim3(1), dim3(256), 0, s, gr, n, gt, nt7, out);
}

__global__ __launch_bounds__(256) void permute_obs_kernel(int64_t n, const int64_t* perm, const double* u_in,
                                                          const double* v_in, const int32_t* corner_in,
                                                          double* u, double* v, int32_t* corner) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int64_t src = perm[i];
    u[i] = u_in[src];
    v[i] = v_in[src];
    corner[i] = corner_in[src];
  }
}
void launch_permute_obs(int64_t n, const int64_t* perm, const double* u_in, const double* v_in,
                        const int32_t* corner_in, double* u, double* v, int32_t* corner, hipStream_t s) {
  if (n <= 0) return;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(permute_obs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n, perm, u_in, v_in,
                     corner_in, u, v, corner);
}

}  // namespace rp
