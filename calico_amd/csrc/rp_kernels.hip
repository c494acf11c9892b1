// rp_kernels.hip — observation-sweep and block-accumulation kernels (gfx950).
//
//  fused7q[k]     : default path.  Residual + analytic Jacobian per corner observation
//                   (replaces AutoDiffCostFunction::Evaluate of the functors at
//                   /root/reference/src/solving-structure.hpp:643-1310), as the 7 chain
//                   columns of rp_model.hpp, evaluated into an LDS slab per wave and
//                   accumulated there on the fp64 matrix cores into per-part Gram records
//                   (replaces Ceres' block-sparse J^T J assembly); the Jacobian never reaches
//                   HBM.  prep_vars_kernel / prep_frs_kernel prepare [R | t] of every block
//                   and the per-FR table for it once per evaluation.
//  jacobian_sweep : the same evaluation with the Jacobian written to HBM as SoA planes,
//                   HBM-bound: 24 B read + 352 B written per observation (materialised
//                   mode and the calico_rp_eval parity hook).
//  accumulate     : Gram records from the Jacobian planes in HBM (materialised mode).
//  cost_sweep     : residual only, block-reduced (Ceres' cost-only evaluation of the
//                   candidate point inside ceres::Solve, solving-structure.cpp:982).
//  fr_error       : per-FR mean squared reprojection error (reporting).
//  gathers        : fixed-order (deterministic) sums of the records into the
//                   camera/pattern blocks and the per-time blocks; the *_w kernels take the
//                   fused kernel's records (rotation-vector columns) and convert the sums
//                   to the reference's seven columns per block.
#include <cstdlib>
#include "rp_internal.hpp"
#include "rp_model.hpp"

namespace rp {

__device__ __forceinline__ double block_sum_256(double v, double* sh);


// ---------------------------------------------------------------- Jacobian sweep
// One workgroup = one tile of SWEEP_TILE (256) consecutive observations.
//  1. The FRs the tile touches (usually 4-5, at most FR_TABLE per pass) are prepared
//     once into LDS: rotation matrices, 1/|q|^2, intrinsics (rp::prep_block), 4 lanes
//     per FR, instead of once per corner.
//  2. Every lane evaluates its observation from the LDS table and streams the 42
//     Jacobian entries + 2 residuals into the tile's [44][256] block through one
//     buffer descriptor: a single VGPR of address serves all 44 stores (a flat
//     pointer per plane costs 84 VGPRs and an occupancy step).
constexpr int FR_TABLE = 32;
constexpr int SWEEP_GROUP = 2;  // tiles per workgroup in the Jacobian sweep (measured best of 1/2/4/8)
#ifndef RP_COST_PAIRS
#define RP_COST_PAIRS 2
#endif
#ifndef RP_COST_GROUP
#define RP_COST_GROUP 8
#endif
constexpr int COST_GROUP = RP_COST_GROUP;   // tiles per table build in the cost sweep

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct TileSink {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
  __device__ __forceinline__ void operator()(int k, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, k * (SWEEP_TILE * 8), 0);
  }
};

__device__ __forceinline__ void build_fr_table(const SweepArgs& a, double* tab, int fb, int nf, int tid) {
  if (tid < 4 * nf) {
    const int fi = tid >> 2, part = tid & 3, ff = fb + fi;
    double* e = tab + fi * FRP_SIZE;
    const int c = a.fr_cam[ff];
    if (part == 0) {
      prep_block(a.x + 7 * (int64_t)c, e + FRP_CAM);
    } else if (part == 1) {
      const int t = a.fr_tim[ff];
      prep_block(t >= 0 ? a.x + 7 * (int64_t)t : nullptr, e + FRP_TIM);
    } else if (part == 2) {
      const int p = a.fr_pat[ff];
      prep_block(p >= 0 ? a.x + 7 * (int64_t)p : nullptr, e + FRP_PAT);
    } else {
      for (int k = 0; k < 12; ++k) e[FRP_K + k] = a.intr[12 * (int64_t)c + k];
    }
  }
}

// The same table for a residual-only sweep, from the blocks prep_vars_kernel prepared
// ([R | t] per variable in a.prep, slot n_vars = identity): no quaternion normalisation (an
// fp64 division) per FR and tile group; eval_prepared<false> reads nothing but R, t and K.
__device__ __forceinline__ void build_fr_table_prepped(const SweepArgs& a, double* tab, int fb, int nf, int tid) {
  if (tid < 4 * nf) {
    const int fi = tid >> 2, part = tid & 3, ff = fb + fi;
    double* e = tab + fi * FRP_SIZE;
    const int c = a.fr_cam[ff];
    const double* src;
    double* dst;
    if (part == 3) {
      src = a.intr + 12 * (int64_t)c;
      dst = e + FRP_K;
    } else {
      const int v = part == 0 ? c : part == 1 ? a.fr_tim[ff] : a.fr_pat[ff];
      src = a.prep + 12 * (int64_t)(v >= 0 ? v : a.n_vars);
      dst = e + (part == 0 ? FRP_CAM : part == 1 ? FRP_TIM : FRP_PAT);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) dst[k] = src[k];
  }
}

// One workgroup handles SWEEP_GROUP consecutive tiles: the dependent load chain of the
// table build (FR -> block indices -> parameters -> 1/|q|^2) is paid once per group,
// and the next tile's observation stream is fetched while the current tile computes.
__global__ __launch_bounds__(SWEEP_BLOCK, 3) void jacobian_sweep_kernel(SweepArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[FR_TABLE * FRP_SIZE];
  const int tid = threadIdx.x;
  const int64_t tile0 = (int64_t)blockIdx.x * SWEEP_GROUP;
  const int64_t i_first = tile0 * SWEEP_TILE;
  if (i_first >= a.n_obs) return;
  const int64_t i_end =
      (i_first + (int64_t)SWEEP_GROUP * SWEEP_TILE < a.n_obs) ? i_first + (int64_t)SWEEP_GROUP * SWEEP_TILE : a.n_obs;
  const int f_first = a.obs_fr[i_first], f_last = a.obs_fr[i_end - 1];
  for (int fb = f_first; fb <= f_last; fb += FR_TABLE) {
    const int nf = (f_last - fb + 1 < FR_TABLE) ? (f_last - fb + 1) : FR_TABLE;
    __syncthreads();
    build_fr_table(a, tab, fb, nf, tid);
    int64_t i = i_first + tid;
    int f = -1, j = 0;
    double u = 0.0, v = 0.0;
    if (i < i_end) {
      f = a.obs_fr[i];
      u = a.obs_u[i];
      v = a.obs_v[i];
      j = a.obs_corner[i];
    }
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < SWEEP_GROUP; ++g) {
      const int64_t in = i + SWEEP_TILE;
      int fn = -1, jn = 0;
      double un = 0.0, vn = 0.0;
      if (g + 1 < SWEEP_GROUP && in < i_end) {
        fn = a.obs_fr[in];
        un = a.obs_u[in];
        vn = a.obs_v[in];
        jn = a.obs_corner[in];
      }
      if (f >= fb && f < fb + nf) {
        TileSink sink{__builtin_amdgcn_make_buffer_rsrc(a.J + (tile0 + g) * (int64_t)(SWEEP_PLANES * SWEEP_TILE), 0,
                                                        SWEEP_PLANES * SWEEP_TILE * 8, 0x00020000),
                      tid * 8};
        const double X[3] = {(double)a.pts[3 * j], (double)a.pts[3 * j + 1], (double)a.pts[3 * j + 2]};
        double r[2];
        eval_prepared<true>(tab + (f - fb) * FRP_SIZE, X, u, v, r, sink);
        sink(42, r[0]);
        sink(43, r[1]);
      }
      i = in;
      f = fn;
      u = un;
      v = vn;
      j = jn;
    }
  }
}

void launch_jacobian_sweep(const SweepArgs& a, hipStream_t s) {
  if (a.n_obs <= 0) return;
  const int64_t per = (int64_t)SWEEP_GROUP * SWEEP_TILE;
  const int64_t blocks = (a.n_obs + per - 1) / per;
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

void launch_prep_vars(const SweepArgs& a, hipStream_t s);

// ---------------------------------------------------------------- cost sweep
// The FRs of a tile group are prepared into LDS in two steps: [R | t] of the three blocks and the
// camera parameters (4 lanes per FR, from the per-variable table), then one lane per FR composes
// them into ONE transform: the residual of a corner needs Xc = R X + t and the projection,
// 24 table values per FR instead of 48 (the broadcast LDS reads were the busiest unit here).
template <bool NODEN>
__global__ __launch_bounds__(SWEEP_BLOCK) void cost_sweep_kernel(SweepArgs a, double* partial) {
  __shared__ __attribute__((aligned(16))) double tab[FR_TABLE * FRP_SIZE];
  __shared__ __attribute__((aligned(16))) double ctab[FR_TABLE * FRR_SIZE];
  __shared__ double sh[4];
  const int tid = threadIdx.x;
  constexpr int64_t PER = (int64_t)COST_GROUP * SWEEP_TILE;
  const int64_t ngroups = (a.n_obs + PER - 1) / PER;
  double acc = 0.0;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t i_first = grp * PER;
    const int64_t i_end = (i_first + PER < a.n_obs) ? i_first + PER : a.n_obs;
    const int f_first = a.obs_fr[i_first], f_last = a.obs_fr[i_end - 1];
    for (int fb = f_first; fb <= f_last; fb += FR_TABLE) {
      const int nf = (f_last - fb + 1 < FR_TABLE) ? (f_last - fb + 1) : FR_TABLE;
      __syncthreads();
      build_fr_table_prepped(a, tab, fb, nf, tid);
      __syncthreads();
      if (tid < nf) {
        const double* e = tab + tid * FRP_SIZE;
        compose_fr(e + FRP_CAM, e + FRP_TIM, e + FRP_PAT, e + FRP_K, ctab + tid * FRR_SIZE);
      }
      __syncthreads();
      // A lane takes two adjacent observations per pair: they nearly always share the FR, and the
      // table values (broadcast LDS reads) then serve both.  Two pairs per step, all sixteen loads
      // requested together through clamped indices (no branch around a load): with one pair per
      // step the kernel's skeleton (loads, barriers, no arithmetic) ran at 0.46 ms on config D, the
      // four workgroups a CU holds had 49 KB in flight per memory round trip.
      auto residual_pair = [&](bool ok, bool two, int f0, int f1, const double* X0, const double* X1, double u0,
                               double v0, double u1, double v1) {
        const bool in0 = ok && f0 >= fb && f0 < fb + nf, in1 = ok && two && f1 >= fb && f1 < fb + nf;
        double r[2];
        if (in0 && in1 && f1 == f0) {
          const double* ent = ctab + (f0 - fb) * FRR_SIZE;
          eval_composed_residual<NODEN>(ent, X0, u0, v0, r);
          acc += r[0] * r[0] + r[1] * r[1];
          eval_composed_residual<NODEN>(ent, X1, u1, v1, r);
          acc += r[0] * r[0] + r[1] * r[1];
        } else {
          if (in0) {
            eval_composed_residual<NODEN>(ctab + (f0 - fb) * FRR_SIZE, X0, u0, v0, r);
            acc += r[0] * r[0] + r[1] * r[1];
          }
          if (in1) {
            eval_composed_residual<NODEN>(ctab + (f1 - fb) * FRR_SIZE, X1, u1, v1, r);
            acc += r[0] * r[0] + r[1] * r[1];
          }
        }
      };
      constexpr int NP = RP_COST_PAIRS;
      for (int64_t i = i_first + 2 * tid; i < i_end; i += 2 * NP * SWEEP_TILE) {
        bool ok[NP], two[NP];
        int f0[NP], f1[NP], j0[NP], j1[NP];
        double u0[NP], v0[NP], u1[NP], v1[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int64_t ip = i + 2 * p * SWEEP_TILE;
          ok[p] = ip < i_end;
          two[p] = ip + 1 < i_end;
          const int64_t x0 = ok[p] ? ip : i, x1 = two[p] ? ip + 1 : x0;
          f0[p] = a.obs_fr[x0]; f1[p] = a.obs_fr[x1];
          j0[p] = a.obs_corner[x0]; j1[p] = a.obs_corner[x1];
          u0[p] = a.obs_u[x0]; v0[p] = a.obs_v[x0];
          u1[p] = a.obs_u[x1]; v1[p] = a.obs_v[x1];
        }
        float xa[NP][3], xb[NP][3];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            xa[p][k] = a.pts[3 * j0[p] + k];
            xb[p][k] = a.pts[3 * j1[p] + k];
          }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const double X0[3] = {(double)xa[p][0], (double)xa[p][1], (double)xa[p][2]};
          const double X1[3] = {(double)xb[p][0], (double)xb[p][1], (double)xb[p][2]};
          residual_pair(ok[p], two[p], f0[p], f1[p], X0, X1, u0[p], v0[p], u1[p], v1[p]);
        }
      }
    }
  }
  const double tot = block_sum_256(acc, sh);
  if (tid == 0) partial[blockIdx.x] = tot;
}

// single block: out = scale * sum(partial[0..n)) in a fixed order (eight loads in flight per thread:
// the per-wave sums of the fused kernel are 25 000 values on config D).  Two more things can ride
// along instead of going through launches of their own in front of the iteration's read-back:
// flag != null: the iteration's failure flag joins the scalars (flag_to_scalar_kernel);
// seg_partial != null: seg_scalars[k] = sum over the segments of seg_partial[.][k], k < 3, in
// segment order (the back-substitution's backsub_sum_kernel).
constexpr int SUMP_THREADS = 1024;
// sum over the workgroup's SUMP_THREADS threads in a fixed order (wave sums, then the 16 waves in order)
__device__ __forceinline__ double block_sum_sump(double v, double* sh /*16*/) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double tot = 0.0;
  if (threadIdx.x == 0)
    for (int k = 0; k < SUMP_THREADS / 64; ++k) tot += sh[k];
  __syncthreads();
  return tot;
}
__global__ __launch_bounds__(SUMP_THREADS) void sum_partials_kernel(const double* __restrict__ partial, int n, double scale,
                                                                    double* __restrict__ out, int* flag, double* flag_out,
                                                                    const double* __restrict__ seg_partial, int n_seg,
                                                                    double* __restrict__ seg_scalars) {
  __shared__ double sh[SUMP_THREADS / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += SUMP_THREADS * 8) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = i + SUMP_THREADS * k;
      v[k] = partial[idx < n ? idx : i];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += (i + SUMP_THREADS * k < n) ? v[k] : 0.0;
  }
  // the segments' partial scalars are requested before the first reduction's barriers
  double a3[3] = {0.0, 0.0, 0.0};
  if (seg_partial) {
    for (int i0 = threadIdx.x; i0 < n_seg; i0 += SUMP_THREADS * 4) {
      double v[4][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + SUMP_THREADS * q;
        const double* p = seg_partial + (int64_t)(i < n_seg ? i : i0) * SEGP;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[q][k] = p[k];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + SUMP_THREADS * q < n_seg)
#pragma unroll
          for (int k = 0; k < 3; ++k) a3[k] += v[q][k];
    }
  }
  const double tot = block_sum_sump(acc, sh);
  if (threadIdx.x == 0) {
    *out = scale * tot;
    if (flag) {
      *flag_out = (double)*flag;
      *flag = 0;
    }
  }
  if (seg_partial) {
    for (int k = 0; k < 3; ++k) {
      const double t = block_sum_sump(a3[k], sh);
      if (threadIdx.x == 0) seg_scalars[k] = t;
    }
  }
}

void launch_sum_partials(const double* partial, int n, double scale, double* out, hipStream_t s, int* flag,
                         double* flag_out, const double* seg_partial, int n_seg, double* seg_scalars) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(SUMP_THREADS), 0, s, partial, n, scale, out, flag, flag_out, seg_partial,
                     n_seg, seg_scalars);
}

void launch_cost_sweep(const SweepArgs& a, double* partial, double* cost_out, hipStream_t s) {
  const int64_t per = (int64_t)COST_GROUP * SWEEP_TILE;
  int64_t blocks = (a.n_obs + per - 1) / per;
  if (blocks > SWEEP_COST_GRID) blocks = SWEEP_COST_GRID;
  if (blocks < 1) blocks = 1;
  launch_prep_vars(a, s);
  if (a.no_denominator)
    hipLaunchKernelGGL(cost_sweep_kernel<true>, dim3((unsigned)blocks), dim3(SWEEP_BLOCK), 0, s, a, partial);
  else
    hipLaunchKernelGGL(cost_sweep_kernel<false>, dim3((unsigned)blocks), dim3(SWEEP_BLOCK), 0, s, a, partial);
  launch_sum_partials(partial, (int)blocks, 0.5, cost_out, s);
}

// ---------------------------------------------------------------- per-FR reprojection error
// Reporting (multicamera.cpp:1860-1891): one wave per FR, the FR prepared once into LDS.
__global__ __launch_bounds__(64) void fr_error_kernel(SweepArgs a, int n_fr, const int64_t* __restrict__ fr_obs_begin,
                                                      double* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) double tab[FRP_SIZE];
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f >= n_fr) return;
  build_fr_table(a, tab, f, 1, lane);
  __syncthreads();
  const int64_t b = fr_obs_begin[f], e = fr_obs_begin[f + 1];
  double acc = 0.0;
  NullSink none;
  for (int64_t i = b + lane; i < e; i += 64) {
    const int j = a.obs_corner[i];
    const double X[3] = {(double)a.pts[3 * j], (double)a.pts[3 * j + 1], (double)a.pts[3 * j + 2]};
    double r[2];
    eval_prepared<false>(tab, X, a.obs_u[i], a.obs_v[i], r, none);
    acc += r[0] * r[0] + r[1] * r[1];
  }
  acc = wave_sum(acc);
  if (lane == 0) out[f] = acc / (double)(e - b);
}
void launch_fr_errors(const SweepArgs& a, int n_fr, const int64_t* fr_obs_begin, double* out, hipStream_t s) {
  if (n_fr > 0) hipLaunchKernelGGL(fr_error_kernel, dim3(n_fr), dim3(64), 0, s, a, n_fr, fr_obs_begin, out);
}

// ---------------------------------------------------------------- block accumulation
// Every wave owns wave_obs consecutive observations (rp::acc_wave_obs) and computes, per part (FR x wave
// range), the Gram blocks of [c~ | p | t] (c~ = 7 camera columns + residual, p / t = 7
// pattern / time columns; rows = the part's 2*n residual rows) on the fp64 matrix cores.
// Two 16x16 tiles per k-step cover every block that is needed downstream:
//     T1 += [c~ p]^T [c~ t]   ->  c~c~ (J_c^T J_c, J_c^T r, r^T r), p c~, c~ t   (record A, per part)
//     T2 += [p  t]^T [p  t]   ->  p p, t p, t t                                 (record B, per piece)
// v_mfma_f64_16x16x4_f64: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15]; a k-step
// takes 4 rows = 4 observations of one residual row (one per lane group g = lane>>4).
// Lane column c = lane&15 needs two streams: S1 = column c of [c~ p], S2 = column c of
// [p t]; then  T1: A = S1, B = (c < 8 ? S1 : S2)   T2: A = B = S2.
// The operands always come out of an LDS slab [plane][observation]:
// (accumulate_kernel: the slab is filled from the Jacobian tiles in HBM.)
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int acc_plane_s1(int c, int rr) {  // [c0..c6 r p0..p6 0]
  if (c < 7) return 21 * rr + c;
  if (c == 7) return 42 + rr;
  if (c < 15) return 21 * rr + 7 + (c - 8);
  return -1;
}
__device__ __forceinline__ int acc_plane_s2(int c, int rr) {  // [p0..p6 0 t0..t6 0]
  if (c < 7) return 21 * rr + 7 + c;
  if (c >= 8 && c < 15) return 21 * rr + 14 + (c - 8);
  return -1;
}

// What a lane needs to know about its role in the two tiles.
struct GramLane {
  int c, g;
  int pls[4];    // slab rows of [S1 row0, S1 row1, S2 row0, S2 row1] (0 where the stream does not exist)
  int vm1, vm2;  // 15 if S1 / S2 exists for this lane, else 0
  bool low;
  __device__ __forceinline__ explicit GramLane(int lane) {
    c = lane & 15;
    g = lane >> 4;
    const int p0 = acc_plane_s1(c, 0), p1 = acc_plane_s1(c, 1), p2 = acc_plane_s2(c, 0), p3 = acc_plane_s2(c, 1);
    pls[0] = p0 < 0 ? 0 : p0;
    pls[1] = p1 < 0 ? 0 : p1;
    pls[2] = p2 < 0 ? 0 : p2;
    pls[3] = p3 < 0 ? 0 : p3;
    vm1 = p0 >= 0 ? 15 : 0;
    vm2 = p2 >= 0 ? 15 : 0;
    low = c < 8;
  }
};

// eight k-steps over 16 observations whose columns start at slab + col0 (row stride ld):
// residual row 0 then row 1, observation q of every lane; mask bit q = observation
// 4g+q of the group belongs to the part
template <bool MFMA>
__device__ __forceinline__ void gram16(const GramLane& L, const double* slab, int ld, int col0, int mask, f64x4& t1,
                                       f64x4& t2) {
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const double* r1 = slab + L.pls[rr] * ld + col0 + 4 * L.g;
    const double* r2 = slab + L.pls[2 + rr] * ld + col0 + 4 * L.g;
    const f64x2 a_lo = *reinterpret_cast<const f64x2*>(r1), a_hi = *reinterpret_cast<const f64x2*>(r1 + 2);
    const f64x2 b_lo = *reinterpret_cast<const f64x2*>(r2), b_hi = *reinterpret_cast<const f64x2*>(r2 + 2);
    const double m1[4] = {a_lo.x, a_lo.y, a_hi.x, a_hi.y}, m2[4] = {b_lo.x, b_lo.y, b_hi.x, b_hi.y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double s1 = ((mask & L.vm1) >> q) & 1 ? m1[q] : 0.0;
      const double s2 = ((mask & L.vm2) >> q) & 1 ? m2[q] : 0.0;
      if (MFMA) {
        t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s1, L.low ? s1 : s2, t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(s2, s2, t2, 0, 0, 0);
      } else {
        t1[q] += s1 * (L.low ? s1 : s2);
        t2[q] += s2 * s2;
      }
    }
  }
}

// D: reg r of lane -> row i = g + 4r, col j = c.
//   T1 rows: i<8 c~_i, 8..14 p_{i-8};  T1 cols: j<8 c~_j, 8..14 t_{j-8}
//   T2 rows/cols: <7 p, 8..14 t
// record A: [c~c~ | p c~ | t c~]; the (p,t) quadrant of T1 is dropped
__device__ __forceinline__ void scatter_record_a(const GramLane& L, double* out, const f64x4& t1) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = L.g + 4 * r, j = L.c;
    if (j < 8) {
      if (i < 8) out[0 * 64 + i * 8 + j] = t1[r];        // (c~,c~)[i][j]
      else out[1 * 64 + (i - 8) * 8 + j] = t1[r];        // (p,c~)[i-8][j]   (row 15: zeros)
    } else if (i < 8) {
      out[2 * 64 + (j - 8) * 8 + i] = t1[r];             // (t,c~)[j-8][i]   (col 15: zeros)
    }
  }
}
// record B: [p p | t p | t t], and in the padding column of t t the piece's J_t^T r:
// (t,t)[a][7] = gt, held by the lanes (c = 8 + a, g = 3) that own (t,c~)[a][7] of T1
__device__ __forceinline__ void scatter_record_b(const GramLane& L, double* out, const f64x4& t2, double gt) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = L.g + 4 * r, j = L.c;
    if (i < 8) {
      if (j < 8) out[0 * 64 + i * 8 + j] = t2[r];        // (p,p)[i][j]      (row/col 7: zeros)
      else out[1 * 64 + (j - 8) * 8 + i] = t2[r];        // (t,p)[j-8][i]
    } else if (j >= 8) {
      out[2 * 64 + (i - 8) * 8 + (j - 8)] = t2[r];       // (t,t)[i-8][j-8]   (row/col 7: zeros)
    }
  }
  // after the zeros of column 7 (one wave: the stores stay in program order)
  if (L.c >= 8 && L.c < 15 && L.g == 3) out[2 * 64 + (L.c - 8) * 8 + 7] = gt;
}

constexpr int ACC_STAGE_ROWS = 48, ACC_STAGE_LD = 18;  // per-wave LDS slab: plane rows of 16 observations (+2 pad)

// VARIANT is 0 in the library; tools/acc_bench.hip instantiates 1 (no record writes),
// 2 (no MFMA) and 4 (no refetch) to see which part bounds the kernel.
#ifndef RP_ACC_MIN_BLOCKS
#define RP_ACC_MIN_BLOCKS 4
#endif
template <int VARIANT>
__global__ __launch_bounds__(256, RP_ACC_MIN_BLOCKS) void accumulate_kernel(
    int64_t n_obs, int wave_obs, const int32_t* __restrict__ obs_fr, const int64_t* __restrict__ fr_obs_begin,
    const int32_t* __restrict__ wave_part0, const int32_t* __restrict__ part_piece, const double* __restrict__ J,
    double* __restrict__ recA, double* __restrict__ recB) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  const int64_t i0 = w * wave_obs;
  if (i0 >= n_obs) return;
  const int64_t i1 = (i0 + wave_obs < n_obs) ? i0 + wave_obs : n_obs;
  const GramLane L(lane);
  const int g = L.g;
  f64x4 t1 = {0.0, 0.0, 0.0, 0.0}, t2 = t1;
  // Groups of 16 observations aligned to 16 (= one 128-B line per plane).  The MFMA
  // operand layout puts 16 different planes into every quarter-wave, i.e. 16 cache lines
  // per 16 lanes if loaded directly.  So the group's 44 plane lines are fetched line-wise
  // (8 lanes x 16 B per line, 8 planes per instruction, whole lines only) into registers,
  // staged through a per-wave LDS slab, and read back in operand layout.  Lanes at a part
  // boundary take their whole quad (the tile is allocated whole) and mask at use.
  // All branch-free, so that the waits are counted and the group ahead stays in flight.
  __shared__ __attribute__((aligned(16))) double stage_all[4][ACC_STAGE_ROWS * ACC_STAGE_LD];
  double* stage = stage_all[wave];
  const int lrow = lane >> 3, lpiece = (lane & 7) * 2;
  f64x2 raw[6];  // the group in flight, line layout
  auto fetch = [&](int64_t o16) {
    const double* base = J + (o16 >> 8) * (int64_t)(SWEEP_PLANES * SWEEP_TILE) + (int)(o16 & 255) + lpiece;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int plane = 8 * i + lrow;
      plane = plane < SWEEP_PLANES ? plane : SWEEP_PLANES - 1;  // rows 44..47 of the slab are never read
      raw[i] = *reinterpret_cast<const f64x2*>(base + plane * SWEEP_TILE);
    }
  };
  auto restage = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      *reinterpret_cast<f64x2*>(stage + (8 * i + lrow) * ACC_STAGE_LD + lpiece) = raw[i];
    // one wave, in-order LDS: the slab accesses may alias, so they stay in program order
    __builtin_amdgcn_wave_barrier();
  };
  // records are scattered into the (idle) slab in record order and written as whole lines
  static_assert(ACC_STAGE_ROWS * ACC_STAGE_LD >= REC_A && REC_A == 192 && REC_B == 192, "record staging");
  auto write_record = [&](double* dst_rec) {
    __builtin_amdgcn_wave_barrier();
    f64x2* dst = reinterpret_cast<f64x2*>(dst_rec);
    const f64x2 v0 = *reinterpret_cast<const f64x2*>(stage + 2 * lane);
    if ((VARIANT & 1) && v0.x != 1.2345e-300) {
      __builtin_amdgcn_wave_barrier();
      return;
    }
    dst[lane] = v0;
    if (lane < 32) dst[64 + lane] = *reinterpret_cast<const f64x2*>(stage + 2 * (64 + lane));
    __builtin_amdgcn_wave_barrier();
  };

  // flat walk over (part, group) with the next group's lines in flight behind the MFMAs
  double gt_piece = 0.0;
  int part = wave_part0[w];
  int f = obs_fr[i0];
  int64_t b = i0, e = fr_obs_begin[f + 1];
  if (e > i1) e = i1;
  int64_t o16 = b & ~(int64_t)15;
  fetch(o16);
  for (;;) {
    restage();
    const int64_t o = o16 + 4 * g;
    int mask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) mask |= (o + q >= b && o + q < e) ? (1 << q) : 0;
    const bool part_ends = o16 + 16 >= e;
    const bool last = part_ends && e >= i1;
    int64_t nb = b, ne = e, no = o16 + 16;
    if (part_ends && !last) {
      nb = e;
      ne = fr_obs_begin[f + 2];
      if (ne > i1) ne = i1;
      no = nb & ~(int64_t)15;
    }
    // past the last part the prefetch re-reads the current group (valid memory, unused)
    if (!(VARIANT & 4)) fetch(last ? o16 : no);
    gram16<!(VARIANT & 2)>(L, stage, ACC_STAGE_LD, 0, mask, t1, t2);
    __builtin_amdgcn_wave_barrier();
    if (part_ends) {
      scatter_record_a(L, stage, t1);
      write_record(recA + (int64_t)part * REC_A);
      gt_piece += t1[1];  // meaningful in the lanes (c = 8 + a, g = 3): (t,c~)[a][7]
      t1 = (f64x4){0.0, 0.0, 0.0, 0.0};
      const int piece = part_piece[part];
      if (last || part_piece[part + 1] != piece) {
        scatter_record_b(L, stage, t2, gt_piece);
        write_record(recB + (int64_t)piece * REC_B);
        t2 = (f64x4){0.0, 0.0, 0.0, 0.0};
        gt_piece = 0.0;
      }
      if (last) break;
      ++part;
      ++f;
    }
    b = nb;
    e = ne;
    o16 = no;
  }
}

// Observations per wave of the accumulation kernels: about 12288 waves (4 rounds of the 3072
// fused-kernel wave slots of the chip: 3 waves per SIMD), at most 4096 observations (a part
// record per extra wave boundary) and at least 256; a multiple of 64.
int acc_wave_obs(int64_t n_obs) {
  int64_t l = (n_obs / 12288 + 63) / 64 * 64;
  if (l < 256) l = 256;
  if (l > 4096) l = 4096;
  return (int)l;
}

void launch_accumulate(int64_t n_obs, int wave_obs, const int32_t* obs_fr, const int64_t* fr_obs_begin,
                       const int32_t* wave_part0, const int32_t* part_piece, const double* J, double* recA,
                       double* recB, hipStream_t s) {
  if (n_obs <= 0) return;
  const int64_t waves = (n_obs + wave_obs - 1) / wave_obs;
  hipLaunchKernelGGL(accumulate_kernel<0>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, n_obs, wave_obs, obs_fr,
                     fr_obs_begin, wave_part0, part_piece, J, recA, recB);
}

// ---------------------------------------------------------------- fused sweep + accumulation
// One wave per workgroup.  Per slab of 64 consecutive observations the wave
//   1. evaluates residual + Jacobian columns, lane = observation, from an LDS ring of per-FR
//      table entries straight into an LDS slab,
//   2. runs the Gram steps of every part that intersects the slab out of the slab;
// the accumulators stay in registers across slabs until the part ends.  Nothing but the
// 24 B/observation inputs, the per-FR table and the records touches HBM.
//
// The records hold Gram blocks of the 16 rotation-vector columns of a residual row
// (rp_model.hpp: UW_*),
//     U = [ wc x a (3) | a (3) | r | wp x b2 (3) | b2 (3) | wt x b1 (3) ]
// (a = row of d(u,v)/dXc, b1 = a Rc, b2 = b1 Rt, w* = the rotated point of each factor).  The
// 22 columns of [J_c | r | J_p | J_t] are U times a matrix that depends only on the camera /
// pattern / time blocks, and every downstream sum of Gram blocks runs over a fixed camera,
// pattern or time, so the records stay in U space and the gathers convert the SUMS (gather
// kernels below).
//   record A (per part)   G[0..15][0..7]: [16][8] doubles (column 7 unused)
//   record B (per piece)  sum over the piece's parts of G[7..15][6..15]: [9][10] doubles
// The kernels below never evaluate U itself: they work on the 7 chain columns V (U = V C).
constexpr int FUSED_T = 4;    // FR table ring entries

// [R(q) | t] of every 7-parameter block, once per evaluation instead of once per FR inside the
// fused kernel (1.28 M FRs share 10 k variables on config D); slot n_vars is the identity.
__global__ __launch_bounds__(256) void prep_vars_kernel(const double* __restrict__ x, int n_vars, double* __restrict__ prep) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v > n_vars) return;
  double blk[PREP_SIZE];
  prep_block(v < n_vars ? x + 7 * (int64_t)v : nullptr, blk);
#pragma unroll
  for (int k = 0; k < 12; ++k) prep[12 * (int64_t)v + k] = blk[k];
}

void launch_prep_vars(const SweepArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(prep_vars_kernel, dim3((a.n_vars + 256) / 256), dim3(256), 0, s, a.x, a.n_vars, a.prep_out);
}

// ---------------------------------------------------------------- fused kernel, chain columns
// The default path (camera parameters fixed).  rp_model.hpp "chain columns": per observation
// row only the 7 columns V = [wc x a | a | r] are evaluated, from the FR's composed transform,
// and the Gram of a part is the 7 x 7 matrix G7 = V^T V; the 16 x 16 Gram of the
// rotation-vector columns that the records hold is C^T G7 C with the FR-constant 8 x 16 matrix
// C, formed once per part.
//   * end of a part: with G7 (symmetric) in the accumulator layout of a 16x16x4 tile, P = G7 C
//     and G = C^T P are 2 + 2 MFMAs: G7 IS the A operand of the first product, P in accumulator
//     layout the B operand of the second, and the lane's two entries of C serve as B of the
//     first and A of the second.
//   * FR table: ring of 4 entries of 64 doubles straight from the per-FR table in HBM
//     (prep_frs_kernel: composed transform, camera parameters, the four 3 x 3 blocks of C), one
//     double per lane and FR, fetched a slab ahead; no dependent index -> parameter loads.
struct FrRing7 {
  static constexpr int T = FUSED_T;
  const double* frtab;
  double* tab;  // LDS, T * FRC_SIZE
  int lane, n_fr;
  int v_lo = 0, v_hi = -1;  // valid FRs (wave-uniform)
  int p2 = -1, n2 = 0;      // FRs p2 .. p2 + n2 - 1: entries in flight in reg0 / reg1
  double reg0 = 0.0, reg1 = 0.0;
  __device__ __forceinline__ FrRing7(const double* frtab_, double* table, int lane_, int n_fr_)
      : frtab(frtab_), tab(table), lane(lane_), n_fr(n_fr_) {}
  __device__ __forceinline__ const double* entry(int fr) const { return tab + (fr & (T - 1)) * FRC_SIZE; }
  __device__ __forceinline__ bool holds(int fr) const { return fr >= v_lo && fr <= v_hi; }
  // entries of FRs [fb, fb + nf), nf <= T: fetch on the spot what the ring lacks
  __device__ __forceinline__ void ensure(int fb, int nf) {
    if (fb < v_lo || fb + nf - 1 > v_hi) {
      for (int k = 0; k < nf; ++k) {
        const int fr = fb + k;
        if (fr < v_lo || fr > v_hi) tab[(fr & (T - 1)) * FRC_SIZE + lane] = frtab[(int64_t)fr * FRC_SIZE + lane];
      }
      if (fb > v_hi + 1 || fb < v_lo) v_lo = fb;
      v_hi = fb + nf - 1;
      if (v_lo < v_hi - T + 1) v_lo = v_hi - T + 1;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // once per slab: the (up to two) entries requested a slab ago go into the ring, the next ones
  // are requested; FR fr overwrites fr - T, which must stay behind keep_from
  __device__ __forceinline__ void advance(int keep_from) {
    if (p2 >= 0) {
      if (p2 == v_hi + 1) {
        if (p2 - T < keep_from) {
          tab[(p2 & (T - 1)) * FRC_SIZE + lane] = reg0;
          v_hi = p2;
          if (n2 > 1 && p2 + 1 - T < keep_from) {
            tab[((p2 + 1) & (T - 1)) * FRC_SIZE + lane] = reg1;
            v_hi = p2 + 1;
          }
        }
        if (v_lo < v_hi - T + 1) v_lo = v_hi - T + 1;
      }
      p2 = -1;
    }
    const int target = v_hi + 1;
    n2 = 0;
    if (target < n_fr && target - T < keep_from) {
      reg0 = frtab[(int64_t)target * FRC_SIZE + lane];
      n2 = 1;
      if (target + 1 < n_fr && target + 1 - T < keep_from) {
        reg1 = frtab[(int64_t)(target + 1) * FRC_SIZE + lane];
        n2 = 2;
      }
      p2 = target;
    }
  }
};

// ---- chain columns, Gram on the quad-block fp64 MFMA (v_mfma_f64_4x4x4_4b_f64)
// A 16x16x4 tile fed with both residual rows of a 7-column V spends 64 cycles on four
// observations and uses two of its four quadrants (the round-2 form of this kernel).  The 4x4x4
// instruction with four independent blocks (A[b][i][k] in lane
// i + 4b + 16k, B[b][k][j] in lane j + 4b + 16k, D[b][i][j] in lane j + 4b + 16i;
// tools/mfma44_probe.hip) issues about four times as often, and with the 8 columns of V split
// into two chunks of four, V = [V0 | V1], three instructions per residual row take SIXTEEN
// observations: lane (x, b, k) feeds column x of a chunk for observation b + 4k of the group, the
// blocks accumulate four partial Grams of V0^T V0, V1^T V0 and V1^T V1 (the upper right chunk is
// the transpose of the lower left), summed over b at the end of the part.
//   * slab: per column c < 7 a plane of 16-byte units [obs] = (row 0, row 1), plane stride 68
//     units: the evaluation stores one unit per column (ds_write_b128, conflict free), a Gram
//     step loads two (chunk 0 / chunk 1: both rows of the lane's column), 4 x + b distinct mod 16
//     inside every quarter wave.  Column 7 is a single zero unit that its lanes (x = 3, chunk 1)
//     read for every group; masked lanes of a partial group read it too.
//   * end of a part: the three accumulators are summed over b by two DPP row rotations, the
//     upper right chunk comes through one ds_bpermute, and the lanes c < 8 of a row of 16 hold
//     G7[g][c] / G7[g + 4][c], the A operand of the part-end products.
constexpr int F7Q_PS = 68, F7Q_ZERO = 7 * F7Q_PS, F7Q_UNITS = F7Q_ZERO + 2;

struct SlabSink7Q {
  f64x2* col;      // the lane's unit of plane 0
  double row0[7];  // row 0 waits for row 1 of the same column (constant indices: stays in registers)
  __device__ __forceinline__ void operator()(int k, double v) {
    if (k < 8) row0[k] = v;
    else col[(k - 8) * F7Q_PS] = (f64x2){row0[k - 8], v};
  }
};

template <int CTRL>
__device__ __forceinline__ double row_dpp(double v) {
  const long long b = __double_as_longlong(v);
  // every lane of a row rotation has a source: no "old" value to set up (mov_dpp leaves it undefined)
  const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double lane_fetch(double v, int byte_lane) {  // value of lane byte_lane / 4
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(byte_lane, (int)(unsigned)b);
  const int hi = __builtin_amdgcn_ds_bpermute(byte_lane, (int)(unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// one group of 16 observations, both residual rows: u0 / u1 = the lane's units of chunk 0 / 1.
// Three accumulators, each used every third instruction (a 4x4x4 result is back well before).
__device__ __forceinline__ void gram7q(const f64x2* slab2, int u0, int u1, double (&q)[3]) {
  const f64x2 x0 = slab2[u0], x1 = slab2[u1];
  q[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(x0.x, x0.x, q[0], 0, 0, 0);
  q[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(x1.x, x0.x, q[1], 0, 0, 0);
  q[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(x1.x, x1.x, q[2], 0, 0, 0);
  q[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(x0.y, x0.y, q[0], 0, 0, 0);
  q[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(x1.y, x0.y, q[1], 0, 0, 0);
  q[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(x1.y, x1.y, q[2], 0, 0, 0);
}

template <bool NODEN>
__device__ __forceinline__ void fused7q_body(const SweepArgs& a, int wave_obs, int n_fr,
                                             const int64_t* __restrict__ fr_obs_begin,
                                             const int32_t* __restrict__ wave_part0,
                                             const int32_t* __restrict__ part_piece, double* __restrict__ recA,
                                             double* __restrict__ recB) {

  __shared__ f64x2 slab2[F7Q_UNITS];
  __shared__ __attribute__((aligned(16))) double tab[FUSED_T * FRC_SIZE];
  __shared__ double tab_late[FRC_SIZE];  // the entry of an FR that left the ring before its part ended
  const int lane = threadIdx.x;
  const int64_t w = blockIdx.x;
  const int64_t i0 = w * wave_obs;
  if (i0 >= a.n_obs) return;
  const int64_t i1 = (i0 + wave_obs < a.n_obs) ? i0 + wave_obs : a.n_obs;
  const int c = lane & 15, g = lane >> 4;
  // Gram lane (x, b, k): column x of a chunk, observation b + 4k of a group
  const int qx = lane & 3, qb = (lane >> 2) & 3;
  const int qo = qb + 4 * g;
  const int base0 = qx * F7Q_PS + qo;
  const int base1 = qx == 3 ? F7Q_ZERO : (4 + qx) * F7Q_PS + qo;
  const int scale1 = qx == 3 ? 0 : 1;
  const int fetch_t = 4 * (g + 4 * qb + 16 * qx);  // the lane that holds the transposed entry of the lower left chunk
  const int cofs0 = chain_map_offset(g, c), cofs1 = chain_map_offset(g + 4, c);
  double q[3] = {0.0, 0.0, 0.0};
  f64x4 bacc = {0.0, 0.0, 0.0, 0.0};  // G summed over the parts of the current piece
  double costacc = 0.0;               // lane 38 (c = 6, g = 2): G[6][6] = r^T r summed over the wave's parts
  if (lane < 2) slab2[F7Q_ZERO + lane] = (f64x2){0.0, 0.0};
  __builtin_amdgcn_wave_barrier();
  int part = wave_part0[w];
  // Vector memory returns in order and the record stores count with the loads, so a wait for
  // any load also waits for everything issued before it, and a wait right behind the Gram steps
  // waits for their record stores.  All loads are therefore issued AND awaited in ONE place per
  // slab, between the evaluation and the Gram steps (unconditional, through clamped indices): the
  // observation a slab ahead and its corner index two slabs ahead, so that the model point of the
  // next slab travels with them.  The youngest operation that wait can meet is a record store of
  // the slab before, an evaluation old.
  // Positions are kept relative to the wave's first observation, as 32-bit scalars (gfx950 has no
  // scalar 64-bit order compare: every int64 test in the walk below was a v_mov_b64 + v_cmp on
  // the vector unit); what lies further away than 2^30 either way is clamped there.
  auto rel = [&](int64_t pos) -> int {
    const uint64_t d = (uint64_t)(pos - i0) + (1u << 30);
    const uint32_t hi = (uint32_t)(d >> 32), lo = (uint32_t)d;
    return (hi | (lo >> 31)) == 0 ? (int)lo - (1 << 30) : (int32_t)hi < 0 ? -(1 << 30) : (1 << 30);
  };
  const int i1r = (int)(i1 - i0);
  const int lastr = rel(a.n_obs - 1);
  const int64_t last = a.n_obs - 1;
  int f, fn, jn2;
  double u, v, un, vn;
  float xf0, xf1, xf2, xn0, xn1, xn2;
  {
    const int64_t ic0 = i0 + lane < last ? i0 + lane : last;
    const int64_t ic1 = i0 + 64 + lane < last ? i0 + 64 + lane : last;
    const int64_t ic2 = i0 + 128 + lane < last ? i0 + 128 + lane : last;
    f = a.obs_fr[ic0];
    u = a.obs_u[ic0];
    v = a.obs_v[ic0];
    const int j0 = a.obs_corner[ic0], j1 = a.obs_corner[ic1];
    if (i0 + lane >= i1) f = -1;
    xf0 = a.pts[3 * j0];
    xf1 = a.pts[3 * j0 + 1];
    xf2 = a.pts[3 * j0 + 2];
    xn0 = a.pts[3 * j1];
    xn1 = a.pts[3 * j1 + 1];
    xn2 = a.pts[3 * j1 + 2];
    fn = a.obs_fr[ic1];
    un = a.obs_u[ic1];
    vn = a.obs_v[ic1];
    jn2 = a.obs_corner[ic2];
  }
  int ff = __builtin_amdgcn_readfirstlane(f);
  int fbeg = rel(fr_obs_begin[ff]), fend = rel(fr_obs_begin[ff + 1]);
  int fend2 = (ff + 2 <= n_fr) ? rel(fr_obs_begin[ff + 2]) : (1 << 30);
  FrRing7 ring(a.frtab, tab, lane, n_fr);

#pragma unroll 1
  for (int s0 = 0; s0 < i1r; s0 += 64) {
    const int s1 = (s0 + 64 < i1r) ? s0 + 64 : i1r;
    const int f_lo = __builtin_amdgcn_readfirstlane(f);
    const int f_hi = __builtin_amdgcn_readlane(f, s1 - s0 - 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int fb = f_lo; fb <= f_hi; fb += FUSED_T) {
      const int nf = (f_hi - fb + 1 < FUSED_T) ? (f_hi - fb + 1) : FUSED_T;
      ring.ensure(fb, nf);
      if (f >= fb && f < fb + nf) {
        const double X[3] = {(double)xf0, (double)xf1, (double)xf2};
        double r[2];
        SlabSink7Q sink{slab2 + lane, {}};
        eval_chain<NODEN>(ring.entry(f), X, u, v, r, sink);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- everything requested a slab ago is taken over; the requests of this slab go out
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    f = s0 + 64 + lane < i1r ? fn : -1;
    u = un;
    v = vn;
    xf0 = xn0;
    xf1 = xn1;
    xf2 = xn2;
    ring.advance(ff);  // FRs before the Gram walk's current one are done with their entries
    {
      // wave-uniform base (scalar) + a 32-bit lane offset, clamped to the last observation
      const int room = lastr - s0;  // >= 0
      const unsigned lim = (unsigned)(room < (1 << 20) ? room : (1 << 20));
      const int64_t sb = i0 + s0;
      const unsigned o2 = (unsigned)(128 + lane) < lim ? (unsigned)(128 + lane) : lim;
      const unsigned o3 = (unsigned)(192 + lane) < lim ? (unsigned)(192 + lane) : lim;
      const float* px = a.pts + 3 * jn2;
      xn0 = px[0];
      xn1 = px[1];
      xn2 = px[2];
      fn = (a.obs_fr + sb)[o2];
      un = (a.obs_u + sb)[o2];
      vn = (a.obs_v + sb)[o2];
      jn2 = (a.obs_corner + sb)[o3];
    }
    // ---- Gram steps of every part that intersects the slab
#pragma unroll 1
    for (;;) {
      if (fend > s0) {
        const int b = fbeg > s0 ? fbeg : s0, e = fend < s1 ? fend : s1;
        const int part_end = fend < i1r ? fend : i1r;
        const int b_rel = b - s0, e_rel = e - s0;
        auto partial_group = [&](int g16) {
          const int o = g16 + qo;
          const bool on = o >= b_rel && o < e_rel;
          gram7q(slab2, on ? base0 + g16 : F7Q_ZERO, on ? base1 + scale1 * g16 : F7Q_ZERO, q);
        };
        int g16 = b_rel & ~15;
        if (g16 < b_rel || g16 + 16 > e_rel) {
          partial_group(g16);
          g16 += 16;
        }
#pragma unroll 1
        for (; g16 + 16 <= e_rel; g16 += 16)
          gram7q(slab2, base0 + g16, base1 + scale1 * g16, q);
        if (g16 < e_rel) partial_group(g16);
        if (e == part_end) {
          // ---- the part is complete: G7 -> G = C^T G7 C, then out
          double c0 = cofs0 == -2 ? 1.0 : 0.0, c1 = cofs1 == -2 ? 1.0 : 0.0;
          const double* ent = ring.entry(ff);
          if (!ring.holds(ff)) {  // a slab with more than FUSED_T FRs has overwritten the early ones
            tab_late[lane] = a.frtab[(int64_t)ff * FRC_SIZE + lane];
            __builtin_amdgcn_wave_barrier();
            ent = tab_late;
          }
          if (cofs0 >= 0) c0 = ent[cofs0];
          if (cofs1 >= 0) c1 = ent[cofs1];
          const int piece = part_piece[part];
          const bool piece_ends = e >= i1r || part_piece[part + 1] != piece;
          // sums over the four blocks: lane (j, b, i) = S[i][j] for every b
          double s00 = q[0], s10 = q[1], s11 = q[2];
          q[0] = q[1] = q[2] = 0.0;
          s00 += row_dpp<0x124>(s00);
          s10 += row_dpp<0x124>(s10);
          s11 += row_dpp<0x124>(s11);
          s00 += row_dpp<0x128>(s00);
          s10 += row_dpp<0x128>(s10);
          s11 += row_dpp<0x128>(s11);
          const double s01 = lane_fetch(s10, fetch_t);  // G7[g][4 + x] = S10[x][g]
          // lanes c < 8: G7[g][c] and G7[g + 4][c] (the others feed rows of P that nothing reads)
          const double t0 = qb == 0 ? s00 : s01, t1 = qb == 0 ? s10 : s11;
          f64x4 P = {0.0, 0.0, 0.0, 0.0};
          f64x4 G = {0.0, 0.0, 0.0, 0.0};
          P = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, c0, P, 0, 0, 0);
          P = __builtin_amdgcn_mfma_f64_16x16x4f64(t1, c1, P, 0, 0, 0);
          G = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, P[0], G, 0, 0, 0);
          G = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, P[1], G, 0, 0, 0);
          if (c < 8) {  // record A: G[0..15][0..7]; 32 consecutive doubles per store
            double* out = recA + (int64_t)part * REC_AW + g * 8 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[32 * r] = G[r];
          }
          bacc += G;
          costacc += G[1];
          if (piece_ends) {
            if (c >= 6) {  // record B: rows 7..15, columns 6..15 of the piece's sum
              double* out = recB + (int64_t)piece * REC_BW + (c - 6);
#pragma unroll
              for (int r = 1; r < 4; ++r) {
                const int row = g + 4 * r;
                if (row >= 7) out[(row - 7) * 10] = bacc[r];
              }
            }
            bacc = (f64x4){0.0, 0.0, 0.0, 0.0};
            asm volatile("" ::: "memory");  // keeps this a (wave-uniform) branch: as selects it is 8 v_cndmask per part
          }
          ++part;
        }
        if (fend >= s1) break;
      }
      ++ff;
      fbeg = fend;
      fend = fend2;
      fend2 = (ff + 2 <= n_fr) ? rel(fr_obs_begin[ff + 2]) : (1 << 30);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the wave's share of the cost (sum r^2; launch_sum_partials adds the waves in a fixed order)
  if (lane == 38) a.wave_cost[w] = costacc;
}

// Four waves per SIMD need 128 registers: the form without the rational model's denominator
// fits, the one with it would spill the ring's requests (and wait for them on the spot), so it
// runs at three waves per SIMD.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void fused7q_kernel(
    SweepArgs a, int wave_obs, int n_fr, const int64_t* __restrict__ fr_obs_begin, const int32_t* __restrict__ wave_part0,
    const int32_t* __restrict__ part_piece, double* __restrict__ recA, double* __restrict__ recB) {
  fused7q_body<true>(a, wave_obs, n_fr, fr_obs_begin, wave_part0, part_piece, recA, recB);
}
__global__ __launch_bounds__(64) void fused7q_den_kernel(SweepArgs a, int wave_obs, int n_fr,
                                                         const int64_t* __restrict__ fr_obs_begin,
                                                         const int32_t* __restrict__ wave_part0,
                                                         const int32_t* __restrict__ part_piece,
                                                         double* __restrict__ recA, double* __restrict__ recB) {
  fused7q_body<false>(a, wave_obs, n_fr, fr_obs_begin, wave_part0, part_piece, recA, recB);
}

// ---- the same with the camera parameters refined (NK = 12): the slab carries their 12 columns as
// planes 7..18 (three more chunks K0 K1 K2), a group of 16 observations takes 15 chunk products per
// residual row (V^T V 3, K^T V 6, K^T K 6 of 9: the rest are transposes) and the part end adds
// K^T U = (K^T V) C by two more 16x16x4 tiles; record K = [12][32] doubles, K x [U (16) | K (12) | 0].
constexpr int F7QK_ZERO = (7 + KP) * F7Q_PS, F7QK_UNITS = F7QK_ZERO + 2;

struct SlabSink7QK {
  f64x2* col;
  double row0[7];
  double krow0[KP];
  __device__ __forceinline__ void operator()(int k, double v) {
    if (k < 8) row0[k] = v;
    else if (k < 16) col[(k - 8) * F7Q_PS] = (f64x2){row0[k - 8], v};
    else if (k < 16 + KP) krow0[k - 16] = v;
    else col[(7 + k - 16 - KP) * F7Q_PS] = (f64x2){krow0[k - 16 - KP], v};
  }
};

__device__ __forceinline__ void gram7qk(const f64x2* slab2, int u0, int u1, int uk, int kstep, double (&q)[3],
                                        double (&kv)[6], double (&kk)[6]) {
  const f64x2 x0 = slab2[u0], x1 = slab2[u1];
  const f64x2 y0 = slab2[uk], y1 = slab2[uk + kstep], y2 = slab2[uk + 2 * kstep];
#define RP_QM(A, B, ACC) ACC = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, ACC, 0, 0, 0)
  RP_QM(x0.x, x0.x, q[0]);  RP_QM(x1.x, x0.x, q[1]);  RP_QM(x1.x, x1.x, q[2]);
  RP_QM(y0.x, x0.x, kv[0]); RP_QM(y0.x, x1.x, kv[1]); RP_QM(y1.x, x0.x, kv[2]);
  RP_QM(y1.x, x1.x, kv[3]); RP_QM(y2.x, x0.x, kv[4]); RP_QM(y2.x, x1.x, kv[5]);
  RP_QM(y0.x, y0.x, kk[0]); RP_QM(y1.x, y0.x, kk[1]); RP_QM(y1.x, y1.x, kk[2]);
  RP_QM(y2.x, y0.x, kk[3]); RP_QM(y2.x, y1.x, kk[4]); RP_QM(y2.x, y2.x, kk[5]);
  RP_QM(x0.y, x0.y, q[0]);  RP_QM(x1.y, x0.y, q[1]);  RP_QM(x1.y, x1.y, q[2]);
  RP_QM(y0.y, x0.y, kv[0]); RP_QM(y0.y, x1.y, kv[1]); RP_QM(y1.y, x0.y, kv[2]);
  RP_QM(y1.y, x1.y, kv[3]); RP_QM(y2.y, x0.y, kv[4]); RP_QM(y2.y, x1.y, kv[5]);
  RP_QM(y0.y, y0.y, kk[0]); RP_QM(y1.y, y0.y, kk[1]); RP_QM(y1.y, y1.y, kk[2]);
  RP_QM(y2.y, y0.y, kk[3]); RP_QM(y2.y, y1.y, kk[4]); RP_QM(y2.y, y2.y, kk[5]);
#undef RP_QM
}

template <bool NODEN>
__device__ __forceinline__ void fused7qk_body(const SweepArgs& a, int wave_obs, int n_fr,
                                             const int64_t* __restrict__ fr_obs_begin,
                                             const int32_t* __restrict__ wave_part0,
                                             const int32_t* __restrict__ part_piece, double* __restrict__ recA,
                                             double* __restrict__ recB) {

  __shared__ f64x2 slab2[F7QK_UNITS];
  __shared__ __attribute__((aligned(16))) double tab[FUSED_T * FRC_SIZE];
  __shared__ double tab_late[FRC_SIZE];  // the entry of an FR that left the ring before its part ended
  const int lane = threadIdx.x;
  const int64_t w = blockIdx.x;
  const int64_t i0 = w * wave_obs;
  if (i0 >= a.n_obs) return;
  const int64_t i1 = (i0 + wave_obs < a.n_obs) ? i0 + wave_obs : a.n_obs;
  const int c = lane & 15, g = lane >> 4;
  // Gram lane (x, b, k): column x of a chunk, observation b + 4k of a group
  const int qx = lane & 3, qb = (lane >> 2) & 3;
  const int qo = qb + 4 * g;
  const int base0 = qx * F7Q_PS + qo;
  const int base1 = qx == 3 ? F7QK_ZERO : (4 + qx) * F7Q_PS + qo;
  const int basek = (7 + qx) * F7Q_PS + qo;  // camera-parameter chunk a: + 4 a planes
  const int scale1 = qx == 3 ? 0 : 1;
  const int fetch_t = 4 * (g + 4 * qb + 16 * qx);  // the lane that holds the transposed entry of the lower left chunk
  const int cofs0 = chain_map_offset(g, c), cofs1 = chain_map_offset(g + 4, c);
  double q[3] = {0.0, 0.0, 0.0};
  double kv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // K_a^T V_h at 2 a + h
  double kk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // K_a^T K_b, a >= b, at a (a + 1) / 2 + b
  f64x4 bacc = {0.0, 0.0, 0.0, 0.0};  // G summed over the parts of the current piece
  double costacc = 0.0;               // lane 38 (c = 6, g = 2): G[6][6] = r^T r summed over the wave's parts
  if (lane < 2) slab2[F7QK_ZERO + lane] = (f64x2){0.0, 0.0};
  __builtin_amdgcn_wave_barrier();
  int part = wave_part0[w];
  // Vector memory returns in order and the record stores count with the loads, so a wait for
  // any load also waits for everything issued before it, and a wait right behind the Gram steps
  // waits for their record stores.  All loads are therefore issued AND awaited in ONE place per
  // slab, between the evaluation and the Gram steps (unconditional, through clamped indices): the
  // observation a slab ahead and its corner index two slabs ahead, so that the model point of the
  // next slab travels with them.  The youngest operation that wait can meet is a record store of
  // the slab before, an evaluation old.
  // Positions are kept relative to the wave's first observation, as 32-bit scalars (gfx950 has no
  // scalar 64-bit order compare: every int64 test in the walk below was a v_mov_b64 + v_cmp on
  // the vector unit); what lies further away than 2^30 either way is clamped there.
  auto rel = [&](int64_t pos) -> int {
    const uint64_t d = (uint64_t)(pos - i0) + (1u << 30);
    const uint32_t hi = (uint32_t)(d >> 32), lo = (uint32_t)d;
    return (hi | (lo >> 31)) == 0 ? (int)lo - (1 << 30) : (int32_t)hi < 0 ? -(1 << 30) : (1 << 30);
  };
  const int i1r = (int)(i1 - i0);
  const int lastr = rel(a.n_obs - 1);
  const int64_t last = a.n_obs - 1;
  int f, fn, jn2;
  double u, v, un, vn;
  float xf0, xf1, xf2, xn0, xn1, xn2;
  {
    const int64_t ic0 = i0 + lane < last ? i0 + lane : last;
    const int64_t ic1 = i0 + 64 + lane < last ? i0 + 64 + lane : last;
    const int64_t ic2 = i0 + 128 + lane < last ? i0 + 128 + lane : last;
    f = a.obs_fr[ic0];
    u = a.obs_u[ic0];
    v = a.obs_v[ic0];
    const int j0 = a.obs_corner[ic0], j1 = a.obs_corner[ic1];
    if (i0 + lane >= i1) f = -1;
    xf0 = a.pts[3 * j0];
    xf1 = a.pts[3 * j0 + 1];
    xf2 = a.pts[3 * j0 + 2];
    xn0 = a.pts[3 * j1];
    xn1 = a.pts[3 * j1 + 1];
    xn2 = a.pts[3 * j1 + 2];
    fn = a.obs_fr[ic1];
    un = a.obs_u[ic1];
    vn = a.obs_v[ic1];
    jn2 = a.obs_corner[ic2];
  }
  int ff = __builtin_amdgcn_readfirstlane(f);
  int fbeg = rel(fr_obs_begin[ff]), fend = rel(fr_obs_begin[ff + 1]);
  int fend2 = (ff + 2 <= n_fr) ? rel(fr_obs_begin[ff + 2]) : (1 << 30);
  FrRing7 ring(a.frtab, tab, lane, n_fr);

#pragma unroll 1
  for (int s0 = 0; s0 < i1r; s0 += 64) {
    const int s1 = (s0 + 64 < i1r) ? s0 + 64 : i1r;
    const int f_lo = __builtin_amdgcn_readfirstlane(f);
    const int f_hi = __builtin_amdgcn_readlane(f, s1 - s0 - 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int fb = f_lo; fb <= f_hi; fb += FUSED_T) {
      const int nf = (f_hi - fb + 1 < FUSED_T) ? (f_hi - fb + 1) : FUSED_T;
      ring.ensure(fb, nf);
      if (f >= fb && f < fb + nf) {
        const double X[3] = {(double)xf0, (double)xf1, (double)xf2};
        double r[2];
        SlabSink7QK sink{slab2 + lane, {}, {}};
        eval_chain<NODEN, true>(ring.entry(f), X, u, v, r, sink, a.refine_mask);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- everything requested a slab ago is taken over; the requests of this slab go out
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    f = s0 + 64 + lane < i1r ? fn : -1;
    u = un;
    v = vn;
    xf0 = xn0;
    xf1 = xn1;
    xf2 = xn2;
    ring.advance(ff);  // FRs before the Gram walk's current one are done with their entries
    {
      // wave-uniform base (scalar) + a 32-bit lane offset, clamped to the last observation
      const int room = lastr - s0;  // >= 0
      const unsigned lim = (unsigned)(room < (1 << 20) ? room : (1 << 20));
      const int64_t sb = i0 + s0;
      const unsigned o2 = (unsigned)(128 + lane) < lim ? (unsigned)(128 + lane) : lim;
      const unsigned o3 = (unsigned)(192 + lane) < lim ? (unsigned)(192 + lane) : lim;
      const float* px = a.pts + 3 * jn2;
      xn0 = px[0];
      xn1 = px[1];
      xn2 = px[2];
      fn = (a.obs_fr + sb)[o2];
      un = (a.obs_u + sb)[o2];
      vn = (a.obs_v + sb)[o2];
      jn2 = (a.obs_corner + sb)[o3];
    }
    // ---- Gram steps of every part that intersects the slab
#pragma unroll 1
    for (;;) {
      if (fend > s0) {
        const int b = fbeg > s0 ? fbeg : s0, e = fend < s1 ? fend : s1;
        const int part_end = fend < i1r ? fend : i1r;
        const int b_rel = b - s0, e_rel = e - s0;
        auto partial_group = [&](int g16) {
          const int o = g16 + qo;
          const bool on = o >= b_rel && o < e_rel;
          gram7qk(slab2, on ? base0 + g16 : F7QK_ZERO, on ? base1 + scale1 * g16 : F7QK_ZERO, on ? basek + g16 : F7QK_ZERO,
                  on ? 4 * F7Q_PS : 0, q, kv, kk);
        };
        int g16 = b_rel & ~15;
        if (g16 < b_rel || g16 + 16 > e_rel) {
          partial_group(g16);
          g16 += 16;
        }
#pragma unroll 1
        for (; g16 + 16 <= e_rel; g16 += 16)
          gram7qk(slab2, base0 + g16, base1 + scale1 * g16, basek + g16, 4 * F7Q_PS, q, kv, kk);
        if (g16 < e_rel) partial_group(g16);
        if (e == part_end) {
          // ---- the part is complete: G7 -> G = C^T G7 C, then out
          double c0 = cofs0 == -2 ? 1.0 : 0.0, c1 = cofs1 == -2 ? 1.0 : 0.0;
          const double* ent = ring.entry(ff);
          if (!ring.holds(ff)) {  // a slab with more than FUSED_T FRs has overwritten the early ones
            tab_late[lane] = a.frtab[(int64_t)ff * FRC_SIZE + lane];
            __builtin_amdgcn_wave_barrier();
            ent = tab_late;
          }
          if (cofs0 >= 0) c0 = ent[cofs0];
          if (cofs1 >= 0) c1 = ent[cofs1];
          const int piece = part_piece[part];
          const bool piece_ends = e >= i1r || part_piece[part + 1] != piece;
          // sums over the four blocks: lane (j, b, i) = S[i][j] for every b
          double s00 = q[0], s10 = q[1], s11 = q[2];
          q[0] = q[1] = q[2] = 0.0;
          s00 += row_dpp<0x124>(s00);
          s10 += row_dpp<0x124>(s10);
          s11 += row_dpp<0x124>(s11);
          s00 += row_dpp<0x128>(s00);
          s10 += row_dpp<0x128>(s10);
          s11 += row_dpp<0x128>(s11);
          const double s01 = lane_fetch(s10, fetch_t);  // G7[g][4 + x] = S10[x][g]
          // lanes c < 8: G7[g][c] and G7[g + 4][c] (the others feed rows of P that nothing reads)
          const double t0 = qb == 0 ? s00 : s01, t1 = qb == 0 ? s10 : s11;
          f64x4 P = {0.0, 0.0, 0.0, 0.0};
          f64x4 G = {0.0, 0.0, 0.0, 0.0};
          P = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, c0, P, 0, 0, 0);
          P = __builtin_amdgcn_mfma_f64_16x16x4f64(t1, c1, P, 0, 0, 0);
          G = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, P[0], G, 0, 0, 0);
          G = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, P[1], G, 0, 0, 0);
          if (c < 8) {  // record A: G[0..15][0..7]; 32 consecutive doubles per store
            double* out = recA + (int64_t)part * REC_AW + g * 8 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[32 * r] = G[r];
          }
          {  // ---- record K: K^T U = (K^T V) C and K^T K, rows = camera parameter g + 4 r, columns [U | K | 0]
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              kv[k] += row_dpp<0x124>(kv[k]);
              kk[k] += row_dpp<0x124>(kk[k]);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              kv[k] += row_dpp<0x128>(kv[k]);
              kk[k] += row_dpp<0x128>(kk[k]);
            }
            // lane (j, b, i) now holds, for every b, entry [i][j] of each chunk product.  (K^T V)[c][g]
            // for parameter c = 4 a + i sits in lane g + 4 b + 16 i of product (a, h): the lane
            // with b = a picks its own a, then the transposing fetch of the Gram above brings it over.
            const double sel0 = qb == 0 ? kv[0] : qb == 1 ? kv[2] : qb == 2 ? kv[4] : 0.0;
            const double sel1 = qb == 0 ? kv[1] : qb == 1 ? kv[3] : qb == 2 ? kv[5] : 0.0;
            const double kv0 = lane_fetch(sel0, fetch_t), kv1 = lane_fetch(sel1, fetch_t);
            f64x4 tku = {0.0, 0.0, 0.0, 0.0};
            tku = __builtin_amdgcn_mfma_f64_16x16x4f64(kv0, c0, tku, 0, 0, 0);
            tku = __builtin_amdgcn_mfma_f64_16x16x4f64(kv1, c1, tku, 0, 0, 0);
            // (K^T K)[g + 4 r][c], c = 4 b + j: chunk product (r, b) holds it in this very lane for
            // b <= r; for b > r it is the transpose of product (b, r), fetched from the lane with
            // that b which picked its own.  kk index: (a, a') -> a (a + 1) / 2 + a'.
            const double tr0 = lane_fetch(qb == 1 ? kk[1] : qb == 2 ? kk[3] : 0.0, fetch_t);  // (1,0) / (2,0) for row chunk 0
            const double tr1 = lane_fetch(qb == 2 ? kk[4] : 0.0, fetch_t);                    // (2,1) for row chunk 1
            f64x4 tkk;
            tkk[0] = qb == 0 ? kk[0] : tr0;
            tkk[1] = qb == 0 ? kk[1] : qb == 1 ? kk[2] : tr1;
            tkk[2] = qb == 0 ? kk[3] : qb == 1 ? kk[4] : qb == 2 ? kk[5] : 0.0;
            if (qb == 3) tkk[0] = tkk[1] = 0.0;
            double* out = a.recK + (int64_t)part * REC_K + g * 32 + c;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
              out[128 * r] = tku[r];
              out[128 * r + 16] = tkk[r];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) kv[k] = kk[k] = 0.0;
          }
          bacc += G;
          costacc += G[1];
          if (piece_ends) {
            if (c >= 6) {  // record B: rows 7..15, columns 6..15 of the piece's sum
              double* out = recB + (int64_t)piece * REC_BW + (c - 6);
#pragma unroll
              for (int r = 1; r < 4; ++r) {
                const int row = g + 4 * r;
                if (row >= 7) out[(row - 7) * 10] = bacc[r];
              }
            }
            bacc = (f64x4){0.0, 0.0, 0.0, 0.0};
            asm volatile("" ::: "memory");  // keeps this a (wave-uniform) branch: as selects it is 8 v_cndmask per part
          }
          ++part;
        }
        if (fend >= s1) break;
      }
      ++ff;
      fbeg = fend;
      fend = fend2;
      fend2 = (ff + 2 <= n_fr) ? rel(fr_obs_begin[ff + 2]) : (1 << 30);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 38) a.wave_cost[w] = costacc;
}

__global__ __launch_bounds__(64) void fused7qk_kernel(SweepArgs a, int wave_obs, int n_fr,
                                                      const int64_t* __restrict__ fr_obs_begin,
                                                      const int32_t* __restrict__ wave_part0,
                                                      const int32_t* __restrict__ part_piece,
                                                      double* __restrict__ recA, double* __restrict__ recB) {
  fused7qk_body<false>(a, wave_obs, n_fr, fr_obs_begin, wave_part0, part_piece, recA, recB);
}
__global__ __launch_bounds__(64) void fused7qk_noden_kernel(SweepArgs a, int wave_obs, int n_fr,
                                                            const int64_t* __restrict__ fr_obs_begin,
                                                            const int32_t* __restrict__ wave_part0,
                                                            const int32_t* __restrict__ part_piece,
                                                            double* __restrict__ recA, double* __restrict__ recB) {
  fused7qk_body<true>(a, wave_obs, n_fr, fr_obs_begin, wave_part0, part_piece, recA, recB);
}

// Per-FR table of the chain form, [n_fr][64] doubles, from the per-variable [R | t] of
// prep_vars_kernel: one lane per FR, written out through LDS so that a workgroup's 64 entries
// leave as one contiguous 32 KB block.
__global__ __launch_bounds__(64) void prep_frs_kernel(SweepArgs a, int n_fr) {
  // transposed through LDS a quarter entry at a time: the whole 64 x 64 tile (33 KB) left room for
  // four waves per CU, and the chain of dependent loads in front of the arithmetic had nothing to
  // hide behind
  constexpr int Q = 16, QS = Q + 2;  // row stride even: 16-byte LDS reads
  __shared__ __attribute__((aligned(16))) double sh[64 * QS];
  const int t = threadIdx.x;
  const int f0 = blockIdx.x * 64;
  const int fr = f0 + t;
  double e[FRC_SIZE];
  if (fr < n_fr) {
    const int ic = a.fr_cam[fr], it = a.fr_tim[fr], ip = a.fr_pat[fr];
    double C[12], T[12], P[12], K[12];
    const double* pc = a.prep + 12 * (int64_t)ic;
    const double* ptm = a.prep + 12 * (int64_t)(it >= 0 ? it : a.n_vars);
    const double* pp = a.prep + 12 * (int64_t)(ip >= 0 ? ip : a.n_vars);
    const double* pk = a.intr + 12 * (int64_t)ic;
    // 96-byte blocks, 16-byte aligned: six 16-byte loads each (a lane's loads share no line with its
    // neighbours', so the instruction count is what the texture unit sees)
#pragma unroll
    for (int k = 0; k < 12; k += 2) {
      const double2 c2 = *reinterpret_cast<const double2*>(pc + k), t2 = *reinterpret_cast<const double2*>(ptm + k);
      const double2 p2 = *reinterpret_cast<const double2*>(pp + k), k2 = *reinterpret_cast<const double2*>(pk + k);
      C[k] = c2.x; C[k + 1] = c2.y;
      T[k] = t2.x; T[k + 1] = t2.y;
      P[k] = p2.x; P[k + 1] = p2.y;
      K[k] = k2.x; K[k + 1] = k2.y;
    }
    prep_fr_chain(C, T, P, K, e);
  }
  const int nf = n_fr - f0 < 64 ? n_fr - f0 : 64;
  double* out = a.frtab + (int64_t)f0 * FRC_SIZE;
#pragma unroll
  for (int q = 0; q < FRC_SIZE / Q; ++q) {
    __syncthreads();
    if (fr < n_fr) {
#pragma unroll
      for (int k = 0; k < Q; ++k) sh[t * QS + k] = e[Q * q + k];
    }
    __syncthreads();
    // eight entries per step, 16 consecutive doubles (128 B) of each as eight 16-byte stores
    for (int k = t >> 3; k < nf; k += 8)
      *reinterpret_cast<double2*>(out + k * FRC_SIZE + Q * q + 2 * (t & 7)) =
          *reinterpret_cast<const double2*>(sh + k * QS + 2 * (t & 7));
  }
}

void launch_fused_prepass(const SweepArgs& a, int n_fr, hipStream_t s) {
  if (a.n_obs <= 0) return;
  launch_prep_vars(a, s);
  hipLaunchKernelGGL(prep_frs_kernel, dim3((n_fr + 63) / 64), dim3(64), 0, s, a, n_fr);
}

void launch_fused(const SweepArgs& a, int wave_obs, int n_fr, const int64_t* fr_obs_begin, const int32_t* wave_part0,
                  const int32_t* part_piece, double* recA, double* recB, hipStream_t s) {
  if (a.n_obs <= 0) return;
  const int64_t waves = (a.n_obs + wave_obs - 1) / wave_obs;
#define RP_LAUNCH(K) \
  hipLaunchKernelGGL(K, dim3((unsigned)waves), dim3(64), 0, s, a, wave_obs, n_fr, fr_obs_begin, wave_part0, part_piece, recA, recB)
  if (a.refine_mask && a.recK) {
    if (a.no_denominator) RP_LAUNCH(fused7qk_noden_kernel);
    else RP_LAUNCH(fused7qk_kernel);
  } else {
    if (a.no_denominator) RP_LAUNCH(fused7q_kernel);
    else RP_LAUNCH(fused7q_den_kernel);
  }
#undef RP_LAUNCH
}

// ---------------------------------------------------------------- gathers
// record offsets
constexpr int RA_CC = 0, RA_TC = 128;              // record A (per FR): (p,c~) at 64 travels with (c~,c~)
constexpr int RB_PP = 0, RB_TP = 64, RB_TT = 128;  // record B (per piece)

// out[b][t] = sum over i in [begin[b], begin[b+1]) of src[(list ? list[i] : i) * stride + off + t],
// t < blockDim.x, rows of out `ostride` apart (entries beyond blockDim.x are left alone)
__global__ __launch_bounds__(128) void list_sum_kernel(const double* __restrict__ src, int stride, int off,
                                                       const int32_t* __restrict__ list,
                                                       const int32_t* __restrict__ begin, double* __restrict__ out,
                                                       int ostride) {
  const int b = blockIdx.x, t = threadIdx.x;
  double acc = 0.0;
  const int i1 = begin[b + 1];
  int i = begin[b];
  // eight records in flight per thread (the one-at-a-time loop waited out a dependent index ->
  // record load chain per term); the terms are still added in list order
  for (; i + 8 <= i1; i += 8) {
    int64_t idx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) idx[k] = list ? list[i + k] : i + k;
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[idx[k] * stride + off + t];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  for (; i < i1; ++i) acc += src[(int64_t)(list ? list[i] : i) * stride + off + t];
  out[(int64_t)b * ostride + t] = acc;
}
// one block per reduced block: diagonal 7x7, gradient, and (for cameras) the
// pattern-camera off-diagonal blocks of its groups.  group_sum: [c~c~ | p c~]
__global__ __launch_bounds__(64) void assemble_reduced_kernel(GatherArgs g) {
  const int rb = blockIdx.x, t = threadIdx.x;
  const int a = t >> 3, b = t & 7;  // 8x8 slot
  double acc = (b < 7) ? g.pp_sum[(int64_t)rb * 64 + t] : 0.0;  // zero for camera blocks
  for (int k = g.rblock_group_begin[rb]; k < g.rblock_group_begin[rb + 1]; ++k) {
    const int gr = g.rblock_groups[k];
    const double* gs = g.group_sum + (int64_t)gr * 128;
    if (g.group_rc[gr] == rb) {
      acc += gs[t];  // cc~
      const int rp = g.group_rp[gr];
      if (rp >= 0 && a < 7 && b < 7) {
        const double v = gs[64 + t];  // pc~[a][b]: row = pattern param a, col = camera param b
        g.Hrr[(int64_t)(7 * rp + a) * g.ldn + 7 * rb + b] = v;
        g.Hrr[(int64_t)(7 * rb + b) * g.ldn + 7 * rp + a] = v;
      }
    } else if (b == 7) {
      acc += gs[64 + a * 8 + 7];  // pattern block: gradient from pc~[a][7]
    }
  }
  if (a < 7) {
    if (b < 7) {
      g.Hrr[(int64_t)(7 * rb + a) * g.ldn + 7 * rb + b] = acc;
      if (a == b) g.hdiag[7 * rb + a] = acc;
    } else
      g.gr[7 * rb + a] = acc;
  }
}
// cost: sum over groups of cc~[7][7]
__global__ __launch_bounds__(256) void cost_from_groups_kernel(GatherArgs g) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < g.n_groups; i += 256) acc += g.group_sum[(int64_t)i * 128 + g.cost_index];
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) *g.cost_sum = g.cost_scale * tot;
}
// per eliminated segment: E = sum over its pieces of tt, g_t = sum of their padding column
__global__ __launch_bounds__(64) void seg_diag_kernel(GatherArgs g) {
  const int sgi = blockIdx.x, t = threadIdx.x;
  const int a = t >> 3, b = t & 7;
  if (a >= 7) return;
  double acc = 0.0;
  for (int k = g.seg_piece_begin[sgi]; k < g.seg_piece_begin[sgi + 1]; ++k)
    acc += g.recB[(int64_t)k * REC_B + RB_TT + a * 8 + b];
  if (b < 7)
    g.Ett[(int64_t)sgi * 49 + a * 7 + b] = acc;
  else
    g.gt[(int64_t)sgi * 7 + a] = acc;
}
// per (segment, reduced block) unit: the coupling block H_rt (7 time parameters x 7 block
// parameters) summed from the records and, in the same thread, taken through the segment's
// forward substitution: Y[7*seg+i][7*rb+j], L y = s_t .* h per column j.  H_rt itself is
// never stored; a retry with another radius re-reads the records.  One workgroup per
// segment; a thread owns (unit, j), so consecutive threads write consecutive columns of Y.
__global__ __launch_bounds__(256) void unit_y_kernel(GatherArgs g, ElimArgs e) {
  const int sgi = blockIdx.x;
  const int u0 = g.seg_unit_begin[sgi], u1 = g.seg_unit_begin[sgi + 1];
  const int v7 = 7 * e.seg_var[sgi];
  const double* L = e.Lt + (int64_t)sgi * 49;
  for (int el = threadIdx.x; el < (u1 - u0) * 7; el += 256) {
    const int u = u0 + el / 7, j = el % 7;
    double acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = 0.0;
    for (int k = g.unit_begin[u]; k < g.unit_begin[u + 1]; ++k) {
      const int c = g.contrib[k];
      const double* src = (c & 1) ? g.recB + (int64_t)(c >> 1) * REC_B + RB_TP : g.recA + (int64_t)(c >> 1) * REC_A + RA_TC;
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] += src[i * 8 + j];
    }
    double* dst = e.Y + (int64_t)(7 * sgi) * e.ldn + 7 * g.unit_rblock[u] + j;
    double y[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      double t = e.scale[v7 + i] * acc[i];
#pragma unroll
      for (int k = 0; k < i; ++k) t -= L[7 * i + k] * y[k];
      y[i] = t / L[7 * i + i];
      dst[(int64_t)i * e.ldn] = y[i];
    }
  }
}

// ---------------------------------------------------------------- gathers, rotation-vector records
// The fused kernel's records are Gram blocks of U (rp_model.hpp).  Each block's seven
// reference columns are U columns times a matrix of that block alone:
//     camera   [wc a]  (6) -> [q_c t_c]:  [E(q_c) 0; 0 I]
//     pattern  [wp b2] (6) -> [q_p t_p]:  [E(q_p) 0; 0 I]
//     time     [wt b2] (6) -> [q_t t_t]:  [E(q_t) 0; 0 Rt^T]
// (E = quat_rotvec_map), and every sum below runs over records that share the block(s) in
// question, so the sums are taken in U space and converted once.
// out[a][b] = sum_ij TL[i][a] G[i][j] TR[j][b] over 6 x 6 (G with row stride ldg); one thread per (a, b)
__device__ __forceinline__ double sandwich66(const double* TL, int a, const double* G, int ldg, const double* TR, int b) {
  double acc = 0.0;
  for (int i = 0; i < 6; ++i) {
    double t = 0.0;
    for (int j = 0; j < 6; ++j) t += G[i * ldg + j] * TR[7 * j + b];
    acc += TL[7 * i + a] * t;
  }
  return acc;
}
// one block per reduced block (see assemble_reduced_kernel); group_sum: [16][8] per group in U space
__global__ __launch_bounds__(64) void assemble_reduced_w_kernel(GatherArgs g) {
  __shared__ double Tb[42], To[42], S[64], S2[48];
  const int rb = blockIdx.x, t = threadIdx.x;
  const int a = t >> 3, b = t & 7;  // 8x8 slot
  if (t == 0) block_map(g.x, g.rvar[rb], false, Tb);
  const int k0 = g.rblock_group_begin[rb], k1 = g.rblock_group_begin[rb + 1];
  const bool camera = k0 < k1 && g.group_rc[g.rblock_groups[k0]] == rb;
  double out = 0.0;  // this thread's entry of the block's [7][8] = [H diagonal block | gradient]
  if (camera) {
    // [wc a r] x [wc a r] summed over the camera's groups
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) acc += g.group_sum[(int64_t)g.rblock_groups[k] * 128 + t];
    S[t] = acc;
    __syncthreads();
    if (a < 7) {
      if (b < 7) {
        out = sandwich66(Tb, a, S, 8, Tb, b);
      } else {  // gradient: column r
        for (int i = 0; i < 6; ++i) out += Tb[7 * i + a] * S[i * 8 + 6];
      }
    }
    // pattern-camera blocks of its groups
    for (int k = k0; k < k1; ++k) {
      const int gr = g.rblock_groups[k];
      const int rp = g.group_rp[gr];
      if (rp < 0) continue;  // wave-uniform
      __syncthreads();
      if (t == 0) block_map(g.x, g.rvar[rp], false, To);
      if (t < 48) S2[t] = g.group_sum[(int64_t)gr * 128 + 56 + t];  // rows 7..12 = [wp b2], [6][8]
      __syncthreads();
      if (a < 7 && b < 7) {
        const double v = sandwich66(To, a, S2, 8, Tb, b);  // row = pattern parameter a, column = camera parameter b
        g.Hrr[(int64_t)(7 * rp + a) * g.ldn + 7 * rb + b] = v;
        g.Hrr[(int64_t)(7 * rb + b) * g.ldn + 7 * rp + a] = v;
      }
    }
  } else {
    // pattern block: pp_sum is the first 64 entries of the summed record B: rows [wp b2] x [r | wp b2 wt]
    S[t] = g.pp_sum[(int64_t)rb * 64 + t];
    __syncthreads();
    if (a < 7) {
      if (b < 7) {
        out = sandwich66(Tb, a, S + 1, 10, Tb, b);
      } else {
        for (int i = 0; i < 6; ++i) out += Tb[7 * i + a] * S[i * 10];
      }
    }
  }
  if (a < 7) {
    if (b < 7) {
      if (a >= b) {  // lower triangle, mirrored: exactly symmetric
        g.Hrr[(int64_t)(7 * rb + a) * g.ldn + 7 * rb + b] = out;
        g.Hrr[(int64_t)(7 * rb + b) * g.ldn + 7 * rb + a] = out;
      }
      if (a == b) g.hdiag[7 * rb + a] = out;
    } else
      g.gr[7 * rb + a] = out;
  }
}
// record B rows / columns of [wt b2]
__device__ __forceinline__ int rb_row_t(int i) { return i < 3 ? 6 + i : i; }
__device__ __forceinline__ int rb_col_t(int j) { return j < 3 ? 7 + j : 1 + j; }
// per eliminated segment: E_tt and g_t from the [wt b2] x [wt b2 | r] part of its pieces' records
__global__ __launch_bounds__(64) void seg_diag_w_kernel(GatherArgs g) {
  __shared__ double Tt[42], S[42];
  const int sgi = blockIdx.x, t = threadIdx.x;
  if (t == 63) block_map(g.x, g.seg_var[sgi], true, Tt);
  if (t < 42) {
    const int i = t / 7, j = t % 7;
    const int off = rb_row_t(i) * 10 + (j < 6 ? rb_col_t(j) : 0);
    double acc = 0.0;
    for (int k = g.seg_piece_begin[sgi]; k < g.seg_piece_begin[sgi + 1]; ++k) acc += g.recB[(int64_t)k * REC_BW + off];
    S[t] = acc;
  }
  __syncthreads();
  const int a = t >> 3, b = t & 7;
  if (a >= 7) return;
  if (b < 7) {
    if (a >= b) {
      const double v = sandwich66(Tt, a, S, 7, Tt, b);
      g.Ett[(int64_t)sgi * 49 + a * 7 + b] = v;
      g.Ett[(int64_t)sgi * 49 + b * 7 + a] = v;
    }
  } else {
    double acc = 0.0;
    for (int i = 0; i < 6; ++i) acc += Tt[7 * i + a] * S[i * 7 + 6];
    g.gt[(int64_t)sgi * 7 + a] = acc;
  }
}
// unit_y_kernel for rotation-vector records: per unit the 6 x 6 block [wt b2] x [wc a] (camera
// block, record A of its parts) or [wt b2] x [wp b2] (pattern block, record B of its pieces)
// is summed in U space into LDS, UNIT_CHUNK units at a time, then a thread per (unit, column)
// applies the time block's map to its column and runs the forward substitution.  The block's
// side stays in its six columns: Y is [7 * segments][e.ldn] with 6 columns per reduced block
// (e.n = 6 * n_rblocks), and the Schur product is converted after the SYRK (rp_linalg.hip).
// Six-row form (ElimArgs::rows6), without the LDS stage: thread = (unit, column jj of its 6 x 6
// block); it sums the six entries of that column from the unit's records, applies R (from
// time_factor_kernel) and writes six entries of Y.  The form with the block summed into LDS first
// (round 2) took 0.25 ms on config D, of which 0.15 ms remained with the record loads and
// the Y stores both switched off: four rounds per segment, each a chain of index loads, an LDS
// write, a barrier and an LDS read.  Here a thread's elements go through the three levels together,
// two per batch, and there is one barrier per segment: 0.18 ms = 5.0 TB/s for 0.52 GB read (three
// lines of every record A) + 0.39 GB written, the mixed read / write ceiling measured in
// tools/sweep_bench.hip.
constexpr int UY6_BATCH = 2;
__global__ __launch_bounds__(256) void unit_y_w6_kernel(GatherArgs g, ElimArgs e) {
  __shared__ double R[36];
  const int sgi = blockIdx.x;
  const int u0 = g.seg_unit_begin[sgi], u1 = g.seg_unit_begin[sgi + 1];
  if (threadIdx.x < 36) R[threadIdx.x] = e.segw[(int64_t)sgi * SEGW + SEGW_R + threadIdx.x];
  __syncthreads();
  const int n_el = (u1 - u0) * 6;
  // entry (m, jj) of the unit's block in contribution c 
  auto fetch = [&](int c, int m, int jj) {
    const int offA = (m < 3 ? 13 + m : 7 + m) * 8 + jj;  // rows wt 13..15, b2 10..12; columns [wc a]
    const int offB = rb_row_t(m) * 10 + 1 + jj;           // columns [wp b2]
    return (c & 1) ? g.recB[(int64_t)(c >> 1) * REC_BW + offB] : g.recA[(int64_t)(c >> 1) * REC_AW + offA];
  };
  for (int base = 0; base < n_el; base += 256 * UY6_BATCH) {
    int kb[UY6_BATCH], ke[UY6_BATCH], rb[UY6_BATCH], c0[UY6_BATCH];
    double hu[UY6_BATCH][6];
#pragma unroll
    for (int b = 0; b < UY6_BATCH; ++b) {
      const int el = base + threadIdx.x + 256 * b;
      const int u = u0 + (el < n_el ? el / 6 : 0);  // clamped: no branch around the loads
      kb[b] = g.unit_begin[u];
      ke[b] = g.unit_begin[u + 1];
      rb[b] = g.unit_rblock[u];
    }
#pragma unroll
    for (int b = 0; b < UY6_BATCH; ++b) c0[b] = g.contrib[kb[b]];  // every unit has a contribution
#pragma unroll
    for (int b = 0; b < UY6_BATCH; ++b) {
      const int jj = (base + threadIdx.x + 256 * b) % 6;
#pragma unroll
      for (int m = 0; m < 6; ++m) hu[b][m] = fetch(c0[b], m, jj);
    }
#pragma unroll
    for (int b = 0; b < UY6_BATCH; ++b) {
      const int el = base + threadIdx.x + 256 * b, jj = el % 6;
      if (el >= n_el) continue;
      for (int k = kb[b] + 1; k < ke[b]; ++k) {  // a camera that sees several patterns at this time
        const int c = g.contrib[k];
#pragma unroll
        for (int m = 0; m < 6; ++m) hu[b][m] += fetch(c, m, jj);
      }
      double* dst = e.Y + (int64_t)(6 * sgi) * e.ldn + 6 * rb[b] + jj;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double y = 0.0;
#pragma unroll
        for (int m = i; m < 6; ++m) y += R[6 * i + m] * hu[b][m];
        dst[(int64_t)i * e.ldn] = y;
      }
    }
  }
}

// Camera parameters refined: the coupling of a segment's time block with the 12 camera
// parameters of every camera seen in it.  The (segment, camera block) units of unit_y_w list
// exactly the parts of that camera in the segment; rows k of their records K, columns
// [wt b2], summed and taken through the same map W: 12 columns of Y per camera behind the
// six-column blocks.  Thread = (unit, parameter k).
__global__ __launch_bounds__(256) void unit_y_k_kernel(GatherArgs g, ElimArgs e) {
  __shared__ double R[36];
  const int sgi = blockIdx.x;
  const int u0 = g.seg_unit_begin[sgi], u1 = g.seg_unit_begin[sgi + 1];
  if (threadIdx.x < 36) R[threadIdx.x] = e.segw[(int64_t)sgi * SEGW + SEGW_R + threadIdx.x];
  __syncthreads();
  for (int el = threadIdx.x; el < (u1 - u0) * KP; el += 256) {
    const int u = u0 + el / KP, k = el % KP;
    const int rb = g.unit_rblock[u];
    if (rb >= g.nk) continue;  // pattern block
    double hu[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) hu[m] = 0.0;
    for (int kk = g.unit_begin[u]; kk < g.unit_begin[u + 1]; ++kk) {
      const double* src = g.recK + (int64_t)(g.contrib[kk] >> 1) * REC_K + k * 32;  // camera units list parts only
#pragma unroll
      for (int m = 0; m < 6; ++m) hu[m] += src[m < 3 ? UW_WT + m : UW_B2 + (m - 3)];
    }
    double* dst = e.Y + (int64_t)(6 * sgi) * e.ldn + 6 * g.n_rblocks + KP * rb + k;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double y = 0.0;
#pragma unroll
      for (int m = i; m < 6; ++m) y += R[6 * i + m] * hu[m];
      dst[(int64_t)i * e.ldn] = y;
    }
  }
}

// out[b][t] = sum over i in [begin[b], begin[b+1]) of src[(list ? list[i] : i) * width + t], t < width
__global__ __launch_bounds__(128) void list_sum_wide_kernel(const double* __restrict__ src, int width,
                                                            const int32_t* __restrict__ list,
                                                            const int32_t* __restrict__ begin, double* __restrict__ out) {
  const int b = blockIdx.x;
  const int i0 = begin[b], i1 = begin[b + 1];
  for (int t = threadIdx.x; t < width; t += 128) {
    double acc = 0.0;
    for (int i = i0; i < i1; ++i) acc += src[(int64_t)(list ? list[i] : i) * width + t];
    out[(int64_t)b * width + t] = acc;
  }
}
// One workgroup per camera-parameter block kb (= reduced camera block kb): from the sums of
// record K over the camera's (camera, pattern) groups,  K x K  -> diagonal block, diag and
// (column r) gradient;  K x [wc a] -> the block against the camera's own 7 parameters;  per
// group K x [wp b2] -> the block against that pattern's 7 parameters.
__global__ __launch_bounds__(128) void assemble_k_kernel(GatherArgs g) {
  __shared__ double Tb[42], To[42], S[REC_K];
  const int kb = blockIdx.x, t = threadIdx.x;
  const int k0 = g.rblock_group_begin[kb], k1 = g.rblock_group_begin[kb + 1];
  if (t == 0) block_map(g.x, g.rvar[kb], false, Tb);
  for (int el = t; el < REC_K; el += 128) {
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) acc += g.groupK_sum[(int64_t)g.rblock_groups[k] * REC_K + el];
    S[el] = acc;
  }
  __syncthreads();
  const int o = g.n7 + KP * kb;
  for (int el = t; el < KP * KP; el += 128) {
    const int i = el / KP, j = el % KP;
    if (i < j) continue;  // lower triangle, mirrored: exactly symmetric
    const double v = S[i * 32 + 16 + j];
    g.Hrr[(int64_t)(o + i) * g.ldn + o + j] = v;
    g.Hrr[(int64_t)(o + j) * g.ldn + o + i] = v;
    if (i == j) g.hdiag[o + i] = v;
  }
  if (t < KP) g.gr[o + t] = S[t * 32 + UW_R];
  for (int el = t; el < KP * 7; el += 128) {
    const int i = el / 7, b = el % 7;
    double v = 0.0;
    for (int m = 0; m < 6; ++m) v += S[i * 32 + m] * Tb[7 * m + b];  // U columns 0..5 = [wc a]
    g.Hrr[(int64_t)(o + i) * g.ldn + 7 * kb + b] = v;
    g.Hrr[(int64_t)(7 * kb + b) * g.ldn + o + i] = v;
  }
  for (int k = k0; k < k1; ++k) {
    const int gr = g.rblock_groups[k];
    const int rp = g.group_rp[gr];
    if (rp < 0) continue;  // workgroup-uniform
    __syncthreads();
    if (t == 0) block_map(g.x, g.rvar[rp], false, To);
    __syncthreads();
    const double* GK = g.groupK_sum + (int64_t)gr * REC_K;
    for (int el = t; el < KP * 7; el += 128) {
      const int i = el / 7, b = el % 7;
      double v = 0.0;
      for (int m = 0; m < 6; ++m) v += GK[i * 32 + UW_WP + m] * To[7 * m + b];  // U columns 7..12 = [wp b2]
      g.Hrr[(int64_t)(o + i) * g.ldn + 7 * rp + b] = v;
      g.Hrr[(int64_t)(7 * rp + b) * g.ldn + o + i] = v;
    }
  }
}

// Parity hook (calico_rp_eval with camera parameters refined): the 2 x 12 columns per observation
struct GlobalSink {
  double* p;
  __device__ __forceinline__ void operator()(int k, double v) { p[k] = v; }
};
__global__ __launch_bounds__(256) void k_columns_kernel(SweepArgs a, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_obs) return;
  const int f = a.obs_fr[i];
  const int c = a.fr_cam[f], p = a.fr_pat[f], t = a.fr_tim[f];
  double frp[FRP_SIZE];
  prep_fr(a.x + 7 * (int64_t)c, p >= 0 ? a.x + 7 * (int64_t)p : nullptr, t >= 0 ? a.x + 7 * (int64_t)t : nullptr,
          a.intr + 12 * (int64_t)c, frp);
  const int j = a.obs_corner[i];
  const double X[3] = {(double)a.pts[3 * j], (double)a.pts[3 * j + 1], (double)a.pts[3 * j + 2]};
  GlobalSink sink{out + 2 * KP * i};
  eval_intrinsics_columns(frp, X, a.refine_mask, sink);
}
void launch_k_columns(const SweepArgs& a, double* out, hipStream_t s) {
  if (a.n_obs > 0) hipLaunchKernelGGL(k_columns_kernel, dim3((unsigned)((a.n_obs + 255) / 256)), dim3(256), 0, s, a, out);
}

void launch_unit_y(const GatherArgs& g, const ElimArgs& e, hipStream_t s) {
  if (g.n_seg <= 0) return;
  // rotation-vector records always take the six-row form (ElimArgs::rows6 is set with wspace:
  // time_factor_kernel has left Q | R | z6 per segment in e.segw)
  if (g.wspace && g.nk > 0) hipLaunchKernelGGL(unit_y_k_kernel, dim3(g.n_seg), dim3(256), 0, s, g, e);
  if (g.wspace)
    hipLaunchKernelGGL(unit_y_w6_kernel, dim3(g.n_seg), dim3(256), 0, s, g, e);
  else
    hipLaunchKernelGGL(unit_y_kernel, dim3(g.n_seg), dim3(256), 0, s, g, e);
}

void launch_gathers(const GatherArgs& g, hipStream_t s, int what) {
  const int strideA = g.wspace ? REC_AW : REC_A, strideB = g.wspace ? REC_BW : REC_B;
  // rotation-vector records: rows 0..12 of [16][8] are all a (camera, pattern) sum needs (rows
  // 13..15, wt x c~, only make sense per time and are read by unit_y_w)
  const int widthA = g.wspace ? 13 * 8 : 128;
  if (g.n_seg > 0 && (what & GATHER_SEG)) {
    if (g.wspace)
      hipLaunchKernelGGL(seg_diag_w_kernel, dim3(g.n_seg), dim3(64), 0, s, g);
    else
      hipLaunchKernelGGL(seg_diag_kernel, dim3(g.n_seg), dim3(64), 0, s, g);
  }
  if (g.n_chunks > 0 && (what & GATHER_PAIRS)) {
    hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_chunks), dim3(widthA), 0, s, g.recA, strideA, RA_CC, g.pair_part,
                       g.chunk_begin, g.chunk_sum, 128);
    hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_groups), dim3(widthA), 0, s, g.chunk_sum, 128, 0, (const int32_t*)nullptr,
                       g.group_chunk_begin, g.group_sum, 128);
    if (g.cost_sum) hipLaunchKernelGGL(cost_from_groups_kernel, dim3(1), dim3(256), 0, s, g);
  }
  if (g.n_chunks > 0 && (what & GATHER_BLOCKS)) {
    if (g.n_pp_chunks > 0)
      hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_pp_chunks), dim3(64), 0, s, g.recB, strideB, RB_PP, g.pp_piece,
                         g.pp_chunk_begin, g.pp_chunk_sum, 64);
    hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_rblocks), dim3(64), 0, s, g.pp_chunk_sum, 64, 0, (const int32_t*)nullptr,
                       g.pp_rblock_chunk_begin, g.pp_sum, 64);
    if (g.wspace)
      hipLaunchKernelGGL(assemble_reduced_w_kernel, dim3(g.n_rblocks), dim3(64), 0, s, g);
    else
      hipLaunchKernelGGL(assemble_reduced_kernel, dim3(g.n_rblocks), dim3(64), 0, s, g);
    if (g.wspace && g.nk > 0) {  // camera parameters refined: record K through the same chunks and groups
      hipLaunchKernelGGL(list_sum_wide_kernel, dim3(g.n_chunks), dim3(128), 0, s, g.recK, REC_K, g.pair_part, g.chunk_begin,
                         g.chunkK_sum);
      hipLaunchKernelGGL(list_sum_wide_kernel, dim3(g.n_groups), dim3(128), 0, s, g.chunkK_sum, REC_K, (const int32_t*)nullptr,
                         g.group_chunk_begin, g.groupK_sum);
      hipLaunchKernelGGL(assemble_k_kernel, dim3(g.nk), dim3(128), 0, s, g);
    }
  }
}

// ---------------------------------------------------------------- misc
// max |g|: per-block maxima combined with an integer atomic max on the bit pattern
// (non-negative doubles order like unsigned integers; max is order-independent, so the
// result is deterministic).  *out must be 0 before the launch.
__global__ __launch_bounds__(256) void gradient_max_kernel(const double* gr, int n, const double* gt,
                                                           int nt7, double* out) {
  __shared__ double sh[4];
  double m = 0.0;
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmax(m, fabs(gr[i]));
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nt7; i += stride) m = fmax(m, fabs(gt[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double bm = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(bm));
  }
}
void launch_gradient_max(const double* gr, int n, const double* gt, int nt7, double* out, hipStream_t s) {
  (void)hipMemsetAsync(out, 0, sizeof(double), s);
  int blocks = (n + nt7 + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(gradient_max_kernel, dim3(blocks), dim3(256), 0, s, gr, n, gt, nt7, out);
}

// Sharded solves: every rank owns the time blocks of its own segments.  pack: tsum = owned ? x : 0
// over the time variables [t0, vn); after the all-reduce, unpack: x = tsum where some rank owns it.
__global__ __launch_bounds__(256) void time_blocks_pack_kernel(const double* x, const double* own, int t0, int tn,
                                                               double* tsum) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 7 * tn) tsum[i] = own[i / 7] != 0.0 ? x[7 * (int64_t)t0 + i] : 0.0;
}
__global__ __launch_bounds__(256) void time_blocks_unpack_kernel(double* x, const double* owners, int t0, int tn,
                                                                 const double* tsum) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 7 * tn && owners[i / 7] > 0.5) x[7 * (int64_t)t0 + i] = tsum[i];
}
void launch_time_blocks_pack(const double* x, const double* own, int t0, int tn, double* tsum, hipStream_t s) {
  if (tn > 0) hipLaunchKernelGGL(time_blocks_pack_kernel, dim3((7 * tn + 255) / 256), dim3(256), 0, s, x, own, t0, tn, tsum);
}
void launch_time_blocks_unpack(double* x, const double* owners, int t0, int tn, const double* tsum, hipStream_t s) {
  if (tn > 0)
    hipLaunchKernelGGL(time_blocks_unpack_kernel, dim3((7 * tn + 255) / 256), dim3(256), 0, s, x, owners, t0, tn, tsum);
}

// the failure flag as a double next to the iteration's scalars (they travel in one all-gather)
// the iteration's failure flag joins the scalars of its one read-back, and is cleared for the next
// iteration (lm_begin clears it for the first)
__global__ void flag_to_scalar_kernel(int* flag, double* out) {
  *out = (double)*flag;
  *flag = 0;
}
void launch_flag_to_scalar(int* flag, double* out, hipStream_t s) {
  hipLaunchKernelGGL(flag_to_scalar_kernel, dim3(1), dim3(1), 0, s, flag, out);
}

__global__ __launch_bounds__(64) void move_runs_kernel(int64_t n_runs, const int64_t* __restrict__ src_begin,
                                                       const int64_t* __restrict__ dst_begin,
                                                       const int64_t* __restrict__ len, const uint32_t* __restrict__ src,
                                                       uint32_t* __restrict__ dst, int words) {
  for (int64_t k = blockIdx.x; k < n_runs; k += gridDim.x) {
    const uint32_t* a = src + src_begin[k] * words;
    uint32_t* b = dst + dst_begin[k] * words;
    const int64_t nw = len[k] * words;
    for (int64_t w = threadIdx.x; w < nw; w += 64) b[w] = a[w];
  }
}
void launch_move_runs(int64_t n_runs, const int64_t* src_begin, const int64_t* dst_begin, const int64_t* len,
                      const uint32_t* src, uint32_t* dst, int words, hipStream_t s) {
  if (n_runs <= 0) return;
  const int64_t blocks = n_runs < ((int64_t)1 << 22) ? n_runs : ((int64_t)1 << 22);
  hipLaunchKernelGGL(move_runs_kernel, dim3((unsigned)blocks), dim3(64), 0, s, n_runs, src_begin, dst_begin, len, src, dst,
                     words);
}

__global__ __launch_bounds__(64) void fill_obs_fr_kernel(int n_fr, const int64_t* __restrict__ fr_obs_begin,
                                                         int32_t* __restrict__ obs_fr) {
  for (int f = blockIdx.x; f < n_fr; f += gridDim.x) {
    const int64_t e = fr_obs_begin[f + 1];
    for (int64_t i = fr_obs_begin[f] + threadIdx.x; i < e; i += 64) obs_fr[i] = f;
  }
}
void launch_fill_obs_fr(int n_fr, const int64_t* fr_obs_begin, int32_t* obs_fr, hipStream_t s) {
  if (n_fr <= 0) return;
  const int blocks = n_fr < (1 << 22) ? n_fr : (1 << 22);
  hipLaunchKernelGGL(fill_obs_fr_kernel, dim3((unsigned)blocks), dim3(64), 0, s, n_fr, fr_obs_begin, obs_fr);
}

}  // namespace rp
