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
constexpr int COST_GROUP = 8;   // tiles per table build in the cost sweep

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

// ---------------------------------------------------------------- cost sweep
__global__ __launch_bounds__(SWEEP_BLOCK) void cost_sweep_kernel(SweepArgs a, double* partial) {
  __shared__ __attribute__((aligned(16))) double tab[FR_TABLE * FRP_SIZE];
  __shared__ double sh[4];
  const int tid = threadIdx.x;
  constexpr int64_t PER = (int64_t)COST_GROUP * SWEEP_TILE;
  const int64_t ngroups = (a.n_obs + PER - 1) / PER;
  double acc = 0.0;
  NullSink none;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t i_first = grp * PER;
    const int64_t i_end = (i_first + PER < a.n_obs) ? i_first + PER : a.n_obs;
    const int f_first = a.obs_fr[i_first], f_last = a.obs_fr[i_end - 1];
    for (int fb = f_first; fb <= f_last; fb += FR_TABLE) {
      const int nf = (f_last - fb + 1 < FR_TABLE) ? (f_last - fb + 1) : FR_TABLE;
      __syncthreads();
      build_fr_table(a, tab, fb, nf, tid);
      __syncthreads();
#pragma unroll 2
      for (int64_t i = i_first + tid; i < i_end; i += SWEEP_TILE) {
        const int f = a.obs_fr[i];
        if (f >= fb && f < fb + nf) {
          const int j = a.obs_corner[i];
          const double X[3] = {(double)a.pts[3 * j], (double)a.pts[3 * j + 1], (double)a.pts[3 * j + 2]};
          double r[2];
          eval_prepared<false>(tab + (f - fb) * FRP_SIZE, X, a.obs_u[i], a.obs_v[i], r, none);
          acc += r[0] * r[0] + r[1] * r[1];
        }
      }
    }
  }
  const double tot = block_sum_256(acc, sh);
  if (tid == 0) partial[blockIdx.x] = tot;
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
  const int64_t per = (int64_t)COST_GROUP * SWEEP_TILE;
  int64_t blocks = (a.n_obs + per - 1) / per;
  if (blocks > SWEEP_COST_GRID) blocks = SWEEP_COST_GRID;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cost_sweep_kernel, dim3((unsigned)blocks), dim3(SWEEP_BLOCK), 0, s, a, partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, (int)blocks, 0.5, cost_out);
}

// ---------------------------------------------------------------- per-FR accumulation
// One wave walks ACC_FR_PER_WAVE consecutive FRs and computes, per FR, the Gram blocks of
// [c~ | p | t] (c~ = 7 camera columns + residual, p / t = 7 pattern / time columns; rows =
// the FR's 2*n residual rows) on the fp64 matrix cores, straight from the Jacobian planes,
// no LDS.  Two 16x16 tiles per k-step cover every block that is needed downstream:
//     T1 += [c~ p]^T [c~ t]   ->  c~c~ (J_c^T J_c, J_c^T r, r^T r), p c~, c~ t, p t
//     T2 += [p  t]^T [p  t]   ->  p p, t t
// (a third tile, as a plain [c~ p | t] split would need, is 50 % more MFMA work on a
// kernel that is bound by the fp64 MFMA rate).
// v_mfma_f64_16x16x4_f64: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15]; a k-step
// takes 4 rows = 4 observations of one residual row (one per lane group g = lane>>4).
// Lane column c = lane&15 needs two streams: S1 = column c of [c~ p], S2 = column c of
// [p t]; then  T1: A = S1, B = (c < 8 ? S1 : S2)   T2: A = B = S2.
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

constexpr int ACC_STAGE_ROWS = 48, ACC_STAGE_LD = 18;  // per-wave LDS slab: plane rows of 16 observations (+2 pad)

// VARIANT is 0 in the library; tools/acc_bench.hip instantiates 1 (no flush), 2 (no MFMA)
// and 4 (no refetch) to see which part bounds the kernel.
template <int VARIANT>
#ifndef RP_ACC_MIN_BLOCKS
#define RP_ACC_MIN_BLOCKS 4
#endif
__global__ __launch_bounds__(256, RP_ACC_MIN_BLOCKS) void accumulate_kernel(int n_fr, const int64_t* __restrict__ fr_obs_begin,
                                                         const int32_t* __restrict__ fr_piece,
                                                         const double* __restrict__ J, double* __restrict__ recA,
                                                         double* __restrict__ recB) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int f0 = (blockIdx.x * 4 + wave) * ACC_FR_PER_WAVE;
  if (f0 >= n_fr) return;
  const int f_end = (f0 + ACC_FR_PER_WAVE < n_fr) ? f0 + ACC_FR_PER_WAVE : n_fr;
  const int c = lane & 15, g = lane >> 4;
  // plane offsets (within a tile) of this lane's streams: [S1 row0, S1 row1, S2 row0, S2 row1]
  const int pl[4] = {acc_plane_s1(c, 0), acc_plane_s1(c, 1), acc_plane_s2(c, 0), acc_plane_s2(c, 1)};
  const bool low = c < 8;

  f64x4 t1 = {0.0, 0.0, 0.0, 0.0}, t2 = t1;
  // Groups of 16 observations aligned to 16 (= one 128-B line per plane).  The MFMA
  // operand layout puts 16 different planes into every quarter-wave, i.e. 16 cache lines
  // per 16 lanes if loaded directly - the texture path, not HBM, then bounds the kernel.
  // So the group's 44 plane lines are fetched line-wise (8 lanes x 16 B per line, 8 planes
  // per instruction, everything the instruction touches is whole lines) into registers,
  // staged through a per-wave LDS slab, and read back in operand layout.  Lanes at an FR
  // boundary take their whole quad (the tile is allocated whole) and mask at use.
  // All branch-free, so that the waits are counted and the group ahead stays in flight.
  __shared__ double stage_all[4][ACC_STAGE_ROWS * ACC_STAGE_LD];
  double* stage = stage_all[wave];
  const int pls[4] = {pl[0] < 0 ? 0 : pl[0], pl[1] < 0 ? 0 : pl[1], pl[2] < 0 ? 0 : pl[2], pl[3] < 0 ? 0 : pl[3]};
  const int vm1 = pl[0] >= 0 ? 15 : 0, vm2 = pl[2] >= 0 ? 15 : 0;
  const int lrow = lane >> 3, lpiece = (lane & 7) * 2;
  f64x2 raw[6];    // the group in flight, line layout
  int fmask;       // bit q: observation q of this lane belongs to the FR
  auto fetch = [&](int64_t o16) {
    const double* base = J + (o16 >> 8) * (int64_t)(SWEEP_PLANES * SWEEP_TILE) + (int)(o16 & 255) + lpiece;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int plane = 8 * i + lrow;
      plane = plane < SWEEP_PLANES ? plane : SWEEP_PLANES - 1;  // rows 44..47 of the slab are never read
      raw[i] = (VARIANT & 8) ? __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(base + plane * SWEEP_TILE))
                             : *reinterpret_cast<const f64x2*>(base + plane * SWEEP_TILE);
    }
  };
  auto restage = [&](int64_t o16, int64_t b, int64_t e) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      *reinterpret_cast<f64x2*>(stage + (8 * i + lrow) * ACC_STAGE_LD + lpiece) = raw[i];
    // one wave, in-order LDS: the slab accesses may alias, so they stay in program order
    __builtin_amdgcn_wave_barrier();
    const int64_t o = o16 + 4 * g;
    int mask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) mask |= (o + q >= b && o + q < e) ? (1 << q) : 0;
    fmask = mask;
  };
  // eight k-steps per group: residual row 0 then row 1, observation q of every lane;
  // operands come out of the slab one residual row at a time (16 registers, not 32)
  auto gram = [&]() {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const f64x2 a_lo = *reinterpret_cast<const f64x2*>(stage + pls[rr] * ACC_STAGE_LD + 4 * g);
      const f64x2 a_hi = *reinterpret_cast<const f64x2*>(stage + pls[rr] * ACC_STAGE_LD + 4 * g + 2);
      const f64x2 b_lo = *reinterpret_cast<const f64x2*>(stage + pls[2 + rr] * ACC_STAGE_LD + 4 * g);
      const f64x2 b_hi = *reinterpret_cast<const f64x2*>(stage + pls[2 + rr] * ACC_STAGE_LD + 4 * g + 2);
      const double m1[4] = {a_lo.x, a_lo.y, a_hi.x, a_hi.y}, m2[4] = {b_lo.x, b_lo.y, b_hi.x, b_hi.y};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double s1 = ((fmask & vm1) >> q) & 1 ? m1[q] : 0.0;
        const double s2 = ((fmask & vm2) >> q) & 1 ? m2[q] : 0.0;
        if (VARIANT & 2) {
          t1[q] += s1 * (low ? s1 : s2);
          t2[q] += s2 * s2;
        } else {
          t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s1, low ? s1 : s2, t1, 0, 0, 0);
          t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(s2, s2, t2, 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  // D: reg r of lane -> row i = g + 4r, col j = c.
  //   T1 rows: i<8 c~_i, 8..14 p_{i-8};  T1 cols: j<8 c~_j, 8..14 t_{j-8}
  //   T2 rows/cols: <7 p, 8..14 t
  // Records are scattered into the (idle) slab in record order and written as whole
  // lines (the record writes, not the MFMAs, are what the kernel pays for on top of
  // reading J: 3.9 GB of them cost 2.2 ms on config D, hence the per-piece record B).
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
  auto flush_fr = [&](int f, const f64x4& t1) {  // record A: [c~c~ | p c~ | t c~]; the (p,t) quadrant of T1 is dropped
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = g + 4 * r, j = c;
      if (j < 8) {
        if (i < 8) stage[0 * 64 + i * 8 + j] = t1[r];        // (c~,c~)[i][j]
        else stage[1 * 64 + (i - 8) * 8 + j] = t1[r];        // (p,c~)[i-8][j]   (row 15: zeros)
      } else if (i < 8) {
        stage[2 * 64 + (j - 8) * 8 + i] = t1[r];             // (t,c~)[j-8][i]   (col 15: zeros)
      }
    }
    write_record(recA + (int64_t)f * REC_A);
  };
  auto flush_piece = [&](int piece, const f64x4& t2) {  // record B: [p p | t p | t t]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = g + 4 * r, j = c;
      if (i < 8) {
        if (j < 8) stage[0 * 64 + i * 8 + j] = t2[r];        // (p,p)[i][j]      (row/col 7: zeros)
        else stage[1 * 64 + (j - 8) * 8 + i] = t2[r];        // (t,p)[j-8][i]
      } else if (j >= 8) {
        stage[2 * 64 + (i - 8) * 8 + (j - 8)] = t2[r];       // (t,t)[i-8][j-8]
      }
    }
    write_record(recB + (int64_t)piece * REC_B);
  };

  // flat walk over (FR, group) with the next group's lines in flight behind the MFMAs
  int f = f0;
  int64_t b = fr_obs_begin[f], e = fr_obs_begin[f + 1];
  int64_t o16 = b & ~(int64_t)15;
  fetch(o16);
  while (f < f_end) {
    restage(o16, b, e);
    int nf = f;
    int64_t nb = b, ne = e, no = o16 + 16;
    if (no >= e) {
      nf = f + 1;
      if (nf < f_end) {
        nb = e;
        ne = fr_obs_begin[nf + 1];
        no = nb & ~(int64_t)15;
      }
    }
    // past the last FR the prefetch re-reads the current group (valid memory, unused)
    if (!(VARIANT & 4)) fetch(nf < f_end ? no : o16);
    gram();
    if (nf != f) {
      flush_fr(f, t1);
      t1 = (f64x4){0.0, 0.0, 0.0, 0.0};
      const int piece = fr_piece[f];
      if (nf >= f_end || fr_piece[nf] != piece) {
        flush_piece(piece, t2);
        t2 = (f64x4){0.0, 0.0, 0.0, 0.0};
      }
    }
    f = nf;
    b = nb;
    e = ne;
    o16 = no;
  }
}

void launch_accumulate(int n_fr, const int64_t* fr_obs_begin, const int32_t* fr_piece, const double* J,
                       double* recA, double* recB, hipStream_t s) {
  if (n_fr <= 0) return;
  const int per_block = 4 * ACC_FR_PER_WAVE;
  hipLaunchKernelGGL(accumulate_kernel<0>, dim3((n_fr + per_block - 1) / per_block), dim3(256), 0, s, n_fr, fr_obs_begin,
                     fr_piece, J, recA, recB);
}

// ---------------------------------------------------------------- gathers
// record offsets
constexpr int RA_CC = 0, RA_TC = 128;              // record A (per FR): (p,c~) at 64 travels with (c~,c~)
constexpr int RB_PP = 0, RB_TP = 64, RB_TT = 128;  // record B (per piece)

// out[b][t] = sum over i in [begin[b], begin[b+1]) of src[(list ? list[i] : i) * stride + off + t]
__global__ __launch_bounds__(128) void list_sum_kernel(const double* __restrict__ src, int stride, int off,
                                                       const int32_t* __restrict__ list,
                                                       const int32_t* __restrict__ begin, double* __restrict__ out) {
  const int b = blockIdx.x, t = threadIdx.x, width = blockDim.x;
  double acc = 0.0;
  const int i1 = begin[b + 1];
  for (int i = begin[b]; i < i1; ++i) acc += src[(int64_t)(list ? list[i] : i) * stride + off + t];
  out[(int64_t)b * width + t] = acc;
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
  for (int i = threadIdx.x; i < g.n_groups; i += 256) acc += g.group_sum[(int64_t)i * 128 + 63];
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) *g.cost_sum = tot;
}
// per eliminated segment: E = sum over its pieces of tt, g_t = sum over its FRs of tc~[.][7]
__global__ __launch_bounds__(64) void seg_diag_kernel(GatherArgs g) {
  const int sgi = blockIdx.x, t = threadIdx.x;
  const int a = t >> 3, b = t & 7;
  if (a >= 7) return;
  double acc = 0.0;
  if (b < 7) {
    for (int k = g.seg_piece_begin[sgi]; k < g.seg_piece_begin[sgi + 1]; ++k)
      acc += g.recB[(int64_t)k * REC_B + RB_TT + a * 8 + b];
    g.Ett[(int64_t)sgi * 49 + a * 7 + b] = acc;
  } else {
    for (int f = g.seg_fr_begin[sgi]; f < g.seg_fr_begin[sgi + 1]; ++f)
      acc += g.recA[(int64_t)f * REC_A + RA_TC + a * 8 + 7];
    g.gt[(int64_t)sgi * 7 + a] = acc;
  }
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
    const int64_t idx = c >> 1;
    acc += (c & 1) ? g.recB[idx * REC_B + RB_TP + i * 8 + j] : g.recA[idx * REC_A + RA_TC + i * 8 + j];
  }
  g.Hrt[(int64_t)(7 * g.unit_seg[u] + i) * g.ldn + 7 * g.unit_rblock[u] + j] = acc;
}

void launch_gathers(const GatherArgs& g, hipStream_t s) {
  if (g.n_chunks > 0) {
    hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_chunks), dim3(128), 0, s, g.recA, REC_A, RA_CC, g.pair_fr, g.chunk_begin,
                       g.chunk_sum);
    hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_groups), dim3(128), 0, s, g.chunk_sum, 128, 0, (const int32_t*)nullptr,
                       g.group_chunk_begin, g.group_sum);
    if (g.n_pp_chunks > 0)
      hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_pp_chunks), dim3(64), 0, s, g.recB, REC_B, RB_PP, g.pp_piece,
                         g.pp_chunk_begin, g.pp_chunk_sum);
    hipLaunchKernelGGL(list_sum_kernel, dim3(g.n_rblocks), dim3(64), 0, s, g.pp_chunk_sum, 64, 0, (const int32_t*)nullptr,
                       g.pp_rblock_chunk_begin, g.pp_sum);
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
