// ae_kernels.hip — Stage-4 algebraic-error refinement on the machinery of
// the reprojection path: the same block structure (one camera, pattern and time 7-vector
// per FR; p* / t* fixed), hence the same records, gathers, time-block elimination, Schur
// SYRK, dense Cholesky and LM loop.  Only what turns parameters into records differs:
// 12 residual rows per relation C - A P T (ae_model.hpp) instead of 2 per corner.
//   ae_records_kernel : per part, [J r]^T [J r] -> record A / record B (one wave per 8 parts)
//   ae_cost_kernel    : 1/2 sum r^2 at the candidate point
//   ae_eval_kernel    : r and J of every relation (parity hook calico_ae_eval)
// One relation per FR: 1/80 of the corner sweep's data on config D, but as many records
// (3 KB per FR) to write as there.
#include "rp_internal.hpp"
#include "ae_model.hpp"

namespace rp {

namespace {

__device__ __forceinline__ double wave_sum_ae(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// prepared blocks of relation i (FR f): 4 x PREP_SIZE doubles at dst: cam, pat, tim, A
__device__ __forceinline__ void ae_prepare(const SweepArgs& a, int f, int64_t i, double* dst) {
  const int c = a.fr_cam[f], p = a.fr_pat[f], t = a.fr_tim[f];
  prep_block(a.x + 7 * (int64_t)c, dst);
  prep_block(p >= 0 ? a.x + 7 * (int64_t)p : nullptr, dst + PREP_SIZE);
  prep_block(t >= 0 ? a.x + 7 * (int64_t)t : nullptr, dst + 2 * PREP_SIZE);
  prep_block(a.a7 + 7 * i, dst + 3 * PREP_SIZE);
}

constexpr int AE_LD = 22;                        // [J(21) | r] per residual row
constexpr int AE_REL = AE_ROWS * AE_LD + 1;      // doubles per relation in LDS (odd: lanes spread over banks)

}  // namespace

// One wave per AE_WAVE_PARTS consecutive parts (a part = one FR's relations inside one wave
// range; with one relation per FR: one relation).  In batches of NB relations, four lanes per
// relation: lane (rel, col) first prepares ONE of the relation's four blocks (camera, pattern,
// time, A: rotation matrix and 1/|q|^2, an fp64 division each) into LDS, then evaluates matrix
// column col of the 3 x 4 residual (ae_eval_column: three residual rows with their Jacobian rows;
// the columns are independent) into the relation's [J | r] (12 x 22) in LDS.  (One lane per
// relation doing all of it kept 8 of 64 lanes busy for half of the kernel's time.)  Then every
// lane adds, relation by relation, the six record entries it owns: lane l = (a, b) =
// (l / 8, l % 8) owns entry [a][b] of the blocks c~c~, p c~, t c~ (record A) and p p, t p, t t
// (record B), i.e. six dot products of three "left" columns (a of c~ / p / t) with three "right"
// ones (b of c~ / p / t), c~ = [camera 7 | r]; (t,t)[a][7] carries t_a . r.  A part's sums are
// written when the part ends, so a part is written exactly once, by one wave; the record-B
// entries keep adding up over the parts of a piece (same time and pattern, the cameras of a
// rig: 16 on config D) and are written when the piece ends.
template <int NB>
__global__ __launch_bounds__(64) void ae_records_kernel(SweepArgs a, int n_parts, const int32_t* __restrict__ part_fr,
                                                        const int64_t* __restrict__ part_obs_begin,
                                                        const int32_t* __restrict__ part_piece, double* __restrict__ recA,
                                                        double* __restrict__ recB) {
  static_assert(4 * NB <= 64, "four lanes per relation");
  __shared__ double sM[NB * AE_REL];
  __shared__ double sP[NB * 4 * PREP_SIZE];
  const int lane = threadIdx.x;
  const int p0 = blockIdx.x * AE_WAVE_PARTS;
  if (p0 >= n_parts) return;
  const int p1 = min(n_parts, p0 + AE_WAVE_PARTS);
  const int aa = lane >> 3, bb = lane & 7;
  // column of [camera 7 | pattern 7 | time 7 | r] for entry k of c~ / p / t; -1: padding
  const int L0 = aa < 7 ? aa : 21, L1 = aa < 7 ? 7 + aa : -1, L2 = aa < 7 ? 14 + aa : -1;
  const int R0 = bb < 7 ? bb : 21, R1 = bb < 7 ? 7 + bb : -1, R2 = bb < 7 ? 14 + bb : 21;
  const double mL1 = L1 >= 0 ? 1.0 : 0.0, mR1 = R1 >= 0 ? 1.0 : 0.0;
  const int cL1 = max(L1, 0), cL2 = max(L2, 0), cR1 = max(R1, 0);
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int part = p0;
  const int64_t i_end = part_obs_begin[p1];
  for (int64_t i0 = part_obs_begin[p0]; i0 < i_end; i0 += NB) {
    const int nb = (int)min((int64_t)NB, i_end - i0);
    const int rel = lane >> 2, col = lane & 3;
    if (rel < nb) {
      const int64_t i = i0 + rel;
      const int f = a.obs_fr[i];
      const int v = col == 0 ? a.fr_cam[f] : col == 1 ? a.fr_pat[f] : col == 2 ? a.fr_tim[f] : 0;
      const double* src = col == 3 ? a.a7 + 7 * i : (v >= 0 ? a.x + 7 * (int64_t)v : nullptr);
      prep_block(src, sP + (rel * 4 + col) * PREP_SIZE);
    }
    __builtin_amdgcn_wave_barrier();  // one wave: its LDS writes are ordered before its later reads
    if (rel < nb) {
      const double* blk = sP + rel * 4 * PREP_SIZE;
      ae_eval_column<AE_LD, 21>(blk, blk + PREP_SIZE, blk + 2 * PREP_SIZE, blk + 3 * PREP_SIZE, col, sM + rel * AE_REL);
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < nb; ++k) {
      while (part_obs_begin[part + 1] <= i0 + k) {  // uniform: the part of relation i0 + k
        const int piece = part_piece[part];
        double* ra = recA + (int64_t)part * REC_A + lane;
        ra[0] = acc[0]; ra[64] = acc[1]; ra[128] = acc[2];
        acc[0] = acc[1] = acc[2] = 0.0;
        if (part_piece[part + 1] != piece) {  // (the next part is this wave's: part + 1 < p1 here)
          double* rb = recB + (int64_t)piece * REC_B + lane;
          rb[0] = acc[3]; rb[64] = acc[4]; rb[128] = acc[5];
          acc[3] = acc[4] = acc[5] = 0.0;
        }
        ++part;
      }
      const double* M = sM + k * AE_REL;
#pragma unroll
      for (int rr = 0; rr < AE_ROWS; ++rr) {
        const double* row = M + rr * AE_LD;
        const double l0 = row[L0], l1 = row[cL1] * mL1, l2 = row[cL2] * mL1;
        const double r0 = row[R0], r1 = row[cR1] * mR1, r2 = row[R2];
        acc[0] += l0 * r0; acc[1] += l1 * r0; acc[2] += l2 * r0;
        acc[3] += l1 * r1; acc[4] += l2 * r1; acc[5] += l2 * r2;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the last part of this wave (every part holds at least one relation)
  {
    const int piece = part_piece[part];
    double* ra = recA + (int64_t)part * REC_A + lane;
    double* rb = recB + (int64_t)piece * REC_B + lane;
    ra[0] = acc[0]; ra[64] = acc[1]; ra[128] = acc[2];
    rb[0] = acc[3]; rb[64] = acc[4]; rb[128] = acc[5];
  }
}

__global__ __launch_bounds__(256) void ae_cost_kernel(SweepArgs a, double* __restrict__ partial) {
  __shared__ double sh[4];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_obs; i += stride) {
    double blk[4 * PREP_SIZE], r[AE_ROWS];
    ae_prepare(a, a.obs_fr[i], i, blk);
    ae_eval_prepared(blk, blk + PREP_SIZE, blk + 2 * PREP_SIZE, blk + 3 * PREP_SIZE, r, nullptr);
    for (int k = 0; k < AE_ROWS; ++k) acc += r[k] * r[k];
  }
  acc = wave_sum_ae(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(64) void ae_eval_kernel(SweepArgs a, double* __restrict__ r_out, double* __restrict__ J_out) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n_obs) return;
  double blk[4 * PREP_SIZE], r[AE_ROWS];
  ae_prepare(a, a.obs_fr[i], i, blk);
  ae_eval_prepared(blk, blk + PREP_SIZE, blk + 2 * PREP_SIZE, blk + 3 * PREP_SIZE, r, J_out ? J_out + i * (AE_ROWS * 21) : nullptr);
  if (r_out)
    for (int k = 0; k < AE_ROWS; ++k) r_out[i * AE_ROWS + k] = r[k];
}

void launch_ae_records(const SweepArgs& a, int n_parts, const int32_t* part_fr, const int64_t* part_obs_begin,
                       const int32_t* part_piece, double* recA, double* recB, hipStream_t s) {
  constexpr int NB = 8;  // 32 lanes evaluate; 16 (all 64 lanes, 44 KB of LDS per wave) and 4 measured slower
  if (n_parts > 0)
    hipLaunchKernelGGL(ae_records_kernel<NB>, dim3((n_parts + AE_WAVE_PARTS - 1) / AE_WAVE_PARTS), dim3(64), 0, s, a,
                       n_parts, part_fr, part_obs_begin, part_piece, recA, recB);
}
// partial: SWEEP_COST_GRID doubles; *cost_out = 1/2 sum (fixed order)
void launch_ae_cost(const SweepArgs& a, double* partial, double* cost_out, hipStream_t s) {
  int64_t blocks = (a.n_obs + 255) / 256;
  if (blocks > SWEEP_COST_GRID) blocks = SWEEP_COST_GRID;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ae_cost_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, partial);
  launch_sum_partials(partial, (int)blocks, 0.5, cost_out, s);
}
void launch_ae_eval(const SweepArgs& a, double* r_out, double* J_out, hipStream_t s) {
  if (a.n_obs > 0)
    hipLaunchKernelGGL(ae_eval_kernel, dim3((unsigned)((a.n_obs + 63) / 64)), dim3(64), 0, s, a, r_out, J_out);
}

}  // namespace rp
