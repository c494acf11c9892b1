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
#include <cstdlib>
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

// [J | r] of a relation in LDS, 12 rows of 24 slots: [camera 7 | r | pattern 7 | 0 | time 7 | 0],
// i.e. six chunks of four columns c0 c1 p0 p1 t0 t1 for the quad-block MFMA below
constexpr int AE_LD = 24, AE_RCOL = 7, AE_PCOL = 8, AE_TCOL = 16;
constexpr int AE_REL = AE_ROWS * AE_LD + 4;  // doubles per relation in LDS

}  // namespace

// One wave per AE_WAVE_PARTS consecutive parts (a part = one FR's relations inside one wave
// range; with one relation per FR: one relation).  In batches of NB relations, four lanes per
// relation: lane (rel, col) first prepares ONE of the relation's four blocks (camera, pattern,
// time, A: rotation matrix and 1/|q|^2, an fp64 division each) into LDS, then evaluates matrix
// column col of the 3 x 4 residual (ae_eval_column: three residual rows with their Jacobian rows;
// the columns are independent) into the relation's [J | r] (12 x 24) in LDS.
//
// Gram blocks on the quad-block fp64 MFMA (v_mfma_f64_4x4x4_4b_f64: A[b][i][k] in lane
// i + 4b + 16k, B[b][k][j] in lane j + 4b + 16k, D[b][i][j] in lane j + 4b + 16i): the four blocks
// are FOUR RELATIONS, k runs over four of the twelve residual rows, and a chunk pair (left chunk,
// right chunk) is one instruction per k-step.  The six 8 x 8 products the records hold,
//   record A: c~ c~, p c~, t c~      record B: p p, t p, t t~     (c~ = [camera | r], t~ = [time | r]
//   on the right of t t only: its column 7 carries t . r; p and t have a zero in slot 7 otherwise)
// are 24 chunk pairs: 72 instructions and 18 LDS reads per lane for four relations, where the
// scalar form (lane = one entry of each product) read six LDS values per residual row, relation
// and lane: 72 reads per relation, the busiest unit of that kernel.  The products of the four
// relations then go through LDS (over the [J | r] they were made from) into the lane = entry
// layout of the records, where the running sums over the relations of a part (record A) and the
// parts of a piece (record B: same time and pattern, the cameras of a rig: 16 on config D) and the
// 512-byte record stores are what they were.
template <int NB>
__global__ __launch_bounds__(64) void ae_records_kernel(SweepArgs a, int n_parts, const int32_t* __restrict__ part_fr,
                                                        const int64_t* __restrict__ part_obs_begin,
                                                        const int32_t* __restrict__ part_piece, double* __restrict__ recA,
                                                        double* __restrict__ recB) {
  static_assert(4 * NB <= 64 && NB % 4 == 0, "four lanes per relation, four relations per MFMA group");
  static_assert(4 * AE_REL >= 3 * 4 * 64, "three products of four relations fit over their [J | r]");
  __shared__ __attribute__((aligned(16))) double sM[NB * AE_REL];
  __shared__ double sP[NB * 4 * PREP_SIZE];
  const int lane = threadIdx.x;
  const int p0 = blockIdx.x * AE_WAVE_PARTS;
  if (p0 >= n_parts) return;
  const int p1 = min(n_parts, p0 + AE_WAVE_PARTS);
  // MFMA lane (x, b, k): column x of a chunk, relation b of the group, residual row 4 ks + k
  const int qx = lane & 3, qb = (lane >> 2) & 3, qk = lane >> 4;
  const int opnd = qb * AE_REL + qk * AE_LD + qx;
  // D[b][i][j] sits in lane j + 4b + 16i: entry [4 ha + i][4 hb + j] of relation b's product
  const int stage = qb * 192 + qk * 8 + qx;
  double accA[3] = {0.0, 0.0, 0.0}, accB[3] = {0.0, 0.0, 0.0};
  int partA = p0, partB = p0;
  const int64_t i_begin = part_obs_begin[p0], i_end = part_obs_begin[p1];
  // The inputs of a relation's four blocks hang on a chain of three dependent loads (relation ->
  // FR -> variable index -> 7 parameters).  Each level is requested for a later batch at ONE point
  // per batch, between the evaluation and the Gram steps, and awaited there a batch later (vector
  // memory returns in order, record stores included: a wait anywhere else would sit behind the
  // stores of the batch before): parameters two batches ahead, indices three, FRs four.  All of
  // it unconditional, through clamped relation numbers.
  const int rel = lane >> 2, col = lane & 3;
  const int32_t* vtab = col == 0 ? a.fr_cam : col == 1 ? a.fr_pat : a.fr_tim;
  auto rel_at = [&](int64_t ib) { return ib + rel < i_end ? ib + rel : i_end - 1; };
  auto load_x = [&](int v, int64_t ib, double* out) {
    const double* src = col == 3 ? a.a7 + 7 * rel_at(ib) : a.x + 7 * (int64_t)(v >= 0 ? v : 0);
#pragma unroll
    for (int k = 0; k < 7; ++k) out[k] = src[k];
  };
  double b0[7], b1[7];
  int v0, v1, v2, f3;
  {
    const int f0 = a.obs_fr[rel_at(i_begin)], f1 = a.obs_fr[rel_at(i_begin + NB)], f2 = a.obs_fr[rel_at(i_begin + 2 * NB)];
    f3 = a.obs_fr[rel_at(i_begin + 3 * NB)];
    v0 = vtab[f0];
    v1 = vtab[f1];
    v2 = vtab[f2];
    load_x(v0, i_begin, b0);
    load_x(v1, i_begin + NB, b1);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // nothing pending at the loop's entry: no waits inside but the one below
  }
  for (int64_t i0 = i_begin; i0 < i_end; i0 += NB) {
    const int nb = (int)min((int64_t)NB, i_end - i0);
    if (rel < nb) prep_block(col == 3 || v0 >= 0 ? b0 : nullptr, sP + (rel * 4 + col) * PREP_SIZE);
    __builtin_amdgcn_wave_barrier();  // one wave: its LDS writes are ordered before its later reads
    if (rel < nb) {
      const double* blk = sP + rel * 4 * PREP_SIZE;
      ae_eval_column<AE_LD, AE_RCOL, AE_PCOL, AE_TCOL, AE_LD>(blk, blk + PREP_SIZE, blk + 2 * PREP_SIZE,
                                                              blk + 3 * PREP_SIZE, col, sM + rel * AE_REL);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- what was requested a batch ago is taken over; the next requests go out
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int k = 0; k < 7; ++k) b0[k] = b1[k];
    v0 = v1;
    v1 = v2;
    load_x(v1, i0 + 2 * NB, b1);
    v2 = vtab[f3];
    f3 = a.obs_fr[rel_at(i0 + 4 * NB)];
    for (int g0 = 0; g0 < nb; g0 += 4) {
      double* Mg = sM + g0 * AE_REL;  // [J | r] of the group's relations, then their products
      double d[24];                   // product P (c~c~ pc~ tc~ pp tp tt~), chunk pair (ha, hb): d[4 P + 2 ha + hb]
#pragma unroll
      for (int k = 0; k < 24; ++k) d[k] = 0.0;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const double* o = Mg + opnd + ks * 4 * AE_LD;
        const double c0 = o[0], c1 = o[4], q0 = o[8], q1 = o[12], t0 = o[16], t1 = o[20];
        const double u1 = qx == 3 ? c1 : t1;  // t~ chunk 1: [time 4..6 | r]
        const double L[3][2] = {{c0, c1}, {q0, q1}, {t0, t1}};
#pragma unroll
        for (int P = 0; P < 6; ++P) {
          const int li = P < 3 ? P : P == 3 ? 1 : 2;  // left vector: c~ p t | p t t
          const double r0 = P < 3 ? c0 : P < 5 ? q0 : t0;
          const double r1 = P < 3 ? c1 : P < 5 ? q1 : u1;
#pragma unroll
          for (int ha = 0; ha < 2; ++ha) {
            d[4 * P + 2 * ha] = __builtin_amdgcn_mfma_f64_4x4x4f64(L[li][ha], r0, d[4 * P + 2 * ha], 0, 0, 0);
            d[4 * P + 2 * ha + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(L[li][ha], r1, d[4 * P + 2 * ha + 1], 0, 0, 0);
          }
        }
      }
      const int ng = nb - g0 < 4 ? nb - g0 : 4;
      // ---- record A: products 0..2 through LDS, running sums over the relations of a part
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int P = 0; P < 3; ++P)
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) Mg[stage + P * 64 + ha * 32 + hb * 4] = d[4 * P + 2 * ha + hb];
      __builtin_amdgcn_wave_barrier();
      for (int k = 0; k < ng; ++k) {
        while (part_obs_begin[partA + 1] <= i0 + g0 + k) {  // uniform: the part of this relation
          double* ra = recA + (int64_t)partA * REC_A + lane;
          ra[0] = accA[0]; ra[64] = accA[1]; ra[128] = accA[2];
          accA[0] = accA[1] = accA[2] = 0.0;
          ++partA;
        }
        const double* sg = Mg + k * 192 + lane;
        accA[0] += sg[0]; accA[1] += sg[64]; accA[2] += sg[128];
      }
      // ---- record B: products 3..5, running sums over the parts of a piece
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int P = 3; P < 6; ++P)
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) Mg[stage + (P - 3) * 64 + ha * 32 + hb * 4] = d[4 * P + 2 * ha + hb];
      __builtin_amdgcn_wave_barrier();
      for (int k = 0; k < ng; ++k) {
        while (part_obs_begin[partB + 1] <= i0 + g0 + k) {
          if (part_piece[partB + 1] != part_piece[partB]) {  // (the next part is this wave's: partB + 1 < p1 here)
            double* rb = recB + (int64_t)part_piece[partB] * REC_B + lane;
            rb[0] = accB[0]; rb[64] = accB[1]; rb[128] = accB[2];
            accB[0] = accB[1] = accB[2] = 0.0;
          }
          ++partB;
        }
        const double* sg = Mg + k * 192 + lane;
        accB[0] += sg[0]; accB[1] += sg[64]; accB[2] += sg[128];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // the last part of this wave (every part holds at least one relation)
  {
    double* ra = recA + (int64_t)partA * REC_A + lane;
    double* rb = recB + (int64_t)part_piece[partB] * REC_B + lane;
    ra[0] = accA[0]; ra[64] = accA[1]; ra[128] = accA[2];
    rb[0] = accB[0]; rb[64] = accB[1]; rb[128] = accB[2];
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
  if (n_parts <= 0) return;
  const dim3 grid((n_parts + AE_WAVE_PARTS - 1) / AE_WAVE_PARTS);
  // batches of 8 relations (4: 1.49 ms, 16: 1.57 ms, 8: 1.20 ms on config D's relations)
  hipLaunchKernelGGL(ae_records_kernel<8>, grid, dim3(64), 0, s, a, n_parts, part_fr, part_obs_begin, part_piece, recA, recB);
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
