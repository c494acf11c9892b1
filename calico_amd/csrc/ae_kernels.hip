// ae_kernels.hip — Stage-4 algebraic-error refinement on the machinery of
// the reprojection path: the same block structure (one camera, pattern and time 7-vector
// per FR; p* / t* fixed), hence the same records, gathers, time-block elimination, Schur
// SYRK, dense Cholesky and LM loop.  Only what turns parameters into records differs:
// 12 residual rows per relation C - A P T (ae_model.hpp) instead of 2 per corner.
//   ae_records_kernel : per part, [J r]^T [J r] -> record A / record B (one wave per part)
//   ae_cost_kernel    : 1/2 sum r^2 at the candidate point
//   ae_eval_kernel    : r and J of every relation (parity hook calico_ae_eval)
// The data volume is 1/85 of the corner sweep (one relation per FR), so these kernels are
// written for clarity: nothing here is on the critical path of BASELINE.json's metric.
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

constexpr int AE_LD = 22;  // [J(21) | r] per residual row

}  // namespace

// One wave per part (= FR x wave range; with one relation per FR: one relation).  Lane 0
// evaluates r and J of each relation of the part into LDS; all lanes then add the Gram
// entries they own.  Lane l owns record slots l, l+64, ... of [A (192) | B (192)].
__global__ __launch_bounds__(256) void ae_records_kernel(SweepArgs a, int n_parts, const int32_t* __restrict__ part_fr,
                                                         const int64_t* __restrict__ part_obs_begin,
                                                         const int32_t* __restrict__ part_piece, double* __restrict__ recA,
                                                         double* __restrict__ recB) {
  __shared__ double sM[4][AE_ROWS * AE_LD];
  __shared__ double sJ[4][AE_ROWS * 21];
  __shared__ double sR[4][AE_ROWS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int part = blockIdx.x * 4 + wave;
  if (part >= n_parts) return;
  const int f = part_fr[part];
  // slot -> (column index into [c(7) r p(7) t(7)] for A-operand, same for B-operand)
  // record A: [c~c~ 8x8 | p c~ 8x8 | t c~ 8x8], record B: [p p | t p | t t] with (t,t)[a][7] = t_a . r
  auto col_of = [](int blk, int k) -> int {  // blk 0: c~ (k 0..7: 7 = r), 1: p, 2: t; k = 7 of p/t -> none
    if (blk == 0) return k < 7 ? k : 21;
    if (k >= 7) return -1;
    return blk == 1 ? 7 + k : 14 + k;
  };
  int ca[6], cb[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int slot = lane + 64 * s;           // 0..383
    const int rec = slot / 192, w = slot % 192, pair = w / 64, aa = (w % 64) / 8, bb = w % 8;
    int ba, bbk;
    if (rec == 0) {  // A: rows blk (c~, p, t) x cols c~
      ba = pair;
      bbk = 0;
    } else {  // B: pp, tp, tt
      ba = pair == 0 ? 1 : 2;
      bbk = pair == 2 ? 2 : 1;
    }
    ca[s] = col_of(ba, aa);
    cb[s] = col_of(bbk, bb);
    if (rec == 1 && pair == 2 && bb == 7 && aa < 7) cb[s] = 21;  // (t,t)[a][7] = J_t^T r
  }
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t i = part_obs_begin[part]; i < part_obs_begin[part + 1]; ++i) {
    if (lane == 0) {
      double blk[4 * PREP_SIZE];
      ae_prepare(a, f, i, blk);
      ae_eval_prepared(blk, blk + PREP_SIZE, blk + 2 * PREP_SIZE, blk + 3 * PREP_SIZE, sR[wave], sJ[wave]);
      for (int rr = 0; rr < AE_ROWS; ++rr) {
        for (int k = 0; k < 21; ++k) sM[wave][rr * AE_LD + k] = sJ[wave][rr * 21 + k];
        sM[wave][rr * AE_LD + 21] = sR[wave][rr];
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if (ca[s] >= 0 && cb[s] >= 0) {
        double t = 0.0;
        for (int rr = 0; rr < AE_ROWS; ++rr) t += sM[wave][rr * AE_LD + ca[s]] * sM[wave][rr * AE_LD + cb[s]];
        acc[s] += t;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  const int piece = part_piece[part];  // AE handles have one piece per part
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int slot = lane + 64 * s;
    if (slot < 192) recA[(int64_t)part * REC_A + slot] = acc[s];
    else recB[(int64_t)piece * REC_B + slot - 192] = acc[s];
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
  if (n_parts > 0)
    hipLaunchKernelGGL(ae_records_kernel, dim3((n_parts + 3) / 4), dim3(256), 0, s, a, n_parts, part_fr, part_obs_begin,
                       part_piece, recA, recB);
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
