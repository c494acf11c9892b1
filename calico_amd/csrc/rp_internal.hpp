// rp_internal.hpp — device data layout and kernel launchers shared by
// rp_kernels.hip / rp_linalg.hip (device side) and rp_api.hip (host side).
//
// Vocabulary: FR = one (camera, pattern, time) triple with its corner
// observations; "segment" = all FRs of one time variable; "reduced" blocks = the
// camera and pattern blocks that survive the Schur elimination of the time blocks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rp {

// ------------------------------------------------------------------ observation sweep
// HBM layout: every per-observation array is SoA over the observations sorted by
// (time, pattern, camera), so that lane i of a wave touches element i:
//   obs_u, obs_v   double[N]     pixel coordinates
//   obs_corner     int32[N]      index into pts (float32 xyz)
//   obs_fr         int32[N]      FR index
//   J              double[ntiles][44][256]  per tile of 256 observations: 42 Jacobian
//                  planes (plane = 21*row + column, columns [camera 7 | pattern 7 |
//                  time 7]) followed by the 2 residual planes.  One workgroup writes
//                  one contiguous 90 KB tile (plane-major over all N measured 15 %
//                  slower: 44 write streams 800 MB apart per workgroup).
struct SweepArgs {
  int64_t n_obs;
  const double* x;      // 7*vn parameter blocks
  const double* intr;   // 12*cn
  const float* pts;     // 3*n_pts
  const double* obs_u;
  const double* obs_v;
  const int32_t* obs_corner;
  const int32_t* obs_fr;
  const int32_t* fr_cam;  // variable index
  const int32_t* fr_pat;  // variable index or -1 (p*)
  const int32_t* fr_tim;  // variable index or -1 (t*)
  double* J;
  const double* a7 = nullptr;  // algebraic-error handles only: the relation's A as a 7-vector, [n_obs][7]
  // fused kernel only: [R | t] per variable (12 doubles, slot n_vars = identity), filled by
  // launch_fused from x before the kernel runs; prep_out is the same buffer, writable
  int n_vars = 0;
  const double* prep = nullptr;
  double* prep_out = nullptr;
  // camera parameters refined (calico_rp_set_refine_intrinsics): bit k frees parameter k of
  // every camera; the fused kernel then also writes record K per part
  unsigned refine_mask = 0;
  double* recK = nullptr;
  // every camera's k4 = k5 = k6 is zero (host check of the intrinsics): the rational model's
  // denominator is 1 and the fused kernel skips its division
  int no_denominator = 0;
  // fused kernel: per-FR table of the chain form ([n_fr][64] doubles, rp_model.hpp:
  // prep_fr_chain), rebuilt by launch_fused_prepass before the kernel runs
  double* frtab = nullptr;
  // fused kernel: sum of r^2 over the observations of every wave, [ceil(n_obs / wave_obs)]
  double* wave_cost = nullptr;
};

constexpr int SWEEP_BLOCK = 256;
constexpr int SWEEP_TILE = 256;          // observations per tile (= workgroup size)
constexpr int SWEEP_PLANES = 44;         // 42 Jacobian entries + 2 residuals
constexpr unsigned SWEEP_MAX_GRID = 1u << 30;  // one workgroup per tile
#ifndef RP_COST_GRID
#define RP_COST_GRID 2048
#endif
constexpr int SWEEP_COST_GRID = RP_COST_GRID; // cost sweep: grid-stride, one partial per workgroup

void launch_jacobian_sweep(const SweepArgs& a, hipStream_t s);
// cost-only sweep: partial[grid] then *cost_out = 1/2 * sum (fixed order)
void launch_cost_sweep(const SweepArgs& a, double* partial /*SWEEP_COST_GRID*/, double* cost_out,
                       hipStream_t s);

// out = scale * sum(partial[0..n)) in a fixed tree order (single workgroup)
// flag != null: also *flag_out = *flag; *flag = 0 (launch_flag_to_scalar folded into this launch);
// seg_partial != null: also seg_scalars[0..2] = the sums BacksubArgs::defer_sum left out
void launch_sum_partials(const double* partial, int n, double scale, double* out, hipStream_t s, int* flag = nullptr,
                         double* flag_out = nullptr, const double* seg_partial = nullptr, int n_seg = 0,
                         double* seg_scalars = nullptr);

// ------------------------------------------------------------------ algebraic error (Stage 4), ae_kernels.hip
// one relation (12 residual rows, C - A P T) per observation slot; records as below
// parts per wave of the records kernel: a piece (record B) is a run of parts of one (time,
// pattern) inside one wave's AE_WAVE_PARTS consecutive parts
constexpr int AE_WAVE_PARTS = 64;
void launch_ae_records(const SweepArgs& a, int n_parts, const int32_t* part_fr, const int64_t* part_obs_begin,
                       const int32_t* part_piece, double* recA, double* recB, hipStream_t s);
void launch_ae_cost(const SweepArgs& a, double* partial /*SWEEP_COST_GRID*/, double* cost_out, hipStream_t s);
void launch_ae_eval(const SweepArgs& a, double* r_out /*12 per relation*/, double* J_out /*12 x 21*/, hipStream_t s);

// per FR: out[f] = sum of r^2 over its observations / number of observations (one wave per FR)
void launch_fr_errors(const SweepArgs& a, int n_fr, const int64_t* fr_obs_begin, double* out, hipStream_t s);

// ------------------------------------------------------------------ block accumulation
// FRs are sorted by (time, pattern, camera).  A wave of the accumulation kernels owns
// acc_wave_obs(N) consecutive observations; a "part" is the intersection of one FR with one
// wave's range, a "piece" a maximal run of consecutive parts inside one wave's range
// that share (time, pattern).  Parts and pieces are numbered in observation order.
// Blocks are 8x8 doubles [a][b] with c~ = [camera Jacobian columns 0..6 | residual],
// p / t = pattern / time columns padded to 8 with a zero column; entry = sum over the
// rows of A[row][a] * B[row][b].
//   record A, one per part:   [ (c~,c~) | (p,c~) | (t,c~) ]
//        (c~,c~)[i][7] = (J_c^T r)_i, (c~,c~)[7][7] = r^T r, (p,c~)[i][7] = (J_p^T r)_i,
//        (t,c~)[i][7] = (J_t^T r)_i
//   record B, one per piece:  [ (p,p) | (t,p) | (t,t) ]   summed over the piece's parts;
//        the padding column of (t,t) carries the piece's gradient: (t,t)[i][7] = (J_t^T r)_i
constexpr int REC_A = 192, REC_B = 192;
// The fused kernel's records hold Gram blocks of the 16 rotation-vector columns U
// (rp_model.hpp: UW_*), not of the 22 reference columns; GatherArgs::wspace tells the
// gathers, which convert the sums.  They live in the same buffers with a shorter stride:
//   record A, one per part:   G[0..15][0..7]  as [16][8]   (rows = all of U, columns = [wc a r | -])
//   record B, one per piece:  G[7..15][6..15] as [9][10]   (rows = [wp b2 wt], columns = [r | wp b2 wt])
constexpr int REC_AW = 128, REC_BW = 96;
// Camera parameters refined: record K, one per part, [12][32] = K^T [U (16) | K (12) | 0 (4)]
// (rows = the camera's 12 parameters).
constexpr int REC_K = 384;
int acc_wave_obs(int64_t n_obs);  // observations per wave: multiple of 64 in [256, 4096]
// from the Jacobian tiles in HBM (the kernel split SURVEY.md 8(d) prices)
void launch_accumulate(int64_t n_obs, int wave_obs, const int32_t* obs_fr, const int64_t* fr_obs_begin,
                       const int32_t* wave_part0, const int32_t* part_piece /*[n_parts+1]*/, const double* J,
                       double* recA, double* recB, hipStream_t s);
// residual + Jacobian + accumulation in one kernel, the Jacobian never leaves the CU (a.J unused);
// launch_fused_prepass first: [R | t] per variable and, with the camera parameters fixed, the
// per-FR table of the chain form, at a.x
void launch_fused_prepass(const SweepArgs& a, int n_fr, hipStream_t s);
void launch_fused(const SweepArgs& a, int wave_obs, int n_fr, const int64_t* fr_obs_begin, const int32_t* wave_part0,
                  const int32_t* part_piece, double* recA, double* recB, hipStream_t s);

// ------------------------------------------------------------------ block gathers
struct GatherArgs {
  const double* recA;
  const double* recB;
  int wspace = 0;                 // 1: records in rotation-vector space (fused kernel), converted here
  const double* x = nullptr;      // wspace: the point the records were made at (7 per variable)
  const int32_t* rvar = nullptr;  // wspace: reduced block -> variable index
  const int32_t* seg_var = nullptr;  // wspace: segment -> time variable index
  int cost_index = 63;            // entry of a group sum that holds r^T r
  // (camera, pattern) groups of FRs for the reduced-block sums
  int n_chunks;                 // chunks of <= PAIR_CHUNK parts
  const int32_t* chunk_begin;   // [n_chunks+1] into pair_part
  const int32_t* pair_part;     // part ids sorted by (reduced camera, reduced pattern)
  double* chunk_sum;            // [n_chunks][128]  c~c~ | p c~
  int n_groups;
  const int32_t* group_chunk_begin;  // [n_groups+1]
  const int32_t* group_rc;      // reduced block index of the camera
  const int32_t* group_rp;      // reduced block index of the pattern or -1
  double* group_sum;            // [n_groups][128]
  // pattern diagonal blocks from the pieces
  int n_pp_chunks;
  const int32_t* pp_chunk_begin;         // [n_pp_chunks+1] into pp_piece
  const int32_t* pp_piece;               // piece ids sorted by reduced pattern block
  double* pp_chunk_sum;                  // [n_pp_chunks][64]
  const int32_t* pp_rblock_chunk_begin;  // [n_rblocks+1] chunk range per reduced block (empty for cameras)
  double* pp_sum;                        // [n_rblocks][64]
  // reduced blocks -> groups
  int n_rblocks;
  const int32_t* rblock_group_begin;  // [n_rblocks+1]
  const int32_t* rblock_groups;       // group ids touching the block
  int n;                        // 7*n_rblocks
  int ldn;                      // leading dimension (>= n+1, multiple of 128)
  double* Hrr;                  // [ldn][ldn] row-major, full symmetric, rows/cols >= n zero
  double* gr;                   // [n]
  double* hdiag;                // [n] diagonal of Hrr
  double* cost_sum;             // scalar: cost_scale * sum r^T r over all FRs; null: not wanted
  double cost_scale = 1.0;
  // time segments
  int n_seg;                    // eliminated segments
  const int32_t* seg_part_begin;   // [n_seg+1] part range (parts are time-sorted)
  const int32_t* seg_piece_begin;  // [n_seg+1] piece range
  double* Ett;                  // [n_seg][49]
  double* gt;                   // [n_seg][7]
  int n_units;                  // (segment, reduced block) pairs
  const int32_t* unit_seg;
  const int32_t* seg_unit_begin;  // [n_seg+1] unit range per segment (units are sorted by segment)
  const int32_t* unit_rblock;
  const int32_t* unit_begin;    // [n_units+1] into contrib
  const int32_t* contrib;       // 2*part (camera block: (t,c~) of record A) or 2*piece+1 (pattern block: (t,p) of record B)
  // Camera parameters refined (wspace only).  The 12-parameter block of a camera follows the
  // 7-parameter blocks in the reduced system: block kb < nk belongs to reduced camera block kb
  // (cameras come first in rvar), reference columns n7 + 12 kb, Y / SYRK columns
  // 6 n_rblocks + 12 kb.  Sums of record K run over the chunks / groups / units above.
  int nk = 0;
  int n7 = 0;                   // 7 * n_rblocks
  const double* recK = nullptr;
  double* chunkK_sum = nullptr; // [n_chunks][REC_K]
  double* groupK_sum = nullptr; // [n_groups][REC_K]
};

// Entry of the extended parameter vector [7 per variable | 12 per camera] behind the reduced
// unknown i: 7-parameter blocks first (n7 = 7 * n_rblocks), then the camera-parameter blocks.
__device__ __forceinline__ int reduced_slot(int i, int n7, int vn7, const int32_t* rvar) {
  if (i < n7) return 7 * rvar[i / 7] + (i % 7);
  const int k = i - n7;
  return vn7 + 12 * rvar[k / 12] + (k % 12);
}
constexpr int PAIR_CHUNK = 64;
// GATHER_SEG: per-segment E_tt / g_t (record B of the segment's pieces);
// GATHER_PAIRS: record A -> chunk_sum -> group_sum (and, if cost_sum is set, the cost from the
// group sums); GATHER_BLOCKS: pattern diagonals, H_rr / g_r / diag, camera-parameter blocks (reads
// group_sum, so GATHER_PAIRS of the same records must have run before).
enum { GATHER_PAIRS = 1, GATHER_BLOCKS = 2, GATHER_SEG = 4, GATHER_ALL = 7 };
void launch_gathers(const GatherArgs& g, hipStream_t s, int what = GATHER_ALL);

// ------------------------------------------------------------------ time elimination
constexpr int SEGW = 96, SEGW_R = 42, SEGW_Z = 78;
constexpr int SEGP = 12, SEGP_V = 4;  // BacksubArgs::seg_partial
struct ElimArgs {
  int n_seg, n, ldn, n_krows;   // n_krows = padded 7*n_seg (multiple of 16)
  const int32_t* seg_var;       // time variable index per segment
  const double* Ett;
  const double* gt;
  double* scale;                // [7*vn] Jacobi column scale (in/out)
  double* diag;                 // [7*vn] clamped squared column norms of the scaled J
  double* gscaled;              // [7*vn] scaled gradient (time part written here)
  double* Lt;                   // [n_seg][49] lower Cholesky factors of the damped blocks
  double* z;                    // [n_seg][7]
  double* Y;                    // [n_krows][ldn]; column n holds z
  int* fail;                    // set non-zero if a block is not positive definite
  // Rotation-vector records (GatherArgs::wspace): the 7 rows of a segment's Y = W M have rank 6
  // (W = L^-1 s T^T is 7 x 6, M the coupling in rotation-vector space), so with the thin QR
  // W = Q R the SYRK, Y and the back-substitution work on the SIX rows R M: 14 % fewer rows.
  // segw: [n_seg][SEGW] = Q (42: [7 m + i]) | R (36, row-major upper) | z6 = Q^T z (6).
  int rows6 = 0;
  double* segw = nullptr;
  const double* x = nullptr;    // rows6: the point the records were made at (7 per variable)
  double radius, min_diag, max_diag;
  int compute_scale;            // 1 on iteration 0 of a solve with Jacobi scaling
  int use_scale;                // 0: scale == 1
  int reuse_diag;               // 1: keep diag (rejected step), only the radius changed
};
// factors the time blocks, then Y = H_rt s_t L^-T straight from the records (g)
struct GatherArgs;
void launch_time_eliminate(const ElimArgs& e, const GatherArgs& g, hipStream_t s);
void launch_unit_y(const GatherArgs& g, const ElimArgs& e, hipStream_t s);
// parity hook, camera parameters refined: out[i][2][12] = d(u,v)/d(camera parameters) of
// observation i (sorted order) at a.x / a.intr
void launch_k_columns(const SweepArgs& a, double* out, hipStream_t s);

// P = Hrr - Y Y^T (lower+upper), including the augmented column/row n:
//   P[i][n] (i<n) = -(Y z)_i ;  q = gr + P[.][n]
struct SyrkArgs {
  int n, ldn, n_krows, ksplit;
  const double* Y;
  const double* Hrr;
  const double* gr;
  double* partial;   // [ksplit][n_tile_pairs][128*128]
  double* P;         // [ldn][ldn]
  double* q;         // [n]
};
int syrk_tile_pairs(int ldn);
void launch_schur_syrk(const SyrkArgs& a, hipStream_t s);
// Rotation-vector records: Y keeps six columns per reduced block ([w | translation], see the
// gathers), so the SYRK runs on nu = 6 * n_rblocks columns (73 % of the flops) into Su =
// -Y^T Y [ldu][ldu] (a: n = nu, ldn = ldu, P = Su, Hrr/gr/q unused), and the product is
// taken to the reference columns block by block:  P = Hrr + T^T Su T,  T = blockdiag[E(q) 0; 0 I].
struct SyrkConvertArgs {
  int n_rblocks, n, ldn, nu, ldu;
  int n7 = 0, nu6 = 0;  // 7 / 6 * n_rblocks: where the camera-parameter blocks start (n7 == n: none)
  const int32_t* rvar;
  const double* x;
  const double* Su;
  const double* Hrr;
  const double* gr;
  double* P;
  double* q;
};
// two launchers: the product needs Y only, the conversion also H_rr / g_r (which the LM loop lets a
// second stream gather while the product runs)
void launch_schur_syrk_w_product(const SyrkArgs& a, hipStream_t s);
void launch_schur_syrk_w_convert(const SyrkConvertArgs& c, hipStream_t s);

// Multi-GPU exchange: [lower triangle of P | q | g_r | diag(H_rr)] of the working buffer
// pack = [P ldn x ldn | q ldn | g_r ldn | diag ldn] to a dense buffer of exchange_count(n)
// doubles and back (unpack mirrors the triangle).
size_t exchange_count(int n);
void launch_exchange_pack(int n, int ldn, const double* pack, double* buf, hipStream_t s);
void launch_exchange_unpack(int n, int ldn, const double* buf, double* pack, hipStream_t s);
void exchange_pack_host(int n, int ldn, const double* pack, double* buf);
void exchange_unpack_host(int n, int ldn, const double* buf, double* pack);

// S = s_r P s_r + D_r^2, rhs = s_r q; also scale/diag/gscaled for the reduced part.
struct ReducedArgs {
  int n, ldn;
  int n7, vn7;              // reduced_slot()
  const int32_t* rvar;      // reduced block -> variable index
  const double* P;
  const double* q;
  const double* gr;         // unscaled gradient (global sum)
  const double* hdiag;      // [n] diag(Hrr) (global sum)
  double* scale;
  double* diag;
  double* gscaled;
  double* S;                // [ldn][ldn]
  double* rhs;              // [n]
  double radius, min_diag, max_diag;
  int compute_scale, use_scale, reuse_diag;
};
void launch_finalize_reduced(const ReducedArgs& a, hipStream_t s);

// Blocked Cholesky (lower) of S[n][n] (ld = ldn; row n of S = rhs^T): of the diagonal blocks'
// factors only the inverses are kept, in invL [ceil(n/64)][64][64], the factor
// below the diagonal blocks (and L^-1 rhs in row n) goes to Lout [ldn][ldn]; then
// solves S y = rhs into rhs.  The upper part of Lout receives the transposed inverse of the
// factor (y = L^-T (L^-1 rhs) is then one product); P: [ldn][ldn] partial sums of that inverse.
void launch_cholesky_solve(int n, int ldn, double* S, double* Lout, double* invL, double* rhs, double* P,
                           int* fail, hipStream_t s);

// Back-substitution of the time blocks and candidate point.
struct BacksubArgs {
  int n, ldn, n_seg, n_rblocks, vn7;
  int n7 = 0;              // 7 * n_rblocks (reduced_slot)
  const int32_t* rvar;
  const int32_t* seg_var;
  const double* yr;        // solution of the reduced system (scaled space)
  const double* Y;
  const double* Lt;
  const double* z;
  const double* scale;
  const double* diag;
  const double* gscaled;
  const double* x;
  double* w;               // [ldn] scratch: s_r * y_r
  double* delta;           // [7*vn]
  double* cand;            // [7*vn]
  double* seg_partial;     // [n_seg][SEGP]: 3 partial scalars at 0, the rows' z - <Y row, w> at SEGP_V
  double* scalars;         // out: [0..2] model_cost_change, |delta|^2, |x|^2 of the reduced blocks,
                           //      [3..5] the same three sums over this rank's time blocks
  double radius;
  int defer_sum = 0;       // 1: scalars[3..5] are summed later, by launch_sum_partials (seg_partial / seg_scalars)
  int wspace = 0;          // 1: Y has six columns per reduced block (ldy, ny); w is mapped to them (wu)
  const double* segw = nullptr;  // wspace: per segment Q | R | z6 (ElimArgs); Y then has SIX rows per segment
  int ldy = 0, ny = 0;
  double* wu = nullptr;    // [ldy] scratch
};
void launch_backsub(const BacksubArgs& a, hipStream_t s);

// max |g| over the unscaled gradient (reduced part gr[n], time part gt[7*n_seg]).
void launch_gradient_max(const double* gr, int n, const double* gt, int nt7, double* out,
                         hipStream_t s);

// sharded solves: merge of the time blocks at the end (own / owners: one double per time variable)
void launch_time_blocks_pack(const double* x, const double* own, int t0, int tn, double* tsum, hipStream_t s);
void launch_time_blocks_unpack(double* x, const double* owners, int t0, int tn, const double* tsum, hipStream_t s);

void launch_flag_to_scalar(int* flag, double* out, hipStream_t s);  // *out = *flag; *flag = 0

// Observations into FR order on the device, run by run: run k = len[k] elements of
// `words` 32-bit words each, from element src_begin[k] of src to element dst_begin[k] of dst.
void launch_move_runs(int64_t n_runs, const int64_t* src_begin, const int64_t* dst_begin, const int64_t* len,
                      const uint32_t* src, uint32_t* dst, int words, hipStream_t s);
// obs_fr[i] = f for fr_obs_begin[f] <= i < fr_obs_begin[f + 1]
void launch_fill_obs_fr(int n_fr, const int64_t* fr_obs_begin, int32_t* obs_fr, hipStream_t s);

}  // namespace rp
