/* calico_rp.h — C ABI of the MI355X reprojection-error LM refinement.
 *
 * This is the drop-in boundary for ONE path of the reference (amy-tabb/calico):
 * Stage 5, the final reprojection-error Levenberg–Marquardt refinement.  The
 * reference has no FFI; the seam is the C++ class CeresProblemClass
 * (src/solving-structure.hpp:611-641) as driven by src/calico.cpp:703-731.  Each
 * entry point below names the reference code it replaces.  All pointers are plain
 * host pointers unless a name ends in `_dev`; no C++ / torch types cross the ABI.
 *
 * Conventions (identical to the reference):
 *  - variable v in [0, vn), vn = cn + pn + tn: cameras [0,cn), patterns
 *    [cn,cn+pn), times [cn+pn,vn)        (multicamera.cpp:261-278);
 *  - parameter vector x = 7*vn doubles, block v at x+7v =
 *    [qw qx qy qz tx ty tz], quaternion NOT normalised, no manifold
 *    (solving-structure.cpp:402,546-548,639-667);
 *  - pattern and time blocks hold the INVERSE transforms (solving-structure.cpp:442-456);
 *  - p_star_var / t_star_var are the variable indices of the exemplar pattern and
 *    time; they are identity and never optimised (solving-structure.cpp:483-484,528-544);
 *  - intrinsics are constants: 12 doubles per camera
 *    [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6]  (solving-structure.cpp:134-159)
 *    (unless calico_rp_set_refine_intrinsics frees them: an extension, see there);
 *  - model points are float32 xyz (patterns.hpp:17, solving-structure.hpp:757-759);
 *  - Jacobian rows are [d/d camera(7) | d/d pattern(7) | d/d time(7)], the block
 *    order of problemRP.AddResidualBlock (solving-structure.cpp:639-667); absent
 *    blocks (p*, t*) are zero.
 *
 * Every function returns 0 on success and a negative CALICO_RP_E* code on error
 * (the reference prints and exit(1)s, e.g. solving-structure.cpp:621-624; a
 * library must not), with text available from calico_rp_last_error().
 * A handle is not re-entrant: one caller thread at a time, as in the reference
 * (calico.cpp is single-threaded around CeresProblemClass).
 */
#ifndef CALICO_RP_H_
#define CALICO_RP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CALICO_RP_OK 0
#define CALICO_RP_EINVAL (-1)   /* bad argument / index out of range          */
#define CALICO_RP_ESTATE (-2)   /* call made in the wrong state               */
#define CALICO_RP_EDEVICE (-3)  /* HIP / RCCL runtime error                   */
#define CALICO_RP_ENUMERIC (-4) /* linear solver failed (non-finite, not PD)  */
#define CALICO_RP_ENOMEM (-5)

typedef struct calico_rp_problem calico_rp_problem;

/* ceres::Solver::Options fields that solving-structure.cpp:973-979 sets, plus the
 * Ceres defaults it relies on.  calico_rp_options_default()
 * fills in exactly those defaults with max_num_iterations = 10 (calico.cpp:709). */
typedef struct calico_rp_options {
  int32_t max_num_iterations;
  int32_t jacobi_scaling;            /* 1 */
  int32_t verbose;                   /* minimizer_progress_to_stdout            */
  int32_t profile;                   /* 1: time every kernel stage with HIP events; 2: the Jacobian /
                                        fused sweep stage only (an event pair per stage costs about
                                        0.1 ms per iteration on 10^8 observations) */
  double initial_trust_region_radius; /* 1e4  */
  double max_trust_region_radius;     /* 1e16 */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3 */
  double min_lm_diagonal;             /* 1e-6 */
  double max_lm_diagonal;             /* 1e32 */
  double function_tolerance;          /* 1e-6 */
  double gradient_tolerance;          /* 1e-10 */
  double parameter_tolerance;         /* 1e-8 */
} calico_rp_options;

enum {
  CALICO_RP_CONVERGENCE = 0,   /* ceres::CONVERGENCE    */
  CALICO_RP_NO_CONVERGENCE = 1, /* ceres::NO_CONVERGENCE (iteration limit) */
  CALICO_RP_FAILURE = 2        /* ceres::FAILURE        */
};

/* Stage timings, milliseconds, summed over the call; filled when options.profile. */
typedef struct calico_rp_stage_ms {
  double jacobian_sweep;  /* residual + analytic Jacobian kernel                */
  double accumulate;      /* per-FR J^T J / J^T r block accumulation; fused mode
                             (where the sweep kernel accumulates itself): the
                             per-variable / per-FR table pre-pass of that kernel   */
  double reduce;          /* camera/pattern block reduction, scaling, damping   */
  double schur;           /* per-time elimination + reduced-system assembly     */
  double cholesky;        /* dense Cholesky + triangular solves                 */
  double backsub;         /* time-block back-substitution + candidate update    */
  double cost_sweep;      /* residual-only sweep at the candidate point         */
  double exchange;        /* RCCL all-reduce                                    */
  int32_t n_jacobian_sweep, n_accumulate, n_reduce, n_schur, n_cholesky, n_backsub,
      n_cost_sweep, n_exchange;
} calico_rp_stage_ms;

/* What ceres::Solver::Summary::BriefReport() prints (solving-structure.cpp:984-991). */
typedef struct calico_rp_summary {
  double initial_cost;
  double final_cost;
  int32_t num_iterations;          /* iteration 0 included, as Ceres counts them */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t termination_type;
  int64_t num_residual_blocks;     /* observations                               */
  double total_ms;                 /* host wall time of the call                 */
  calico_rp_stage_ms stage;
  char message[128];
} calico_rp_summary;

void calico_rp_options_default(calico_rp_options* o);

/* "Ceres Solver Report: Iterations: N, Initial cost: %e, Final cost: %e,
 * Termination: ..." — the line solving-structure.cpp:985-989 writes. */
int calico_rp_brief_report(const calico_rp_summary* s, char* buf, size_t len);

/* Replaces CeresProblemClass::CeresProblemClass (solving-structure.cpp:394-419):
 * fixes the variable layout.  `device` is the HIP device ordinal to run on. */
int calico_rp_create(int32_t cn, int32_t pn, int32_t tn, int32_t p_star_var,
                     int32_t t_star_var, int32_t device, calico_rp_problem** out);

/* Replaces CeresProblemClass::~CeresProblemClass (solving-structure.cpp:1026-1051). */
int calico_rp_destroy(calico_rp_problem* h);

const char* calico_rp_last_error(const calico_rp_problem* h);

/* Replaces the &camera_params[12*camera_index] pointers handed to every cost
 * functor (solving-structure.cpp:565,581,595,609; filled by CopyFromCalibration
 * :134-159).  Copied; may be called again to change intrinsics. */
int calico_rp_set_intrinsics(calico_rp_problem* h, const double* cam_params /*12*cn*/);

/* Replaces CCV[c]->P_class->three_d_points[j] (solving-structure.cpp:559). */
int calico_rp_set_points(calico_rp_problem* h, const float* xyz /*3*n*/, int32_t n_model_pts);

/* Replaces the body of CeresProblemClass::AddEqsToProblemReprojectionError
 * (solving-structure.cpp:506-668): one call appends n residual blocks.  Arrays are
 * borrowed only for the duration of the call.  pat_var / time_var are VARIABLE
 * indices (== p_star_var / t_star_var selects cases 1–3).  Residual blocks
 * accumulate over calls, as they do in problemRP. */
int calico_rp_add_observations(calico_rp_problem* h, int64_t n, const int32_t* cam,
                               const int32_t* pat_var, const int32_t* time_var,
                               const int32_t* corner, const double* u, const double* v);

int64_t calico_rp_num_observations(const calico_rp_problem* h);

/* Parity hook: what one AutoDiffCostFunction::Evaluate per residual block returns
 * (solving-structure.hpp:795-805 etc.), for all blocks at once, in the order the
 * observations were added.  cost = 1/2 sum r^2.  r: 2 per observation; J: 2x21
 * row-major per observation (42 doubles); either may be NULL. */
int calico_rp_eval(calico_rp_problem* h, const double* x, double* cost, double* r, double* J);

/* Parity hook: the damped, (optionally Jacobi-scaled) Schur-reduced normal equations
 * of one LM step at x with trust-region radius `radius`, i.e. what Ceres'
 * SchurEliminator hands to the dense Cholesky.  Reduced unknowns are the ACTIVE
 * camera and pattern blocks in variable order (p* and unobserved blocks skipped):
 * n = 7*n_active.  S is n*n row-major (full symmetric), rhs is n.
 * Pass S = rhs = NULL to query n only. */
int calico_rp_normal_eq(calico_rp_problem* h, const double* x, double radius,
                        int32_t jacobi_scaling, int32_t* n, double* S, double* rhs);

/* Replaces ceres::Solve(options, &problemRP, &summary)
 * (solving-structure.cpp:973-982): LM on all residual blocks added so far,
 * x optimised in place. */
int calico_rp_solve(calico_rp_problem* h, double* x, const calico_rp_options* opt,
                    calico_rp_summary* summary);

/* The same loop, one LM iteration at a time (bench.py times these):
 *   begin: take x, reset radius; nothing is evaluated yet;
 *   step : one LM iteration = [Jacobian sweep + accumulation if x changed] +
 *          damped Schur solve + cost at the candidate (cost sweep, or the speculative
 *          evaluation described at calico_rp_set_speculation) + accept/reject;
 *          *done is set when a Ceres termination test fires;
 *   end  : write x back and fill the summary. */
int calico_rp_lm_begin(calico_rp_problem* h, const double* x, const calico_rp_options* opt);
int calico_rp_lm_step(calico_rp_problem* h, int32_t* accepted, int32_t* done);
int calico_rp_lm_end(calico_rp_problem* h, double* x, calico_rp_summary* summary);

/* Cumulative stage timings (options.profile = 1) over every lm_step since the handle
 * was created or last reset; unlike calico_rp_summary.stage it survives lm_begin. */
int calico_rp_stage_times(calico_rp_problem* h, calico_rp_stage_ms* out, int32_t reset);

/* Reporting on the state x: the per-FR reprojection error of
 * MCcali::ReprojectionError (multicamera.cpp:1860-1891) with
 * CameraCali::ComputeReprojectionErrorOneImagePattern (camera-calibration.cpp:821-933):
 * for every FR (camera, pattern, time) the sum of squared pixel distances of its corners
 * divided by the number of corners, and rrmse = sqrt(sum of those / number of FRs)
 * (multicamera.cpp:717-731).  Computed in fp64 over the observations held by the handle
 * (the reference goes through cv::projectPoints with float32 image points, ~1e-4 px, and
 * uses every detected corner: add them all to a handle to reproduce that number).
 * FRs are reported in the library's order (time, pattern, camera; t* last); the arrays
 * hold calico_rp_num_frs() entries and any of them may be NULL. */
int32_t calico_rp_num_frs(calico_rp_problem* h);
int calico_rp_fr_errors(calico_rp_problem* h, const double* x, int32_t* cam, int32_t* pat_var,
                        int32_t* time_var, int32_t* n_corners, double* mean_sq_err, double* rrmse);

/* calico_rp_fr_errors without any allocation.  No counterpart in the reference; it is for
 * callers that cannot allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc.  (It is NOT legal inside HIP stream capture: it copies from and to the caller's
 * pageable host arrays and synchronises the handle's stream before it returns, and its error
 * paths build messages on the host heap.)  Same arguments, same per-FR numbers (same kernel) as
 * calico_rp_fr_errors, but it calls no hipMalloc / hipFree and builds no host container: the
 * device copy of x, the per-FR means and their sum live in a file-scope device array of the
 * library (18.7 MB per GPU, held until the process ends), which bounds the sizes it takes:
 * at most CALICO_RP_NOALLOC_MAX_VARS variables (cn + pn + tn) and CALICO_RP_NOALLOC_MAX_FRS
 * FRs; beyond that it answers CALICO_RP_EINVAL and names the limit.  The handle's own
 * device data must already be in place (it answers CALICO_RP_ESTATE otherwise): call
 * calico_rp_num_frs, which the caller needs anyway to size the arrays, or any evaluating
 * entry point once after add_observations / set_points / set_intrinsics.  rrmse is summed on
 * the device in a fixed tree order (calico_rp_fr_errors sums on the host: last-bit
 * differences).  All handles on one GPU share that GPU's array, which keeps the last call's
 * contents (every call overwrites what it reads): calls that overlap on one GPU race. */
#define CALICO_RP_NOALLOC_MAX_VARS 32768
#define CALICO_RP_NOALLOC_MAX_FRS (1 << 21)
int calico_rp_fr_errors_noalloc(calico_rp_problem* h, const double* x, int32_t* cam, int32_t* pat_var,
                                int32_t* time_var, int32_t* n_corners, double* mean_sq_err, double* rrmse);

/* How the Jacobian travels from the residual/Jacobian evaluation to the J^T J / J^T r
 * block accumulation.  No counterpart in the reference (Ceres always materialises its
 * block-sparse Jacobian, ceres::Solve at solving-structure.cpp:982); both modes give the
 * same normal equations up to summation order.
 *   FUSED (default)  one kernel evaluates 64 observations per wave into LDS and
 *                    accumulates them on the matrix cores; the Jacobian never reaches HBM.
 *   MATERIALISED     the kernel split SURVEY.md 8(d) prices: a sweep kernel writes
 *                    [ntiles][44][256] doubles (352 B/observation) to HBM, an accumulation
 *                    kernel reads them back.  calico_rp_eval always uses this sweep.
 * Not allowed between lm_begin and lm_end. */
enum { CALICO_RP_JACOBIAN_FUSED = 0, CALICO_RP_JACOBIAN_MATERIALISED = 1 };
int calico_rp_set_jacobian_mode(calico_rp_problem* h, int32_t mode);

/* Speculative evaluation of the candidate point.  No counterpart in the reference: Ceres
 * evaluates the cost alone at x + dx and, once the step is accepted, residuals and Jacobian
 * at the same point again (ceres::Solve, solving-structure.cpp:982).  With speculation on
 * (default; fused mode, reprojection-error handles) lm_step runs the fused kernel at the
 * candidate into a second set of records and takes the candidate's cost from its r^T r; an
 * accepted step then costs no second sweep.  After a rejected or invalid step the cost-only
 * sweep is used until a step is accepted again, and the last iteration the options allow
 * never speculates.  Iterates are the same up to the summation order of the candidate's
 * cost.  Costs a second record set (2.2 GB on 102 M observations).
 * Not allowed between lm_begin and lm_end. */
int calico_rp_set_speculation(calico_rp_problem* h, int32_t on);

/* Camera parameters as a fourth parameter block ("intrinsics + distortion refined",
 * BASELINE.json configs[2]).  EXTENSION: no counterpart in the reference, where the 12 numbers
 * are constants inside the functors (solving-structure.hpp:738-749); with mask = 0 (the
 * default) the library does exactly what it does without this call.  Bit k of mask (k < 12)
 * frees parameter k of every camera's [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6]; the other
 * parameters stay constant (zero Jacobian column, what a ceres::SubsetManifold would do).
 * With a non-zero mask every residual has the block order
 *     AddResidualBlock(cost, NULL, camera, pattern, time, &camera_params[12 * camera])
 * and
 *   - calico_rp_eval returns J as 2 x 33 row-major per observation,
 *     columns [camera 7 | pattern 7 | time 7 | camera parameters 12];
 *   - calico_rp_normal_eq: n = 7 * n_active + 12 * n_active_cameras, the 12-parameter blocks
 *     follow the 7-parameter blocks, in camera order;
 *   - calico_rp_solve / lm_* start from the intrinsics the handle holds (set_intrinsics) and
 *     leave the refined values there: read them with calico_rp_get_intrinsics, which is also
 *     what fr_errors then uses.  x stays 7 * vn doubles.
 * Needs the fused Jacobian mode.  Not allowed between lm_begin and lm_end. */
int calico_rp_set_refine_intrinsics(calico_rp_problem* h, uint32_t mask);
int calico_rp_get_intrinsics(calico_rp_problem* h, double* cam_params /*12*cn*/);

/* ---- Stage 4: algebraic-error refinement on the same machinery ------
 * Replaces CeresProblemClass::AddToProblemAlgebraicError (solving-structure.cpp:673-915)
 * and the ceres::Solve of SolveWriteBackToMCAlgebraicError (solving-structure.cpp:918-935;
 * SPARSE_NORMAL_CHOLESKY there, an exact solve of the same normal equations that this
 * library solves by eliminating the time blocks).  One relation per foundational
 * relationship: residual = rows 0..2 of  C - A P T  (12 numbers, row-major), with the
 * p* / t* factors dropped, i.e. the four functors AlgebraicErrorCam / CamPattern /
 * CamPatternTime / CamTime (solving-structure.hpp:173-462).  "Nothing is inverted"
 * (solving-structure.cpp:750): x holds the 7-vectors of V_initial itself, and A is given
 * as the 7-vector ConvertMatrixTo7QuaternionRepresentationx makes of MC.A[i] (cpp:684).
 *
 * calico_ae_create returns an ordinary handle whose observations are relations:
 * calico_rp_normal_eq / solve / lm_* / stage_times / comm_* / synchronize / destroy work
 * on it unchanged; set_points, add_observations, fr_errors and set_jacobian_mode answer
 * CALICO_RP_ESTATE, set_intrinsics is ignored.  The caller adds the relations whose
 * variables are initialised (the reference adds them incrementally, cpp:732-742); blocks
 * no relation mentions keep their values.
 * calico_ae_eval: cost = 1/2 sum r^2; r [12 n]; J [n][12][21], columns
 * [camera 7 | pattern 7 | time 7], zero for p* / t*; in the caller's relation order. */
int calico_ae_create(int32_t cn, int32_t pn, int32_t tn, int32_t p_star_var, int32_t t_star_var,
                     int32_t device, calico_rp_problem** out);
int calico_ae_add_relations(calico_rp_problem* h, int64_t n, const int32_t* cam,
                            const int32_t* pat_var, const int32_t* time_var,
                            const double* a7 /*7*n: qw qx qy qz tx ty tz of A_i*/);
int calico_ae_eval(calico_rp_problem* h, const double* x, double* cost, double* r, double* J);

/* ---- Point reconstruction ("RAE") -----------------------------------
 * Replaces ReconstructXFunctionIDsMCwithDLTs (solving-structure.cpp:162-390): for every
 * model corner seen in at least two views, a DLT estimate in normalised image coordinates
 * (:309-355) refined by Levenberg-Marquardt on its 3 coordinates with one ReconstructXStruct
 * residual block per view (solving-structure.hpp:1313-1409; Solver::Options defaults with
 * DENSE_QR, cpp:358-362).  All corners are solved together: one sweep over the
 * observations per LM iteration, every corner keeping its own trust region.
 *   cam_params        12 per camera, as calico_rp_set_intrinsics
 *   image_cols_rows   2 per camera: CameraCali::cols, ::rows (cpp:326-327)
 *   fr_cam, fr_a_hat  per view (FR with all variables initialised, cpp:203): its camera and
 *                     rows 0..2 of A_hat = C T^-1 P^-1 (cpp:217), row-major 3x4
 *   obs_*             one entry per detected corner of a view (points_present, cpp:278):
 *                     view index, corner index k in [0, n_points), pixel; grouped by view
 *                     (obs_fr non-decreasing), corner indices increasing inside a view and
 *                     all within one pattern's range [p*corners_per_pattern, (p+1)*...)
 *   opt               NULL = Ceres defaults with 50 iterations; max_num_iterations = 0
 *                     returns the DLT estimates
 *   xyz, has_values   out: 3 per corner (0 where unsolved) and the reference's flag
 *   iterations, final_cost, summary   optional per-corner / overall diagnostics
 * Stateless: device buffers live for the call.  Returns 0 or a CALICO_RP_E* code with the
 * text in err (if given).  Not bit-identical to the reference's DLT (normal equations
 * instead of a pivoted QR; the LM refinement starts from it). */
typedef struct calico_rae_summary {
  int32_t n_points, n_solved, sweeps, max_point_iterations;
  double total_ms, sweep_ms /* average LM sweep kernel time, HIP events */, update_ms;
} calico_rae_summary;
int calico_rae_reconstruct(int32_t device, int32_t cn, const double* cam_params,
                           const int32_t* image_cols_rows, int32_t n_frs, const int32_t* fr_cam,
                           const double* fr_a_hat, int32_t n_points, int32_t corners_per_pattern,
                           int64_t n_obs, const int32_t* obs_fr, const int32_t* obs_point,
                           const double* u, const double* v, const calico_rp_options* opt,
                           double* xyz, uint8_t* has_values, int32_t* iterations,
                           double* final_cost, calico_rae_summary* summary, char* err,
                           size_t err_len);

/* The same with the views and their observations kept on the device between calls: the reference
 * reconstructs the corners once per solution type (multicamera.cpp:832, called from
 * calico.cpp:740 for up to three variable vectors) from the same detections each time, and the
 * stateless call spends its time on checking, regrouping and uploading them (20 bytes per
 * observation), not on its kernels.  calico_rae_create takes what does not change (arguments as
 * above, without cam_params and fr_a_hat) and answers errors in err; calico_rae_solve takes the
 * camera parameters and one A_hat per view (96 bytes per view to upload) and returns what
 * calico_rae_reconstruct returns, which is create + solve + destroy.  Not thread-safe per handle. */
typedef struct calico_rae_problem calico_rae_problem;
int calico_rae_create(int32_t device, int32_t cn, const int32_t* image_cols_rows, int32_t n_frs, const int32_t* fr_cam,
                      int32_t n_points, int32_t corners_per_pattern, int64_t n_obs, const int32_t* obs_fr,
                      const int32_t* obs_point, const double* u, const double* v, calico_rae_problem** out, char* err,
                      size_t err_len);
int calico_rae_solve(calico_rae_problem* h, const double* cam_params, const double* fr_a_hat,
                     const calico_rp_options* options, double* xyz, uint8_t* has_values, int32_t* iterations,
                     double* final_cost, calico_rae_summary* summary);
const char* calico_rae_last_error(const calico_rae_problem* h);
int calico_rae_destroy(calico_rae_problem* h);

/* Block until all work queued by this handle has finished. */
int calico_rp_synchronize(calico_rp_problem* h);

/* ---- multi-GPU: one process per GPU, observations sharded by time index ------
 * No counterpart in the reference (single process, OpenMP only).  Each rank
 * creates a handle with the FULL variable layout, adds only the observations of
 * its own time range, and attaches an RCCL communicator; solve/step then
 * all-reduce the packed reduced normal equations and the cost. */

/* Contiguous time ranges balanced by observation count: boundaries[r] ..
 * boundaries[r+1] is rank r's half-open range of time SUPERSCRIPTS (variable
 * index minus cn+pn).  Pure host function. */
int calico_rp_partition_times(int32_t tn, const int64_t* obs_per_time, int32_t nranks,
                              int32_t* boundaries /*nranks+1*/);

/* What the ranks all-reduce per linear solve, as pure host functions (the device kernels use
 * the same offset functions; multi-process tests without a GPU go through these).  The working
 * layout of a rank is  [P ld x ld row-major | q ld | g_r ld | diag(H_rr) ld]  with
 * ld = calico_rp_exchange_ld(n) (n + 1 rounded up to 128); the exchanged buffer is
 * [lower triangle of P, row i at i (i + 1) / 2 | q n | g_r n | diag n] =
 * calico_rp_exchange_count(n) doubles.  unpack mirrors the triangle. */
int64_t calico_rp_exchange_count(int32_t n);
int32_t calico_rp_exchange_ld(int32_t n);
int calico_rp_exchange_pack_host(int32_t n, const double* work, double* buf);
int calico_rp_exchange_unpack_host(int32_t n, const double* buf, double* work);

/* The reduced system's solver on its own: x = S^-1 b for a symmetric positive definite S
 * (n x n, row-major, lower triangle read), through the blocked Cholesky, the inverse of its
 * factor and the two triangular products every LM iteration runs on the Schur complement of
 * the time blocks.  In the reference this is the dense solve inside Ceres' DENSE_SCHUR
 * (third-party; selected at solving-structure.cpp:975), which no reference file calls by
 * itself: the entry point exists so that the solver can be checked against a known answer at
 * any n (tests/test_gpu_dense_solve.py).  Host buffers; device buffers live for the call.
 * CALICO_RP_ENUMERIC if a pivot is not positive. */
int calico_rp_dense_solve(calico_rp_problem* h, int32_t n, const double* S, const double* b, double* x);

#define CALICO_RP_UNIQUE_ID_BYTES 128
int calico_rp_comm_unique_id(void* id_out /*128 bytes*/);
int calico_rp_comm_init(calico_rp_problem* h, int32_t nranks, int32_t rank,
                        const void* id /*128 bytes, same on all ranks*/);

#ifdef __cplusplus
}
#endif
#endif /* CALICO_RP_H_ */
