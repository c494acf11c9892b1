// rp_model.hpp — closed-form reprojection residual and Jacobian for one corner
// observation (the arithmetic inside the Jacobian-sweep and cost-sweep kernels).
//
// Replaces, for the GPU path, the four autodiff cost functors of the reference
// (/root/reference/src/solving-structure.hpp:643-817 case 0, :984-1147 p == p*,
// :1150-1310 t == t*, :820-982 both) together with
// Convert7ParameterQuaternionRepresentationIntoMatrix (:32-58) and the two
// MatrixMultiply calls (:733-734, :763).  Instead of pushing 21-wide dual numbers
// through 4x4 matrix products, the chain
//     z = R(q_pi) X + t_pi,   y = R(q_tau) z + t_tau,   Xc = R(q_c) y + t_c
// (A_hat = C * T * P, hpp:733-734; pattern/time blocks hold inverses) is
// differentiated by hand:
//     d(R(q) y)/dq = (2/|q|^2) * ( S(q,y) - (R y) q^T ),
//     S = [ s0  s3  s2 -s1 ;  s1 -s2  s3  s0 ;  s2  s1 -s0  s3 ],
//     s0 = a y0 - d y1 + c y2,  s1 = d y0 + a y1 - b y2,
//     s2 = -c y0 + b y1 + a y2, s3 = b y0 + c y1 + d y2,      q = (a,b,c,d)
// which includes the derivative of the 1/|q|^2 normalisation that
// ceres::QuaternionToRotation applies (quaternions are NOT unit, no manifold:
// solving-structure.cpp:639-667).  Distortion / projection follow hpp:765-777.
//
// The work is split in two so that the kernels can do the per-block part once per
// FR instead of once per corner:
//     prep_block()      7 parameters -> rotation matrix, 1/|q|^2   (per FR)
//     eval_prepared()   everything that depends on the corner      (per observation)
// The header is plain C++ so that tests can run the very same code on the host
// against the dual-number oracle (tests/helpers/model_host.cpp); the product only
// ever calls it from device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RP_HD __host__ __device__ __forceinline__
#else
#define RP_HD inline
#endif

// Keeps hipcc from hoisting every LDS read of the prepared FR to the top of
// eval_prepared (which costs 240+ VGPRs): reads stay next to the stage that uses them.
#if defined(__HIP_DEVICE_COMPILE__)
#define RP_STAGE_FENCE()              \
  do {                                \
    asm volatile("" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
#else
#define RP_STAGE_FENCE() ((void)0)
#endif

namespace rp {

// One prepared 7-parameter block: 20 doubles.
//   [0..8] R row-major, [9..11] t, [12..15] q, [16] 1/|q|^2, [17] present (1.0/0.0)
constexpr int PREP_R = 0, PREP_T = 9, PREP_Q = 12, PREP_INV = 16, PREP_ON = 17, PREP_SIZE = 20;
// One prepared FR: camera, time, pattern blocks + 12 intrinsics = 72 doubles.
constexpr int FRP_CAM = 0, FRP_TIM = 20, FRP_PAT = 40, FRP_K = 60, FRP_SIZE = 72;

RP_HD void prep_block(const double* p7, double* out) {
  if (p7) {
    const double a = p7[0], b = p7[1], c = p7[2], d = p7[3];
    const double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    const double inv = 1.0 / (aa + bb + cc + dd);
    out[0] = (aa + bb - cc - dd) * inv;
    out[1] = 2.0 * (b * c - a * d) * inv;
    out[2] = 2.0 * (a * c + b * d) * inv;
    out[3] = 2.0 * (a * d + b * c) * inv;
    out[4] = (aa - bb + cc - dd) * inv;
    out[5] = 2.0 * (c * d - a * b) * inv;
    out[6] = 2.0 * (b * d - a * c) * inv;
    out[7] = 2.0 * (a * b + c * d) * inv;
    out[8] = (aa - bb - cc + dd) * inv;
    out[9] = p7[4]; out[10] = p7[5]; out[11] = p7[6];
    out[12] = a; out[13] = b; out[14] = c; out[15] = d;
    out[16] = inv;
    out[17] = 1.0;
  } else {  // identity factor (p* / t*): hpp:861-874, :1028-1038, :1197-1204
    out[0] = 1.0; out[1] = 0.0; out[2] = 0.0;
    out[3] = 0.0; out[4] = 1.0; out[5] = 0.0;
    out[6] = 0.0; out[7] = 0.0; out[8] = 1.0;
    out[9] = 0.0; out[10] = 0.0; out[11] = 0.0;
    out[12] = 1.0; out[13] = 0.0; out[14] = 0.0; out[15] = 0.0;
    out[16] = 1.0;
    out[17] = 0.0;
  }
  out[18] = 0.0;
  out[19] = 0.0;
}

RP_HD void rot_apply(const double* R, const double* y, double* w) {
  w[0] = R[0] * y[0] + R[1] * y[1] + R[2] * y[2];
  w[1] = R[3] * y[0] + R[4] * y[1] + R[5] * y[2];
  w[2] = R[6] * y[0] + R[7] * y[1] + R[8] * y[2];
}

// B (2x3) times d(R(q) y)/dq (3x4); w = R y (rotation only).  Emits columns
// col0..col0+3 of both Jacobian rows, then B itself as the translation columns.
template <typename Out>
RP_HD void jac_block(const double B[6], const double* blk, const double* y, const double* w, int col0,
                     Out& out) {
  const double a = blk[PREP_Q], b = blk[PREP_Q + 1], c = blk[PREP_Q + 2], d = blk[PREP_Q + 3];
  const double s0 = a * y[0] - d * y[1] + c * y[2];
  const double s1 = d * y[0] + a * y[1] - b * y[2];
  const double s2 = -c * y[0] + b * y[1] + a * y[2];
  const double s3 = b * y[0] + c * y[1] + d * y[2];
  const double k = 2.0 * blk[PREP_INV];
  const double Bw0 = B[0] * w[0] + B[1] * w[1] + B[2] * w[2];
  const double Bw1 = B[3] * w[0] + B[4] * w[1] + B[5] * w[2];
  // columns of S: a:(s0,s1,s2) b:(s3,-s2,s1) c:(s2,s3,-s0) d:(-s1,s0,s3)
  out(col0 + 0, k * (B[0] * s0 + B[1] * s1 + B[2] * s2 - Bw0 * a));
  out(col0 + 1, k * (B[0] * s3 - B[1] * s2 + B[2] * s1 - Bw0 * b));
  out(col0 + 2, k * (B[0] * s2 + B[1] * s3 - B[2] * s0 - Bw0 * c));
  out(col0 + 3, k * (-B[0] * s1 + B[1] * s0 + B[2] * s3 - Bw0 * d));
  out(col0 + 4, B[0]);
  out(col0 + 5, B[1]);
  out(col0 + 6, B[2]);
  out(21 + col0 + 0, k * (B[3] * s0 + B[4] * s1 + B[5] * s2 - Bw1 * a));
  out(21 + col0 + 1, k * (B[3] * s3 - B[4] * s2 + B[5] * s1 - Bw1 * b));
  out(21 + col0 + 2, k * (B[3] * s2 + B[4] * s3 - B[5] * s0 - Bw1 * c));
  out(21 + col0 + 3, k * (-B[3] * s1 + B[4] * s0 + B[5] * s3 - Bw1 * d));
  out(21 + col0 + 4, B[3]);
  out(21 + col0 + 5, B[4]);
  out(21 + col0 + 6, B[5]);
}

template <typename Out>
RP_HD void zero_block(int col0, Out& out) {
  for (int k = 0; k < 7; ++k) {
    out(col0 + k, 0.0);
    out(21 + col0 + k, 0.0);
  }
}

// One observation from a prepared FR (layout FRP_*).
//   r[2]    residual (predicted - observed), weight 1 (hpp:677-683)
//   out(k, value): Jacobian entry k = 21*row + column, columns [camera | pattern | time]
template <bool WITH_J, typename Out>
RP_HD void eval_prepared(const double* frp, const double X[3], double u, double v, double* r, Out& out) {
  const double* C = frp + FRP_CAM;
  const double* T = frp + FRP_TIM;
  const double* P = frp + FRP_PAT;
  const double* K = frp + FRP_K;
  double wp[3], z[3], wt[3], y[3], wc[3], xc[3];
  rot_apply(P + PREP_R, X, wp);
  z[0] = wp[0] + P[PREP_T]; z[1] = wp[1] + P[PREP_T + 1]; z[2] = wp[2] + P[PREP_T + 2];
  rot_apply(T + PREP_R, z, wt);
  y[0] = wt[0] + T[PREP_T]; y[1] = wt[1] + T[PREP_T + 1]; y[2] = wt[2] + T[PREP_T + 2];
  rot_apply(C + PREP_R, y, wc);
  xc[0] = wc[0] + C[PREP_T]; xc[1] = wc[1] + C[PREP_T + 1]; xc[2] = wc[2] + C[PREP_T + 2];

  const double fx = K[0], cx = K[1], fy = K[2], cy = K[3];
  const double k1 = K[4], k2 = K[5], p1 = K[6], p2 = K[7], k3 = K[8], k4 = K[9], k5 = K[10], k6 = K[11];
  const double iz = 1.0 / xc[2];
  const double xn = xc[0] * iz, yn = xc[1] * iz;
  const double r2 = xn * xn + yn * yn;
  const double numer = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;   // hpp:770
  const double denom = 1.0 + k4 * r2 + k5 * r2 * r2 + k6 * r2 * r2 * r2;   // hpp:771
  const double idn = 1.0 / denom;
  const double xpp = xn * numer * idn + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);  // :773
  const double ypp = yn * numer * idn + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);  // :774
  r[0] = xpp * fx + cx - u;  // :776,783
  r[1] = ypp * fy + cy - v;  // :777,784
  if (!WITH_J) return;
  RP_STAGE_FENCE();

  const double ratio = numer * idn;
  const double dnum = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r2 * r2;
  const double dden = k4 + 2.0 * k5 * r2 + 3.0 * k6 * r2 * r2;
  const double dratio = (dnum - ratio * dden) * idn;  // d(numer/denom)/d(r2)
  const double dxx = ratio + 2.0 * xn * xn * dratio + 2.0 * p1 * yn + 6.0 * p2 * xn;
  const double dxy = 2.0 * xn * yn * dratio + 2.0 * p1 * xn + 2.0 * p2 * yn;
  const double dyy = ratio + 2.0 * yn * yn * dratio + 2.0 * p2 * xn + 6.0 * p1 * yn;
  // A = d(u,v)/d(Xc), 2x3
  double A[6];
  const double fz = fx * iz, gz = fy * iz;
  A[0] = fz * dxx; A[1] = fz * dxy; A[2] = -fz * (dxx * xn + dxy * yn);
  A[3] = gz * dxy; A[4] = gz * dyy; A[5] = -gz * (dxy * xn + dyy * yn);

  RP_STAGE_FENCE();
  jac_block(A, C, y, wc, 0, out);  // camera block: [A * dq(Rc y) | A]
  RP_STAGE_FENCE();

  double B1[6], B2[6];  // B1 = A Rc, B2 = B1 Rt
  const double* Rc = C + PREP_R;
  B1[0] = A[0] * Rc[0] + A[1] * Rc[3] + A[2] * Rc[6];
  B1[1] = A[0] * Rc[1] + A[1] * Rc[4] + A[2] * Rc[7];
  B1[2] = A[0] * Rc[2] + A[1] * Rc[5] + A[2] * Rc[8];
  B1[3] = A[3] * Rc[0] + A[4] * Rc[3] + A[5] * Rc[6];
  B1[4] = A[3] * Rc[1] + A[4] * Rc[4] + A[5] * Rc[7];
  B1[5] = A[3] * Rc[2] + A[4] * Rc[5] + A[5] * Rc[8];
  RP_STAGE_FENCE();
  if (T[PREP_ON] != 0.0) jac_block(B1, T, z, wt, 14, out); else zero_block(14, out);
  RP_STAGE_FENCE();
  const double* Rt = T + PREP_R;
  B2[0] = B1[0] * Rt[0] + B1[1] * Rt[3] + B1[2] * Rt[6];
  B2[1] = B1[0] * Rt[1] + B1[1] * Rt[4] + B1[2] * Rt[7];
  B2[2] = B1[0] * Rt[2] + B1[1] * Rt[5] + B1[2] * Rt[8];
  B2[3] = B1[3] * Rt[0] + B1[4] * Rt[3] + B1[5] * Rt[6];
  B2[4] = B1[3] * Rt[1] + B1[4] * Rt[4] + B1[5] * Rt[7];
  B2[5] = B1[3] * Rt[2] + B1[4] * Rt[5] + B1[5] * Rt[8];
  RP_STAGE_FENCE();
  if (P[PREP_ON] != 0.0) jac_block(B2, P, X, wp, 7, out); else zero_block(7, out);
}

// ---- camera parameters as a fourth parameter block ("intrinsics + distortion refined") ----
// Extension asked for by BASELINE.json configs[2]; in the reference the 12 numbers are
// constants inside the functor (hpp:738-749).  Derivative of the two residual rows with
// respect to K = [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6] (the formulas of hpp:765-777):
//     u = fx x'' + cx,  v = fy y'' + cy,
//     x'' = xn numer/denom + 2 p1 xn yn + p2 (r2 + 2 xn^2),  y'' likewise with p1 <-> p2
// out(base + 12 * row + k, value); parameters whose bit in `mask` is clear are held
// constant (zero column).  The arguments are the intermediate values of the projection
// (eval_prepared / eval_prepared_w have them at hand).
constexpr int KP = 12;
template <typename Out>
RP_HD void intrinsics_columns(double fx, double fy, double xn, double yn, double r2, double numer, double idn,
                              double xpp, double ypp, unsigned mask, int base, Out& out) {
  const double r4 = r2 * r2, r6 = r4 * r2;
  const double q = -numer * idn * idn;  // d(numer/denom)/d(denom)
  const double fxn = fx * xn, fyn = fy * yn;
  const double cu[KP] = {xpp, 1.0, 0.0, 0.0, fxn * r2 * idn, fxn * r4 * idn, fx * 2.0 * xn * yn,
                         fx * (r2 + 2.0 * xn * xn), fxn * r6 * idn, fxn * q * r2, fxn * q * r4, fxn * q * r6};
  const double cv[KP] = {0.0, 0.0, ypp, 1.0, fyn * r2 * idn, fyn * r4 * idn, fy * (r2 + 2.0 * yn * yn),
                         fy * 2.0 * xn * yn, fyn * r6 * idn, fyn * q * r2, fyn * q * r4, fyn * q * r6};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < KP; ++k) {
    const bool on = (mask >> k) & 1u;
    out(base + k, on ? cu[k] : 0.0);
    out(base + KP + k, on ? cv[k] : 0.0);
  }
}

// The same columns from a prepared FR (parity hook kernel, host tests): out(12 * row + k).
template <typename Out>
RP_HD void eval_intrinsics_columns(const double* frp, const double X[3], unsigned mask, Out& out) {
  const double* C = frp + FRP_CAM;
  const double* T = frp + FRP_TIM;
  const double* P = frp + FRP_PAT;
  const double* K = frp + FRP_K;
  double wp[3], z[3], wt[3], y[3], wc[3], xc[3];
  rot_apply(P + PREP_R, X, wp);
  z[0] = wp[0] + P[PREP_T]; z[1] = wp[1] + P[PREP_T + 1]; z[2] = wp[2] + P[PREP_T + 2];
  rot_apply(T + PREP_R, z, wt);
  y[0] = wt[0] + T[PREP_T]; y[1] = wt[1] + T[PREP_T + 1]; y[2] = wt[2] + T[PREP_T + 2];
  rot_apply(C + PREP_R, y, wc);
  xc[0] = wc[0] + C[PREP_T]; xc[1] = wc[1] + C[PREP_T + 1]; xc[2] = wc[2] + C[PREP_T + 2];
  const double fx = K[0], fy = K[2];
  const double k1 = K[4], k2 = K[5], p1 = K[6], p2 = K[7], k3 = K[8], k4 = K[9], k5 = K[10], k6 = K[11];
  const double iz = 1.0 / xc[2];
  const double xn = xc[0] * iz, yn = xc[1] * iz;
  const double r2 = xn * xn + yn * yn;
  const double numer = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
  const double denom = 1.0 + k4 * r2 + k5 * r2 * r2 + k6 * r2 * r2 * r2;
  const double idn = 1.0 / denom;
  const double xpp = xn * numer * idn + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
  const double ypp = yn * numer * idn + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);
  intrinsics_columns(fx, fy, xn, yn, r2, numer, idn, xpp, ypp, mask, 0, out);
}

// ---- rotation-vector columns (the fused kernel's Gram space) --------------------------
// A 7-parameter block enters the chain only through R(q) and t, and a change of q moves
// R(q) by a left rotation  exp([dw]x) R(q)  with
//     dw = E(q) dq,   E(q) = (2/|q|^2) [ -qv | a I + [qv]x ]      (3 x 4, q = (a, qv)),
// so  d(b . R(q) y)/dq = (w x b)^T E(q),  w = R(q) y:  the four quaternion columns of a block
// are FR-constant mixes of the THREE columns  w x b  (they span the same space: J_q q = 0).
// With b1 = A Rc, b2 = b1 Rt the translation columns are  J_tc = A, J_tt = b1 = b2 Rt^T,
// J_tp = b2.  eval_prepared_w emits, per residual row, the 16 columns
//     U = [ wc x a (3) | a (3) | r | wp x b2 (3) | b2 (3) | wt x b1 (3) ]
// as out(16 * row + column); the 22 columns of [J_c | r | J_p | J_t] are U times an
// FR-constant matrix that the gathers apply once per SUM of Gram blocks (per (camera,
// pattern) group, per time segment, per (segment, block) unit) instead of once per corner.
// Nothing here depends on whether the pattern / time factor is the fixed exemplar: its
// prepared block is the identity and its columns are simply never gathered.
constexpr int UW_WC = 0, UW_A = 3, UW_R = 6, UW_WP = 7, UW_B2 = 10, UW_WT = 13, UW_COLS = 16;

// E(q) of a raw 7-parameter block, row-major 3 x 4
RP_HD void quat_rotvec_map(const double* p7, double* E) {
  const double a = p7[0], b = p7[1], c = p7[2], d = p7[3];
  const double k = 2.0 / (a * a + b * b + c * c + d * d);
  E[0] = -k * b; E[1] = k * a;  E[2] = -k * d; E[3] = k * c;
  E[4] = -k * c; E[5] = k * d;  E[6] = k * a;  E[7] = -k * b;
  E[8] = -k * d; E[9] = -k * c; E[10] = k * b; E[11] = k * a;
}

// Map of a 7-parameter block from its six rotation-vector columns [w (3) | translation (3)] to
// its seven parameters (the gathers of the fused kernel's records convert sums with it).
// T[i][j], 6 x 7 row-major, of variable v; with_rt: the time block's translation mix
RP_HD void block_map(const double* x, int v, bool with_rt, double* T) {
  double E[12];
  quat_rotvec_map(x + 7 * (int64_t)v, E);
  for (int i = 0; i < 42; ++i) T[i] = 0.0;
  for (int m = 0; m < 3; ++m)
    for (int j = 0; j < 4; ++j) T[7 * m + j] = E[4 * m + j];
  if (with_rt) {
    double blk[PREP_SIZE];
    prep_block(x + 7 * (int64_t)v, blk);
    for (int m = 0; m < 3; ++m)
      for (int j = 0; j < 3; ++j) T[7 * (3 + m) + 4 + j] = blk[PREP_R + 3 * j + m];  // b1 = b2 Rt^T
  } else {
    for (int m = 0; m < 3; ++m) T[7 * (3 + m) + 4 + m] = 1.0;
  }
}

// WITH_K (camera parameters refined): also the 2 x 12 columns of intrinsics_columns as
// out(32 + 12 * row + k).
// NODEN: the caller guarantees k4 = k5 = k6 = 0 (what CopyFromCalibration hands over: OpenCV
// returns five distortion terms, solving-structure.cpp:134-159), so denom = 1 exactly and its
// division drops out; the results are bit-identical to the general path (x * 1.0, x - y * 0.0).
template <bool WITH_K = false, bool NODEN = false, typename Out>
RP_HD void eval_prepared_w(const double* frp, const double X[3], double u, double v, double* r, Out& out,
                           unsigned kmask = 0) {
  const double* C = frp + FRP_CAM;
  const double* T = frp + FRP_TIM;
  const double* P = frp + FRP_PAT;
  const double* K = frp + FRP_K;
  double wp[3], z[3], wt[3], y[3], wc[3], xc[3];
  rot_apply(P + PREP_R, X, wp);
  z[0] = wp[0] + P[PREP_T]; z[1] = wp[1] + P[PREP_T + 1]; z[2] = wp[2] + P[PREP_T + 2];
  rot_apply(T + PREP_R, z, wt);
  y[0] = wt[0] + T[PREP_T]; y[1] = wt[1] + T[PREP_T + 1]; y[2] = wt[2] + T[PREP_T + 2];
  rot_apply(C + PREP_R, y, wc);
  xc[0] = wc[0] + C[PREP_T]; xc[1] = wc[1] + C[PREP_T + 1]; xc[2] = wc[2] + C[PREP_T + 2];

  const double fx = K[0], cx = K[1], fy = K[2], cy = K[3];
  const double k1 = K[4], k2 = K[5], p1 = K[6], p2 = K[7], k3 = K[8], k4 = K[9], k5 = K[10], k6 = K[11];
  const double iz = 1.0 / xc[2];
  const double xn = xc[0] * iz, yn = xc[1] * iz;
  const double r2 = xn * xn + yn * yn;
  const double numer = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;   // hpp:770
  const double denom = NODEN ? 1.0 : 1.0 + k4 * r2 + k5 * r2 * r2 + k6 * r2 * r2 * r2;   // hpp:771
  const double idn = NODEN ? 1.0 : 1.0 / denom;
  const double xpp = xn * numer * idn + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);  // :773
  const double ypp = yn * numer * idn + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);  // :774
  r[0] = xpp * fx + cx - u;  // :776,783
  r[1] = ypp * fy + cy - v;  // :777,784
  out(UW_R, r[0]);
  out(16 + UW_R, r[1]);
  if (WITH_K) {
    RP_STAGE_FENCE();
    intrinsics_columns(fx, fy, xn, yn, r2, numer, idn, xpp, ypp, kmask, 32, out);
  }
  RP_STAGE_FENCE();

  const double ratio = numer * idn;
  const double dnum = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r2 * r2;
  const double dden = NODEN ? 0.0 : k4 + 2.0 * k5 * r2 + 3.0 * k6 * r2 * r2;
  const double dratio = NODEN ? dnum : (dnum - ratio * dden) * idn;  // d(numer/denom)/d(r2)
  const double dxx = ratio + 2.0 * xn * xn * dratio + 2.0 * p1 * yn + 6.0 * p2 * xn;
  const double dxy = 2.0 * xn * yn * dratio + 2.0 * p1 * xn + 2.0 * p2 * yn;
  const double dyy = ratio + 2.0 * yn * yn * dratio + 2.0 * p2 * xn + 6.0 * p1 * yn;
  double A[6];  // d(u,v)/d(Xc), 2x3
  const double fz = fx * iz, gz = fy * iz;
  A[0] = fz * dxx; A[1] = fz * dxy; A[2] = -fz * (dxx * xn + dxy * yn);
  A[3] = gz * dxy; A[4] = gz * dyy; A[5] = -gz * (dxy * xn + dyy * yn);
  RP_STAGE_FENCE();
#define RP_CROSS_OUT(col, w, b, row)                                  \
  out(16 * (row) + (col) + 0, (w)[1] * (b)[2] - (w)[2] * (b)[1]);     \
  out(16 * (row) + (col) + 1, (w)[2] * (b)[0] - (w)[0] * (b)[2]);     \
  out(16 * (row) + (col) + 2, (w)[0] * (b)[1] - (w)[1] * (b)[0]);
  for (int row = 0; row < 2; ++row) {
    const double* a = A + 3 * row;
    out(16 * row + UW_A + 0, a[0]);
    out(16 * row + UW_A + 1, a[1]);
    out(16 * row + UW_A + 2, a[2]);
    RP_CROSS_OUT(UW_WC, wc, a, row)
  }
  RP_STAGE_FENCE();
  const double* Rc = C + PREP_R;
  const double* Rt = T + PREP_R;
  for (int row = 0; row < 2; ++row) {
    const double* a = A + 3 * row;
    double b1[3], b2[3];
    b1[0] = a[0] * Rc[0] + a[1] * Rc[3] + a[2] * Rc[6];
    b1[1] = a[0] * Rc[1] + a[1] * Rc[4] + a[2] * Rc[7];
    b1[2] = a[0] * Rc[2] + a[1] * Rc[5] + a[2] * Rc[8];
    RP_CROSS_OUT(UW_WT, wt, b1, row)
    b2[0] = b1[0] * Rt[0] + b1[1] * Rt[3] + b1[2] * Rt[6];
    b2[1] = b1[0] * Rt[1] + b1[1] * Rt[4] + b1[2] * Rt[7];
    b2[2] = b1[0] * Rt[2] + b1[1] * Rt[5] + b1[2] * Rt[8];
    out(16 * row + UW_B2 + 0, b2[0]);
    out(16 * row + UW_B2 + 1, b2[1]);
    out(16 * row + UW_B2 + 2, b2[2]);
    RP_CROSS_OUT(UW_WP, wp, b2, row)
  }
#undef RP_CROSS_OUT
}

// The FR-constant matrix that takes U to the reference's columns: J[22] = U[16] * M, with
// J = [camera 7 | pattern 7 | time 7 | r] (host tests; the gathers apply it blockwise).
// cam/pat/tim: raw 7-parameter blocks, pat/tim null for the fixed exemplars (zero columns).
RP_HD void rotvec_to_reference_columns(const double* cam, const double* pat, const double* tim, double* M /*16 x 22*/) {
  for (int i = 0; i < 16 * 22; ++i) M[i] = 0.0;
  double E[12], blk[PREP_SIZE];
  quat_rotvec_map(cam, E);
  for (int m = 0; m < 3; ++m) {
    for (int j = 0; j < 4; ++j) M[(UW_WC + m) * 22 + j] = E[4 * m + j];
    M[(UW_A + m) * 22 + 4 + m] = 1.0;
  }
  if (pat) {
    quat_rotvec_map(pat, E);
    for (int m = 0; m < 3; ++m) {
      for (int j = 0; j < 4; ++j) M[(UW_WP + m) * 22 + 7 + j] = E[4 * m + j];
      M[(UW_B2 + m) * 22 + 7 + 4 + m] = 1.0;
    }
  }
  if (tim) {
    quat_rotvec_map(tim, E);
    prep_block(tim, blk);
    for (int m = 0; m < 3; ++m) {
      for (int j = 0; j < 4; ++j) M[(UW_WT + m) * 22 + 14 + j] = E[4 * m + j];
      for (int j = 0; j < 3; ++j) M[(UW_B2 + m) * 22 + 14 + 4 + j] = blk[PREP_R + 3 * j + m];  // b1 = b2 Rt^T
    }
  }
  M[UW_R * 22 + 21] = 1.0;
}

// ---- chain columns (the fused kernel's compact Gram space) ----------------------------
// The three factors act on the corner as ONE rigid transform Xc = Rc (Rt (Rp X + tp) + tt) + tc,
// and a left perturbation of an inner factor is a left perturbation of the whole chain seen
// through the adjoint of what stands to its left.  With c1 = wc x a (the camera's own rotation
// columns) the other rotation-vector columns of a residual row are FR-constant mixes of c1 and a:
//     wt x b1 = Rc^T c1 - [tt]x Rc^T a
//     b2      = Rt^T Rc^T a
//     wp x b2 = Rt^T (wt x b1) - [tp]x b2
// i.e. the 16 columns U of eval_prepared_w are  U = V C  with the 8 columns
//     V = [ wc x a (3) | a (3) | r | 0 ]
// and C = [ I_7 0 | C9 ] (8 x 16), C9 (rows of V x columns [wp x b2 | b2 | wt x b1]):
//     rows wc x a :  [ Q        | 0 | Rc       ]        Q = Rc Rt
//     rows a      :  [ Y        | Q | X        ]        X = Rc [tt]x,  Y = X Rt + Q [tp]x
// so the Gram of a part needs 7 columns per observation row (G7 = V^T V) and the 16-column
// Gram the records hold is  C^T G7 C, formed once per part.  The corner itself needs only the
// composed transform:  Xc = (Q Rp) X + (Q tp + Rc tt + tc),  wc = Xc - tc.
// One prepared FR for this form (prep_fr_chain), 64 doubles:
constexpr int FRC_R = 0, FRC_T = 9, FRC_TC = 12, FRC_K = 16, FRC_Q = 28, FRC_Y = 37, FRC_RC = 46, FRC_X = 55,
              FRC_SIZE = 64;
constexpr int CW_WC = 0, CW_A = 3, CW_R = 6, CW_COLS = 8;

// C, T, P: [R | t] of the camera / time / pattern block (12 doubles each, the identity for the
// fixed exemplars); K: the camera's 12 parameters.
RP_HD void prep_fr_chain(const double* C, const double* T, const double* P, const double* K, double* e) {
  const double* Rc = C;
  const double* Rt = T;
  const double* Rp = P;
  const double* tc = C + 9;
  const double* tt = T + 9;
  const double* tp = P + 9;
  double Q[9], X[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Q[3 * i + j] = Rc[3 * i] * Rt[j] + Rc[3 * i + 1] * Rt[3 + j] + Rc[3 * i + 2] * Rt[6 + j];
    X[3 * i + 0] = Rc[3 * i + 1] * tt[2] - Rc[3 * i + 2] * tt[1];
    X[3 * i + 1] = Rc[3 * i + 2] * tt[0] - Rc[3 * i + 0] * tt[2];
    X[3 * i + 2] = Rc[3 * i + 0] * tt[1] - Rc[3 * i + 1] * tt[0];
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      e[FRC_R + 3 * i + j] = Q[3 * i] * Rp[j] + Q[3 * i + 1] * Rp[3 + j] + Q[3 * i + 2] * Rp[6 + j];
      e[FRC_Q + 3 * i + j] = Q[3 * i + j];
      e[FRC_RC + 3 * i + j] = Rc[3 * i + j];
      e[FRC_X + 3 * i + j] = X[3 * i + j];
    }
    e[FRC_T + i] = (Q[3 * i] * tp[0] + Q[3 * i + 1] * tp[1] + Q[3 * i + 2] * tp[2]) +
                   (Rc[3 * i] * tt[0] + Rc[3 * i + 1] * tt[1] + Rc[3 * i + 2] * tt[2]) + tc[i];
    e[FRC_TC + i] = tc[i];
    // Q [tp]x, row i
    const double qx0 = Q[3 * i + 1] * tp[2] - Q[3 * i + 2] * tp[1];
    const double qx1 = Q[3 * i + 2] * tp[0] - Q[3 * i + 0] * tp[2];
    const double qx2 = Q[3 * i + 0] * tp[1] - Q[3 * i + 1] * tp[0];
    e[FRC_Y + 3 * i + 0] = X[3 * i] * Rt[0] + X[3 * i + 1] * Rt[3] + X[3 * i + 2] * Rt[6] + qx0;
    e[FRC_Y + 3 * i + 1] = X[3 * i] * Rt[1] + X[3 * i + 1] * Rt[4] + X[3 * i + 2] * Rt[7] + qx1;
    e[FRC_Y + 3 * i + 2] = X[3 * i] * Rt[2] + X[3 * i + 1] * Rt[5] + X[3 * i + 2] * Rt[8] + qx2;
  }
  e[15] = 0.0;
  for (int k = 0; k < 12; ++k) e[FRC_K + k] = K[k];
}

// Entry (k, j) of C (8 x 16) as an offset into a prepared FR, or -1 / -2 for the constants 0 / 1.
RP_HD int chain_map_offset(int k, int j) {
  if (j < 7) return k == j ? -2 : -1;
  const int m = j - 7;
  if (k >= 6) return -1;
  if (k < 3) return m < 3 ? FRC_Q + 3 * k + m : m < 6 ? -1 : FRC_RC + 3 * k + (m - 6);
  return m < 3 ? FRC_Y + 3 * (k - 3) + m : m < 6 ? FRC_Q + 3 * (k - 3) + (m - 3) : FRC_X + 3 * (k - 3) + (m - 6);
}

// One observation from a prepared FR of this form: r[2] and out(8 * row + column, value) for the
// seven columns V of both residual rows (column 7 is never emitted).
// WITH_K (camera parameters refined): also the 2 x 12 columns of intrinsics_columns as
// out(16 + 12 * row + k).
template <bool NODEN = false, bool WITH_K = false, typename Out>
RP_HD void eval_chain(const double* e, const double X[3], double u, double v, double* r, Out& out, unsigned kmask = 0) {
  const double* R = e + FRC_R;
  const double* K = e + FRC_K;
  double xc[3], wc[3];
  xc[0] = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + e[FRC_T];
  xc[1] = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + e[FRC_T + 1];
  xc[2] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + e[FRC_T + 2];
  wc[0] = xc[0] - e[FRC_TC]; wc[1] = xc[1] - e[FRC_TC + 1]; wc[2] = xc[2] - e[FRC_TC + 2];

  const double fx = K[0], cx = K[1], fy = K[2], cy = K[3];
  const double k1 = K[4], k2 = K[5], p1 = K[6], p2 = K[7], k3 = K[8], k4 = K[9], k5 = K[10], k6 = K[11];
  const double iz = 1.0 / xc[2];
  const double xn = xc[0] * iz, yn = xc[1] * iz;
  const double r2 = xn * xn + yn * yn;
  const double numer = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;   // hpp:770
  const double denom = NODEN ? 1.0 : 1.0 + k4 * r2 + k5 * r2 * r2 + k6 * r2 * r2 * r2;   // hpp:771
  const double idn = NODEN ? 1.0 : 1.0 / denom;
  const double xpp = xn * numer * idn + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);  // :773
  const double ypp = yn * numer * idn + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);  // :774
  r[0] = xpp * fx + cx - u;  // :776,783
  r[1] = ypp * fy + cy - v;  // :777,784
  out(CW_R, r[0]);
  out(8 + CW_R, r[1]);
  if (WITH_K) {
    RP_STAGE_FENCE();
    intrinsics_columns(fx, fy, xn, yn, r2, numer, idn, xpp, ypp, kmask, 16, out);
  }
  RP_STAGE_FENCE();

  const double ratio = numer * idn;
  const double dnum = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r2 * r2;
  const double dden = NODEN ? 0.0 : k4 + 2.0 * k5 * r2 + 3.0 * k6 * r2 * r2;
  const double dratio = NODEN ? dnum : (dnum - ratio * dden) * idn;  // d(numer/denom)/d(r2)
  const double dxx = ratio + 2.0 * xn * xn * dratio + 2.0 * p1 * yn + 6.0 * p2 * xn;
  const double dxy = 2.0 * xn * yn * dratio + 2.0 * p1 * xn + 2.0 * p2 * yn;
  const double dyy = ratio + 2.0 * yn * yn * dratio + 2.0 * p2 * xn + 6.0 * p1 * yn;
  double A[6];  // d(u,v)/d(Xc), 2x3
  const double fz = fx * iz, gz = fy * iz;
  A[0] = fz * dxx; A[1] = fz * dxy; A[2] = -fz * (dxx * xn + dxy * yn);
  A[3] = gz * dxy; A[4] = gz * dyy; A[5] = -gz * (dxy * xn + dyy * yn);
  RP_STAGE_FENCE();
  for (int row = 0; row < 2; ++row) {
    const double* a = A + 3 * row;
    out(8 * row + CW_A + 0, a[0]);
    out(8 * row + CW_A + 1, a[1]);
    out(8 * row + CW_A + 2, a[2]);
    out(8 * row + CW_WC + 0, wc[1] * a[2] - wc[2] * a[1]);
    out(8 * row + CW_WC + 1, wc[2] * a[0] - wc[0] * a[2]);
    out(8 * row + CW_WC + 2, wc[0] * a[1] - wc[1] * a[0]);
  }
}

// Residual only, from the composed transform: e = [R (9) | t (3) | K (12)] (cost sweep).
constexpr int FRR_SIZE = 24;
RP_HD void compose_fr(const double* C, const double* T, const double* P, const double* K, double* e) {
  double Q[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Q[3 * i + j] = C[3 * i] * T[j] + C[3 * i + 1] * T[3 + j] + C[3 * i + 2] * T[6 + j];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) e[3 * i + j] = Q[3 * i] * P[j] + Q[3 * i + 1] * P[3 + j] + Q[3 * i + 2] * P[6 + j];
    e[9 + i] = (Q[3 * i] * P[9] + Q[3 * i + 1] * P[10] + Q[3 * i + 2] * P[11]) +
               (C[3 * i] * T[9] + C[3 * i + 1] * T[10] + C[3 * i + 2] * T[11]) + C[9 + i];
  }
  for (int k = 0; k < 12; ++k) e[12 + k] = K[k];
}
template <bool NODEN = false>
RP_HD void eval_composed_residual(const double* e, const double X[3], double u, double v, double* r) {
  const double* K = e + 12;
  const double x0 = e[0] * X[0] + e[1] * X[1] + e[2] * X[2] + e[9];
  const double x1 = e[3] * X[0] + e[4] * X[1] + e[5] * X[2] + e[10];
  const double x2 = e[6] * X[0] + e[7] * X[1] + e[8] * X[2] + e[11];
  const double fx = K[0], cx = K[1], fy = K[2], cy = K[3];
  const double k1 = K[4], k2 = K[5], p1 = K[6], p2 = K[7], k3 = K[8], k4 = K[9], k5 = K[10], k6 = K[11];
  const double iz = 1.0 / x2;
  const double xn = x0 * iz, yn = x1 * iz;
  const double r2 = xn * xn + yn * yn;
  const double numer = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;   // hpp:770
  const double denom = NODEN ? 1.0 : 1.0 + k4 * r2 + k5 * r2 * r2 + k6 * r2 * r2 * r2;   // hpp:771
  const double idn = NODEN ? 1.0 : 1.0 / denom;
  const double xpp = xn * numer * idn + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);  // :773
  const double ypp = yn * numer * idn + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);  // :774
  r[0] = xpp * fx + cx - u;  // :776,783
  r[1] = ypp * fy + cy - v;  // :777,784
}

// Prepare one FR: cam/pat/tim 7-parameter blocks (pat/tim may be null), 12 intrinsics.
RP_HD void prep_fr(const double* cam, const double* pat, const double* tim, const double* K, double* frp) {
  prep_block(cam, frp + FRP_CAM);
  prep_block(tim, frp + FRP_TIM);
  prep_block(pat, frp + FRP_PAT);
  for (int k = 0; k < 12; ++k) frp[FRP_K + k] = K[k];
}

struct ArraySink {
  double* J;
  RP_HD void operator()(int k, double v) { J[k] = v; }
};
struct NullSink {
  RP_HD void operator()(int, double) {}
};

// Convenience: unprepared single observation (host tests, small paths).
//   cam / pat / tim : 7-parameter blocks; pat == nullptr <=> p*, tim == nullptr <=> t*
//   K               : [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6]
//   J[42] (WITH_J)  : row-major 2 x 21, columns [camera | pattern | time]
template <bool WITH_J>
RP_HD void eval_observation(const double* cam, const double* pat, const double* tim, const double* K,
                            const double X[3], double u, double v, double* r, double* J) {
  double frp[FRP_SIZE];
  prep_fr(cam, pat, tim, K, frp);
  if (WITH_J) {
    ArraySink s{J};
    eval_prepared<true>(frp, X, u, v, r, s);
  } else {
    NullSink s;
    eval_prepared<false>(frp, X, u, v, r, s);
  }
}

}  // namespace rp
