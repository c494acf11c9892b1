// ae_model.hpp — closed-form algebraic-error residual and Jacobian for one FR
// (the Stage-4 "AE" refinement).
//
// Replaces, for the GPU path, the four autodiff cost functors
//   AlgebraicErrorCamPatternTime  residual = rows 0..2 of  C - A P T   (solving-structure.hpp, case p != p*, t != t*)
//   AlgebraicErrorCamTime         C - A T                              (hpp:388-462, p == p*)
//   AlgebraicErrorCamPattern      C - A P                              (hpp:231-310, t == t*)
//   AlgebraicErrorCam             C - A                                (hpp:173-229, both)
// where C, P, T are the 4x4 matrices of the camera / pattern / time 7-vectors
// (Convert7ParameterQuaternionRepresentationIntoMatrix, hpp:32-58: R(q)/|q|^2 and t, NOT
// inverted in this stage, "A*P*T = C no inverses" hpp:397-398) and A is the FR's measured
// transform, given as a 7-vector too.  12 residuals, row-major (index 4*row + column).
//
//   A P T = [ Ra Rp Rt | Ra (Rp tt + tp) + ta ]
//   d/dq_c : d(Rc e_j)/dq_c                      d/dt_c : identity on column 3
//   d/dq_p : -Ra d(Rp y)/dq_p,  y = Rt e_j (rotation columns), y = tt (column 3)
//   d/dt_p : -Ra on column 3
//   d/dq_t : -(Ra Rp) d(Rt e_j)/dq_t             d/dt_t : -(Ra Rp) on column 3
// with d(R(q) y)/dq = (2/|q|^2) (S(q,y) - (R y) q^T) as in rp_model.hpp.
// Plain C++ so that tests run the same code on the host against the dual-number oracle.
#pragma once
#include "rp_model.hpp"

namespace rp {

constexpr int AE_ROWS = 12;

// D[3][4] = d(R(q) y)/dq for a prepared block (PREP_* layout)
RP_HD void drot_dq(const double* blk, const double y[3], double D[12]) {
  const double a = blk[PREP_Q], b = blk[PREP_Q + 1], c = blk[PREP_Q + 2], d = blk[PREP_Q + 3];
  const double s0 = a * y[0] - d * y[1] + c * y[2];
  const double s1 = d * y[0] + a * y[1] - b * y[2];
  const double s2 = -c * y[0] + b * y[1] + a * y[2];
  const double s3 = b * y[0] + c * y[1] + d * y[2];
  const double k = 2.0 * blk[PREP_INV];
  double w[3];
  rot_apply(blk + PREP_R, y, w);
  const double S[12] = {s0, s3, s2, -s1, s1, -s2, s3, s0, s2, s1, -s0, s3};
  const double q[4] = {a, b, c, d};
  for (int i = 0; i < 3; ++i)
    for (int m = 0; m < 4; ++m) D[4 * i + m] = k * (S[4 * i + m] - w[i] * q[m]);
}

// M (3x3 row-major) times D (3x4): G = -M D
RP_HD void neg_mat_d(const double* M, const double D[12], double G[12]) {
  for (int i = 0; i < 3; ++i)
    for (int m = 0; m < 4; ++m) G[4 * i + m] = -(M[3 * i] * D[m] + M[3 * i + 1] * D[4 + m] + M[3 * i + 2] * D[8 + m]);
}

// cam / pat / tim / a : prepared blocks (prep_block); pat / tim absent <=> blk[PREP_ON] == 0
//   r[12]; J: 12 rows of LD doubles, columns [camera 7 | pattern 7 | time 7] (zero where absent; columns
//   21.. of a row are left alone), may be null
template <int LD = 21>
RP_HD void ae_eval_prepared(const double* C, const double* P, const double* T, const double* A, double* r, double* J) {
  const bool on_p = P[PREP_ON] != 0.0, on_t = T[PREP_ON] != 0.0;
  const double* Ra = A + PREP_R;
  double M2[9];  // Ra Rp
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      M2[3 * i + j] = on_p ? Ra[3 * i] * P[PREP_R + j] + Ra[3 * i + 1] * P[PREP_R + 3 + j] + Ra[3 * i + 2] * P[PREP_R + 6 + j]
                           : Ra[3 * i + j];
  double tin[3];  // Rp tt + tp (the point Ra acts on)
  {
    double tt[3] = {0.0, 0.0, 0.0};
    if (on_t) { tt[0] = T[PREP_T]; tt[1] = T[PREP_T + 1]; tt[2] = T[PREP_T + 2]; }
    if (on_p) {
      rot_apply(P + PREP_R, tt, tin);
      tin[0] += P[PREP_T]; tin[1] += P[PREP_T + 1]; tin[2] += P[PREP_T + 2];
    } else {
      tin[0] = tt[0]; tin[1] = tt[1]; tin[2] = tt[2];
    }
  }
  double dtr[3];
  rot_apply(Ra, tin, dtr);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      const double dij = on_t ? M2[3 * i] * T[PREP_R + j] + M2[3 * i + 1] * T[PREP_R + 3 + j] + M2[3 * i + 2] * T[PREP_R + 6 + j]
                              : M2[3 * i + j];
      r[4 * i + j] = C[PREP_R + 3 * i + j] - dij;
    }
    r[4 * i + 3] = C[PREP_T + i] - (dtr[i] + A[PREP_T + i]);
  }
  if (!J) return;
  for (int rr = 0; rr < AE_ROWS; ++rr)
    for (int k = 0; k < 21; ++k) J[rr * LD + k] = 0.0;
  for (int j = 0; j < 3; ++j) {
    double e[3] = {0.0, 0.0, 0.0};
    e[j] = 1.0;
    double D[12], G[12];
    // camera: d(Rc e_j)/dq_c
    drot_dq(C, e, D);
    for (int i = 0; i < 3; ++i)
      for (int m = 0; m < 4; ++m) J[(4 * i + j) * LD + m] = D[4 * i + m];
    // pattern: -Ra d(Rp y)/dq_p, y = column j of Rt (or e_j)
    if (on_p) {
      double y[3] = {e[0], e[1], e[2]};
      if (on_t) { y[0] = T[PREP_R + j]; y[1] = T[PREP_R + 3 + j]; y[2] = T[PREP_R + 6 + j]; }
      drot_dq(P, y, D);
      neg_mat_d(Ra, D, G);
      for (int i = 0; i < 3; ++i)
        for (int m = 0; m < 4; ++m) J[(4 * i + j) * LD + 7 + m] = G[4 * i + m];
    }
    // time: -(Ra Rp) d(Rt e_j)/dq_t
    if (on_t) {
      drot_dq(T, e, D);
      neg_mat_d(M2, D, G);
      for (int i = 0; i < 3; ++i)
        for (int m = 0; m < 4; ++m) J[(4 * i + j) * LD + 14 + m] = G[4 * i + m];
    }
  }
  // column 3 (translations)
  for (int i = 0; i < 3; ++i) {
    J[(4 * i + 3) * LD + 4 + i] = 1.0;
    if (on_p)
      for (int m = 0; m < 3; ++m) J[(4 * i + 3) * LD + 7 + 4 + m] = -Ra[3 * i + m];
    if (on_t)
      for (int m = 0; m < 3; ++m) J[(4 * i + 3) * LD + 14 + 4 + m] = -M2[3 * i + m];
  }
  if (on_p && on_t) {
    double tt[3] = {T[PREP_T], T[PREP_T + 1], T[PREP_T + 2]};
    double D[12], G[12];
    drot_dq(P, tt, D);
    neg_mat_d(Ra, D, G);
    for (int i = 0; i < 3; ++i)
      for (int m = 0; m < 4; ++m) J[(4 * i + 3) * LD + 7 + m] = G[4 * i + m];
  }
}

// The same, one matrix column at a time: column j of the 3 x 4 residual matrix, i.e. residual
// rows 4 i + j (i = 0..2) with their Jacobian rows, and nothing else.  The four columns are
// independent, so four lanes can share a relation (ae_records_kernel); the arithmetic of every
// entry is that of ae_eval_prepared, bit for bit.  r is written at M[(4 i + j) * LD + RCOL], the
// pattern / time columns from PCOL / TCOL on; columns 0 .. NCOL - 1 are cleared first.
template <int LD, int RCOL, int PCOL = 7, int TCOL = 14, int NCOL = 21>
RP_HD void ae_eval_column(const double* C, const double* P, const double* T, const double* A, int j, double* M) {
  const bool on_p = P[PREP_ON] != 0.0, on_t = T[PREP_ON] != 0.0;
  const double* Ra = A + PREP_R;
  double M2[9];  // Ra Rp
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k)
      M2[3 * i + k] = on_p ? Ra[3 * i] * P[PREP_R + k] + Ra[3 * i + 1] * P[PREP_R + 3 + k] + Ra[3 * i + 2] * P[PREP_R + 6 + k]
                           : Ra[3 * i + k];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < NCOL; ++k) M[(4 * i + j) * LD + k] = 0.0;
  if (j < 3) {
    for (int i = 0; i < 3; ++i) {
      const double dij = on_t ? M2[3 * i] * T[PREP_R + j] + M2[3 * i + 1] * T[PREP_R + 3 + j] + M2[3 * i + 2] * T[PREP_R + 6 + j]
                              : M2[3 * i + j];
      M[(4 * i + j) * LD + RCOL] = C[PREP_R + 3 * i + j] - dij;
    }
    double e[3] = {0.0, 0.0, 0.0};
    e[0] = j == 0 ? 1.0 : 0.0;
    e[1] = j == 1 ? 1.0 : 0.0;
    e[2] = j == 2 ? 1.0 : 0.0;
    double D[12], G[12];
    drot_dq(C, e, D);
    for (int i = 0; i < 3; ++i)
      for (int m = 0; m < 4; ++m) M[(4 * i + j) * LD + m] = D[4 * i + m];
    if (on_p) {
      double y[3] = {e[0], e[1], e[2]};
      if (on_t) { y[0] = T[PREP_R + j]; y[1] = T[PREP_R + 3 + j]; y[2] = T[PREP_R + 6 + j]; }
      drot_dq(P, y, D);
      neg_mat_d(Ra, D, G);
      for (int i = 0; i < 3; ++i)
        for (int m = 0; m < 4; ++m) M[(4 * i + j) * LD + PCOL + m] = G[4 * i + m];
    }
    if (on_t) {
      drot_dq(T, e, D);
      neg_mat_d(M2, D, G);
      for (int i = 0; i < 3; ++i)
        for (int m = 0; m < 4; ++m) M[(4 * i + j) * LD + TCOL + m] = G[4 * i + m];
    }
  } else {
    double tin[3];  // Rp tt + tp (the point Ra acts on)
    double tt[3] = {0.0, 0.0, 0.0};
    if (on_t) { tt[0] = T[PREP_T]; tt[1] = T[PREP_T + 1]; tt[2] = T[PREP_T + 2]; }
    if (on_p) {
      rot_apply(P + PREP_R, tt, tin);
      tin[0] += P[PREP_T]; tin[1] += P[PREP_T + 1]; tin[2] += P[PREP_T + 2];
    } else {
      tin[0] = tt[0]; tin[1] = tt[1]; tin[2] = tt[2];
    }
    double dtr[3];
    rot_apply(Ra, tin, dtr);
    for (int i = 0; i < 3; ++i) {
      M[(4 * i + 3) * LD + RCOL] = C[PREP_T + i] - (dtr[i] + A[PREP_T + i]);
      M[(4 * i + 3) * LD + 4 + i] = 1.0;
      if (on_p)
        for (int m = 0; m < 3; ++m) M[(4 * i + 3) * LD + PCOL + 4 + m] = -Ra[3 * i + m];
      if (on_t)
        for (int m = 0; m < 3; ++m) M[(4 * i + 3) * LD + TCOL + 4 + m] = -M2[3 * i + m];
    }
    if (on_p && on_t) {
      double D[12], G[12];
      drot_dq(P, tt, D);
      neg_mat_d(Ra, D, G);
      for (int i = 0; i < 3; ++i)
        for (int m = 0; m < 4; ++m) M[(4 * i + 3) * LD + PCOL + m] = G[4 * i + m];
    }
  }
}

// Unprepared convenience (host tests): 7-vectors; pat / tim null <=> p* / t*
RP_HD void ae_eval(const double* cam, const double* pat, const double* tim, const double* a7, double* r, double* J) {
  double C[PREP_SIZE], P[PREP_SIZE], T[PREP_SIZE], A[PREP_SIZE];
  prep_block(cam, C);
  prep_block(pat, P);
  prep_block(tim, T);
  prep_block(a7, A);
  ae_eval_prepared(C, P, T, A, r, J);
}

}  // namespace rp
