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
// The header is plain C++ so that tests can run the very same code on the host
// against the dual-number oracle (tests/helpers/model_host.cpp); the product only
// ever calls it from device code.
#pragma once

#if defined(__HIPCC__)
#define RP_HD __host__ __device__ __forceinline__
#else
#define RP_HD inline
#endif

namespace rp {

struct Rot {
  double m[9];  // row-major R = M(q)/|q|^2
  double inv_n; // 1/|q|^2
};

RP_HD void quat_to_rot(const double* q, Rot& R) {
  const double a = q[0], b = q[1], c = q[2], d = q[3];
  const double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
  const double inv = 1.0 / (aa + bb + cc + dd);
  R.inv_n = inv;
  R.m[0] = (aa + bb - cc - dd) * inv;
  R.m[1] = 2.0 * (b * c - a * d) * inv;
  R.m[2] = 2.0 * (a * c + b * d) * inv;
  R.m[3] = 2.0 * (a * d + b * c) * inv;
  R.m[4] = (aa - bb + cc - dd) * inv;
  R.m[5] = 2.0 * (c * d - a * b) * inv;
  R.m[6] = 2.0 * (b * d - a * c) * inv;
  R.m[7] = 2.0 * (a * b + c * d) * inv;
  R.m[8] = (aa - bb - cc + dd) * inv;
}

RP_HD void rot_apply(const Rot& R, const double* y, double* w) {
  w[0] = R.m[0] * y[0] + R.m[1] * y[1] + R.m[2] * y[2];
  w[1] = R.m[3] * y[0] + R.m[4] * y[1] + R.m[5] * y[2];
  w[2] = R.m[6] * y[0] + R.m[7] * y[1] + R.m[8] * y[2];
}

// B (2x3) times d(R(q) y)/dq (3x4) -> out[2][4]; w = R y (rotation only).
RP_HD void jac_quat(const double B[6], const double* q, const Rot& R, const double* y,
                    const double* w, double* out_row0, double* out_row1) {
  const double a = q[0], b = q[1], c = q[2], d = q[3];
  const double s0 = a * y[0] - d * y[1] + c * y[2];
  const double s1 = d * y[0] + a * y[1] - b * y[2];
  const double s2 = -c * y[0] + b * y[1] + a * y[2];
  const double s3 = b * y[0] + c * y[1] + d * y[2];
  const double k = 2.0 * R.inv_n;
  // columns of S
  const double Sa[3] = {s0, s1, s2}, Sb[3] = {s3, -s2, s1}, Sc[3] = {s2, s3, -s0}, Sd[3] = {-s1, s0, s3};
  const double Bw0 = B[0] * w[0] + B[1] * w[1] + B[2] * w[2];
  const double Bw1 = B[3] * w[0] + B[4] * w[1] + B[5] * w[2];
  out_row0[0] = k * (B[0] * Sa[0] + B[1] * Sa[1] + B[2] * Sa[2] - Bw0 * a);
  out_row0[1] = k * (B[0] * Sb[0] + B[1] * Sb[1] + B[2] * Sb[2] - Bw0 * b);
  out_row0[2] = k * (B[0] * Sc[0] + B[1] * Sc[1] + B[2] * Sc[2] - Bw0 * c);
  out_row0[3] = k * (B[0] * Sd[0] + B[1] * Sd[1] + B[2] * Sd[2] - Bw0 * d);
  out_row1[0] = k * (B[3] * Sa[0] + B[4] * Sa[1] + B[5] * Sa[2] - Bw1 * a);
  out_row1[1] = k * (B[3] * Sb[0] + B[4] * Sb[1] + B[5] * Sb[2] - Bw1 * b);
  out_row1[2] = k * (B[3] * Sc[0] + B[4] * Sc[1] + B[5] * Sc[2] - Bw1 * c);
  out_row1[3] = k * (B[3] * Sd[0] + B[4] * Sd[1] + B[5] * Sd[2] - Bw1 * d);
}

// One observation.
//   cam / pat / tim : 7-parameter blocks; pat == nullptr <=> p*, tim == nullptr <=> t*
//   K               : [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6]
//   X               : model point (float32 promoted by the caller)
//   r[2]            : residual (predicted - observed), weight 1 (hpp:677-683)
//   J[42] (WITH_J)  : row-major 2 x 21, columns [camera | pattern | time]
template <bool WITH_J>
RP_HD void eval_observation(const double* cam, const double* pat, const double* tim, const double* K,
                            const double X[3], double u, double v, double* r, double* J) {
  Rot Rc, Rt, Rp;
  double z[3], y[3], xc[3], wp[3] = {0, 0, 0}, wt[3] = {0, 0, 0}, wc[3];
  if (pat) {
    quat_to_rot(pat, Rp);
    rot_apply(Rp, X, wp);
    z[0] = wp[0] + pat[4]; z[1] = wp[1] + pat[5]; z[2] = wp[2] + pat[6];
  } else {
    z[0] = X[0]; z[1] = X[1]; z[2] = X[2];
  }
  if (tim) {
    quat_to_rot(tim, Rt);
    rot_apply(Rt, z, wt);
    y[0] = wt[0] + tim[4]; y[1] = wt[1] + tim[5]; y[2] = wt[2] + tim[6];
  } else {
    y[0] = z[0]; y[1] = z[1]; y[2] = z[2];
  }
  quat_to_rot(cam, Rc);
  rot_apply(Rc, y, wc);
  xc[0] = wc[0] + cam[4]; xc[1] = wc[1] + cam[5]; xc[2] = wc[2] + cam[6];

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

  // camera block: [A * dq(Rc y) | A]
  jac_quat(A, cam, Rc, y, wc, J + 0, J + 21);
  J[4] = A[0]; J[5] = A[1]; J[6] = A[2];
  J[25] = A[3]; J[26] = A[4]; J[27] = A[5];

  // B1 = A Rc
  double B1[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      B1[3 * i + j] = A[3 * i] * Rc.m[j] + A[3 * i + 1] * Rc.m[3 + j] + A[3 * i + 2] * Rc.m[6 + j];

  double B2[6];
  if (tim) {
    jac_quat(B1, tim, Rt, z, wt, J + 14, J + 35);
    J[18] = B1[0]; J[19] = B1[1]; J[20] = B1[2];
    J[39] = B1[3]; J[40] = B1[4]; J[41] = B1[5];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j)
        B2[3 * i + j] = B1[3 * i] * Rt.m[j] + B1[3 * i + 1] * Rt.m[3 + j] + B1[3 * i + 2] * Rt.m[6 + j];
  } else {
    for (int k = 0; k < 7; ++k) { J[14 + k] = 0.0; J[35 + k] = 0.0; }
    for (int k = 0; k < 6; ++k) B2[k] = B1[k];
  }
  if (pat) {
    jac_quat(B2, pat, Rp, X, wp, J + 7, J + 28);
    J[11] = B2[0]; J[12] = B2[1]; J[13] = B2[2];
    J[32] = B2[3]; J[33] = B2[4]; J[34] = B2[5];
  } else {
    for (int k = 0; k < 7; ++k) { J[7 + k] = 0.0; J[28 + k] = 0.0; }
  }
}

}  // namespace rp
