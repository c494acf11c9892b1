// oracle/rp_functors.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the four reprojection-error cost functors of the reference,
// templated on the scalar type so that they can be evaluated on double (cost only)
// and on oracle::Jet<N> (autodiff Jacobian), as ceres::AutoDiffCostFunction does.
// Every function cites the reference lines whose arithmetic (operation order
// included) it restates.  Nothing here is shipped or called by the product path.
#pragma once
#include "jet.hpp"

namespace oracle {

// ceres::QuaternionToRotation (ceres/rotation.h; not in /root/reference, formula
// recorded in SURVEY.md appendix): scaled rotation from (a,b,c,d), then every entry
// multiplied by 1/(a^2+b^2+c^2+d^2).  Called from solving-structure.hpp:36,64.
template <typename T>
inline void quaternion_to_rotation(const T* q, T* R /*row-major 3x3*/) {
  const T a = q[0], b = q[1], c = q[2], d = q[3];
  const T aa = a * a, ab = a * b, ac = a * c, ad = a * d;
  const T bb = b * b, bc = b * c, bd = b * d;
  const T cc = c * c, cd = c * d, dd = d * d;
  R[0] = aa + bb - cc - dd;
  R[1] = T(2.0) * (bc - ad);
  R[2] = T(2.0) * (ac + bd);
  R[3] = T(2.0) * (ad + bc);
  R[4] = aa - bb + cc - dd;
  R[5] = T(2.0) * (cd - ab);
  R[6] = T(2.0) * (bd - ac);
  R[7] = T(2.0) * (ab + cd);
  R[8] = aa - bb - cc + dd;
  T normalizer = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  normalizer = T(1.0) / normalizer;
  for (int i = 0; i < 9; ++i) R[i] *= normalizer;
}

// Convert7ParameterQuaternionRepresentationIntoMatrix<T>(const T*, T*)
// solving-structure.hpp:32-58: [qw qx qy qz tx ty tz] -> row-major 4x4 [R|t; 0 0 0 1].
template <typename T>
inline void seven_to_mat4(const T* X, T* M) {
  T R[9];
  quaternion_to_rotation(X, R);
  M[0] = R[0]; M[1] = R[1]; M[2]  = R[2]; M[3]  = X[4];
  M[4] = R[3]; M[5] = R[4]; M[6]  = R[5]; M[7]  = X[5];
  M[8] = R[6]; M[9] = R[7]; M[10] = R[8]; M[11] = X[6];
  M[12] = T(0.0); M[13] = T(0.0); M[14] = T(0.0); M[15] = T(1.0);
}

// The identity 4x4 the reference writes out when a factor is p* / t*
// (solving-structure.hpp:861-874 both; :1028-1038 pattern; :1197-1204 time).
template <typename T>
inline void identity_mat4(T* M) {
  for (int i = 0; i < 16; ++i) M[i] = T(0.0);
  M[0] = T(1.0); M[5] = T(1.0); M[10] = T(1.0); M[15] = T(1.0);
}

// MatrixMultiply<T>(X, Y, M, row0, col0, row1, col1), solving-structure.hpp:134-160
// (and the square overload :112-132, same loop nest): r starts at T(0) and the
// products are added in index order.
template <typename T>
inline void mat_mul(const T* X, const T* Y, T* M, int row0, int col0, int col1) {
  for (int i = 0; i < row0; ++i) {
    for (int j = 0; j < col1; ++j) {
      T r(0.0);
      for (int k = 0; k < col0; ++k) r += X[i * col0 + k] * Y[k * col1 + j];
      M[i * col1 + j] = r;
    }
  }
}

// MultiCameraReprojectionError::operator() solving-structure.hpp:658-790 (case 0),
// ...ErrorCamTime :994-1147 (pattern == p*), ...ErrorCamPattern :1160-1310
// (time == t*; note :1290-1297 actually instantiates the CamTime functor with the
// pattern block in the time slot, which multiplies the same two matrices and an
// identity in a different order — same value), ...ErrorCam :829-961 (both).
//   pattern_var == nullptr  <=> the pattern factor is the identity (p*)
//   time_var    == nullptr  <=> the time factor is the identity (t*)
// camera_parameters = [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6] (solving-structure.cpp:134-159).
// The functors used by Stage 5 are built through the (im_x, im_y, Point3f) overload,
// so weighting == 0 and w == 1 (hpp:652-656,677-683), the model point is float32
// promoted to T (hpp:757-759).
// KT is the type of the 12 camera parameters: double in the reference (constants inside the
// functor, hpp:738-749) or T when they are a fourth parameter block (the "intrinsics +
// distortion refined" extension of BASELINE.json configs[2]; same formula, T(camera_parameters[k])
// is then a copy of the dual number instead of a promotion of a constant).
template <typename T, typename KT = double>
inline void reprojection_residual(const T* camera_var, const T* pattern_var, const T* time_var,
                                  const KT* camera_parameters, double im_x, double im_y,
                                  const float* pt3f, T* residuals) {
  T CM[16], PM[16], tM[16];
  seven_to_mat4(camera_var, CM);
  if (pattern_var) seven_to_mat4(pattern_var, PM); else identity_mat4(PM);
  if (time_var) seven_to_mat4(time_var, tM); else identity_mat4(tM);

  // A_hat = C * T * P  (hpp:733-734; P and T blocks already hold the inverses)
  T term0[16], A_hat[16];
  mat_mul(CM, tM, term0, 4, 4, 4);
  mat_mul(term0, PM, A_hat, 4, 4, 4);

  const T fx(camera_parameters[0]), cx(camera_parameters[1]);
  const T fy(camera_parameters[2]), cy(camera_parameters[3]);
  const T k1(camera_parameters[4]), k2(camera_parameters[5]);
  const T p1(camera_parameters[6]), p2(camera_parameters[7]);
  const T k3(camera_parameters[8]), k4(camera_parameters[9]);
  const T k5(camera_parameters[10]), k6(camera_parameters[11]);

  T Xp[4] = {T(double(pt3f[0])), T(double(pt3f[1])), T(double(pt3f[2])), T(1.0)};
  T xp[3];
  mat_mul(A_hat, Xp, xp, 3, 4, 1);  // hpp:763 (first three rows of the 4x4)

  xp[0] = xp[0] / xp[2];  // hpp:765-766
  xp[1] = xp[1] / xp[2];
  const T r_sqr = xp[0] * xp[0] + xp[1] * xp[1];  // hpp:768

  const T numer = T(1.0) + k1 * r_sqr + k2 * r_sqr * r_sqr + k3 * r_sqr * r_sqr * r_sqr;  // :770
  const T denom = T(1.0) + k4 * r_sqr + k5 * r_sqr * r_sqr + k6 * r_sqr * r_sqr * r_sqr;  // :771

  const T xpp0 = xp[0] * numer / denom + T(2.0) * p1 * xp[0] * xp[1] +
                 p2 * (r_sqr + T(2.0) * xp[0] * xp[0]);  // :773
  const T xpp1 = xp[1] * numer / denom + T(2.0) * p2 * xp[0] * xp[1] +
                 p1 * (r_sqr + T(2.0) * xp[1] * xp[1]);  // :774

  const T predicted_x = xpp0 * fx + cx;  // :776
  const T predicted_y = xpp1 * fy + cy;  // :777

  const T w(1.0);
  residuals[0] = w * (predicted_x - T(im_x));  // :783-784
  residuals[1] = w * (predicted_y - T(im_y));
}

}  // namespace oracle
