// rae_model.hpp — closed-form residual and 2x3 Jacobian of ReconstructXStruct
// (solving-structure.hpp:1313-1409): the reprojection of a free 3-D point X
// through a fixed 3x4 transform A_hat and fixed intrinsics,
//   xc = A_hat [X; 1]   (hpp:1368),  (xn, yn) = (xc.x, xc.y) / xc.z   (:1370-1371),
//   r2, numer, denom, tangential terms   (:1373-1379),  (u, v) = (fx x'' + cx, fy y'' + cy)   (:1381-1382),
//   r = (u - im_x, v - im_y)   (:1384-1385);   dr/dX = d(u,v)/d(xc) * A_hat[:, 0:3].
// The reference differentiates it with Jets (AutoDiffCostFunction<ReconstructXStruct,2,3>, :1392).
// Plain C++ so that the tests run the same code on the host.
#pragma once
#include "rp_model.hpp"

namespace rp {

// Ah: rows 0..2 of A_hat, row-major 3x4; K: [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6];
// r[2]; J: 2x3 row-major or null
RP_HD void rae_eval(const double* Ah, const double* K, const double X[3], double u, double v, double* r, double* J) {
  const double xc0 = Ah[0] * X[0] + Ah[1] * X[1] + Ah[2] * X[2] + Ah[3];
  const double xc1 = Ah[4] * X[0] + Ah[5] * X[1] + Ah[6] * X[2] + Ah[7];
  const double xc2 = Ah[8] * X[0] + Ah[9] * X[1] + Ah[10] * X[2] + Ah[11];
  const double fx = K[0], cx = K[1], fy = K[2], cy = K[3];
  const double k1 = K[4], k2 = K[5], p1 = K[6], p2 = K[7], k3 = K[8], k4 = K[9], k5 = K[10], k6 = K[11];
  const double iz = 1.0 / xc2;
  const double xn = xc0 * iz, yn = xc1 * iz;
  const double r2 = xn * xn + yn * yn;
  const double numer = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
  const double denom = 1.0 + k4 * r2 + k5 * r2 * r2 + k6 * r2 * r2 * r2;
  const double idn = 1.0 / denom;
  const double xpp = xn * numer * idn + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
  const double ypp = yn * numer * idn + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);
  r[0] = xpp * fx + cx - u;
  r[1] = ypp * fy + cy - v;
  if (!J) return;
  const double ratio = numer * idn;
  const double dnum = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r2 * r2;
  const double dden = k4 + 2.0 * k5 * r2 + 3.0 * k6 * r2 * r2;
  const double dratio = (dnum - ratio * dden) * idn;
  const double dxx = ratio + 2.0 * xn * xn * dratio + 2.0 * p1 * yn + 6.0 * p2 * xn;
  const double dxy = 2.0 * xn * yn * dratio + 2.0 * p1 * xn + 2.0 * p2 * yn;
  const double dyy = ratio + 2.0 * yn * yn * dratio + 2.0 * p2 * xn + 6.0 * p1 * yn;
  const double fz = fx * iz, gz = fy * iz;
  const double a0 = fz * dxx, a1 = fz * dxy, a2 = -fz * (dxx * xn + dxy * yn);
  const double a3 = gz * dxy, a4 = gz * dyy, a5 = -gz * (dxy * xn + dyy * yn);
  for (int k = 0; k < 3; ++k) {
    J[k] = a0 * Ah[k] + a1 * Ah[4 + k] + a2 * Ah[8 + k];
    J[3 + k] = a3 * Ah[k] + a4 * Ah[4 + k] + a5 * Ah[8 + k];
  }
}

// The two DLT rows of one view (solving-structure.cpp:324-346): with
// T = [2/cols 0 -1; 0 2/rows -1; 0 0 1], P = T K A_hat (3x4) and (x, y) = T (u, v):
//   row1 = x P[2] - P[0],  row2 = y P[2] - P[1];  LHS takes columns 0..2, RHS = -column 3.
// K here is [fx 0 cx; 0 fy cy; 0 0 1] (zero skew, as OpenCV's calibrateCamera returns it).
RP_HD void rae_dlt_rows(const double* Ah, const double* K, double cols, double rows, double u, double v, double row1[4],
                        double row2[4]) {
  const double sx = 2.0 / cols, sy = 2.0 / rows;
  const double x = sx * u - 1.0, y = sy * v - 1.0;
  for (int k = 0; k < 4; ++k) {
    const double ka0 = K[0] * Ah[k] + K[1] * Ah[8 + k];  // (K A_hat) row 0
    const double ka1 = K[2] * Ah[4 + k] + K[3] * Ah[8 + k];
    const double ka2 = Ah[8 + k];
    const double p0 = sx * ka0 - ka2, p1 = sy * ka1 - ka2, p2 = ka2;
    row1[k] = x * p2 - p0;
    row2[k] = y * p2 - p1;
  }
}

}  // namespace rp
