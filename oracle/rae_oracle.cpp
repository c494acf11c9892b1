// oracle/rae_oracle.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of ReconstructXFunctionIDsMCwithDLTs
// (/root/reference/src/solving-structure.cpp:162-390): for every model corner k seen in at
// least two views, a DLT estimate (normalised image coordinates, least squares through a
// column-pivoted Householder QR as Eigen's colPivHouseholderQr().solve, :309-355) refined by
// ceres::Solve on the 3 coordinates with one ReconstructXStruct residual block per view
// (:295-301, :358-362: DENSE_QR, everything else Ceres defaults, 50 iterations).
// The functor is differentiated with oracle::Jet<3> as AutoDiffCostFunction<.,2,3> does
// (solving-structure.hpp:1391-1394).  The trust-region loop is the restated Ceres loop of
// rp_oracle.cpp (PARITY UNPINNED for the minimiser: Ceres is not in the reference tree); the
// r/J arithmetic is pinned by tests/golden/rae_golden.json.
// Takes the same arguments as calico_rae_reconstruct (include/calico_rp.h).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/calico_rp.h"
#include "jet.hpp"

namespace {

using oracle::Jet;

// ReconstructXStruct::operator(), solving-structure.hpp:1321-1389 (operation order kept)
template <typename T>
void reconstruct_residual(const double* camera_parameters, const double* a_hat /*3x4*/, double im_x, double im_y,
                          const T* parameters, T* residuals) {
  T Xp[4] = {parameters[0], parameters[1], parameters[2], T(1.0)};
  T xp[3];
  for (int i = 0; i < 3; ++i) {  // MatrixMultiply(A_hat_T, Xp, xp, 3, 4, 4, 1), hpp:1368
    T r(0.0);
    for (int k = 0; k < 4; ++k) r += T(a_hat[4 * i + k]) * Xp[k];
    xp[i] = r;
  }
  const T k1(camera_parameters[4]), k2(camera_parameters[5]), p1(camera_parameters[6]), p2(camera_parameters[7]);
  const T k3(camera_parameters[8]), k4(camera_parameters[9]), k5(camera_parameters[10]), k6(camera_parameters[11]);
  xp[0] = xp[0] / xp[2];
  xp[1] = xp[1] / xp[2];
  const T r_sqr = xp[0] * xp[0] + xp[1] * xp[1];
  const T numer = T(1.0) + k1 * r_sqr + k2 * r_sqr * r_sqr + k3 * r_sqr * r_sqr * r_sqr;
  const T denom = T(1.0) + k4 * r_sqr + k5 * r_sqr * r_sqr + k6 * r_sqr * r_sqr * r_sqr;
  const T xpp0 = xp[0] * numer / denom + T(2.0) * p1 * xp[0] * xp[1] + p2 * (r_sqr + T(2.0) * xp[0] * xp[0]);
  const T xpp1 = xp[1] * numer / denom + T(2.0) * p2 * xp[0] * xp[1] + p1 * (r_sqr + T(2.0) * xp[1] * xp[1]);
  residuals[0] = xpp0 * T(camera_parameters[0]) + T(camera_parameters[1]) - T(im_x);
  residuals[1] = xpp1 * T(camera_parameters[2]) + T(camera_parameters[3]) - T(im_y);
}

struct View {
  const double* K;
  const double* Ah;
  double u, v, cols, rows;
};

// least squares min |A x - b| for m x 3 A through Householder QR with column pivoting
void qr_least_squares(std::vector<double>& A /*m x 3 row-major*/, std::vector<double>& b, int m, double x[3]) {
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < 3; ++k) {
    // pivot: remaining column of largest norm
    int best = k;
    double bn = -1;
    for (int j = k; j < 3; ++j) {
      double s = 0;
      for (int i = k; i < m; ++i) s += A[3 * i + j] * A[3 * i + j];
      if (s > bn) { bn = s; best = j; }
    }
    if (best != k) {
      for (int i = 0; i < m; ++i) std::swap(A[3 * i + k], A[3 * i + best]);
      std::swap(perm[k], perm[best]);
    }
    double norm = 0;
    for (int i = k; i < m; ++i) norm += A[3 * i + k] * A[3 * i + k];
    norm = std::sqrt(norm);
    if (norm == 0.0) continue;
    const double alpha = A[3 * k + k] > 0 ? -norm : norm;
    std::vector<double> w(m, 0.0);
    for (int i = k; i < m; ++i) w[i] = A[3 * i + k];
    w[k] -= alpha;
    double wn = 0;
    for (int i = k; i < m; ++i) wn += w[i] * w[i];
    if (wn == 0.0) continue;
    for (int j = k; j < 3; ++j) {
      double d = 0;
      for (int i = k; i < m; ++i) d += w[i] * A[3 * i + j];
      d = 2.0 * d / wn;
      for (int i = k; i < m; ++i) A[3 * i + j] -= d * w[i];
    }
    double d = 0;
    for (int i = k; i < m; ++i) d += w[i] * b[i];
    d = 2.0 * d / wn;
    for (int i = k; i < m; ++i) b[i] -= d * w[i];
  }
  double y[3];
  for (int k = 2; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < 3; ++j) s -= A[3 * k + j] * y[j];
    y[k] = s / A[3 * k + k];
  }
  for (int k = 0; k < 3; ++k) x[perm[k]] = y[k];
}

bool chol3_solve(const double M[9], const double rhs[3], double y[3]) {
  double L[9] = {0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = M[3 * i + j];
      for (int k = 0; k < j; ++k) s -= L[3 * i + k] * L[3 * j + k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[3 * i + i] = std::sqrt(s);
      } else {
        L[3 * i + j] = s / L[3 * j + j];
      }
    }
  double z[3];
  for (int i = 0; i < 3; ++i) {
    double s = rhs[i];
    for (int k = 0; k < i; ++k) s -= L[3 * i + k] * z[k];
    z[i] = s / L[3 * i + i];
  }
  for (int i = 2; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 3; ++k) s -= L[3 * k + i] * y[k];
    y[i] = s / L[3 * i + i];
  }
  return std::isfinite(y[0]) && std::isfinite(y[1]) && std::isfinite(y[2]);
}

struct Eval {
  double cost, g[3], H[9];
};

Eval evaluate(const std::vector<View>& views, const double x[3], bool with_jacobian) {
  Eval e;
  std::memset(&e, 0, sizeof e);
  for (const View& w : views) {
    if (with_jacobian) {
      Jet<3> p[3] = {Jet<3>(x[0], 0), Jet<3>(x[1], 1), Jet<3>(x[2], 2)}, r[2];
      reconstruct_residual(w.K, w.Ah, w.u, w.v, p, r);
      for (int rr = 0; rr < 2; ++rr) {
        e.cost += r[rr].a * r[rr].a;
        for (int a = 0; a < 3; ++a) {
          e.g[a] += r[rr].v[a] * r[rr].a;
          for (int b = 0; b < 3; ++b) e.H[3 * a + b] += r[rr].v[a] * r[rr].v[b];
        }
      }
    } else {
      double r[2];
      reconstruct_residual(w.K, w.Ah, w.u, w.v, x, r);
      e.cost += r[0] * r[0] + r[1] * r[1];
    }
  }
  e.cost *= 0.5;
  return e;
}

}  // namespace

extern "C" int oracle_rae_reconstruct(int32_t /*device*/, int32_t cn, const double* cam_params, const int32_t* image_cols_rows,
                                      int32_t n_frs, const int32_t* fr_cam, const double* fr_a_hat, int32_t n_points,
                                      int32_t /*corners_per_pattern*/, int64_t n_obs, const int32_t* obs_fr,
                                      const int32_t* obs_point, const double* u, const double* v,
                                      const calico_rp_options* opt_in, double* xyz, uint8_t* has_values, int32_t* iterations,
                                      double* final_cost, calico_rae_summary* summary, char* /*err*/, size_t /*err_len*/) {
  calico_rp_options o;
  if (opt_in) {
    o = *opt_in;
  } else {  // Solver::Options defaults (solving-structure.cpp:358-360)
    o = calico_rp_options{50, 1, 0, 0, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 1e-6, 1e-10, 1e-8};
  }
  std::vector<std::vector<View>> per_point(n_points);
  for (int64_t i = 0; i < n_obs; ++i) {
    const int f = obs_fr[i], k = obs_point[i];
    if (f < 0 || f >= n_frs || k < 0 || k >= n_points || fr_cam[f] < 0 || fr_cam[f] >= cn) return CALICO_RP_EINVAL;
    const int c = fr_cam[f];
    per_point[k].push_back(View{cam_params + 12 * c, fr_a_hat + 12 * (size_t)f, u[i], v[i],
                                (double)(float)image_cols_rows[2 * c], (double)(float)image_cols_rows[2 * c + 1]});
  }
  int n_solved = 0, max_it = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : n_solved) reduction(max : max_it)
  for (int k = 0; k < n_points; ++k) {
    const std::vector<View>& views = per_point[k];
    xyz[3 * k] = xyz[3 * k + 1] = xyz[3 * k + 2] = 0.0;  // est_threed_vars zero-initialised, cpp:174-176
    has_values[k] = 0;
    if (iterations) iterations[k] = 0;
    if (final_cost) final_cost[k] = 0.0;
    const int m = 2 * (int)views.size();
    if (views.size() < 2) continue;  // cpp:309
    // ---- DLT, cpp:312-355
    std::vector<double> LHS((size_t)3 * m), RHS(m);
    for (size_t c = 0; c < views.size(); ++c) {
      const View& w = views[c];
      double T[9] = {2.0 / w.cols, 0, -1, 0, 2.0 / w.rows, -1, 0, 0, 1};
      const double K[9] = {w.K[0], 0, w.K[1], 0, w.K[2], w.K[3], 0, 0, 1};
      double KA[12], P[12];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
          double s = 0;
          for (int l = 0; l < 3; ++l) s += K[3 * i + l] * w.Ah[4 * l + j];
          KA[4 * i + j] = s;
        }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
          double s = 0;
          for (int l = 0; l < 3; ++l) s += T[3 * i + l] * KA[4 * l + j];
          P[4 * i + j] = s;
        }
      const double xcam0 = T[0] * w.u + T[2], xcam1 = T[4] * w.v + T[5];
      for (int j = 0; j < 3; ++j) {
        LHS[3 * (2 * c) + j] = xcam0 * P[8 + j] - P[j];
        LHS[3 * (2 * c + 1) + j] = xcam1 * P[8 + j] - P[4 + j];
      }
      RHS[2 * c] = -(xcam0 * P[11] - P[3]);
      RHS[2 * c + 1] = -(xcam1 * P[11] - P[7]);
    }
    double x[3];
    qr_least_squares(LHS, RHS, m, x);
    has_values[k] = 1;
    ++n_solved;

    // ---- ceres::Solve, restated (same loop as rp_oracle.cpp, 3 parameters, dense)
    Eval e = evaluate(views, x, true);
    double cost = e.cost, scale[3] = {1, 1, 1}, diag[3] = {0, 0, 0};
    if (o.jacobi_scaling)
      for (int a = 0; a < 3; ++a) scale[a] = 1.0 / (1.0 + std::sqrt(e.H[4 * a]));
    int iteration = 0, n_succ = 1, n_unsucc = 0, consecutive_invalid = 0;
    double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
    bool reuse_diagonal = false, at_least_one_successful = false, step_was_successful = true;
    double x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (;;) {
      if (iteration >= o.max_num_iterations) break;
      const double gmax = std::max(std::fabs(e.g[0]), std::max(std::fabs(e.g[1]), std::fabs(e.g[2])));
      if (step_was_successful && gmax <= o.gradient_tolerance) break;
      if (radius < o.min_trust_region_radius) break;
      ++iteration;
      step_was_successful = false;
      if (!reuse_diagonal)
        for (int a = 0; a < 3; ++a)
          diag[a] = std::min(std::max(e.H[4 * a] * scale[a] * scale[a], o.min_lm_diagonal), o.max_lm_diagonal);
      double M[9], gs[3], y[3];
      for (int a = 0; a < 3; ++a) {
        gs[a] = e.g[a] * scale[a];
        for (int b = 0; b < 3; ++b) M[3 * a + b] = e.H[3 * a + b] * scale[a] * scale[b];
      }
      double Md[9];
      std::memcpy(Md, M, sizeof M);
      for (int a = 0; a < 3; ++a) Md[4 * a] += diag[a] / radius;
      bool valid = chol3_solve(Md, gs, y);
      reuse_diagonal = true;
      double step[3], model_cost_change = 0;
      if (valid) {
        for (int a = 0; a < 3; ++a) step[a] = -y[a];
        double q = 0;
        for (int a = 0; a < 3; ++a) {
          double Ms = 0;
          for (int b = 0; b < 3; ++b) Ms += M[3 * a + b] * step[b];
          q += step[a] * (gs[a] + 0.5 * Ms);
        }
        model_cost_change = -q;
        if (!(model_cost_change > 0.0)) valid = false;
      }
      if (!valid) {
        ++n_unsucc;
        if (++consecutive_invalid >= 5) break;
        radius /= decrease_factor;
        decrease_factor *= 2.0;
        continue;
      }
      consecutive_invalid = 0;
      double cand[3], step_norm = 0;
      for (int a = 0; a < 3; ++a) {
        cand[a] = x[a] + step[a] * scale[a];
        step_norm += (x[a] - cand[a]) * (x[a] - cand[a]);
      }
      step_norm = std::sqrt(step_norm);
      double cand_cost = evaluate(views, cand, false).cost;
      if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
      if (at_least_one_successful && step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
        ++n_unsucc;
        break;
      }
      const double cost_change = cost - cand_cost;
      if (std::fabs(cost_change) <= o.function_tolerance * cost) {
        ++n_unsucc;
        break;
      }
      const double rho = cost_change / model_cost_change;
      if (rho > o.min_relative_decrease) {
        for (int a = 0; a < 3; ++a) x[a] = cand[a];
        x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        e = evaluate(views, x, true);
        cost = e.cost;
        radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
        decrease_factor = 2.0;
        reuse_diagonal = false;
        at_least_one_successful = true;
        step_was_successful = true;
        ++n_succ;
      } else {
        radius /= decrease_factor;
        decrease_factor *= 2.0;
        ++n_unsucc;
      }
    }
    for (int a = 0; a < 3; ++a) xyz[3 * k + a] = x[a];
    if (iterations) iterations[k] = n_succ + n_unsucc;
    if (final_cost) final_cost[k] = cost;
    max_it = std::max(max_it, n_succ + n_unsucc);
  }
  if (summary) {
    std::memset(summary, 0, sizeof *summary);
    summary->n_points = n_points;
    summary->n_solved = n_solved;
    summary->max_point_iterations = max_it;
  }
  return 0;
}

// parity hook: one ReconstructXStruct block through Jet<3>: r[2], J[2][3]
extern "C" void oracle_rae_eval(const double* cam_params, const double* a_hat, const double* X, double u, double v,
                                double* r, double* J) {
  Jet<3> p[3] = {Jet<3>(X[0], 0), Jet<3>(X[1], 1), Jet<3>(X[2], 2)}, res[2];
  reconstruct_residual(cam_params, a_hat, u, v, p, res);
  for (int rr = 0; rr < 2; ++rr) {
    r[rr] = res[rr].a;
    for (int k = 0; k < 3; ++k) J[3 * rr + k] = res[rr].v[k];
  }
}
