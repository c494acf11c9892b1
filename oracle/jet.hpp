// oracle/jet.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Forward-mode dual number used by the CPU oracle to differentiate the restated
// reprojection functors exactly the way the reference does: the reference wraps
// its functors in ceres::AutoDiffCostFunction<F,2,7[,7[,7]]>
// (/root/reference/src/solving-structure.hpp:795-797,803-805,966-968,1131-1133,1293-1295),
// which evaluates them on ceres::Jet<double,N>.  Ceres is not vendored in the
// reference (Dockerfile:59 clones HEAD) and is absent from this image, so this is
// a restatement of the published Jet algebra (ceres/jet.h): value `a`, derivative
// vector `v`, product rule for *, quotient rule evaluated as
//   g_inv = 1/g.a ; q = f.a*g_inv ; (q, (f.v - q*g.v)*g_inv).
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Jet {
  double a;
  double v[N];

  Jet() : a(0.0) {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  // NOLINTNEXTLINE: implicit on purpose, mirrors T(double) promotions in the functors
  Jet(double s) : a(s) {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  Jet(double s, int k) : a(s) {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
    v[k] = 1.0;
  }

  Jet& operator+=(const Jet& o) {
    a += o.a;
    for (int i = 0; i < N; ++i) v[i] += o.v[i];
    return *this;
  }
  Jet& operator-=(const Jet& o) {
    a -= o.a;
    for (int i = 0; i < N; ++i) v[i] -= o.v[i];
    return *this;
  }
  Jet& operator*=(const Jet& o) {
    *this = *this * o;
    return *this;
  }
  Jet& operator/=(const Jet& o) {
    *this = *this / o;
    return *this;
  }

  friend Jet operator+(const Jet& f, const Jet& g) {
    Jet h;
    h.a = f.a + g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i];
    return h;
  }
  friend Jet operator-(const Jet& f, const Jet& g) {
    Jet h;
    h.a = f.a - g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i];
    return h;
  }
  friend Jet operator-(const Jet& f) {
    Jet h;
    h.a = -f.a;
    for (int i = 0; i < N; ++i) h.v[i] = -f.v[i];
    return h;
  }
  friend Jet operator*(const Jet& f, const Jet& g) {
    Jet h;
    h.a = f.a * g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
  }
  friend Jet operator/(const Jet& f, const Jet& g) {
    Jet h;
    const double g_inv = 1.0 / g.a;
    const double q = f.a * g_inv;
    h.a = q;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - q * g.v[i]) * g_inv;
    return h;
  }
};

// Lets the functor templates be instantiated on plain double as well (cost-only path).
inline double value_of(double x) { return x; }
template <int N>
inline double value_of(const Jet<N>& x) { return x.a; }

}  // namespace oracle
