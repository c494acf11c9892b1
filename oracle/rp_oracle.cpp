// oracle/rp_oracle.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU oracle for the Stage-5 reprojection-error LM refinement of amy-tabb/calico.
// It is the checker for the HIP path and the "port" CPU baseline of bench.py; the
// product never links, imports or calls it.
//
// What is restated, and from where:
//  * residual blocks / case split / block order:
//      /root/reference/src/solving-structure.cpp:506-668
//  * cost functors (evaluated on dual numbers, as AutoDiffCostFunction does):
//      /root/reference/src/solving-structure.hpp:643-1310  (rp_functors.hpp)
//  * solver options: solving-structure.cpp:973-982 (DENSE_SCHUR, max iterations,
//      everything else Ceres defaults)
//  * the minimiser itself is Ceres Solver, a third-party dependency that is NOT in
//      /root/reference (un-vendored, unpinned: Dockerfile:59 clones HEAD;
//      CMakeLists.txt:12) and not installed in this image.  trust-region LM,
//      Jacobi scaling, LM diagonal clamping, radius schedule, termination tests and
//      the Schur elimination order below are restated from Ceres 2.x's published
//      algorithm (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
//      schur_eliminator_impl.h).  The reference holds no golden vector, test or
//      fixture for this arithmetic  ==>  PARITY UNPINNED for the LM/Schur/Cholesky
//      iterates.  The residual/Jacobian part is pinned independently by the mpmath
//      golden vectors in tests/golden/ (oracle/gen_golden.py).
//
// Build: make -C oracle   ->  oracle/_build/liboracle_rp.so
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/calico_rp.h"
#include "ae_functors.hpp"
#include "rp_functors.hpp"

namespace {

using oracle::Jet;

struct Problem {
  int cn = 0, pn = 0, tn = 0, vn = 0, p_star = -1, t_star = -1;
  std::vector<double> intr;  // 12*cn
  std::vector<float> pts;    // 3*n
  std::vector<int32_t> cam, pat, tim, corner;
  std::vector<double> u, v;
  // kind 1 (algebraic error, Stage 4): one residual block of 12 rows per FR, a7 = its A
  int kind = 0;
  std::vector<double> a7;
  int rows() const { return kind ? 12 : 2; }
  // Extension (BASELINE.json configs[2], "intrinsics + distortion refined"; NOT in the
  // reference, where the 12 camera parameters are constants inside the functor,
  // solving-structure.hpp:738-749): bit k set = parameter k of every camera
  // [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6] is a free variable.  The 12 numbers of camera c
  // then form a fourth parameter block of every residual of that camera, as
  // AddResidualBlock(cost, NULL, c_, p_, t_, &camera_params[12 * c]) would make them; the
  // masked-out entries are constants inside the functor (what ceres::SubsetManifold does).
  // The extended parameter vector is [7 * vn pose parameters | 12 * cn camera parameters].
  uint32_t refine_mask = 0;
  bool refine() const { return refine_mask != 0 && kind == 0; }
  int ncols() const { return refine() ? 33 : 21; }  // Jacobian columns per residual row
  int nx() const { return 7 * vn + (refine() ? 12 * cn : 0); }
  std::string err;
};

// The parameter blocks of observation i: Jacobian columns [col, col + size) of a row belong
// to the entries [slot, slot + size) of the extended parameter vector.
struct Blk {
  int col, size, slot;
};
inline int obs_blocks(const Problem& P, int cam, int pat, int tim, Blk* b) {
  b[0] = Blk{0, 7, 7 * cam};
  b[1] = Blk{7, 7, 7 * pat};
  b[2] = Blk{14, 7, 7 * tim};
  if (!P.refine()) return 3;
  b[3] = Blk{21, 12, 7 * P.vn + 12 * cam};
  return 4;
}

inline int case_of(const Problem& P, int64_t i) {
  // solving-structure.cpp:528-544
  const bool ps = P.pat[i] == P.p_star, ts = P.tim[i] == P.t_star;
  return (ps ? 1 : 0) + (ts ? 2 : 0);
}

// One AutoDiffCostFunction::Evaluate: residual (2) and Jacobian rows in the
// [camera | pattern | time] block layout (zeros for absent blocks).
template <int NJ>
void ae_block(const double* c, const double* p, const double* t, const double* a7, double* r, double* J,
              int off_p, int off_t) {
  using T = Jet<NJ>;
  T C[7], Pv[7], Tv[7], res[12];
  int next = 0;
  for (int k = 0; k < 7; ++k) C[k] = T(c[k], next + k);
  next += 7;
  if (p) {
    for (int k = 0; k < 7; ++k) Pv[k] = T(p[k], next + k);
    next += 7;
  }
  if (t)
    for (int k = 0; k < 7; ++k) Tv[k] = T(t[k], next + k);
  oracle::algebraic_residual<T>(C, p ? Pv : nullptr, t ? Tv : nullptr, a7, res);
  for (int rr = 0; rr < 12; ++rr) {
    r[rr] = res[rr].a;
    if (!J) continue;
    int col = 0;
    for (int k = 0; k < 7; ++k) J[rr * 21 + k] = res[rr].v[col + k];
    col += 7;
    if (p) {
      for (int k = 0; k < 7; ++k) J[rr * 21 + off_p + k] = res[rr].v[col + k];
      col += 7;
    }
    if (t)
      for (int k = 0; k < 7; ++k) J[rr * 21 + off_t + k] = res[rr].v[col + k];
  }
}

void eval_block(const Problem& P, const double* x, int64_t i, double* r, double* J /*rows*21 or null*/) {
  const double* c = x + 7 * (int64_t)P.cam[i];
  const double* p = x + 7 * (int64_t)P.pat[i];
  const double* t = x + 7 * (int64_t)P.tim[i];
  if (P.kind == 1) {
    // AutoDiffCostFunction<AlgebraicError..., 12, 7[, 7[, 7]]> (solving-structure.hpp:228,309,385,461)
    const int cs = case_of(P, i);
    const double* a7 = &P.a7[7 * (size_t)i];
    if (J)
      for (int k = 0; k < 12 * 21; ++k) J[k] = 0.0;
    switch (cs) {
      case 0: ae_block<21>(c, p, t, a7, r, J, 7, 14); break;
      case 1: ae_block<14>(c, nullptr, t, a7, r, J, 7, 14); break;
      case 2: ae_block<14>(c, p, nullptr, a7, r, J, 7, 14); break;
      default: ae_block<7>(c, nullptr, nullptr, a7, r, J, 7, 14); break;
    }
    return;
  }
  const float* X = &P.pts[3 * (size_t)P.corner[i]];
  const int cs = case_of(P, i);
  if (P.refine()) {
    // the camera parameters are the tail of the extended parameter vector
    const double* Kx = x + 7 * (size_t)P.vn + 12 * (size_t)P.cam[i];
    if (!J) {
      double res[2];
      oracle::reprojection_residual<double>(c, (cs & 1) ? nullptr : p, (cs & 2) ? nullptr : t, Kx, P.u[i], P.v[i], X, res);
      r[0] = res[0];
      r[1] = res[1];
      return;
    }
    // one functor evaluation on dual numbers over [camera | pattern | time | camera parameters];
    // the slots of an absent (p*, t*) block and of masked-out camera parameters stay unused
    using T = Jet<33>;
    T C[7], Pv[7], Tv[7], Kv[12], res[2];
    for (int k = 0; k < 7; ++k) {
      C[k] = T(c[k], k);
      if (!(cs & 1)) Pv[k] = T(p[k], 7 + k);
      if (!(cs & 2)) Tv[k] = T(t[k], 14 + k);
    }
    for (int k = 0; k < 12; ++k) Kv[k] = ((P.refine_mask >> k) & 1u) ? T(Kx[k], 21 + k) : T(Kx[k]);
    oracle::reprojection_residual<T, T>(C, (cs & 1) ? nullptr : Pv, (cs & 2) ? nullptr : Tv, Kv, P.u[i], P.v[i], X, res);
    for (int rr = 0; rr < 2; ++rr) {
      r[rr] = res[rr].a;
      for (int k = 0; k < 33; ++k) J[rr * 33 + k] = res[rr].v[k];
    }
    return;
  }
  const double* K = &P.intr[12 * (size_t)P.cam[i]];
  if (!J) {
    double res[2];
    oracle::reprojection_residual<double>(c, (cs & 1) ? nullptr : p, (cs & 2) ? nullptr : t, K,
                                          P.u[i], P.v[i], X, res);
    r[0] = res[0];
    r[1] = res[1];
    return;
  }
  for (int k = 0; k < 42; ++k) J[k] = 0.0;
  switch (cs) {
    case 0: {  // MultiCameraReprojectionError, AutoDiff<.,2,7,7,7> (hpp:803-805)
      using T = Jet<21>;
      T C[7], Pv[7], Tv[7], res[2];
      for (int k = 0; k < 7; ++k) {
        C[k] = T(c[k], k);
        Pv[k] = T(p[k], 7 + k);
        Tv[k] = T(t[k], 14 + k);
      }
      oracle::reprojection_residual<T>(C, Pv, Tv, K, P.u[i], P.v[i], X, res);
      for (int rr = 0; rr < 2; ++rr) {
        r[rr] = res[rr].a;
        for (int k = 0; k < 21; ++k) J[rr * 21 + k] = res[rr].v[k];
      }
    } break;
    case 1: {  // ...ErrorCamTime, AutoDiff<.,2,7,7> (hpp:1131-1133): blocks (camera, time)
      using T = Jet<14>;
      T C[7], Tv[7], res[2];
      for (int k = 0; k < 7; ++k) {
        C[k] = T(c[k], k);
        Tv[k] = T(t[k], 7 + k);
      }
      oracle::reprojection_residual<T>(C, nullptr, Tv, K, P.u[i], P.v[i], X, res);
      for (int rr = 0; rr < 2; ++rr) {
        r[rr] = res[rr].a;
        for (int k = 0; k < 7; ++k) {
          J[rr * 21 + k] = res[rr].v[k];
          J[rr * 21 + 14 + k] = res[rr].v[7 + k];
        }
      }
    } break;
    case 2: {  // ...ErrorCamPattern::Create (hpp:1290-1297) builds the CamTime functor
               // with the pattern block as its second argument: blocks (camera, pattern).
      using T = Jet<14>;
      T C[7], Pv[7], res[2];
      for (int k = 0; k < 7; ++k) {
        C[k] = T(c[k], k);
        Pv[k] = T(p[k], 7 + k);
      }
      // second slot of the CamTime functor is the "time" matrix: A_hat = C * M(p) * I
      oracle::reprojection_residual<T>(C, nullptr, Pv, K, P.u[i], P.v[i], X, res);
      for (int rr = 0; rr < 2; ++rr) {
        r[rr] = res[rr].a;
        for (int k = 0; k < 7; ++k) {
          J[rr * 21 + k] = res[rr].v[k];
          J[rr * 21 + 7 + k] = res[rr].v[7 + k];
        }
      }
    } break;
    default: {  // ...ErrorCam, AutoDiff<.,2,7> (hpp:966-968)
      using T = Jet<7>;
      T C[7], res[2];
      for (int k = 0; k < 7; ++k) C[k] = T(c[k], k);
      oracle::reprojection_residual<T>(C, nullptr, nullptr, K, P.u[i], P.v[i], X, res);
      for (int rr = 0; rr < 2; ++rr) {
        r[rr] = res[rr].a;
        for (int k = 0; k < 7; ++k) J[rr * 21 + k] = res[rr].v[k];
      }
    } break;
  }
}

double eval_all(const Problem& P, const double* x, double* r, double* J) {
  const int64_t N = (int64_t)P.cam.size();
  double cost = 0.0;
  const int R = P.rows();
#pragma omp parallel for schedule(static) reduction(+ : cost)
  for (int64_t i = 0; i < N; ++i) {
    double rl[12];
    eval_block(P, x, i, rl, J ? J + (int64_t)P.ncols() * R * i : nullptr);
    for (int k = 0; k < R; ++k) {
      if (r) r[(int64_t)R * i + k] = rl[k];
      cost += rl[k] * rl[k];
    }
  }
  return 0.5 * cost;
}

// ---------------------------------------------------------------- dense helpers
// In-place lower Cholesky of a row-major n x n SPD matrix (Eigen LLT stand-in:
// Ceres' DENSE_SCHUR with the default EIGEN dense backend factorises the reduced
// system with LLT).  Returns false if a pivot is not positive / finite.
bool cholesky_lower(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    const double inv = 1.0 / d;
#pragma omp parallel for schedule(static) if (n - j > 256)
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s * inv;
    }
  }
  return true;
}
void chol_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}

// ---------------------------------------------------------------- problem structure
struct FR {
  int c, p, t;  // variable indices; p/t may be p*/t*
  std::vector<int64_t> obs;
};

struct Structure {
  std::vector<FR> frs;
  std::vector<int> ridx;  // variable -> reduced block index (cameras+patterns), -1 if none
  std::vector<int> tidx;  // variable -> eliminated (time) block index, -1 if none
  std::vector<int> kidx;  // camera -> index of its camera-parameter block (refinement only), -1 if none
  std::vector<int> rvar, tvar, kcam;
  int nr = 0, nt = 0, nk = 0;
  // reduced unknowns: the 7-parameter camera / pattern blocks in variable order, then (camera
  // parameters refined) the 12-parameter blocks of the observed cameras in camera order
  int n() const { return 7 * nr + 12 * nk; }
  std::vector<int> rslot;  // reduced index -> entry of the extended parameter vector
};

Structure build_structure(const Problem& P) {
  Structure S;
  std::map<std::tuple<int, int, int>, int> ids;  // (t, c, p) -> fr
  const int64_t N = (int64_t)P.cam.size();
  for (int64_t i = 0; i < N; ++i) {
    auto key = std::make_tuple(P.tim[i], P.cam[i], P.pat[i]);
    auto it = ids.find(key);
    if (it == ids.end()) {
      it = ids.emplace(key, (int)S.frs.size()).first;
      S.frs.push_back(FR{P.cam[i], P.pat[i], P.tim[i], {}});
    }
    S.frs[it->second].obs.push_back(i);
  }
  std::vector<char> used(P.vn, 0);
  for (auto& f : S.frs) {
    used[f.c] = 1;
    if (f.p != P.p_star) used[f.p] = 1;
    if (f.t != P.t_star) used[f.t] = 1;
  }
  S.ridx.assign(P.vn, -1);
  S.tidx.assign(P.vn, -1);
  S.kidx.assign(P.cn, -1);
  for (int v = 0; v < P.cn + P.pn; ++v)
    if (used[v]) {
      S.ridx[v] = S.nr++;
      S.rvar.push_back(v);
    }
  for (int v = P.cn + P.pn; v < P.vn; ++v)
    if (used[v]) {
      S.tidx[v] = S.nt++;
      S.tvar.push_back(v);
    }
  if (P.refine())
    for (int c = 0; c < P.cn; ++c)
      if (used[c]) {
        S.kidx[c] = S.nk++;
        S.kcam.push_back(c);
      }
  for (int ib = 0; ib < S.nr; ++ib)
    for (int k = 0; k < 7; ++k) S.rslot.push_back(7 * S.rvar[ib] + k);
  for (int kb = 0; kb < S.nk; ++kb)
    for (int k = 0; k < 12; ++k) S.rslot.push_back(7 * P.vn + 12 * S.kcam[kb] + k);
  return S;
}

// offset of a block of an FR in the reduced system, or -1 (eliminated time block, p*, t*)
inline int reduced_offset(const Problem& P, const Structure& S, const FR& fr, int which) {
  switch (which) {
    case 0: return 7 * S.ridx[fr.c];
    case 1: return fr.p == P.p_star ? -1 : 7 * S.ridx[fr.p];
    case 3: return 7 * S.nr + 12 * S.kidx[fr.c];
    default: return -1;
  }
}

// Normal equations of the (column-scaled) Jacobian in Schur-ready block form.
struct Coupling {
  int size = 0;           // rows of the reduced block
  std::vector<double> m;  // size x 7, [reduced parameter][time parameter]
};
struct Normal {
  int n = 0;                      // reduced unknowns
  std::vector<double> Hrr;        // n x n (full symmetric)
  std::vector<double> gr;         // n
  std::vector<double> Ett;        // nt * 49
  std::vector<double> gt;         // nt * 7
  // coupling blocks H_rt: per time, reduced offset -> block
  std::vector<std::map<int, Coupling>> Hrt;
};

// scale: column scaling over the extended parameter vector (or null).  J is the UNSCALED
// per-observation Jacobian.
Normal build_normal(const Problem& P, const Structure& S, const double* J, const double* r,
                    const double* scale) {
  Normal Nq;
  Nq.n = S.n();
  Nq.Hrr.assign((size_t)Nq.n * Nq.n, 0.0);
  Nq.gr.assign(Nq.n, 0.0);
  Nq.Ett.assign((size_t)S.nt * 49, 0.0);
  Nq.gt.assign((size_t)S.nt * 7, 0.0);
  Nq.Hrt.resize(S.nt);
  const int F = (int)S.frs.size();
  const int W = P.ncols(), R = P.rows();
  std::vector<double> Wm((size_t)F * W * W), wv((size_t)F * W);
#pragma omp parallel for schedule(dynamic, 16)
  for (int f = 0; f < F; ++f) {
    const FR& fr = S.frs[f];
    double* Wf = &Wm[(size_t)f * W * W];
    double* wf = &wv[(size_t)f * W];
    for (int k = 0; k < W * W; ++k) Wf[k] = 0.0;
    for (int k = 0; k < W; ++k) wf[k] = 0.0;
    Blk blk[4];
    const int nb = obs_blocks(P, fr.c, fr.p, fr.t, blk);
    double sc[33];
    for (int b = 0; b < nb; ++b)
      for (int k = 0; k < blk[b].size; ++k) sc[blk[b].col + k] = scale ? scale[blk[b].slot + k] : 1.0;
    for (int64_t i : fr.obs) {
      for (int rr = 0; rr < R; ++rr) {
        double row[33];
        for (int k = 0; k < W; ++k) row[k] = J[(int64_t)W * R * i + W * rr + k] * sc[k];
        const double ri = r[(int64_t)R * i + rr];
        for (int a = 0; a < W; ++a) {
          wf[a] += row[a] * ri;
          for (int b = 0; b < W; ++b) Wf[a * W + b] += row[a] * row[b];
        }
      }
    }
  }
  for (int f = 0; f < F; ++f) {
    const FR& fr = S.frs[f];
    const double* Wf = &Wm[(size_t)f * W * W];
    const double* wf = &wv[(size_t)f * W];
    Blk blk[4];
    const int nb = obs_blocks(P, fr.c, fr.p, fr.t, blk);
    const int n = Nq.n;
    // reduced x reduced
    for (int a = 0; a < nb; ++a) {
      const int oa = reduced_offset(P, S, fr, a);
      if (oa < 0) continue;
      for (int k = 0; k < blk[a].size; ++k) Nq.gr[oa + k] += wf[blk[a].col + k];
      for (int b = 0; b < nb; ++b) {
        const int ob = reduced_offset(P, S, fr, b);
        if (ob < 0) continue;
        for (int k = 0; k < blk[a].size; ++k)
          for (int l = 0; l < blk[b].size; ++l)
            Nq.Hrr[(size_t)(oa + k) * n + ob + l] += Wf[(blk[a].col + k) * W + blk[b].col + l];
      }
    }
    if (fr.t != P.t_star) {
      const int it = S.tidx[fr.t];
      for (int k = 0; k < 7; ++k) {
        Nq.gt[7 * it + k] += wf[14 + k];
        for (int l = 0; l < 7; ++l) Nq.Ett[(size_t)it * 49 + 7 * k + l] += Wf[(14 + k) * W + 14 + l];
      }
      for (int a = 0; a < nb; ++a) {
        const int oa = reduced_offset(P, S, fr, a);
        if (oa < 0) continue;
        Coupling& cb = Nq.Hrt[it][oa];
        if (cb.m.empty()) {
          cb.size = blk[a].size;
          cb.m.assign((size_t)cb.size * 7, 0.0);
        }
        for (int k = 0; k < cb.size; ++k)
          for (int l = 0; l < 7; ++l) cb.m[7 * k + l] += Wf[(blk[a].col + k) * W + 14 + l];
      }
    }
  }
  return Nq;
}

// Solve (H + D^2) y = g by eliminating the time blocks first (Ceres DENSE_SCHUR:
// e-blocks = an independent set of the Hessian graph; the time blocks never share
// a residual, solving-structure.cpp:546-548).  D2 is the squared LM diagonal over the
// extended parameter vector.  Optionally returns the reduced system.
bool schur_solve(const Problem& P, const Structure& S, const Normal& Nq, const double* D2,
                 double* y /*nx, zero where inactive*/, std::vector<double>* S_out,
                 std::vector<double>* rhs_out) {
  const int n = Nq.n;
  std::vector<double> Sm(Nq.Hrr), rhs(Nq.gr);
  for (int i = 0; i < n; ++i) Sm[(size_t)i * n + i] += D2[S.rslot[i]];
  std::vector<double> Einv((size_t)S.nt * 49);
  for (int it = 0; it < S.nt; ++it) {
    double E[49];
    for (int k = 0; k < 49; ++k) E[k] = Nq.Ett[(size_t)it * 49 + k];
    for (int k = 0; k < 7; ++k) E[8 * k] += D2[7 * S.tvar[it] + k];
    if (!cholesky_lower(E, 7)) return false;
    // E^-1 via LLT solves of the identity (InvertPSDMatrix in Ceres)
    double* Ei = &Einv[(size_t)it * 49];
    for (int col = 0; col < 7; ++col) {
      double e[7] = {0, 0, 0, 0, 0, 0, 0};
      e[col] = 1.0;
      chol_solve(E, 7, e);
      for (int k = 0; k < 7; ++k) Ei[7 * k + col] = e[k];
    }
    // S -= B Einv B^T ; rhs -= B Einv g_t
    double Eg[7];
    for (int k = 0; k < 7; ++k) {
      double s = 0;
      for (int l = 0; l < 7; ++l) s += Ei[7 * k + l] * Nq.gt[7 * it + l];
      Eg[k] = s;
    }
    for (auto& bi : Nq.Hrt[it]) {
      const int ni = bi.second.size;
      double BE[12 * 7];
      for (int k = 0; k < ni; ++k)
        for (int l = 0; l < 7; ++l) {
          double s = 0;
          for (int m = 0; m < 7; ++m) s += bi.second.m[7 * k + m] * Ei[7 * m + l];
          BE[7 * k + l] = s;
        }
      for (int k = 0; k < ni; ++k) {
        double s = 0;
        for (int l = 0; l < 7; ++l) s += bi.second.m[7 * k + l] * Eg[l];
        rhs[bi.first + k] -= s;
      }
      for (auto& bj : Nq.Hrt[it]) {
        const int nj = bj.second.size;
        for (int k = 0; k < ni; ++k)
          for (int l = 0; l < nj; ++l) {
            double s = 0;
            for (int m = 0; m < 7; ++m) s += BE[7 * k + m] * bj.second.m[7 * l + m];
            Sm[(size_t)(bi.first + k) * n + bj.first + l] -= s;
          }
      }
    }
  }
  if (S_out) *S_out = Sm;
  if (rhs_out) *rhs_out = rhs;
  if (!y) return true;
  const int nx = P.nx();
  for (int k = 0; k < nx; ++k) y[k] = 0.0;
  if (n > 0) {
    if (!cholesky_lower(Sm.data(), n)) return false;
    chol_solve(Sm.data(), n, rhs.data());
    for (int i = 0; i < n; ++i) y[S.rslot[i]] = rhs[i];
  }
  // back-substitution: y_t = Einv (g_t - B^T y_r)
  for (int it = 0; it < S.nt; ++it) {
    double q[7];
    for (int k = 0; k < 7; ++k) q[k] = Nq.gt[7 * it + k];
    for (auto& bi : Nq.Hrt[it])
      for (int k = 0; k < bi.second.size; ++k)
        for (int l = 0; l < 7; ++l) q[l] -= bi.second.m[7 * k + l] * rhs[bi.first + k];
    const double* Ei = &Einv[(size_t)it * 49];
    for (int k = 0; k < 7; ++k) {
      double s = 0;
      for (int l = 0; l < 7; ++l) s += Ei[7 * k + l] * q[l];
      y[7 * S.tvar[it] + k] = s;
    }
  }
  for (int k = 0; k < nx; ++k)
    if (!std::isfinite(y[k])) return false;
  return true;
}

// Squared column norms of the scaled Jacobian, per entry of the extended parameter vector.
void column_sqnorm(const Problem& P, const double* J, const double* scale, std::vector<double>& out) {
  out.assign((size_t)P.nx(), 0.0);
  const int64_t N = (int64_t)P.cam.size();
  const int R = P.rows(), W = P.ncols();
  for (int64_t i = 0; i < N; ++i) {
    Blk blk[4];
    const int nb = obs_blocks(P, P.cam[i], P.pat[i], P.tim[i], blk);
    for (int b = 0; b < nb; ++b)
      for (int k = 0; k < blk[b].size; ++k) {
        const double s = scale ? scale[blk[b].slot + k] : 1.0;
        for (int rr = 0; rr < R; ++rr) {
          const double a = J[(int64_t)W * R * i + W * rr + blk[b].col + k] * s;
          out[blk[b].slot + k] += a * a;
        }
      }
  }
}

// [x (7 * vn) | camera parameters (12 * cn)] when they are refined, else x
std::vector<double> extended_x(const Problem& P, const double* x) {
  std::vector<double> xe(x, x + 7 * (size_t)P.vn);
  if (P.refine()) xe.insert(xe.end(), P.intr.begin(), P.intr.end());
  return xe;
}

struct LmState {
  std::vector<double> x, r, J, scale, diag, g;
  double cost = 0;
};

}  // namespace

// ------------------------------------------------------------------------ C ABI
extern "C" {

struct oracle_rp_problem {
  Problem P;
};

int oracle_rp_create(int32_t cn, int32_t pn, int32_t tn, int32_t p_star_var, int32_t t_star_var,
                     int32_t /*device: unused on the CPU*/, oracle_rp_problem** out) {
  if (!out || cn <= 0 || pn <= 0 || tn <= 0) return CALICO_RP_EINVAL;
  auto* h = new oracle_rp_problem;
  h->P.cn = cn;
  h->P.pn = pn;
  h->P.tn = tn;
  h->P.vn = cn + pn + tn;
  h->P.p_star = p_star_var;
  h->P.t_star = t_star_var;
  h->P.intr.assign((size_t)12 * cn, 0.0);
  *out = h;
  return 0;
}
int oracle_rp_destroy(oracle_rp_problem* h) {
  delete h;
  return 0;
}
int oracle_rp_set_intrinsics(oracle_rp_problem* h, const double* k) {
  std::copy(k, k + 12 * h->P.cn, h->P.intr.begin());
  return 0;
}
int oracle_rp_set_points(oracle_rp_problem* h, const float* xyz, int32_t n) {
  h->P.pts.assign(xyz, xyz + 3 * (size_t)n);
  return 0;
}
int oracle_rp_add_observations(oracle_rp_problem* h, int64_t n, const int32_t* cam,
                               const int32_t* pat, const int32_t* tim, const int32_t* corner,
                               const double* u, const double* v) {
  Problem& P = h->P;
  for (int64_t i = 0; i < n; ++i) {
    if (cam[i] < 0 || cam[i] >= P.cn || pat[i] < P.cn || pat[i] >= P.cn + P.pn ||
        tim[i] < P.cn + P.pn || tim[i] >= P.vn || corner[i] < 0 ||
        (size_t)corner[i] * 3 >= P.pts.size())
      return CALICO_RP_EINVAL;
  }
  P.cam.insert(P.cam.end(), cam, cam + n);
  P.pat.insert(P.pat.end(), pat, pat + n);
  P.tim.insert(P.tim.end(), tim, tim + n);
  P.corner.insert(P.corner.end(), corner, corner + n);
  P.u.insert(P.u.end(), u, u + n);
  P.v.insert(P.v.end(), v, v + n);
  return 0;
}
int64_t oracle_rp_num_observations(const oracle_rp_problem* h) { return (int64_t)h->P.cam.size(); }

// camera parameters refined (extension, see Problem::refine_mask); 0 = the reference's constants
int oracle_rp_set_refine_intrinsics(oracle_rp_problem* h, uint32_t mask) {
  if (mask > 0xFFFu) return CALICO_RP_EINVAL;
  h->P.refine_mask = mask;
  return 0;
}
int oracle_rp_get_intrinsics(const oracle_rp_problem* h, double* k) {
  std::copy(h->P.intr.begin(), h->P.intr.end(), k);
  return 0;
}

// Stage 4 (algebraic error): same handle, one 12-row residual block per FR
int oracle_ae_create(int32_t cn, int32_t pn, int32_t tn, int32_t p_star_var, int32_t t_star_var, int32_t device,
                     oracle_rp_problem** out) {
  const int rc = oracle_rp_create(cn, pn, tn, p_star_var, t_star_var, device, out);
  if (rc == 0) (*out)->P.kind = 1;
  return rc;
}
int oracle_ae_add_relations(oracle_rp_problem* h, int64_t n, const int32_t* cam, const int32_t* pat,
                            const int32_t* tim, const double* a7) {
  Problem& P = h->P;
  if (P.kind != 1) return CALICO_RP_ESTATE;
  for (int64_t i = 0; i < n; ++i)
    if (cam[i] < 0 || cam[i] >= P.cn || pat[i] < P.cn || pat[i] >= P.cn + P.pn || tim[i] < P.cn + P.pn ||
        tim[i] >= P.vn)
      return CALICO_RP_EINVAL;
  P.cam.insert(P.cam.end(), cam, cam + n);
  P.pat.insert(P.pat.end(), pat, pat + n);
  P.tim.insert(P.tim.end(), tim, tim + n);
  P.a7.insert(P.a7.end(), a7, a7 + 7 * n);
  return 0;
}

// J: 2 x 21 per observation, 2 x 33 ([camera | pattern | time | camera parameters]) with refinement
int oracle_rp_eval(oracle_rp_problem* h, const double* x, double* cost, double* r, double* J) {
  const std::vector<double> xe = extended_x(h->P, x);
  const double c = eval_all(h->P, xe.data(), r, J);
  if (cost) *cost = c;
  return 0;
}

int oracle_ae_eval(oracle_rp_problem* h, const double* x, double* cost, double* r, double* J) {
  if (h->P.kind != 1) return CALICO_RP_ESTATE;
  return oracle_rp_eval(h, x, cost, r, J);
}

int oracle_rp_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

int oracle_rp_normal_eq(oracle_rp_problem* h, const double* x, double radius, int32_t jacobi,
                        int32_t* n_out, double* S_out, double* rhs_out) {
  const Problem& P = h->P;
  Structure S = build_structure(P);
  if (n_out) *n_out = S.n();
  if (!S_out && !rhs_out) return 0;
  const int64_t N = (int64_t)P.cam.size();
  std::vector<double> r((size_t)P.rows() * N), J((size_t)P.ncols() * P.rows() * N);
  const std::vector<double> xe = extended_x(P, x);
  eval_all(P, xe.data(), r.data(), J.data());
  std::vector<double> scale, cs;
  if (jacobi) {
    column_sqnorm(P, J.data(), nullptr, cs);
    scale.resize(cs.size());
    for (size_t k = 0; k < cs.size(); ++k) scale[k] = 1.0 / (1.0 + std::sqrt(cs[k]));
  }
  Normal Nq = build_normal(P, S, J.data(), r.data(), jacobi ? scale.data() : nullptr);
  column_sqnorm(P, J.data(), jacobi ? scale.data() : nullptr, cs);
  std::vector<double> D2(cs.size());
  for (size_t k = 0; k < cs.size(); ++k) D2[k] = std::min(std::max(cs[k], 1e-6), 1e32) / radius;
  std::vector<double> Sm, rhs;
  if (!schur_solve(P, S, Nq, D2.data(), nullptr, &Sm, &rhs)) return CALICO_RP_ENUMERIC;
  if (S_out) std::copy(Sm.begin(), Sm.end(), S_out);
  if (rhs_out) std::copy(rhs.begin(), rhs.end(), rhs_out);
  return 0;
}

// ceres::Solve with the options of solving-structure.cpp:973-979 (restated; see header).
// iter_ms (optional, length >= max_num_iterations+1) receives per-iteration wall ms.
int oracle_rp_solve_timed(oracle_rp_problem* h, double* x, const calico_rp_options* opt,
                          calico_rp_summary* sum, double* jac_ms, double* iter_ms) {
  Problem& P = h->P;
  const auto t_start = std::chrono::steady_clock::now();
  Structure S = build_structure(P);
  const int64_t N = (int64_t)P.cam.size();
  const int nv = P.nx();  // with refinement the camera parameters ride behind the pose blocks
  const int R = P.rows(), W = P.ncols();
  std::vector<double> xc = extended_x(P, x), cand(nv), r((size_t)R * N), J((size_t)W * R * N), g(nv), scale(nv, 1.0), diag,
      step(nv), delta(nv), y(nv);
  calico_rp_summary s;
  std::memset(&s, 0, sizeof(s));
  s.num_residual_blocks = N;
  double jac_total_ms = 0;
  int n_jac = 0;

  auto evaluate = [&](const std::vector<double>& xx) -> double {
    const auto t0 = std::chrono::steady_clock::now();
    const double c = eval_all(P, xx.data(), r.data(), J.data());
    jac_total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ++n_jac;
    // gradient of the UNSCALED problem (what Ceres' evaluator returns)
    std::fill(g.begin(), g.end(), 0.0);
    for (int64_t i = 0; i < N; ++i) {
      Blk blk[4];
      const int nb = obs_blocks(P, P.cam[i], P.pat[i], P.tim[i], blk);
      for (int b = 0; b < nb; ++b)
        for (int k = 0; k < blk[b].size; ++k)
          for (int rr = 0; rr < R; ++rr)
            g[blk[b].slot + k] += J[(int64_t)W * R * i + W * rr + blk[b].col + k] * r[(int64_t)R * i + rr];
    }
    return c;
  };
  auto gmax = [&]() {
    double m = 0;
    for (double a : g) m = std::max(m, std::fabs(a));
    return m;
  };

  // ---- iteration 0 (TrustRegionMinimizer::IterationZero)
  double cost = evaluate(xc);
  if (opt->jacobi_scaling) {
    std::vector<double> cs;
    column_sqnorm(P, J.data(), nullptr, cs);
    for (int k = 0; k < nv; ++k) scale[k] = 1.0 / (1.0 + std::sqrt(cs[k]));
  }
  s.initial_cost = cost;
  s.num_successful_steps = 1;
  int iteration = 0;
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false, at_least_one_successful = false, step_was_successful = true;
  int consecutive_invalid = 0;
  double x_norm = 0;
  for (double a : xc) x_norm += a * a;
  x_norm = std::sqrt(x_norm);
  s.termination_type = CALICO_RP_NO_CONVERGENCE;
  std::snprintf(s.message, sizeof s.message, "Maximum number of iterations reached.");
  if (opt->verbose)
    std::printf("oracle iter %3d cost %.12e |g|max %.3e radius %.3e\n", 0, cost, gmax(), radius);
  auto t_iter = std::chrono::steady_clock::now();

  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iter_ms && iteration > 0) {
      auto now = std::chrono::steady_clock::now();
      iter_ms[iteration - 1] = std::chrono::duration<double, std::milli>(now - t_iter).count();
      t_iter = now;
    } else if (iteration == 0) {
      t_iter = std::chrono::steady_clock::now();
    }
    if (iteration >= opt->max_num_iterations) break;
    if (step_was_successful && gmax() <= opt->gradient_tolerance) {
      s.termination_type = CALICO_RP_CONVERGENCE;
      std::snprintf(s.message, sizeof s.message, "Gradient tolerance reached.");
      break;
    }
    if (radius < opt->min_trust_region_radius) {
      s.termination_type = CALICO_RP_CONVERGENCE;
      std::snprintf(s.message, sizeof s.message, "Minimum trust region radius reached.");
      break;
    }
    ++iteration;
    step_was_successful = false;

    // LevenbergMarquardtStrategy::ComputeStep
    if (!reuse_diagonal) {
      column_sqnorm(P, J.data(), scale.data(), diag);
      for (double& d : diag) d = std::min(std::max(d, opt->min_lm_diagonal), opt->max_lm_diagonal);
    }
    std::vector<double> D2(nv);
    for (int k = 0; k < nv; ++k) D2[k] = diag[k] / radius;
    Normal Nq = build_normal(P, S, J.data(), r.data(), scale.data());
    bool valid = schur_solve(P, S, Nq, D2.data(), y.data(), nullptr, nullptr);
    reuse_diagonal = true;
    double model_cost_change = 0;
    if (valid) {
      for (int k = 0; k < nv; ++k) step[k] = -y[k];
      // model_cost_change = -(J_s step)^T (r + J_s step / 2)
      double acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
      for (int64_t i = 0; i < N; ++i) {
        Blk blk[4];
        const int nb = obs_blocks(P, P.cam[i], P.pat[i], P.tim[i], blk);
        for (int rr = 0; rr < R; ++rr) {
          double m = 0;
          for (int b = 0; b < nb; ++b)
            for (int k = 0; k < blk[b].size; ++k)
              m += J[(int64_t)W * R * i + W * rr + blk[b].col + k] * scale[blk[b].slot + k] * step[blk[b].slot + k];
          acc += m * (r[(int64_t)R * i + rr] + 0.5 * m);
        }
      }
      model_cost_change = -acc;
      if (!(model_cost_change > 0.0)) valid = false;
    }
    if (!valid) {
      if (++consecutive_invalid >= 5) {
        s.termination_type = CALICO_RP_FAILURE;
        std::snprintf(s.message, sizeof s.message, "Number of consecutive invalid steps more than 5.");
        ++s.num_unsuccessful_steps;
        break;
      }
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      ++s.num_unsuccessful_steps;
      continue;
    }
    consecutive_invalid = 0;
    double step_norm = 0;
    for (int k = 0; k < nv; ++k) {
      delta[k] = step[k] * scale[k];
      cand[k] = xc[k] + delta[k];
      step_norm += (xc[k] - cand[k]) * (xc[k] - cand[k]);
    }
    step_norm = std::sqrt(step_norm);
    double cand_cost = eval_all(P, cand.data(), nullptr, nullptr);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();

    if (at_least_one_successful &&
        step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
      s.termination_type = CALICO_RP_CONVERGENCE;
      std::snprintf(s.message, sizeof s.message, "Parameter tolerance reached.");
      ++s.num_unsuccessful_steps;  // the candidate is not taken
      break;
    }
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= opt->function_tolerance * cost) {
      s.termination_type = CALICO_RP_CONVERGENCE;
      std::snprintf(s.message, sizeof s.message, "Function tolerance reached.");
      ++s.num_unsuccessful_steps;
      break;
    }
    const double rho = cost_change / model_cost_change;
    if (rho > opt->min_relative_decrease) {
      xc = cand;
      x_norm = 0;
      for (double a : xc) x_norm += a * a;
      x_norm = std::sqrt(x_norm);
      cost = evaluate(xc);
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
      radius = std::min(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
      at_least_one_successful = true;
      step_was_successful = true;
      ++s.num_successful_steps;
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
      ++s.num_unsuccessful_steps;
    }
    if (opt->verbose)
      std::printf("oracle iter %3d cost %.12e change %.3e |g|max %.3e |step| %.3e rho %.3e radius %.3e\n",
                  iteration, cost, cost_change, gmax(), step_norm, rho, radius);
  }
  std::copy(xc.begin(), xc.begin() + 7 * (size_t)P.vn, x);
  if (P.refine()) std::copy(xc.begin() + 7 * (size_t)P.vn, xc.end(), P.intr.begin());  // optimised in place, like x
  s.final_cost = cost;
  s.num_iterations = s.num_successful_steps + s.num_unsuccessful_steps;
  s.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  s.stage.jacobian_sweep = jac_total_ms;
  s.stage.n_jacobian_sweep = n_jac;
  if (jac_ms) *jac_ms = n_jac ? jac_total_ms / n_jac : 0.0;
  if (sum) *sum = s;
  return 0;
}

int oracle_rp_solve(oracle_rp_problem* h, double* x, const calico_rp_options* opt,
                    calico_rp_summary* sum) {
  return oracle_rp_solve_timed(h, x, opt, sum, nullptr, nullptr);
}

}  // extern "C"
