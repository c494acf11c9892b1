// rae.hip — point reconstruction: calico_rae_reconstruct of
// include/calico_rp.h, replacing ReconstructXFunctionIDsMCwithDLTs
// (solving-structure.cpp:162-390).
//
// The reference solves the corners one after the other, each with its own ceres::Problem
// over the views that see it.  Here every corner advances together:
//   rae_sweep_kernel   one pass over all observations; a wave owns a chunk of views of ONE
//                      pattern and keeps that pattern's per-corner sums (10 doubles each:
//                      upper J^T J, J^T r, cost — or the DLT normal equations) in LDS.
//                      Corner indices are distinct inside a view, so the lanes of one view
//                      update different rows and no atomics are needed; views follow each
//                      other in program order.  Partial sums go out per chunk.
//   rae_update_kernel  one wave per corner: adds the chunk partials in a fixed order, then
//                      lane 0 runs that corner's trust-region step (Ceres' loop, as in
//                      rp_api.hip: Jacobi scaling, clamped LM diagonal, radius schedule,
//                      tolerance tests) and leaves the next candidate point.
// One sweep per LM iteration evaluates cost, gradient and J^T J at every corner's candidate,
// so an accepted step needs no second evaluation.  Data per sweep: 20 B per observation
// (u, v, corner), 192 B per view (A_hat, intrinsics): HBM-bound, SURVEY.md 8(d) style.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/calico_rp.h"
#include "rae_model.hpp"

namespace {

constexpr int NV = 10;       // sums per corner
constexpr int VIEW_D = 26;   // doubles per view: A_hat 12, intrinsics 12, cols, rows

enum { PH_DLT = 0, PH_ZERO = 1, PH_CAND = 2 };

struct PointState {
  double x[3], cand[3], H[6], g[3], cost, scale[3], diag[3];
  double radius, decrease_factor, step_norm, model_cost_change, x_norm;
  int phase, done, has_value, iteration, n_succ, n_unsucc, consecutive_invalid;
  int reuse_diagonal, at_least_one_successful, step_was_successful;
};

struct SweepArgs {
  int mode;  // 0: DLT normal equations, 1: LM sums
  int ncp, n_chunks;
  const int32_t* chunk_pattern;    // pattern of each chunk
  const int32_t* chunk_fr_begin;   // n_chunks + 1
  const int64_t* fr_obs_begin;     // sorted views
  const double* view;              // VIEW_D per sorted view
  const int32_t* obs_corner;       // corner inside the pattern
  const double* obs_u;
  const double* obs_v;
  const double* epos;              // 3 per corner: where to evaluate
  double* partial;                 // [chunk][NV][ncp]
};

// One wave per chunk.  LDS: sums [NV][ncp] (corner-contiguous: conflict-free for the usual
// consecutive corners), evaluation points [3][ncp].
__global__ __launch_bounds__(64) void rae_sweep_kernel(SweepArgs a) {
  extern __shared__ double lds[];
  const int ncp = a.ncp, lane = threadIdx.x, chunk = blockIdx.x;
  double* sT = lds;
  double* sX = lds + NV * ncp;
  const int pattern = a.chunk_pattern[chunk];
  for (int k = lane; k < NV * ncp; k += 64) sT[k] = 0.0;
  for (int k = lane; k < ncp; k += 64) {
    const double* e = a.epos + 3 * ((int64_t)pattern * ncp + k);
    sX[k] = e[0];
    sX[ncp + k] = e[1];
    sX[2 * ncp + k] = e[2];
  }
  __builtin_amdgcn_wave_barrier();
  // The first 64 observations of the next view are fetched while the current view is worked on.
  const int f_begin = a.chunk_fr_begin[chunk], f_end = a.chunk_fr_begin[chunk + 1];
  int64_t nb = f_begin < f_end ? a.fr_obs_begin[f_begin] : 0, ne = f_begin < f_end ? a.fr_obs_begin[f_begin + 1] : 0;
  int pc = 0;
  double pu = 0.0, pv = 0.0;
  if (nb + lane < ne) {
    pc = a.obs_corner[nb + lane];
    pu = a.obs_u[nb + lane];
    pv = a.obs_v[nb + lane];
  }
  for (int f = f_begin; f < f_end; ++f) {
    const double* vw = a.view + (int64_t)VIEW_D * f;  // same address in every lane
    double Ah[12], K[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      Ah[k] = vw[k];
      K[k] = vw[12 + k];
    }
    const double cols = vw[24], rows = vw[25];
    const int64_t b = nb, e = ne;
    int c = pc;
    double u = pu, v = pv;
    if (f + 1 < f_end) {
      nb = e;
      ne = a.fr_obs_begin[f + 2];
      if (nb + lane < ne) {
        pc = a.obs_corner[nb + lane];
        pu = a.obs_u[nb + lane];
        pv = a.obs_v[nb + lane];
      }
    }
    for (int64_t i = b + lane; i < e; i += 64) {
      if (i >= b + 64) {
        c = a.obs_corner[i];
        u = a.obs_u[i];
        v = a.obs_v[i];
      }
      double s[NV];
      if (a.mode == 0) {
        double r1[4], r2[4];
        rp::rae_dlt_rows(Ah, K, cols, rows, u, v, r1, r2);
        s[0] = r1[0] * r1[0] + r2[0] * r2[0];
        s[1] = r1[0] * r1[1] + r2[0] * r2[1];
        s[2] = r1[0] * r1[2] + r2[0] * r2[2];
        s[3] = r1[1] * r1[1] + r2[1] * r2[1];
        s[4] = r1[1] * r1[2] + r2[1] * r2[2];
        s[5] = r1[2] * r1[2] + r2[2] * r2[2];
        s[6] = -(r1[0] * r1[3] + r2[0] * r2[3]);  // LHS^T RHS, RHS = -column 3
        s[7] = -(r1[1] * r1[3] + r2[1] * r2[3]);
        s[8] = -(r1[2] * r1[3] + r2[2] * r2[3]);
        s[9] = 1.0;                               // views of this corner
      } else {
        const double X[3] = {sX[c], sX[ncp + c], sX[2 * ncp + c]};
        double r[2], J[6];
        rp::rae_eval(Ah, K, X, u, v, r, J);
        s[0] = J[0] * J[0] + J[3] * J[3];
        s[1] = J[0] * J[1] + J[3] * J[4];
        s[2] = J[0] * J[2] + J[3] * J[5];
        s[3] = J[1] * J[1] + J[4] * J[4];
        s[4] = J[1] * J[2] + J[4] * J[5];
        s[5] = J[2] * J[2] + J[5] * J[5];
        s[6] = J[0] * r[0] + J[3] * r[1];
        s[7] = J[1] * r[0] + J[4] * r[1];
        s[8] = J[2] * r[0] + J[5] * r[1];
        s[9] = r[0] * r[0] + r[1] * r[1];
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) sT[k * ncp + c] += s[k];
    }
    __builtin_amdgcn_wave_barrier();  // the next view may touch the same corners
  }
  double* out = a.partial + (int64_t)chunk * NV * ncp;
  for (int k = lane; k < NV * ncp; k += 64) out[k] = sT[k];
}

__device__ bool chol3(const double* M /*3x3 sym, row-major*/, const double* rhs, double* y) {
  const double l00s = M[0];
  if (!(l00s > 0.0)) return false;
  const double l00 = sqrt(l00s), l10 = M[3] / l00, l20 = M[6] / l00;
  const double l11s = M[4] - l10 * l10;
  if (!(l11s > 0.0)) return false;
  const double l11 = sqrt(l11s), l21 = (M[7] - l20 * l10) / l11;
  const double l22s = M[8] - l20 * l20 - l21 * l21;
  if (!(l22s > 0.0)) return false;
  const double l22 = sqrt(l22s);
  const double z0 = rhs[0] / l00, z1 = (rhs[1] - l10 * z0) / l11, z2 = (rhs[2] - l20 * z0 - l21 * z1) / l22;
  y[2] = z2 / l22;
  y[1] = (z1 - l21 * y[2]) / l11;
  y[0] = (z0 - l10 * y[1] - l20 * y[2]) / l00;
  return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}

__device__ void finish(PointState& s) { s.done = 1; }

// TrustRegionMinimizer's loop from FinalizeIterationAndCheckIfMinimizerCanContinue up to the
// point where the candidate must be evaluated; invalid steps shrink the radius and retry.
__device__ void next_step(PointState& s, const calico_rp_options& o) {
  for (;;) {
    if (s.iteration >= o.max_num_iterations) return finish(s);
    const double gmax = fmax(fabs(s.g[0]), fmax(fabs(s.g[1]), fabs(s.g[2])));
    if (s.step_was_successful && gmax <= o.gradient_tolerance) return finish(s);
    if (s.radius < o.min_trust_region_radius) return finish(s);
    ++s.iteration;
    s.step_was_successful = 0;
    const double Hd[3] = {s.H[0], s.H[3], s.H[5]};
    if (!s.reuse_diagonal)
      for (int a = 0; a < 3; ++a)
        s.diag[a] = fmin(fmax(Hd[a] * s.scale[a] * s.scale[a], o.min_lm_diagonal), o.max_lm_diagonal);
    const double Hf[9] = {s.H[0], s.H[1], s.H[2], s.H[1], s.H[3], s.H[4], s.H[2], s.H[4], s.H[5]};
    double M[9], Md[9], gs[3], y[3];
    for (int a = 0; a < 3; ++a) {
      gs[a] = s.g[a] * s.scale[a];
      for (int b = 0; b < 3; ++b) Md[3 * a + b] = M[3 * a + b] = Hf[3 * a + b] * s.scale[a] * s.scale[b];
    }
    for (int a = 0; a < 3; ++a) Md[4 * a] += s.diag[a] / s.radius;
    bool valid = chol3(Md, gs, y);
    s.reuse_diagonal = 1;
    double step[3] = {0, 0, 0};
    if (valid) {
      double q = 0;
      for (int a = 0; a < 3; ++a) step[a] = -y[a];
      for (int a = 0; a < 3; ++a) {
        double Ms = 0;
        for (int b = 0; b < 3; ++b) Ms += M[3 * a + b] * step[b];
        q += step[a] * (gs[a] + 0.5 * Ms);
      }
      s.model_cost_change = -q;
      if (!(s.model_cost_change > 0.0)) valid = false;
    }
    if (!valid) {
      ++s.n_unsucc;
      if (++s.consecutive_invalid >= 5) return finish(s);
      s.radius /= s.decrease_factor;
      s.decrease_factor *= 2.0;
      continue;
    }
    s.consecutive_invalid = 0;
    double sn = 0;
    for (int a = 0; a < 3; ++a) {
      s.cand[a] = s.x[a] + step[a] * s.scale[a];
      sn += (s.x[a] - s.cand[a]) * (s.x[a] - s.cand[a]);
    }
    s.step_norm = sqrt(sn);
    s.phase = PH_CAND;
    return;
  }
}

struct UpdateArgs {
  int n_points, ncp;
  const int32_t* pattern_chunk_begin;  // n_patterns + 1
  const double* partial;
  PointState* state;
  double* epos;
  int* n_active;
  calico_rp_options opt;
};

__global__ __launch_bounds__(64) void rae_update_kernel(UpdateArgs a) {
  const int point = blockIdx.x, lane = threadIdx.x;
  PointState& s = a.state[point];
  if (s.done) return;  // uniform
  const int pattern = point / a.ncp, c = point % a.ncp;
  double sum[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) sum[k] = 0.0;
  for (int ch = a.pattern_chunk_begin[pattern] + lane; ch < a.pattern_chunk_begin[pattern + 1]; ch += 64) {
    const double* p = a.partial + (int64_t)ch * NV * a.ncp + c;
#pragma unroll
    for (int k = 0; k < NV; ++k) sum[k] += p[k * a.ncp];
  }
#pragma unroll
  for (int k = 0; k < NV; ++k)
    for (int off = 32; off > 0; off >>= 1) sum[k] += __shfl_down(sum[k], off, 64);
  if (lane != 0) return;
  const calico_rp_options& o = a.opt;
  if (s.phase == PH_DLT) {
    if (sum[9] < 2.0) {  // fewer than two views: left at zero, has_values false (cpp:309)
      s.done = 1;
      return;
    }
    const double M[9] = {sum[0], sum[1], sum[2], sum[1], sum[3], sum[4], sum[2], sum[4], sum[5]};
    double y[3];
    if (!chol3(M, sum + 6, y)) {
      s.done = 1;
      return;
    }
    s.has_value = 1;
    for (int k = 0; k < 3; ++k) s.x[k] = s.cand[k] = a.epos[3 * point + k] = y[k];
    s.phase = PH_ZERO;
    if (o.max_num_iterations <= 0) s.done = 1;
    else atomicAdd(a.n_active, 1);
    return;
  }
  const double cost = 0.5 * sum[9];
  if (s.phase == PH_ZERO) {  // TrustRegionMinimizer::IterationZero
    for (int k = 0; k < 6; ++k) s.H[k] = sum[k];
    for (int k = 0; k < 3; ++k) s.g[k] = sum[6 + k];
    s.cost = cost;
    const double Hd[3] = {s.H[0], s.H[3], s.H[5]};
    for (int k = 0; k < 3; ++k) s.scale[k] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(Hd[k])) : 1.0;
    s.x_norm = sqrt(s.x[0] * s.x[0] + s.x[1] * s.x[1] + s.x[2] * s.x[2]);
    s.radius = o.initial_trust_region_radius;
    s.decrease_factor = 2.0;
    s.n_succ = 1;
    s.step_was_successful = 1;
  } else {  // the candidate has been evaluated
    double cand_cost = isfinite(cost) ? cost : 1.7976931348623157e308;
    if (s.at_least_one_successful && s.step_norm <= o.parameter_tolerance * (s.x_norm + o.parameter_tolerance)) {
      ++s.n_unsucc;
      s.done = 1;
    } else {
      const double cost_change = s.cost - cand_cost;
      if (fabs(cost_change) <= o.function_tolerance * s.cost) {
        ++s.n_unsucc;
        s.done = 1;
      } else {
        const double rho = cost_change / s.model_cost_change;
        if (rho > o.min_relative_decrease) {
          for (int k = 0; k < 3; ++k) s.x[k] = s.cand[k];
          s.x_norm = sqrt(s.x[0] * s.x[0] + s.x[1] * s.x[1] + s.x[2] * s.x[2]);
          s.cost = cand_cost;
          for (int k = 0; k < 6; ++k) s.H[k] = sum[k];
          for (int k = 0; k < 3; ++k) s.g[k] = sum[6 + k];
          const double t = 2.0 * rho - 1.0;
          s.radius = fmin(o.max_trust_region_radius, s.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
          s.decrease_factor = 2.0;
          s.reuse_diagonal = 0;
          s.at_least_one_successful = 1;
          s.step_was_successful = 1;
          ++s.n_succ;
        } else {
          s.radius /= s.decrease_factor;
          s.decrease_factor *= 2.0;
          ++s.n_unsucc;
        }
      }
    }
  }
  if (!s.done) next_step(s, o);
  if (s.done) {
    atomicAdd(a.n_active, -1);
    for (int k = 0; k < 3; ++k) a.epos[3 * point + k] = s.x[k];
  } else {
    for (int k = 0; k < 3; ++k) a.epos[3 * point + k] = s.cand[k];
  }
}

struct Fail {
  int code;
  std::string msg;
};
#define RAE_HIP(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess) throw Fail{CALICO_RP_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)}; \
  } while (0)

// host loops over the observations on up to 16 threads (15 M observations of config C: the serial
// checks and copies took most of the call's 140 ms)
int rae_threads(int64_t n) {
  if (n < (int64_t)1 << 18) return 1;
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(16, std::max<unsigned>(1, hw));
}
template <typename Fn>
void rae_parallel(int64_t n, int nt, Fn&& fn) {
  if (nt <= 1 || n <= 0) {
    fn((int64_t)0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  const int64_t per = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t b = std::min(n, per * t), e = std::min(n, per * (t + 1));
    th.emplace_back([&fn, b, e, t] { fn(b, e, t); });
  }
  for (auto& x : th) x.join();
}

template <typename T>
struct Dev {
  T* p = nullptr;
  ~Dev() {
    if (p) (void)hipFree(p);
  }
  void put(const std::vector<T>& h, hipStream_t s) {
    alloc(h.size());
    if (!h.empty()) RAE_HIP(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void put(const T* h, size_t n, hipStream_t s) {
    alloc(n);
    if (n) RAE_HIP(hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void alloc(size_t n) {
    if (n == 0) n = 1;
    RAE_HIP(hipMalloc((void**)&p, n * sizeof(T)));
  }
};

// view[s] = [A_hat of the view | its camera's 12 parameters | image columns, rows] for the sorted
// views of a handle: A_hat and the camera parameters change from call to call, the rest is the handle's
__global__ __launch_bounds__(256) void rae_build_views_kernel(int F, const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ fr_cam,
                                                              const double* __restrict__ a_hat,
                                                              const double* __restrict__ cam_params,
                                                              const double* __restrict__ cols_rows,
                                                              double* __restrict__ view) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)F * VIEW_D) return;
  const int s = (int)(t / VIEW_D), k = (int)(t % VIEW_D);
  const int f = order[s], c = fr_cam[f];
  view[t] = k < 12 ? a_hat[12 * (int64_t)f + k] : k < 24 ? cam_params[12 * (int64_t)c + (k - 12)] : cols_rows[2 * c + (k - 24)];
}

template <typename T>
struct DevArr {  // owned device array of a handle
  T* p = nullptr;
  size_t n = 0;
  DevArr() = default;
  DevArr(const DevArr&) = delete;
  DevArr& operator=(const DevArr&) = delete;
  ~DevArr() {
    if (p) (void)hipFree(p);
  }
  void alloc(size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = count;
    RAE_HIP(hipMalloc((void**)&p, std::max<size_t>(1, count) * sizeof(T)));
  }
  void put(const T* h, size_t count, hipStream_t s) {
    alloc(count);
    if (count) RAE_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void put(const std::vector<T>& h, hipStream_t s) { put(h.data(), h.size(), s); }
};

}  // namespace

// The handle: everything about the views and their observations that does not change between calls
// (regrouped by pattern, chunked, on the device), so that a call with new poses / camera
// parameters uploads 96 bytes per view and runs the sweeps.
struct calico_rae_problem {
  int device = 0, cn = 0, n_frs = 0, n_points = 0, ncp = 0, F = 0, n_chunks = 0;
  int64_t n_obs = 0;
  size_t lds_bytes = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  std::string err;
  DevArr<int32_t> d_chunk_pattern, d_chunk_fr_begin, d_pattern_chunk_begin, d_corner, d_order, d_fr_cam;
  DevArr<int64_t> d_sbeg;
  DevArr<double> d_view, d_u, d_v, d_epos, d_partial, d_a_hat, d_cam, d_cols_rows;
  DevArr<PointState> d_state;
  DevArr<int> d_active;
};

extern "C" const char* calico_rae_last_error(const calico_rae_problem* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int calico_rae_destroy(calico_rae_problem* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  hipStream_t s = h->stream;
  delete h;
  if (s) (void)hipStreamDestroy(s);
  return 0;
}

extern "C" int calico_rae_create(int32_t device, int32_t cn, const int32_t* image_cols_rows, int32_t n_frs,
                                 const int32_t* fr_cam, int32_t n_points, int32_t ncp, int64_t n_obs, const int32_t* obs_fr,
                                 const int32_t* obs_point, const double* u, const double* v, calico_rae_problem** out,
                                 char* err, size_t err_len) {
  auto fail = [&](int code, const std::string& m) {
    if (err && err_len) std::snprintf(err, err_len, "%s", m.c_str());
    return code;
  };
  if (!out) return fail(CALICO_RP_EINVAL, "rae_create: null output");
  *out = nullptr;
  if (cn <= 0 || !image_cols_rows || n_frs < 0 || n_points <= 0 || ncp <= 0 || n_points % ncp != 0 || n_obs < 0)
    return fail(CALICO_RP_EINVAL, "rae_reconstruct: bad argument");
  if (n_obs > 0 && (!fr_cam || !obs_fr || !obs_point || !u || !v)) return fail(CALICO_RP_EINVAL, "rae_reconstruct: null array");
  const size_t lds_bytes = (size_t)(NV + 3) * ncp * sizeof(double);
  if (lds_bytes > 64 * 1024) return fail(CALICO_RP_EINVAL, "rae_reconstruct: more than 630 corners per pattern");
  const int n_patterns = n_points / ncp;
  calico_rae_problem* h = nullptr;
  int rc = 0;
  try {
    // ---- views with observations, grouped by pattern (stable), chunked
    std::vector<int64_t> fbeg((size_t)n_frs + 1, -1);
    std::vector<int32_t> fpat(n_frs, -1);
    for (int f = 0; f < n_frs; ++f)
      if (fr_cam[f] < 0 || fr_cam[f] >= cn) return fail(CALICO_RP_EINVAL, "rae_reconstruct: camera index out of range");
    for (int c = 0; c < cn; ++c)
      if (!(image_cols_rows[2 * c] > 0) || !(image_cols_rows[2 * c + 1] > 0))
        return fail(CALICO_RP_EINVAL, "rae_reconstruct: image size must be positive");
    const int nt = rae_threads(n_obs);
    {
      // every observation is checked against its predecessor; the first one of a view records
      // where the view starts (each view's start is written by exactly one thread)
      enum { OK = 0, BAD_FR, BAD_CORNER, BAD_ORDER, BAD_PIXEL };
      std::vector<int> what(nt, OK);
      std::vector<int64_t> where(nt, -1);
      rae_parallel(n_obs, nt, [&](int64_t b, int64_t e, int t) {
        for (int64_t i = b; i < e; ++i) {
          const int f = obs_fr[i], k = obs_point[i];
          int bad = OK;
          if (f < 0 || f >= n_frs || (i > 0 && f < obs_fr[i - 1])) bad = BAD_FR;
          else if (k < 0 || k >= n_points) bad = BAD_CORNER;
          else if (i > 0 && obs_fr[i - 1] == f && (k <= obs_point[i - 1] || k / ncp != obs_point[i - 1] / ncp)) bad = BAD_ORDER;
          else if (!std::isfinite(u[i]) || !std::isfinite(v[i])) bad = BAD_PIXEL;
          if (bad != OK) {
            what[t] = bad;
            where[t] = i;
            return;
          }
          if (i == 0 || obs_fr[i - 1] != f) {
            fbeg[f] = i;
            fpat[f] = k / ncp;
          }
        }
      });
      for (int t = 0; t < nt; ++t) {  // the first offender in observation order
        if (what[t] == OK) continue;
        const std::string at = " (observation " + std::to_string(where[t]) + ")";
        if (what[t] == BAD_FR) return fail(CALICO_RP_EINVAL, "rae_reconstruct: obs_fr must be non-decreasing and < n_frs" + at);
        if (what[t] == BAD_CORNER) return fail(CALICO_RP_EINVAL, "rae_reconstruct: corner index out of range" + at);
        if (what[t] == BAD_ORDER)
          return fail(CALICO_RP_EINVAL, "rae_reconstruct: corners of a view must increase within one pattern" + at);
        return fail(CALICO_RP_EINVAL, "rae_reconstruct: non-finite pixel" + at);
      }
      fbeg[n_frs] = n_obs;
      for (int f = n_frs - 1; f >= 0; --f)  // views without observations start where the next one does
        if (fbeg[f] < 0) fbeg[f] = fbeg[f + 1];
    }
    // views with observations, grouped by pattern (stable counting sort)
    std::vector<int32_t> order;
    {
      std::vector<int32_t> cnt(n_patterns + 1, 0);
      for (int f = 0; f < n_frs; ++f)
        if (fpat[f] >= 0) ++cnt[fpat[f] + 1];
      for (int p = 0; p < n_patterns; ++p) cnt[p + 1] += cnt[p];
      order.resize(cnt[n_patterns]);
      for (int f = 0; f < n_frs; ++f)
        if (fpat[f] >= 0) order[cnt[fpat[f]]++] = f;
    }
    const int F = (int)order.size();
    std::vector<int64_t> sbeg((size_t)F + 1, 0);
    // regrouped payload, written once by the threads below: no zero-fill pass first
    std::unique_ptr<int32_t[]> corner(new int32_t[(size_t)std::max<int64_t>(1, n_obs)]);
    std::unique_ptr<double[]> su(new double[(size_t)std::max<int64_t>(1, n_obs)]), sv(new double[(size_t)std::max<int64_t>(1, n_obs)]);
    const int per_chunk = std::max(1, (F + 16383) / 16384);
    std::vector<int32_t> chunk_pattern, chunk_fr_begin, pattern_chunk_begin(n_patterns + 1, 0);
    {
      int prev_pat = -1, in_chunk = 0;
      for (int s = 0; s < F; ++s) {
        const int f = order[s];
        if (fpat[f] != prev_pat || in_chunk == per_chunk) {
          chunk_pattern.push_back(fpat[f]);
          chunk_fr_begin.push_back(s);
          in_chunk = 0;
        }
        prev_pat = fpat[f];
        ++in_chunk;
        sbeg[s + 1] = sbeg[s] + (fbeg[f + 1] - fbeg[f]);
      }
      chunk_fr_begin.push_back(F);
      rae_parallel(F, nt, [&](int64_t sb, int64_t se, int) {
        for (int64_t s = sb; s < se; ++s) {
          const int f = order[s];
          const int64_t n = fbeg[f + 1] - fbeg[f], src = fbeg[f], dst = sbeg[s];
          const int base = fpat[f] * ncp;
          for (int64_t i = 0; i < n; ++i) corner[dst + i] = obs_point[src + i] - base;
          std::memcpy(&su[dst], u + src, n * sizeof(double));
          std::memcpy(&sv[dst], v + src, n * sizeof(double));
        }
      });
      const int n_chunks = (int)chunk_pattern.size();
      int ch = 0;
      for (int p = 0; p < n_patterns; ++p) {
        pattern_chunk_begin[p] = ch;
        while (ch < n_chunks && chunk_pattern[ch] == p) ++ch;
      }
      pattern_chunk_begin[n_patterns] = ch;
    }
    std::vector<double> cols_rows(2 * (size_t)cn);
    for (int c = 0; c < 2 * cn; ++c) cols_rows[c] = (double)(float)image_cols_rows[c];  // 2.0 / float(cols), cpp:326

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
      throw Fail{CALICO_RP_EDEVICE, "rae_reconstruct: no such HIP device"};
    RAE_HIP(hipSetDevice(device));
    h = new calico_rae_problem;
    h->device = device; h->cn = cn; h->n_frs = n_frs; h->n_points = n_points; h->ncp = ncp; h->F = F;
    h->n_chunks = (int)chunk_pattern.size(); h->n_obs = n_obs; h->lds_bytes = lds_bytes;
    RAE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto& e : h->ev) RAE_HIP(hipEventCreate(&e));
    hipStream_t s = h->stream;
    h->d_chunk_pattern.put(chunk_pattern, s);
    h->d_chunk_fr_begin.put(chunk_fr_begin, s);
    h->d_pattern_chunk_begin.put(pattern_chunk_begin, s);
    h->d_corner.put(corner.get(), (size_t)n_obs, s);
    h->d_sbeg.put(sbeg, s);
    h->d_u.put(su.get(), (size_t)n_obs, s);
    h->d_v.put(sv.get(), (size_t)n_obs, s);
    h->d_order.put(order, s);
    h->d_fr_cam.put(fr_cam, (size_t)n_frs, s);
    h->d_cols_rows.put(cols_rows, s);
    h->d_view.alloc((size_t)VIEW_D * std::max(1, F));
    h->d_a_hat.alloc((size_t)12 * std::max(1, n_frs));
    h->d_cam.alloc((size_t)12 * cn);
    h->d_epos.alloc((size_t)3 * n_points);
    h->d_partial.alloc((size_t)std::max(1, h->n_chunks) * NV * ncp);
    h->d_state.alloc(n_points);
    h->d_active.alloc(1);
    RAE_HIP(hipStreamSynchronize(s));  // the host staging buffers go out of scope
  } catch (const Fail& f) {
    rc = fail(f.code, f.msg);
  } catch (const std::bad_alloc&) {
    rc = fail(CALICO_RP_ENOMEM, "rae_reconstruct: host allocation failed");
  }
  if (rc) {
    calico_rae_destroy(h);
    return rc;
  }
  *out = h;
  return 0;
}

extern "C" int calico_rae_solve(calico_rae_problem* h, const double* cam_params, const double* fr_a_hat,
                                const calico_rp_options* opt_in, double* xyz, uint8_t* has_values, int32_t* iterations,
                                double* final_cost, calico_rae_summary* summary) {
  if (!h) return CALICO_RP_EINVAL;
  auto fail = [&](int code, const std::string& m) {
    h->err = m;
    return code;
  };
  if (!cam_params || !xyz || !has_values || (h->n_frs > 0 && !fr_a_hat)) return fail(CALICO_RP_EINVAL, "rae_solve: bad argument");
  calico_rp_options o;
  if (opt_in) {
    o = *opt_in;
  } else {
    calico_rp_options_default(&o);
    o.max_num_iterations = 50;  // Solver::Options default (solving-structure.cpp:358)
  }
  const auto t_begin = std::chrono::steady_clock::now();
  const int n_points = h->n_points, ncp = h->ncp, n_chunks = h->n_chunks;
  {
    const int nt = rae_threads((int64_t)12 * h->n_frs);
    std::vector<int> bad(nt, 0);
    rae_parallel((int64_t)12 * h->n_frs, nt, [&](int64_t b, int64_t e, int t) {
      for (int64_t i = b; i < e; ++i)
        if (!std::isfinite(fr_a_hat[i])) bad[t] = 1;
    });
    for (int t = 0; t < nt; ++t)
      if (bad[t]) return fail(CALICO_RP_EINVAL, "rae_reconstruct: non-finite A_hat");
  }
  int rc = 0;
  try {
    RAE_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    hipEvent_t* ev = h->ev;
    if (h->n_frs) RAE_HIP(hipMemcpyAsync(h->d_a_hat.p, fr_a_hat, (size_t)12 * h->n_frs * sizeof(double), hipMemcpyHostToDevice, s));
    RAE_HIP(hipMemcpyAsync(h->d_cam.p, cam_params, (size_t)12 * h->cn * sizeof(double), hipMemcpyHostToDevice, s));
    if (h->F > 0)
      hipLaunchKernelGGL(rae_build_views_kernel, dim3((unsigned)(((int64_t)h->F * VIEW_D + 255) / 256)), dim3(256), 0, s, h->F,
                         h->d_order.p, h->d_fr_cam.p, h->d_a_hat.p, h->d_cam.p, h->d_cols_rows.p, h->d_view.p);
    RAE_HIP(hipMemsetAsync(h->d_epos.p, 0, (size_t)3 * n_points * sizeof(double), s));
    RAE_HIP(hipMemsetAsync(h->d_state.p, 0, (size_t)n_points * sizeof(PointState), s));
    RAE_HIP(hipMemsetAsync(h->d_active.p, 0, sizeof(int), s));

    SweepArgs sa;
    sa.ncp = ncp; sa.n_chunks = n_chunks; sa.chunk_pattern = h->d_chunk_pattern.p; sa.chunk_fr_begin = h->d_chunk_fr_begin.p;
    sa.fr_obs_begin = h->d_sbeg.p; sa.view = h->d_view.p; sa.obs_corner = h->d_corner.p; sa.obs_u = h->d_u.p; sa.obs_v = h->d_v.p;
    sa.epos = h->d_epos.p; sa.partial = h->d_partial.p;
    UpdateArgs ua;
    ua.n_points = n_points; ua.ncp = ncp; ua.pattern_chunk_begin = h->d_pattern_chunk_begin.p; ua.partial = h->d_partial.p;
    ua.state = h->d_state.p; ua.epos = h->d_epos.p; ua.n_active = h->d_active.p; ua.opt = o;

    int sweeps = 0;
    double sweep_ms = 0, update_ms = 0;
    int n_lm_sweeps = 0;
    for (int it = 0;; ++it) {
      sa.mode = it == 0 ? 0 : 1;
      RAE_HIP(hipEventRecord(ev[0], s));
      if (n_chunks > 0) hipLaunchKernelGGL(rae_sweep_kernel, dim3(n_chunks), dim3(64), h->lds_bytes, s, sa);
      RAE_HIP(hipEventRecord(ev[1], s));
      hipLaunchKernelGGL(rae_update_kernel, dim3(n_points), dim3(64), 0, s, ua);
      RAE_HIP(hipEventRecord(ev[2], s));
      int active = 0;
      RAE_HIP(hipMemcpyAsync(&active, h->d_active.p, sizeof(int), hipMemcpyDeviceToHost, s));
      RAE_HIP(hipStreamSynchronize(s));
      RAE_HIP(hipGetLastError());
      ++sweeps;
      float ms = 0;
      if (it > 0) {
        RAE_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        sweep_ms += ms;
        ++n_lm_sweeps;
      }
      RAE_HIP(hipEventElapsedTime(&ms, ev[1], ev[2]));
      update_ms += ms;
      if (active <= 0) break;
      if (it > o.max_num_iterations + 8) throw Fail{CALICO_RP_ENUMERIC, "rae_reconstruct: corners still active after the iteration limit"};
    }
    std::vector<PointState> st(n_points);
    RAE_HIP(hipMemcpyAsync(st.data(), h->d_state.p, (size_t)n_points * sizeof(PointState), hipMemcpyDeviceToHost, s));
    RAE_HIP(hipStreamSynchronize(s));
    int n_solved = 0, max_it = 0;
    for (int k = 0; k < n_points; ++k) {
      has_values[k] = st[k].has_value ? 1 : 0;
      for (int a = 0; a < 3; ++a) xyz[3 * k + a] = st[k].has_value ? st[k].x[a] : 0.0;
      const int its = st[k].n_succ + st[k].n_unsucc;
      if (iterations) iterations[k] = its;
      if (final_cost) final_cost[k] = st[k].cost;
      n_solved += st[k].has_value;
      max_it = std::max(max_it, its);
    }
    if (summary) {
      summary->n_points = n_points;
      summary->n_solved = n_solved;
      summary->sweeps = sweeps;
      summary->max_point_iterations = max_it;
      summary->sweep_ms = n_lm_sweeps ? sweep_ms / n_lm_sweeps : 0.0;
      summary->update_ms = update_ms / sweeps;
      summary->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
  } catch (const Fail& f) {
    rc = fail(f.code, f.msg);
  } catch (const std::bad_alloc&) {
    rc = fail(CALICO_RP_ENOMEM, "rae_reconstruct: host allocation failed");
  }
  return rc;
}

// The stateless form: create + solve + destroy.
extern "C" int calico_rae_reconstruct(int32_t device, int32_t cn, const double* cam_params, const int32_t* image_cols_rows,
                                      int32_t n_frs, const int32_t* fr_cam, const double* fr_a_hat, int32_t n_points,
                                      int32_t ncp, int64_t n_obs, const int32_t* obs_fr, const int32_t* obs_point,
                                      const double* u, const double* v, const calico_rp_options* opt_in, double* xyz,
                                      uint8_t* has_values, int32_t* iterations, double* final_cost,
                                      calico_rae_summary* summary, char* err, size_t err_len) {
  auto fail = [&](int code, const std::string& m) {
    if (err && err_len) std::snprintf(err, err_len, "%s", m.c_str());
    return code;
  };
  if (!cam_params || !xyz || !has_values) return fail(CALICO_RP_EINVAL, "rae_reconstruct: bad argument");
  if (n_obs > 0 && !fr_a_hat) return fail(CALICO_RP_EINVAL, "rae_reconstruct: null array");
  const auto t_begin = std::chrono::steady_clock::now();
  calico_rae_problem* h = nullptr;
  int rc = calico_rae_create(device, cn, image_cols_rows, n_frs, fr_cam, n_points, ncp, n_obs, obs_fr, obs_point, u, v, &h, err,
                             err_len);
  if (rc) return rc;
  rc = calico_rae_solve(h, cam_params, fr_a_hat, opt_in, xyz, has_values, iterations, final_cost, summary);
  if (rc) fail(rc, h->err);
  if (summary && !rc)
    summary->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  calico_rae_destroy(h);
  return rc;
}
