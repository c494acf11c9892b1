// rp_api.hip — host side of libcalico_rp.so: the C ABI of include/calico_rp.h,
// problem structure (FR grouping, time segments, Schur bookkeeping), the
// Levenberg–Marquardt loop and the RCCL exchange.
//
// Reference seams replaced (see include/calico_rp.h for the per-entry citations):
//   CeresProblemClass::AddEqsToProblemReprojectionError  solving-structure.cpp:463-669
//   ceres::Solve(options{DENSE_SCHUR, 10 iterations})    solving-structure.cpp:973-982
// The minimiser semantics (trust-region LM, Jacobi scaling, LM diagonal clamping,
// radius schedule, termination tests) are those of Ceres 2.x' TrustRegionMinimizer
// and LevenbergMarquardtStrategy, which the reference relies on through defaults;
// Ceres is not in the reference tree, so those are restated here
// and in oracle/rp_oracle.cpp independently of each other's code.
//
// Differences from Ceres that do not change the iterates beyond rounding:
//   * the Jacobian after an accepted step is evaluated lazily at the start of the
//     next iteration (and skipped when the iteration limit ends the solve), or, with
//     speculative evaluation (calico_rp_set_speculation, default in fused mode), already
//     at the candidate: the fused kernel there gives the candidate's cost and, if the step
//     is accepted, the records the next iteration starts from (lm_step_impl);
//   * model_cost_change is computed from the solved system,
//       1/2 y^T g_s + 1/2 y^T D^2 y   (== -(J_s d)^T (r + J_s d / 2) for exact solves),
//     instead of a second sweep over J;
//   * time blocks are always the eliminated set (Ceres picks an independent set of
//     the Hessian graph, which for this problem is the time blocks,
//     solving-structure.cpp:546-548: every residual touches exactly one).
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/calico_rp.h"
#include "rp_internal.hpp"

namespace {

using rp::REC_A;
using rp::REC_B;

struct HipError {
  std::string msg;
};
#define HIP_CHECK(expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      throw HipError{std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                     std::to_string(__LINE__) + ")"};                                        \
  } while (0)

// Host-side loops over the observations (10^8 on config D) run on a few threads:
// fn(begin, end, thread_index) over [0, n) cut into equal chunks.
int host_threads(int64_t n) {
  if (n < (int64_t)1 << 20) return 1;
  unsigned hw = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(16, std::max<unsigned>(1, hw));
}
template <typename Fn>
void parallel_chunks(int64_t n, int nt, Fn&& fn) {
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
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count == n && p) return;
    release();
    if (count == 0) return;
    HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void zero(hipStream_t s) {
    if (p) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  void upload(const std::vector<T>& h, hipStream_t s) {
    alloc(h.size());
    if (!h.empty()) HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
  }
  // [first `keep` elements that are already here | `count` elements from host memory]; the
  // caller synchronises the stream before `src` goes away
  void append_from_host(size_t keep, const T* src, size_t count, hipStream_t s) {
    DevBuf bigger;
    bigger.alloc(keep + count);
    if (keep) HIP_CHECK(hipMemcpyAsync(bigger.p, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (count) HIP_CHECK(hipMemcpyAsync(bigger.p + keep, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    swap(bigger);
  }
};

// ------------------------------------------------------------------ RCCL (loaded lazily)
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& err) {
    if (lib) return true;
#ifdef CALICO_RP_TEST_HOOKS
    // Test build only (tests/_build/libcalico_rp_testhooks.so, never the product library):
    // CALICO_RP_COLLECTIVES_LIB names another library with the same six entry points, so that
    // the tests can run two ranks on ONE GPU (RCCL refuses two ranks per device) through a
    // shared-memory stand-in, tests/helpers/shm_collectives.cpp.
    if (const char* other = getenv("CALICO_RP_COLLECTIVES_LIB")) {
      lib = dlopen(other, RTLD_NOW | RTLD_LOCAL);
      if (!lib) {
        err = std::string("cannot load CALICO_RP_COLLECTIVES_LIB: ") + dlerror();
        return false;
      }
    }
#endif
    // the soname torch already mapped (if any) is reused by the loader
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (lib) break;
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!lib) {
      err = std::string("cannot load librccl: ") + dlerror();
      return false;
    }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !AllGather || !GetErrorString) {
      err = "librccl lacks a required symbol";
      return false;
    }
    return true;
  }
};
Rccl g_rccl;

// ------------------------------------------------------------------ scratch of the entry point that allocates nothing
// calico_rp_fr_errors_noalloc (include/calico_rp.h) keeps everything calico_rp_fr_errors
// allocates per call (the device copy of x, one mean per FR, the sum) in this file-scope
// array: [x: 7 * NOALLOC_MAX_VARS | per-FR means: NOALLOC_MAX_FRS | sum: 1 (+ padding)].
// Every GPU the library runs on holds its own copy (18.7 MB) until the process ends.  In host
// code the array's name is only a stand-in: the device address is looked up once per device
// (scratch_setup, called from calico_rp_create with that device current) and that pointer is
// what the kernels, copies and memsets get.
constexpr int NOALLOC_MAX_VARS = CALICO_RP_NOALLOC_MAX_VARS;
constexpr int NOALLOC_MAX_FRS = CALICO_RP_NOALLOC_MAX_FRS;
constexpr size_t NOALLOC_X = 0, NOALLOC_OUT = (size_t)7 * NOALLOC_MAX_VARS, NOALLOC_SUM = NOALLOC_OUT + NOALLOC_MAX_FRS;
constexpr size_t NOALLOC_DOUBLES = NOALLOC_SUM + 32;
static __device__ __align__(256) double scratch[NOALLOC_DOUBLES];
constexpr int NOALLOC_MAX_DEVICES = 64;
double* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};  // per device: each GPU has the array at its own address
std::mutex g_scratch_mutex;

// `device` must be the current device.  hipSuccess, or the error of the symbol lookup.
hipError_t scratch_setup(int device) {
  if (device < 0 || device >= NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return hipSuccess;
  double* ptr = nullptr;
  const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return e;
  if (!ptr) return hipErrorInvalidSymbol;
  g_scratch_ptr[device] = ptr;
  return hipSuccess;
}

// ------------------------------------------------------------------ structure
// A maximal run of consecutive observations of one (camera, pattern, time): where its payload
// sits in the device arrays (begin) and which caller's index its first observation has (orig).
struct ObsRun {
  int32_t t, c, p;  // t: variable index, INT32_MAX for t* (sorts last)
  int64_t begin, len, orig;
};

struct Structure {
  int64_t N = 0;
  int F = 0;
  bool identity_perm = true;  // sorted position == caller's index for every observation
  std::vector<ObsRun> run_map;  // sorted runs: begin = sorted position, orig = caller's index (eval hooks)
  // payload moves the device still has to make (run k: move_len[k] observations from position
  // move_src[k] to move_dst[k]); empty when the arrival order is the sorted order
  std::vector<int64_t> move_src, move_dst, move_len;
  std::vector<int32_t> fr_cam, fr_pat, fr_tim;
  std::vector<int64_t> fr_obs_begin;
  int n_rblocks = 0, n = 0, ldn = 128;
  std::vector<int32_t> rvar, ridx;
  int n_seg = 0, n_krows = 16;
  std::vector<int32_t> seg_var, seg_fr_begin;
  // pieces (runs of FRs sharing (time, pattern) inside one accumulate wave)
  int n_parts = 0, n_pieces = 0, wave_obs = 4096;
  std::vector<int32_t> wave_part0, part_fr, part_piece, seg_part_begin, seg_piece_begin, pp_piece, pp_chunk_begin,
      pp_rblock_chunk_begin;
  std::vector<int64_t> part_obs_begin;  // first observation of each part (+ N): parts tile the sorted observations
  // pair groups
  std::vector<int32_t> pair_part, chunk_begin, group_chunk_begin, group_rc, group_rp;
  std::vector<int32_t> rblock_group_begin, rblock_groups;
  // units
  std::vector<int32_t> unit_seg, unit_rblock, unit_begin, contrib, seg_unit_begin;
  int ksplit = 1, n_pairs = 1;
  int ksplit_w = 1, n_pairs_w = 1;  // the same for the six-column layout of the fused mode (ldu)
  // camera parameters refined: nk 12-parameter blocks (one per reduced camera block, which come
  // first in rvar) follow the 7-parameter blocks: n = 7 n_rblocks + 12 nk
  int nk = 0;
  int n7() const { return 7 * n_rblocks; }
  int nu() const { return 6 * n_rblocks + 12 * nk; }
  int ldu() const { return (nu() + 1 + 127) / 128 * 128; }
};

enum { SC_COSTSUM = 0, SC_GMAX = 1, SC_CAND = 2, SC_BACK = 3 /*3..8*/, SC_FAIL = 9, SC_N = 16 };

struct Lm {
  calico_rp_options opt;
  bool active = false, done = false;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diag = false, jac_valid = false, first_eval = true;
  // speculative evaluation (see lm_step_impl): spec_ok is cleared by a rejected step until a
  // step is accepted again; gathers_pending: the records and group sums already belong to
  // d_x (they were made for the accepted candidate), the rest of the gathers has not run
  bool spec_ok = true, gathers_pending = false;
  bool at_least_one_successful = false, step_was_successful = true;
  int iteration = 0, consecutive_invalid = 0;
  double cost = 0.0;
  calico_rp_summary sum;
  std::chrono::steady_clock::time_point t_begin;
};

}  // namespace

struct calico_rp_problem {
  int cn = 0, pn = 0, tn = 0, vn = 0, p_star = -1, t_star = -1, device = 0;
  // 0: reprojection error (Stage 5, the hot path); 1: algebraic error (Stage 4, calico_ae_*):
  // an "observation" is then one relation A_i with its 7-vector in h_a7, 12 residual rows.
  int kind = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // Observations go straight to the device (d_u / d_v / d_corner, or d_a7 for relations), in
  // arrival order behind what is already there; the host keeps one ObsRun per maximal run of
  // observations of one (camera, pattern, time) and nothing per observation.  ensure_ready
  // sorts the runs into FR order and moves the payload on the device (build_structure).
  std::vector<ObsRun> runs;
  int64_t n_obs = 0;        // observations held
  int64_t n_added = 0;      // caller's index of the next observation
  int32_t max_corner = -1;  // largest model point index any observation refers to
  std::vector<double> h_intr;
  std::vector<float> h_pts;
  bool structure_dirty = true, intr_dirty = true, pts_dirty = true;
  int jac_mode = CALICO_RP_JACOBIAN_FUSED;
  int speculation = 1;
  // Camera parameters refined (calico_rp_set_refine_intrinsics): bit k frees parameter k of
  // every camera.  d_x / d_cand / d_delta / d_scale / d_diag / d_gscaled are always the extended
  // vector [7 per variable | 12 per camera]; its tail is only used when the mask is non-zero.
  uint32_t refine_mask = 0;
  bool no_denominator = false;  // k4 = k5 = k6 = 0 for every camera (set where the intrinsics are uploaded)
  DevBuf<double> d_recK, d_recK2, d_chunkK_sum, d_groupK_sum;
  size_t nx() const { return (size_t)7 * vn + (size_t)12 * cn; }
  Structure st;
  // device: observations and structure
  DevBuf<double> d_u, d_v, d_intr, d_J, d_recA, d_recB, d_pp_chunk_sum, d_pp_sum, d_a7;
  DevBuf<double> d_recA2, d_recB2;  // the candidate's records while a step is evaluated speculatively
  // ... and their segment diagonals (second set, swapped in with the records when the step is
  // accepted).  The gathers of a speculative evaluation run on a stream of their own: the segment
  // diagonals right behind the fused kernel, beside the cost sum and the host's decision; the pair
  // sums and H_rr / g_r / diag after an accepted step, beside the next iteration's SYRK
  // (evaluate_candidate, assemble_reduced)
  DevBuf<double> d_Ett2, d_gt2;
  DevBuf<double> d_wave_cost;       // fused kernel: sum r^2 per wave
  hipStream_t gstream = nullptr;    // the gathers' stream
  hipEvent_t ev_fused = nullptr, ev_seg = nullptr, ev_y = nullptr, ev_hrr = nullptr;
  bool side_busy = false;           // work queued on gstream that the main stream has not waited for yet (ev_hrr)
  DevBuf<double> d_frtab;           // [F][64]: per-FR table of the chain form (fused kernel)
  DevBuf<double> d_prep;            // [vn + 1][12]: R | t of every block at the point being evaluated (fused kernel)
  DevBuf<int32_t> d_part_fr;
  DevBuf<int64_t> d_part_obs_begin;
  DevBuf<int32_t> d_wave_part0, d_part_piece, d_seg_part_begin, d_seg_piece_begin, d_pp_piece, d_pp_chunk_begin,
      d_pp_rblock_chunk_begin;
  DevBuf<float> d_pts;
  DevBuf<int32_t> d_corner, d_obs_fr, d_fr_cam, d_fr_pat, d_fr_tim;
  DevBuf<int64_t> d_fr_obs_begin;
  DevBuf<int32_t> d_rvar, d_seg_var, d_pair_part, d_chunk_begin, d_group_chunk_begin,
      d_group_rc, d_group_rp, d_rblock_group_begin, d_rblock_groups, d_unit_seg, d_unit_rblock,
      d_unit_begin, d_contrib, d_seg_unit_begin;
  // device: normal equations
  DevBuf<double> d_chunk_sum, d_group_sum, d_Hrr, d_gr_local, d_hdiag_local, d_Ett, d_gt, d_Y, d_Lt,
      d_z, d_partial, d_pack /*P | q | gr | hdiag*/, d_S, d_Lfac, d_rhs, d_invL, d_bacc, d_Su, d_wu, d_segw;
  DevBuf<double> d_scale, d_diag, d_gscaled, d_x, d_cand, d_delta, d_w, d_seg_partial, d_cost_partial,
      d_scal, d_scal_all;
  DevBuf<int> d_fail;
  // multi-GPU
  ncclComm_t comm = nullptr;
  bool comm_lost = false;          // the communicator was aborted after an error on this rank
  double comm_timeout_s = 120.0;   // CALICO_RP_COMM_TIMEOUT_S: how long a collective may keep the stream busy
  double* h_scal_all = nullptr;    // pinned, [SC_N][nranks]: the per-iteration scalars of all ranks
  double* h_scal = nullptr;        // pinned, [SC_N]: the same read-back without a communicator
  int nranks = 1, rank = 0;
  DevBuf<double> d_xchg;  // packed [lower(P) | q | g_r | diag] for the all-reduce
  DevBuf<double> d_own, d_owners, d_tsum;  // lm_end: time blocks this rank owns / that any rank owns / their sum
  bool owners_valid = false;
  // profiling
  std::vector<hipEvent_t> evpool;
  size_t ev_used = 0;
  struct Pending {
    double* acc;
    int32_t* cnt;
    int e0, e1;
  };
  std::vector<Pending> pending;
  calico_rp_stage_ms cum_stage = {};
  int grab_event() {
    if (ev_used == evpool.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return -1;
      evpool.push_back(e);
    }
    return (int)ev_used++;
  }
  Lm lm;

  double* P() { return d_pack.p; }
  double* q() { return d_pack.p + (size_t)st.ldn * st.ldn; }
  double* gr_sum() { return q() + st.ldn; }
  double* hdiag_sum() { return gr_sum() + st.ldn; }
  // what the reduced system is finalised from: the ranks' sums in the pack buffer, or, without a
  // communicator, the gathers' own output (no copy into the pack buffer then)
  double* gr_cur() { return comm ? gr_sum() : d_gr_local.p; }
  double* hdiag_cur() { return comm ? hdiag_sum() : d_hdiag_local.p; }
  size_t pack_count() const { return (size_t)st.ldn * st.ldn + 3 * (size_t)st.ldn; }
};

namespace {

// ---- stage timing with HIP events on the library's own stream.  Events are only
// recorded while work is queued; they are read after the iteration's single
// stream synchronisation, so profiling adds no host/device round trips.
struct StageTimer {
  calico_rp_problem* h;
  double* acc;
  int32_t* cnt;
  bool on;
  int e0 = -1;
  hipStream_t stream;
  StageTimer(calico_rp_problem* hh, double* a, int32_t* c, bool enable, hipStream_t on_stream = nullptr);
  ~StageTimer();
};
void resolve_timers(calico_rp_problem* h);

int fail(calico_rp_problem* h, int code, const std::string& m) {
  if (h) h->err = m;
  return code;
}

StageTimer::StageTimer(calico_rp_problem* hh, double* a, int32_t* c, bool enable, hipStream_t on_stream)
    : h(hh), acc(a), cnt(c), on(enable), stream(on_stream ? on_stream : hh->stream) {
  if (!on) return;
  e0 = h->grab_event();
  if (e0 < 0) { on = false; return; }
  (void)hipEventRecord(h->evpool[e0], stream);
}
StageTimer::~StageTimer() {
  if (!on) return;
  const int e1 = h->grab_event();
  if (e1 < 0) return;
  (void)hipEventRecord(h->evpool[e1], stream);
  h->pending.push_back({acc, cnt, e0, e1});
}
// call only after the stream has been synchronised
void resolve_timers(calico_rp_problem* h) {
  for (auto& p : h->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->evpool[p.e0], h->evpool[p.e1]) == hipSuccess) {
      *p.acc += ms;
      *p.cnt += 1;
      // the same slot of the handle's cumulative record
      char* base = reinterpret_cast<char*>(&h->lm.sum.stage);
      char* cum = reinterpret_cast<char*>(&h->cum_stage);
      const ptrdiff_t oa = reinterpret_cast<char*>(p.acc) - base, oc = reinterpret_cast<char*>(p.cnt) - base;
      if (oa >= 0 && oa < (ptrdiff_t)sizeof(calico_rp_stage_ms)) {
        *reinterpret_cast<double*>(cum + oa) += ms;
        *reinterpret_cast<int32_t*>(cum + oc) += 1;
      }
    }
  }
  h->pending.clear();
  h->ev_used = 0;
}

// CALICO_RP_TIMING=1: wall time of the set-up phases on stderr (tools/setup_time.py)
struct PhaseTimer {
  bool on = getenv("CALICO_RP_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "calico_rp setup: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// Groups the observations into FRs sorted by (time [t* last], pattern, camera) and
// derives every index list the kernels need.  O(N + F log F).
void build_structure(calico_rp_problem* h) {
  PhaseTimer pt;
  Structure& S = h->st;
  S = Structure();
  const int64_t N = h->n_obs;
  S.N = N;
  typedef ObsRun Run;
  std::vector<Run> runs = h->runs;  // arrival order: run starts were found when the observations were added
  auto less = [](const Run& a, const Run& b) {
    if (a.t != b.t) return a.t < b.t;
    if (a.p != b.p) return a.p < b.p;
    return a.c < b.c;
  };
  bool sorted = true;
  for (size_t k = 1; k < runs.size() && sorted; ++k) sorted = less(runs[k - 1], runs[k]);
  if (!sorted) {
    // stable merge sort on a few threads: sorted chunks, then pairwise in-place merges
    const int nt = host_threads((int64_t)runs.size());
    const int64_t R = (int64_t)runs.size(), per = (R + nt - 1) / nt;
    parallel_chunks(R, nt, [&](int64_t b, int64_t e, int) { std::stable_sort(runs.begin() + b, runs.begin() + e, less); });
    for (int64_t width = per; width < R; width *= 2) {
      const int64_t pairs = (R + 2 * width - 1) / (2 * width);
      parallel_chunks(pairs, (int)std::min<int64_t>(pairs, nt), [&](int64_t pb, int64_t pe, int) {
        for (int64_t pi = pb; pi < pe; ++pi) {
          const int64_t lo = pi * 2 * width, mid = std::min(R, lo + width), hi = std::min(R, lo + 2 * width);
          if (mid < hi) std::inplace_merge(runs.begin() + lo, runs.begin() + mid, runs.begin() + hi, less);
        }
      });
    }
  }
  pt.lap("runs + sort");
  S.fr_cam.reserve(runs.size());
  S.fr_pat.reserve(runs.size());
  S.fr_tim.reserve(runs.size());
  S.fr_obs_begin.reserve(runs.size() + 1);
  int64_t pos = 0;
  bool in_place = true;
  S.identity_perm = true;
  for (size_t k = 0; k < runs.size(); ++k) {
    const bool new_fr = (k == 0) || less(runs[k - 1], runs[k]);
    if (new_fr) {
      S.fr_cam.push_back(runs[k].c);
      S.fr_pat.push_back(runs[k].p == h->p_star ? -1 : runs[k].p);
      S.fr_tim.push_back(runs[k].t == std::numeric_limits<int32_t>::max() ? -1 : runs[k].t);
      S.fr_obs_begin.push_back(pos);
    }
    in_place = in_place && runs[k].begin == pos;
    S.identity_perm = S.identity_perm && runs[k].orig == pos;
    pos += runs[k].len;
  }
  S.fr_obs_begin.push_back(pos);
  S.F = (int)S.fr_cam.size();
  if (!in_place) {
    S.move_src.resize(runs.size());
    S.move_dst.resize(runs.size());
    S.move_len.resize(runs.size());
  }
  pos = 0;
  for (size_t k = 0; k < runs.size(); ++k) {
    if (!in_place) {
      S.move_src[k] = runs[k].begin;
      S.move_dst[k] = pos;
      S.move_len[k] = runs[k].len;
    }
    runs[k].begin = pos;  // from now on: the sorted position
    pos += runs[k].len;
  }
  S.run_map.swap(runs);

  pt.lap("FRs + perm");
  // reduced blocks
  std::vector<char> used(h->vn, 0);
  for (int f = 0; f < S.F; ++f) {
    used[S.fr_cam[f]] = 1;
    if (S.fr_pat[f] >= 0) used[S.fr_pat[f]] = 1;
  }
  // Sharded runs must agree on the reduced unknowns whatever each rank happens to
  // observe: every camera and every pattern but p* is a reduced block there.  An
  // unobserved block only gets its LM diagonal (min_lm_diagonal / radius): step 0.
  if (h->comm)
    for (int v = 0; v < h->cn + h->pn; ++v) used[v] = (v != h->p_star);
  S.ridx.assign(h->vn, -1);
  for (int v = 0; v < h->cn + h->pn; ++v)
    if (used[v]) {
      S.ridx[v] = S.n_rblocks++;
      S.rvar.push_back(v);
    }
  if (h->refine_mask)
    for (int v : S.rvar) S.nk += v < h->cn ? 1 : 0;
  S.n = 7 * S.n_rblocks + 12 * S.nk;
  S.ldn = ((S.n + 1) + 127) / 128 * 128;

  // eliminated time segments (FRs are time-sorted; t* FRs are last)
  S.seg_fr_begin.clear();
  for (int f = 0; f < S.F; ++f) {
    if (S.fr_tim[f] < 0) break;
    if (f == 0 || S.fr_tim[f] != S.fr_tim[f - 1]) {
      S.seg_var.push_back(S.fr_tim[f]);
      S.seg_fr_begin.push_back(f);
    }
  }
  S.n_seg = (int)S.seg_var.size();
  {
    int f_elim = 0;
    while (f_elim < S.F && S.fr_tim[f_elim] >= 0) ++f_elim;
    S.seg_fr_begin.push_back(f_elim);
  }
  S.n_krows = std::max(16, (7 * S.n_seg + 15) / 16 * 16);

  // parts and pieces.  The accumulation kernels give every wave a fixed range of
  // S.wave_obs consecutive observations; a "part" is the intersection of one FR with one
  // wave's range (record A), a "piece" a run of consecutive parts inside one wave that
  // share (time, pattern) (record B).  Parts are numbered in observation order.
  S.wave_obs = rp::acc_wave_obs(N);
  const int64_t L = S.wave_obs;
  const int64_t n_waves = (N + L - 1) / L;
  S.wave_part0.assign((size_t)n_waves + 1, 0);
  S.part_fr.clear();
  S.part_obs_begin.clear();
  {
    int f = 0;
    for (int64_t w = 0; w < n_waves; ++w) {
      const int64_t i0 = w * L, i1 = std::min(N, i0 + L);
      S.wave_part0[w] = (int)S.part_fr.size();
      while (f < S.F && S.fr_obs_begin[f + 1] <= i0) ++f;
      for (int g = f; g < S.F && S.fr_obs_begin[g] < i1; ++g) {
        S.part_fr.push_back(g);
        S.part_obs_begin.push_back(std::max(i0, S.fr_obs_begin[g]));
      }
    }
    S.wave_part0[n_waves] = (int)S.part_fr.size();
    S.part_obs_begin.push_back(N);
  }
  S.n_parts = (int)S.part_fr.size();
  S.part_piece.assign((size_t)S.n_parts + 1, -1);
  S.seg_part_begin.assign((size_t)S.n_seg + 1, S.n_parts);
  S.seg_piece_begin.assign((size_t)S.n_seg + 1, 0);
  std::vector<int32_t> piece_pat;
  {
    int w = 0;
    for (int p = 0; p < S.n_parts; ++p) {
      while (S.wave_part0[w + 1] <= p) ++w;
      const int f = S.part_fr[p], fp = p ? S.part_fr[p - 1] : -1;
      // the algebraic-error records kernel gives a wave rp::AE_WAVE_PARTS consecutive parts
      const bool wave_start = h->kind == 1 ? p % rp::AE_WAVE_PARTS == 0 : p == S.wave_part0[w];
      const bool fresh = p == 0 || wave_start || S.fr_tim[f] != S.fr_tim[fp] || S.fr_pat[f] != S.fr_pat[fp];
      if (fresh) piece_pat.push_back(S.fr_pat[f]);
      S.part_piece[p] = (int)piece_pat.size() - 1;
    }
    S.n_pieces = (int)piece_pat.size();
    // segment -> part / piece ranges (a time change always starts a new part and piece)
    int p = 0;
    for (int sgi = 0; sgi <= S.n_seg; ++sgi) {
      while (p < S.n_parts && S.part_fr[p] < S.seg_fr_begin[sgi]) ++p;
      S.seg_part_begin[sgi] = p;
      S.seg_piece_begin[sgi] = p < S.n_parts ? S.part_piece[p] : S.n_pieces;
    }
  }
  pt.lap("parts + pieces");
  // pattern diagonal blocks: pieces per reduced pattern block, in chunks
  {
    std::vector<std::vector<int32_t>> per_rb(S.n_rblocks);
    for (int k = 0; k < S.n_pieces; ++k)
      if (piece_pat[k] >= 0) per_rb[S.ridx[piece_pat[k]]].push_back(k);
    S.pp_rblock_chunk_begin.assign(1, 0);
    for (auto& v : per_rb) {
      for (size_t i = 0; i < v.size(); i += rp::PAIR_CHUNK) S.pp_chunk_begin.push_back((int)S.pp_piece.size() + (int)i);
      S.pp_piece.insert(S.pp_piece.end(), v.begin(), v.end());
      S.pp_rblock_chunk_begin.push_back((int)S.pp_chunk_begin.size());
    }
    S.pp_chunk_begin.push_back((int)S.pp_piece.size());
  }

  // (camera, pattern) groups of parts
  std::vector<int32_t> order(S.n_parts);
  for (int p = 0; p < S.n_parts; ++p) order[p] = p;
  const int64_t kmul = (int64_t)S.n_rblocks + 1;
  std::vector<int64_t> keys(S.n_parts);
  for (int p = 0; p < S.n_parts; ++p) {
    const int f = S.part_fr[p];
    const int64_t rc = S.ridx[S.fr_cam[f]];
    const int64_t rp = S.fr_pat[f] >= 0 ? S.ridx[S.fr_pat[f]] : -1;
    keys[p] = rc * kmul + (rp + 1);  // dense: < n_rblocks * (n_rblocks + 1)
  }
  auto gkey = [&](int p) { return keys[p]; };
  if (kmul * kmul <= ((int64_t)1 << 24)) {
    // stable counting sort by group key
    std::vector<int32_t> count((size_t)(kmul * kmul) + 1, 0);
    for (int p = 0; p < S.n_parts; ++p) ++count[keys[p] + 1];
    for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
    for (int p = 0; p < S.n_parts; ++p) order[count[keys[p]]++] = p;
  } else {
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return keys[a] < keys[b]; });
  }
  S.pair_part = order;
  S.chunk_begin.clear();
  S.group_chunk_begin.clear();
  for (int i = 0; i < S.n_parts;) {
    int j = i + 1;
    while (j < S.n_parts && gkey(order[j]) == gkey(order[i])) ++j;
    const int f = S.part_fr[order[i]];
    S.group_chunk_begin.push_back((int)S.chunk_begin.size());
    S.group_rc.push_back(S.ridx[S.fr_cam[f]]);
    S.group_rp.push_back(S.fr_pat[f] >= 0 ? S.ridx[S.fr_pat[f]] : -1);
    for (int k = i; k < j; k += rp::PAIR_CHUNK) S.chunk_begin.push_back(k);
    i = j;
  }
  S.group_chunk_begin.push_back((int)S.chunk_begin.size());
  S.chunk_begin.push_back(S.n_parts);
  const int n_groups = (int)S.group_rc.size();
  std::vector<std::vector<int32_t>> rb_groups(S.n_rblocks);
  for (int g = 0; g < n_groups; ++g) {
    rb_groups[S.group_rc[g]].push_back(g);
    if (S.group_rp[g] >= 0) rb_groups[S.group_rp[g]].push_back(g);
  }
  S.rblock_group_begin.assign(1, 0);
  for (auto& v : rb_groups) {
    S.rblock_groups.insert(S.rblock_groups.end(), v.begin(), v.end());
    S.rblock_group_begin.push_back((int)S.rblock_groups.size());
  }

  pt.lap("pattern lists + groups");
  // (segment, reduced block) units
  S.unit_begin.assign(1, 0);
  S.seg_unit_begin.assign(1, 0);
  std::vector<std::pair<int32_t, int32_t>> tmp;  // (rblock, contrib)
  for (int sgi = 0; sgi < S.n_seg; ++sgi) {
    tmp.clear();
    for (int p = S.seg_part_begin[sgi]; p < S.seg_part_begin[sgi + 1]; ++p)
      tmp.emplace_back(S.ridx[S.fr_cam[S.part_fr[p]]], 2 * p);
    for (int k = S.seg_piece_begin[sgi]; k < S.seg_piece_begin[sgi + 1]; ++k)
      if (piece_pat[k] >= 0) tmp.emplace_back(S.ridx[piece_pat[k]], 2 * k + 1);
    std::sort(tmp.begin(), tmp.end());
    for (size_t k = 0; k < tmp.size(); ++k) {
      if (k == 0 || tmp[k].first != tmp[k - 1].first) {
        if (k != 0) S.unit_begin.push_back((int)S.contrib.size());
        S.unit_seg.push_back(sgi);
        S.unit_rblock.push_back(tmp[k].first);
      }
      S.contrib.push_back(tmp[k].second);
    }
    if (!tmp.empty()) S.unit_begin.push_back((int)S.contrib.size());
    S.seg_unit_begin.push_back((int)S.unit_seg.size());
  }

  // split-K: a multiple of 8 (slab s runs on XCD s % 8, rp_linalg.hip: syrk_kernel) that gives
  // every XCD about two rounds of its 96 workgroup slots (3 workgroups of 162 VGPRs per CU)
  auto split = [&](int pairs) {
    int per_xcd = (2 * 96 + pairs / 2) / pairs;
    // short K (a shard of a sharded solve): fewer, longer slabs; the slab reduce reads every slab's tiles,
    // whatever their K (one eighth of config D: elimination + SYRK 0.375 -> 0.323 ms with 3 instead of 7)
    per_xcd = std::min(per_xcd, std::max(2, S.n_krows / 2500));
    int ks = 8 * std::max(1, per_xcd);
    const int most = std::max(1, S.n_krows / 64);
    if (ks > most) ks = most >= 8 ? most / 8 * 8 : most;
    return ks;
  };
  S.n_pairs = rp::syrk_tile_pairs(S.ldn);
  S.ksplit = split(S.n_pairs);
  S.n_pairs_w = rp::syrk_tile_pairs(S.ldu());
  S.ksplit_w = split(S.n_pairs_w);
  pt.lap("units");
}

void upload_structure(calico_rp_problem* h) {
  PhaseTimer pt;
  Structure& S = h->st;
  hipStream_t s = h->stream;
  const int64_t N = S.N;
  // the payload into sorted order, on the device (nothing to do when it arrived that way)
  if (!S.move_src.empty()) {
    DevBuf<int64_t> d_src, d_dst, d_len;
    d_src.upload(S.move_src, s);
    d_dst.upload(S.move_dst, s);
    d_len.upload(S.move_len, s);
    const int64_t nr = (int64_t)S.move_src.size();
    // every array into a fresh buffer first, then all of them swapped in together: an allocation
    // that fails half way (config D moves 2 GB) leaves the handle as it was, runs and payload still
    // in arrival order, and the next call starts over
    auto move = [&](auto& buf, auto& sorted_buf, int words_per_element) {
      sorted_buf.alloc(buf.n);
      rp::launch_move_runs(nr, d_src.p, d_dst.p, d_len.p, reinterpret_cast<const uint32_t*>(buf.p),
                           reinterpret_cast<uint32_t*>(sorted_buf.p), words_per_element, s);
    };
    if (h->kind == 1) {
      DevBuf<double> a7;
      move(h->d_a7, a7, 14);
      HIP_CHECK(hipStreamSynchronize(s));
      h->d_a7.swap(a7);
    } else {
      DevBuf<int32_t> corner;
      DevBuf<double> u, v;
      move(h->d_corner, corner, 1);
      move(h->d_u, u, 2);
      move(h->d_v, v, 2);
      HIP_CHECK(hipStreamSynchronize(s));
      h->d_corner.swap(corner);
      h->d_u.swap(u);
      h->d_v.swap(v);
    }
    S.move_src = std::vector<int64_t>();
    S.move_dst = std::vector<int64_t>();
    S.move_len = std::vector<int64_t>();
  }
  h->runs = S.run_map;  // the payload now sits at the sorted positions
  if (h->kind == 1) {
    h->d_part_fr.upload(S.part_fr, s);
    h->d_part_obs_begin.upload(S.part_obs_begin, s);
  }
  pt.lap("observations to FR order");
  h->d_fr_obs_begin.upload(S.fr_obs_begin, s);
  h->d_obs_fr.alloc((size_t)std::max<int64_t>(1, N));
  rp::launch_fill_obs_fr(S.F, h->d_fr_obs_begin.p, h->d_obs_fr.p, s);
  h->d_fr_cam.upload(S.fr_cam, s);
  h->d_fr_pat.upload(S.fr_pat, s);
  h->d_fr_tim.upload(S.fr_tim, s);
  h->d_rvar.upload(S.rvar, s);
  h->d_seg_var.upload(S.seg_var, s);
  h->d_pair_part.upload(S.pair_part, s);
  h->d_chunk_begin.upload(S.chunk_begin, s);
  h->d_group_chunk_begin.upload(S.group_chunk_begin, s);
  h->d_group_rc.upload(S.group_rc, s);
  h->d_group_rp.upload(S.group_rp, s);
  h->d_rblock_group_begin.upload(S.rblock_group_begin, s);
  h->d_rblock_groups.upload(S.rblock_groups, s);
  h->d_unit_seg.upload(S.unit_seg, s);
  h->d_unit_rblock.upload(S.unit_rblock, s);
  h->d_unit_begin.upload(S.unit_begin, s);
  h->d_contrib.upload(S.contrib, s);
  h->d_seg_unit_begin.upload(S.seg_unit_begin, s);
  h->d_wave_part0.upload(S.wave_part0, s);
  h->d_part_piece.upload(S.part_piece, s);
  h->d_seg_part_begin.upload(S.seg_part_begin, s);
  h->d_seg_piece_begin.upload(S.seg_piece_begin, s);
  h->d_pp_piece.upload(S.pp_piece, s);
  h->d_pp_chunk_begin.upload(S.pp_chunk_begin, s);
  h->d_pp_rblock_chunk_begin.upload(S.pp_rblock_chunk_begin, s);
  HIP_CHECK(hipStreamSynchronize(s));  // host vectors above go out of scope
  pt.lap("index lists upload");

  const size_t ld2 = (size_t)S.ldn * S.ldn;
  h->d_J.release();  // allocated on first use (ensure_jacobian_tiles): the fused path never needs it
  h->d_recA.alloc((size_t)std::max(1, S.n_parts) * REC_A);
  h->d_recB.alloc((size_t)std::max(1, S.n_pieces) * REC_B);
  // the fused kernel never writes the padding row/column 7 of the pattern/time blocks
  h->d_recA.zero(s);
  h->d_recB.zero(s);
  h->d_recA2.release();  // allocated by the first speculative evaluation
  h->d_recB2.release();
  h->d_recK2.release();
  h->d_Ett2.release();
  h->d_gt2.release();
  h->d_wave_cost.alloc((size_t)std::max<int64_t>(1, (N + S.wave_obs - 1) / S.wave_obs));
  if (S.nk > 0) {
    h->d_recK.alloc((size_t)std::max(1, S.n_parts) * rp::REC_K);
    h->d_chunkK_sum.alloc((size_t)(S.chunk_begin.size() - 1) * rp::REC_K);
    h->d_groupK_sum.alloc((size_t)std::max<size_t>(1, S.group_rc.size()) * rp::REC_K);
  } else {
    h->d_recK.release();
    h->d_chunkK_sum.release();
    h->d_groupK_sum.release();
  }
  h->d_chunk_sum.alloc((size_t)(S.chunk_begin.size() - 1) * 128);
  h->d_group_sum.alloc((size_t)S.group_rc.size() * 128);
  h->d_pp_chunk_sum.alloc((size_t)std::max<size_t>(1, S.pp_chunk_begin.size() - 1) * 64);
  h->d_pp_sum.alloc((size_t)std::max(1, S.n_rblocks) * 64);
  h->d_Hrr.alloc(ld2);
  h->d_Hrr.zero(s);
  h->d_gr_local.alloc(S.ldn);
  h->d_gr_local.zero(s);
  h->d_hdiag_local.alloc(S.ldn);
  h->d_hdiag_local.zero(s);
  h->d_Ett.alloc((size_t)std::max(1, S.n_seg) * 49);
  h->d_gt.alloc((size_t)std::max(1, S.n_seg) * 7);
  h->d_Y.alloc((size_t)S.n_krows * S.ldn);
  h->d_Y.zero(s);
  h->d_Su.alloc(ld2);
  h->d_wu.alloc(S.ldn);
  h->d_Lt.alloc((size_t)std::max(1, S.n_seg) * 49);
  h->d_z.alloc((size_t)std::max(1, S.n_seg) * 7);
  h->d_segw.alloc((size_t)std::max(1, S.n_seg) * rp::SEGW);
  h->d_partial.alloc((size_t)std::max(S.ksplit * S.n_pairs, S.ksplit_w * S.n_pairs_w) * 128 * 128);
  h->d_pack.alloc(h->pack_count());
  h->d_pack.zero(s);
  h->d_S.alloc(ld2);
  h->d_Lfac.alloc(ld2);
  h->d_Lfac.zero(s);  // the factor below the diagonal blocks, its transposed inverse above (rp_linalg.hip); what a
                      // partial last block leaves unwritten is read times zero
  h->d_bacc.alloc(ld2);  // partial sums of the factor's inverse (rp_linalg.hip: chol_inverse_ahead)
  h->d_bacc.zero(s);
  h->d_rhs.alloc(S.ldn);
  h->d_invL.alloc((size_t)((S.n + 63) / 64 + 1) * 4096);
  h->d_scale.alloc(h->nx());
  h->d_diag.alloc(h->nx());
  h->d_gscaled.alloc(h->nx());
  h->d_x.alloc(h->nx());
  h->d_cand.alloc(h->nx());
  h->d_delta.alloc(h->nx());
  h->d_w.alloc(S.ldn);
  h->d_seg_partial.alloc((size_t)std::max(1, S.n_seg) * rp::SEGP);
  h->d_cost_partial.alloc(rp::SWEEP_COST_GRID);
  h->d_scal.alloc(SC_N);
  h->d_scal.zero(s);
  h->d_scal_all.alloc((size_t)SC_N * std::max(1, h->nranks));
  h->d_fail.alloc(1);
  h->d_fail.zero(s);
  HIP_CHECK(hipStreamSynchronize(s));
  pt.lap("device buffers");
}

// sorted position -> caller's index (the eval hooks return results in the caller's order)
std::vector<int64_t> caller_index(const calico_rp_problem* h) {
  std::vector<int64_t> perm((size_t)h->st.N);
  for (const ObsRun& r : h->st.run_map)
    for (int64_t i = 0; i < r.len; ++i) perm[(size_t)(r.begin + i)] = r.orig + i;
  return perm;
}

void ensure_ready(calico_rp_problem* h) {
  HIP_CHECK(hipSetDevice(h->device));
  if (h->structure_dirty) {
    build_structure(h);
    upload_structure(h);
    h->structure_dirty = false;
    h->owners_valid = false;
  }
  if (h->intr_dirty) {
    h->d_intr.upload(h->h_intr, h->stream);
    h->intr_dirty = false;
    h->no_denominator = true;
    for (int c = 0; c < h->cn; ++c)
      for (int k = 9; k < 12; ++k) h->no_denominator = h->no_denominator && h->h_intr[(size_t)12 * c + k] == 0.0;
  }
  if (h->pts_dirty && h->kind == 0) {
    h->d_pts.upload(h->h_pts, h->stream);
    h->pts_dirty = false;
  }
  HIP_CHECK(hipStreamSynchronize(h->stream));
}

// The materialised Jacobian, [ntiles][44][256] doubles (36 GB on config D).
void ensure_jacobian_tiles(calico_rp_problem* h) {
  const size_t need = (size_t)((h->st.N + rp::SWEEP_TILE - 1) / rp::SWEEP_TILE) * rp::SWEEP_PLANES * rp::SWEEP_TILE;
  if (h->d_J.n != need) h->d_J.alloc(need);
}

rp::SweepArgs sweep_args(calico_rp_problem* h, const double* x_dev) {
  rp::SweepArgs a;
  a.n_obs = h->st.N;
  a.x = x_dev;
  // camera parameters refined: they are the tail of the point being evaluated
  const bool extended = h->refine_mask && (x_dev == h->d_x.p || x_dev == h->d_cand.p);
  a.intr = extended ? x_dev + (size_t)7 * h->vn : h->d_intr.p;
  a.refine_mask = h->refine_mask;
  // the kernels without the rational model's denominator are only right while k4..k6 stay
  // zero: not when the mask frees one of them (they then move away from the zeros handed over)
  a.no_denominator = (h->no_denominator && !(h->refine_mask & 0xE00u)) ? 1 : 0;
  a.pts = h->d_pts.p;
  a.obs_u = h->d_u.p;
  a.obs_v = h->d_v.p;
  a.obs_corner = h->d_corner.p;
  a.obs_fr = h->d_obs_fr.p;
  a.fr_cam = h->d_fr_cam.p;
  a.fr_pat = h->d_fr_pat.p;
  a.fr_tim = h->d_fr_tim.p;
  a.J = h->d_J.p;
  a.a7 = h->d_a7.p;
  a.n_vars = h->vn;
  if (h->d_prep.n != (size_t)12 * (h->vn + 1)) h->d_prep.alloc((size_t)12 * (h->vn + 1));
  a.prep = a.prep_out = h->d_prep.p;
  if (h->kind == 0 && h->jac_mode == CALICO_RP_JACOBIAN_FUSED) {
    const size_t need = (size_t)64 * (size_t)std::max(1, h->st.F);
    if (h->d_frtab.n != need) h->d_frtab.alloc(need);
    a.frtab = h->d_frtab.p;
    a.wave_cost = h->d_wave_cost.p;
  }
  return a;
}

rp::GatherArgs gather_args(calico_rp_problem* h) {
  Structure& S = h->st;
  rp::GatherArgs g;
  g.recA = h->d_recA.p;
  g.recB = h->d_recB.p;
  // the fused kernel leaves its records in rotation-vector space, made at d_x
  g.wspace = (h->kind == 0 && h->jac_mode == CALICO_RP_JACOBIAN_FUSED) ? 1 : 0;
  g.nk = g.wspace ? S.nk : 0;
  g.n7 = S.n7();
  g.recK = h->d_recK.p;
  g.chunkK_sum = h->d_chunkK_sum.p;
  g.groupK_sum = h->d_groupK_sum.p;
  g.x = h->d_x.p;
  g.rvar = h->d_rvar.p;
  g.seg_var = h->d_seg_var.p;
  g.cost_index = g.wspace ? 6 * 8 + 6 : 63;
  g.n_pp_chunks = (int)S.pp_chunk_begin.size() - 1;
  g.pp_chunk_begin = h->d_pp_chunk_begin.p;
  g.pp_piece = h->d_pp_piece.p;
  g.pp_chunk_sum = h->d_pp_chunk_sum.p;
  g.pp_rblock_chunk_begin = h->d_pp_rblock_chunk_begin.p;
  g.pp_sum = h->d_pp_sum.p;
  g.seg_piece_begin = h->d_seg_piece_begin.p;
  g.n_chunks = (int)S.chunk_begin.size() - 1;
  g.chunk_begin = h->d_chunk_begin.p;
  g.pair_part = h->d_pair_part.p;
  g.chunk_sum = h->d_chunk_sum.p;
  g.n_groups = (int)S.group_rc.size();
  g.group_chunk_begin = h->d_group_chunk_begin.p;
  g.group_rc = h->d_group_rc.p;
  g.group_rp = h->d_group_rp.p;
  g.group_sum = h->d_group_sum.p;
  g.n_rblocks = S.n_rblocks;
  g.rblock_group_begin = h->d_rblock_group_begin.p;
  g.rblock_groups = h->d_rblock_groups.p;
  g.n = S.n;
  g.ldn = S.ldn;
  g.Hrr = h->d_Hrr.p;
  g.gr = h->d_gr_local.p;
  g.hdiag = h->d_hdiag_local.p;
  g.cost_sum = h->d_scal.p + SC_COSTSUM;
  g.n_seg = S.n_seg;
  g.seg_part_begin = h->d_seg_part_begin.p;
  g.Ett = h->d_Ett.p;
  g.gt = h->d_gt.p;
  g.n_units = (int)S.unit_seg.size();
  g.unit_seg = h->d_unit_seg.p;
  g.seg_unit_begin = h->d_seg_unit_begin.p;
  g.unit_rblock = h->d_unit_rblock.p;
  g.unit_begin = h->d_unit_begin.p;
  g.contrib = h->d_contrib.p;
  return g;
}

void nccl_check(calico_rp_problem*, ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw HipError{std::string(what) + ": " + g_rccl.GetErrorString(r)};
}

// A rank that fails between collectives must not leave its peers blocked in the next one, and a
// rank whose peer has gone must not wait for ever:
//   * this rank fails (any error that unwinds to guarded()): the communicator is aborted
//     (ncclCommAbort) before the error is returned; every later call on the handle answers
//     CALICO_RP_ESTATE;
//   * a peer fails: this rank's collective never completes, stream_sync() gives up after
//     comm_timeout_s, and the error takes the path above (abort, non-zero status).
void comm_abort(calico_rp_problem* h) {
  if (!h->comm || h->nranks <= 1) return;
  if (g_rccl.CommAbort) g_rccl.CommAbort(h->comm);  // frees the communicator; a stuck kernel of ours ends
  h->comm = nullptr;
  h->comm_lost = true;
}
void stream_sync(calico_rp_problem* h) {
  if (!h->comm || h->nranks <= 1) {
    HIP_CHECK(hipStreamSynchronize(h->stream));
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(h->stream);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) HIP_CHECK(e);
    if (spins > 4096) {  // past the first instants: look at the clock, be polite to the core
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (waited > h->comm_timeout_s)
        throw HipError{"a collective did not complete within " + std::to_string((int)h->comm_timeout_s) +
                       " s: a peer rank has failed or left (CALICO_RP_COMM_TIMEOUT_S)"};
      if (waited > 0.002) usleep(50);
    }
  }
}

// Speculative evaluation needs a second set of records: the current point's stay valid
// until the candidate is accepted.
// (Independent of what this rank holds: in a sharded solve every rank must take the same
// branch, or their costs and accept / reject decisions part ways.)
bool can_speculate(const calico_rp_problem* h) {
  return h->speculation && h->kind == 0 && h->jac_mode == CALICO_RP_JACOBIAN_FUSED;
}
void ensure_candidate_records(calico_rp_problem* h) {
  if (h->d_recA2.n == h->d_recA.n && h->d_recB2.n == h->d_recB.n && h->d_recK2.n == h->d_recK.n &&
      h->d_Ett2.n == h->d_Ett.n)
    return;
  hipStream_t s = h->stream;
  h->d_recA2.alloc(h->d_recA.n);
  h->d_recB2.alloc(h->d_recB.n);
  h->d_recK2.alloc(h->d_recK.n);  // fully written by the fused kernel
  h->d_recA2.zero(s);  // padding rows / columns are never written by the fused kernel
  h->d_recB2.zero(s);
  h->d_Ett2.alloc(h->d_Ett.n);
  h->d_gt2.alloc(h->d_gt.n);
}

// The main stream waits for whatever the gathers' stream still has queued (their scratch buffers
// and the second set of sums are free again afterwards).
void join_side(calico_rp_problem* h) {
  if (!h->side_busy) return;
  HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_hrr, 0));
  h->side_busy = false;
}

// Whether the block gathers go to the second stream.  Each hand-over between the streams costs an
// event wait (about 10 us): worth it when the gathers are long enough to hide (0.3 ms on config D),
// not on small networks (config A, 3 200 FRs: 0.18 -> 0.20 ms per iteration with it).
bool gathers_beside(const calico_rp_problem* h) {
#ifdef CALICO_RP_TEST_HOOKS
  // test build only (tests/test_gpu_streams.py): "0" / "1" force the choice, so that both orders of the
  // same solve can be compared bit for bit on one network
  if (const char* force = getenv("CALICO_RP_TEST_BESIDE")) return force[0] == '1';
#endif
  return h->st.n_parts >= 32768;
}

int64_t fused_waves(const calico_rp_problem* h) { return (h->st.N + h->st.wave_obs - 1) / h->st.wave_obs; }

// Fused kernel at the candidate into the second record set; *cost_out = 1/2 sum r^2 (this rank's
// observations) from the kernel's per-wave sums.  The segment diagonals of those records go to the
// second set on the gathers' stream, beside the cost sum and the host's accept / reject decision;
// the rest of what the next iteration needs from the records (pair sums, H_rr / g_r / diag) is
// gathered after an accepted step, beside that iteration's SYRK (assemble_reduced).
void evaluate_candidate(calico_rp_problem* h, double* cost_out, calico_rp_stage_ms* st, bool prof, bool prof_sweep,
                        bool sum_segments) {
  hipStream_t s = h->stream, gs = h->gstream;
  ensure_candidate_records(h);
  if (h->st.N <= 0) {  // a shard without observations contributes nothing
    HIP_CHECK(hipMemsetAsync(cost_out, 0, sizeof(double), s));
    rp::launch_flag_to_scalar(h->d_fail.p, h->d_scal.p + SC_FAIL, s);
    return;
  }
  join_side(h);
  rp::SweepArgs a = sweep_args(h, h->d_cand.p);
  a.recK = h->d_recK2.p;
  {
    StageTimer t(h, &st->accumulate, &st->n_accumulate, prof);
    rp::launch_fused_prepass(a, h->st.F, s);
  }
  {
    StageTimer t(h, &st->jacobian_sweep, &st->n_jacobian_sweep, prof_sweep);
    rp::launch_fused(a, h->st.wave_obs, h->st.F, h->d_fr_obs_begin.p, h->d_wave_part0.p, h->d_part_piece.p, h->d_recA2.p,
                     h->d_recB2.p, s);
  }
  HIP_CHECK(hipEventRecord(h->ev_fused, s));
  // (with the iteration's failure flag: the read-back follows this launch directly;
  // and, if the back-substitution left them out (sum_segments), the sums over its segments)
  rp::launch_sum_partials(h->d_wave_cost.p, (int)fused_waves(h), 0.5, cost_out, s, h->d_fail.p, h->d_scal.p + SC_FAIL,
                          sum_segments ? h->d_seg_partial.p : nullptr, h->st.n_seg, h->d_scal.p + SC_BACK + 3);
  if (!gathers_beside(h)) return;  // small network: all gathers inline at the start of the next iteration
  HIP_CHECK(hipStreamWaitEvent(gs, h->ev_fused, 0));
  {
    StageTimer t(h, &st->reduce, &st->n_reduce, prof, gs);
    rp::GatherArgs g = gather_args(h);
    g.recB = h->d_recB2.p;
    g.x = h->d_cand.p;
    g.Ett = h->d_Ett2.p;
    g.gt = h->d_gt2.p;
    rp::launch_gathers(g, gs, rp::GATHER_SEG);
  }
  HIP_CHECK(hipEventRecord(h->ev_seg, gs));
  HIP_CHECK(hipEventRecord(h->ev_hrr, gs));  // what join_side waits for
  h->side_busy = true;
}

// Jacobian sweep + accumulation + gathers at d_x.  Returns true when the pair sums and H_rr / g_r /
// diag are still to be gathered (fused mode: assemble_reduced does that on the gathers' stream
// beside the SYRK, as after an accepted speculative step; the segment diagonals are done here).
bool evaluate(calico_rp_problem* h, calico_rp_stage_ms* st, bool prof, bool prof_sweep) {
  hipStream_t s = h->stream;
  join_side(h);  // the gathers below share their scratch buffers with those of a speculative evaluation
  bool cost_from_waves = false;
  if (h->kind == 1) {
    StageTimer t(h, &st->jacobian_sweep, &st->n_jacobian_sweep, prof_sweep);
    rp::launch_ae_records(sweep_args(h, h->d_x.p), h->st.n_parts, h->d_part_fr.p, h->d_part_obs_begin.p, h->d_part_piece.p,
                          h->d_recA.p, h->d_recB.p, s);
  } else if (h->jac_mode == CALICO_RP_JACOBIAN_FUSED) {
    // the kernel's time is booked under jacobian_sweep, its pre-pass (per-variable / per-FR tables)
    // under accumulate
    rp::SweepArgs a = sweep_args(h, h->d_x.p);
    a.recK = h->d_recK.p;
    {
      StageTimer t(h, &st->accumulate, &st->n_accumulate, prof);
      rp::launch_fused_prepass(a, h->st.F, s);
    }
    StageTimer t(h, &st->jacobian_sweep, &st->n_jacobian_sweep, prof_sweep);
    rp::launch_fused(a, h->st.wave_obs, h->st.F, h->d_fr_obs_begin.p, h->d_wave_part0.p, h->d_part_piece.p, h->d_recA.p,
                     h->d_recB.p, s);
    cost_from_waves = h->st.N > 0;
  } else {
    ensure_jacobian_tiles(h);
    {
      StageTimer t(h, &st->jacobian_sweep, &st->n_jacobian_sweep, prof_sweep);
      rp::launch_jacobian_sweep(sweep_args(h, h->d_x.p), s);
    }
    {
      StageTimer t(h, &st->accumulate, &st->n_accumulate, prof);
      rp::launch_accumulate(h->st.N, h->st.wave_obs, h->d_obs_fr.p, h->d_fr_obs_begin.p, h->d_wave_part0.p, h->d_part_piece.p,
                            h->d_J.p, h->d_recA.p, h->d_recB.p, s);
    }
  }
  {
    StageTimer t(h, &st->reduce, &st->n_reduce, prof);
    rp::GatherArgs g = gather_args(h);
    if (cost_from_waves) {  // the same sum a speculative evaluation of this point would have given
      rp::launch_sum_partials(h->d_wave_cost.p, (int)fused_waves(h), 1.0, h->d_scal.p + SC_COSTSUM, s);
      rp::launch_gathers(g, s, rp::GATHER_SEG);
      return true;
    }
    rp::launch_gathers(g, s);
  }
  return false;
}

// 1/2 sum r^2 at x_dev -> *out (device), without the Jacobian
void cost_at(calico_rp_problem* h, const double* x_dev, double* out, hipStream_t s) {
  if (h->kind == 1)
    rp::launch_ae_cost(sweep_args(h, x_dev), h->d_cost_partial.p, out, s);
  else
    rp::launch_cost_sweep(sweep_args(h, x_dev), h->d_cost_partial.p, out, s);
}

// Damped Schur system for the current radius: fills S/rhs (ready for Cholesky).
void assemble_reduced(calico_rp_problem* h, double radius, int compute_scale, int use_scale, int reuse_diag,
                      const calico_rp_options& o, calico_rp_stage_ms* st, bool prof, bool gathers_pending = false,
                      bool seg_on_side = false, bool* gmax_done = nullptr) {
  Structure& S = h->st;
  hipStream_t s = h->stream;
  {
    StageTimer t(h, &st->schur, &st->n_schur, prof);
    rp::ElimArgs e;
    e.n_seg = S.n_seg; e.n = S.n; e.ldn = S.ldn; e.n_krows = S.n_krows;
    const rp::GatherArgs ga = gather_args(h);
    // rotation-vector records: Y and the SYRK keep six columns per reduced block
    const int nu = S.nu(), ldu = S.ldu();
    // ... and six rows per segment (ElimArgs::rows6)
    const int krows6 = std::max(16, (6 * S.n_seg + 15) / 16 * 16);
    if (ga.wspace) {
      e.n = nu;
      e.ldn = ldu;
      e.rows6 = 1;
      e.segw = h->d_segw.p;
      e.x = h->d_x.p;
      e.n_krows = krows6;
    }
    e.seg_var = h->d_seg_var.p; e.Ett = h->d_Ett.p; e.gt = h->d_gt.p;
    e.scale = h->d_scale.p; e.diag = h->d_diag.p; e.gscaled = h->d_gscaled.p;
    e.Lt = h->d_Lt.p; e.z = h->d_z.p; e.Y = h->d_Y.p; e.fail = h->d_fail.p;
    e.radius = radius; e.min_diag = o.min_lm_diagonal; e.max_diag = o.max_lm_diagonal;
    e.compute_scale = compute_scale; e.use_scale = use_scale; e.reuse_diag = reuse_diag;
    // after an accepted speculative step the segment diagonals come from the gathers' stream ...
    const bool beside = gathers_beside(h);
    if (seg_on_side && beside) HIP_CHECK(hipStreamWaitEvent(s, h->ev_seg, 0));
    if (gathers_pending && !beside) {  // small network: inline, in front of the elimination
      StageTimer t2(h, &st->reduce, &st->n_reduce, prof);
      rp::GatherArgs g2 = ga;
      g2.cost_sum = nullptr;
      rp::launch_gathers(g2, s, rp::GATHER_PAIRS | rp::GATHER_BLOCKS | (seg_on_side ? rp::GATHER_SEG : 0));
    }
    rp::launch_time_eliminate(e, ga, s);
    if (gathers_pending && beside) {
      // ... and the pair sums, H_rr, g_r and diag of the new records are gathered there while the
      // SYRK runs here: they start when Y is done (both stream through HBM) and are needed by the
      // conversion behind the product
      hipStream_t gs = h->gstream;
      HIP_CHECK(hipEventRecord(h->ev_y, s));
      HIP_CHECK(hipStreamWaitEvent(gs, h->ev_y, 0));
      {
        StageTimer t2(h, &st->reduce, &st->n_reduce, prof, gs);
        rp::GatherArgs g2 = ga;
        g2.cost_sum = nullptr;
        rp::launch_gathers(g2, gs, rp::GATHER_PAIRS | rp::GATHER_BLOCKS);
        if (gmax_done && !h->comm) {  // max |g| of the new point: only the host reads it (with a communicator: of the ranks' sum)
          rp::launch_gradient_max(g2.gr, S.n, h->d_gt.p, 7 * S.n_seg, h->d_scal.p + SC_GMAX, gs);
          *gmax_done = true;
        }
      }
      HIP_CHECK(hipEventRecord(h->ev_hrr, gs));
      h->side_busy = true;
    }
    rp::SyrkArgs a;
    a.n = S.n; a.ldn = S.ldn; a.n_krows = S.n_krows; a.ksplit = S.ksplit;
    a.Y = h->d_Y.p; a.Hrr = h->d_Hrr.p; a.gr = h->d_gr_local.p;
    a.partial = h->d_partial.p; a.P = h->P(); a.q = h->q();
    if (ga.wspace) {
      a.n = nu; a.ldn = ldu; a.Hrr = nullptr; a.P = h->d_Su.p; a.ksplit = S.ksplit_w;
      if (e.rows6) a.n_krows = krows6;
      rp::SyrkConvertArgs c;
      c.n_rblocks = S.n_rblocks; c.n = S.n; c.ldn = S.ldn; c.nu = nu; c.ldu = ldu;
      c.n7 = S.n7(); c.nu6 = 6 * S.n_rblocks;
      c.rvar = h->d_rvar.p; c.x = h->d_x.p; c.Su = h->d_Su.p; c.Hrr = h->d_Hrr.p; c.gr = h->d_gr_local.p;
      c.P = h->P(); c.q = h->q();
      rp::launch_schur_syrk_w_product(a, s);
      join_side(h);  // H_rr, g_r, diag: gathered beside the product
      rp::launch_schur_syrk_w_convert(c, s);
    } else {
      join_side(h);
      rp::launch_schur_syrk(a, s);
    }
    if (h->comm) {  // the all-reduce below replaces them by the ranks' sums; the local ones are needed again after a rejected step
      HIP_CHECK(hipMemcpyAsync(h->gr_sum(), h->d_gr_local.p, S.ldn * sizeof(double), hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(h->hdiag_sum(), h->d_hdiag_local.p, S.ldn * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
  }
  if (h->comm) {
    StageTimer t(h, &st->exchange, &st->n_exchange, prof);
    const size_t cnt = rp::exchange_count(S.n);
    if (h->d_xchg.n != cnt) h->d_xchg.alloc(cnt);
    rp::launch_exchange_pack(S.n, S.ldn, h->d_pack.p, h->d_xchg.p, s);
    nccl_check(h, g_rccl.AllReduce(h->d_xchg.p, h->d_xchg.p, cnt, ncclDouble, ncclSum, h->comm, s),
               "ncclAllReduce(reduced system)");
    rp::launch_exchange_unpack(S.n, S.ldn, h->d_xchg.p, h->d_pack.p, s);
  }
  {
    StageTimer t(h, &st->schur, &st->n_schur, false);
    rp::ReducedArgs a;
    a.n = S.n; a.ldn = S.ldn; a.n7 = S.n7(); a.vn7 = 7 * h->vn;
    a.rvar = h->d_rvar.p; a.P = h->P(); a.q = h->q(); a.gr = h->gr_cur();
    a.hdiag = h->hdiag_cur(); a.scale = h->d_scale.p; a.diag = h->d_diag.p; a.gscaled = h->d_gscaled.p;
    a.S = h->d_S.p; a.rhs = h->d_rhs.p; a.radius = radius; a.min_diag = o.min_lm_diagonal;
    a.max_diag = o.max_lm_diagonal; a.compute_scale = compute_scale; a.use_scale = use_scale;
    a.reuse_diag = reuse_diag;
    rp::launch_finalize_reduced(a, s);
  }
}

void termination(Lm& lm, int type, const char* msg) {
  lm.done = true;
  lm.sum.termination_type = type;
  std::snprintf(lm.sum.message, sizeof lm.sum.message, "%s", msg);
}

// One LM iteration.  See the file header for how it maps onto Ceres' loop.
void lm_step_impl(calico_rp_problem* h, int32_t* accepted) {
  Lm& lm = h->lm;
  Structure& S = h->st;
  hipStream_t s = h->stream;
  const calico_rp_options& o = lm.opt;
  // profile 1: every stage; 2: the Jacobian / fused sweep only (an event pair per stage costs the
  // iteration 0.1 ms on config D: the packets keep consecutive kernels from overlapping)
  const bool prof = o.profile == 1, prof_sweep = o.profile != 0;
  calico_rp_stage_ms* st = &lm.sum.stage;
  if (accepted) *accepted = 0;
  if (lm.done) return;
#ifdef CALICO_RP_TEST_HOOKS
  if (h->comm) {
    // test build only (tests/test_gpu_comm.py): "rank:iteration" makes that rank fail there,
    // between two collectives, the way a HIP error would
    static const char* inject = getenv("CALICO_RP_TEST_FAIL_AT");
    int fr = -1, fi = -1;
    if (inject && sscanf(inject, "%d:%d", &fr, &fi) == 2 && fr == h->rank && fi == lm.iteration)
      throw HipError{"injected failure (CALICO_RP_TEST_FAIL_AT)"};
  }
#endif
  if (lm.iteration >= o.max_num_iterations) {
    termination(lm, CALICO_RP_NO_CONVERGENCE, "Maximum number of iterations reached.");
    return;
  }
  if (lm.radius < o.min_trust_region_radius) {
    termination(lm, CALICO_RP_CONVERGENCE, "Minimum trust region radius reached.");
    return;
  }
  // fresh: the normal equations are formed at a new point in this call, either by a full
  // evaluation or by finishing the gathers of an accepted speculative evaluation
  const bool full_eval = !lm.jac_valid;
  const bool new_eval = full_eval || lm.gathers_pending;
  // pending: the records at d_x are new (a full evaluation, or those of the accepted candidate, which
  // were swapped in with their segment diagonals) and their other sums are still to be gathered:
  // beside the SYRK, see assemble_reduced
  const bool pending = full_eval ? evaluate(h, st, prof, prof_sweep) : lm.gathers_pending;
  const int compute_scale = lm.first_eval ? 1 : 0;
  bool gmax_done = false;
  assemble_reduced(h, lm.radius, compute_scale, o.jacobi_scaling, lm.reuse_diag ? 1 : 0, o, st, prof, pending,
                   !full_eval && pending, new_eval ? &gmax_done : nullptr);
  lm.gathers_pending = false;
  if (new_eval && !gmax_done) rp::launch_gradient_max(h->gr_cur(), S.n, h->d_gt.p, 7 * S.n_seg, h->d_scal.p + SC_GMAX, s);
  {
    StageTimer t(h, &st->cholesky, &st->n_cholesky, prof);
    rp::launch_cholesky_solve(S.n, S.ldn, h->d_S.p, h->d_Lfac.p, h->d_invL.p, h->d_rhs.p, h->d_bacc.p, h->d_fail.p, s);
  }
  const bool speculative = can_speculate(h) && lm.spec_ok && lm.iteration + 1 < o.max_num_iterations;
  // the sums over the segments' partial scalars are only read by the host: the speculative
  // evaluation's cost sum takes them along (one launch less in front of the fused kernel)
  const bool defer_sum = speculative && S.N > 0 && S.n_seg > 0;
  {
    StageTimer t(h, &st->backsub, &st->n_backsub, prof);
    rp::BacksubArgs b;
    b.defer_sum = defer_sum ? 1 : 0;
    b.n = S.n; b.ldn = S.ldn; b.n_seg = S.n_seg; b.n_rblocks = S.n_rblocks; b.vn7 = 7 * h->vn; b.n7 = S.n7();
    b.rvar = h->d_rvar.p; b.seg_var = h->d_seg_var.p; b.yr = h->d_rhs.p; b.Y = h->d_Y.p; b.Lt = h->d_Lt.p;
    b.z = h->d_z.p; b.scale = h->d_scale.p; b.diag = h->d_diag.p; b.gscaled = h->d_gscaled.p;
    b.x = h->d_x.p; b.w = h->d_w.p; b.delta = h->d_delta.p; b.cand = h->d_cand.p;
    b.seg_partial = h->d_seg_partial.p; b.scalars = h->d_scal.p + SC_BACK; b.radius = lm.radius;
    if (gather_args(h).wspace) {
      b.wspace = 1;
      b.ny = S.nu();
      b.ldy = S.ldu();
      b.wu = h->d_wu.p;
      b.segw = h->d_segw.p;
    }
    // blocks that are not optimised keep their values: d_cand started as a copy of d_x (lm_begin),
    // the kernels below write every optimised block, and the two buffers only ever swap
    if (S.n_seg == 0) HIP_CHECK(hipMemsetAsync(h->d_scal.p + SC_BACK + 3, 0, 3 * sizeof(double), s));
    rp::launch_backsub(b, s);
  }
  // Cost at the candidate.  Speculative evaluation: most LM steps are accepted and the next
  // iteration then needs residuals, Jacobian and records at exactly this point, so the fused
  // kernel runs there now, into the second record set, and its r^T r is the candidate's cost
  // (Ceres evaluates the cost alone and the Jacobian after accepting,
  // TrustRegionMinimizer::EvaluateGradientAndJacobian; same numbers, one sweep less per
  // accepted step).  Not after a rejected step (then the cost-only sweep, 5x cheaper, until a
  // step is accepted again) and not on the last iteration the options allow.
  if (speculative) {
    evaluate_candidate(h, h->d_scal.p + SC_CAND, st, prof, prof_sweep, defer_sum);
  } else {
    StageTimer t(h, &st->cost_sweep, &st->n_cost_sweep, prof);
    cost_at(h, h->d_cand.p, h->d_scal.p + SC_CAND, s);
    rp::launch_flag_to_scalar(h->d_fail.p, h->d_scal.p + SC_FAIL, s);  // (the speculative evaluation's cost sum does it)
  }
  // ---- one read-back per iteration
  double sc[SC_N];
  int failflag = 0;
  double cost_sum = 0, gmax = 0, cand_cost = 0, mcc = 0, d2 = 0, x2 = 0;
  if (h->comm) {
    // (the fail flag travels as a double in slot SC_FAIL)
    nccl_check(h, g_rccl.AllGather(h->d_scal.p, h->d_scal_all.p, SC_N, ncclDouble, h->comm, s), "ncclAllGather(scalars)");
    const double* all = h->h_scal_all;  // pinned, sized by comm_init
    HIP_CHECK(hipMemcpyAsync(h->h_scal_all, h->d_scal_all.p, (size_t)SC_N * h->nranks * sizeof(double), hipMemcpyDeviceToHost, s));
    stream_sync(h);
    for (int rk = 0; rk < h->nranks; ++rk) {
      const double* a = &all[(size_t)SC_N * rk];
      cost_sum += a[SC_COSTSUM];
      gmax = std::max(gmax, a[SC_GMAX]);
      cand_cost += a[SC_CAND];
      mcc += a[SC_BACK + 3];
      d2 += a[SC_BACK + 4];
      x2 += a[SC_BACK + 5];
      if (a[SC_FAIL] != 0.0) failflag = 1;
    }
    const double* a0 = &all[(size_t)SC_N * h->rank];
    mcc += a0[SC_BACK + 0];
    d2 += a0[SC_BACK + 1];
    x2 += a0[SC_BACK + 2];
  } else {
    // one DMA into pinned memory: the fail flag has joined the scalars on the device
    if (!h->h_scal) HIP_CHECK(hipHostMalloc((void**)&h->h_scal, SC_N * sizeof(double), hipHostMallocDefault));
    HIP_CHECK(hipMemcpyAsync(h->h_scal, h->d_scal.p, SC_N * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    std::memcpy(sc, h->h_scal, sizeof sc);
    failflag = sc[SC_FAIL] != 0.0;
    cost_sum = sc[SC_COSTSUM];
    gmax = sc[SC_GMAX];
    cand_cost = sc[SC_CAND];
    mcc = sc[SC_BACK + 0] + sc[SC_BACK + 3];
    d2 = sc[SC_BACK + 1] + sc[SC_BACK + 4];
    x2 = sc[SC_BACK + 2] + sc[SC_BACK + 5];
  }

  if (prof && h->side_busy) HIP_CHECK(hipStreamSynchronize(h->gstream));  // its event pair is read below
  resolve_timers(h);

  if (new_eval) {
    if (full_eval) lm.cost = 0.5 * cost_sum;  // else: the speculative evaluation's cost, already in lm.cost
    lm.jac_valid = true;
    if (lm.first_eval) {
      lm.sum.initial_cost = lm.cost;
      lm.first_eval = false;
    }
    if (o.verbose)
      std::printf("calico_rp iter %3d cost %.12e |g|max %.3e radius %.3e\n", lm.iteration, lm.cost, gmax, lm.radius);
    if (lm.step_was_successful && gmax <= o.gradient_tolerance) {
      termination(lm, CALICO_RP_CONVERGENCE, "Gradient tolerance reached.");
      return;
    }
  }
  ++lm.iteration;
  lm.step_was_successful = false;
  lm.reuse_diag = true;  // LevenbergMarquardtStrategy::ComputeStep sets reuse_diagonal_ = true

  const bool valid = !failflag && std::isfinite(mcc) && mcc > 0.0 && std::isfinite(d2);
  if (!valid) {
    ++lm.sum.num_unsuccessful_steps;
    lm.spec_ok = false;
    if (++lm.consecutive_invalid >= 5) {
      termination(lm, CALICO_RP_FAILURE, "Number of consecutive invalid steps more than 5.");
      return;
    }
    lm.radius /= lm.decrease_factor;
    lm.decrease_factor *= 2.0;
    if (o.verbose) std::printf("calico_rp iter %3d invalid step (fail=%d mcc=%.3e)\n", lm.iteration, failflag, mcc);
    return;
  }
  lm.consecutive_invalid = 0;
  if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
  const double step_norm = std::sqrt(d2), x_norm = std::sqrt(x2);
  if (lm.at_least_one_successful && step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
    ++lm.sum.num_unsuccessful_steps;
    termination(lm, CALICO_RP_CONVERGENCE, "Parameter tolerance reached.");
    return;
  }
  const double cost_change = lm.cost - cand_cost;
  if (std::fabs(cost_change) <= o.function_tolerance * lm.cost) {
    ++lm.sum.num_unsuccessful_steps;
    termination(lm, CALICO_RP_CONVERGENCE, "Function tolerance reached.");
    return;
  }
  const double rho = cost_change / mcc;
  if (rho > o.min_relative_decrease) {
    std::swap(h->d_x.p, h->d_cand.p);
    lm.cost = cand_cost;  // replaced by the evaluated cost at the next iteration (full evaluation)
    lm.jac_valid = false;
    if (speculative) {  // the candidate's records become the current ones
      std::swap(h->d_recA.p, h->d_recA2.p);
      std::swap(h->d_recB.p, h->d_recB2.p);
      std::swap(h->d_recK.p, h->d_recK2.p);
      std::swap(h->d_Ett.p, h->d_Ett2.p);
      std::swap(h->d_gt.p, h->d_gt2.p);
      lm.jac_valid = true;
      lm.gathers_pending = true;
    }
    lm.spec_ok = true;
    lm.radius = lm.radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
    lm.radius = std::min(o.max_trust_region_radius, lm.radius);
    lm.decrease_factor = 2.0;
    lm.reuse_diag = false;
    lm.at_least_one_successful = true;
    lm.step_was_successful = true;
    ++lm.sum.num_successful_steps;
    if (accepted) *accepted = 1;
  } else {
    lm.radius /= lm.decrease_factor;
    lm.decrease_factor *= 2.0;
    ++lm.sum.num_unsuccessful_steps;
    lm.spec_ok = false;
  }
  if (o.verbose)
    std::printf("calico_rp iter %3d cand %.12e change %.3e |step| %.3e rho %.3e radius %.3e %s\n", lm.iteration,
                cand_cost, cost_change, step_norm, rho, lm.radius, lm.step_was_successful ? "accepted" : "rejected");
}

// x (7 per variable, host) -> d_x; with the camera parameters refined the handle's
// intrinsics become the tail of the point.  Asynchronous on the handle's stream.
void upload_point(calico_rp_problem* h, const double* x) {
  HIP_CHECK(hipMemcpyAsync(h->d_x.p, x, (size_t)7 * h->vn * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->refine_mask)
    HIP_CHECK(hipMemcpyAsync(h->d_x.p + (size_t)7 * h->vn, h->h_intr.data(), (size_t)12 * h->cn * sizeof(double),
                             hipMemcpyHostToDevice, h->stream));
}

void lm_begin_impl(calico_rp_problem* h, const double* x, const calico_rp_options* opt) {
  ensure_ready(h);
  Lm& lm = h->lm;
  lm = Lm();
  lm.opt = *opt;
  lm.active = true;
  lm.radius = opt->initial_trust_region_radius;
  std::memset(&lm.sum, 0, sizeof lm.sum);
  lm.sum.num_successful_steps = 1;  // iteration 0, as Ceres counts it
  lm.sum.num_residual_blocks = h->st.N;
  lm.sum.termination_type = CALICO_RP_NO_CONVERGENCE;
  std::snprintf(lm.sum.message, sizeof lm.sum.message, "Maximum number of iterations reached.");
  lm.t_begin = std::chrono::steady_clock::now();
  join_side(h);  // (a solve that was left without lm_end may still have gathers reading the candidate's buffers)
  upload_point(h, x);
  // the candidate buffer starts as a copy (blocks no kernel writes stay equal in both through every
  // swap), the failure flag clear (flag_to_scalar_kernel re-clears it at every read-back)
  HIP_CHECK(hipMemcpyAsync(h->d_cand.p, h->d_x.p, h->nx() * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIP_CHECK(hipMemsetAsync(h->d_fail.p, 0, sizeof(int), h->stream));
  stream_sync(h);
}

void lm_end_impl(calico_rp_problem* h, double* x, calico_rp_summary* sum) {
  Lm& lm = h->lm;
  hipStream_t s = h->stream;
  join_side(h);
  if (h->comm && h->st.N >= 0) {
    // every rank owns the time blocks of its own segments: zero the others and sum, on the
    // device; which blocks have an owner at all is a property of the sharding (one more
    // all-reduce the first time, cached until the structure or the communicator changes)
    const int t0 = h->cn + h->pn;
    if (!h->owners_valid) {
      std::vector<double> own((size_t)h->tn, 0.0);
      for (int v : h->st.seg_var) own[v - t0] = 1.0;
      h->d_own.upload(own, s);
      h->d_owners.upload(own, s);
      h->d_tsum.alloc((size_t)7 * h->tn);
      stream_sync(h);  // `own` goes out of scope
      nccl_check(h, g_rccl.AllReduce(h->d_owners.p, h->d_owners.p, own.size(), ncclDouble, ncclSum, h->comm, s),
                 "ncclAllReduce(owners)");
      h->owners_valid = true;
    }
    rp::launch_time_blocks_pack(h->d_x.p, h->d_own.p, t0, h->tn, h->d_tsum.p, s);
    nccl_check(h, g_rccl.AllReduce(h->d_tsum.p, h->d_tsum.p, (size_t)7 * h->tn, ncclDouble, ncclSum, h->comm, s),
               "ncclAllReduce(time blocks)");
    rp::launch_time_blocks_unpack(h->d_x.p, h->d_owners.p, t0, h->tn, h->d_tsum.p, s);
  }
  if (x) HIP_CHECK(hipMemcpyAsync(x, h->d_x.p, (size_t)7 * h->vn * sizeof(double), hipMemcpyDeviceToHost, s));
  if (h->refine_mask) {
    // the refined camera parameters replace the handle's intrinsics (calico_rp_get_intrinsics);
    // they are reduced unknowns, so every rank of a sharded solve holds the same values
    HIP_CHECK(hipMemcpyAsync(h->h_intr.data(), h->d_x.p + (size_t)7 * h->vn, (size_t)12 * h->cn * sizeof(double),
                             hipMemcpyDeviceToHost, s));
    h->intr_dirty = true;
  }
  if (x || h->refine_mask || h->comm) stream_sync(h);
  lm.sum.final_cost = lm.cost;
  lm.sum.num_iterations = lm.sum.num_successful_steps + lm.sum.num_unsuccessful_steps;
  lm.sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lm.t_begin).count();
  if (sum) *sum = lm.sum;
  lm.active = false;
}

template <typename Fn>
int guarded(calico_rp_problem* h, Fn&& fn) {
  if (h->comm_lost)
    return fail(h, CALICO_RP_ESTATE, "the communicator of this handle was aborted after an error; destroy the handle");
  try {
    return fn();
  } catch (const HipError& e) {
    const bool sharded = h->comm && h->nranks > 1;
    comm_abort(h);
    return fail(h, CALICO_RP_EDEVICE, sharded ? e.msg + " (communicator aborted)" : e.msg);
  } catch (const std::bad_alloc&) {
    comm_abort(h);
    return fail(h, CALICO_RP_ENOMEM, "host allocation failed");
  }
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

void calico_rp_options_default(calico_rp_options* o) {
  if (!o) return;
  o->max_num_iterations = 10;  // calico.cpp:709
  o->jacobi_scaling = 1;
  o->verbose = 0;
  o->profile = 0;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
}

int calico_rp_brief_report(const calico_rp_summary* s, char* buf, size_t len) {
  if (!s || !buf || len == 0) return CALICO_RP_EINVAL;
  const char* term = s->termination_type == CALICO_RP_CONVERGENCE      ? "CONVERGENCE"
                     : s->termination_type == CALICO_RP_NO_CONVERGENCE ? "NO_CONVERGENCE"
                                                                       : "FAILURE";
  std::snprintf(buf, len, "Ceres Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
                s->num_iterations, s->initial_cost, s->final_cost, term);
  return 0;
}

int calico_rp_create(int32_t cn, int32_t pn, int32_t tn, int32_t p_star_var, int32_t t_star_var, int32_t device,
                     calico_rp_problem** out) {
  if (!out) return CALICO_RP_EINVAL;
  *out = nullptr;
  if (cn <= 0 || pn <= 0 || tn <= 0) return CALICO_RP_EINVAL;
  if (p_star_var < cn || p_star_var >= cn + pn || t_star_var < cn + pn || t_star_var >= cn + pn + tn)
    return CALICO_RP_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CALICO_RP_EDEVICE;
  auto* h = new (std::nothrow) calico_rp_problem;
  if (!h) return CALICO_RP_ENOMEM;
  h->cn = cn; h->pn = pn; h->tn = tn; h->vn = cn + pn + tn;
  h->p_star = p_star_var; h->t_star = t_star_var; h->device = device;
  h->h_intr.assign((size_t)12 * cn, 0.0);
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return CALICO_RP_EDEVICE;
  }
  if (hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fused, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_seg, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_y, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_hrr, hipEventDisableTiming) != hipSuccess) {
    calico_rp_destroy(h);
    return CALICO_RP_EDEVICE;
  }
  // the library's set-up call for this device: address of the no-allocation scratch array
  // (calico_rp_fr_errors_noalloc); a device beyond the table only loses that entry point
  if (device < NOALLOC_MAX_DEVICES && scratch_setup(device) != hipSuccess) {
    calico_rp_destroy(h);
    return CALICO_RP_EDEVICE;
  }
  *out = h;
  return 0;
}

int calico_rp_destroy(calico_rp_problem* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->gstream) (void)hipStreamSynchronize(h->gstream);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  for (hipEvent_t e : {h->ev_fused, h->ev_seg, h->ev_y, h->ev_hrr})
    if (e) (void)hipEventDestroy(e);
  if (h->h_scal_all) (void)hipHostFree(h->h_scal_all);
  if (h->h_scal) (void)hipHostFree(h->h_scal);
  for (auto& e : h->evpool)
    if (e) (void)hipEventDestroy(e);
  hipStream_t s = h->stream, gs = h->gstream;
  delete h;  // frees the device buffers
  if (gs) (void)hipStreamDestroy(gs);
  if (s) (void)hipStreamDestroy(s);
  return 0;
}

const char* calico_rp_last_error(const calico_rp_problem* h) { return h ? h->err.c_str() : "null handle"; }

int calico_rp_set_intrinsics(calico_rp_problem* h, const double* k) {
  if (!h || !k) return CALICO_RP_EINVAL;
  h->h_intr.assign(k, k + (size_t)12 * h->cn);
  h->intr_dirty = true;
  return 0;
}

int calico_rp_set_points(calico_rp_problem* h, const float* xyz, int32_t n) {
  if (!h || !xyz || n <= 0) return CALICO_RP_EINVAL;
  if (h->kind != 0) return fail(h, CALICO_RP_ESTATE, "set_points: not a reprojection-error handle");
  if (h->max_corner >= n) return fail(h, CALICO_RP_EINVAL, "set_points: fewer points than an observation already refers to");
  h->h_pts.assign(xyz, xyz + (size_t)3 * n);
  h->pts_dirty = true;
  return 0;
}

int calico_rp_add_observations(calico_rp_problem* h, int64_t n, const int32_t* cam, const int32_t* pat,
                               const int32_t* tim, const int32_t* corner, const double* u, const double* v) {
  if (!h || n < 0) return CALICO_RP_EINVAL;
  if (n == 0) return 0;
  if (h->kind != 0) return fail(h, CALICO_RP_ESTATE, "add_observations: not a reprojection-error handle");
  if (!cam || !pat || !tim || !corner || !u || !v) return fail(h, CALICO_RP_EINVAL, "add_observations: null array");
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "add_observations inside lm_begin/lm_end");
  const int64_t npts = (int64_t)h->h_pts.size() / 3;
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    // The payload (corner, u, v: 20 of the 32 bytes per observation) goes to the device behind
    // what is already there while the host threads check the indices and find the run starts;
    // no per-observation host copy is kept.
    std::string copy_error;
    const int nt = host_threads(n);
    std::vector<int64_t> bad(nt, -1);  // first offending observation of each chunk
    std::vector<int32_t> mx(nt, -1);
    std::vector<std::vector<int64_t>> starts(nt);  // (allocated before the thread starts: nothing below it throws
                                                   // between its start and its join but parallel_chunks' own threads)
    std::thread copier([&] {
      try {
        HIP_CHECK(hipSetDevice(h->device));
        h->d_corner.append_from_host((size_t)h->n_obs, corner, (size_t)n, h->stream);
        h->d_u.append_from_host((size_t)h->n_obs, u, (size_t)n, h->stream);
        h->d_v.append_from_host((size_t)h->n_obs, v, (size_t)n, h->stream);
      } catch (const HipError& e) {
        copy_error = e.msg;
      } catch (const std::bad_alloc&) {
        copy_error = "host allocation failed";
      }
    });
    struct Joiner {  // an exception on the way (bad_alloc in a chunk) must not leave a joinable thread behind
      std::thread& t;
      ~Joiner() {
        if (t.joinable()) t.join();
      }
    } joiner{copier};
    parallel_chunks(n, nt, [&](int64_t b, int64_t e, int t) {
      int32_t m = -1;
      for (int64_t i = b; i < e; ++i) {
        if (cam[i] < 0 || cam[i] >= h->cn || pat[i] < h->cn || pat[i] >= h->cn + h->pn || tim[i] < h->cn + h->pn ||
            tim[i] >= h->vn || corner[i] < 0 || corner[i] >= npts || !std::isfinite(u[i]) || !std::isfinite(v[i])) {
          bad[t] = i;
          return;
        }
        m = corner[i] > m ? corner[i] : m;
        if (i == 0 || cam[i] != cam[i - 1] || pat[i] != pat[i - 1] || tim[i] != tim[i - 1]) starts[t].push_back(i);
      }
      mx[t] = m;
    });
    copier.join();
    for (int64_t i : bad)
      if (i >= 0)  // nothing was committed: the appended tail lies beyond n_obs
        return fail(h, CALICO_RP_EINVAL, "add_observations: index out of range at observation " + std::to_string(i) +
                                             " (set_points must precede add_observations)");
    if (!copy_error.empty()) return fail(h, CALICO_RP_EDEVICE, copy_error);
    size_t total = 0;
    for (auto& sv : starts) total += sv.size();
    const size_t first = h->runs.size();
    h->runs.reserve(first + total);
    for (auto& sv : starts)
      for (int64_t i : sv)
        h->runs.push_back(ObsRun{tim[i] == h->t_star ? std::numeric_limits<int32_t>::max() : tim[i], cam[i], pat[i],
                                 h->n_obs + i, 0, h->n_added + i});
    for (size_t k = first; k < h->runs.size(); ++k)
      h->runs[k].len = (k + 1 < h->runs.size() ? h->runs[k + 1].begin : h->n_obs + n) - h->runs[k].begin;
    for (int32_t m : mx) h->max_corner = std::max(h->max_corner, m);
    h->n_obs += n;
    h->n_added += n;
    h->structure_dirty = true;
    return 0;
  });
}

int64_t calico_rp_num_observations(const calico_rp_problem* h) { return h ? h->n_obs : -1; }

int calico_rp_stage_times(calico_rp_problem* h, calico_rp_stage_ms* out, int32_t reset) {
  if (!h) return CALICO_RP_EINVAL;
  if (out) *out = h->cum_stage;
  if (reset) std::memset(&h->cum_stage, 0, sizeof h->cum_stage);
  return 0;
}

int32_t calico_rp_num_frs(calico_rp_problem* h) {
  if (!h || h->n_obs == 0) return 0;
  const int rc = guarded(h, [&] {
    ensure_ready(h);
    return 0;
  });
  return rc == 0 ? h->st.F : 0;
}

int calico_rp_fr_errors(calico_rp_problem* h, const double* x, int32_t* cam, int32_t* pat_var, int32_t* time_var,
                        int32_t* n_corners, double* mean_sq_err, double* rrmse) {
  if (!h || !x) return CALICO_RP_EINVAL;
  if (h->kind != 0) return fail(h, CALICO_RP_ESTATE, "fr_errors: not a reprojection-error handle");
  if (h->n_obs == 0) return fail(h, CALICO_RP_ESTATE, "fr_errors: no observations");
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "fr_errors: a solve is in progress");
  return guarded(h, [&] {
    ensure_ready(h);
    hipStream_t s = h->stream;
    const Structure& S = h->st;
    DevBuf<double> d_x, d_out;
    d_x.alloc((size_t)7 * h->vn);
    d_out.alloc((size_t)S.F);
    HIP_CHECK(hipMemcpyAsync(d_x.p, x, (size_t)7 * h->vn * sizeof(double), hipMemcpyHostToDevice, s));
    rp::launch_fr_errors(sweep_args(h, d_x.p), S.F, h->d_fr_obs_begin.p, d_out.p, s);
    std::vector<double> e((size_t)S.F);
    HIP_CHECK(hipMemcpyAsync(e.data(), d_out.p, e.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    double sum = 0.0;
    for (int f = 0; f < S.F; ++f) {
      sum += e[f];
      if (cam) cam[f] = S.fr_cam[f];
      if (pat_var) pat_var[f] = S.fr_pat[f] >= 0 ? S.fr_pat[f] : h->p_star;
      if (time_var) time_var[f] = S.fr_tim[f] >= 0 ? S.fr_tim[f] : h->t_star;
      if (n_corners) n_corners[f] = (int32_t)(S.fr_obs_begin[f + 1] - S.fr_obs_begin[f]);
      if (mean_sq_err) mean_sq_err[f] = e[f];
    }
    if (rrmse) *rrmse = std::sqrt(sum / (double)S.F);  // multicamera.cpp:730
    return 0;
  });
}

// calico_rp_fr_errors for callers that cannot allocate: the same kernel, its buffers in the
// file-scope scratch array (through the per-device pointer, never the array's name).  No
// hipMalloc / hipFree, no host container: the handle must already be set up on the device.
int calico_rp_fr_errors_noalloc(calico_rp_problem* h, const double* x, int32_t* cam, int32_t* pat_var, int32_t* time_var,
                                int32_t* n_corners, double* mean_sq_err, double* rrmse) {
  if (!h || !x) return CALICO_RP_EINVAL;
  if (h->kind != 0) return fail(h, CALICO_RP_ESTATE, "fr_errors_noalloc: not a reprojection-error handle");
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "fr_errors_noalloc: a solve is in progress");
  if (h->structure_dirty || h->intr_dirty || h->pts_dirty || h->n_obs == 0)
    return fail(h, CALICO_RP_ESTATE,
                "fr_errors_noalloc: the handle's observations, points or intrinsics are not on the device yet; call "
                "calico_rp_num_frs (or any evaluating entry point) once after changing them");
  const Structure& S = h->st;
  if (h->vn > NOALLOC_MAX_VARS)
    return fail(h, CALICO_RP_EINVAL, "fr_errors_noalloc: " + std::to_string(h->vn) + " variables exceed the limit of " +
                                         std::to_string(NOALLOC_MAX_VARS) + " (CALICO_RP_NOALLOC_MAX_VARS)");
  if (S.F > NOALLOC_MAX_FRS)
    return fail(h, CALICO_RP_EINVAL, "fr_errors_noalloc: " + std::to_string(S.F) + " FRs exceed the limit of " +
                                         std::to_string(NOALLOC_MAX_FRS) + " (CALICO_RP_NOALLOC_MAX_FRS)");
  if (h->device < 0 || h->device >= NOALLOC_MAX_DEVICES || !g_scratch_ptr[h->device])
    return fail(h, CALICO_RP_EDEVICE, "fr_errors_noalloc: no scratch array on this device");
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    double* const base = g_scratch_ptr[h->device];  // this GPU's copy of `scratch`
    double* const d_x = base + NOALLOC_X;
    double* const d_out = base + NOALLOC_OUT;
    double* const d_sum = base + NOALLOC_SUM;
    HIP_CHECK(hipMemcpyAsync(d_x, x, (size_t)7 * h->vn * sizeof(double), hipMemcpyHostToDevice, s));
    rp::SweepArgs a;  // by hand: sweep_args() may (re)allocate the fused kernel's table
    a.n_obs = S.N;
    a.x = d_x;
    a.intr = h->d_intr.p;
    a.pts = h->d_pts.p;
    a.obs_u = h->d_u.p;
    a.obs_v = h->d_v.p;
    a.obs_corner = h->d_corner.p;
    a.obs_fr = h->d_obs_fr.p;
    a.fr_cam = h->d_fr_cam.p;
    a.fr_pat = h->d_fr_pat.p;
    a.fr_tim = h->d_fr_tim.p;
    a.J = nullptr;
    a.n_vars = h->vn;
    rp::launch_fr_errors(a, S.F, h->d_fr_obs_begin.p, d_out, s);  // writes every one of the F means: nothing to clear
    rp::launch_sum_partials(d_out, S.F, 1.0, d_sum, s);
    double sum = 0.0;
    if (mean_sq_err) HIP_CHECK(hipMemcpyAsync(mean_sq_err, d_out, (size_t)S.F * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&sum, d_sum, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int f = 0; f < S.F; ++f) {
      if (cam) cam[f] = S.fr_cam[f];
      if (pat_var) pat_var[f] = S.fr_pat[f] >= 0 ? S.fr_pat[f] : h->p_star;
      if (time_var) time_var[f] = S.fr_tim[f] >= 0 ? S.fr_tim[f] : h->t_star;
      if (n_corners) n_corners[f] = (int32_t)(S.fr_obs_begin[f + 1] - S.fr_obs_begin[f]);
    }
    if (rrmse) *rrmse = std::sqrt(sum / (double)S.F);  // multicamera.cpp:730
    return 0;
  });
}

int calico_rp_set_jacobian_mode(calico_rp_problem* h, int32_t mode) {
  if (!h) return CALICO_RP_EINVAL;
  if (h->kind != 0) return fail(h, CALICO_RP_ESTATE, "set_jacobian_mode: not a reprojection-error handle");
  if (mode != CALICO_RP_JACOBIAN_FUSED && mode != CALICO_RP_JACOBIAN_MATERIALISED)
    return fail(h, CALICO_RP_EINVAL, "set_jacobian_mode: unknown mode");
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "set_jacobian_mode: a solve is in progress");
  if (mode == CALICO_RP_JACOBIAN_MATERIALISED && h->refine_mask)
    return fail(h, CALICO_RP_ESTATE, "set_jacobian_mode: refined camera parameters need the fused mode");
  h->jac_mode = mode;
  if (mode == CALICO_RP_JACOBIAN_FUSED) h->d_J.release();
  // the two modes lay Y out differently (6 / 7 columns per reduced block)
  if (h->d_Y.p) (void)hipMemsetAsync(h->d_Y.p, 0, h->d_Y.n * sizeof(double), h->stream);
  h->lm.jac_valid = false;
  return 0;
}

int calico_rp_set_speculation(calico_rp_problem* h, int32_t on) {
  if (!h) return CALICO_RP_EINVAL;
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "set_speculation: a solve is in progress");
  h->speculation = on ? 1 : 0;
  if (!on) {
    h->d_recA2.release();
    h->d_recB2.release();
  }
  return CALICO_RP_OK;
}

int calico_rp_set_refine_intrinsics(calico_rp_problem* h, uint32_t mask) {
  if (!h) return CALICO_RP_EINVAL;
  if (mask > 0xFFFu) return fail(h, CALICO_RP_EINVAL, "set_refine_intrinsics: the mask has 12 bits");
  if (h->kind != 0) return fail(h, CALICO_RP_ESTATE, "set_refine_intrinsics: not a reprojection-error handle");
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "set_refine_intrinsics: a solve is in progress");
  if (mask && h->jac_mode != CALICO_RP_JACOBIAN_FUSED)
    return fail(h, CALICO_RP_ESTATE, "set_refine_intrinsics: refined camera parameters need the fused mode");
  if ((mask != 0) != (h->refine_mask != 0)) h->structure_dirty = true;  // the reduced system changes size
  h->refine_mask = mask;
  h->lm.jac_valid = false;
  return CALICO_RP_OK;
}

int calico_rp_get_intrinsics(calico_rp_problem* h, double* cam_params) {
  if (!h || !cam_params) return CALICO_RP_EINVAL;
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "get_intrinsics: a solve is in progress");
  std::copy(h->h_intr.begin(), h->h_intr.end(), cam_params);
  return CALICO_RP_OK;
}

int calico_rp_synchronize(calico_rp_problem* h) {
  if (!h) return CALICO_RP_EINVAL;
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    return 0;
  });
}

int calico_rp_eval(calico_rp_problem* h, const double* x, double* cost, double* r, double* J) {
  if (!h || !x) return CALICO_RP_EINVAL;
  if (h->kind != 0) return calico_ae_eval(h, x, cost, r, J);
  if (h->n_obs == 0) return fail(h, CALICO_RP_ESTATE, "eval: no observations");
  return guarded(h, [&] {
    ensure_ready(h);
    hipStream_t s = h->stream;
    const int64_t N = h->st.N;
    upload_point(h, x);
    if (r || J) {
      ensure_jacobian_tiles(h);
      rp::launch_jacobian_sweep(sweep_args(h, h->d_x.p), s);
    }
    // camera parameters refined: 12 more columns per residual row
    const int W = h->refine_mask ? 33 : 21;
    DevBuf<double> d_kc;
    std::vector<double> kc;
    if (J && h->refine_mask) {
      d_kc.alloc((size_t)24 * N);
      kc.resize((size_t)24 * N);
      rp::launch_k_columns(sweep_args(h, h->d_x.p), d_kc.p, s);
      HIP_CHECK(hipMemcpyAsync(kc.data(), d_kc.p, kc.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    rp::launch_cost_sweep(sweep_args(h, h->d_x.p), h->d_cost_partial.p, h->d_scal.p + SC_CAND, s);
    double c = 0;
    HIP_CHECK(hipMemcpyAsync(&c, h->d_scal.p + SC_CAND, sizeof(double), hipMemcpyDeviceToHost, s));
    std::vector<double> Jt;
    if (r || J) {
      Jt.resize(h->d_J.n);
      HIP_CHECK(hipMemcpyAsync(Jt.data(), h->d_J.p, Jt.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    if (cost) *cost = c;
    const bool id = h->st.identity_perm;
    if (r || J) {
      const std::vector<int64_t> perm = id ? std::vector<int64_t>() : caller_index(h);
      const int T = rp::SWEEP_TILE, PL = rp::SWEEP_PLANES;
      for (int64_t i = 0; i < N; ++i) {
        const int64_t dst = id ? i : perm[i];
        const double* t = &Jt[(size_t)(i / T) * PL * T + (i % T)];
        if (r) {
          r[2 * dst] = t[42 * T];
          r[2 * dst + 1] = t[43 * T];
        }
        if (J)
          for (int rr = 0; rr < 2; ++rr) {
            double* row = J + (2 * dst + rr) * W;
            for (int k = 0; k < 21; ++k) row[k] = t[(21 * rr + k) * T];
            for (int k = 21; k < W; ++k) row[k] = kc[(size_t)24 * i + 12 * rr + (k - 21)];
          }
      }
    }
    h->lm.jac_valid = false;
    return 0;
  });
}

int calico_rp_normal_eq(calico_rp_problem* h, const double* x, double radius, int32_t jacobi, int32_t* n_out,
                        double* S_out, double* rhs_out) {
  if (!h || !x) return CALICO_RP_EINVAL;
  if (h->n_obs == 0) return fail(h, CALICO_RP_ESTATE, "normal_eq: no observations");
  return guarded(h, [&] {
    ensure_ready(h);
    const int n = h->st.n, ldn = h->st.ldn;
    if (n_out) *n_out = n;
    if (!S_out && !rhs_out) return 0;
    hipStream_t s = h->stream;
    upload_point(h, x);
    calico_rp_options o;
    calico_rp_options_default(&o);
    calico_rp_stage_ms st;
    std::memset(&st, 0, sizeof st);
    const bool pending = evaluate(h, &st, false, false);
    HIP_CHECK(hipMemsetAsync(h->d_fail.p, 0, sizeof(int), s));
    assemble_reduced(h, radius, 1, jacobi ? 1 : 0, 0, o, &st, false, pending);
    std::vector<double> Sh((size_t)ldn * ldn), rh(ldn);
    int failflag = 0;
    HIP_CHECK(hipMemcpyAsync(Sh.data(), h->d_S.p, Sh.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(rh.data(), h->d_rhs.p, rh.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&failflag, h->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
    stream_sync(h);  // on a sharded handle assemble_reduced has queued an all-reduce: a lost peer must not block for ever
    h->lm.jac_valid = false;
    if (failflag) return fail(h, CALICO_RP_ENUMERIC, "normal_eq: a time block is not positive definite");
    if (S_out)
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S_out[(size_t)i * n + j] = Sh[(size_t)i * ldn + j];
    if (rhs_out)
      for (int i = 0; i < n; ++i) rhs_out[i] = rh[i];
    return 0;
  });
}

int calico_rp_lm_begin(calico_rp_problem* h, const double* x, const calico_rp_options* opt) {
  if (!h || !x || !opt) return CALICO_RP_EINVAL;
  if (h->n_obs == 0 && !h->comm) return fail(h, CALICO_RP_ESTATE, "lm_begin: no observations");
  if (h->kind == 0 && h->h_pts.empty()) return fail(h, CALICO_RP_ESTATE, "lm_begin: no model points");
  return guarded(h, [&] {
    lm_begin_impl(h, x, opt);
    return 0;
  });
}

int calico_rp_lm_step(calico_rp_problem* h, int32_t* accepted, int32_t* done) {
  if (!h) return CALICO_RP_EINVAL;
  if (!h->lm.active) return fail(h, CALICO_RP_ESTATE, "lm_step without lm_begin");
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    lm_step_impl(h, accepted);
    if (done) *done = h->lm.done ? 1 : 0;
    return 0;
  });
}

int calico_rp_lm_end(calico_rp_problem* h, double* x, calico_rp_summary* summary) {
  if (!h) return CALICO_RP_EINVAL;
  if (!h->lm.active) return fail(h, CALICO_RP_ESTATE, "lm_end without lm_begin");
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    lm_end_impl(h, x, summary);
    return 0;
  });
}

int calico_rp_solve(calico_rp_problem* h, double* x, const calico_rp_options* opt, calico_rp_summary* summary) {
  if (!h || !x) return CALICO_RP_EINVAL;
  calico_rp_options o;
  if (opt) o = *opt; else calico_rp_options_default(&o);
  int rc = calico_rp_lm_begin(h, x, &o);
  if (rc) return rc;
  rc = guarded(h, [&] {
    while (!h->lm.done) lm_step_impl(h, nullptr);
    return 0;
  });
  if (rc) {
    h->lm.active = false;
    return rc;
  }
  return calico_rp_lm_end(h, x, summary);
}

// ---- Stage 4: algebraic error on the same handle type
int calico_ae_create(int32_t cn, int32_t pn, int32_t tn, int32_t p_star_var, int32_t t_star_var, int32_t device,
                     calico_rp_problem** out) {
  const int rc = calico_rp_create(cn, pn, tn, p_star_var, t_star_var, device, out);
  if (rc == 0) (*out)->kind = 1;
  return rc;
}

int calico_ae_add_relations(calico_rp_problem* h, int64_t n, const int32_t* cam, const int32_t* pat, const int32_t* tim,
                            const double* a7) {
  if (!h || n < 0) return CALICO_RP_EINVAL;
  if (h->kind != 1) return fail(h, CALICO_RP_ESTATE, "add_relations: not an algebraic-error handle");
  if (n == 0) return 0;
  if (!cam || !pat || !tim || !a7) return fail(h, CALICO_RP_EINVAL, "add_relations: null array");
  if (h->lm.active) return fail(h, CALICO_RP_ESTATE, "add_relations inside lm_begin/lm_end");
  for (int64_t i = 0; i < n; ++i) {
    if (cam[i] < 0 || cam[i] >= h->cn || pat[i] < h->cn || pat[i] >= h->cn + h->pn || tim[i] < h->cn + h->pn ||
        tim[i] >= h->vn)
      return fail(h, CALICO_RP_EINVAL, "add_relations: index out of range at relation " + std::to_string(i));
    double q2 = 0.0;
    for (int k = 0; k < 7; ++k) {
      if (!std::isfinite(a7[7 * i + k])) return fail(h, CALICO_RP_EINVAL, "add_relations: non-finite A at relation " + std::to_string(i));
      if (k < 4) q2 += a7[7 * i + k] * a7[7 * i + k];
    }
    if (q2 == 0.0) return fail(h, CALICO_RP_EINVAL, "add_relations: zero quaternion at relation " + std::to_string(i));
  }
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    h->d_a7.append_from_host((size_t)7 * h->n_obs, a7, (size_t)7 * n, h->stream);
    for (int64_t i = 0; i < n; ++i)
      if (i == 0 || cam[i] != cam[i - 1] || pat[i] != pat[i - 1] || tim[i] != tim[i - 1]) {
        if (i > 0) h->runs.back().len = h->n_obs + i - h->runs.back().begin;
        h->runs.push_back(ObsRun{tim[i] == h->t_star ? std::numeric_limits<int32_t>::max() : tim[i], cam[i], pat[i],
                                 h->n_obs + i, 0, h->n_added + i});
      }
    h->runs.back().len = h->n_obs + n - h->runs.back().begin;
    h->n_obs += n;
    h->n_added += n;
    h->structure_dirty = true;
    return 0;
  });
}

int calico_ae_eval(calico_rp_problem* h, const double* x, double* cost, double* r, double* J) {
  if (!h || !x) return CALICO_RP_EINVAL;
  if (h->kind != 1) return fail(h, CALICO_RP_ESTATE, "ae_eval: not an algebraic-error handle");
  if (h->n_obs == 0) return fail(h, CALICO_RP_ESTATE, "ae_eval: no relations");
  return guarded(h, [&] {
    ensure_ready(h);
    hipStream_t s = h->stream;
    const int64_t N = h->st.N;
    HIP_CHECK(hipMemcpyAsync(h->d_x.p, x, (size_t)7 * h->vn * sizeof(double), hipMemcpyHostToDevice, s));
    DevBuf<double> d_r, d_Jout;
    if (r) d_r.alloc((size_t)12 * N);
    if (J) d_Jout.alloc((size_t)12 * 21 * N);
    if (r || J) rp::launch_ae_eval(sweep_args(h, h->d_x.p), d_r.p, d_Jout.p, s);
    cost_at(h, h->d_x.p, h->d_scal.p + SC_CAND, s);
    double c = 0;
    HIP_CHECK(hipMemcpyAsync(&c, h->d_scal.p + SC_CAND, sizeof(double), hipMemcpyDeviceToHost, s));
    std::vector<double> rh(d_r.n), Jh(d_Jout.n);
    if (r) HIP_CHECK(hipMemcpyAsync(rh.data(), d_r.p, rh.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    if (J) HIP_CHECK(hipMemcpyAsync(Jh.data(), d_Jout.p, Jh.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (cost) *cost = c;
    const bool id = h->st.identity_perm;
    const std::vector<int64_t> perm = id ? std::vector<int64_t>() : caller_index(h);
    for (int64_t i = 0; i < N; ++i) {
      const int64_t dst = id ? i : perm[i];
      if (r) std::memcpy(r + 12 * dst, &rh[(size_t)12 * i], 12 * sizeof(double));
      if (J) std::memcpy(J + 252 * dst, &Jh[(size_t)252 * i], 252 * sizeof(double));
    }
    h->lm.jac_valid = false;
    return 0;
  });
}

int calico_rp_partition_times(int32_t tn, const int64_t* obs_per_time, int32_t nranks, int32_t* boundaries) {
  if (tn <= 0 || nranks <= 0 || !obs_per_time || !boundaries) return CALICO_RP_EINVAL;
  long double total = 0;
  for (int t = 0; t < tn; ++t) {
    if (obs_per_time[t] < 0) return CALICO_RP_EINVAL;
    total += obs_per_time[t];
  }
  boundaries[0] = 0;
  long double acc = 0;
  int t = 0;
  for (int r = 1; r < nranks; ++r) {
    const long double target = total * r / nranks;
    // advance while adding the next time keeps us closer to the target
    while (t < tn && acc + obs_per_time[t] * 0.5L <= target) acc += obs_per_time[t++];
    boundaries[r] = t;
  }
  boundaries[nranks] = tn;
  return 0;
}

int64_t calico_rp_exchange_count(int32_t n) { return n < 0 ? -1 : (int64_t)rp::exchange_count(n); }
int32_t calico_rp_exchange_ld(int32_t n) { return n < 0 ? -1 : ((n + 1) + 127) / 128 * 128; }
int calico_rp_exchange_pack_host(int32_t n, const double* work, double* buf) {
  if (n < 0 || !work || !buf) return CALICO_RP_EINVAL;
  rp::exchange_pack_host(n, calico_rp_exchange_ld(n), work, buf);
  return 0;
}
int calico_rp_exchange_unpack_host(int32_t n, const double* buf, double* work) {
  if (n < 0 || !work || !buf) return CALICO_RP_EINVAL;
  rp::exchange_unpack_host(n, calico_rp_exchange_ld(n), buf, work);
  return 0;
}

int calico_rp_dense_solve(calico_rp_problem* h, int32_t n, const double* S, const double* b, double* x) {
  if (!h || !S || !b || !x || n < 1) return CALICO_RP_EINVAL;
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int ldn = calico_rp_exchange_ld(n);
    // the solver's own layout: lower triangle in [ldn][ldn], right-hand side as row n
    std::vector<double> aug((size_t)ldn * ldn, 0.0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) aug[(size_t)i * ldn + j] = S[(size_t)i * n + j];
    for (int j = 0; j < n; ++j) aug[(size_t)n * ldn + j] = b[j];
    DevBuf<double> d_S, d_L, d_inv, d_rhs, d_P;
    DevBuf<int> d_fail;
    d_S.upload(aug, s);
    d_L.alloc(aug.size());
    d_L.zero(s);
    d_P.alloc(aug.size());
    d_P.zero(s);
    d_inv.alloc((size_t)((n + 63) / 64 + 1) * 4096);
    d_rhs.alloc((size_t)ldn);
    d_fail.alloc(1);
    d_fail.zero(s);
    rp::launch_cholesky_solve(n, ldn, d_S.p, d_L.p, d_inv.p, d_rhs.p, d_P.p, d_fail.p, s);
    int failed = 0;
    HIP_CHECK(hipMemcpyAsync(x, d_rhs.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&failed, d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (failed) return fail(h, CALICO_RP_ENUMERIC, "dense_solve: the matrix is not positive definite");
    return 0;
  });
}

int calico_rp_comm_unique_id(void* id_out) {
  if (!id_out) return CALICO_RP_EINVAL;
  std::string err;
  if (!g_rccl.load(err)) return CALICO_RP_EDEVICE;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return CALICO_RP_EDEVICE;
  static_assert(sizeof(id) == CALICO_RP_UNIQUE_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id_out, &id, sizeof id);
  return 0;
}

int calico_rp_comm_init(calico_rp_problem* h, int32_t nranks, int32_t rank, const void* idp) {
  if (!h || !idp || nranks < 1 || rank < 0 || rank >= nranks) return CALICO_RP_EINVAL;
  if (h->comm) return fail(h, CALICO_RP_ESTATE, "communicator already attached");
  std::string err;
  if (!g_rccl.load(err)) return fail(h, CALICO_RP_EDEVICE, err);
  return guarded(h, [&] {
    HIP_CHECK(hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, idp, sizeof id);
    nccl_check(h, g_rccl.CommInitRank(&h->comm, nranks, id, rank), "ncclCommInitRank");
    h->nranks = nranks;
    h->rank = rank;
    h->structure_dirty = true;  // reduced-block set is global from now on
    h->d_scal_all.alloc((size_t)SC_N * nranks);
    if (h->h_scal_all) HIP_CHECK(hipHostFree(h->h_scal_all));
    h->h_scal_all = nullptr;
    HIP_CHECK(hipHostMalloc((void**)&h->h_scal_all, (size_t)SC_N * nranks * sizeof(double), hipHostMallocDefault));
    if (const char* e = getenv("CALICO_RP_COMM_TIMEOUT_S")) h->comm_timeout_s = std::max(1.0, atof(e));
    return 0;
  });
}

}  // extern "C"
