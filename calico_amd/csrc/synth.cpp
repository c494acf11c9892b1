// synth.cpp — deterministic synthetic camera-network generator (host only, C ABI).
//
// Produces the arrays the Stage-5 refinement consumes, in the reference's layout:
//   * intrinsics   12 per camera  (solving-structure.cpp:134-159)
//   * model points float32, ncp per pattern, pattern p at [ncp*p, ncp*(p+1))
//     (solving-structure.cpp:523-526; ChArUco grid patterns.cpp:442-446, AprilTag
//     corner quadruples patterns.cpp:252-281)
//   * observations (camera, pattern var, time var, corner, u, v), one per used
//     corner (solving-structure.cpp:550-559), FR-contiguous and time-major
//   * ground truth / perturbed initial x in the 7-parameter layout with pattern and
//     time blocks holding the inverse transforms (solving-structure.cpp:422-460)
// It stands in for Stages 1–4 of the reference (detection + closed-form
// initialisation), which are out of scope.  The generator is counter-based
// (every value is a hash of (seed, stream, index)), so any time range can be
// produced on any rank without generating the rest — SURVEY.md §8(d) suggested
// std::mt19937_64; a sequential PRNG cannot shard by time, hence this choice.
//
// Geometry (millimetres): boards stand on a ring of radius 450 around the rig
// origin; the rig follows a smooth trajectory (<= 30 deg, <= 300 mm) that is the
// identity at t*; cameras sit on a 3 m sphere looking at the rig.  Which board a
// camera sees at a time is decided combinatorially (no occlusion model): either
// every board or exactly one per (camera, time).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

typedef struct calico_synth_config {
  int32_t cn, pn, tn;
  int32_t board_type;  // 0 ChArUco (sx-1)*(sy-1) corners, 1 AprilTag 4*sx*sy corners
  int32_t sx, sy;
  double square_mm;
  int32_t one_board_per_cam_time;  // 1: exactly one board per (camera,time); 0: all boards
  double fr_drop_prob;             // probability that a (c,p,t) triple is absent
  double corner_drop_prob;         // probability that a corner of a present FR is absent
  int32_t k_select;                // >0: keep at most k corners per FR (--k of calico.cpp:291-304)
  double noise_px;
  double perturb_rot_deg, perturb_trans_mm;
  int32_t tangential;              // 1: p1,p2,k3 non-zero
  int32_t p_star_idx, t_star_idx;  // pattern / time SUPERSCRIPTS of the exemplars
  uint64_t seed;
} calico_synth_config;

}  // extern "C"

namespace {

inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t h3(uint64_t seed, uint64_t stream, uint64_t a, uint64_t b = 0, uint64_t c = 0) {
  uint64_t z = mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull));
  z = mix64(z ^ a);
  z = mix64(z ^ (b + 0x632BE59BD9B4E019ull));
  z = mix64(z ^ (c + 0xA24BAED4963EE407ull));
  return z;
}
inline double u01(uint64_t h) { return ((h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
inline double gauss(uint64_t h) {
  const double a = u01(h), b = u01(mix64(h));
  return std::sqrt(-2.0 * std::log(a)) * std::cos(6.283185307179586 * b);
}

struct Pose {  // 7-vector semantics: y = R(q) x + t
  double q[4], t[3];
};
inline void rot(const double* q, double R[9]) {
  const double a = q[0], b = q[1], c = q[2], d = q[3];
  const double n = 1.0 / (a * a + b * b + c * c + d * d);
  R[0] = (a * a + b * b - c * c - d * d) * n; R[1] = 2 * (b * c - a * d) * n; R[2] = 2 * (a * c + b * d) * n;
  R[3] = 2 * (a * d + b * c) * n; R[4] = (a * a - b * b + c * c - d * d) * n; R[5] = 2 * (c * d - a * b) * n;
  R[6] = 2 * (b * d - a * c) * n; R[7] = 2 * (a * b + c * d) * n; R[8] = (a * a - b * b - c * c + d * d) * n;
}
inline void apply(const Pose& P, const double x[3], double y[3]) {
  double R[9];
  rot(P.q, R);
  for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2] + P.t[i];
}
inline void axis_angle(const double ax[3], double ang, double q[4]) {
  const double n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  const double s = std::sin(0.5 * ang) / (n > 0 ? n : 1.0);
  q[0] = std::cos(0.5 * ang); q[1] = s * ax[0]; q[2] = s * ax[1]; q[3] = s * ax[2];
}
inline void qmul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
inline void rand_axis(uint64_t seed, uint64_t stream, uint64_t i, double ax[3]) {
  for (int k = 0; k < 3; ++k) ax[k] = gauss(h3(seed, stream, i, k));
}

int ncp_of(const calico_synth_config& c) {
  return c.board_type == 0 ? (c.sx - 1) * (c.sy - 1) : 4 * c.sx * c.sy;
}

// pattern block: pattern-local -> rig frame; identity for p*.
Pose pattern_pose(const calico_synth_config& c, int p) {
  Pose P{{1, 0, 0, 0}, {0, 0, 0}};
  if (p == c.p_star_idx) return P;
  const int rel = (p - c.p_star_idx + c.pn) % c.pn;
  const double yaw = 6.283185307179586 * rel / c.pn;
  double qy[4], qt[4], ax[3] = {0, 1, 0}, tilt[3];
  axis_angle(ax, yaw, qy);
  rand_axis(c.seed, 11, p, tilt);
  axis_angle(tilt, 0.35 * (u01(h3(c.seed, 12, p)) - 0.5), qt);
  qmul(qy, qt, P.q);
  // ring of radius 450 mm, expressed relative to p* (which sits at the rig origin)
  P.t[0] = 450.0 * std::sin(yaw) + 40.0 * gauss(h3(c.seed, 13, p, 0));
  P.t[1] = 60.0 * gauss(h3(c.seed, 13, p, 1));
  P.t[2] = 450.0 * (1.0 - std::cos(yaw)) + 40.0 * gauss(h3(c.seed, 13, p, 2));
  return P;
}

// time block: rig frame -> world; identity at t*.  Smooth (three incommensurate
// sinusoids) in the signed distance from t*.
Pose time_pose(const calico_synth_config& c, int t) {
  Pose P{{1, 0, 0, 0}, {0, 0, 0}};
  if (t == c.t_star_idx) return P;
  const double s = double(t - c.t_star_idx);
  const double w = 6.283185307179586 / 97.0;
  double ax[3] = {std::sin(0.31 * w * s + 1.0), std::cos(0.47 * w * s + 2.0), 0.6 + 0.3 * std::sin(0.13 * w * s)};
  const double ang = (30.0 * 0.017453292519943295) * std::sin(w * s) * (0.6 + 0.4 * std::cos(0.37 * w * s));
  axis_angle(ax, ang, P.q);
  P.t[0] = 300.0 * std::sin(0.71 * w * s);
  P.t[1] = 200.0 * std::sin(0.53 * w * s + 0.4) - 200.0 * std::sin(0.4);
  P.t[2] = 250.0 * std::sin(0.29 * w * s + 1.1) - 250.0 * std::sin(1.1);
  return P;
}

// camera block: world -> camera.  Camera centre on a 3 m sphere, optical axis
// through the rig origin (offset by the board ring centre), random roll.
Pose camera_pose(const calico_synth_config& c, int cam) {
  double dir[3];
  rand_axis(c.seed, 21, cam, dir);
  double n = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  for (double& d : dir) d /= n;
  const double dist = 3000.0 + 300.0 * (u01(h3(c.seed, 22, cam)) - 0.5);
  const double target[3] = {0.0, 0.0, 450.0};
  double centre[3] = {target[0] + dist * dir[0], target[1] + dist * dir[1], target[2] + dist * dir[2]};
  // camera z axis = -dir (looking at the target)
  double z[3] = {-dir[0], -dir[1], -dir[2]};
  double up[3] = {0.3, 1.0, 0.2};
  if (std::fabs(z[1]) > 0.9) { up[0] = 1.0; up[1] = 0.1; }
  double xax[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
  n = std::sqrt(xax[0] * xax[0] + xax[1] * xax[1] + xax[2] * xax[2]);
  for (double& d : xax) d /= n;
  double yax[3] = {z[1] * xax[2] - z[2] * xax[1], z[2] * xax[0] - z[0] * xax[2], z[0] * xax[1] - z[1] * xax[0]};
  // R rows = camera axes in world coordinates
  const double R[9] = {xax[0], xax[1], xax[2], yax[0], yax[1], yax[2], z[0], z[1], z[2]};
  // rotation matrix -> quaternion (trace method)
  Pose P;
  const double tr = R[0] + R[4] + R[8];
  if (tr >= 0) {
    double t = std::sqrt(tr + 1.0);
    P.q[0] = 0.5 * t; t = 0.5 / t;
    P.q[1] = (R[7] - R[5]) * t; P.q[2] = (R[2] - R[6]) * t; P.q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    P.q[i + 1] = 0.5 * t; t = 0.5 / t;
    P.q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    P.q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t;
    P.q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  // roll about the optical axis
  double qr[4], zc[3] = {0, 0, 1}, q2[4];
  axis_angle(zc, 6.283185307179586 * u01(h3(c.seed, 23, cam)), qr);
  qmul(qr, P.q, q2);
  std::memcpy(P.q, q2, sizeof q2);
  double Rm[9];
  rot(P.q, Rm);
  for (int i = 0; i < 3; ++i) P.t[i] = -(Rm[3 * i] * centre[0] + Rm[3 * i + 1] * centre[1] + Rm[3 * i + 2] * centre[2]);
  return P;
}

void intrinsics_of(const calico_synth_config& c, int cam, double K[12]) {
  for (int k = 0; k < 12; ++k) K[k] = 0.0;
  K[0] = 1400.0 + 40.0 * (u01(h3(c.seed, 31, cam, 0)) - 0.5);
  K[1] = 600.0 + 20.0 * (u01(h3(c.seed, 31, cam, 1)) - 0.5);
  K[2] = 1400.0 + 40.0 * (u01(h3(c.seed, 31, cam, 2)) - 0.5);
  K[3] = 450.0 + 20.0 * (u01(h3(c.seed, 31, cam, 3)) - 0.5);
  K[4] = 0.1 * (u01(h3(c.seed, 31, cam, 4)) - 0.5);
  K[5] = 0.1 * (u01(h3(c.seed, 31, cam, 5)) - 0.5);
  if (c.tangential) {
    K[6] = 2e-3 * (u01(h3(c.seed, 31, cam, 6)) - 0.5);
    K[7] = 2e-3 * (u01(h3(c.seed, 31, cam, 7)) - 0.5);
    K[8] = 0.02 * (u01(h3(c.seed, 31, cam, 8)) - 0.5);
  }
}

void model_point(const calico_synth_config& c, int j, float X[3]) {
  if (c.board_type == 0) {  // interior chessboard corners (patterns.cpp:442-446)
    const int cols = c.sx - 1;
    X[0] = float((j % cols) * c.square_mm);
    X[1] = float((j / cols) * c.square_mm);
  } else {  // four corners per tag, tags on a grid with a 25% gap (patterns.cpp:252-281)
    const int tag = j / 4, k = j % 4;
    const double pitch = 1.25 * c.square_mm;
    const double x0 = (tag % c.sx) * pitch, y0 = (tag / c.sx) * pitch;
    const double dx = (k == 1 || k == 2) ? c.square_mm : 0.0, dy = (k >= 2) ? c.square_mm : 0.0;
    X[0] = float(x0 + dx);
    X[1] = float(y0 + dy);
  }
  X[2] = 0.0f;
}

void project(const Pose& C, const Pose& T, const Pose& P, const double K[12], const float Xf[3], double uv[2]) {
  const double X[3] = {Xf[0], Xf[1], Xf[2]};
  double z[3], y[3], xc[3];
  apply(P, X, z);
  apply(T, z, y);
  apply(C, y, xc);
  const double xn = xc[0] / xc[2], yn = xc[1] / xc[2], r2 = xn * xn + yn * yn;
  const double num = 1 + K[4] * r2 + K[5] * r2 * r2 + K[8] * r2 * r2 * r2;
  const double den = 1 + K[9] * r2 + K[10] * r2 * r2 + K[11] * r2 * r2 * r2;
  const double xpp = xn * num / den + 2 * K[6] * xn * yn + K[7] * (r2 + 2 * xn * xn);
  const double ypp = yn * num / den + 2 * K[7] * xn * yn + K[6] * (r2 + 2 * yn * yn);
  uv[0] = K[0] * xpp + K[1];
  uv[1] = K[2] * ypp + K[3];
}

inline bool fr_present(const calico_synth_config& c, int cam, int p, int t) {
  if (c.one_board_per_cam_time) {
    if (p != (cam * 3 + t) % c.pn) return false;
  }
  // the exemplar time keeps everything so that t* is always observed
  if (t == c.t_star_idx) return true;
  return u01(h3(c.seed, 41, cam, p, t)) >= c.fr_drop_prob;
}

// corners of FR (cam,p,t) that are used, ascending local index
int fr_corners(const calico_synth_config& c, int cam, int p, int t, int ncp, int* out) {
  int m = 0;
  const uint64_t frk = h3(c.seed, 42, cam, p, t);
  for (int j = 0; j < ncp; ++j)
    if (c.corner_drop_prob <= 0.0 || u01(mix64(frk ^ (uint64_t)(j + 1))) >= c.corner_drop_prob) out[m++] = j;
  if (c.k_select > 0 && m > c.k_select) {
    // keep k corners spread over the present ones (stand-in for the k-means pick of
    // multicamera.cpp:1060-1172, which is randomised in the reference)
    const int k = c.k_select;
    for (int i = 0; i < k; ++i) out[i] = out[(int)(((int64_t)i * m + m / 2) / k) % m];
    m = k;
  }
  return m;
}

}  // namespace

extern "C" {

int calico_synth_preset(int32_t config_id, calico_synth_config* c) {
  if (!c) return -1;
  std::memset(c, 0, sizeof *c);
  c->square_mm = 45.0;
  c->noise_px = 0.1;
  c->perturb_rot_deg = 1.0;
  c->perturb_trans_mm = 5.0;
  c->seed = 0xCA11C0ull + (uint64_t)config_id;
  switch (config_id) {
    case 1:  // A: 8 cameras x 2 ChArUco 7x9 boards x 200 times (BASELINE.json configs[1])
      c->cn = 8; c->pn = 2; c->tn = 200; c->board_type = 0; c->sx = 7; c->sy = 9;
      c->fr_drop_prob = 0.15; c->corner_drop_prob = 0.1;
      break;
    case 2:  // B: 32 cameras x 4 AprilTag 6x6 boards x 2000 times (configs[2])
      c->cn = 32; c->pn = 4; c->tn = 2000; c->board_type = 1; c->sx = 6; c->sy = 6;
      c->tangential = 1; c->one_board_per_cam_time = 1;
      break;
    case 3:  // C: 64 non-overlapping cameras x 6 boards 7x9 x 5000 times (configs[3])
      c->cn = 64; c->pn = 6; c->tn = 5000; c->board_type = 0; c->sx = 7; c->sy = 9;
      c->one_board_per_cam_time = 1;
      break;
    case 4:  // D: 128 cameras x 8 boards 9x11 x 10000 times, ~102.4M observations (configs[4])
      c->cn = 128; c->pn = 8; c->tn = 10000; c->board_type = 0; c->sx = 9; c->sy = 11;
      c->one_board_per_cam_time = 1;
      break;
    default:
      return -1;
  }
  return 0;
}

int32_t calico_synth_corners_per_pattern(const calico_synth_config* c) { return ncp_of(*c); }

void calico_synth_points(const calico_synth_config* c, float* xyz /*3*pn*ncp*/) {
  const int ncp = ncp_of(*c);
  for (int p = 0; p < c->pn; ++p)
    for (int j = 0; j < ncp; ++j) model_point(*c, j, xyz + 3 * ((size_t)p * ncp + j));
}

void calico_synth_intrinsics(const calico_synth_config* c, double* k /*12*cn*/) {
  for (int cam = 0; cam < c->cn; ++cam) intrinsics_of(*c, cam, k + 12 * cam);
}

void calico_synth_ground_truth(const calico_synth_config* c, double* x /*7*vn*/) {
  auto put = [&](int v, const Pose& P) {
    for (int k = 0; k < 4; ++k) x[7 * v + k] = P.q[k];
    for (int k = 0; k < 3; ++k) x[7 * v + 4 + k] = P.t[k];
  };
  for (int i = 0; i < c->cn; ++i) put(i, camera_pose(*c, i));
  for (int p = 0; p < c->pn; ++p) put(c->cn + p, pattern_pose(*c, p));
  for (int t = 0; t < c->tn; ++t) put(c->cn + c->pn + t, time_pose(*c, t));
}

// ground truth with every non-exemplar block perturbed by a rotation of
// perturb_rot_deg about a random axis and a translation of perturb_trans_mm.
void calico_synth_initial_guess(const calico_synth_config* c, double* x) {
  calico_synth_ground_truth(c, x);
  const int vn = c->cn + c->pn + c->tn;
  const int ps = c->cn + c->p_star_idx, ts = c->cn + c->pn + c->t_star_idx;
  for (int v = 0; v < vn; ++v) {
    if (v == ps || v == ts) continue;
    double ax[3], dq[4], q[4];
    rand_axis(c->seed, 51, v, ax);
    axis_angle(ax, c->perturb_rot_deg * 0.017453292519943295, dq);
    qmul(dq, x + 7 * v, q);
    for (int k = 0; k < 4; ++k) x[7 * v + k] = q[k];
    double d[3];
    rand_axis(c->seed, 52, v, d);
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; ++k) x[7 * v + 4 + k] += c->perturb_trans_mm * d[k] / n;
  }
}

// Observations with time superscript in [t0, t1).  Pass cam == NULL to count only.
// Returns the number of observations (or -1 if capacity is too small).
int64_t calico_synth_observations(const calico_synth_config* c, int32_t t0, int32_t t1, int32_t* cam,
                                  int32_t* pat_var, int32_t* time_var, int32_t* corner, double* u,
                                  double* v, int64_t capacity) {
  const int ncp = ncp_of(*c);
  std::vector<Pose> Cs(c->cn), Ps(c->pn);
  std::vector<double> Ks((size_t)12 * c->cn);
  std::vector<float> pts((size_t)3 * ncp);
  for (int i = 0; i < c->cn; ++i) {
    Cs[i] = camera_pose(*c, i);
    intrinsics_of(*c, i, &Ks[12 * i]);
  }
  for (int p = 0; p < c->pn; ++p) Ps[p] = pattern_pose(*c, p);
  for (int j = 0; j < ncp; ++j) model_point(*c, j, &pts[3 * j]);

  // pass 1: per-time counts (parallel), pass 2: fill at prefix offsets (parallel)
  const int nt = t1 - t0;
  std::vector<int64_t> off((size_t)nt + 1, 0);
#pragma omp parallel
  {
    std::vector<int> idx(ncp);
#pragma omp for schedule(static)
    for (int tt = 0; tt < nt; ++tt) {
      int64_t m = 0;
      for (int i = 0; i < c->cn; ++i)
        for (int p = 0; p < c->pn; ++p)
          if (fr_present(*c, i, p, t0 + tt)) m += fr_corners(*c, i, p, t0 + tt, ncp, idx.data());
      off[tt + 1] = m;
    }
  }
  for (int tt = 0; tt < nt; ++tt) off[tt + 1] += off[tt];
  const int64_t total = off[nt];
  if (!cam) return total;
  if (total > capacity) return -1;
#pragma omp parallel
  {
    std::vector<int> idx(ncp);
#pragma omp for schedule(static)
    for (int tt = 0; tt < nt; ++tt) {
      const int t = t0 + tt;
      const Pose T = time_pose(*c, t);
      int64_t o = off[tt];
      for (int i = 0; i < c->cn; ++i)
        for (int p = 0; p < c->pn; ++p) {
          if (!fr_present(*c, i, p, t)) continue;
          const int m = fr_corners(*c, i, p, t, ncp, idx.data());
          for (int q = 0; q < m; ++q, ++o) {
            const int j = idx[q];
            double uv[2];
            project(Cs[i], T, Ps[p], &Ks[12 * i], &pts[3 * j], uv);
            const uint64_t hk = h3(c->seed, 61, (uint64_t)i * c->pn + p, t, j);
            // detections are float32 in the reference (camera-calibration.cpp:257-258)
            cam[o] = i;
            pat_var[o] = c->cn + p;
            time_var[o] = c->cn + c->pn + t;
            corner[o] = p * ncp + j;
            u[o] = (double)(float)(uv[0] + c->noise_px * gauss(hk));
            v[o] = (double)(float)(uv[1] + c->noise_px * gauss(mix64(hk ^ 0x5555ull)));
          }
        }
    }
  }
  return total;
}

// Observation count per time superscript (input of calico_rp_partition_times).
void calico_synth_obs_per_time(const calico_synth_config* c, int64_t* out /*tn*/) {
  const int ncp = ncp_of(*c);
#pragma omp parallel
  {
    std::vector<int> idx(ncp);
#pragma omp for schedule(static)
    for (int t = 0; t < c->tn; ++t) {
      int64_t m = 0;
      for (int i = 0; i < c->cn; ++i)
        for (int p = 0; p < c->pn; ++p)
          if (fr_present(*c, i, p, t)) m += fr_corners(*c, i, p, t, ncp, idx.data());
      out[t] = m;
    }
  }
}

}  // extern "C"
