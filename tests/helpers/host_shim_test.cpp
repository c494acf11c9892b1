// Drives calico_amd/host/ceres_problem_class.hpp the way the reference drives its
// CeresProblemClass: the Stage-5 loop of /root/reference/src/calico.cpp:703-731,
// re-typed here (batch size = round(perc_rp * #equations), 10 iterations per solve)
// against small stand-ins for MCcali / CameraCali / PatternsCreated that expose only
// the members the path reads.  Data comes from the synthetic generator.
//   host_shim_test --selfcheck        CPU: pose <-> 7-vector conversions and inverse
//   host_shim_test --run <tn>         GPU: config-A-like network with tn times
//   host_shim_test --run-ae <tn>      GPU: the Stage-4 loop (calico.cpp:637-697) on the same network
//   host_shim_test --run-rae <tn>     GPU: ReconstructXFunctionIDsMCwithDLTs at the ground truth
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <vector>
#include <unistd.h>

#include "../../calico_amd/host/ceres_problem_class.hpp"
#include "../../calico_amd/host/points_txt.hpp"
#include "../../calico_amd/host/results_io.hpp"
#include "../../calico_amd/host/assess.hpp"
#include "../../calico_amd/host/reconstruct_x.hpp"
#include <sstream>
#include <iomanip>

extern "C" {
typedef struct calico_synth_config {
  int32_t cn, pn, tn, board_type, sx, sy;
  double square_mm;
  int32_t one_board_per_cam_time;
  double fr_drop_prob, corner_drop_prob;
  int32_t k_select;
  double noise_px, perturb_rot_deg, perturb_trans_mm;
  int32_t tangential, p_star_idx, t_star_idx;
  uint64_t seed;
} calico_synth_config;
int calico_synth_preset(int32_t, calico_synth_config*);
int32_t calico_synth_corners_per_pattern(const calico_synth_config*);
void calico_synth_points(const calico_synth_config*, float*);
void calico_synth_intrinsics(const calico_synth_config*, double*);
void calico_synth_initial_guess(const calico_synth_config*, double*);
void calico_synth_ground_truth(const calico_synth_config*, double*);
int64_t calico_synth_observations(const calico_synth_config*, int32_t, int32_t, int32_t*, int32_t*, int32_t*,
                                  int32_t*, double*, double*, int64_t);
}

// ---- stand-ins (member names as in the reference: multicamera.hpp, camera-calibration.hpp, patterns.hpp)
struct Mat4 {
  double m[16];
  double& operator()(int i, int j) { return m[4 * i + j]; }
  double operator()(int i, int j) const { return m[4 * i + j]; }
};
struct Point3f { float x, y, z; };
struct PatternsCreated {
  std::vector<Point3f> three_d_points;
  int ncp = 0;
  int NumberCornersPerPattern() const { return ncp; }
};
struct Dense2 {  // MatrixXd stand-in: rows x 2
  std::vector<double> d;
  double operator()(int j, int c) const { return d[2 * j + c]; }
};
struct Mat3 {
  double m[9];
  double operator()(int i, int j) const { return m[3 * i + j]; }
  double& operator()(int i, int j) { return m[3 * i + j]; }
};
struct Dist {  // VectorXd stand-in
  std::vector<double> d;
  long rows() const { return (long)d.size(); }
  double operator()(int j) const { return d[j]; }
  double& operator()(int j) { return d[j]; }
  void conservativeResize(long n) { d.resize((size_t)n); }
};
struct Vec3 {
  double d[3] = {0, 0, 0};
  double& operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
};
struct CameraCali {
  int start_time_this_camera = 0;
  int rows = 900, cols = 1200;
  PatternsCreated* P_class = nullptr;
  std::vector<std::vector<bool>> points_used_min, points_present;
  std::vector<Dense2> two_d_point_coordinates_dense;
  Mat3 internal_parameters;
  Dist distortion;
};
struct single_relationship_container {
  int lhs;
  std::vector<int> rhs;
};
struct MCcali {
  int cn = 0, pn = 0, tn = 0, ps = 0, ts = 0;
  std::vector<Mat4> V_initial;
  std::vector<int> V_index, pattern_a_vars, time_f_vars;
  std::vector<single_relationship_container> singles;
  std::vector<Mat4> A;               // multicamera.hpp: the measured camera-pattern transform of each FR
  std::vector<bool> singles_open;    // true while a variable of the equation is uninitialised
  std::vector<bool> var_initialised;  // stand-in for V_has_initialization
  void UpdateSinglesOpenFlag() {
    singles_open.assign(singles.size(), true);
    for (size_t e = 0; e < singles.size(); ++e)
      singles_open[e] = !(var_initialised[singles[e].lhs] && var_initialised[singles[e].rhs[1]] &&
                          var_initialised[singles[e].rhs[2]]);
  }
  // what the assessment / report functions fill and print (multicamera.hpp:31-200)
  int an = 0;
  std::vector<std::vector<Mat4>> V_progressive_solutions;
  std::vector<bool> V_has_initialization;
  bool AllInitialized(const std::vector<int>& side) const {
    for (size_t k = 1; k < side.size(); ++k)
      if (!V_has_initialization[side[k]]) return false;
    return true;
  }
  std::vector<int> type_recorder;
  std::vector<std::string> descriptor_recorder;
  std::vector<double> summed_c1_cost_function_error_by_type;
  std::vector<std::vector<double>> cost_function_error_by_type_c1, reprojection_error_by_term_and_type;
  std::vector<Vec3> reconstructed_points;
  std::vector<bool> valid_reconstructed_points;
  int number_valid_points_rae_strict = 0, number_valid_points_rae_multi = 0;
  std::vector<double> average_rae_ba, stddev_rae_ba, median_rae_ba, average_rae_multi, stddev_rae_multi, median_rae_multi;
  int NumberVariables() const { return cn + pn + tn; }
  int NumberSingles() const { return (int)singles.size(); }
  int NumberCameras() const { return cn; }
  int NumberPatterns() const { return pn; }
  int p_star() const { return ps; }
  int t_star() const { return ts; }
  // multicamera.cpp:1493-1514 (asserts are live in the reference build: no -DNDEBUG)
  int identity_checks = 0;
  void TestSolutionHasIdentityPstarTstar(const std::vector<Mat4>& vector_to_use) {
    for (int var : {pattern_a_vars[p_star()], time_f_vars[t_star()]}) {
      double norm2 = 0.0;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const double d = vector_to_use[var](i, j) - (i == j ? 1.0 : 0.0);
          norm2 += d * d;
        }
      if (!(std::sqrt(norm2) < 1e-20)) {
        std::printf("TestSolutionHasIdentityPstarTstar failed for variable %d\n", var);
        std::abort();
      }
    }
    ++identity_checks;
  }
};

// points.txt round trip: write the synthetic corners of one camera the way
// camera-calibration.cpp:283-294 does, read them back, rebuild the SoA observation arrays.
static int points_txt_check() {
  calico_synth_config cfg;
  calico_synth_preset(1, &cfg);
  cfg.tn = 6;
  const int ncp = calico_synth_corners_per_pattern(&cfg), cn = cfg.cn, pn = cfg.pn, tn = cfg.tn;
  const int64_t N = calico_synth_observations(&cfg, 0, tn, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  std::vector<int32_t> cam(N), pat(N), tim(N), cor(N);
  std::vector<double> u(N), v(N);
  calico_synth_observations(&cfg, 0, tn, cam.data(), pat.data(), tim.data(), cor.data(), u.data(), v.data(), N);
  std::vector<int> pattern_a_vars, time_f_vars;
  for (int p = 0; p < pn; ++p) pattern_a_vars.push_back(cn + p);
  for (int t = 0; t < tn; ++t) time_f_vars.push_back(cn + pn + t);
  int bad = 0;
  for (int c = 0; c < cn; c += 3) {
    const std::string path = "/tmp/calico_points_txt_check_" + std::to_string(c) + ".txt";
    {
      std::ofstream out(path);
      for (int i = 0; i < tn; ++i) {
        out << "image " << i << std::endl;
        for (int64_t k = 0; k < N; ++k)
          if (cam[k] == c && tim[k] == cn + pn + i)
            out << cor[k] << " " << std::setprecision(9) << u[k] << " " << std::setprecision(9) << v[k] << " ";
        out << std::endl << "end " << i << std::endl;
      }
      out << "Image-board truth table " << std::endl << "Image 0 : 1 0" << std::endl;
    }
    std::vector<calico_amd::PointsTxtImage> images;
    std::string err;
    if (!calico_amd::ReadPointsTxt(path, images, &err) || (int)images.size() != tn) {
      std::printf("points.txt: %s\n", err.c_str());
      return 1;
    }
    std::vector<int32_t> c2, p2, t2, j2;
    std::vector<double> u2, v2;
    calico_amd::AppendCameraObservations(images, c, 0, ncp, pattern_a_vars, time_f_vars, [](int, int) { return true; },
                                         c2, p2, t2, j2, u2, v2);
    // the generator's order within a camera is (time, pattern, corner) = the file's order
    size_t m = 0;
    for (int i = 0; i < tn; ++i)
      for (int64_t k = 0; k < N; ++k)
        if (cam[k] == c && tim[k] == cn + pn + i) {
          if (m >= c2.size() || c2[m] != c || p2[m] != pat[k] || t2[m] != tim[k] || j2[m] != cor[k] ||
              std::fabs(u2[m] - u[k]) > 1e-8 * std::fabs(u[k]) || std::fabs(v2[m] - v[k]) > 1e-8 * std::fabs(v[k]))
            ++bad;
          ++m;
        }
    if (m != c2.size() || m == 0) ++bad;
    std::remove(path.c_str());
  }
  return bad;
}

// result files and the pose metric (results_io.hpp) on small known inputs
static int results_io_check() {
  int bad = 0;
  auto rotz = [](double deg, double tx, double ty, double tz) {
    Mat4 M;
    const double a = deg * M_PI / 180.0;
    const double v[16] = {std::cos(a), 0.0 - std::sin(a), 0, tx, std::sin(a), std::cos(a), 0, ty, 0, 0, 1, tz, 0, 0, 0, 1};
    std::memcpy(M.m, v, sizeof v);
    return M;
  };
  // cameras.txt: "2\n name K R t dist \n ..." with default formatting
  CameraCali c0, c1;
  c0.internal_parameters = Mat3{{1400.5, 0, 600.25, 0, 1399.75, 450, 0, 0, 1}};
  c0.distortion.d = {-0.0123456789, 0.02, 0, 0, 0};
  c1 = c0;
  c1.internal_parameters.m[0] = 1234.5678;
  std::vector<CameraCali*> CCV = {&c0, &c1};
  std::vector<Mat4> ext = {rotz(0, 0, 0, 0), rotz(10, 100.5, -20.25, 3)};
  const std::string f1 = "/tmp/calico_results_io_check_cameras.txt";
  if (!calico_amd::WriteCameraCalibrationResult(CCV, {"cam0", "cam1"}, ext, f1)) ++bad;
  {
    std::ifstream in(f1);
    int n = 0;
    in >> n;
    if (n != 2) ++bad;
    for (int i = 0; i < 2; ++i) {
      std::string name;
      in >> name;
      if (name != (i ? "cam1" : "cam0")) ++bad;
      double K[9], R[9], t[3], d[5];
      for (double& k : K) in >> k;
      for (double& k : R) in >> k;
      for (double& k : t) in >> k;
      for (double& k : d) in >> k;
      if (!in.good()) ++bad;
      if (std::fabs(K[0] - CCV[i]->internal_parameters(0, 0)) > 1e-5 * K[0] || K[2] != 600.25 || K[8] != 1) ++bad;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          if (std::fabs(R[3 * r + c] - ext[i](r, c)) > 1e-6) ++bad;
        if (std::fabs(t[r] - ext[i](r, 3)) > 1e-5 * (1 + std::fabs(t[r]))) ++bad;
      }
      if (std::fabs(d[0] + 0.0123457) > 1e-9) ++bad;  // 6 significant digits, as `out << double` prints
    }
    std::string rest;
    if (in >> rest) ++bad;
  }
  std::remove(f1.c_str());
  if (bad) std::printf("results_io: cameras file check failed (%d)\n", bad);
  // variables.txt: headers and Eigen-style aligned 4x4 blocks
  const std::string f2 = "/tmp/calico_results_io_check_variables.txt";
  std::vector<Mat4> V = {rotz(0, 0, 0, 0), rotz(0, 0, 0, 0), rotz(90, 1, 2, 3)};
  if (!calico_amd::WriteVariablesTxt(f2, 1, 1, 1, V)) ++bad;
  {
    std::ifstream in(f2);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string want_head = "C 0\n1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\nP 0, 1\n1 0 0 0\n";
    if (ss.str().compare(0, want_head.size(), want_head) != 0) ++bad;
    if (ss.str().find("T 0, 2\n") == std::string::npos) ++bad;
    // aligned columns: every row of the last block has the same length
    const size_t p = ss.str().find("T 0, 2\n") + 7;
    std::istringstream blk(ss.str().substr(p));
    std::string line;
    size_t len = 0;
    for (int r = 0; r < 4 && std::getline(blk, line); ++r) {
      if (r == 0) len = line.size();
      if (line.size() != len || len == 0) ++bad;
    }
  }
  std::remove(f2.c_str());
  if (bad) std::printf("results_io: after variables.txt check (%d)\n", bad);
  // pose metric: camera 1 off by 10 degrees about z and (3, 4, 0): 10 / (2-1) degrees, 5 units
  std::vector<Mat4> gt = {rotz(30, 5, 6, 7), rotz(50, 1, 2, 3)};
  std::vector<Mat4> gt_rel = calico_amd::RelativeExternals(gt);
  std::vector<Mat4> est_rel = gt_rel;
  {
    Mat4 d = rotz(10, 0, 0, 0), m = gt_rel[1];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double sacc = 0;
        for (int k = 0; k < 3; ++k) sacc += d(i, k) * gt_rel[1](k, j);
        m(i, j) = sacc;
      }
    m(0, 3) += 3;
    m(1, 3) += 4;
    est_rel[1] = m;
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (std::fabs(gt_rel[0](i, j) - (i == j)) > 1e-12) ++bad;  // camera 0 relative to itself
  double te = 0, re = 0;
  calico_amd::CameraPoseError(gt_rel, est_rel, &te, &re);
  if (std::fabs(te - 5.0) > 1e-12 || std::fabs(re - 10.0) > 1e-10) {
    std::printf("results_io: pose metric %.17g %.17g\n", te, re);
    ++bad;
  }
  return bad;
}

// CopyToCalibration is the inverse of CopyFromCalibration (solving-structure.cpp:134-159)
static int copy_calibration_check() {
  int bad = 0;
  CameraCali c0, c1;
  c0.internal_parameters = Mat3{{1400.5, 0.25, 600.25, 0, 1399.75, 450, 0, 0, 1}};  // skew 0.25: dropped by the reference
  c0.distortion.d = {-0.0123456789, 0.02, 1e-4, -2e-4, 0.003};
  c1 = c0;
  c1.internal_parameters.m[0] = 1234.5678;
  std::vector<CameraCali*> CCV = {&c0, &c1};
  double p[24], back[24];
  calico_amd::CopyFromCalibration(CCV, p);
  if (p[0] != 1400.5 || p[1] != 600.25 || p[2] != 1399.75 || p[3] != 450 || p[8] != 0.003 || p[9] != 0 || p[12] != 1234.5678) ++bad;
  // round trip leaves everything as it was
  calico_amd::CopyToCalibration(p, CCV);
  if (c0.internal_parameters(0, 1) != 0.25 || c0.distortion.rows() != 5 || c0.distortion(0) != -0.0123456789) ++bad;
  // refined values arrive; k4 of camera 1 (freed by a full mask) grows its vector, camera 0 keeps five
  for (int k = 0; k < 9; ++k) { p[k] += 0.5 + k; p[12 + k] -= 0.25 * k; }
  p[12 + 9] = 0.004;
  calico_amd::CopyToCalibration(p, CCV);
  calico_amd::CopyFromCalibration(CCV, back);
  for (int k = 0; k < 24; ++k)
    if (back[k] != p[k]) ++bad;
  if (c0.distortion.rows() != 5 || c1.distortion.rows() != 6 || c1.distortion(5) != 0.004) ++bad;
  if (c0.internal_parameters(0, 1) != 0.25 || c0.internal_parameters(2, 2) != 1) ++bad;
  if (bad) std::printf("CopyToCalibration check failed (%d)\n", bad);
  return bad;
}

static int selfcheck() {
  int bad = 0;
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / double(1 << 24) - 0.5; };
  for (int it = 0; it < 200; ++it) {
    double q[7] = {rnd(), rnd(), rnd(), rnd(), 100 * rnd(), 100 * rnd(), 100 * rnd()};
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) q[k] /= n;
    Mat4 M;
    calico_amd::Convert7ParameterQuaternionRepresentationIntoMatrix(q, M);
    double back[7];
    calico_amd::ConvertMatrixTo7QuaternionRepresentationx(M, back);
    double sign = (back[0] * q[0] + back[1] * q[1] + back[2] * q[2] + back[3] * q[3]) < 0 ? -1 : 1;
    for (int k = 0; k < 4; ++k) if (std::fabs(sign * back[k] - q[k]) > 1e-12) ++bad;
    for (int k = 4; k < 7; ++k) if (back[k] != q[k]) ++bad;
    Mat4 Mi = calico_amd::Inverse4(M);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += M(i, k) * Mi(k, j);
        if (std::fabs(acc - (i == j)) > 1e-10) ++bad;
      }
  }
  bad += points_txt_check();
  bad += results_io_check();
  bad += copy_calibration_check();
  std::printf("selfcheck %s\n", bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}

static Mat4 mul4(const Mat4& a, const Mat4& b) {
  Mat4 c;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a(i, k) * b(k, j);
      c(i, j) = s;
    }
  return c;
}

// Stage 4 through the mirror class: variables become "solved" in three waves (cameras and
// patterns, then each half of the times; stands in for IterativelySolveForVariables2), with
// an AE solve after each wave and the cleanup solve of calico.cpp:694-697.
static int run_ae(int tn_arg) {
  calico_synth_config cfg;
  calico_synth_preset(1, &cfg);
  cfg.tn = tn_arg;
  const int cn = cfg.cn, pn = cfg.pn, tn = cfg.tn, vn = cn + pn + tn;
  const int64_t N = calico_synth_observations(&cfg, 0, tn, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  std::vector<int32_t> cam(N), pat(N), tim(N), cor(N);
  std::vector<double> u(N), v(N);
  calico_synth_observations(&cfg, 0, tn, cam.data(), pat.data(), tim.data(), cor.data(), u.data(), v.data(), N);
  MCcali MC;
  MC.cn = cn; MC.pn = pn; MC.tn = tn; MC.ps = cfg.p_star_idx; MC.ts = cfg.t_star_idx;
  for (int p = 0; p < pn; ++p) MC.pattern_a_vars.push_back(cn + p);
  for (int t = 0; t < tn; ++t) MC.time_f_vars.push_back(cn + pn + t);
  std::vector<double> xg(7 * (size_t)vn), x0(7 * (size_t)vn);
  calico_synth_ground_truth(&cfg, xg.data());
  calico_synth_initial_guess(&cfg, x0.data());
  // generator 7-vectors hold patterns/times inverted (Stage-5 convention): un-invert
  std::vector<Mat4> Vg(vn);
  MC.V_initial.resize(vn);
  for (int vv = 0; vv < vn; ++vv) {
    Mat4 M;
    calico_amd::Convert7ParameterQuaternionRepresentationIntoMatrix(&xg[7 * vv], M);
    Vg[vv] = (vv < cn) ? M : calico_amd::Inverse4(M);
    calico_amd::Convert7ParameterQuaternionRepresentationIntoMatrix(&x0[7 * vv], M);
    MC.V_initial[vv] = (vv < cn) ? M : calico_amd::Inverse4(M);
  }
  std::map<long long, int> fr_id;
  unsigned s = 99;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / double(1 << 24) - 0.5; };
  for (int64_t i = 0; i < N; ++i) {
    const long long key = ((long long)tim[i] * 4096 + cam[i]) * 4096 + pat[i];
    if (fr_id.count(key)) continue;
    fr_id[key] = (int)MC.singles.size();
    MC.singles.push_back({cam[i], {-1, pat[i], tim[i]}});
    // C = A P T  =>  A = C (P T)^-1, measured with a little noise on the translation (mm)
    Mat4 A = mul4(Vg[cam[i]], calico_amd::Inverse4(mul4(Vg[pat[i]], Vg[tim[i]])));
    for (int r = 0; r < 3; ++r) A(r, 3) += 0.5 * rnd();
    MC.A.push_back(A);
  }
  MC.var_initialised.assign(vn, false);
  const int p_star_var = MC.pattern_a_vars[MC.p_star()], t_star_var = MC.time_f_vars[MC.t_star()];
  MC.var_initialised[p_star_var] = MC.var_initialised[t_star_var] = true;

  std::ofstream ceres_out("/dev/null"), out("/dev/null");
  calico_amd::CeresProblemClass<MCcali, CameraCali> CPC(calico_amd::Cali_Quaternion, MC, ceres_out);
  std::vector<int> variable_order, equation_order;
  const int number_iters_local = 10;  // calico.cpp:626
  const int t_half = cn + pn + tn / 2;
  int total = 0;
  for (int wave = 0; wave < 3; ++wave) {
    int var_accumulator = 0;
    const int lo = wave == 0 ? 0 : (wave == 1 ? cn + pn : t_half), hi = wave == 0 ? cn + pn : (wave == 1 ? t_half : vn);
    for (int vv = lo; vv < hi; ++vv)
      if (vv != p_star_var && vv != t_star_var) {
        MC.var_initialised[vv] = true;
        variable_order.push_back(vv);
        ++var_accumulator;
      }
    const int num_equations = CPC.AddToProblemAlgebraicError(MC, variable_order, equation_order, out, var_accumulator);
    total += num_equations;
    if (total == 0) continue;  // cameras and patterns alone: only FRs at t* are closed
    CPC.SolveWriteBackToMCAlgebraicError(MC, ceres_out, number_iters_local, false);
    std::printf("AE_SOLVE added=%d relations=%lld iterations=%d initial=%.17g final=%.17g\n", num_equations,
                (long long)CPC.summary.num_residual_blocks, CPC.summary.num_iterations, CPC.summary.initial_cost,
                CPC.summary.final_cost);
  }
  if (total != (int)MC.singles.size() || equation_order.size() != MC.singles.size()) {
    std::printf("NOT ALL EQUATIONS ADDED\n");
    return 1;
  }
  for (int vv : {p_star_var, t_star_var})
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (std::fabs(MC.V_initial[vv](i, j) - (i == j)) > 1e-20) { std::printf("EXEMPLAR MOVED\n"); return 1; }
  // distance of the refined poses from the truth (the measured A carry 0.25 mm of noise)
  double worst_r = 0, worst_t = 0;
  for (int vv = 0; vv < vn; ++vv)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        const double d = std::fabs(MC.V_initial[vv](i, j) - Vg[vv](i, j));
        (j < 3 ? worst_r : worst_t) = std::max(j < 3 ? worst_r : worst_t, d);
      }
  std::printf("AE_TRUTH rot=%.3e trans=%.3e\n", worst_r, worst_t);
  return 0;
}

// ReconstructXFunctionIDsMCwithDLTs through the mirror function, variables at the ground
// truth (calico.cpp calls it after the refinements): prints the corners.
static int run_rae(int tn_arg) {
  calico_synth_config cfg;
  calico_synth_preset(1, &cfg);
  cfg.tn = tn_arg;
  const int cn = cfg.cn, pn = cfg.pn, tn = cfg.tn, vn = cn + pn + tn;
  const int ncp = calico_synth_corners_per_pattern(&cfg);
  const int64_t N = calico_synth_observations(&cfg, 0, tn, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  std::vector<int32_t> cam(N), pat(N), tim(N), cor(N);
  std::vector<double> u(N), v(N);
  calico_synth_observations(&cfg, 0, tn, cam.data(), pat.data(), tim.data(), cor.data(), u.data(), v.data(), N);
  PatternsCreated P;
  P.ncp = ncp;
  std::vector<float> xyz(3 * (size_t)pn * ncp);
  calico_synth_points(&cfg, xyz.data());
  for (int j = 0; j < pn * ncp; ++j) P.three_d_points.push_back({xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]});
  std::vector<double> K(12 * cn);
  calico_synth_intrinsics(&cfg, K.data());
  std::vector<CameraCali*> CCV;
  for (int c = 0; c < cn; ++c) {
    auto* cc = new CameraCali;
    cc->P_class = &P;
    cc->points_present.assign(tn, std::vector<bool>(pn * ncp, false));
    cc->two_d_point_coordinates_dense.assign(tn, Dense2{std::vector<double>(2 * (size_t)pn * ncp, 0.0)});
    CCV.push_back(cc);
  }
  MCcali MC;
  MC.cn = cn; MC.pn = pn; MC.tn = tn; MC.ps = cfg.p_star_idx; MC.ts = cfg.t_star_idx;
  MC.V_index.resize(vn);
  for (int c = 0; c < cn; ++c) MC.V_index[c] = c;
  for (int p = 0; p < pn; ++p) MC.V_index[cn + p] = p;
  for (int t = 0; t < tn; ++t) MC.V_index[cn + pn + t] = t;
  std::map<long long, int> fr_id;
  for (int64_t i = 0; i < N; ++i) {
    const int t = tim[i] - cn - pn;
    CCV[cam[i]]->points_present[t][cor[i]] = true;
    CCV[cam[i]]->two_d_point_coordinates_dense[t].d[2 * cor[i]] = u[i];
    CCV[cam[i]]->two_d_point_coordinates_dense[t].d[2 * cor[i] + 1] = v[i];
    const long long key = ((long long)tim[i] * 4096 + cam[i]) * 4096 + pat[i];
    if (!fr_id.count(key)) {
      fr_id[key] = (int)MC.singles.size();
      MC.singles.push_back({cam[i], {-1, pat[i], tim[i]}});
    }
  }
  MC.singles_open.assign(MC.singles.size(), false);
  std::vector<double> xg(7 * (size_t)vn);
  calico_synth_ground_truth(&cfg, xg.data());
  std::vector<Mat4> vars(vn + 1);  // one temporary variable at the end, cpp:193
  for (int vv = 0; vv < vn; ++vv) {
    Mat4 M;
    calico_amd::Convert7ParameterQuaternionRepresentationIntoMatrix(&xg[7 * vv], M);
    vars[vv] = (vv < cn) ? M : calico_amd::Inverse4(M);
  }
  vars[vn] = vars[0];
  std::vector<Vec3> est;
  std::vector<bool> has_values;
  std::ofstream out("/dev/null");
  calico_amd::ReconstructXFunctionIDsMCwithDLTs(MC, vars, CCV, K.data(), est, has_values, out);
  for (int k = 0; k < pn * ncp; ++k)
    std::printf("RAE %d %d %.17g %.17g %.17g\n", k, (int)has_values[k], est[k](0), est[k](1), est[k](2));
  for (auto* c : CCV) delete c;
  return 0;
}

// Ingest -> GPU: every camera's corners written as camera-calibration.cpp:283-294
// writes points.txt, read back with points_txt.hpp, fed to a handle; the reduced normal equations
// must equal those of a handle fed with the generator's arrays directly.
static int run_points(int tn_arg) {
  calico_synth_config cfg;
  calico_synth_preset(1, &cfg);
  cfg.tn = tn_arg;
  const int ncp = calico_synth_corners_per_pattern(&cfg), cn = cfg.cn, pn = cfg.pn, tn = cfg.tn, vn = cn + pn + tn;
  const int64_t N = calico_synth_observations(&cfg, 0, tn, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  std::vector<int32_t> cam(N), pat(N), tim(N), cor(N);
  std::vector<double> u(N), v(N);
  calico_synth_observations(&cfg, 0, tn, cam.data(), pat.data(), tim.data(), cor.data(), u.data(), v.data(), N);
  std::vector<int> pattern_a_vars, time_f_vars;
  for (int p = 0; p < pn; ++p) pattern_a_vars.push_back(cn + p);
  for (int t = 0; t < tn; ++t) time_f_vars.push_back(cn + pn + t);
  std::vector<int32_t> c2, p2, t2, j2;
  std::vector<double> u2, v2;
  for (int c = 0; c < cn; ++c) {
    const std::string path = "/tmp/calico_run_points_" + std::to_string(c) + ".txt";
    {
      // per image: index -> (x, y) of the present corners, then the reference's own output statements
      std::vector<std::map<int, std::pair<double, double>>> present(tn);
      for (int64_t k = 0; k < N; ++k)
        if (cam[k] == c) present[tim[k] - cn - pn][cor[k]] = {u[k], v[k]};
      std::ofstream out(path);
      for (int i = 0; i < tn; ++i) {
        out << "image " << i << std::endl;
        for (int j = 0; j < pn * ncp; ++j)
          if (present[i].count(j))
            out << j << " " << std::setprecision(9) << present[i][j].first << " " << std::setprecision(9)
                << present[i][j].second << " ";
        out << std::endl << "end " << i << std::endl;
      }
      out << "Image-board truth table " << std::endl;
    }
    std::vector<calico_amd::PointsTxtImage> images;
    std::string err;
    if (!calico_amd::ReadPointsTxt(path, images, &err) || (int)images.size() != tn) {
      std::printf("points.txt: %s\n", err.c_str());
      return 1;
    }
    calico_amd::AppendCameraObservations(images, c, 0, ncp, pattern_a_vars, time_f_vars, [](int, int) { return true; }, c2, p2,
                                         t2, j2, u2, v2);
    std::remove(path.c_str());
  }
  if ((int64_t)c2.size() != N) {
    std::printf("points.txt: %lld observations read, %lld written\n", (long long)c2.size(), (long long)N);
    return 1;
  }
  std::vector<float> xyz(3 * (size_t)pn * ncp);
  calico_synth_points(&cfg, xyz.data());
  std::vector<double> K(12 * cn), x0(7 * (size_t)vn);
  calico_synth_intrinsics(&cfg, K.data());
  calico_synth_initial_guess(&cfg, x0.data());
  std::vector<std::vector<double>> S(2), rhs(2);
  int32_t n = 0;
  for (int which = 0; which < 2; ++which) {
    calico_rp_problem* h = nullptr;
    if (calico_rp_create(cn, pn, tn, cn + cfg.p_star_idx, cn + pn + cfg.t_star_idx, 0, &h)) return 1;
    calico_rp_set_intrinsics(h, K.data());
    calico_rp_set_points(h, xyz.data(), pn * ncp);
    int rc = which == 0 ? calico_rp_add_observations(h, N, c2.data(), p2.data(), t2.data(), j2.data(), u2.data(), v2.data())
                        : calico_rp_add_observations(h, N, cam.data(), pat.data(), tim.data(), cor.data(), u.data(), v.data());
    if (rc == 0) rc = calico_rp_normal_eq(h, x0.data(), 1e4, 1, &n, nullptr, nullptr);
    S[which].resize((size_t)n * n);
    rhs[which].resize(n);
    if (rc == 0) rc = calico_rp_normal_eq(h, x0.data(), 1e4, 1, &n, S[which].data(), rhs[which].data());
    if (rc) {
      std::printf("points.txt: %s\n", calico_rp_last_error(h));
      return 1;
    }
    calico_rp_destroy(h);
  }
  double smax = 0, sdiff = 0, rmax = 0, rdiff = 0;
  for (size_t k = 0; k < S[0].size(); ++k) {
    smax = std::max(smax, std::fabs(S[1][k]));
    sdiff = std::max(sdiff, std::fabs(S[0][k] - S[1][k]));
  }
  for (size_t k = 0; k < rhs[0].size(); ++k) {
    rmax = std::max(rmax, std::fabs(rhs[1][k]));
    rdiff = std::max(rdiff, std::fabs(rhs[0][k] - rhs[1][k]));
  }
  std::printf("POINTS n_obs=%lld n=%d S_rel=%.3e rhs_rel=%.3e\n", (long long)N, n, sdiff / smax, rdiff / rmax);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !std::strcmp(argv[1], "--run-points")) return run_points(std::atoi(argv[2]));
  if (argc >= 2 && !std::strcmp(argv[1], "--selfcheck")) return selfcheck();
  if (argc >= 3 && !std::strcmp(argv[1], "--run-rae")) return run_rae(std::atoi(argv[2]));
  if (argc >= 3 && !std::strcmp(argv[1], "--run-ae")) return run_ae(std::atoi(argv[2]));
  const bool report = argc >= 3 && !std::strcmp(argv[1], "--run-report");
  if (argc < 3 || (std::strcmp(argv[1], "--run") && !report)) {
    std::printf("usage: %s --selfcheck | --run <tn> [config 1..4 [refine mask]] | --run-report <tn> [config [refine mask]] | "
                "--run-ae <tn> | --run-rae <tn>\n", argv[0]);
    return 2;
  }
  calico_synth_config cfg;
  const int config_id = argc >= 4 ? std::atoi(argv[3]) : 1;  // 1 = A .. 4 = D (BASELINE.json configs[1..4])
  const unsigned refine_mask = argc >= 5 ? (unsigned)std::strtoul(argv[4], nullptr, 0) : 0u;
  if (calico_synth_preset(config_id, &cfg) != 0) return 2;
  cfg.tn = std::atoi(argv[2]);
  const int cn = cfg.cn, pn = cfg.pn, tn = cfg.tn, vn = cn + pn + tn;
  const int ncp = calico_synth_corners_per_pattern(&cfg);

  // observations -> the reference's per-camera, per-time dense arrays
  const int64_t N = calico_synth_observations(&cfg, 0, tn, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  std::vector<int32_t> cam(N), pat(N), tim(N), cor(N);
  std::vector<double> u(N), v(N);
  calico_synth_observations(&cfg, 0, tn, cam.data(), pat.data(), tim.data(), cor.data(), u.data(), v.data(), N);

  PatternsCreated P;
  P.ncp = ncp;
  std::vector<float> xyz(3 * (size_t)pn * ncp);
  calico_synth_points(&cfg, xyz.data());
  for (int j = 0; j < pn * ncp; ++j) P.three_d_points.push_back({xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]});
  std::vector<double> K(12 * cn);
  calico_synth_intrinsics(&cfg, K.data());
  if (refine_mask)  // a per-camera calibration that is a little off (the test repeats this rule)
    for (int c = 0; c < cn; ++c) {
      K[12 * c] *= 1.0 + 0.004 * ((c % 3) - 1);
      K[12 * c + 2] *= 1.0 - 0.003 * ((c % 2) ? 1 : -1);
      K[12 * c + 1] += 1.5 * ((c % 4) - 1.5);
      K[12 * c + 3] -= 1.0 * ((c % 5) - 2);
      K[12 * c + 4] += 0.004 * ((c % 3) - 1);
    }
  std::vector<CameraCali*> CCV;
  for (int c = 0; c < cn; ++c) {
    auto* cc = new CameraCali;
    cc->P_class = &P;
    cc->points_used_min.assign(tn, std::vector<bool>(pn * ncp, false));
    cc->points_present.assign(tn, std::vector<bool>(pn * ncp, false));
    cc->two_d_point_coordinates_dense.assign(tn, Dense2{std::vector<double>(2 * (size_t)pn * ncp, 0.0)});
    cc->internal_parameters = Mat3{{K[12 * c], 0, K[12 * c + 1], 0, K[12 * c + 2], K[12 * c + 3], 0, 0, 1}};
    cc->distortion.d = {K[12 * c + 4], K[12 * c + 5], K[12 * c + 6], K[12 * c + 7], K[12 * c + 8]};
    CCV.push_back(cc);
  }
  MCcali MC;
  MC.cn = cn; MC.pn = pn; MC.tn = tn; MC.ps = cfg.p_star_idx; MC.ts = cfg.t_star_idx;
  for (int p = 0; p < pn; ++p) MC.pattern_a_vars.push_back(cn + p);
  for (int t = 0; t < tn; ++t) MC.time_f_vars.push_back(cn + pn + t);
  MC.V_index.resize(vn);
  for (int c = 0; c < cn; ++c) MC.V_index[c] = c;
  for (int p = 0; p < pn; ++p) MC.V_index[cn + p] = p;
  for (int t = 0; t < tn; ++t) MC.V_index[cn + pn + t] = t;
  std::map<long long, int> fr_id;
  for (int64_t i = 0; i < N; ++i) {
    const int t = tim[i] - cn - pn;
    // report mode: every other detected corner is left out of the minimisation (stands in for the k-means
    // selection, multicamera.cpp:1060-1172), so that "all detected corners" and "corners used" differ
    CCV[cam[i]]->points_used_min[t][cor[i]] = !report || (cor[i] % 2 == 0);
    CCV[cam[i]]->points_present[t][cor[i]] = true;
    CCV[cam[i]]->two_d_point_coordinates_dense[t].d[2 * cor[i]] = u[i];
    CCV[cam[i]]->two_d_point_coordinates_dense[t].d[2 * cor[i] + 1] = v[i];
    const long long key = ((long long)tim[i] * 4096 + cam[i]) * 4096 + pat[i];
    if (!fr_id.count(key)) {
      fr_id[key] = (int)MC.singles.size();
      MC.singles.push_back({cam[i], {-1, pat[i], tim[i]}});
    }
  }
  // initial V: generator's x0 (7-vectors; patterns/times stored inverted) -> 4x4, P and T un-inverted
  std::vector<double> x0(7 * (size_t)vn);
  calico_synth_initial_guess(&cfg, x0.data());
  MC.V_initial.resize(vn);
  for (int vv = 0; vv < vn; ++vv) {
    Mat4 M;
    calico_amd::Convert7ParameterQuaternionRepresentationIntoMatrix(&x0[7 * vv], M);
    MC.V_initial[vv] = (vv < cn) ? M : calico_amd::Inverse4(M);
  }

  std::vector<double> camera_params(12 * cn);
  calico_amd::CopyFromCalibration(CCV, camera_params.data());
  std::ofstream ceres_out("/dev/null");
  std::vector<int> equation_order(MC.singles.size()), variable_order;
  for (size_t e = 0; e < equation_order.size(); ++e) equation_order[e] = (int)e;
  for (int vv = 0; vv < vn; ++vv)
    if (vv != MC.pattern_a_vars[MC.p_star()] && vv != MC.time_f_vars[MC.t_star()]) variable_order.push_back(vv);

  // ---- calico.cpp:637,707-729
  calico_amd::CeresProblemClass<MCcali, CameraCali> CPC(calico_amd::Cali_Quaternion, MC, ceres_out);
  const float percentage_global_rp = 0.5f;
  int number_equations = (int)equation_order.size();
  int number_equations_before_solve = int(round(percentage_global_rp * float(number_equations)));
  int number_iters_rp = 10;
  CPC.SetUpXForReprojectionError(MC);
  if (refine_mask) CPC.SetRefineIntrinsics(refine_mask);  // extension, not in calico.cpp
  int end_index = 0;
  for (int start_index = 0; start_index < number_equations; start_index += number_equations_before_solve) {
    end_index = std::min(start_index + number_equations_before_solve, number_equations);
    CPC.AddEqsToProblemReprojectionError(MC, CCV, camera_params.data(), start_index, end_index, equation_order);
    CPC.SolveWriteBackToMCRP(MC, ceres_out, number_iters_rp, variable_order, false);
    std::printf("SOLVE n_obs=%lld iterations=%d initial=%.17g final=%.17g\n",
                (long long)CPC.summary.num_residual_blocks, CPC.summary.num_iterations, CPC.summary.initial_cost,
                CPC.summary.final_cost);
  }
  // exemplars untouched (MCcali::TestSolutionHasIdentityPstarTstar, multicamera.cpp:1493-1514)
  for (int vv : {MC.pattern_a_vars[MC.p_star()], MC.time_f_vars[MC.t_star()]})
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (std::fabs(MC.V_initial[vv](i, j) - (i == j)) > 1e-20) { std::printf("EXEMPLAR MOVED\n"); return 1; }
  std::printf("X");
  for (int k = 0; k < 7 * vn; ++k) std::printf(" %.17g", CPC.x[k]);
  std::printf("\n");
  std::printf("K");
  for (int k = 0; k < 12 * cn; ++k) std::printf(" %.17g", camera_params[k]);
  std::printf("\n");
  std::printf("IDENTITY_CHECKS %d\n", MC.identity_checks);
  if (report) {
    // ---- calico.cpp:731-798: V_progressive_solutions, RAE, WriteSolutionAssessErrorII, OutputRunResults
    // measured A of every FR: C (P T)^-1 at the ground truth with a little translation noise (as --run-ae)
    std::vector<double> xg(7 * (size_t)vn);
    calico_synth_ground_truth(&cfg, xg.data());
    std::vector<Mat4> Vg(vn);
    for (int vv = 0; vv < vn; ++vv) {
      Mat4 M;
      calico_amd::Convert7ParameterQuaternionRepresentationIntoMatrix(&xg[7 * vv], M);
      Vg[vv] = (vv < cn) ? M : calico_amd::Inverse4(M);
    }
    unsigned sd = 99;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (sd >> 8) / double(1 << 24) - 0.5; };
    for (size_t e = 0; e < MC.singles.size(); ++e) {
      Mat4 A = mul4(Vg[MC.singles[e].lhs], calico_amd::Inverse4(mul4(Vg[MC.singles[e].rhs[1]], Vg[MC.singles[e].rhs[2]])));
      for (int r = 0; r < 3; ++r) A(r, 3) += 0.5 * rnd();
      MC.A.push_back(A);
    }
    MC.an = (int)MC.singles.size();
    MC.V_has_initialization.assign(vn, true);
    MC.singles_open.assign(MC.singles.size(), false);
    std::vector<Mat4> V = MC.V_initial;
    V.push_back(V[0]);  // "one temporary variable at the end" (solving-structure.cpp:193)
    MC.V_progressive_solutions.push_back(V);
    const std::string dir = "/tmp/calico_shim_report_" + std::to_string((long)getpid()) + "/";
    mkdir(dir.c_str(), 0700);
    std::vector<std::string> names;
    for (int c = 0; c < cn; ++c) names.push_back("cam" + std::to_string(c));
    std::ofstream rae_out((dir + "rae_detail_ceres.txt").c_str());
    calico_amd::ReconstructionAccuracyErrorAndWriteII(MC, dir, 2, CCV, camera_params.data(), rae_out);
    calico_amd::WriteSolutionAssessErrorII(MC, dir, names, CCV, 2, false, 0.f, 0.f);
    calico_amd::OutputRunResults(MC, dir + "total_results.txt");
    for (const char* f : {"total_results.txt", "EstimatedVsGroundTruth2.txt", "cameras-incremental/A_est.txt",
                          "cameras-incremental/variables.txt", "camera_cali_incremental.txt"}) {
      std::ifstream in((dir + f).c_str());
      std::string line;
      int k = 0;
      while (std::getline(in, line) && k++ < (std::strcmp(f, "total_results.txt") ? 12 : 100000))
        std::printf("FILE[%s] %s\n", f, line.c_str());
      std::remove((dir + f).c_str());
    }
    std::remove((dir + "rae_detail_ceres.txt").c_str());
    rmdir((dir + "cameras-incremental").c_str());
    rmdir(dir.c_str());
  }
  // What the outputs read is the CameraCali objects, not camera_params (multicamera.cpp:1538,1553):
  // their values after the loop, and the line WriteCameraCalibrationResult prints for each camera.
  for (int c = 0; c < cn; ++c) {
    std::printf("CCV %d %.17g %.17g %.17g %.17g", c, CCV[c]->internal_parameters(0, 0), CCV[c]->internal_parameters(0, 2),
                CCV[c]->internal_parameters(1, 1), CCV[c]->internal_parameters(1, 2));
    for (long j = 0; j < CCV[c]->distortion.rows(); ++j) std::printf(" %.17g", CCV[c]->distortion(j));
    std::printf("\n");
  }
  {
    std::vector<std::string> names;
    for (int c = 0; c < cn; ++c) names.push_back("cam" + std::to_string(c));
    std::vector<Mat4> ext(MC.V_initial.begin(), MC.V_initial.begin() + cn);
    const std::string f = "/tmp/calico_shim_cameras_" + std::to_string((long)getpid()) + ".txt";
    if (!calico_amd::WriteCameraCalibrationResult(CCV, names, ext, f)) { std::printf("RESULT FILE FAILED\n"); return 1; }
    std::ifstream in(f);
    std::string line;
    std::getline(in, line);
    std::printf("FILE_HEAD %s\n", line.c_str());
    while (std::getline(in, line)) std::printf("FILE %s\n", line.c_str());
    std::remove(f.c_str());
  }
  for (auto* c : CCV) delete c;
  return 0;
}
