// ceres_problem_class.hpp — host-side mirror of the reference's CeresProblemClass
// for the reprojection-error path (Stage 5) and the algebraic-error path (Stage 4), on
// top of the C ABI (include/calico_rp.h).
//
// The reference seam is the C++ class CeresProblemClass
// (/root/reference/src/solving-structure.hpp:611-641) as driven by the Stage-5 loop
// of /root/reference/src/calico.cpp:703-731.  The three methods below keep the
// reference's names, argument lists and meaning, so that loop compiles against this
// class unchanged; only their bodies differ (HIP instead of Ceres):
//
//   SetUpXForReprojectionError          solving-structure.cpp:422-460
//   AddEqsToProblemReprojectionError    solving-structure.cpp:463-669
//   SolveWriteBackToMCRP                solving-structure.cpp:967-1023
//   CopyFromCalibration (free)          solving-structure.cpp:134-159   (+ its inverse, CopyToCalibration)
// and, for the Stage-4 loop of calico.cpp:637-697,
//   AddToProblemAlgebraicError          solving-structure.cpp:673-915
//   SolveWriteBackToMCAlgebraicError    solving-structure.cpp:918-964
//
// The reference's MCcali / CameraCali / PatternsCreated drag in Eigen, OpenCV and the
// AprilTag libraries, none of which exist in this image, so the class is a template
// over the two types and only uses the members the reference's own bodies touch
// (listed at each use).  With the real types it is a drop-in; the
// tests instantiate it with small stand-ins (tests/helpers/host_shim_test.cpp).
//
// Error convention: as in the reference (cout + exit(1), e.g.
// solving-structure.cpp:621-624), a failing C-ABI call prints calico_rp_last_error()
// and exits; there is no CPU fallback.
#pragma once
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <ostream>
#include <vector>

#include "../../include/calico_rp.h"

namespace calico_amd {

enum PARAM_TYPE { Cali_Quaternion };  // solving-structure.hpp:30

// ceres::RotationMatrixToQuaternion (ceres/rotation.h; formula in SURVEY.md appendix)
// applied to the rotation part of M, then the translation
// (ConvertMatrixTo7QuaternionRepresentationx, solving-structure.hpp:89-110; the
// reference hands Ceres a column-major copy, i.e. R(i,j) below is M(i,j)).
template <class Mat4>
inline void ConvertMatrixTo7QuaternionRepresentationx(const Mat4& M, double* X) {
  const double R00 = M(0, 0), R01 = M(0, 1), R02 = M(0, 2);
  const double R10 = M(1, 0), R11 = M(1, 1), R12 = M(1, 2);
  const double R20 = M(2, 0), R21 = M(2, 1), R22 = M(2, 2);
  const double trace = R00 + R11 + R22;
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    X[0] = 0.5 * t;
    t = 0.5 / t;
    X[1] = (R21 - R12) * t;
    X[2] = (R02 - R20) * t;
    X[3] = (R10 - R01) * t;
  } else {
    const double R[3][3] = {{R00, R01, R02}, {R10, R11, R12}, {R20, R21, R22}};
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    X[i + 1] = 0.5 * t;
    t = 0.5 / t;
    X[0] = (R[k][j] - R[j][k]) * t;
    X[j + 1] = (R[j][i] + R[i][j]) * t;
    X[k + 1] = (R[k][i] + R[i][k]) * t;
  }
  X[4] = M(0, 3);
  X[5] = M(1, 3);
  X[6] = M(2, 3);
}

// Convert7ParameterQuaternionRepresentationIntoMatrix(const T*, Matrix4d&),
// solving-structure.hpp:60-87 (ceres::QuaternionToRotation normalises by |q|^2).
template <class Mat4>
inline void Convert7ParameterQuaternionRepresentationIntoMatrix(const double* X, Mat4& M) {
  const double a = X[0], b = X[1], c = X[2], d = X[3];
  const double inv = 1.0 / (a * a + b * b + c * c + d * d);
  M(0, 0) = (a * a + b * b - c * c - d * d) * inv;
  M(0, 1) = 2.0 * (b * c - a * d) * inv;
  M(0, 2) = 2.0 * (a * c + b * d) * inv;
  M(1, 0) = 2.0 * (a * d + b * c) * inv;
  M(1, 1) = (a * a - b * b + c * c - d * d) * inv;
  M(1, 2) = 2.0 * (c * d - a * b) * inv;
  M(2, 0) = 2.0 * (b * d - a * c) * inv;
  M(2, 1) = 2.0 * (a * b + c * d) * inv;
  M(2, 2) = (a * a - b * b - c * c + d * d) * inv;
  M(0, 3) = X[4];
  M(1, 3) = X[5];
  M(2, 3) = X[6];
  M(3, 0) = 0.0;
  M(3, 1) = 0.0;
  M(3, 2) = 0.0;
  M(3, 3) = 1.0;
}

// General 4x4 inverse (the reference calls Eigen's Matrix4d::inverse(),
// solving-structure.cpp:445,1013), Gauss-Jordan with partial pivoting.
template <class Mat4>
inline Mat4 Inverse4(const Mat4& M) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = M(i, j);
      a[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int p = c;
    for (int r = c + 1; r < 4; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
    if (p != c)
      for (int j = 0; j < 8; ++j) std::swap(a[p][j], a[c][j]);
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      if (f != 0.0)
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
    }
  }
  Mat4 R = M;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) R(i, j) = a[i][4 + j];
  return R;
}

// CopyFromCalibration, solving-structure.cpp:134-159.  Uses
// CCV[i]->internal_parameters(r,c), CCV[i]->distortion.rows(), CCV[i]->distortion(j).
template <class CameraCaliT>
inline void CopyFromCalibration(const std::vector<CameraCaliT*>& CCV, double* camera_params) {
  const int number_cameras = (int)CCV.size();
  for (int i = 0; i < number_cameras; i++) {
    for (int j = 0; j < 12; j++) camera_params[12 * i + j] = 0;
    camera_params[12 * i] = CCV[i]->internal_parameters(0, 0);
    camera_params[12 * i + 1] = CCV[i]->internal_parameters(0, 2);
    camera_params[12 * i + 2] = CCV[i]->internal_parameters(1, 1);
    camera_params[12 * i + 3] = CCV[i]->internal_parameters(1, 2);
    for (int j = 0; j < (int)CCV[i]->distortion.rows() && j < 8; j++)
      camera_params[12 * i + 4 + j] = CCV[i]->distortion(j);
  }
}

// CopyToCalibration: the inverse of CopyFromCalibration (solving-structure.cpp:134-159), for the
// refined camera parameters of the extension below.  The reference never needs it (its camera
// parameters are constants in Stage 5), but everything downstream reads the CameraCali objects,
// not camera_params: WriteCameraCalibrationResult prints CCV[i]->internal_parameters /
// distortion (multicamera.cpp:1538,1553) and the rrmse report projects with them
// (camera-calibration.cpp:926).  Skew K(0,1) and the last row of K are left as they are.  A
// distortion vector shorter than the last non-zero refined coefficient (cv::calibrateCamera
// hands over five, camera-calibration.cpp:619-623) grows to hold it
// (VectorXd::conservativeResize), so k4..k6 freed by the mask are not lost; otherwise its length,
// and with it the output file's format, is unchanged.
template <class CameraCaliT>
inline void CopyToCalibration(const double* camera_params, std::vector<CameraCaliT*>& CCV) {
  const int number_cameras = (int)CCV.size();
  for (int i = 0; i < number_cameras; i++) {
    const double* p = camera_params + 12 * i;
    CCV[i]->internal_parameters(0, 0) = p[0];
    CCV[i]->internal_parameters(0, 2) = p[1];
    CCV[i]->internal_parameters(1, 1) = p[2];
    CCV[i]->internal_parameters(1, 2) = p[3];
    int have = (int)CCV[i]->distortion.rows(), need = 0;
    for (int j = 0; j < 8; j++)
      if (p[4 + j] != 0.0) need = j + 1;
    if (need > have) {
      CCV[i]->distortion.conservativeResize(need);
      for (int j = have; j < need; j++) CCV[i]->distortion(j) = 0.0;
      have = need;
    }
    for (int j = 0; j < have && j < 8; j++) CCV[i]->distortion(j) = p[4 + j];
  }
}

template <class MCcaliT, class CameraCaliT>
class CeresProblemClass {
 public:
  PARAM_TYPE param_type = Cali_Quaternion;
  int number_variables = 0;
  int number_equations = 0;
  double* x = 0;  // 7 * number_variables, as in the reference (solving-structure.cpp:402)
  calico_rp_options options;  // Ceres defaults; max_num_iterations is set per solve
  calico_rp_summary summary;  // of the last solve

  // solving-structure.cpp:394-419.  Uses MC.NumberVariables(), NumberSingles(),
  // NumberCameras(), NumberPatterns(), pattern_a_vars[p_star()], time_f_vars[t_star()].
  CeresProblemClass(PARAM_TYPE parameter_type, MCcaliT& MC, std::ostream& out, int device = 0) {
    out << "Setting up the problem in CeresProblemClass. " << std::endl;
    param_type = parameter_type;
    number_variables = MC.NumberVariables();
    number_equations = MC.NumberSingles();
    x = new double[7 * number_variables];
    for (int i = 0; i < number_variables * 7; i++) x[i] = 0;
    calico_rp_options_default(&options);
    const int cn = MC.NumberCameras(), pn = MC.NumberPatterns();
    const int tn = number_variables - cn - pn;
    device_ = device;
    check(calico_rp_create(cn, pn, tn, MC.pattern_a_vars[MC.p_star()], MC.time_f_vars[MC.t_star()], device,
                           &handle_),
          "calico_rp_create");
  }
  CeresProblemClass(const CeresProblemClass&) = delete;
  CeresProblemClass& operator=(const CeresProblemClass&) = delete;

  ~CeresProblemClass() {  // solving-structure.cpp:1026-1051
    if (x != 0) {
      delete[] x;
      x = 0;
    }
    calico_rp_destroy(handle_);
    calico_rp_destroy(ae_handle_);
  }

  // solving-structure.cpp:673-915.  The last vars_to_process entries of solved_vars are
  // new: their 7-vectors are taken from MC.V_initial, nothing inverted (:747-763); every
  // equation all of whose variables are initialised (MC.singles_open[i] == false after
  // MC.UpdateSinglesOpenFlag(), :729-742) and that was not added before joins the problem
  // as one relation C = A_i P T (:786-910; the four functor cases are told apart inside
  // the library).  Returns the number of equations added.  Uses MC.A[i] (4x4),
  // MC.singles[e].lhs / .rhs[1] / .rhs[2], MC.singles_open, MC.UpdateSinglesOpenFlag().
  int AddToProblemAlgebraicError(MCcaliT& MC, std::vector<int>& solved_vars, std::vector<int>& equations_added,
                                 std::ostream& out, int vars_to_process) {
    if (!ae_handle_) {
      const int cn = MC.NumberCameras(), pn = MC.NumberPatterns();
      check(calico_ae_create(cn, pn, number_variables - cn - pn, MC.pattern_a_vars[MC.p_star()],
                             MC.time_f_vars[MC.t_star()], device_, &ae_handle_),
            "calico_ae_create");
      bit_vector_bool.assign(number_variables, false);
      bit_vector_equations.assign(number_equations, false);
    }
    if (a_vectors.empty()) {
      a_vectors.resize(7 * (size_t)number_equations);
      for (int i = 0; i < number_equations; i++) ConvertMatrixTo7QuaternionRepresentationx(MC.A[i], &a_vectors[7 * (size_t)i]);
    }
    out << "New vars in algebraic error ";
    const int start_index = (int)solved_vars.size() - vars_to_process, end_index = (int)solved_vars.size();
    for (int i = start_index; i < end_index; i++) {
      bit_vector_bool[solved_vars[i]] = true;
      out << solved_vars[i] << " ";
      ConvertMatrixTo7QuaternionRepresentationx(MC.V_initial[solved_vars[i]], &x[7 * solved_vars[i]]);
    }
    out << std::endl;
    MC.UpdateSinglesOpenFlag();
    std::vector<int32_t> cam, pat, tim;
    std::vector<double> a7;
    int added = 0;
    for (int i = 0; i < number_equations; i++) {
      if (MC.singles_open[i] == false && bit_vector_equations[i] == false) {
        bit_vector_equations[i] = true;
        equations_added.push_back(i);
        out << "Will add eq " << i << std::endl;
        cam.push_back(MC.singles[i].lhs);
        pat.push_back(MC.singles[i].rhs[1]);
        tim.push_back(MC.singles[i].rhs[2]);
        a7.insert(a7.end(), &a_vectors[7 * (size_t)i], &a_vectors[7 * (size_t)i] + 7);
        ++added;
      }
    }
    check_ae(calico_ae_add_relations(ae_handle_, (int64_t)cam.size(), cam.data(), pat.data(), tim.data(), a7.data()),
             "calico_ae_add_relations");
    return added;
  }

  // solving-structure.cpp:918-964: LM on every relation added so far, then x -> V_initial
  // for every variable flagged so far, "no inverses" (:957).
  void SolveWriteBackToMCAlgebraicError(MCcaliT& MC, std::ostream& out, int iterations, bool output_to_terminal) {
    options.max_num_iterations = iterations;
    options.verbose = output_to_terminal ? 1 : 0;
    check_ae(calico_rp_solve(ae_handle_, x, &options, &summary), "calico_rp_solve (algebraic error)");
    char line[256];
    calico_rp_brief_report(&summary, line, sizeof line);
    out << line << std::endl;
    if (output_to_terminal) {
      std::cout << line << "\n";
      std::cout << "After running solver " << std::endl;
    }
    for (int i = 0; i < number_variables; i++) {
      if (bit_vector_bool[i] == true) {
        auto M = MC.V_initial[i];
        Convert7ParameterQuaternionRepresentationIntoMatrix(&x[7 * i], M);
        MC.V_initial[i] = M;
      }
    }
  }

  // solving-structure.cpp:422-460: cameras as they are, patterns and times INVERTED.
  // Uses MC.V_initial[v] (4x4 with operator()(i,j)).
  void SetUpXForReprojectionError(MCcaliT& MC) {
    for (int v_index = 0, vn = MC.NumberVariables(); v_index < vn; v_index++) {
      if (v_index < MC.NumberCameras()) {
        ConvertMatrixTo7QuaternionRepresentationx(MC.V_initial[v_index], &x[7 * v_index]);
      } else {
        auto Minv = Inverse4(MC.V_initial[v_index]);
        ConvertMatrixTo7QuaternionRepresentationx(Minv, &x[7 * v_index]);
      }
    }
  }

  // solving-structure.cpp:463-669: one residual block per used corner of every FR in
  // equation_ordering[start_eq_index, end_eq_index).  Uses MC.singles[e].lhs / .rhs[1]
  // / .rhs[2], MC.V_index.at(), CCV[c]->start_time_this_camera,
  // CCV[c]->P_class->NumberCornersPerPattern() / three_d_points[j].{x,y,z},
  // CCV[c]->points_used_min[lt].at(j), CCV[c]->two_d_point_coordinates_dense[lt](j,0|1).
  // The case split on p*/t* (:528-544) happens inside the library.
  void AddEqsToProblemReprojectionError(MCcaliT& MC, std::vector<CameraCaliT*>& CCV, double* camera_params,
                                        int start_eq_index, int end_eq_index,
                                        const std::vector<int>& equation_ordering) {
    if (!points_set_) {
      const auto& pts = CCV[0]->P_class->three_d_points;
      std::vector<float> xyz(3 * pts.size());
      for (size_t j = 0; j < pts.size(); ++j) {
        xyz[3 * j] = pts[j].x;
        xyz[3 * j + 1] = pts[j].y;
        xyz[3 * j + 2] = pts[j].z;
      }
      check(calico_rp_set_points(handle_, xyz.data(), (int32_t)pts.size()), "calico_rp_set_points");
      points_set_ = true;
    }
    check(calico_rp_set_intrinsics(handle_, camera_params), "calico_rp_set_intrinsics");
    camera_params_ = camera_params;  // Ceres keeps the raw pointers too (solving-structure.cpp:565)
    ccv_ = &CCV;                     // ... here for CopyToCalibration after a solve that refined them

    std::vector<int32_t> cam, pat, tim, corner;
    std::vector<double> u, v;
    for (int eq_index = start_eq_index; eq_index < end_eq_index; eq_index++) {
      const int e = equation_ordering[eq_index];
      const int camera_index = MC.singles[e].lhs;
      const int pattern_graph_index = MC.singles[e].rhs[1];
      const int time_graph_index = MC.singles[e].rhs[2];
      const int time_superscript = MC.V_index.at(time_graph_index);
      const int pattern_superscript = MC.V_index.at(pattern_graph_index);
      const int local_time_index = time_superscript - CCV[camera_index]->start_time_this_camera;
      const int number_corners_per = CCV[camera_index]->P_class->NumberCornersPerPattern();
      const int start_j = number_corners_per * pattern_superscript;
      const int end_j = number_corners_per * (pattern_superscript + 1);
      for (int j = start_j; j < end_j; j++) {
        if (CCV[camera_index]->points_used_min[local_time_index].at(j) == true) {
          cam.push_back(camera_index);
          pat.push_back(pattern_graph_index);
          tim.push_back(time_graph_index);
          corner.push_back(j);
          u.push_back(CCV[camera_index]->two_d_point_coordinates_dense[local_time_index](j, 0));
          v.push_back(CCV[camera_index]->two_d_point_coordinates_dense[local_time_index](j, 1));
        }
      }
    }
    check(calico_rp_add_observations(handle_, (int64_t)cam.size(), cam.data(), pat.data(), tim.data(),
                                     corner.data(), u.data(), v.data()),
          "calico_rp_add_observations");
  }

  // solving-structure.cpp:967-1023: LM on everything added so far, then x -> V_initial
  // (patterns and times inverted back) for every solved variable.
  void SolveWriteBackToMCRP(MCcaliT& MC, std::ostream& out, int iterations, std::vector<int>& solved_vars,
                            bool output_to_terminal) {
    MC.TestSolutionHasIdentityPstarTstar(MC.V_initial);  // solving-structure.cpp:971
    options.max_num_iterations = iterations;
    options.verbose = output_to_terminal ? 1 : 0;
    check(calico_rp_solve(handle_, x, &options, &summary), "calico_rp_solve");
    // extension: refined camera parameters are optimised in place, like x (what Ceres would do
    // with &camera_params[12 * c] as a fourth parameter block)
    if (refine_intrinsics_mask_ && camera_params_) {
      check(calico_rp_get_intrinsics(handle_, camera_params_), "calico_rp_get_intrinsics");
      // ... and reach the calibration objects, which is what the result files and the rrmse
      // report read (multicamera.cpp:1538,1553; camera-calibration.cpp:926)
      if (ccv_) CopyToCalibration(camera_params_, *ccv_);
    }
    char line[256];
    calico_rp_brief_report(&summary, line, sizeof line);
    out << line << std::endl;
    if (output_to_terminal) {
      std::cout << line << "\n";
      std::cout << "After running solver " << std::endl;
    }
    for (int i = 0, nsv = (int)solved_vars.size(); i < nsv; i++) {
      const int current_var = solved_vars[i];
      auto M = MC.V_initial[current_var];
      Convert7ParameterQuaternionRepresentationIntoMatrix(&x[7 * current_var], M);
      if (current_var < MC.NumberCameras())
        MC.V_initial[current_var] = M;
      else
        MC.V_initial[current_var] = Inverse4(M);
    }
    MC.TestSolutionHasIdentityPstarTstar(MC.V_initial);  // solving-structure.cpp:1021
  }

  // EXTENSION (BASELINE.json configs[2]; not in the reference, where the camera parameters are
  // constants inside the functors, solving-structure.hpp:738-749): bit k of mask frees
  // parameter k of every camera's [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6] in the following
  // SolveWriteBackToMCRP calls, which then update camera_params in place and copy them back into
  // the CameraCali objects handed to AddEqsToProblemReprojectionError (CopyToCalibration).
  // 0 = the reference.
  void SetRefineIntrinsics(unsigned mask) {
    check(calico_rp_set_refine_intrinsics(handle_, mask), "calico_rp_set_refine_intrinsics");
    refine_intrinsics_mask_ = mask;
  }

 private:
  void check(int rc, const char* what) {
    if (rc != 0) {
      std::cout << what << " failed (" << rc << "): " << calico_rp_last_error(handle_) << std::endl;
      std::exit(1);
    }
  }
  void check_ae(int rc, const char* what) {
    if (rc != 0) {
      std::cout << what << " failed (" << rc << "): " << calico_rp_last_error(ae_handle_) << std::endl;
      std::exit(1);
    }
  }
  calico_rp_problem* handle_ = nullptr;
  calico_rp_problem* ae_handle_ = nullptr;  // Stage 4, created at the first AddToProblemAlgebraicError
  int device_ = 0;
  bool points_set_ = false;
  unsigned refine_intrinsics_mask_ = 0;
  double* camera_params_ = nullptr;
  std::vector<CameraCaliT*>* ccv_ = nullptr;
  std::vector<double> a_vectors;            // 7 per equation (solving-structure.cpp:677-693)
  std::vector<bool> bit_vector_bool, bit_vector_equations;  // hpp:620-621
};

}  // namespace calico_amd
