// reconstruct_x.hpp — host-side mirror of ReconstructXFunctionIDsMCwithDLTs
// (/root/reference/src/solving-structure.cpp:162-390) on top of calico_rae_reconstruct
// (include/calico_rp.h): same name, argument list and outputs; the body gathers what the
// reference's loops gather and makes one library call instead of one ceres::Problem per
// corner.  Template over the reference's types, like ceres_problem_class.hpp.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <ostream>
#include <vector>

#include "../../include/calico_rp.h"
#include "ceres_problem_class.hpp"

namespace calico_amd {

// Uses MC.NumberVariables(), NumberSingles(), singles_open[i], singles[i].lhs / .rhs[1] /
// .rhs[2], V_index.at(); CCV[c]->P_class->three_d_points.size(), NumberCornersPerPattern(),
// start_time_this_camera, points_present[lt].at(k), two_d_point_coordinates_dense[lt](k,0|1),
// cols, rows; vector_variables[v](i,j).  estimated_threed_points[k](0..2) is assigned.
template <class MCcaliT, class CameraCaliT, class Mat4, class Vec3>
inline void ReconstructXFunctionIDsMCwithDLTs(MCcaliT& MC, std::vector<Mat4>& vector_variables,
                                              const std::vector<CameraCaliT*>& CCV, double* camera_params,
                                              std::vector<Vec3>& estimated_threed_points, std::vector<bool>& has_values,
                                              std::ostream& out, int device = 0, calico_rae_problem** reuse = nullptr) {
  const int max_number_obs = (int)CCV[0]->P_class->three_d_points.size();
  if (MC.NumberVariables() + 1 != int(vector_variables.size())) {  // cpp:194-199
    std::cout << "The size of the vector sent to ReconstructXFunctionIDsMC is " << vector_variables.size() << std::endl;
    std::cout << "the number of variables according to MC.NumberVariables() + 1 is " << MC.NumberVariables() + 1 << std::endl;
    std::cout << "They should be the same, quitting." << std::endl;
    std::exit(1);
  }
  const int number_corners_per = CCV[0]->P_class->NumberCornersPerPattern();
  std::vector<int32_t> fr_cam, obs_fr, obs_point, cols_rows;
  std::vector<double> fr_a_hat, u, v;
  for (size_t c = 0; c < CCV.size(); ++c) {
    cols_rows.push_back(CCV[c]->cols);
    cols_rows.push_back(CCV[c]->rows);
  }
  for (int i = 0; i < MC.NumberSingles(); i++) {
    if (MC.singles_open[i] != false) continue;  // cpp:203: all variables have an initial value
    const int camera_index = MC.singles[i].lhs;
    const int pattern_graph_index = MC.singles[i].rhs[1];
    const int time_graph_index = MC.singles[i].rhs[2];
    const int time_superscript = MC.V_index.at(time_graph_index);
    const int pattern_superscript = MC.V_index.at(pattern_graph_index);
    // A_hat = C * T.inverse() * P.inverse(), cpp:217
    const Mat4& Cm = vector_variables.at(camera_index);
    const Mat4 Ti = Inverse4(vector_variables.at(time_graph_index));
    const Mat4 Pi = Inverse4(vector_variables.at(pattern_graph_index));
    double CT[12], A[12];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += Cm(r, k) * Ti(k, c);
        CT[4 * r + c] = s;
      }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) {
        double s = (c == 3) ? CT[4 * r + 3] : 0.0;  // bottom row of a rigid transform is 0 0 0 1
        for (int k = 0; k < 3; ++k) s += CT[4 * r + k] * Pi(k, c);
        A[4 * r + c] = s;
      }
    const int view = (int)fr_cam.size();
    fr_cam.push_back(camera_index);
    fr_a_hat.insert(fr_a_hat.end(), A, A + 12);
    const int local_time_index = time_superscript - CCV[camera_index]->start_time_this_camera;
    const int start_j = number_corners_per * pattern_superscript, end_j = number_corners_per * (pattern_superscript + 1);
    for (int k = start_j; k < end_j; k++) {
      if (CCV[camera_index]->points_present[local_time_index].at(k) == true) {  // cpp:278: is k detected
        obs_fr.push_back(view);
        obs_point.push_back(k);
        u.push_back(CCV[camera_index]->two_d_point_coordinates_dense[local_time_index](k, 0));
        v.push_back(CCV[camera_index]->two_d_point_coordinates_dense[local_time_index](k, 1));
      }
    }
  }
  std::vector<double> xyz(3 * (size_t)max_number_obs, 0.0);
  std::vector<uint8_t> has((size_t)max_number_obs, 0);
  calico_rae_summary summary;
  char err[256] = {0};
  int rc = 0;
  if (reuse) {
    // The reference reconstructs once per solution type from the same detections (multicamera.cpp:812-833):
    // a caller that passes the same handle slot every time uploads them once (calico_rae_create) and
    // only the poses afterwards.  The set of FRs with all variables initialised must not change in between.
    if (!*reuse)
      rc = calico_rae_create(device, (int32_t)CCV.size(), cols_rows.data(), (int32_t)fr_cam.size(), fr_cam.data(), max_number_obs,
                             number_corners_per, (int64_t)obs_fr.size(), obs_fr.data(), obs_point.data(), u.data(), v.data(),
                             reuse, err, sizeof err);
    if (rc == 0) {
      rc = calico_rae_solve(*reuse, camera_params, fr_a_hat.data(), nullptr, xyz.data(), has.data(), nullptr, nullptr, &summary);
      if (rc != 0) std::snprintf(err, sizeof err, "%s", calico_rae_last_error(*reuse));
    }
  } else {
    rc = calico_rae_reconstruct(device, (int32_t)CCV.size(), camera_params, cols_rows.data(), (int32_t)fr_cam.size(),
                                fr_cam.data(), fr_a_hat.data(), max_number_obs, number_corners_per, (int64_t)obs_fr.size(),
                                obs_fr.data(), obs_point.data(), u.data(), v.data(), nullptr, xyz.data(), has.data(), nullptr,
                                nullptr, &summary, err, sizeof err);
  }
  if (rc != 0) {
    std::cout << "calico_rae_reconstruct failed (" << rc << "): " << err << std::endl;
    std::exit(1);
  }
  out << "Point reconstruction: " << summary.n_solved << " of " << summary.n_points << " corners, "
      << summary.max_point_iterations << " iterations at most" << std::endl;
  has_values.assign(max_number_obs, false);
  estimated_threed_points.clear();
  estimated_threed_points.resize(max_number_obs, Vec3());
  for (int k = 0; k < max_number_obs; k++) {
    has_values[k] = has[k] != 0;
    estimated_threed_points[k](0) = xyz[3 * k];
    estimated_threed_points[k](1) = xyz[3 * k + 1];
    estimated_threed_points[k](2) = xyz[3 * k + 2];
  }
}

}  // namespace calico_amd
