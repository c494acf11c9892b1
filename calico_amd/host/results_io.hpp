// results_io.hpp — the reference's result files and its pose-error metric, for the state
// the GPU path leaves in MC.V_initial.  Host-only text and 3x3 algebra;
// templates over the reference's types like ceres_problem_class.hpp, so that with the real
// MCcali / CameraCali / Eigen::Matrix4d they are drop-ins and the tests can use stand-ins.
//
//   WriteCameraCalibrationResult   multicamera.cpp:1519-1560  (format: README.md:377-392)
//   WriteVariablesTxt              multicamera.cpp:1751-1774  ("variables.txt")
//   RelativeExternals, CameraPoseError   compute-error.cpp:809-819, 860-896
#pragma once
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace calico_amd {

// multicamera.cpp:1519-1560: first line the number of cameras, then per camera
//   name K(3x3 row-major) R(3x3 row-major) t(3) distortion(all rows)
// every number followed by one space, default stream formatting (6 significant digits).
// Uses CCV[i]->internal_parameters(r,c), CCV[i]->distortion.rows(), CCV[i]->distortion(r),
// ext_to_use[i](r,c).  Returns false where the reference prints and exits.
template <class CameraCaliT, class Mat4>
inline bool WriteCameraCalibrationResult(const std::vector<CameraCaliT*>& CCV, const std::vector<std::string>& camera_names,
                                         const std::vector<Mat4>& ext_to_use, const std::string& filename) {
  std::ofstream out(filename.c_str());
  if (!out.good()) return false;
  const int cn = (int)CCV.size();
  out << cn << std::endl;
  for (int i = 0; i < cn; i++) {
    out << camera_names[i] << " ";
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) out << CCV[i]->internal_parameters(r, c) << " ";
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) out << ext_to_use[i](r, c) << " ";
    for (int r = 0; r < 3; r++) out << ext_to_use[i](r, 3) << " ";
    for (int r = 0, rn = (int)CCV[i]->distortion.rows(); r < rn; r++) out << CCV[i]->distortion(r) << " ";
    out << std::endl;
  }
  return out.good();
}

// `out << Matrix4d` as Eigen's default IOFormat prints it: every coefficient formatted with
// the stream's precision, right-aligned to the widest one, " " between columns, "\n"
// between rows, no trailing newline.
template <class Mat4>
inline void PrintMatrix4LikeEigen(std::ostream& out, const Mat4& M) {
  std::string cell[16];
  size_t width = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      std::ostringstream s;
      s.copyfmt(out);
      s.width(0);
      s << M(i, j);
      cell[4 * i + j] = s.str();
      width = std::max(width, cell[4 * i + j].size());
    }
  for (int i = 0; i < 4; ++i) {
    if (i) out << "\n";
    for (int j = 0; j < 4; ++j) {
      if (j) out << " ";
      out << std::string(width - cell[4 * i + j].size(), ' ') << cell[4 * i + j];
    }
  }
}

// multicamera.cpp:1751-1774: "C i" / "P i, v" / "T i, v" headers, each followed by the 4x4.
template <class Mat4>
inline bool WriteVariablesTxt(const std::string& write_file, int cn, int pn, int tn, const std::vector<Mat4>& vector_to_use) {
  std::ofstream out(write_file.c_str());
  if (!out.good()) return false;
  int v = 0;
  for (int i = 0; i < cn; i++, v++) {
    out << "C " << i << std::endl;
    PrintMatrix4LikeEigen(out, vector_to_use[v]);
    out << std::endl;
  }
  for (int i = 0; i < pn; i++, v++) {
    out << "P " << i << ", " << v << std::endl;
    PrintMatrix4LikeEigen(out, vector_to_use[v]);
    out << std::endl;
  }
  for (int i = 0; i < tn; i++, v++) {
    out << "T " << i << ", " << v << std::endl;
    PrintMatrix4LikeEigen(out, vector_to_use[v]);
    out << std::endl;
  }
  return out.good();
}

// compute-error.cpp:809-811: externals relative to camera 0, E_i * E_0^-1 (rigid inverse).
template <class Mat4>
inline std::vector<Mat4> RelativeExternals(const std::vector<Mat4>& ext) {
  std::vector<Mat4> rel(ext.size(), ext.empty() ? Mat4() : ext[0]);
  if (ext.empty()) return rel;
  Mat4 inv0 = ext[0];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) inv0(i, j) = ext[0](j, i);
    inv0(i, 3) = -(ext[0](0, i) * ext[0](0, 3) + ext[0](1, i) * ext[0](1, 3) + ext[0](2, i) * ext[0](2, 3));
  }
  for (int j = 0; j < 4; ++j) inv0(3, j) = j == 3 ? 1.0 : 0.0;
  for (size_t n = 0; n < ext.size(); ++n)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += ext[n](i, k) * inv0(k, j);
        rel[n](i, j) = s;
      }
  return rel;
}

// compute-error.cpp:860-896: sum over cameras of |t_gt - t| and of the angle (degrees) of
// Rb^T Ra (Eigen::AngleAxisd: angle in [0, pi]), both divided by (number_cameras - 1).
template <class Mat4>
inline void CameraPoseError(const std::vector<Mat4>& externals_gt_relative, const std::vector<Mat4>& externals_relative,
                            double* translation_error_average, double* rotation_error_degrees_average) {
  const int number_cameras = (int)externals_gt_relative.size();
  double translation_error = 0, rot_error_degrees = 0;
  for (int n = 0; n < number_cameras; ++n) {
    const Mat4& A = externals_gt_relative[n];
    const Mat4& B = externals_relative[n];
    double d2 = 0;
    for (int j = 0; j < 3; ++j) d2 += (A(j, 3) - B(j, 3)) * (A(j, 3) - B(j, 3));
    translation_error += std::sqrt(d2);
    double Rc[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rc[i][j] = B(0, i) * A(0, j) + B(1, i) * A(1, j) + B(2, i) * A(2, j);
    const double vx = Rc[2][1] - Rc[1][2], vy = Rc[0][2] - Rc[2][0], vz = Rc[1][0] - Rc[0][1];
    const double angle = std::atan2(0.5 * std::sqrt(vx * vx + vy * vy + vz * vz), 0.5 * (Rc[0][0] + Rc[1][1] + Rc[2][2] - 1.0));
    rot_error_degrees += angle * (180.0 / M_PI);
  }
  if (translation_error_average) *translation_error_average = translation_error / double(number_cameras - 1);
  if (rotation_error_degrees_average) *rotation_error_degrees_average = rot_error_degrees / double(number_cameras - 1);
}

// compute-error.cpp:898-908: the two lines of calicoErrorFile.txt
inline bool WritePoseErrorFile(const std::string& filename, double translation_error_average, double rotation_error_degrees_average) {
  std::ofstream out(filename.c_str());
  if (!out.good()) return false;
  out << "translation error, average: " << translation_error_average << std::endl;
  out << "rotation error, degrees, average: " << rotation_error_degrees_average << std::endl;
  return out.good();
}

}  // namespace calico_amd
