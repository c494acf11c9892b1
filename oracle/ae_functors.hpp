// oracle/ae_functors.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the four algebraic-error cost functors of the reference (Stage 4),
// templated on the scalar type like rp_functors.hpp:
//   AlgebraicErrorCam             solving-structure.hpp:173-229   residual = C - A
//   AlgebraicErrorCamPattern      :231-310                        C - A P
//   AlgebraicErrorCamPatternTime  :312-386                        C - (A P) T
//   AlgebraicErrorCamTime         :388-462                        C - A T
// 12 residuals = the first three rows of the 4x4, row-major (hpp:224-226 etc.); the
// pattern / time blocks are NOT inverted in this stage (hpp:397-398).
#pragma once
#include "rp_functors.hpp"

namespace oracle {

//   pattern_var == nullptr <=> p*, time_var == nullptr <=> t*; a7 = the FR's A as a 7-vector
template <typename T>
inline void algebraic_residual(const T* camera_var, const T* pattern_var, const T* time_var, const double* a7,
                               T* residuals /*12*/) {
  T CM[16], AM[16], a[7];
  for (int i = 0; i < 7; ++i) a[i] = T(a7[i]);
  seven_to_mat4(camera_var, CM);
  seven_to_mat4(a, AM);
  T term0[16], term1[16];
  const T* rhs = AM;
  if (pattern_var && time_var) {
    T PM[16], tM[16];
    seven_to_mat4(pattern_var, PM);
    seven_to_mat4(time_var, tM);
    mat_mul(AM, PM, term0, 4, 4, 4);     // hpp:366
    mat_mul(term0, tM, term1, 4, 4, 4);  // hpp:367
    rhs = term1;
  } else if (pattern_var) {
    T PM[16];
    seven_to_mat4(pattern_var, PM);
    mat_mul(AM, PM, term0, 4, 4, 4);  // hpp:289
    rhs = term0;
  } else if (time_var) {
    T tM[16];
    seven_to_mat4(time_var, tM);
    mat_mul(AM, tM, term0, 4, 4, 4);  // hpp:441
    rhs = term0;
  }
  for (int i = 0; i < 12; ++i) residuals[i] = CM[i] - rhs[i];
}

}  // namespace oracle
