// Test helper: runs calico_amd/csrc/rp_model.hpp (the arithmetic of the HIP sweep
// kernels) on the HOST so that the analytic Jacobian can be checked against the
// dual-number oracle and the mpmath golden vectors without a GPU.  Built only by
// tests (tests/_build/libmodel_host.so); the product library never contains it.
#include "../../calico_amd/csrc/rp_model.hpp"

extern "C" void rp_model_host_eval(const double* cam, const double* pat, const double* tim,
                                   int has_pat, int has_tim, const double* K, const float* Xf,
                                   double u, double v, double* r, double* J) {
  const double X[3] = {Xf[0], Xf[1], Xf[2]};
  rp::eval_observation<true>(cam, has_pat ? pat : nullptr, has_tim ? tim : nullptr, K, X, u, v, r, J);
}

// The fused kernel's Gram space: U[2][16] (rotation-vector columns) and the FR-constant
// matrix M[16][22] with [J_c | J_p | J_t | r] = U M.
namespace {
struct USink {
  double* U;
  void operator()(int k, double v) { U[k] = v; }
};
}  // namespace
extern "C" void rp_model_host_eval_w(const double* cam, const double* pat, const double* tim, int has_pat, int has_tim,
                                     const double* K, const float* Xf, double u, double v, double* r, double* U,
                                     double* M) {
  const double X[3] = {Xf[0], Xf[1], Xf[2]};
  double frp[rp::FRP_SIZE];
  rp::prep_fr(cam, has_pat ? pat : nullptr, has_tim ? tim : nullptr, K, frp);
  USink s{U};
  rp::eval_prepared_w(frp, X, u, v, r, s);
  rp::rotvec_to_reference_columns(cam, has_pat ? pat : nullptr, has_tim ? tim : nullptr, M);
}

// The compact form of the same Gram space: V[2][8] (chain columns) and C[8][16] with U = V C.
namespace {
struct VSink {
  double* V;
  void operator()(int k, double v) { V[k] = v; }
};
}  // namespace
extern "C" void rp_model_host_eval_chain(const double* cam, const double* pat, const double* tim, int has_pat, int has_tim,
                                         const double* K, const float* Xf, double u, double v, double* r, double* V,
                                         double* C) {
  const double X[3] = {Xf[0], Xf[1], Xf[2]};
  double blk[3][rp::PREP_SIZE], e[rp::FRC_SIZE];
  rp::prep_block(cam, blk[0]);
  rp::prep_block(has_tim ? tim : nullptr, blk[1]);
  rp::prep_block(has_pat ? pat : nullptr, blk[2]);
  rp::prep_fr_chain(blk[0], blk[1], blk[2], K, e);
  for (int k = 0; k < 16; ++k) V[k] = 0.0;
  VSink s{V};
  rp::eval_chain(e, X, u, v, r, s);
  for (int k = 0; k < 8; ++k)
    for (int j = 0; j < 16; ++j) {
      const int o = rp::chain_map_offset(k, j);
      C[16 * k + j] = o >= 0 ? e[o] : (o == -2 ? 1.0 : 0.0);
    }
}

// Camera parameters as a fourth block: Kc[2][12] from eval_intrinsics_columns, and the fused
// kernel's variant (eval_prepared_w<true>): U[2][16] and the same K columns in one pass.
namespace {
struct UKSink {
  double* U;
  double* Kc;
  void operator()(int k, double v) { (k < 32 ? U[k] : Kc[k - 32]) = v; }
};
}  // namespace
extern "C" void rp_model_host_eval_k(const double* cam, const double* pat, const double* tim, int has_pat, int has_tim,
                                     const double* K, const float* Xf, double u, double v, unsigned mask, double* Kc,
                                     double* r, double* U, double* Kc_w) {
  const double X[3] = {Xf[0], Xf[1], Xf[2]};
  double frp[rp::FRP_SIZE];
  rp::prep_fr(cam, has_pat ? pat : nullptr, has_tim ? tim : nullptr, K, frp);
  rp::ArraySink s{Kc};
  rp::eval_intrinsics_columns(frp, X, mask, s);
  UKSink w{U, Kc_w};
  rp::eval_prepared_w<true>(frp, X, u, v, r, w, mask);
}

#include "../../calico_amd/csrc/ae_model.hpp"

// Stage-4 algebraic error: r[12], J[12][21]
extern "C" void ae_model_host_eval(const double* cam, const double* pat, const double* tim, int has_pat, int has_tim,
                                   const double* a7, double* r, double* J) {
  rp::ae_eval(cam, has_pat ? pat : nullptr, has_tim ? tim : nullptr, a7, r, J);
}

// the column-wise form the records kernel uses (four lanes per relation): M[12][22] = [J | r]
extern "C" void ae_model_host_eval_columns(const double* cam, const double* pat, const double* tim, int has_pat,
                                           int has_tim, const double* a7, double* M) {
  double C[rp::PREP_SIZE], P[rp::PREP_SIZE], T[rp::PREP_SIZE], A[rp::PREP_SIZE];
  rp::prep_block(cam, C);
  rp::prep_block(has_pat ? pat : nullptr, P);
  rp::prep_block(has_tim ? tim : nullptr, T);
  rp::prep_block(a7, A);
  for (int j = 0; j < 4; ++j) rp::ae_eval_column<22, 21>(C, P, T, A, j, M);
}

#include "../../calico_amd/csrc/rae_model.hpp"

// point reconstruction: r[2], J[2][3]; and the two DLT rows of a view
extern "C" void rae_model_host_eval(const double* a_hat, const double* K, const double* X, double u, double v, double* r,
                                    double* J) {
  rp::rae_eval(a_hat, K, X, u, v, r, J);
}
extern "C" void rae_model_host_dlt_rows(const double* a_hat, const double* K, double cols, double rows, double u, double v,
                                        double* row1, double* row2) {
  rp::rae_dlt_rows(a_hat, K, cols, rows, u, v, row1, row2);
}
