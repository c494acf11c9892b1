// acc_bench.hip — which part bounds accumulate_kernel? (the 16-column fused kernel it also timed was removed in round 3)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I calico_amd/csrc -o tools/_build/acc_bench tools/acc_bench.hip
#include "../calico_amd/csrc/rp_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <typename T>
T* up(const std::vector<T>& v) {
  T* d;
  CK(hipMalloc(&d, v.size() * sizeof(T)));
  CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int n_fr = argc > 1 ? atoi(argv[1]) : 1280000;
  const int per_fr = argc > 2 ? atoi(argv[2]) : 80;
  const int per_piece = argc > 3 ? atoi(argv[3]) : 16;  // FRs sharing (time, pattern)
  const int cn = 128, pn = 8;
  const int64_t n_obs = (int64_t)n_fr * per_fr;
  const int64_t n_tiles = (n_obs + 255) / 256 + 1;
  const int64_t L = rp::acc_wave_obs(n_obs), n_waves = (n_obs + L - 1) / L;
  std::vector<int64_t> begin(n_fr + 1);
  for (int f = 0; f <= n_fr; ++f) begin[f] = (int64_t)f * per_fr;
  std::vector<int32_t> obs_fr(n_obs), corner(n_obs), wave_part0(n_waves + 1), part_fr, part_piece;
  std::vector<double> u(n_obs), v(n_obs);
  for (int64_t i = 0; i < n_obs; ++i) {
    obs_fr[i] = (int)(i / per_fr);
    corner[i] = (int)(i % per_fr);
    u[i] = 600 + (i * 37 % 100);
    v[i] = 450 + (i * 91 % 100);
  }
  int n_pieces = 0;
  for (int64_t w = 0; w < n_waves; ++w) {
    const int64_t i0 = w * L, i1 = std::min(n_obs, i0 + L);
    wave_part0[w] = (int)part_fr.size();
    for (int f = (int)(i0 / per_fr); f < n_fr && begin[f] < i1; ++f) {
      const bool fresh = (int)part_fr.size() == wave_part0[w] || f % per_piece == 0;
      if (fresh) ++n_pieces;
      part_fr.push_back(f);
      part_piece.push_back(n_pieces - 1);
    }
  }
  wave_part0[n_waves] = (int)part_fr.size();
  part_piece.push_back(-1);
  const int n_parts = (int)part_fr.size();
  // FR -> blocks as in config D; parameters near identity
  const int tn = n_fr / cn + 1, vn = cn + pn + tn;
  std::vector<int32_t> fc(n_fr), fp(n_fr), ft(n_fr);
  for (int f = 0; f < n_fr; ++f) {
    fc[f] = f % cn;
    fp[f] = cn + (f / per_piece) % pn;
    ft[f] = cn + pn + f / cn;
  }
  std::vector<double> x(7 * (size_t)vn), intr(12 * cn, 0.0);
  for (int i = 0; i < vn; ++i) {
    double* q = &x[7 * i];
    q[0] = 1; q[1] = 0.01 * (i % 7); q[2] = 0.02; q[3] = 0.01; q[4] = i % 5; q[5] = i % 3; q[6] = (i < cn) ? 3000 : 1;
  }
  for (int c = 0; c < cn; ++c) { intr[12 * c] = 1400; intr[12 * c + 1] = 600; intr[12 * c + 2] = 1400; intr[12 * c + 3] = 450; intr[12 * c + 4] = 0.01; }
  std::vector<float> pts(3 * 640);
  for (int j = 0; j < 640; ++j) { pts[3 * j] = 45.f * (j % 10); pts[3 * j + 1] = 45.f * ((j / 10) % 8); pts[3 * j + 2] = 0; }

  rp::SweepArgs a;
  a.n_obs = n_obs; a.x = up(x); a.intr = up(intr); a.pts = up(pts); a.obs_u = up(u); a.obs_v = up(v);
  a.obs_corner = up(corner); a.obs_fr = up(obs_fr); a.fr_cam = up(fc); a.fr_pat = up(fp); a.fr_tim = up(ft);
  double* prep;
  CK(hipMalloc(&prep, (size_t)(vn + 1) * 12 * 8));
  a.n_vars = vn;
  a.prep = a.prep_out = prep;
  hipLaunchKernelGGL(rp::prep_vars_kernel, dim3((vn + 256) / 256), dim3(256), 0, 0, a.x, vn, prep);
  const int64_t* d_begin = up(begin);
  const int32_t* d_wp0 = up(wave_part0);
  const int32_t* d_pp = up(part_piece);
  double *J, *recA, *recB;
  CK(hipMalloc(&J, n_tiles * 44 * 256 * 8));
  CK(hipMemset(J, 0, n_tiles * 44 * 256 * 8));
  a.J = J;
  CK(hipMalloc(&recA, (int64_t)n_parts * rp::REC_A * 8));
  CK(hipMalloc(&recB, (int64_t)n_pieces * rp::REC_B * 8));
  CK(hipMemset(recA, 0, (int64_t)n_parts * rp::REC_A * 8));
  CK(hipMemset(recB, 0, (int64_t)n_pieces * rp::REC_B * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = (double)n_obs * 44 * 8;
  auto time = [&](const char* name, auto&& launch) {
    double tot = 0;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) tot += ms;
    }
    CK(hipGetLastError());
    printf("%-34s avg %.3f ms -> %.0f GB/s of J, %.0f Mobs/s\n", name, tot / 3, bytes / (tot / 3 * 1e-3) / 1e9, n_obs / (tot / 3 * 1e-3) / 1e6);
    fflush(stdout);
  };
  const dim3 gacc((unsigned)((n_waves + 3) / 4));
#define ACC(V) [&] { hipLaunchKernelGGL(rp::accumulate_kernel<V>, gacc, dim3(256), 0, 0, n_obs, (int)L, a.obs_fr, d_begin, d_wp0, d_pp, (const double*)J, recA, recB); }
  time("sweep (materialised)", [&] { rp::launch_jacobian_sweep(a, 0); });
  time("accumulate full", ACC(0));
  time("accumulate no records", ACC(1));
  time("accumulate no mfma", ACC(2));
  time("accumulate no refetch", ACC(4));
  return 0;
}
