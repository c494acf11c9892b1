// acc_bench.hip — which part bounds accumulate_kernel: fetch, MFMA or flush?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I calico_amd/csrc -o tools/_build/acc_bench tools/acc_bench.hip
#include "../calico_amd/csrc/rp_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int n_fr = argc > 1 ? atoi(argv[1]) : 1280000;
  const int per_fr = argc > 2 ? atoi(argv[2]) : 80;
  const int64_t n_obs = (int64_t)n_fr * per_fr;
  const int64_t n_tiles = (n_obs + 255) / 256 + 1;
  std::vector<int64_t> begin(n_fr + 1);
  for (int f = 0; f <= n_fr; ++f) begin[f] = (int64_t)f * per_fr;
  std::vector<int32_t> piece(n_fr + 1, -1);
  const int per_piece = argc > 3 ? atoi(argv[3]) : 16;  // FRs sharing (time, pattern)
  int n_pieces = 0;
  for (int f = 0; f < n_fr; ++f) {
    if (f % rp::ACC_FR_PER_WAVE == 0 || f % per_piece == 0) ++n_pieces;
    piece[f] = n_pieces - 1;
  }
  int64_t* d_begin;
  int32_t* d_piece;
  double *J, *rec, *recB;
  CK(hipMalloc(&d_piece, (n_fr + 1) * 4));
  CK(hipMemcpy(d_piece, piece.data(), (n_fr + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_begin, (n_fr + 1) * 8));
  CK(hipMemcpy(d_begin, begin.data(), (n_fr + 1) * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&J, n_tiles * 44 * 256 * 8));
  CK(hipMemset(J, 0, n_tiles * 44 * 256 * 8));
  CK(hipMalloc(&rec, (int64_t)n_fr * rp::REC_A * 8));
  CK(hipMalloc(&recB, (int64_t)n_pieces * rp::REC_B * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = (double)n_obs * 44 * 8;
  const int per_block = 4 * rp::ACC_FR_PER_WAVE;
  const dim3 grid((n_fr + per_block - 1) / per_block);
  auto run = [&](const char* name, auto kern) {
    double tot = 0;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, n_fr, (const int64_t*)d_begin, (const int32_t*)d_piece, (const double*)J, rec, recB);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) tot += ms;
    }
    printf("%-28s avg %.3f ms -> %.0f GB/s of J\n", name, tot / 3, bytes / (tot / 3 * 1e-3) / 1e9);
    fflush(stdout);
  };
  run("full", rp::accumulate_kernel<0>);
  run("nt load", rp::accumulate_kernel<8>);
  run("nt store", rp::accumulate_kernel<16>);
  run("nt load+store", rp::accumulate_kernel<24>);
  run("no flush", rp::accumulate_kernel<1>);
  run("no mfma", rp::accumulate_kernel<2>);
  run("no mfma, no flush", rp::accumulate_kernel<3>);
  run("no refetch", rp::accumulate_kernel<4>);
  run("no refetch, no flush", rp::accumulate_kernel<5>);
  run("no refetch, no mfma", rp::accumulate_kernel<6>);
  return 0;
}
