// tools/mfma44_probe.hip — v_mfma_f64_4x4x4_4b_f64 on gfx950: issue rate against the 16x16x4
// tile and the operand / result lane layout (found by one-hot probing).  Dev tool.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma44_probe tools/mfma44_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void rate44(double* out, int iters) {
  double acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0;
  for (int k = 0; k < NACC; ++k) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void rate16(double* out, int iters) {
  f64x4 acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = (f64x4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// one wave per (la, lb): A one-hot at lane la, B one-hot at lane lb
__global__ __launch_bounds__(64) void probe(double* out) {
  const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, l = threadIdx.x;
  const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
  out[blockIdx.x * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
int main() {
  double* out; CK(hipMalloc(&out, 8192 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
  const int iters = 20000, blocks = 1024;
  float ms = time([&] { hipLaunchKernelGGL(rate16<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("16x16x4, 4 acc: %.3f ms  %.1f TFLOP/s\n", ms, (double)blocks * 4 * iters * 4 * 2048 / ms / 1e9);
  ms = time([&] { hipLaunchKernelGGL(rate44<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("4x4x4_4b, 4 acc: %.3f ms  %.1f TFLOP/s\n", ms, (double)blocks * 4 * iters * 4 * 512 / ms / 1e9);
  ms = time([&] { hipLaunchKernelGGL(rate44<8>, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("4x4x4_4b, 8 acc: %.3f ms  %.1f TFLOP/s\n", ms, (double)blocks * 4 * iters * 8 * 512 / ms / 1e9);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, out);
  std::vector<double> h(4096 * 64);
  CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  // for A lane la: which B lanes pair with it, and where the product lands
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
    printf("\n");
  }
  return 0;
}
