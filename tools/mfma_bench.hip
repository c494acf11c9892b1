// tools/mfma_bench.hip — fp64 MFMA issue rate and accumulate-kernel ablations (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate(double* out, int iters) {
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
__global__ __launch_bounds__(256) void fma_rate(double* out, int iters) {
  double acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = k;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fma(acc[k], b, a);
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// Do fp64 MFMA and fp64 FMA overlap?  Each wave issues NM MFMAs and NF FMAs per iteration, all
// independent; time = max(parts) if the pipes are separate, = sum if they share the datapath.
template <int NM, int NF>
__global__ __launch_bounds__(256) void mixed_rate(double* out, int iters) {
  f64x4 macc[NM > 0 ? NM : 1];
  for (int k = 0; k < (NM > 0 ? NM : 1); ++k) macc[k] = (f64x4){0, 0, 0, 0};
  double facc[NF > 0 ? NF : 1];
  for (int k = 0; k < (NF > 0 ? NF : 1); ++k) facc[k] = k;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NM; ++k) macc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, macc[k], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NF; ++k) facc[k] = fma(facc[k], b, a);
  }
  double s = 0;
  for (int k = 0; k < (NM > 0 ? NM : 1); ++k) s += macc[k][0] + macc[k][1] + macc[k][2] + macc[k][3];
  for (int k = 0; k < (NF > 0 ? NF : 1); ++k) s += facc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// waves 0,1 of a workgroup issue MFMAs only, waves 2,3 FMAs only
__global__ __launch_bounds__(256) void split_rate(double* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9, s = 0;
  if ((mode & 1) && wave < 2) {
    f64x4 m0 = {0, 0, 0, 0}, m1 = m0;
    for (int it = 0; it < iters; ++it) {
      m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m1, 0, 0, 0);
    }
    s = m0[0] + m1[1];
  } else if ((mode & 2) && wave >= 2) {
    double f[8];
    for (int k = 0; k < 8; ++k) f[k] = k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fma(f[k], b, a);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fma(f[k], b, a);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fma(f[k], b, a);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fma(f[k], b, a);
    }
    for (int k = 0; k < 8; ++k) s += f[k];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* out; CK(hipMalloc(&out, 8192 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
  const int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    float ms = time([&] { hipLaunchKernelGGL(mfma_rate<3>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    double n = (double)blocks * 4 * iters * 3;
    printf("mfma f64 16x16x4, 3 acc, %4d blocks: %.3f ms  %.1f TFLOP/s  %.1f ns/MFMA/wave-slot\n", blocks, ms, n * 2048 / ms / 1e9, ms * 1e6 / (iters * 3.0));
    ms = time([&] { hipLaunchKernelGGL(mfma_rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    n = (double)blocks * 4 * iters;
    printf("mfma f64 16x16x4, 1 acc, %4d blocks: %.3f ms  %.1f TFLOP/s\n", blocks, ms, n * 2048 / ms / 1e9);
  }
  for (int blocks : {1024, 2048}) {
    float ms = time([&] { hipLaunchKernelGGL(fma_rate, dim3(blocks), dim3(256), 0, 0, out, iters); });
    double n = (double)blocks * 256 * iters * 8.0 * 2;
    printf("v_fma_f64, %4d blocks: %.3f ms  %.1f TFLOP/s\n", blocks, ms, n / ms / 1e9);
  }
  {
    const int blocks = 1024, it2 = 5000;
    float m = time([&] { hipLaunchKernelGGL((mixed_rate<2, 0>), dim3(blocks), dim3(256), 0, 0, out, it2); });
    float f = time([&] { hipLaunchKernelGGL((mixed_rate<0, 32>), dim3(blocks), dim3(256), 0, 0, out, it2); });
    float b = time([&] { hipLaunchKernelGGL((mixed_rate<2, 32>), dim3(blocks), dim3(256), 0, 0, out, it2); });
    printf("same wave: 2 MFMA %.3f ms, 32 FMA %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", m, f, b, m + f, m > f ? m : f);
    m = time([&] { hipLaunchKernelGGL(split_rate, dim3(blocks), dim3(256), 0, 0, out, it2, 1); });
    f = time([&] { hipLaunchKernelGGL(split_rate, dim3(blocks), dim3(256), 0, 0, out, it2, 2); });
    b = time([&] { hipLaunchKernelGGL(split_rate, dim3(blocks), dim3(256), 0, 0, out, it2, 3); });
    printf("different waves: MFMA waves %.3f ms, FMA waves %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", m, f, b, m + f, m > f ? m : f);
  }
  return 0;
}
