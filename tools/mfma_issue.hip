// tools/mfma_issue.hip — issue interval of back-to-back v_mfma_f64_16x16x4_f64 (dev tool):
// s_memtime ticks per MFMA for one wave per SIMD and four, on one CU and on the whole chip,
// plus wall time per launch (so that ticks can be told from core cycles under power management).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void issue(double* out, long long* ticks, int iters) {
  f64x4 acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = (f64x4){0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  double* out; long long* ticks;
  CK(hipMalloc(&out, 8 * 1024 * 1024 * sizeof(double)));
  CK(hipMalloc(&ticks, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4096, NACC = 8;
  for (int threads : {256, 1024}) {
    for (int blocks : {1, 8, 64, 256, 1024}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(issue<NACC>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> t((size_t)blocks * threads / 64);
      CK(hipMemcpy(t.data(), ticks, t.size() * sizeof(long long), hipMemcpyDeviceToHost));
      double mean = 0; for (long long v : t) mean += (double)v; mean /= t.size();
      const double mfmas_per_wave = (double)iters * NACC;
      const double total = mfmas_per_wave * t.size();
      printf("%4d threads x %4d blocks: %.1f ticks per MFMA per wave, launch %.3f ms -> %.1f TFLOP/s, %.1f ns per MFMA per wave\n",
             threads, blocks, mean / mfmas_per_wave, ms, total * 2048 / ms / 1e9, ms * 1e6 / mfmas_per_wave);
    }
  }
  return 0;
}
