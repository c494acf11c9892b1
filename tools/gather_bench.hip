// tools/gather_bench.hip — what bounds the record gathers (dev tool): list_sum_kernel's access
// pattern on synthetic records.  1.28 M records of 128 doubles; a block sums `width` doubles of
// the records of one list of `len` entries, which are either adjacent (pair-major layout) or
// `stride` records apart (FR order, time-major: the parts of one (camera, pattern) pair on config D).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(128) void sum_kernel(const double* __restrict__ src, int rec, int len, int stride, int nrec,
                                                  double* __restrict__ out) {
  const int b = blockIdx.x, t = threadIdx.x;
  // list b: records first + k * stride, k < len
  const long long first = stride == 1 ? (long long)b * len : (long long)(b / stride) * stride * len + b % stride;
  double acc = 0.0;
  for (int i = 0; i + 8 <= len; i += 8) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(first + (long long)(i + k) * stride) * rec + t];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  out[(long long)b * 128 + t] = acc;
}

// the same with 16-byte loads: width / 2 lanes per record
__global__ __launch_bounds__(64) void sum2_kernel(const double* __restrict__ src, int rec, int len, int stride, int nrec,
                                                  double* __restrict__ out) {
  const int b = blockIdx.x, t = threadIdx.x;
  const long long first = stride == 1 ? (long long)b * len : (long long)(b / stride) * stride * len + b % stride;
  double2 acc = {0.0, 0.0};
  for (int i = 0; i + 8 <= len; i += 8) {
    double2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double2*>(src + (first + (long long)(i + k) * stride) * rec + 2 * t);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc.x += v[k].x;
      acc.y += v[k].y;
    }
  }
  *reinterpret_cast<double2*>(out + (long long)b * 128 + 2 * t) = acc;
}

int main() {
  const int nrec = 1280000, rec = 128, len = 80;  // 16 000 lists of 80 records
  double *src, *out;
  CK(hipMalloc(&src, (size_t)nrec * rec * 8));
  CK(hipMalloc(&out, (size_t)(nrec / len) * 128 * 8));
  CK(hipMemset(src, 0, (size_t)nrec * rec * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int width : {104, 128, 64}) {
    for (int stride : {1, 128}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(sum_kernel, dim3(nrec / len), dim3(width), 0, 0, src, rec, len, stride, nrec, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      float best2 = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(sum2_kernel, dim3(nrec / len), dim3(width / 2), 0, 0, src, rec, len, stride, nrec, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best2) best2 = ms;
      }
      printf("  16-byte loads, %d lanes: %.3f ms\n", width / 2, best2);
      const double bytes = (double)nrec * width * 8;
      printf("width %3d doubles of %d, records %s: %.3f ms, %.2f TB/s of bytes used\n", width, rec,
             stride == 1 ? "adjacent        " : "128 records apart", best, bytes / best * 1e-9);
    }
  }
  return 0;
}
