// read_bench.hip — HBM read ceilings for the access shapes of the accumulate kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o read_bench read_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double f64x2 __attribute__((ext_vector_type(2)));

// (1) plain streaming read, one 16-B load per lane per step, `unroll` independent loads in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read(const f64x2* __restrict__ p, int64_t n16, double* out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  f64x2 acc = {0.0, 0.0};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    f64x2 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc.x + acc.y == 1.2345e-300) out[0] = acc.x;
}

// (2) block-contiguous streaming: each workgroup reads a contiguous chunk of `chunk16` 16-B items
__global__ __launch_bounds__(256) void chunk_read(const f64x2* __restrict__ p, int64_t n16, int chunk16, double* out) {
  const int64_t b = (int64_t)blockIdx.x * chunk16;
  f64x2 acc = {0.0, 0.0};
  for (int i = threadIdx.x; i < chunk16; i += 1024) {
    f64x2 v0 = p[b + i], v1 = {0, 0}, v2 = {0, 0}, v3 = {0, 0};
    if (i + 256 < chunk16) v1 = p[b + i + 256];
    if (i + 512 < chunk16) v2 = p[b + i + 512];
    if (i + 768 < chunk16) v3 = p[b + i + 768];
    acc += v0 + v1 + v2 + v3;
  }
  if (acc.x + acc.y == 1.2345e-300) out[0] = acc.x;
}

// (3) the accumulate kernel's line-wise group fetch: a wave walks `groups` consecutive 16-observation
// groups of the tiled layout [tile][44][256], 6 x 16-B loads per lane per group, one group ahead in flight
__global__ __launch_bounds__(256) void group_read(const double* __restrict__ J, int64_t n_groups, int groups_per_wave, double* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t g0 = ((int64_t)blockIdx.x * 4 + wave) * groups_per_wave;
  const int lrow = lane >> 3, lpiece = (lane & 7) * 2;
  f64x2 acc = {0.0, 0.0};
  f64x2 raw[6];
  auto fetch = [&](int64_t grp) {
    const int64_t o16 = grp * 16;
    const double* base = J + (o16 >> 8) * (int64_t)(44 * 256) + (int)(o16 & 255) + lpiece;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int plane = 8 * i + lrow;
      plane = plane < 44 ? plane : 43;
      raw[i] = *reinterpret_cast<const f64x2*>(base + plane * 256);
    }
  };
  if (g0 >= n_groups) return;
  fetch(g0);
  for (int k = 0; k < groups_per_wave; ++k) {
    f64x2 cur[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) cur[i] = raw[i];
    int64_t nx = g0 + k + 1;
    if (nx >= n_groups) nx = g0 + k;
    fetch(nx);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += cur[i];
  }
  if (acc.x + acc.y == 1.2345e-300) out[0] = acc.x;
}

// (4) whole tiles per workgroup: 256 lanes read a tile's 44 planes, plane by plane (2 KB rows)
__global__ __launch_bounds__(256) void tile_read(const double* __restrict__ J, int64_t n_tiles, int tiles_per_wg, double* out) {
  f64x2 acc = {0.0, 0.0};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int64_t tile = (int64_t)blockIdx.x * tiles_per_wg + t;
    if (tile >= n_tiles) break;
    const f64x2* base = reinterpret_cast<const f64x2*>(J + tile * (int64_t)(44 * 256));
    // 44*256 doubles = 5632 f64x2 = 22 per lane
    f64x2 v[22];
#pragma unroll
    for (int i = 0; i < 22; ++i) v[i] = base[i * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 22; ++i) acc += v[i];
  }
  if (acc.x + acc.y == 1.2345e-300) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const int64_t n_obs = argc > 1 ? atoll(argv[1]) : 102400000;
  const int64_t n_tiles = (n_obs + 255) / 256;
  const int64_t n_dbl = n_tiles * 44 * 256;
  const double bytes = (double)n_dbl * 8;
  double* J;
  double* out;
  CK(hipMalloc(&J, n_dbl * 8));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(J, 0, n_dbl * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto&& launch) {
    double tot = 0;
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) { tot += ms; if (ms < best) best = ms; }
    }
    printf("%-44s avg %.3f ms best %.3f ms -> %.0f GB/s\n", name, tot / 3, best, bytes / (tot / 3 * 1e-3) / 1e9);
    fflush(stdout);
  };
  const int64_t n16 = n_dbl / 2;
  for (int grid : {2048, 8192, 32768}) {
    char nm[64];
    snprintf(nm, 64, "stream_read<4> grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL(stream_read<4>, dim3(grid), dim3(256), 0, 0, (const f64x2*)J, n16, out); });
    snprintf(nm, 64, "stream_read<8> grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL(stream_read<8>, dim3(grid), dim3(256), 0, 0, (const f64x2*)J, n16, out); });
  }
  for (int chunk : {4096, 16384, 65536}) {
    char nm[64];
    snprintf(nm, 64, "chunk_read chunk=%d KB", chunk * 16 / 1024);
    run(nm, [&] { hipLaunchKernelGGL(chunk_read, dim3((unsigned)(n16 / chunk)), dim3(256), 0, 0, (const f64x2*)J, n16, chunk, out); });
  }
  const int64_t n_groups = n_tiles * 16;
  for (int gpw : {5, 20, 80}) {
    char nm[64];
    snprintf(nm, 64, "group_read groups/wave=%d", gpw);
    const int64_t waves = (n_groups + gpw - 1) / gpw;
    run(nm, [&] { hipLaunchKernelGGL(group_read, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, J, n_groups, gpw, out); });
  }
  for (int tpw : {1, 4}) {
    char nm[64];
    snprintf(nm, 64, "tile_read tiles/wg=%d", tpw);
    run(nm, [&] { hipLaunchKernelGGL(tile_read, dim3((unsigned)((n_tiles + tpw - 1) / tpw)), dim3(256), 0, 0, J, n_tiles, tpw, out); });
  }
  return 0;
}
