// tools/sweep_bench.hip — A/B harness for Jacobian-sweep kernel variants (dev tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I calico_amd/csrc tools/sweep_bench.hip -o tools/_build/sweep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "rp_model.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args {
  int64_t n; const double* x; const double* intr; const float* pts;
  const double* u; const double* v; const int* corner; const int* fr;
  const int* fc; const int* fp; const int* ft; double* r; double* J;
};

// ---- V0: current product kernel (plane-major)
__global__ __launch_bounds__(256) void v0(Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
    const int f = a.fr[i]; const int c = a.fc[f], p = a.fp[f], t = a.ft[f]; const int j = a.corner[i];
    const double X[3] = {(double)a.pts[3*j], (double)a.pts[3*j+1], (double)a.pts[3*j+2]};
    double r[2], J[42];
    rp::eval_observation<true>(a.x + 7*(int64_t)c, p >= 0 ? a.x + 7*(int64_t)p : nullptr, t >= 0 ? a.x + 7*(int64_t)t : nullptr, a.intr + 12*(int64_t)c, X, a.u[i], a.v[i], r, J);
    a.r[i] = r[0]; a.r[a.n + i] = r[1];
#pragma unroll
    for (int k = 0; k < 42; ++k) a.J[(int64_t)k * a.n + i] = J[k];
  }
}
// ---- store-only ceilings
__global__ __launch_bounds__(256) void store_planes(Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
    const double val = a.u[i];
    a.r[i] = val; a.r[a.n + i] = val;
#pragma unroll
    for (int k = 0; k < 42; ++k) a.J[(int64_t)k * a.n + i] = val + k;
  }
}
template <int TILE>
__global__ __launch_bounds__(256) void store_tiled(Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
    const double val = a.u[i];
    const int64_t tile = i / TILE; const int off = (int)(i - tile * TILE);
    double* base = a.J + tile * (44 * TILE) + off;
#pragma unroll
    for (int k = 0; k < 44; ++k) base[k * TILE] = val + k;
  }
}
// ---- V1: tiled layout [tile][44 planes][TILE]
template <int TILE, int MINW>
__global__ __launch_bounds__(256, MINW) void v1(Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
    const int f = a.fr[i]; const int c = a.fc[f], p = a.fp[f], t = a.ft[f]; const int j = a.corner[i];
    const double X[3] = {(double)a.pts[3*j], (double)a.pts[3*j+1], (double)a.pts[3*j+2]};
    double r[2], J[42];
    rp::eval_observation<true>(a.x + 7*(int64_t)c, p >= 0 ? a.x + 7*(int64_t)p : nullptr, t >= 0 ? a.x + 7*(int64_t)t : nullptr, a.intr + 12*(int64_t)c, X, a.u[i], a.v[i], r, J);
    const int64_t tile = i / TILE; const int off = (int)(i - tile * TILE);
    double* base = a.J + tile * (44 * TILE) + off;
#pragma unroll
    for (int k = 0; k < 42; ++k) base[k * TILE] = J[k];
    base[42 * TILE] = r[0]; base[43 * TILE] = r[1];
  }
}

// ---- V2: per-FR table in LDS, tiled layout, stores through a sink
constexpr int MAXF = 32;
struct TiledSink { double* base; unsigned off; __device__ __forceinline__ void operator()(int k, double v) { base[(unsigned)(k * 256) + off] = v; } };
template <int MINW>
__global__ __launch_bounds__(256, MINW) void v2(Args a) {
  __shared__ __attribute__((aligned(16))) double tab[MAXF * rp::FRP_SIZE];
  const int tid = threadIdx.x;
  const int64_t ntiles = (a.n + 255) / 256;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t i0 = tile * 256, i = i0 + tid;
    const bool valid = i < a.n;
    const int f = valid ? a.fr[i] : -1;
    const int f_first = a.fr[i0];
    const int64_t ilast = (i0 + 255 < a.n) ? i0 + 255 : a.n - 1;
    const int f_last = a.fr[ilast];
    double u = 0, v = 0; int j = 0;
    if (valid) { u = a.u[i]; v = a.v[i]; j = a.corner[i]; }
    for (int fb = f_first; fb <= f_last; fb += MAXF) {
      const int nf = (f_last - fb + 1 < MAXF) ? (f_last - fb + 1) : MAXF;
      __syncthreads();
      if (tid < 4 * nf) {
        const int fi = tid >> 2, part = tid & 3, ff = fb + fi;
        double* e = tab + fi * rp::FRP_SIZE;
        const int c = a.fc[ff];
        if (part == 0) rp::prep_block(a.x + 7 * (int64_t)c, e + rp::FRP_CAM);
        else if (part == 1) { const int t = a.ft[ff]; rp::prep_block(t >= 0 ? a.x + 7 * (int64_t)t : nullptr, e + rp::FRP_TIM); }
        else if (part == 2) { const int p = a.fp[ff]; rp::prep_block(p >= 0 ? a.x + 7 * (int64_t)p : nullptr, e + rp::FRP_PAT); }
        else { for (int k = 0; k < 12; ++k) e[rp::FRP_K + k] = a.intr[12 * (int64_t)c + k]; }
      }
      __syncthreads();
      if (valid && f >= fb && f < fb + nf) {
        const double X[3] = {(double)a.pts[3*j], (double)a.pts[3*j+1], (double)a.pts[3*j+2]};
        double r[2];
        TiledSink sink{a.J + tile * (44 * 256), (unsigned)tid};
        rp::eval_prepared<true>(tab + (f - fb) * rp::FRP_SIZE, X, u, v, r, sink);
        sink(42, r[0]); sink(43, r[1]);
      }
    }
  }
}
// ---- V3: as V2 but stores through a buffer descriptor (one VGPR of address for all 44 stores)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct BufSink {
  __amdgpu_buffer_rsrc_t rs; int voff;
  __device__ __forceinline__ void operator()(int k, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, k * 2048, 0);
  }
};
template <int MINW, bool SKIPTAB = false>
__global__ __launch_bounds__(256, MINW) void v3(Args a) {
  __shared__ __attribute__((aligned(16))) double tab[MAXF * rp::FRP_SIZE];
  const int tid = threadIdx.x;
  const int64_t ntiles = (a.n + 255) / 256;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t i0 = tile * 256, i = i0 + tid;
    const bool valid = i < a.n;
    const int f = valid ? a.fr[i] : -1;
    const int f_first = a.fr[i0];
    const int64_t ilast = (i0 + 255 < a.n) ? i0 + 255 : a.n - 1;
    const int f_last = a.fr[ilast];
    double u = 0, v = 0; int j = 0;
    if (valid) { u = a.u[i]; v = a.v[i]; j = a.corner[i]; }
    BufSink sink{__builtin_amdgcn_make_buffer_rsrc(a.J + tile * (44 * 256), 0, 44 * 256 * 8, 0x00020000), tid * 8};
    for (int fb = f_first; fb <= f_last; fb += MAXF) {
      const int nf = (f_last - fb + 1 < MAXF) ? (f_last - fb + 1) : MAXF;
      if (!SKIPTAB || tile == blockIdx.x) {
      __syncthreads();
      if (tid < 4 * nf) {
        const int fi = tid >> 2, part = tid & 3, ff = fb + fi;
        double* e = tab + fi * rp::FRP_SIZE;
        const int c = a.fc[ff];
        if (part == 0) rp::prep_block(a.x + 7 * (int64_t)c, e + rp::FRP_CAM);
        else if (part == 1) { const int t = a.ft[ff]; rp::prep_block(t >= 0 ? a.x + 7 * (int64_t)t : nullptr, e + rp::FRP_TIM); }
        else if (part == 2) { const int p = a.fp[ff]; rp::prep_block(p >= 0 ? a.x + 7 * (int64_t)p : nullptr, e + rp::FRP_PAT); }
        else { for (int k = 0; k < 12; ++k) e[rp::FRP_K + k] = a.intr[12 * (int64_t)c + k]; }
      }
      __syncthreads();
      }
      if (valid && f >= fb && f < fb + nf) {
        const double X[3] = {(double)a.pts[3*j], (double)a.pts[3*j+1], (double)a.pts[3*j+2]};
        double r[2];
        rp::eval_prepared<true>(tab + (SKIPTAB ? ((f - fb) & 3) : (f - fb)) * rp::FRP_SIZE, X, u, v, r, sink);
        sink(42, r[0]); sink(43, r[1]);
      }
    }
  }
}
// ---- V4: G consecutive tiles per workgroup, one table build for all of them, next tile's
// observation data prefetched while the current tile is computed
template <int MINW, int G>
__global__ __launch_bounds__(256, MINW) void v4(Args a) {
  __shared__ __attribute__((aligned(16))) double tab[MAXF * rp::FRP_SIZE];
  const int tid = threadIdx.x;
  const int64_t tile0 = (int64_t)blockIdx.x * G;
  const int64_t i_first = tile0 * 256;
  if (i_first >= a.n) return;
  const int64_t i_end = (i_first + (int64_t)G * 256 < a.n) ? i_first + (int64_t)G * 256 : a.n;
  const int f_first = a.fr[i_first], f_last = a.fr[i_end - 1];
  for (int fb = f_first; fb <= f_last; fb += MAXF) {
    const int nf = (f_last - fb + 1 < MAXF) ? (f_last - fb + 1) : MAXF;
    __syncthreads();
    if (tid < 4 * nf) {
      const int fi = tid >> 2, part = tid & 3, ff = fb + fi;
      double* e = tab + fi * rp::FRP_SIZE;
      const int c = a.fc[ff];
      if (part == 0) rp::prep_block(a.x + 7 * (int64_t)c, e + rp::FRP_CAM);
      else if (part == 1) { const int t = a.ft[ff]; rp::prep_block(t >= 0 ? a.x + 7 * (int64_t)t : nullptr, e + rp::FRP_TIM); }
      else if (part == 2) { const int p = a.fp[ff]; rp::prep_block(p >= 0 ? a.x + 7 * (int64_t)p : nullptr, e + rp::FRP_PAT); }
      else { for (int k = 0; k < 12; ++k) e[rp::FRP_K + k] = a.intr[12 * (int64_t)c + k]; }
    }
    // first tile's data while the table is being built
    int64_t i = i_first + tid;
    int f = -1, j = 0; double u = 0, v = 0;
    if (i < i_end) { f = a.fr[i]; u = a.u[i]; v = a.v[i]; j = a.corner[i]; }
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      const int64_t tile = tile0 + g;
      const int64_t in = i + 256;
      int fn = -1, jn = 0; double un = 0, vn = 0;
      if (g + 1 < G && in < i_end) { fn = a.fr[in]; un = a.u[in]; vn = a.v[in]; jn = a.corner[in]; }
      if (f >= fb && f < fb + nf) {
        BufSink sink{__builtin_amdgcn_make_buffer_rsrc(a.J + tile * (44 * 256), 0, 44 * 256 * 8, 0x00020000), tid * 8};
        const double X[3] = {(double)a.pts[3*j], (double)a.pts[3*j+1], (double)a.pts[3*j+2]};
        double r[2];
        rp::eval_prepared<true>(tab + (f - fb) * rp::FRP_SIZE, X, u, v, r, sink);
        sink(42, r[0]); sink(43, r[1]);
      }
      i = in; f = fn; u = un; v = vn; j = jn;
    }
  }
}
// ---- ALU-only: full compute, one store per observation
__global__ __launch_bounds__(256) void alu_only(Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
    const int f = a.fr[i]; const int c = a.fc[f], p = a.fp[f], t = a.ft[f]; const int j = a.corner[i];
    const double X[3] = {(double)a.pts[3*j], (double)a.pts[3*j+1], (double)a.pts[3*j+2]};
    double r[2], J[42];
    rp::eval_observation<true>(a.x + 7*(int64_t)c, p >= 0 ? a.x + 7*(int64_t)p : nullptr, t >= 0 ? a.x + 7*(int64_t)t : nullptr, a.intr + 12*(int64_t)c, X, a.u[i], a.v[i], r, J);
    double s = r[0] + r[1];
#pragma unroll
    for (int k = 0; k < 42; ++k) s += J[k];
    a.r[i] = s;
  }
}
// ---- loads-only: the gather chain, no math
__global__ __launch_bounds__(256) void loads_only(Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
    const int f = a.fr[i]; const int c = a.fc[f], p = a.fp[f], t = a.ft[f]; const int j = a.corner[i];
    double s = a.pts[3*j] + a.pts[3*j+1] + a.pts[3*j+2] + a.u[i] + a.v[i];
    for (int k = 0; k < 7; ++k) s += a.x[7*(int64_t)c + k] + (p >= 0 ? a.x[7*(int64_t)p + k] : 0.0) + (t >= 0 ? a.x[7*(int64_t)t + k] : 0.0);
    for (int k = 0; k < 12; ++k) s += a.intr[12*(int64_t)c + k];
    a.r[i] = s;
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 40000000;
  const int per_fr = 80, cn = 128, pn = 8;
  const int64_t F = n / per_fr; const int tn = (int)(F / cn) + 1;
  const int vn = cn + pn + tn;
  std::vector<double> x(7 * (size_t)vn), intr(12 * cn), u(n), v(n);
  std::vector<float> pts(3 * 640);
  std::vector<int> corner(n), fr(n), fc(F), fp(F), ft(F);
  srand(1);
  auto rnd = []() { return rand() / (double)RAND_MAX; };
  for (int i = 0; i < vn; ++i) { double* q = &x[7*i]; q[0] = 1; q[1] = 0.1*rnd(); q[2] = 0.1*rnd(); q[3] = 0.1*rnd(); q[4] = 10*rnd(); q[5] = 10*rnd(); q[6] = (i < cn) ? 3000 : 10*rnd(); }
  for (int c = 0; c < cn; ++c) { double* K = &intr[12*c]; for (int k = 0; k < 12; ++k) K[k] = 0; K[0] = 1400; K[1] = 600; K[2] = 1400; K[3] = 450; K[4] = 0.01; K[5] = 0.01; }
  for (int j = 0; j < 640; ++j) { pts[3*j] = 45.f * (j % 10); pts[3*j+1] = 45.f * ((j / 10) % 8); pts[3*j+2] = 0; }
  for (int64_t f = 0; f < F; ++f) { fc[f] = f % cn; fp[f] = (f % pn == 0) ? -1 : cn + (int)(f % pn); ft[f] = (f / cn == 0) ? -1 : cn + pn + (int)(f / cn); }
  for (int64_t i = 0; i < n; ++i) { fr[i] = (int)std::min<int64_t>(i / per_fr, F - 1); corner[i] = (fr[i] % pn) * 80 + (int)(i % per_fr); u[i] = 600 + 100*rnd(); v[i] = 450 + 100*rnd(); }
  Args a; a.n = n;
  double *dx, *dintr, *du, *dv, *dr, *dJ; float* dpts; int *dcorner, *dfr, *dfc, *dfp, *dft;
  CK(hipMalloc(&dx, x.size()*8)); CK(hipMalloc(&dintr, intr.size()*8)); CK(hipMalloc(&du, n*8)); CK(hipMalloc(&dv, n*8));
  CK(hipMalloc(&dr, 2*n*8)); CK(hipMalloc(&dJ, (size_t)44*(n + 4096)*8)); CK(hipMalloc(&dpts, pts.size()*4));
  CK(hipMalloc(&dcorner, n*4)); CK(hipMalloc(&dfr, n*4)); CK(hipMalloc(&dfc, F*4)); CK(hipMalloc(&dfp, F*4)); CK(hipMalloc(&dft, F*4));
  CK(hipMemcpy(dx, x.data(), x.size()*8, hipMemcpyHostToDevice)); CK(hipMemcpy(dintr, intr.data(), intr.size()*8, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, u.data(), n*8, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, v.data(), n*8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpts, pts.data(), pts.size()*4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcorner, corner.data(), n*4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfr, fr.data(), n*4, hipMemcpyHostToDevice)); CK(hipMemcpy(dfc, fc.data(), F*4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfp, fp.data(), F*4, hipMemcpyHostToDevice)); CK(hipMemcpy(dft, ft.data(), F*4, hipMemcpyHostToDevice));
  a.x = dx; a.intr = dintr; a.pts = dpts; a.u = du; a.v = dv; a.corner = dcorner; a.fr = dfr; a.fc = dfc; a.fp = dfp; a.ft = dft; a.r = dr; a.J = dJ;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)n * 376.0;
  auto run = [&](const char* name, auto launch) {
    float best = 1e9f, tot = 0;
    for (int it = 0; it < 6; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) { tot += ms; if (ms < best) best = ms; }
    }
    CK(hipGetLastError());
    printf("%-28s avg %.3f ms  best %.3f ms  -> %.0f GB/s (376 B/obs)\n", name, tot / 5, best, bytes / (tot / 5 * 1e-3) / 1e9);
  };
  for (int g : {2048, 4096, 8192}) {
    char nm[64]; snprintf(nm, 64, "v0 planes grid=%d", g);
    run(nm, [&] { hipLaunchKernelGGL(v0, dim3(g), dim3(256), 0, 0, a); });
  }
  run("alu_only 2048", [&] { hipLaunchKernelGGL(alu_only, dim3(2048), dim3(256), 0, 0, a); });
  run("loads_only 2048", [&] { hipLaunchKernelGGL(loads_only, dim3(2048), dim3(256), 0, 0, a); });
  run("store_planes 2048", [&] { hipLaunchKernelGGL(store_planes, dim3(2048), dim3(256), 0, 0, a); });
  run("store_tiled<256> 2048", [&] { hipLaunchKernelGGL(store_tiled<256>, dim3(2048), dim3(256), 0, 0, a); });
  run("store_tiled<64> 2048", [&] { hipLaunchKernelGGL(store_tiled<64>, dim3(2048), dim3(256), 0, 0, a); });
  run("store_tiled<256> full", [&] { hipLaunchKernelGGL(store_tiled<256>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v1 tiled<256> w1 2048", [&] { hipLaunchKernelGGL((v1<256, 1>), dim3(2048), dim3(256), 0, 0, a); });
  run("v1 tiled<256> w1 full", [&] { hipLaunchKernelGGL((v1<256, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v1 tiled<64> w1 2048", [&] { hipLaunchKernelGGL((v1<64, 1>), dim3(2048), dim3(256), 0, 0, a); });
  run("v1 tiled<256> w3 2048", [&] { hipLaunchKernelGGL((v1<256, 3>), dim3(2048), dim3(256), 0, 0, a); });
  run("v1 tiled<256> w4 2048", [&] { hipLaunchKernelGGL((v1<256, 4>), dim3(2048), dim3(256), 0, 0, a); });
  run("v1 tiled<256> w4 full", [&] { hipLaunchKernelGGL((v1<256, 4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v2 lds-table w1 2048", [&] { hipLaunchKernelGGL((v2<1>), dim3(2048), dim3(256), 0, 0, a); });
  run("v2 lds-table w1 full", [&] { hipLaunchKernelGGL((v2<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v2 lds-table w4 2048", [&] { hipLaunchKernelGGL((v2<4>), dim3(2048), dim3(256), 0, 0, a); });
  run("v2 lds-table w4 full", [&] { hipLaunchKernelGGL((v2<4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v2 lds-table w3 full", [&] { hipLaunchKernelGGL((v2<3>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v2 lds-table w6 full", [&] { hipLaunchKernelGGL((v2<6>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v2 lds-table w8 full", [&] { hipLaunchKernelGGL((v2<8>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v3 bufstore w1 full", [&] { hipLaunchKernelGGL((v3<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v3 bufstore w3 full", [&] { hipLaunchKernelGGL((v3<3>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v3 SKIPTAB w3 2048", [&] { hipLaunchKernelGGL((v3<3, true>), dim3(2048), dim3(256), 0, 0, a); });
  run("v3 SKIPTAB w3 768", [&] { hipLaunchKernelGGL((v3<3, true>), dim3(768), dim3(256), 0, 0, a); });
  run("v3 w3 768", [&] { hipLaunchKernelGGL((v3<3>), dim3(768), dim3(256), 0, 0, a); });
  run("v3 w3 1536", [&] { hipLaunchKernelGGL((v3<3>), dim3(1536), dim3(256), 0, 0, a); });
  run("v4 G=2 w3", [&] { hipLaunchKernelGGL((v4<3, 2>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, a); });
  run("v4 G=4 w3", [&] { hipLaunchKernelGGL((v4<3, 4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a); });
  run("v4 G=8 w3", [&] { hipLaunchKernelGGL((v4<3, 8>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, a); });
  run("v3 bufstore w4 full", [&] { hipLaunchKernelGGL((v3<4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v3 bufstore w4 2048", [&] { hipLaunchKernelGGL((v3<4>), dim3(2048), dim3(256), 0, 0, a); });
  run("v3 bufstore w5 full", [&] { hipLaunchKernelGGL((v3<5>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  run("v3 bufstore w6 full", [&] { hipLaunchKernelGGL((v3<6>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a); });
  return 0;
}
