// Stand-alone micro-benchmark of the row-panel GEMM kernels (random data, HIP events).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I style_transfer_amd/csrc tools/kbench.cpp -o build_variants/kbench [-D...]
// run:   kbench <rows> <generator splits, 0 = auto> <gemm_format 0|1> <kernel letters: g s f q o 2 u t>
// -D switches (ablations, never in the library build): ST_ABL_NOSTORE (skip output stores), ST_NO_NT (temporal
// stores), ST_TAIL_STAMPS (cycle stamps of the half-layer kernel's phases, printed after its timing)
#define ST_RP_DEFINE_SWITCHES
#include "st_rowpanel.h"

#include <stdarg.h>
#include <stdlib.h>
#include <vector>

void st_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fprintf(stderr, "\n");
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((s >> 8) / 16777216.0f * 2.0f - 1.0f) * scale;
  }
  float* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}
static int g_fmt = ST_GEMM_F32;
static float* pack(const float* W, int N, int K) {
  float* out;
  CK(hipMalloc(&out, rp_packed_floats(N, K) * 4));
  if (g_fmt == ST_GEMM_F16X3) {
    size_t total = (size_t)((N + 31) / 32) * (K >> 4) * 64;
    hipLaunchKernelGGL(pack_weight_h_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, W, N, K, out);
  } else {
    size_t total = (size_t)((N + 31) / 32) * (K >> 3) * 64;
    hipLaunchKernelGGL(pack_weight_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, W, N, K, out);
  }
  CK(hipDeviceSynchronize());
  return out;
}
template <class F>
static double timeit(F f, int iters = 10) {
  for (int i = 0; i < 3; ++i) f();
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e-3 / iters;
}

// Sustained MFMA rate of the chip as it runs here (power-managed clock): nothing but f16 MFMAs on register operands,
// RANDOM operand bits (trivial operands draw less power and hold a higher clock), `waves` waves per SIMD.
// SHAPE 0: v_mfma_f32_32x32x16_f16, 2 accumulators;  SHAPE 1: v_mfma_f32_16x16x32_f16, 8 accumulators (same 32x32x32... flops per iteration).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_peak_kernel(float* out, const float* seed, int iters) {
  h8 a[4], b[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      a[q][i] = (_Float16)seed[(threadIdx.x * 37 + q * 8 + i) & 4095];
      b[q][i] = (_Float16)seed[(threadIdx.x * 53 + blockIdx.x + q * 8 + i + 1000) & 4095];
    }
  float s = 0;
  if (SHAPE == 0) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {  // 4 x 32x32x16 = 131072 flop
      c0 = mfma16h(a[0], b[0], c0);
      c1 = mfma16h(a[1], b[1], c1);
      c2 = mfma16h(a[2], b[2], c2);
      c3 = mfma16h(a[3], b[3], c3);
    }
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  } else {
    f32x4v c[8] = {};
    for (int it = 0; it < iters; ++it) {  // 8 x 16x16x32 = 131072 flop
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q & 3], b[(q + (q >> 2)) & 3], c[q], 0, 0, 0);
    }
    for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
  }
  if (s == 1.2345f) out[threadIdx.x] = s;
}

// Do VALU operations hide under MFMAs?  The 16x16x32 loop above with NV independent v_max3_f32 per 8 MFMAs (8 separate
// registers, no dependence on the MFMAs), random operands.
template <int NV>
__global__ __launch_bounds__(512) void mfma_valu_kernel(float* out, const float* seed, int iters) {
  h8 a[4], b[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      a[q][i] = (_Float16)seed[(threadIdx.x * 37 + q * 8 + i) & 4095];
      b[q][i] = (_Float16)seed[(threadIdx.x * 53 + blockIdx.x + q * 8 + i + 1000) & 4095];
    }
  float k[8], u = seed[threadIdx.x & 4095], w = seed[(threadIdx.x + 77) & 4095];
  for (int j = 0; j < 8; ++j) k[j] = seed[(threadIdx.x + j) & 4095];
  f32x4v c[8] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q & 3], b[(q + (q >> 2)) & 3], c[q], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 8; ++j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(k[(q + j) & 7]) : "v"(u), "v"(w));
    }
  }
  float s = 0;
  for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3] + k[q];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

// Memory floor of the QKV projection: read a [n,256] f32 matrix once, write a [n,768] one (nontemporal 16-byte stores),
// no arithmetic.  One workgroup per 64-row panel like the real kernels.
__global__ __launch_bounds__(256) void qkv_floor_kernel(const float* x, float* out, int n) {
  const int m0 = blockIdx.x * 64, tid = threadIdx.x;
  f32x4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = tid + i * 256, r = f >> 6, c4 = f & 63;
    v[i] = *(const f32x4*)(x + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = tid + i * 256, r = f >> 6, c4 = f & 63;
    if (m0 + r >= n) continue;
    float* o = out + (size_t)(m0 + r) * 768 + c4 * 4;
    __builtin_nontemporal_store(v[i], (f32x4*)o);
    __builtin_nontemporal_store(v[i], (f32x4*)(o + 256));
    __builtin_nontemporal_store(v[i], (f32x4*)(o + 512));
  }
}

// Pure-read floor: every wave streams 16-byte loads over a large f32 array and keeps a sum (nothing stored).
template <int UNROLL>
__global__ __launch_bounds__(256) void read_floor_kernel(const float* x, size_t n4, float* out) {
  const f32x4* p = (const f32x4*)x;
  f32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[threadIdx.x] = acc.x;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 65536;
  g_fmt = argc > 3 ? atoi(argv[3]) : ST_GEMM_F32;  // kbench <rows> <generator splits, 0 = auto> <format>
  const char* only = argc > 4 ? argv[4] : "gsfqo2ut";  // which kernels: g f q o 2 u
  auto want = [&](char c) { return strchr(only, c) != nullptr; };
  printf("gemm_format %d\n", g_fmt);
  const int V = 9339, DFF = 1024;
  float* x = dev_rand((size_t)n * 256, 1.0f, 1);
  float* att = dev_rand((size_t)n * 256, 1.0f, 2);
  float* h;
  CK(hipMalloc(&h, (size_t)n * DFF * 4));
  float* q;
  CK(hipMalloc(&q, (size_t)n * 768 * 4));
  float* ln_a = dev_rand(256, 1.0f, 3);
  float* ln_b = dev_rand(256, 0.1f, 4);
  float* wo = dev_rand(256 * 256, 0.06f, 5);
  float* w1 = dev_rand((size_t)DFF * 256, 0.06f, 6);
  float* w2 = dev_rand((size_t)256 * DFF, 0.03f, 7);
  float* wqkv = dev_rand((size_t)768 * 256, 0.06f, 8);
  float* wg = dev_rand((size_t)V * 256, 0.06f, 9);
  float* bias = dev_rand(16384, 0.1f, 10);
  float *wo_p = pack(wo, 256, 256), *w1_p = pack(w1, DFF, 256), *w2_p = pack(w2, 256, DFF), *wqkv_p = pack(wqkv, 768, 256),
        *wg_p = pack(wg, V, 256);
  float4* partial;
  CK(hipMalloc(&partial, (size_t)n * 16 * 16));
  hipStream_t st = 0;
  double t;

  if (want('p')) {  // sustained MFMA peak, both f16 shapes, random operands
    float* seed = dev_rand(4096, 2.0f, 77);
    for (int rep = 0; rep < 2; ++rep)
      for (int shape = 0; shape < 2; ++shape)
        for (int wg_threads : {256, 512}) {
          const int iters = 20000;
          t = timeit([&] {
            if (shape == 0)
              hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(256 * 4), dim3(wg_threads), 0, st, x, seed, iters);
            else
              hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(256 * 4), dim3(wg_threads), 0, st, x, seed, iters);
          }, 5);
          const double fl = 131072.0 * iters * (wg_threads / 64) * 256 * 4;
          printf("bare %s f16 MFMA loop, random operands, %d waves per SIMD: %8.1f us  %7.1f TF  (= %.2f GHz x 1024 flop/clk/SIMD x 1024 SIMDs)\n",
                 shape ? "16x16x32" : "32x32x16", wg_threads / 256, t * 1e6, fl / t / 1e12, fl / t / 1024.0 / 1024.0 / 1e9);
        }
  }
  if (want('v')) {  // VALU operations between MFMAs: hidden or added?
    float* seed = dev_rand(4096, 2.0f, 77);
    for (int wg_threads : {256, 512}) {
      const int iters = 20000;
      double tv[5];
      tv[0] = timeit([&] { hipLaunchKernelGGL(mfma_valu_kernel<0>, dim3(1024), dim3(wg_threads), 0, st, x, seed, iters); }, 5);
      tv[1] = timeit([&] { hipLaunchKernelGGL(mfma_valu_kernel<8>, dim3(1024), dim3(wg_threads), 0, st, x, seed, iters); }, 5);
      tv[2] = timeit([&] { hipLaunchKernelGGL(mfma_valu_kernel<16>, dim3(1024), dim3(wg_threads), 0, st, x, seed, iters); }, 5);
      tv[3] = timeit([&] { hipLaunchKernelGGL(mfma_valu_kernel<24>, dim3(1024), dim3(wg_threads), 0, st, x, seed, iters); }, 5);
      tv[4] = timeit([&] { hipLaunchKernelGGL(mfma_valu_kernel<32>, dim3(1024), dim3(wg_threads), 0, st, x, seed, iters); }, 5);
      const double fl = 131072.0 * iters * (wg_threads / 64) * 1024;
      for (int j = 0; j < 5; ++j)
        printf("16x16x32 MFMA loop + %d v_max3_f32 per MFMA, %d waves per SIMD: %8.1f us  %7.1f TF  (x %.3f)\n", j, wg_threads / 256,
               tv[j] * 1e6, fl / tv[j] / 1e12, tv[j] / tv[0]);
    }
  }
  if (want('r')) {  // pure-read floor over the q output buffer ([n,768] f32)
    const size_t n4 = (size_t)n * 768 / 4;
    for (int wgs : {1024, 2048, 4096, 8192}) {
      t = timeit([&] { hipLaunchKernelGGL(read_floor_kernel<8>, dim3(wgs), dim3(256), 0, st, q, n4, x); }, 10);
      printf("pure read, %d workgroups x 256 threads, 8 x 16 B in flight per thread, %.2f GB: %8.1f us  %.2f TB/s\n", wgs, n4 * 16 / 1e9, t * 1e6, n4 * 16.0 / t / 1e12);
    }
  }
  if (want('m')) {  // memory floor of the QKV projection
    t = timeit([&] { hipLaunchKernelGGL(qkv_floor_kernel, dim3((n + 63) / 64), dim3(256), 0, st, x, q, n); }, 10);
    printf("qkv memory floor (read [n,256], write [n,768], no arithmetic) n=%d: %8.1f us  %.2f TB/s\n", n, t * 1e6, (double)n * 4096 / t / 1e12);
  }
  if (want('g')) {  // generator (exact argmax); argv[5] = rows of the launch if not n (the redo pass sees few rows and many splits)
    GemmArgs g = {};
    g.fmt = g_fmt;
    const int ng = argc > 5 ? atoi(argv[5]) : n;
    g.X = x; g.n_host = ng; g.ln_a = ln_a; g.ln_b = ln_b; g.W = wg_p; g.bias = bias; g.N = V; g.partial = partial;
    const int sp = (argc > 2 && atoi(argv[2]) > 0) ? atoi(argv[2]) : rp_splits(ng, V, 16);
    t = timeit([&] { launch_rp<EPI_ARGMAX>(g, ng, sp, st); });
    printf("generator  n=%d splits=%d: %8.1f us %6.1f TF\n", ng, sp, t * 1e6, 2.0 * ng * 256 * V / t / 1e12);
    if (g_fmt == ST_GEMM_F16X3) {
      std::vector<float4> pa((size_t)ng * sp), pb(pa.size());
      CK(hipMemcpy(pa.data(), partial, pa.size() * 16, hipMemcpyDeviceToHost));
      float* colpad;
      CK(hipMalloc(&colpad, rp_screen_colpad_floats(V) * 4));
      launch_rp_screen_colpad(g.bias, nullptr, V, colpad, st);
      g.colpad = colpad;
      const double t16 = timeit([&] { launch_rp<EPI_ARGMAX>(g, ng, sp, st); });
      printf("generator 16x16x32 n=%d splits=%d: %8.1f us %6.1f TF  (x %.3f)\n", ng, sp, t16 * 1e6, 2.0 * ng * 256 * V / t16 / 1e12, t16 / t);
      CK(hipMemcpy(pb.data(), partial, pb.size() * 16, hipMemcpyDeviceToHost));
      size_t di = 0;
      double dm = 0;
      for (size_t i = 0; i < pa.size(); ++i) {
        di += memcmp(&pa[i].y, &pb[i].y, 4) != 0;
        dm = fmax(dm, fmax(fabs((double)pa[i].x - pb[i].x), fabs((double)pa[i].z - pb[i].z)));
      }
      printf("  partials against the 32x32x16 form: %zu of %zu best columns differ, max |best / second difference| %.3g\n", di, pa.size(), dm);
    }
  }
  if (want('s') && g_fmt == ST_GEMM_F16X3) {  // generator screening pass (one product)
    GemmArgs g = {};
    g.fmt = g_fmt;
    g.X = x; g.n_host = n; g.ln_a = ln_a; g.ln_b = ln_b; g.W = wg_p; g.bias = bias; g.N = V; g.partial = partial;
    if (getenv("KBENCH_UKEYS")) {  // the production form: keys are upper bounds s + scale |x| |W_v|
      g.wnorm = dev_rand(V, 0.7f, 77);  // the kernels only need SOME per-column factor here
      g.scale = 1.25f * (1.0f / 1024 + 1.0f / 65536);
    }
    float* colpad;
    CK(hipMalloc(&colpad, rp_screen_colpad_floats(V) * 4));
    launch_rp_screen_colpad(g.bias, g.wnorm, V, colpad, st);
    g.colpad = colpad;
    const int sp = (argc > 2 && atoi(argv[2]) > 0) ? atoi(argv[2]) : rp_splits(n, V, 16);
    double t16p = 0, t16 = 0, t32x = 0;
    for (int rep = 0; rep < 3; ++rep) {
      rp_screen_shape16 = 1; rp_screen_paired = 1;
      t16p += timeit([&] { launch_rp_screen_pipelined(g, n, sp, st); }, 5);
      rp_screen_paired = 0;
      t16 += timeit([&] { launch_rp_screen_pipelined(g, n, sp, st); }, 5);
      rp_screen_shape16 = 0;
      t32x += timeit([&] { launch_rp_screen_pipelined(g, n, sp, st); }, 5);
    }
    rp_screen_paired = 1;
    const double fl = 2.0 * n * 256 * V;
    printf("gen screen n=%d splits=%d (one product): 16x16x32 paired keys %8.1f us %6.1f TF   16x16x32 %8.1f us %6.1f TF   32x32x16 %8.1f us %6.1f TF\n",
           n, sp, t16p / 3 * 1e6, fl / (t16p / 3) / 1e12, t16 / 3 * 1e6, fl / (t16 / 3) / 1e12, t32x / 3 * 1e6, fl / (t32x / 3) / 1e12);
    std::vector<float4> pa((size_t)n * sp), pb(pa.size());
    rp_screen_shape16 = 1; rp_screen_paired = 1;
    launch_rp_screen_pipelined(g, n, sp, st);
    CK(hipMemcpy(pa.data(), partial, pa.size() * 16, hipMemcpyDeviceToHost));
    rp_screen_paired = 0;  // reference: the unpaired 16x16x32 form (8-bit keys)
    launch_rp_screen_pipelined(g, n, sp, st);
    CK(hipMemcpy(pb.data(), partial, pb.size() * 16, hipMemcpyDeviceToHost));
    rp_screen_paired = 1;
    size_t di = 0;
    double dm = 0;
    for (size_t i = 0; i < pa.size(); ++i) {
      di += memcmp(&pa[i].y, &pb[i].y, 4) != 0;
      dm = fmax(dm, fmax(fabs((double)pa[i].x - pb[i].x), fabs((double)pa[i].z - pb[i].z)) / fmax(fabs((double)pb[i].x), fabs((double)pb[i].z)));
    }
    printf("  partials, paired against unpaired keys: %zu of %zu best columns differ, max relative |best / second difference| %.3g (key grid 2^-14 = 6.1e-5)\n", di, pa.size(), dm);
  }
  if (want('f')) {  // FFN1 (rp RELU)
    GemmArgs g = {};
    g.fmt = g_fmt;
    g.X = x; g.n_host = n; g.ln_a = ln_a; g.ln_b = ln_b; g.W = w1_p; g.bias = bias; g.N = DFF; g.out = h; g.ldo = DFF;
    const int sp = rp_splits(n, DFF, 8);
    t = timeit([&] { launch_rp<EPI_RELU>(g, n, sp, st); });
    printf("ffn1       n=%d splits=%d: %8.1f us %6.1f TF\n", n, sp, t * 1e6, 2.0 * n * 256 * DFF / t / 1e12);
    t = timeit([&] { launch_rp<EPI_RELU, 8>(g, n, sp, st); });
    printf("ffn1 NW=8  n=%d splits=%d: %8.1f us %6.1f TF\n", n, sp, t * 1e6, 2.0 * n * 256 * DFF / t / 1e12);
  }
  if (want('q')) {  // QKV as plain BIAS store to [n,768]
    GemmArgs g = {};
    g.fmt = g_fmt;
    g.X = x; g.n_host = n; g.ln_a = ln_a; g.ln_b = ln_b; g.W = wqkv_p; g.bias = bias; g.N = 768; g.out = q; g.ldo = 768;
    if (getenv("KBENCH_NOLN")) g.ln_a = g.ln_b = nullptr;  // ablation: no LayerNorm in the panel prologue
    const int sp = rp_splits(n, 768, 6);
    t = timeit([&] { launch_rp<EPI_BIAS>(g, n, sp, st); });
    printf("qkv(bias)  n=%d splits=%d: %8.1f us %6.1f TF\n", n, sp, t * 1e6, 2.0 * n * 256 * 768 / t / 1e12);
    t = timeit([&] { launch_rp<EPI_BIAS, 8>(g, n, sp, st); });
    printf("qkv NW=8   n=%d splits=%d: %8.1f us %6.1f TF\n", n, sp, t * 1e6, 2.0 * n * 256 * 768 / t / 1e12);
    if (g_fmt == ST_GEMM_F16X3) {
      double t16 = 0, t32x = 0;
      for (int rep = 0; rep < 3; ++rep) {
        rp_lin_shape16 = 1;
        t16 += timeit([&] { launch_rp_lin_pipelined<EPI_BIAS>(g, n, sp, st); }, 5);
        rp_lin_shape16 = 0;
        t32x += timeit([&] { launch_rp_lin_pipelined<EPI_BIAS>(g, n, sp, st); }, 5);
      }
      printf("qkv pipelined stores n=%d: 16x16x32 %8.1f us %6.1f TF   32x32x16 %8.1f us %6.1f TF  (%.3f)\n", n, t16 / 3 * 1e6,
             2.0 * n * 256 * 768 / (t16 / 3) / 1e12, t32x / 3 * 1e6, 2.0 * n * 256 * 768 / (t32x / 3) / 1e12, t16 / t32x);
      std::vector<float> ya((size_t)n * 768), yb(ya.size());
      rp_lin_shape16 = 1;
      launch_rp_lin_pipelined<EPI_BIAS>(g, n, sp, st);
      CK(hipMemcpy(ya.data(), q, ya.size() * 4, hipMemcpyDeviceToHost));
      rp_lin_shape16 = 0;
      launch_rp_lin_pipelined<EPI_BIAS>(g, n, sp, st);
      CK(hipMemcpy(yb.data(), q, ya.size() * 4, hipMemcpyDeviceToHost));
      rp_lin_shape16 = 1;
      double mx = 0, ref = 0;
      for (size_t i = 0; i < ya.size(); ++i) {
        mx = fmax(mx, fabs((double)ya[i] - (double)yb[i]));
        ref = fmax(ref, fabs((double)yb[i]));
      }
      printf("  outputs: max |diff| %.3g against max |y| %.3g\n", mx, ref);
    }
  }
  if (want('o')) {  // O-proj RESID
    GemmArgs g = {};
    g.fmt = g_fmt;
    g.X = att; g.n_host = n; g.W = wo_p; g.bias = bias; g.N = 256; g.out = x; g.ldo = 256; g.resid = x;
    const int sp = rp_splits(n, 256, 2);
    t = timeit([&] { launch_rp<EPI_RESID>(g, n, sp, st); });
    printf("oproj      n=%d splits=%d: %8.1f us %6.1f TF\n", n, sp, t * 1e6, 2.0 * n * 256 * 256 / t / 1e12);
  }
  if (want('2')) {  // FFN2 kchunk
    GemmArgs g = {};
    g.fmt = g_fmt;
    g.X = h; g.n_host = n; g.K = DFF; g.W = w2_p; g.bias = bias; g.N = 256; g.out = x; g.ldo = 256; g.resid = x;
    t = timeit([&] { launch_rp_kchunk<EPI_RESID>(g, n, st); });
    printf("ffn2       n=%d: %8.1f us %6.1f TF\n", n, t * 1e6, 2.0 * n * 256 * DFF / t / 1e12);
  }
  if (want('u')) {  // fused O + FFN1
    OFArgs a = {};
    a.fmt = g_fmt;
    a.att = att; a.n_host = n; a.wo = wo_p; a.bo = bias; a.x = x; a.ln_a = ln_a; a.ln_b = ln_b; a.w1 = w1_p; a.b1 = bias;
    a.h = h; a.d_ff = DFF;
    t = timeit([&] { launch_oproj_ffn1(a, n, st); });
    printf("oproj+ffn1 n=%d: %8.1f us %6.1f TF\n", n, t * 1e6, 2.0 * n * 256 * (256 + DFF) / t / 1e12);
  }
  if (want('t') && g_fmt == ST_GEMM_F16X3) {  // whole post-attention half-layer
    OFArgs a = {};
    a.fmt = g_fmt;
    a.att = att; a.n_host = n; a.wo = wo_p; a.bo = bias; a.x = x; a.ln_a = ln_a; a.ln_b = ln_b; a.w1 = w1_p; a.b1 = bias;
    a.w2 = w2_p; a.b2 = bias; a.d_ff = DFF;
    t = timeit([&] { launch_layer_tail64(a, n, st); });
    printf("layer tail n=%d: %8.1f us %6.1f TF\n", n, t * 1e6, 2.0 * n * 256 * (256 + 2 * DFF) / t / 1e12);
    {
      // A/B: persistent (one workgroup per CU, next panel prefetched) vs one workgroup per panel; bitwise comparison
      double tp = 0, t1 = 0;
      for (int rep = 0; rep < 3; ++rep) {
        rp_tail_persistent = 1;
        tp += timeit([&] { launch_layer_tail64(a, n, st); }, 5);
        rp_tail_persistent = 0;
        t1 += timeit([&] { launch_layer_tail64(a, n, st); }, 5);
      }
      printf("layer tail A/B n=%d: persistent %8.1f us   one panel per workgroup %8.1f us  (%.3f)\n", n, tp / 3 * 1e6, t1 / 3 * 1e6, tp / t1);
      std::vector<float> x0((size_t)n * 256), yp(x0.size()), y1(x0.size());
      float* xs = dev_rand((size_t)n * 256, 1.0f, 1);
      CK(hipMemcpy(x0.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      OFArgs b = a;
      b.x = xs;
      rp_tail_persistent = 1;
      launch_layer_tail64(b, n, st);
      CK(hipMemcpy(yp.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(xs, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
      rp_tail_persistent = 0;
      launch_layer_tail64(b, n, st);
      CK(hipMemcpy(y1.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      rp_tail_persistent = 1;
      size_t diff = 0;
      for (size_t i = 0; i < x0.size(); ++i) diff += memcmp(&yp[i], &y1[i], 4) != 0;
      printf("  outputs: %zu of %zu elements differ bitwise\n", diff, x0.size());
      CK(hipFree(xs));
    }
    {
      // A/B: 16x16x32 tiles vs 32x32x16 tiles (both persistent); outputs compared by value (the MFMA-internal summation order differs)
      double t16 = 0, t32x = 0;
      for (int rep = 0; rep < 3; ++rep) {
        t16 += timeit([&] { launch_layer_tail16<0>(a, n, st); }, 5);
        t32x += timeit([&] { launch_layer_tail64(a, n, st); }, 5);
      }
      printf("layer tail A/B n=%d: 16x16x32 tiles %8.1f us   32x32x16 tiles %8.1f us  (%.3f)\n", n, t16 / 3 * 1e6, t32x / 3 * 1e6, t16 / t32x);
      std::vector<float> x0((size_t)n * 256), ya(x0.size()), yb(x0.size());
      float* xs = dev_rand((size_t)n * 256, 1.0f, 1);
      CK(hipMemcpy(x0.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      OFArgs b = a;
      b.x = xs;
      launch_layer_tail16<0>(b, n, st);
      CK(hipMemcpy(ya.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(xs, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
      launch_layer_tail64(b, n, st);
      CK(hipMemcpy(yb.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      double mx = 0, ref = 0;
      size_t diff = 0;
      for (size_t i = 0; i < x0.size(); ++i) {
        mx = fmax(mx, fabs((double)ya[i] - (double)yb[i]));
        ref = fmax(ref, fabs((double)yb[i]));
        diff += memcmp(&ya[i], &yb[i], 4) != 0;
      }
      printf("  outputs: max |diff| %.3g against max |y| %.3g, %zu of %zu elements differ bitwise; samples %g %g | %g %g\n", mx, ref, diff, x0.size(),
             ya[12345], yb[12345], ya[x0.size() - 7], yb[x0.size() - 7]);
      CK(hipFree(xs));
    }
    if (getenv("KBENCH_TAIL32")) {
      // A/B: the 32-row / 4-wave form (two workgroups per CU), interleaved rounds, and a bitwise comparison of the outputs
      double t64 = 0, t32 = 0;
      for (int rep = 0; rep < 3; ++rep) {
        t64 += timeit([&] { launch_layer_tail64(a, n, st); }, 5);
        t32 += timeit([&] { launch_layer_tail32(a, n, st); }, 5);
      }
      printf("layer tail A/B n=%d: 64-row x 8 waves %8.1f us   32-row x 4 waves %8.1f us  (%.3f)\n", n, t64 / 3 * 1e6, t32 / 3 * 1e6, t32 / t64);
      std::vector<float> x0((size_t)n * 256), y64(x0.size()), y32(x0.size());
      float* xs = dev_rand((size_t)n * 256, 1.0f, 1);
      CK(hipMemcpy(x0.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      OFArgs b = a;
      b.x = xs;
      launch_layer_tail64(b, n, st);
      CK(hipMemcpy(y64.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(xs, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
      launch_layer_tail32(b, n, st);
      CK(hipMemcpy(y32.data(), xs, x0.size() * 4, hipMemcpyDeviceToHost));
      size_t diff = 0;
      double mx = 0;
      for (size_t i = 0; i < x0.size(); ++i) {
        if (memcmp(&y64[i], &y32[i], 4)) ++diff;
        const double d = fabs((double)y64[i] - (double)y32[i]);
        if (d > mx) mx = d;
      }
      printf("  outputs: %zu of %zu elements differ bitwise, max |diff| %.3g, sample y64 %g y32 %g\n", diff, x0.size(), mx, y64[12345], y32[12345]);
    }
#ifdef ST_TAIL_STAMPS
    rp_tail_persistent = getenv("KBENCH_ONE_PANEL") ? 0 : 1;
    rp_tail_shape16 = getenv("KBENCH_SHAPE32") ? 0 : 1;
    launch_layer_tail<0>(a, n, st);
    CK(hipDeviceSynchronize());
    rp_tail_persistent = 1;
    unsigned long long hs[64];
    CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_tail_stamps), sizeof(hs)));
    const char* nm[] = {"entry -> loads issued", "wait + barrier0", "O-proj", "barrier1", "x1->LDS + LN + barriers", "bias init", "FFN1(0)"};
    printf("  (stamps of the LAST panel of workgroup 100)  kernel entry -> this panel's loop top %llu\n", hs[35] - hs[31]);
    printf("  %-32s %8llu\n", "loop top -> ring + acc init", hs[0] - hs[35]);
    for (int i = 1; i <= 6; ++i) printf("  %-32s %8llu\n", nm[i], hs[i] - hs[i - 1]);
    printf("    of which: x1 -> LDS %llu, barrier %llu, LayerNorm %llu, barrier %llu\n", hs[32] - hs[3], hs[33] - hs[32], hs[34] - hs[33], hs[4] - hs[34]);
    for (int c = 0; c < 4; ++c) {
      const unsigned long long prev = c ? hs[9 + 5 * (c - 1)] : hs[6];
      printf("  chunk %d: FFN1(c+1) + convert(c) %6llu  barrier %6llu  FFN2 %6llu\n", c, hs[7 + 5 * c] - prev,
             hs[8 + 5 * c] - hs[7 + 5 * c], hs[9 + 5 * c] - hs[8 + 5 * c]);
    }
    printf("  store %llu   panel total %llu   (previous panel: att_store after its store %llu)\n", hs[30] - hs[24], hs[30] - hs[35], hs[35] - hs[36]);
#endif
  }
  CK(hipDeviceSynchronize());
  return 0;
}
