// f32-MFMA GEMM building blocks for the decoder:  C[n,N] = op(X[n,K]) . W[N,K]^T (+ epilogue)
//
//  * panel_gemm (K == 256):  one 512-thread workgroup owns a 64-row panel of X, resident in LDS for
//    the whole launch (optionally LayerNorm'ed in place), and streams 256-column tiles of W through a
//    double-buffered LDS stage.  8 waves, wave w owns columns [32w,32w+32) of the tile and all 64
//    rows (two 32x32 accumulators).  W bytes per MFMA cycle: 8 B/clk/CU (512/BM), well under L2 rate.
//  * kloop_gemm (any K % 32 == 0): classic 128x128 tile, both operands streamed, 256 threads.
//
// MFMA operand trick: v_mfma_f32_32x32x2_f32 wants, per lane, ONE k of A and ONE k of B (lane half
// h = lane>>5 selects k in {0,1}).  Each lane instead reads 4 consecutive k (one ds_read_b128) at
// k = 8*kg + 4*h + {0..3}; MFMA step s then pairs k = 8kg+s (h=0) with k = 8kg+4+s (h=1).  A and B
// use the same permutation, so the sum over k is complete; only the summation order differs.
// LDS rows are padded to a stride = 4 (mod 64) dwords, which makes those b128 reads conflict-free.
#pragma once
#include "st_common.h"

#define ST_LDA 260  // panel row stride in floats (256 + 4)
#define ST_LDW 36   // W-stage row stride in floats (32 + 4)

enum { EPI_BIAS = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_QKV = 3, EPI_ARGMAX = 4, EPI_LSE = 5, EPI_DS = 6 };

struct GemmArgs {
  const float* X;      // [n, K] row-major
  const int* n_dev;    // device row count (may be null -> n_host)
  int n_host;
  int K;               // kloop only (panel: 256)
  int ldx, ldw;        // kloop with AT / WT only
  const float* ln_a;   // panel only: LayerNorm gain/bias applied to X rows (null = none)
  const float* ln_b;
  const float* W;      // [N, K] row-major (nn.Linear layout)
  const float* bias;   // [N]
  int N;
  // BIAS / RELU / RESID
  float* out;          // [n, ldo]
  int ldo;
  const float* resid;  // RESID: [n, ldo]
  const float* cconst; // RESID: optional [S, ld_cc], row picked by idx[row]
  int ld_cc;
  const int* idx;      // row -> sentence (RESID with cconst, QKV)
  // QKV: columns [0,256) -> q[row], [256,512) -> kbase[idx[row]*kv_stride + col-256], [512,768) -> vbase
  float* q;
  float* kbase;
  float* vbase;
  long kv_stride;
  // ARGMAX: partial[row*gridDim.y + blockIdx.y] = (best, bits(best index), second best, 0)
  float4* partial;
  // LSE / DS (in-batch InfoNCE, st_simloss.hip): s = acc * scale; target column of row r is row_offset + r.
  //  LSE: partial[row*gridDim.y + blockIdx.y] = (running max, sum exp(s - max), s[target] or 0, 0)
  //  DS : out[row*ldo + col] = g[row] * (exp(s - lse[row]) - [col == target]) * scale
  float scale;
  int row_offset;
  const float* lse;
  const float* g;
};

struct Lse {
  float m, l;
};
__device__ __forceinline__ Lse lse_merge(Lse a, Lse b) {
  if (b.m == -INFINITY) return a;
  if (a.m == -INFINITY) return b;
  const float m = fmaxf(a.m, b.m);
  Lse r;
  r.m = m;
  r.l = a.l * expf(a.m - m) + b.l * expf(b.m - m);
  return r;
}

struct Top2 {
  float m1;
  int i1;
  float m2;
};
__device__ __forceinline__ Top2 top2_merge(Top2 a, Top2 b) {
  const bool bwins = (b.m1 > a.m1) || (b.m1 == a.m1 && b.i1 < a.i1);
  Top2 r;
  r.m1 = bwins ? b.m1 : a.m1;
  r.i1 = bwins ? b.i1 : a.i1;
  const float loser = bwins ? a.m1 : b.m1;
  r.m2 = fmaxf(fmaxf(a.m2, b.m2), loser);
  return r;
}

// model.py:97-108: a * (x - mean) / (std_unbiased + 1e-6) + b over 256 features; one wave per row,
// lane holds 4 features.
__device__ __forceinline__ f32x4 layernorm256(f32x4 v, f32x4 ga, f32x4 gb) {
  float s = wave_sum(v.x + v.y + v.z + v.w);
  const float mean = s * (1.0f / 256.0f);
  f32x4 d = v - mean;
  float ss = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
  const float den = sqrtf(ss * (1.0f / 255.0f)) + 1e-6f;
  return (ga * d) / den + gb;
}

template <int EPI>
__global__ __launch_bounds__(512) void panel_gemm_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                 // [64][ST_LDA]
  float* Ws = smem + 64 * ST_LDA;   // [2][256][ST_LDW]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;

  const int ntiles = (a.N + 255) >> 8;
  const int per = (ntiles + gridDim.y - 1) / gridDim.y;
  const int t_begin = blockIdx.y * per;
  const int t_end = min(ntiles, t_begin + per);

  // ---- X panel -> LDS (64 rows x 256 floats = 4096 float4, 8 per thread) ----
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = tid + i * 512;
    const int r = f >> 6, c4 = f & 63;
    const int gr = min(m0 + r, n - 1);
    *(f32x4*)(As + r * ST_LDA + c4 * 4) = *(const f32x4*)(a.X + (size_t)gr * 256 + c4 * 4);
  }
  __syncthreads();
  if (a.ln_a) {
    const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
    const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float* p = As + (wave * 8 + rr) * ST_LDA + lane * 4;
      *(f32x4*)p = layernorm256(*(f32x4*)p, ga, gb);
    }
    __syncthreads();
  }

  Top2 best[2][16];
  if (EPI == EPI_ARGMAX) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        best[mi][r].m1 = -INFINITY;
        best[mi][r].i1 = 0x7fffffff;
        best[mi][r].m2 = -INFINITY;
      }
  }

  Lse run[2][16];
  float tgt[2][16];
  if (EPI == EPI_LSE) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        run[mi][r].m = -INFINITY;
        run[mi][r].l = 0.0f;
        tgt[mi][r] = 0.0f;
      }
  }

  const int total = (t_end - t_begin) * 8;
  f32x4 wreg[4];
  // W tile stage: 256 rows x 32 floats = 2048 float4, 4 per thread
  auto gload = [&](int it) {
    const int n0 = (t_begin + (it >> 3)) << 8;
    const int k0 = (it & 7) << 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 512;
      const int r = f >> 3, c4 = f & 7;
      const int gr = min(n0 + r, a.N - 1);
      wreg[i] = *(const f32x4*)(a.W + (size_t)gr * 256 + k0 + c4 * 4);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 512;
      const int r = f >> 3, c4 = f & 7;
      *(f32x4*)(Ws + buf * 256 * ST_LDW + r * ST_LDW + c4 * 4) = wreg[i];
    }
  };
  if (total > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();

  f32x16 acc0 = {0}, acc1 = {0};
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    const int ks = it & 7;
    if (it + 1 < total) gload(it + 1);
    const float* Ab = As + l31 * ST_LDA + ks * 32 + 4 * half;
    const float* Wb = Ws + buf * 256 * ST_LDW + (wave * 32 + l31) * ST_LDW + 4 * half;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const f32x4 a0 = *(const f32x4*)(Ab + kg * 8);
      const f32x4 a1 = *(const f32x4*)(Ab + 32 * ST_LDA + kg * 8);
      const f32x4 b = *(const f32x4*)(Wb + kg * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = mfma32(a0[s], b[s], acc0);
        acc1 = mfma32(a1[s], b[s], acc1);
      }
    }
    if (it + 1 < total) sstore(buf ^ 1);
    __syncthreads();
    if (ks == 7) {
      // ---- epilogue for N tile t ----
      const int col = ((t_begin + (it >> 3)) << 8) + wave * 32 + l31;
      const bool colok = col < a.N;
      const float bv = (colok && a.bias) ? a.bias[col] : 0.0f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x16 acc = mi ? acc1 : acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + mi * 32 + mfma32_row(r, lane);
          float v = acc[r] + bv;
          if (EPI == EPI_ARGMAX) {
            v = colok ? v : -INFINITY;
            Top2& t = best[mi][r];
            const bool gt = v > t.m1;
            t.m2 = gt ? t.m1 : fmaxf(t.m2, v);
            t.i1 = gt ? col : t.i1;
            t.m1 = gt ? v : t.m1;
          } else if (EPI == EPI_LSE) {
            if (colok) {
              const float sv = v * a.scale;
              Lse& u = run[mi][r];
              const float mn = fmaxf(u.m, sv);
              u.l = u.l * expf(u.m - mn) + expf(sv - mn);
              u.m = mn;
              if (col == a.row_offset + row) tgt[mi][r] = sv;
            }
          } else if (row < n && colok) {
            if (EPI == EPI_BIAS) {
              a.out[(size_t)row * a.ldo + col] = v;
            } else if (EPI == EPI_RELU) {
              a.out[(size_t)row * a.ldo + col] = fmaxf(v, 0.0f);
            } else if (EPI == EPI_RESID) {
              v += a.resid[(size_t)row * a.ldo + col];
              if (a.cconst) v += a.cconst[(size_t)a.idx[row] * a.ld_cc + col];
              a.out[(size_t)row * a.ldo + col] = v;
            } else if (EPI == EPI_DS) {
              const float sv = v * a.scale;
              const float pr = expf(sv - a.lse[row]) - (col == a.row_offset + row ? 1.0f : 0.0f);
              a.out[(size_t)row * a.ldo + col] = a.g[row] * pr * a.scale;
            } else if (EPI == EPI_QKV) {
              if (col < 256) {
                a.q[(size_t)row * 256 + col] = v;
              } else {
                const long s = a.idx[row];
                float* base = (col < 512) ? a.kbase : a.vbase;
                base[s * a.kv_stride + ((col - 256) & 255)] = v;
              }
            }
          }
        }
      }
      acc0 = (f32x16){0};
      acc1 = (f32x16){0};
    }
  }

  if (EPI == EPI_ARGMAX) {
    // reduce each (mi, reg) row over the 32 lanes of its half, then over the 8 waves through LDS
    float* red = Ws;  // [8 waves][64 rows][4]  (all waves are past the last barrier of the k-loop)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Top2 t = best[mi][r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          Top2 u;
          u.m1 = __shfl_xor(t.m1, o, 64);
          u.i1 = __shfl_xor(t.i1, o, 64);
          u.m2 = __shfl_xor(t.m2, o, 64);
          t = top2_merge(t, u);
        }
        if (l31 == 0) {
          float* p = red + (wave * 64 + mi * 32 + mfma32_row(r, lane)) * 4;
          p[0] = t.m1;
          p[1] = __int_as_float(t.i1);
          p[2] = t.m2;
        }
      }
    __syncthreads();
    if (tid < 64 && m0 + tid < n) {
      Top2 t;
      t.m1 = red[tid * 4 + 0];
      t.i1 = __float_as_int(red[tid * 4 + 1]);
      t.m2 = red[tid * 4 + 2];
      for (int w = 1; w < 8; ++w) {
        Top2 u;
        u.m1 = red[(w * 64 + tid) * 4 + 0];
        u.i1 = __float_as_int(red[(w * 64 + tid) * 4 + 1]);
        u.m2 = red[(w * 64 + tid) * 4 + 2];
        t = top2_merge(t, u);
      }
      a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m1, __int_as_float(t.i1), t.m2, 0.0f);
    }
  }

  if (EPI == EPI_LSE) {
    float* red = Ws;  // [8 waves][64 rows][4]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Lse t = run[mi][r];
        float tg = tgt[mi][r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          Lse u;
          u.m = __shfl_xor(t.m, o, 64);
          u.l = __shfl_xor(t.l, o, 64);
          t = lse_merge(t, u);
          tg += __shfl_xor(tg, o, 64);
        }
        if (l31 == 0) {
          float* p = red + (wave * 64 + mi * 32 + mfma32_row(r, lane)) * 4;
          p[0] = t.m;
          p[1] = t.l;
          p[2] = tg;
        }
      }
    __syncthreads();
    if (tid < 64 && m0 + tid < n) {
      Lse t = {red[tid * 4 + 0], red[tid * 4 + 1]};
      float tg = red[tid * 4 + 2];
      for (int w = 1; w < 8; ++w) {
        Lse u = {red[(w * 64 + tid) * 4 + 0], red[(w * 64 + tid) * 4 + 1]};
        t = lse_merge(t, u);
        tg += red[(w * 64 + tid) * 4 + 2];
      }
      a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m, t.l, tg, 0.0f);
    }
  }
}

#define ST_PANEL_LDS ((64 * ST_LDA + 2 * 256 * ST_LDW) * 4)

// ------------------------------------------------------------------------------------------------
// kloop_gemm: 128x128 tile, BK = 32, 256 threads (2x2 waves, each 64x64 = 2x2 MFMA tiles).
// Epilogue: BIAS / RELU / RESID (same meaning as above).
// ------------------------------------------------------------------------------------------------
// AT: X is given transposed, [K, n] row-major (ld = a.ldx);  WT: W is given transposed, [K, N] (ld = a.ldw)
template <int EPI, bool AT = false, bool WT = false>
__global__ __launch_bounds__(256) void kloop_gemm_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                        // [2][128][ST_LDW]
  float* Ws = smem + 2 * 128 * ST_LDW;     // [2][128][ST_LDW]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 128;
  if (m0 >= n) return;
  const int n0 = blockIdx.y * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wr = wave >> 1, wc = wave & 1;
  const int K = a.K;
  const int total = K >> 5;

  f32x4 areg[4], wreg[4];
  auto gload = [&](int it) {
    const int k0 = it << 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 3, c4 = f & 7;
      if (!AT) {
        areg[i] = *(const f32x4*)(a.X + (size_t)min(m0 + r, n - 1) * K + k0 + c4 * 4);
      } else {  // 32 k-rows x 32 float4 along m; caller guarantees n % 4 == 0 and ldx % 4 == 0
        const int kk = f >> 5, m4 = f & 31;
        areg[i] = *(const f32x4*)(a.X + (size_t)(k0 + kk) * a.ldx + min(m0 + m4 * 4, n - 4));
      }
      if (!WT) {
        wreg[i] = *(const f32x4*)(a.W + (size_t)min(n0 + r, a.N - 1) * K + k0 + c4 * 4);
      } else {
        const int kk = f >> 5, m4 = f & 31;
        wreg[i] = *(const f32x4*)(a.W + (size_t)(k0 + kk) * a.ldw + min(n0 + m4 * 4, a.N - 4));
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 3, c4 = f & 7;
      const int kk = f >> 5, m4 = f & 31;
      if (!AT) {
        *(f32x4*)(As + buf * 128 * ST_LDW + r * ST_LDW + c4 * 4) = areg[i];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) As[buf * 128 * ST_LDW + (m4 * 4 + j) * ST_LDW + kk] = areg[i][j];
      }
      if (!WT) {
        *(f32x4*)(Ws + buf * 128 * ST_LDW + r * ST_LDW + c4 * 4) = wreg[i];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) Ws[buf * 128 * ST_LDW + (m4 * 4 + j) * ST_LDW + kk] = wreg[i][j];
      }
    }
  };
  gload(0);
  sstore(0);
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    if (it + 1 < total) gload(it + 1);
    const float* Ab = As + buf * 128 * ST_LDW + (wr * 64 + l31) * ST_LDW + 4 * half;
    const float* Wb = Ws + buf * 128 * ST_LDW + (wc * 64 + l31) * ST_LDW + 4 * half;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const f32x4 a0 = *(const f32x4*)(Ab + kg * 8);
      const f32x4 a1 = *(const f32x4*)(Ab + 32 * ST_LDW + kg * 8);
      const f32x4 b0 = *(const f32x4*)(Wb + kg * 8);
      const f32x4 b1 = *(const f32x4*)(Wb + 32 * ST_LDW + kg * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0][0] = mfma32(a0[s], b0[s], acc[0][0]);
        acc[0][1] = mfma32(a0[s], b1[s], acc[0][1]);
        acc[1][0] = mfma32(a1[s], b0[s], acc[1][0]);
        acc[1][1] = mfma32(a1[s], b1[s], acc[1][1]);
      }
    }
    if (it + 1 < total) sstore(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wc * 64 + ni * 32 + l31;
      if (col >= a.N) continue;
      const float bv = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + mi * 32 + mfma32_row(r, lane);
        if (row >= n) continue;
        float v = acc[mi][ni][r] + bv;
        if (EPI == EPI_RELU) v = fmaxf(v, 0.0f);
        if (EPI == EPI_RESID) {
          v += a.resid[(size_t)row * a.ldo + col];
          if (a.cconst) v += a.cconst[(size_t)a.idx[row] * a.ld_cc + col];
        }
        a.out[(size_t)row * a.ldo + col] = v;
      }
    }
}
#define ST_KLOOP_LDS (4 * 128 * ST_LDW * 4)

// ---- host-side launch helpers ----
template <int EPI>
static int launch_panel(const GemmArgs& a, int n_bound, int splits, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    ST_HIP(hipFuncSetAttribute((const void*)panel_gemm_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               ST_PANEL_LDS));
    attr_done = true;
  }
  if (n_bound <= 0) return ST_OK;
  dim3 grid((n_bound + 63) / 64, splits);
  hipLaunchKernelGGL(panel_gemm_kernel<EPI>, grid, dim3(512), ST_PANEL_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
template <int EPI, bool AT = false, bool WT = false>
static int launch_kloop(const GemmArgs& a, int n_bound, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    ST_HIP(hipFuncSetAttribute((const void*)kloop_gemm_kernel<EPI, AT, WT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               ST_KLOOP_LDS));
    attr_done = true;
  }
  if (n_bound <= 0) return ST_OK;
  dim3 grid((n_bound + 127) / 128, (a.N + 127) / 128);
  hipLaunchKernelGGL((kloop_gemm_kernel<EPI, AT, WT>), grid, dim3(256), ST_KLOOP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

#define ST_TRY(x)            \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != ST_OK) return rc_; \
  } while (0)

