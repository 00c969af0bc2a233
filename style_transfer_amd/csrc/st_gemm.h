// f32-MFMA GEMM building blocks:  C[n,N] = op(X[n,K]) . W[N,K]^T (+ epilogue)
//
//  * shared argument block, epilogue kinds, top-2 / log-sum-exp merges, the reference's LayerNorm;
//  * kloop_gemm (any K % 32 == 0): classic 128x128 tile, both operands streamed through LDS, 256 threads;
//    optional transposed operands.  Used where BOTH operands are activations (the two dS products of
//    st_simloss_bwd); every weight GEMM uses the row-panel kernels of st_rowpanel.h instead.
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

enum { EPI_BIAS = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_QKV = 3, EPI_ARGMAX = 4, EPI_LSE = 5, EPI_DS = 6, EPI_DST = 7 };

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
  //  DS : out[row*ldo + col] = g[row] * (exp(s - lse[row]) - [col == target]) * scale;  g == null: the bare exp(..) - [..]
  //  DST: the transposed block (rows = keys, columns = queries): out[row*ldo + col] = exp(s - lse[col]) - [row == row_offset + col]
  float scale;
  int pos;             // dec_qkv_attn16_kernel: the decode step (keys 0 .. pos are attended; kbase / vbase are the layer's bases)
  int row_offset;
  const float* lse;
  const float* g;
  int fmt;             // row-panel kernels: operand format of the packed W (ST_GEMM_F32 / ST_GEMM_F16X3)
  const int* xrows;    // row-panel kernels: optional gather, panel row r reads X[xrows[r]] (null = X[r])
  int* range_flag;     // f16x3: set to 1 when an activation leaves the f16 range (null = unchecked)
  const float* colpad; // rp_screen16p_kernel: [2][32 ceil(N / 32)] bias (padding: RS_PAD_LOGIT) and column norms (padding: 0), see rp_screen_colpad_kernel
  const float* wnorm;  // rp_screen16_kernel: optional [N] 2-norms of the rows of W (rounded up): keys become UPPER BOUNDS
                       // s + scale * ||x|| * ||W_v|| of the exact logits (scale = slack * (2^-10 + 2^-16)), see finish_kernel
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
// The same with one division per row instead of four (f16x3 kernels: the result is split into f16 hi + lo right
// away, so the <= 1 ulp difference to the form above is below the format's own 2^-22)
__device__ __forceinline__ f32x4 layernorm256r(f32x4 v, f32x4 ga, f32x4 gb) {
  float s = wave_sum(v.x + v.y + v.z + v.w);
  const float mean = s * (1.0f / 256.0f);
  f32x4 d = v - mean;
  float ss = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
  const float inv = 1.0f / (sqrtf(ss * (1.0f / 255.0f)) + 1e-6f);
  return (ga * d) * inv + gb;
}

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
template <int EPI, bool AT = false, bool WT = false>
static int launch_kloop(const GemmArgs& a, int n_bound, hipStream_t st) {
  ST_ENSURE_LDS((kloop_gemm_kernel<EPI, AT, WT>), ST_KLOOP_LDS);
  if (n_bound <= 0) return ST_OK;
  dim3 grid((n_bound + 127) / 128, (a.N + 127) / 128);
  hipLaunchKernelGGL((kloop_gemm_kernel<EPI, AT, WT>), grid, dim3(256), ST_KLOOP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

// in-library kernel timing (st_profile_enable / st_profile_read, defined in st_decode.hip)
struct ProfPair {
  hipEvent_t a, b;
  double flops;
};
extern int g_prof_on;
// flops = per_row * (the live row count read ON THE DEVICE when the launch runs: *n_dev, or n_host when n_dev is null) —
// the host's row bound may be stale (st_decode_greedy reads the count only every few steps)
int prof_begin(ProfPair* p, double per_row, const int* n_dev, int n_host, hipStream_t st);
int prof_end(ProfPair* p, hipStream_t st);

#define ST_TRY(x)            \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != ST_OK) return rc_; \
  } while (0)

