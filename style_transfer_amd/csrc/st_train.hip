// One optimisation step of the style scorer (first half): the body of the reference's
// train_contrastive loop (siamese_contrastive.py:918-931) for SiameseNet (:29-62) + Predictor (:66-86) in TRAIN mode.
//
//   forward   G = 1 + Kp + Kn SiameseNet calls of B rows each (ori, like_0.., unlike_0..: :380-382), every call with
//             its OWN BatchNorm batch statistics and Dropout(0.2) masks; Predictor on the ori call (:405);
//             cosine scores / tau (:409-417); NLL of log_softmax([pos_i, neg_0..]) per positive (:438-453)
//   backward  to every parameter.  like / unlike outputs are detached in Predictor mode (:406-407), so only the B rows
//             of the ori call carry gradient; with --simplest_siamese (:384-400) nothing is detached and all G calls do
//   running statistics of the three BatchNorm layers: G (resp. 1) momentum-0.1 updates in call order
//   st_adam_step: torch.optim.Adam (defaults) over the flat parameter vector
//
// Layout: all trainable parameters live in ONE flat f32 vector in the order of
// list(siamese_net.parameters()) + list(predictor_net.parameters()) (:881-882; offsets ST_TP_* in st_kernels.h), so the
// optimizer is one elementwise kernel and a data-parallel job all-reduces the gradient as one RCCL bucket.
// The matrix products are f32 MFMA (v_mfma_f32_32x32x2_f32) on LDS tiles with arbitrary operand strides (one
// kernel serves Y = X.W^T, dX = dY.W and dW = dY^T.X); dW contracts over rows in slabs that a second kernel sums in a
// fixed order, so a step is bit-reproducible.  At the reference's batch (128) the step is 24 launches, 0.19 ms
// (DESIGN.md section 7b has the measured steps from round 2's 47 launches and 0.41 ms).
#include <math.h>

#include "st_common.h"

// ST_TRAIN_SCALAR_GEMM=1: one element per load in the small products (the form that serves misaligned operands; the
// parity tests run it once so that it stays pinned)
int g_tr_scalar_gemm = 0;
static void tr_env() {
  static const bool once = [] {
    const char* v = getenv("ST_TRAIN_SCALAR_GEMM");
    if (v && v[0] == '1') g_tr_scalar_gemm = 1;
    return true;
  }();
  (void)once;
}

#define TR_EPS 1e-5f
#define TR_MOM 0.1f

// ------------------------------------------------------------------------------------------------
// C[i,j] (+bias[j]) = sum_r A(i,r) * B(r,j),   A(i,r) = A[i*sai + r*sar],  B(r,j) = B[r*sbr + j*sbj]
// blockIdx.z = slab of the contraction range [z*rper, (z+1)*rper) -> C + z*slab
// ------------------------------------------------------------------------------------------------
struct TgArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, R;
  long sai, sar, sbr, sbj;
  int ldc, rper;
  long slab;
  float* colsum;  // tg_gemm_kernel only
};
#define TG_KS 64  /* contraction rows per LDS stage */
// TS x TS tile per workgroup of four waves; one LDS stage holds 64 contraction rows.  <64, 1>: a wave per 32 x 32 sub-tile,
// contracting the whole stage.  <32, 4>: one sub-tile, a wave per quarter of the stage.  Round 2 had four waves on a 64 x 64 tile and
// 16-row stages: at the reference's batch a backward product has 4 to 64 such tiles, so a handful of CUs walked the whole
// contraction (7.8 us of f32 MFMA time per 64 x 64 x 512 tile on ONE CU, an LDS round trip in front of every MFMA on top:
// 12 us at 128 contraction rows, 29 us at 512) while the rest of the chip idled.  The 32 x 32 form spreads the same work
// over four times as many CUs and is taken while the 64 x 64 form would leave CUs empty; the large forward products keep
// 64 x 64 tiles (twice the flops per LDS byte), two workgroups a CU.  The four partial tiles are added through LDS as
// (k0 + k2) + (k1 + k3): a fixed order, the step stays bit-reproducible.
// colsum (optional; dW = dY^T X products, where A(i, r) = dY[r][i]): the blocks of tile column 0 also sum their A tile
// over r, colsum[z][i] = sum_r A(i, r) = this slab's part of the bias gradient -- the rows are in LDS anyway.
// VEC: both operands are read four elements at a time along their unit-stride axis (the host checks alignment and
// divisibility by 4); the scalar form serves everything else.
template <int TS, int KQ, int RING, bool VEC>
__device__ __forceinline__ void tg_gemm_body(const TgArgs& a, const int bx, const int by, const int bz) {
  constexpr int NSUB = (TS / 32) * (TS / 32), NW = KQ * NSUB, NT = 64 * NW, NQ = TG_KS * TS / NT, IB = TS == 64 ? 6 : 5;
  constexpr int W = VEC ? 4 : 1, NG = NQ / W;  // elements per load, loads per operand and stage
  static_assert(KQ == 1 || KQ == 4, "a wave contracts a whole stage or a quarter of it");
  static_assert(NT == 256, "four waves");
  __shared__ __align__(16) float smem[2 * TG_KS * (TS + 4)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = wave & (NSUB - 1), kq = wave / NSUB;
  const int i0 = by * TS, j0 = bx * TS;
  const int r0 = bz * a.rper, r1 = min(a.R, r0 + a.rper);
  const bool a_r_fast = a.sar == 1, b_r_fast = a.sbr == 1;
  // LDS row stride: TS + 4 keeps 16-byte rows for the vector writes of an operand whose unit stride runs along the tile;
  // TS + 1 lets the 16 x 4 (r, i) lanes of an r-fast operand's four-row writes (VEC) fall on 64 different banks
  const int lda = a_r_fast ? TS + 1 : TS + 4, ldb = b_r_fast ? TS + 1 : TS + 4;
  float* const As = smem;
  float* const Bs = smem + TG_KS * (TS + 4);
  // load g of a stage covers elements e = W (tid + 256 g) ...: r-fast operands put 16 lanes along r (64-byte or 256-byte
  // runs), the others put the lanes along the tile row
  auto r_of = [&](int g, bool fast) {
    const int e = tid + NT * g;
    if (VEC) return fast ? 4 * (e & 15) : (e >> (IB - 2));
    return fast ? (e & 15) + 16 * (e >> (4 + IB)) : (e >> IB);
  };
  auto i_of = [&](int g, bool fast) {
    const int e = tid + NT * g;
    if (VEC) return fast ? (e >> 4) : 4 * (e & (TS / 4 - 1));
    return fast ? (e >> 4) & (TS - 1) : (e & (TS - 1));
  };
  // the stages are requested RING ahead
  const float* const Ab = a.A + (long)i0 * a.sai + (long)r0 * a.sar;  // 32-bit element offsets from here (a tile spans < 2^31)
  const float* const Bb = a.B + (long)r0 * a.sbr + (long)j0 * a.sbj;
  int pa[NG], pb[NG];  // (-1: the element's tile row / column is out of range)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int ar = r_of(g, a_r_fast), ai = i_of(g, a_r_fast);
    const int br = r_of(g, b_r_fast), bj = i_of(g, b_r_fast);
    pa[g] = i0 + ai < a.M ? ai * (int)a.sai + ar * (int)a.sar : -1;
    pb[g] = j0 + bj < a.N ? br * (int)a.sbr + bj * (int)a.sbj : -1;
  }
  typedef float vecw __attribute__((ext_vector_type(W)));
  vecw ra[RING][NG], rb[RING][NG];
  const int nrow = r1 - r0;  // (<= 0: an empty slab)
  auto fetch = [&](vecw (&va)[NG], vecw (&vb)[NG], int st) {
    const int left = nrow - st * TG_KS;
    const float* As_ = Ab + (long)st * TG_KS * a.sar;  // wave-uniform
    const float* Bs_ = Bb + (long)st * TG_KS * a.sbr;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      va[g] = (pa[g] >= 0 && r_of(g, a_r_fast) < left) ? *(const vecw*)(As_ + pa[g]) : vecw(0.0f);
      vb[g] = (pb[g] >= 0 && r_of(g, b_r_fast) < left) ? *(const vecw*)(Bs_ + pb[g]) : vecw(0.0f);
    }
  };
  auto stage_store = [&](float* T, int ld, bool fast, int g, vecw v) {
    float* d = T + r_of(g, fast) * ld + i_of(g, fast);
    if (!VEC) {
      d[0] = v[0];
    } else if (fast) {  // four contraction rows of one tile row
#pragma unroll
      for (int k = 0; k < W; ++k) d[k * ld] = v[k];
    } else {
      *(vecw*)d = v;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int wi = (sub & 1) * 32, wj = (sub >> 1) * 32, l31 = lane & 31, half = lane >> 5;
  const bool do_colsum = a.colsum && bx == 0;  // block-uniform
  constexpr int RPW = TG_KS / NW;                       // rows of a stage a wave adds up for the column sums
  float csum = 0.0f;
  const int nst = nrow > 0 ? (nrow + TG_KS - 1) / TG_KS : 0;
#pragma unroll
  for (int u = 0; u < RING; ++u) fetch(ra[u], rb[u], u);
  for (int sb = 0; sb < nst; sb += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const int st = sb + u;
      if (st >= nst) break;
      __syncthreads();
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        stage_store(As, lda, a_r_fast, g, ra[u][g]);
        stage_store(Bs, ldb, b_r_fast, g, rb[u][g]);
      }
      __syncthreads();
      fetch(ra[u], rb[u], st + RING);
      constexpr int RW = TG_KS / KQ;          // contraction rows of a stage per wave
      if (st * TG_KS + RW * kq < nrow) {  // (rows past r1 are zero; a quarter wholly past it is skipped)
        const float* Ar = As + (RW * kq + half) * lda + wi + l31;
        const float* Br = Bs + (RW * kq + half) * ldb + wj + l31;
#pragma unroll
        for (int kk = 0; kk < RW / 2; ++kk) acc = mfma32(Ar[2 * kk * lda], Br[2 * kk * ldb], acc);
      }
      if (do_colsum) {
        const float* Ac = As + RPW * wave * lda + (lane & (TS - 1));
#pragma unroll
        for (int r = 0; r < RPW; r += 4) csum += (Ac[r * lda] + Ac[(r + 1) * lda]) + (Ac[(r + 2) * lda] + Ac[(r + 3) * lda]);
      }
    }
  }
  // (k0 + k2) + (k1 + k3) through LDS; a wave's tile sits as [register][lane] at sub * 1024 of its region
  __syncthreads();
  float* reg0 = smem;
  float* reg1 = smem + NSUB * 1024;
  if (KQ == 4) {
    if (kq >= 2) {
      float* dst = (kq == 2 ? reg0 : reg1) + sub * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc[r];
    }
    __syncthreads();
    if (kq < 2) {
      const float* src = (kq == 0 ? reg0 : reg1) + sub * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += src[r * 64];
    }
    __syncthreads();
    if (kq == 1) {
      float* dst = reg0 + sub * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc[r];
    }
  }
  if (do_colsum) reg1[wave * 64 + lane] = csum;
  __syncthreads();
  if (kq == 0) {
    const float* src = reg0 + sub * 1024 + lane;
    float* C = a.C + (long)bz * a.slab;
    const int j = j0 + wj + l31;
    const float bj = (a.bias && j < a.N) ? a.bias[j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wi + mfma32_row(r, lane);
      if (i < a.M && j < a.N) C[(long)i * a.ldc + j] = (KQ == 4 ? acc[r] + src[r * 64] : acc[r]) + bj;
    }
  }
  if (do_colsum && tid < TS && i0 + tid < a.M) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += reg1[w * 64 + tid];
    a.colsum[(long)bz * a.M + i0 + tid] = s;
  }
}
template <int TS, int KQ, int RING, bool VEC>
__global__ __launch_bounds__(256, 2) void tg_gemm_kernel(TgArgs a) {
  tg_gemm_body<TS, KQ, RING, VEC>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}
// two independent small products in one launch (the weight gradient dY^T X and the input gradient dY W of one layer: at
// the reference's batch each is a 6 us launch of a few tiles): blocks [0, na) work on a, the rest on b
template <bool VEC>
__global__ __launch_bounds__(256, 2) void tg_gemm_pair_kernel(TgArgs a, TgArgs b, int na, int gxa, int gxb) {
  const int id = blockIdx.x;
  if (id < na)
    tg_gemm_body<32, 4, 3, VEC>(a, id % gxa, id / gxa, 0);
  else
    tg_gemm_body<32, 4, 3, VEC>(b, (id - na) % gxb, (id - na) / gxb, 0);
}

// The same product on a 128 x 128 tile per workgroup (64 x 64 per wave, contraction steps of 32): 4x the MFMA work per
// LDS byte and per barrier.  Used when it still fills the chip (large batches); the 64 x 64 form above keeps the small ones.
#define TG_LDW 36
// VEC: four elements a load along each operand's unit-stride axis, as in tg_gemm_kernel (round 3: the sixteen one-element
// loads per operand and stage, each with its 64-bit address and range test, cost as much as the stage's MFMAs)
template <bool VEC>
__global__ __launch_bounds__(256) void tg_gemm128_kernel(TgArgs a) {
  __shared__ __align__(16) float As[128 * TG_LDW], Bs[128 * TG_LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
  const int r0 = blockIdx.z * a.rper, r1 = min(a.R, r0 + a.rper);
  const bool a_r_fast = a.sar == 1, b_r_fast = a.sbr == 1;
  constexpr int W = VEC ? 4 : 1, NG = 16 / W;
  typedef float vecw __attribute__((ext_vector_type(W)));
  vecw ra[NG], rb[NG];
  // load q covers W elements from (r, i): r-fast operands run along r, the others along the tile
  auto r_of = [&](int q, bool fast) {
    const int e = tid + 256 * q;
    if (VEC) return fast ? 4 * (e & 7) : (e >> 5);
    return fast ? (e & 31) : (e >> 7);
  };
  auto i_of = [&](int q, bool fast) {
    const int e = tid + 256 * q;
    if (VEC) return fast ? (e >> 3) : 4 * (e & 31);
    return fast ? (e >> 5) : (e & 127);
  };
  auto fetch = [&](int rc) {
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int ar = r_of(q, a_r_fast), ai = i_of(q, a_r_fast);
      const int br = r_of(q, b_r_fast), bj = i_of(q, b_r_fast);
      ra[q] = (i0 + ai < a.M && rc + ar < r1) ? *(const vecw*)(a.A + (long)(i0 + ai) * a.sai + (long)(rc + ar) * a.sar) : vecw(0.0f);
      rb[q] = (j0 + bj < a.N && rc + br < r1) ? *(const vecw*)(a.B + (long)(rc + br) * a.sbr + (long)(j0 + bj) * a.sbj) : vecw(0.0f);
    }
  };
  auto stage_store = [&](float* T, bool fast, int q, vecw v) {  // LDS rows: tile index, TG_LDW apart; r runs along a row
    float* d = T + i_of(q, fast) * TG_LDW + r_of(q, fast);
    if (!VEC) {
      d[0] = v[0];
    } else if (fast) {
      *(vecw*)d = v;
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) d[k * TG_LDW] = v[k];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, half = lane >> 5;
  if (r0 < r1) fetch(r0);
  for (int rc = r0; rc < r1; rc += 32) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      stage_store(As, a_r_fast, q, ra[q]);
      stage_store(Bs, b_r_fast, q, rb[q]);
    }
    __syncthreads();
    if (rc + 32 < r1) fetch(rc + 32);
    const float* Ab = As + (wr * 64 + l31) * TG_LDW + 4 * half;
    const float* Bb = Bs + (wc * 64 + l31) * TG_LDW + 4 * half;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const f32x4 a0 = *(const f32x4*)(Ab + kg * 8), a1 = *(const f32x4*)(Ab + 32 * TG_LDW + kg * 8);
      const f32x4 b0 = *(const f32x4*)(Bb + kg * 8), b1 = *(const f32x4*)(Bb + 32 * TG_LDW + kg * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0][0] = mfma32(a0[q], b0[q], acc[0][0]);
        acc[0][1] = mfma32(a0[q], b1[q], acc[0][1]);
        acc[1][0] = mfma32(a1[q], b0[q], acc[1][0]);
        acc[1][1] = mfma32(a1[q], b1[q], acc[1][1]);
      }
    }
  }
  float* C = a.C + (long)blockIdx.z * a.slab;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int j = j0 + wc * 64 + ni * 32 + l31;
      if (j >= a.N) continue;
      const float bj = a.bias ? a.bias[j] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wr * 64 + mi * 32 + mfma32_row(r, lane);
        if (i < a.M) C[(long)i * a.ldc + j] = acc[mi][ni][r] + bj;
      }
    }
}

// out[e] = sum_s in[s*stride + e]   (fixed order: reproducible)
__global__ __launch_bounds__(256) void tr_slab_sum_kernel(const float* __restrict__ in, int nslab, long stride, long n,
                                                          float* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.0f;
  for (int k = 0; k < nslab; ++k) s += in[k * stride + e];
  out[e] = s;
}

// column sums of X[rows, N] over row slabs: part[z][col] = sum_{row in slab z} X[row][col]
__global__ __launch_bounds__(256) void tr_colsum_kernel(const float* __restrict__ X, int rows, int N, int rper,
                                                        float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rper, r1 = min(rows, r0 + rper);
  float s = 0.0f;
  if (c < N)
    for (int r = r0 + rg; r < r1; r += 4) s += X[(long)r * N + c];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < N) part[(long)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- BatchNorm (train mode), one statistic set per (call g, feature) ----
// grid (ceil(N/64), G): mean and BIASED variance of H[g*B .. g*B+B, col], two passes
// 1024 threads = 64 columns x 16 row groups, two rows in flight per thread: with 4 row groups the kernel walked 8192 rows
// twice at 0.56 TB/s and was 37 % of the B = 8192 step.
#define TR_RG 16
__device__ __forceinline__ float tr_rg_sum(float (*red)[64], int cl) {  // fixed order over the 16 row groups
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < TR_RG; k += 4) s += (red[k][cl] + red[k + 1][cl]) + (red[k + 2][cl] + red[k + 3][cl]);
  return s;
}
__global__ __launch_bounds__(64 * TR_RG) void tr_bn_stats_kernel(const float* __restrict__ H, int B, int N, float* __restrict__ mean,
                                                                 float* __restrict__ var) {
  __shared__ float red[TR_RG][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rg = threadIdx.x >> 6, g = blockIdx.y;
  const float* Hg = H + (long)g * B * N;
  float s0 = 0.0f, s1 = 0.0f;
  if (c < N) {
    int r = rg;
    for (; r + TR_RG < B; r += 2 * TR_RG) {
      s0 += Hg[(long)r * N + c];
      s1 += Hg[(long)(r + TR_RG) * N + c];
    }
    if (r < B) s0 += Hg[(long)r * N + c];
  }
  red[rg][cl] = s0 + s1;
  __syncthreads();
  const float mu = tr_rg_sum(red, cl) / (float)B;
  __syncthreads();
  float q0 = 0.0f, q1 = 0.0f;
  if (c < N) {
    int r = rg;
    for (; r + TR_RG < B; r += 2 * TR_RG) {
      const float d0 = Hg[(long)r * N + c] - mu, d1 = Hg[(long)(r + TR_RG) * N + c] - mu;
      q0 = fmaf(d0, d0, q0);
      q1 = fmaf(d1, d1, q1);
    }
    if (r < B) {
      const float d0 = Hg[(long)r * N + c] - mu;
      q0 = fmaf(d0, d0, q0);
    }
  }
  red[rg][cl] = q0 + q1;
  __syncthreads();
  if (rg == 0 && c < N) {
    mean[(long)g * N + c] = mu;
    var[(long)g * N + c] = tr_rg_sum(red, cl) / (float)B;
  }
}
// running statistics: G sequential momentum updates in call order (unbiased variance, as nn.BatchNorm1d)
struct RunArgs {  // blockIdx.y = layer
  const float* mean[3];
  const float* var[3];
  float* rm[3];
  float* rv[3];
  int G[3], N[3];
  int B;
};
__device__ __forceinline__ void tr_bn_running(const RunArgs& a, int l, int c) {
  const int N = a.N[l], G = a.G[l], B = a.B;
  if (c >= N) return;
  const float* __restrict__ mean = a.mean[l];
  const float* __restrict__ var = a.var[l];
  float* __restrict__ rm = a.rm[l];
  float* __restrict__ rv = a.rv[l];
  float m = rm[c], v = rv[c];
  const float ub = (float)B / (float)(B - 1);
  for (int g = 0; g < G; ++g) {
    m = (1.0f - TR_MOM) * m + TR_MOM * mean[(long)g * N + c];
    v = (1.0f - TR_MOM) * v + TR_MOM * (var[(long)g * N + c] * ub);
  }
  rm[c] = m;
  rv[c] = v;
}

// Dropout keep decision: explicit mask (parity tests replay the reference's masks) or a counter hash of
// (seed, element index) — stateless, so the backward pass regenerates the forward's decision.
__device__ __forceinline__ bool tr_keep(const uint8_t* mask, unsigned long long seed, long idx, float drop_p) {
  if (mask) return mask[idx] != 0;
  unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(idx + 1);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (float)(x >> 40) * (1.0f / 16777216.0f) >= drop_p;
}

// A = dropout(act(gamma * (H - mean) / sqrt(var + eps) + beta)); slope 0 = ReLU, 0.2 = LeakyReLU; one thread per element
// keep-mask layout [G calls][2 layers][B][N]
struct BnActArgs {
  const float* H;
  const float* mean;
  const float* var;
  const float* gamma;
  const float* beta;
  float* A;
  const uint8_t* mask;
  unsigned long long seed;
  int B, N, layer, rows;
  float slope, drop_p;
  // backward only
  const float* dA;
  const float* sums;  // [G][2][N]: sum dy, sum dy*xhat
  float* dH;
  float *dbeta, *dgamma;  // one call carries gradient (Predictor mode): its two sums ARE the gradients, written here as well
};
__device__ __forceinline__ long tr_mask_index(int g, int layer, int b, int c, int B, int N) {
  return (((long)g * 2 + layer) * B + b) * N + c;
}
__global__ __launch_bounds__(256) void tr_bn_act_kernel(BnActArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)a.rows * a.N) return;
  const int row = (int)(e / a.N), c = (int)(e - (long)row * a.N), g = row / a.B, b = row - g * a.B;
  const float inv = 1.0f / sqrtf(a.var[(long)g * a.N + c] + TR_EPS);
  const float y = (a.H[e] - a.mean[(long)g * a.N + c]) * inv * a.gamma[c] + a.beta[c];
  float v = y > 0.0f ? y : a.slope * y;
  if (a.drop_p > 0.0f) v = tr_keep(a.mask, a.seed, tr_mask_index(g, a.layer, b, c, a.B, a.N), a.drop_p) ? v * (1.0f / (1.0f - a.drop_p)) : 0.0f;
  a.A[e] = v;
}
// statistics and activation of one (64 features, call g) slab in one launch: tr_bn_stats_kernel's two passes, then
// tr_bn_act_kernel's expression over the same rows (at the reference's batch each of the two was a 5 us launch of 0.1 us of work)
__global__ __launch_bounds__(64 * TR_RG) void tr_bn_fwd_kernel(BnActArgs a, float* __restrict__ mean, float* __restrict__ var) {
  __shared__ float red[TR_RG][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rg = threadIdx.x >> 6, g = blockIdx.y, B = a.B, N = a.N;
  const float* Hg = a.H + (long)g * B * N;
  float s0 = 0.0f, s1 = 0.0f;
  if (c < N) {
    int r = rg;
    for (; r + TR_RG < B; r += 2 * TR_RG) {
      s0 += Hg[(long)r * N + c];
      s1 += Hg[(long)(r + TR_RG) * N + c];
    }
    if (r < B) s0 += Hg[(long)r * N + c];
  }
  red[rg][cl] = s0 + s1;
  __syncthreads();
  const float mu = tr_rg_sum(red, cl) / (float)B;
  __syncthreads();
  float q0 = 0.0f, q1 = 0.0f;
  if (c < N) {
    int r = rg;
    for (; r + TR_RG < B; r += 2 * TR_RG) {
      const float d0 = Hg[(long)r * N + c] - mu, d1 = Hg[(long)(r + TR_RG) * N + c] - mu;
      q0 = fmaf(d0, d0, q0);
      q1 = fmaf(d1, d1, q1);
    }
    if (r < B) {
      const float d0 = Hg[(long)r * N + c] - mu;
      q0 = fmaf(d0, d0, q0);
    }
  }
  red[rg][cl] = q0 + q1;
  __syncthreads();
  if (c >= N) return;
  const float vr = tr_rg_sum(red, cl) / (float)B;
  if (rg == 0) {
    mean[(long)g * N + c] = mu;
    var[(long)g * N + c] = vr;
  }
  const float inv = 1.0f / sqrtf(vr + TR_EPS), gm = a.gamma[c], bt = a.beta[c];
  for (int b = rg; b < B; b += TR_RG) {
    const long e = ((long)g * B + b) * N + c;
    const float y = (a.H[e] - mu) * inv * gm + bt;
    float v = y > 0.0f ? y : a.slope * y;
    if (a.drop_p > 0.0f) v = tr_keep(a.mask, a.seed, tr_mask_index(g, a.layer, b, c, B, N), a.drop_p) ? v * (1.0f / (1.0f - a.drop_p)) : 0.0f;
    a.A[e] = v;
  }
}
// dy = dA * dropout' * act'(y);  the two column sums BatchNorm's backward needs: sums[g][0][c] = sum dy (= d beta),
// sums[g][1][c] = sum dy * xhat (= d gamma).  grid (ceil(N/64), G)
__device__ __forceinline__ float tr_dy(const BnActArgs& a, long e, int g, int b, int c, float& xhat) {
  const float inv = 1.0f / sqrtf(a.var[(long)g * a.N + c] + TR_EPS);
  xhat = (a.H[e] - a.mean[(long)g * a.N + c]) * inv;
  const float y = xhat * a.gamma[c] + a.beta[c];
  float d = a.dA[e];
  if (a.drop_p > 0.0f) d = tr_keep(a.mask, a.seed, tr_mask_index(g, a.layer, b, c, a.B, a.N), a.drop_p) ? d * (1.0f / (1.0f - a.drop_p)) : 0.0f;
  return y > 0.0f ? d : a.slope * d;
}
// gridDim.z > 1 (large batches with few calls carrying gradient: one call of 8192 rows was 8 workgroups walking 512 rows a
// thread, 244 us): block z sums rows [z rper, (z + 1) rper) into part[z][g][2][N] and tr_bn_bwd_sums_finish_kernel adds
// the parts in order.
__global__ __launch_bounds__(64 * TR_RG) void tr_bn_bwd_sums_kernel(BnActArgs a, int rper, float* __restrict__ part) {
  __shared__ float red[2][TR_RG][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rg = threadIdx.x >> 6, g = blockIdx.y;
  const int b_lo = blockIdx.z * rper, b_hi = min(a.B, b_lo + rper);
  float s1 = 0.0f, s2 = 0.0f;
  if (c < a.N)
    for (int b = b_lo + rg; b < b_hi; b += TR_RG) {
      float xh;
      const float dy = tr_dy(a, ((long)g * a.B + b) * a.N + c, g, b, c, xh);
      s1 += dy;
      s2 = fmaf(dy, xh, s2);
    }
  red[0][rg][cl] = s1;
  red[1][rg][cl] = s2;
  __syncthreads();
  if (rg == 0 && c < a.N) {
    const float sb = tr_rg_sum(red[0], cl), sg = tr_rg_sum(red[1], cl);
    if (gridDim.z > 1) {
      float* o = part + ((long)blockIdx.z * gridDim.y + g) * 2 * a.N;
      o[c] = sb;
      o[a.N + c] = sg;
      return;
    }
    float* o = (float*)a.sums + (long)g * 2 * a.N;
    o[c] = sb;
    o[a.N + c] = sg;
    if (a.dbeta) {
      a.dbeta[c] = sb;
      a.dgamma[c] = sg;
    }
  }
}
__global__ __launch_bounds__(256) void tr_bn_bwd_sums_finish_kernel(BnActArgs a, int G, int nz, const float* __restrict__ part) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // over [G][2][N]
  if (e >= G * 2 * a.N) return;
  float s = 0.0f;
  for (int z = 0; z < nz; ++z) s += part[(long)z * G * 2 * a.N + e];
  ((float*)a.sums)[e] = s;
  if (a.dbeta) {  // (one call: e = k N + c)
    if (e < a.N) a.dbeta[e] = s;
    else a.dgamma[e - a.N] = s;
  }
}
// dH = gamma * inv / B * (B*dy - sum dy - xhat * sum dy*xhat)
__global__ __launch_bounds__(256) void tr_bn_bwd_apply_kernel(BnActArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)a.rows * a.N) return;
  const int row = (int)(e / a.N), c = (int)(e - (long)row * a.N), g = row / a.B, b = row - g * a.B;
  float xh;
  const float dy = tr_dy(a, e, g, b, c, xh);
  const float inv = 1.0f / sqrtf(a.var[(long)g * a.N + c] + TR_EPS);
  const float* s = a.sums + (long)g * 2 * a.N;
  a.dH[e] = a.gamma[c] * inv / (float)a.B * ((float)a.B * dy - s[c] - xh * s[a.N + c]);
}

// ---- scores, loss, d(loss)/d(scores) ----
// one wave per batch row b: cosine of p[b] with the K = Kp+Kn reference outputs Z[1+k][b]; per-positive NLL term and
// the reference's `choice` (argmax of the log-softmax row, first maximum wins)
__global__ __launch_bounds__(256) void tr_score_kernel(const float* __restrict__ P, const float* __restrict__ Z, int B, int Kp, int Kn,
                                                       float tau, float* __restrict__ S, float* __restrict__ lossrow,
                                                       int32_t* __restrict__ choice) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int K = Kp + Kn;
  const f32x4 p = *(const f32x4*)(P + (long)b * 256 + lane * 4);
  const float pn = fmaxf(sqrtf(wave_sum(p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w)), 1e-8f);
  float mine = -INFINITY;
  for (int k0 = 0; k0 < K; k0 += 4) {  // four rows requested together: the loop is a chain of load latencies otherwise
    f32x4 zb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) zb[u] = *(const f32x4*)(Z + ((long)(1 + min(k0 + u, K - 1)) * B + b) * 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 z = zb[u];
      const float zn = fmaxf(sqrtf(wave_sum(z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w)), 1e-8f);
      const float s = wave_sum(p.x * z.x + p.y * z.y + p.z * z.z + p.w * z.w) / (pn * zn) / tau;
      if (lane == k0 + u) mine = s;  // (rows past K - 1 repeat row K - 1 and land on lanes that are not read)
    }
  }
  if (lane < K) S[(long)b * K + lane] = mine;
  // negatives: lanes Kp..K-1
  const bool isneg = lane >= Kp && lane < K;
  const float nmax = wave_max(isneg ? mine : -INFINITY);
  const float nsum = wave_sum(isneg ? expf(mine - nmax) : 0.0f);
  // first negative lane holding the maximum
  const unsigned long long eq = __ballot(isneg && mine == nmax);
  const int nfirst = eq ? __builtin_ctzll(eq) - Kp : 0;
  for (int i = 0; i < Kp; ++i) {
    const float si = __shfl(mine, i, 64);
    const float m = Kn > 0 ? fmaxf(si, nmax) : si;
    const float sum = expf(si - m) + (Kn > 0 ? nsum * expf(nmax - m) : 0.0f);
    if (lane == 0) {
      lossrow[(long)i * B + b] = -(si - m - logf(sum));
      choice[(long)i * B + b] = (Kn > 0 && nmax > si) ? 1 + nfirst : 0;
    }
  }
}
// one block: L_i = mean_b lossrow[i][b]; loss = mean_i L_i, or min_i (first minimum) with just_use_one (:448-452)
// blocks 1 .. 2 nrun: the running statistics of BatchNorm layer (blockIdx.x - 1) / 2 (they depend on the forward pass only,
// and a launch of their own was 7 us at the end of the step)
__global__ __launch_bounds__(256) void tr_loss_kernel(const float* __restrict__ lossrow, int B, int Kp, int just_one,
                                                      float* __restrict__ loss_out, int32_t* __restrict__ sel, RunArgs run) {
  if (blockIdx.x > 0) {
    tr_bn_running(run, (blockIdx.x - 1) >> 1, ((blockIdx.x - 1) & 1) * 256 + threadIdx.x);
    return;
  }
  __shared__ float Ls[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < Kp; i += 4) {  // a wave per positive: lane partial sums in row order, then the wave tree
    float s0 = 0.0f, s1 = 0.0f;
    int b = lane;
    for (; b + 64 < B; b += 128) {
      s0 += lossrow[(long)i * B + b];
      s1 += lossrow[(long)i * B + b + 64];
    }
    if (b < B) s0 += lossrow[(long)i * B + b];
    const float t = wave_sum(s0 + s1);
    if (lane == 0) Ls[i] = t / (float)B;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (just_one) {
      int k = 0;
      for (int i = 1; i < Kp; ++i)
        if (Ls[i] < Ls[k]) k = i;
      *loss_out = Ls[k];
      *sel = k;
    } else {
      float s = 0.0f;
      for (int i = 0; i < Kp; ++i) s += Ls[i];
      *loss_out = s / (float)Kp;
      *sel = -1;
    }
  }
}
// one wave per row b: d loss / d scores from the softmax rows, then through cos = pu . zu:
//   dP[b] = (sum_k dcos_k zu_k - (sum_k dcos_k cos_k) pu) / |p|;   simplest mode also dZ[1+k][b] = dcos_k (pu - cos_k zu_k) / |z_k|
// (dOut rows: group 0 = dP; in simplest mode groups 1..K = dZ)
__global__ __launch_bounds__(256) void tr_dscore_kernel(const float* __restrict__ P, const float* __restrict__ Z, const float* __restrict__ S,
                                                        const int32_t* __restrict__ sel, int B, int Kp, int Kn, float tau,
                                                        int want_dz, float* __restrict__ dOut) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int K = Kp + Kn, only = *sel;
  const float mine = lane < K ? S[(long)b * K + lane] : -INFINITY;
  const bool isneg = lane >= Kp && lane < K;
  const float nmax = wave_max(isneg ? mine : -INFINITY);
  const float nsum = wave_sum(isneg ? expf(mine - nmax) : 0.0f);
  float ds = 0.0f;
  const float wgt = (only >= 0 ? 1.0f : 1.0f / (float)Kp) / (float)B;
  for (int i = 0; i < Kp; ++i) {
    if (only >= 0 && i != only) continue;
    const float si = __shfl(mine, i, 64);
    const float m = Kn > 0 ? fmaxf(si, nmax) : si;
    const float lse = m + logf(expf(si - m) + (Kn > 0 ? nsum * expf(nmax - m) : 0.0f));
    if (lane == i) ds += wgt * (expf(si - lse) - 1.0f);
    if (isneg) ds += wgt * expf(mine - lse);
  }
  const float dcos = ds / tau;
  const f32x4 p = *(const f32x4*)(P + (long)b * 256 + lane * 4);
  const float pn = fmaxf(sqrtf(wave_sum(p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w)), 1e-8f);
  const f32x4 pu = p / pn;
  f32x4 acc = {0, 0, 0, 0};
  float csum = 0.0f;
  for (int k0 = 0; k0 < K; k0 += 4) {
    f32x4 zb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) zb[u] = *(const f32x4*)(Z + ((long)(1 + min(k0 + u, K - 1)) * B + b) * 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u;
      if (k < K) {
        const f32x4 z = zb[u];
        const float zn = fmaxf(sqrtf(wave_sum(z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w)), 1e-8f);
        const f32x4 zu = z / zn;
        const float cs = wave_sum(pu.x * zu.x + pu.y * zu.y + pu.z * zu.z + pu.w * zu.w);
        const float dk = __shfl(dcos, k, 64);
        acc += dk * zu;
        csum = fmaf(dk, cs, csum);
        if (want_dz) *(f32x4*)(dOut + ((long)(1 + k) * B + b) * 256 + lane * 4) = dk * (pu - cs * zu) / zn;
      }
    }
  }
  *(f32x4*)(dOut + (long)b * 256 + lane * 4) = (acc - csum * pu) / pn;
}

// torch.optim.Adam, defaults (no weight decay, no amsgrad)
__global__ __launch_bounds__(256) void tr_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, long n, float b1, float b2, float eps,
                                                      float lr_over_bc1, float inv_sqrt_bc2) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float ge = g[e];
  const float me = m[e] + (1.0f - b1) * (ge - m[e]);
  const float ve = v[e] * b2 + (1.0f - b2) * ge * ge;
  m[e] = me;
  v[e] = ve;
  p[e] -= lr_over_bc1 * (me / (sqrtf(ve) * inv_sqrt_bc2 + eps));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {
struct TrainWs {
  float *H1, *A1, *H2, *A2, *Z, *mean1, *var1, *mean2, *var2;
  float *HP, *Q, *P, *pmean, *pvar, *S, *lossrow, *dP, *dZ, *d512a, *d512b, *dQ, *sums, *slabs, *colpart;
  int32_t *choice, *sel;
  size_t bytes;
};
inline size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }
constexpr int TR_MAX_SLABS = 16;
TrainWs train_layout(char* base, int B, int Kp, int Kn, bool simplest) {
  const size_t G = 1 + Kp + Kn, Gb = simplest ? G : 1, GB = G * B, GbB = Gb * B;
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = (float*)(base + off);
    off += up64(nfloat) * 4;
    return p;
  };
  TrainWs w;
  w.H1 = take(GB * 512); w.A1 = take(GB * 512); w.H2 = take(GB * 512); w.A2 = take(GB * 512); w.Z = take(GB * 256);
  w.mean1 = take(G * 512); w.var1 = take(G * 512); w.mean2 = take(G * 512); w.var2 = take(G * 512);
  w.HP = take((size_t)B * 100); w.Q = take((size_t)B * 100); w.P = take((size_t)B * 256); w.pmean = take(128); w.pvar = take(128);
  w.S = take((size_t)B * 64); w.lossrow = take((size_t)B * 64);
  w.dP = take((size_t)B * 256); w.dZ = take(GbB * 256); w.d512a = take(GbB * 512); w.d512b = take(GbB * 512); w.dQ = take((size_t)B * 100);
  w.sums = take(Gb * 2 * 512); w.slabs = take((size_t)TR_MAX_SLABS * 512 * 512); w.colpart = take((size_t)TR_MAX_SLABS * 512);
  w.choice = (int32_t*)take((size_t)B * 64); w.sel = (int32_t*)take(64);
  w.bytes = off;
  return w;
}
int tr_slabs(int R) {
  int s = R / 256;
  return s < 1 ? 1 : (s > TR_MAX_SLABS ? TR_MAX_SLABS : s);
}
bool gemm_big(int M, int N, int nslab) { return (long)((N + 127) / 128) * ((M + 127) / 128) * nslab >= 192; }
TgArgs gemm_args(const float* A, long sai, long sar, const float* Bm, long sbr, long sbj, float* C, int ldc, const float* bias, int M,
                 int N, int R, int nslab = 1, long slab = 0, float* colsum = nullptr) {
  TgArgs a{A, Bm, C, bias, M, N, R, sai, sar, sbr, sbj, ldc, 0, slab, colsum};
  a.rper = ((R + nslab - 1) / nslab + 31) / 32 * 32;
  return a;
}
// four elements a load when every group of four along an operand's unit-stride axis is whole and 16-byte aligned
bool gemm_vec(const TgArgs& a) {
  auto ok = [&](const float* P, long s_tile, long s_r, int tile_extent) {
    if (((uintptr_t)P & 15) != 0) return false;
    if (s_r == 1) return s_tile % 4 == 0 && a.R % 4 == 0;
    return s_tile == 1 && s_r % 4 == 0 && tile_extent % 4 == 0;
  };
  return !g_tr_scalar_gemm && ok(a.A, a.sai, a.sar, a.M) && ok(a.B, a.sbj, a.sbr, a.N);
}
bool gemm_wide(int M, int N, int nslab) { return (long)((N + 63) / 64) * ((M + 63) / 64) * nslab >= 128; }
int gemm(hipStream_t st, const float* A, long sai, long sar, const float* Bm, long sbr, long sbj, float* C, int ldc, const float* bias,
         int M, int N, int R, int nslab = 1, long slab = 0, float* colsum = nullptr) {
  const TgArgs a = gemm_args(A, sai, sar, Bm, sbr, sbj, C, ldc, bias, M, N, R, nslab, slab, colsum);
  if (gemm_big(M, N, nslab)) {
    if (colsum) return ST_EINVAL;  // (the 128 x 128 form has no fused column sum: weight_grads checks)
    if (gemm_vec(a)) hipLaunchKernelGGL(tg_gemm128_kernel<true>, dim3((N + 127) / 128, (M + 127) / 128, nslab), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(tg_gemm128_kernel<false>, dim3((N + 127) / 128, (M + 127) / 128, nslab), dim3(256), 0, st, a);
  } else {
    const bool vec = gemm_vec(a), wide = gemm_wide(M, N, nslab);
    const dim3 grid = wide ? dim3((N + 63) / 64, (M + 63) / 64, nslab) : dim3((N + 31) / 32, (M + 31) / 32, nslab);
    if (wide && vec) hipLaunchKernelGGL((tg_gemm_kernel<64, 1, 1, true>), grid, dim3(256), 0, st, a);
    else if (wide) hipLaunchKernelGGL((tg_gemm_kernel<64, 1, 1, false>), grid, dim3(256), 0, st, a);
    else if (vec) hipLaunchKernelGGL((tg_gemm_kernel<32, 4, 3, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((tg_gemm_kernel<32, 4, 3, false>), grid, dim3(256), 0, st, a);
  }
  ST_LAUNCH_CHECK();
  return ST_OK;
}
// dW[N,K] = dY[rows,N]^T . X[rows,K]  and  db[N] = column sums of dY, both reduced over row slabs in a fixed order
int weight_grads(hipStream_t st, const TrainWs& w, const float* dY, const float* X, int rows, int N, int K, float* dW, float* db) {
  const int ns = tr_slabs(rows);
  const bool fused = !gemm_big(N, K, ns);  // small products: the bias gradient comes out of the same launch
  float* part = ns == 1 ? db : w.colpart;
  int rc = gemm(st, dY, 1, N, X, K, 1, ns == 1 ? dW : w.slabs, K, nullptr, N, K, rows, ns, (long)N * K, fused ? part : nullptr);
  if (rc != ST_OK) return rc;
  if (ns > 1) {
    hipLaunchKernelGGL(tr_slab_sum_kernel, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, st, w.slabs, ns, (long)N * K, (long)N * K, dW);
    ST_LAUNCH_CHECK();
  }
  if (!fused) {
    const int rper = (rows + ns - 1) / ns;
    hipLaunchKernelGGL(tr_colsum_kernel, dim3((N + 63) / 64, ns), dim3(256), 0, st, dY, rows, N, rper, part);
    ST_LAUNCH_CHECK();
  }
  if (ns > 1) {
    hipLaunchKernelGGL(tr_slab_sum_kernel, dim3((N + 255) / 256), dim3(256), 0, st, w.colpart, ns, (long)N, (long)N, db);
    ST_LAUNCH_CHECK();
  }
  return ST_OK;
}
// one layer's backward products: dW = dY^T X (+ db) and dXin = dY W, W = [N, K].  While both are a few tiles they go out
// as one launch; otherwise as weight_grads + gemm.
int layer_grads(hipStream_t st, const TrainWs& w, const float* dY, const float* X, const float* Wt, int rows, int N, int K, float* dW,
                float* db, float* dXin) {
  const int ns = tr_slabs(rows);
  if (ns == 1 && !gemm_big(N, K, 1) && !gemm_wide(N, K, 1) && !gemm_big(rows, K, 1) && !gemm_wide(rows, K, 1)) {
    const TgArgs a = gemm_args(dY, 1, N, X, K, 1, dW, K, nullptr, N, K, rows, 1, 0, db);
    const TgArgs b = gemm_args(dY, N, 1, Wt, K, 1, dXin, K, nullptr, rows, K, N);
    const bool vec = gemm_vec(a) && gemm_vec(b);
    const int gxa = (a.N + 31) / 32, na = gxa * ((a.M + 31) / 32), gxb = (b.N + 31) / 32, nb = gxb * ((b.M + 31) / 32);
    if (vec) hipLaunchKernelGGL((tg_gemm_pair_kernel<true>), dim3(na + nb), dim3(256), 0, st, a, b, na, gxa, gxb);
    else hipLaunchKernelGGL((tg_gemm_pair_kernel<false>), dim3(na + nb), dim3(256), 0, st, a, b, na, gxa, gxb);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  int rc = weight_grads(st, w, dY, X, rows, N, K, dW, db);
  if (rc != ST_OK) return rc;
  return gemm(st, dY, N, 1, Wt, K, 1, dXin, K, nullptr, rows, K, N);
}
}  // namespace

extern "C" int st_scorer_train_workspace_bytes(int B, int Kp, int Kn, int flags, size_t* bytes) {
  ST_CHECK_ARG(bytes && B >= 2 && Kp >= 1 && Kn >= 1 && Kp + Kn <= 64, "B=%d Kp=%d Kn=%d (B >= 2, Kp >= 1, Kn >= 1, Kp+Kn <= 64)", B, Kp, Kn);
  *bytes = train_layout(nullptr, B, Kp, Kn, (flags & ST_F_SIMPLEST) != 0).bytes;
  return ST_OK;
}

extern "C" int st_scorer_train_step(const float* params, float* bn, const float* x, int B, int Kp, int Kn, float tau, int flags,
                                    float drop_p, const uint8_t* keep_masks, uint64_t seed, float* grads, float* loss_out,
                                    int32_t* choice_out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(params && bn && x && grads && loss_out && workspace, "NULL buffer");
  ST_CHECK_ARG(B >= 2 && Kp >= 1 && Kn >= 1 && Kp + Kn <= 64, "B=%d Kp=%d Kn=%d (B >= 2: BatchNorm needs a batch; Kp, Kn >= 1: the NLL form needs both; Kp+Kn <= 64)", B, Kp, Kn);
  ST_CHECK_ARG(tau > 0.0f && drop_p >= 0.0f && drop_p < 1.0f, "tau=%g drop_p=%g", tau, drop_p);
  ST_CHECK_ARG((flags & ~(ST_F_SIMPLEST | ST_F_JUST_USE_ONE)) == 0, "flags=%d: only ST_F_SIMPLEST / ST_F_JUST_USE_ONE apply to training", flags);
  const bool simplest = (flags & ST_F_SIMPLEST) != 0;
  tr_env();
  hipStream_t st = (hipStream_t)stream;
  const TrainWs w = train_layout((char*)workspace, B, Kp, Kn, simplest);
  if (w.bytes > workspace_bytes) {
    st_set_error("st_scorer_train_step: workspace %zu bytes < %zu needed", workspace_bytes, w.bytes);
    return ST_ENOMEM;
  }
  const int G = 1 + Kp + Kn, Gb = simplest ? G : 1, GB = G * B, GbB = Gb * B, K = Kp + Kn;
  const float* pr = params;
  float* gr = grads;
  int rc;
#define TR_RUN(call)        \
  do {                      \
    rc = (call);            \
    if (rc != ST_OK) return rc; \
  } while (0)
  auto blocks = [](long n) { return dim3((unsigned)((n + 255) / 256)); };
  // BatchNorm statistics and the activation pass behind them in one launch per layer while a (64 features, call) slab is
  // small; large batches keep the elementwise pass on the whole chip.  (The same fusion of the backward's sums and its
  // elementwise pass measured SLOWER at batch 128, 10.7 against 5.4 + 3.0 us: one call = 8 workgroups walk both passes.)
  const bool fuse_bn = B <= 512;
  // BatchNorm backward sums of `calls` calls: rows split over workgroups until the grid fills the chip
  auto bn_bwd_sums = [&](const BnActArgs& b, int col_blocks, int calls) {
    const int want = (256 + col_blocks * calls - 1) / (col_blocks * calls), most = B / 256 > 1 ? B / 256 : 1;
    const int nz = want < most ? want : most;
    const int rper = ((B + nz - 1) / nz + TR_RG - 1) / TR_RG * TR_RG;
    hipLaunchKernelGGL(tr_bn_bwd_sums_kernel, dim3(col_blocks, calls, nz), dim3(64 * TR_RG), 0, st, b, rper, w.slabs);
    if (nz > 1)
      hipLaunchKernelGGL(tr_bn_bwd_sums_finish_kernel, dim3((calls * 2 * b.N + 255) / 256), dim3(256), 0, st, b, calls, nz, (const float*)w.slabs);
  };
  auto bnargs = [&](const float* H, const float* mean, const float* var, int gofs, int bofs, float* A, int layer, int N, int rows, float slope, float dp) {
    BnActArgs a{};
    a.H = H; a.mean = mean; a.var = var; a.gamma = pr + gofs; a.beta = pr + bofs; a.A = A; a.mask = keep_masks; a.seed = seed;
    a.B = B; a.N = N; a.layer = layer; a.rows = rows; a.slope = slope; a.drop_p = dp;
    return a;
  };
  // ---------------- forward ----------------
  TR_RUN(gemm(st, x, 256, 1, pr + ST_TP_S_FC1_W, 1, 256, w.H1, 512, pr + ST_TP_S_FC1_B, GB, 512, 256));
  BnActArgs f1 = bnargs(w.H1, w.mean1, w.var1, ST_TP_S_BN1_G, ST_TP_S_BN1_B, w.A1, 0, 512, GB, 0.0f, drop_p);
  if (fuse_bn) {
    hipLaunchKernelGGL(tr_bn_fwd_kernel, dim3(8, G), dim3(64 * TR_RG), 0, st, f1, w.mean1, w.var1);
  } else {
    hipLaunchKernelGGL(tr_bn_stats_kernel, dim3(8, G), dim3(64 * TR_RG), 0, st, w.H1, B, 512, w.mean1, w.var1);
    hipLaunchKernelGGL(tr_bn_act_kernel, blocks((long)GB * 512), dim3(256), 0, st, f1);
  }
  TR_RUN(gemm(st, w.A1, 512, 1, pr + ST_TP_S_FC2_W, 1, 512, w.H2, 512, pr + ST_TP_S_FC2_B, GB, 512, 512));
  BnActArgs f2 = bnargs(w.H2, w.mean2, w.var2, ST_TP_S_BN2_G, ST_TP_S_BN2_B, w.A2, 1, 512, GB, 0.0f, drop_p);
  if (fuse_bn) {
    hipLaunchKernelGGL(tr_bn_fwd_kernel, dim3(8, G), dim3(64 * TR_RG), 0, st, f2, w.mean2, w.var2);
  } else {
    hipLaunchKernelGGL(tr_bn_stats_kernel, dim3(8, G), dim3(64 * TR_RG), 0, st, w.H2, B, 512, w.mean2, w.var2);
    hipLaunchKernelGGL(tr_bn_act_kernel, blocks((long)GB * 512), dim3(256), 0, st, f2);
  }
  TR_RUN(gemm(st, w.A2, 512, 1, pr + ST_TP_S_FC3_W, 1, 512, w.Z, 256, pr + ST_TP_S_FC3_B, GB, 256, 512));
  const float* Pv = w.Z;  // simplest: the ori call's output is scored directly
  BnActArgs fp{};
  if (!simplest) {
    TR_RUN(gemm(st, w.Z, 256, 1, pr + ST_TP_P_FC1_W, 1, 256, w.HP, 100, pr + ST_TP_P_FC1_B, B, 100, 256));
    fp = bnargs(w.HP, w.pmean, w.pvar, ST_TP_P_BN_G, ST_TP_P_BN_B, w.Q, 0, 100, B, 0.2f, 0.0f);
    if (fuse_bn) {
      hipLaunchKernelGGL(tr_bn_fwd_kernel, dim3(2, 1), dim3(64 * TR_RG), 0, st, fp, w.pmean, w.pvar);
    } else {
      hipLaunchKernelGGL(tr_bn_stats_kernel, dim3(2, 1), dim3(64 * TR_RG), 0, st, w.HP, B, 100, w.pmean, w.pvar);
      hipLaunchKernelGGL(tr_bn_act_kernel, blocks((long)B * 100), dim3(256), 0, st, fp);
    }
    TR_RUN(gemm(st, w.Q, 100, 1, pr + ST_TP_P_FC2_W, 1, 100, w.P, 256, pr + ST_TP_P_FC2_B, B, 256, 100));
    Pv = w.P;
  }
  hipLaunchKernelGGL(tr_score_kernel, dim3((B + 3) / 4), dim3(256), 0, st, Pv, w.Z, B, Kp, Kn, tau, w.S, w.lossrow,
                     choice_out ? choice_out : w.choice);
  // running statistics: every one of the G calls moved them, in call order
  RunArgs ra{};
  ra.B = B;
  ra.mean[0] = w.mean1; ra.var[0] = w.var1; ra.rm[0] = bn + ST_TB_S_RM1; ra.rv[0] = bn + ST_TB_S_RV1; ra.G[0] = G; ra.N[0] = 512;
  ra.mean[1] = w.mean2; ra.var[1] = w.var2; ra.rm[1] = bn + ST_TB_S_RM2; ra.rv[1] = bn + ST_TB_S_RV2; ra.G[1] = G; ra.N[1] = 512;
  ra.mean[2] = w.pmean; ra.var[2] = w.pvar; ra.rm[2] = bn + ST_TB_P_RM; ra.rv[2] = bn + ST_TB_P_RV; ra.G[2] = 1; ra.N[2] = 100;
  hipLaunchKernelGGL(tr_loss_kernel, dim3(1 + 2 * (simplest ? 2 : 3)), dim3(256), 0, st, w.lossrow, B, Kp,
                     (flags & ST_F_JUST_USE_ONE) ? 1 : 0, loss_out, w.sel, ra);
  ST_LAUNCH_CHECK();
  // ---------------- backward ----------------
  // every Siamese tensor's gradient is written below; the Predictor's only when it takes part
  if (simplest) ST_HIP(hipMemsetAsync(grads + ST_TP_SIAMESE_COUNT, 0, (size_t)(ST_TP_COUNT - ST_TP_SIAMESE_COUNT) * 4, st));
  // d loss / d P (and, simplest mode, d loss / d Z of the like / unlike calls: rows of w.dZ in call order)
  hipLaunchKernelGGL(tr_dscore_kernel, dim3((B + 3) / 4), dim3(256), 0, st, Pv, w.Z, w.S, w.sel, B, Kp, Kn, tau, simplest ? 1 : 0,
                     simplest ? w.dZ : w.dP);
  ST_LAUNCH_CHECK();
  if (!simplest) {
    // Predictor: fc2 <- LeakyReLU <- BatchNorm <- fc1
    TR_RUN(layer_grads(st, w, w.dP, w.Q, pr + ST_TP_P_FC2_W, B, 256, 100, gr + ST_TP_P_FC2_W, gr + ST_TP_P_FC2_B, w.dQ));
    BnActArgs bp = fp;
    bp.dA = w.dQ; bp.sums = w.sums; bp.dH = w.dQ; bp.dbeta = gr + ST_TP_P_BN_B; bp.dgamma = gr + ST_TP_P_BN_G;
    bn_bwd_sums(bp, 2, 1);
    hipLaunchKernelGGL(tr_bn_bwd_apply_kernel, blocks((long)B * 100), dim3(256), 0, st, bp);
    TR_RUN(layer_grads(st, w, w.dQ, w.Z, pr + ST_TP_P_FC1_W, B, 100, 256, gr + ST_TP_P_FC1_W, gr + ST_TP_P_FC1_B, w.dZ));
  }
  // SiameseNet over the Gb calls that carry gradient (the leading Gb*B rows of every activation buffer)
  TR_RUN(layer_grads(st, w, w.dZ, w.A2, pr + ST_TP_S_FC3_W, GbB, 256, 512, gr + ST_TP_S_FC3_W, gr + ST_TP_S_FC3_B, w.d512a));
  BnActArgs b2 = f2;
  b2.rows = GbB; b2.dA = w.d512a; b2.sums = w.sums; b2.dH = w.d512a;
  if (Gb == 1) {
    b2.dbeta = gr + ST_TP_S_BN2_B;
    b2.dgamma = gr + ST_TP_S_BN2_G;
  }
  bn_bwd_sums(b2, 8, Gb);
  hipLaunchKernelGGL(tr_bn_bwd_apply_kernel, blocks((long)GbB * 512), dim3(256), 0, st, b2);
  if (Gb > 1) {  // d beta / d gamma: sum over the calls of sums[g][0] / sums[g][1]
    hipLaunchKernelGGL(tr_slab_sum_kernel, dim3(2), dim3(256), 0, st, w.sums, Gb, 1024L, 512L, gr + ST_TP_S_BN2_B);
    hipLaunchKernelGGL(tr_slab_sum_kernel, dim3(2), dim3(256), 0, st, w.sums + 512, Gb, 1024L, 512L, gr + ST_TP_S_BN2_G);
  }
  ST_LAUNCH_CHECK();
  TR_RUN(layer_grads(st, w, w.d512a, w.A1, pr + ST_TP_S_FC2_W, GbB, 512, 512, gr + ST_TP_S_FC2_W, gr + ST_TP_S_FC2_B, w.d512b));
  BnActArgs b1 = f1;
  b1.rows = GbB; b1.dA = w.d512b; b1.sums = w.sums; b1.dH = w.d512b;
  if (Gb == 1) {
    b1.dbeta = gr + ST_TP_S_BN1_B;
    b1.dgamma = gr + ST_TP_S_BN1_G;
  }
  bn_bwd_sums(b1, 8, Gb);
  hipLaunchKernelGGL(tr_bn_bwd_apply_kernel, blocks((long)GbB * 512), dim3(256), 0, st, b1);
  if (Gb > 1) {  // d beta / d gamma: sum over the calls of sums[g][0] / sums[g][1]
    hipLaunchKernelGGL(tr_slab_sum_kernel, dim3(2), dim3(256), 0, st, w.sums, Gb, 1024L, 512L, gr + ST_TP_S_BN1_B);
    hipLaunchKernelGGL(tr_slab_sum_kernel, dim3(2), dim3(256), 0, st, w.sums + 512, Gb, 1024L, 512L, gr + ST_TP_S_BN1_G);
  }
  ST_LAUNCH_CHECK();
  TR_RUN(weight_grads(st, w, w.d512b, x, GbB, 512, 256, gr + ST_TP_S_FC1_W, gr + ST_TP_S_FC1_B));
  ST_LAUNCH_CHECK();
#undef TR_RUN
  (void)K;
  return ST_OK;
}

extern "C" int st_adam_step(float* params, const float* grads, float* adam_m, float* adam_v, int64_t n, float lr, float beta1,
                            float beta2, float eps, int step, void* stream) {
  ST_CHECK_ARG(n >= 0 && step >= 1 && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "n=%lld step=%d beta=(%g, %g)",
               (long long)n, step, beta1, beta2);
  if (n == 0) return ST_OK;
  ST_CHECK_ARG(params && grads && adam_m && adam_v, "NULL buffer");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(tr_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, adam_m, adam_v,
                     (long)n, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
  ST_LAUNCH_CHECK();
  return ST_OK;
}
