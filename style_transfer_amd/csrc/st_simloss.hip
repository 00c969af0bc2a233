// In-batch InfoNCE similarity loss, forward and backward (the B x B block of
// train_ae_model_constraint, /root/reference/method/yelp/siamese_contrastive.py:1060-1069:
// sim = cosine_similarity(ori[:,None,:], like[None,:,:]) / tau; log_softmax; NLL(target = arange(B))).
// The reference materialises a [B,B,256] broadcast (68 GB at B = 8192); here
//   fwd: unit-normalise rows once, then one panel GEMM whose epilogue keeps an online
//        (max, sum-exp, target) per row — the [Bq,Bk] matrix never exists;
//   bwd: dS is produced in row chunks of <= 1024 rows (<= 32 MiB at Bk = 8192: Infinity-Cache
//        resident) by recomputing S from the saved logsumexp, and consumed at once by two GEMMs:
//        d_ou = dS . ku   and   d_ku += dS^T . ou.
#include "st_rowpanel.h"

#define SL_SPLITS 4
#ifndef SL_CHUNK
#define SL_CHUNK 2048
#endif
#define SL_KSPLIT 16

// u = x / max(||x||, 1e-8), nrm = max(||x||, 1e-8); wave per row.  ut (optional) = u transposed [256, n]
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* __restrict__ x, int n, float* __restrict__ u,
                                                        float* __restrict__ nrm, float* __restrict__ ut) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  f32x4 v = *(const f32x4*)(x + (size_t)r * 256 + lane * 4);
  const float nn = fmaxf(sqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w)), 1e-8f);
  v = v / nn;
  *(f32x4*)(u + (size_t)r * 256 + lane * 4) = v;
  if (lane == 0) nrm[r] = nn;
  if (ut) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ut[(size_t)(lane * 4 + j) * n + r] = v[j];
  }
}

__global__ __launch_bounds__(256) void lse_merge_kernel(const float4* __restrict__ partial, int nsplit, int n,
                                                        float* __restrict__ loss, float* __restrict__ lse) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  Lse t = {-INFINITY, 0.0f};
  float tg = 0.0f;
  for (int k = 0; k < nsplit; ++k) {
    const float4 p = partial[(size_t)r * nsplit + k];
    Lse u = {p.x, p.y};
    t = lse_merge(t, u);
    tg += p.z;
  }
  const float l = t.m + logf(t.l);
  lse[r] = l;
  loss[r] = l - tg;
}

// dx = (du - (du . u) u) / nrm  (backward of x -> x / max(||x||, eps));  out = or += dx
__global__ __launch_bounds__(256) void unit_rows_bwd_kernel(const float* __restrict__ du, const float* __restrict__ u,
                                                            const float* __restrict__ nrm, int n,
                                                            float* __restrict__ dx, int accumulate,
                                                            const float* __restrict__ rowscale) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const f32x4 g = *(const f32x4*)(du + (size_t)r * 256 + lane * 4);
  const f32x4 v = *(const f32x4*)(u + (size_t)r * 256 + lane * 4);
  const float d = wave_sum(g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w);
  f32x4 o = (g - d * v) * ((rowscale ? rowscale[r] : 1.0f) / nrm[r]);
  float* p = dx + (size_t)r * 256 + lane * 4;
  if (accumulate) o += *(const f32x4*)p;
  *(f32x4*)p = o;
}

// ut[c][r] = s[r] * u[r][c]  (rows scaled by g / tau, transposed: the B operand of d_ku = dS^T . (s * ou))
__global__ __launch_bounds__(256) void scale_transpose_kernel(const float* __restrict__ u, const float* __restrict__ g, float scale,
                                                              int n, float* __restrict__ ut, float* __restrict__ rowscale) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const float sc = g[r] * scale;
  const f32x4 v = *(const f32x4*)(u + (size_t)r * 256 + lane * 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) ut[(size_t)(lane * 4 + j) * n + r] = sc * v[j];
  if (lane == 0) rowscale[r] = sc;
}

// ------------------------------------------------------------------------------------------------
// Small blocks (Bq, Bk <= SL_SMALL: the reference's training batch of 128, the goldens' 32): one kernel per pass, wave per
// row, plain f32 FMAs -- a 128 x 128 block is 4 MFLOP, and the GEMM path above spends 5 launches forward and 9 backward
// (~ 11 us each) on it.  Same formulas: rows normalised to unit length, s = u . v / tau, online log-sum-exp; backward
// d u_i = sum_j g_i (softmax_ij - [j = target]) / tau * v_j through the normalisation's Jacobian, and the same with the
// roles swapped for the keys (d_like is ADDED to, d_ori written, as in the GEMM path).
// ------------------------------------------------------------------------------------------------
#define SL_SMALL 128  /* 256: the wave-per-row loops get long, the AE step at B = 256 measured 8.4 ms against 7.4 on the GEMM path */
__device__ __forceinline__ f32x4 sl_unit(const float* __restrict__ x, int r, int lane, float& nrm) {
  f32x4 v = *(const f32x4*)(x + (size_t)r * 256 + lane * 4);
  nrm = fmaxf(sqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w)), 1e-8f);
  return v / nrm;
}
__device__ __forceinline__ float sl_dot(const f32x4& a, const f32x4& b) { return wave_sum(a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w); }
// blockIdx.y = view: its keys at like + y * like_vs, its outputs at loss / lse + y * out_vs (sl_small_views: the views of
// the AE constraint share their queries and go out as one launch; a single block has strides 0 and gridDim.y = 1)
__global__ __launch_bounds__(256) void sl_small_fwd_kernel(const float* __restrict__ ori, const float* __restrict__ like, int Bq, int Bk,
                                                           float scale, int row_offset, float* __restrict__ loss, float* __restrict__ lse,
                                                           long like_vs, int out_vs) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= Bq) return;
  like += blockIdx.y * like_vs;
  loss += blockIdx.y * out_vs;
  lse += blockIdx.y * out_vs;
  const int lane = threadIdx.x & 63;
  float nn;
  const f32x4 u = sl_unit(ori, i, lane, nn);
  Lse t = {-INFINITY, 0.0f};
  float tg = 0.0f;
  auto fold = [&](int j, float sv) {
    const float d = sv - t.m;
    const float e = expf(-fabsf(d));
    t.l = d > 0.0f ? fmaf(t.l, e, 1.0f) : t.l + e;
    t.m = fmaxf(t.m, sv);
    if (j == row_offset + i) tg = sv;
  };
  int j = 0;
  for (; j + 1 < Bk; j += 2) {  // two keys at a time: two independent reduction chains in flight
    float n0, n1;
    const f32x4 v0 = sl_unit(like, j, lane, n0), v1 = sl_unit(like, j + 1, lane, n1);
    const float s0 = sl_dot(u, v0) * scale, s1 = sl_dot(u, v1) * scale;
    fold(j, s0);
    fold(j + 1, s1);
  }
  if (j < Bk) {
    float n0;
    fold(j, sl_dot(u, sl_unit(like, j, lane, n0)) * scale);
  }
  if (lane == 0) {
    const float l = t.m + logf(t.l);
    lse[i] = l;
    loss[i] = l - tg;
  }
}
// blocks [0, ceil(Bq / 4)): wave per query -> d_ori;  the rest: wave per key -> d_like +=
__global__ __launch_bounds__(256) void sl_small_bwd_kernel(const float* __restrict__ ori, const float* __restrict__ like, int Bq, int Bk,
                                                           float scale, int row_offset, const float* __restrict__ lse,
                                                           const float* __restrict__ g, float* __restrict__ d_ori, float* __restrict__ d_like,
                                                           long like_vs, int lse_vs, long dori_vs) {
  like += blockIdx.y * like_vs;
  d_like += blockIdx.y * like_vs;
  lse += blockIdx.y * lse_vs;
  d_ori += blockIdx.y * dori_vs;
  const int qblocks = (Bq + 3) / 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  if ((int)blockIdx.x < qblocks) {
    const int i = blockIdx.x * 4 + w;
    if (i >= Bq) return;
    float nn;
    const f32x4 u = sl_unit(ori, i, lane, nn);
    const float li = lse[i], gi = g[i] * scale;
    auto term = [&](int j, const f32x4& v, float dot) {
      const float p = expf(dot * scale - li) - (j == row_offset + i ? 1.0f : 0.0f);
      acc += (gi * p) * v;
    };
    int j = 0;
    for (; j + 1 < Bk; j += 2) {
      float n0, n1;
      const f32x4 v0 = sl_unit(like, j, lane, n0), v1 = sl_unit(like, j + 1, lane, n1);
      const float d0 = sl_dot(u, v0), d1 = sl_dot(u, v1);
      term(j, v0, d0);
      term(j + 1, v1, d1);
    }
    if (j < Bk) {
      float n0;
      const f32x4 v0 = sl_unit(like, j, lane, n0);
      term(j, v0, sl_dot(u, v0));
    }
    const f32x4 o = (acc - sl_dot(acc, u) * u) / nn;
    *(f32x4*)(d_ori + (size_t)i * 256 + lane * 4) = o;
  } else {
    const int j = (blockIdx.x - qblocks) * 4 + w;
    if (j >= Bk) return;
    float nk;
    const f32x4 v = sl_unit(like, j, lane, nk);
    auto term = [&](int i, const f32x4& u, float dot) {
      const float p = expf(dot * scale - lse[i]) - (j == row_offset + i ? 1.0f : 0.0f);
      acc += (g[i] * scale * p) * u;
    };
    int i = 0;
    for (; i + 1 < Bq; i += 2) {
      float n0, n1;
      const f32x4 u0 = sl_unit(ori, i, lane, n0), u1 = sl_unit(ori, i + 1, lane, n1);
      const float d0 = sl_dot(u0, v), d1 = sl_dot(u1, v);
      term(i, u0, d0);
      term(i + 1, u1, d1);
    }
    if (i < Bq) {
      float n0;
      const f32x4 u0 = sl_unit(ori, i, lane, n0);
      term(i, u0, sl_dot(u0, v));
    }
    float* dst = d_like + (size_t)j * 256 + lane * 4;
    *(f32x4*)dst = *(const f32x4*)dst + (acc - sl_dot(acc, v) * v) / nk;
  }
}
int g_sl_small = 1;  // 0: small blocks through the GEMM path as well (tests flip it)
// internal (st_common.h): n_views blocks of B queries x B keys that share the queries, keys of view v at like0 + v B 256
// (the AE constraint, st_aetrain.hip): one forward and one backward launch for all of them instead of one pair per view
// (at B = 128 a view's launches are 32 + 64 workgroups of 49 + 38 us: eleven views in a row were 18 % of the step).
// loss_rows, lse: [n_views][B]; d_ori_parts: [n_views][B][256] (the caller adds them up in view order); d_like0 += per view.
// blocks up to this size go through sl_small_views (all views in one launch pair).  Beyond SL_SMALL a single view is faster
// on the GEMM path, but eleven of them in one launch fill the chip: AE step at B = 256 6.92 -> 6.65 ms, B = 384 10.97 -> 9.37
// (384 is no multiple of 256, its GEMM path is the chunked one); at B = 512 the wave-per-row loops lose, 11.05 -> 11.91
// (tools/ae_views_limit_probe.py)
int g_sl_small_views_max = 384;
bool sl_small_views_ok(int B) { return g_sl_small && B <= g_sl_small_views_max && B % 32 == 0; }
int sl_small_views(const float* ori, const float* like0, int n_views, int B, float tau, float* loss_rows, float* lse, const float* g,
                   float* d_ori_parts, float* d_like0, hipStream_t st) {
  ST_CHECK_ARG(sl_small_views_ok(B) && n_views >= 1 && n_views <= 65535 && tau > 0.0f, "sl_small_views: B=%d n_views=%d", B, n_views);
  const long vs = (long)B * 256;
  hipLaunchKernelGGL(sl_small_fwd_kernel, dim3((B + 3) / 4, n_views), dim3(256), 0, st, ori, like0, B, B, 1.0f / tau, 0, loss_rows, lse, vs, B);
  hipLaunchKernelGGL(sl_small_bwd_kernel, dim3(2 * ((B + 3) / 4), n_views), dim3(256), 0, st, ori, like0, B, B, 1.0f / tau, 0,
                     (const float*)lse, g, d_ori_parts, d_like0, vs, B, vs);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

// Operand format of the similarity GEMMs (ST_GEMM_*).  F16X3 (default): unit rows lie in [-1, 1] and dS in [-1, 1], the
// ideal domain of the hi + lo split (f32-grade: ~2^-22 per term), and every product of the block runs on the f16 MFMA.
static int g_sl_format = ST_GEMM_F16X3;
int g_sl_fused = getenv("ST_AB_ENABLE") && getenv("ST_AB_SIMLOSS_CHUNKED") ? 0 : 1;  // A/B: 0 = the round-2 chunked backward
extern "C" int st_simloss_format(int gemm_format) {
  const int prev = g_sl_format;
  if (gemm_format == ST_GEMM_F32 || gemm_format == ST_GEMM_F16X3) g_sl_format = gemm_format;
  return prev;
}
static int sl_pack(const float* W, int N, int K, float* out, hipStream_t st) {
  if (g_sl_format == ST_GEMM_F16X3) {
    const size_t total = (size_t)((N + 31) / 32) * (K >> 4) * 64;
    hipLaunchKernelGGL(pack_weight_h_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, out);
  } else {
    const size_t total = (size_t)((N + 31) / 32) * (K >> 3) * 64;
    hipLaunchKernelGGL(pack_weight_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, out);
  }
  ST_LAUNCH_CHECK();
  return ST_OK;
}


// ------------------------------------------------------------------------------------------------
// Fused backward block (f16x3 format, round 3): for a 64-row panel of one side's unit rows and a range of 256-column
// chunks of the other side,
//     S  = panel . other^T            (split-f16 operands, three products)        [64, 256] per chunk
//     dS = exp(S / tau - lse) - [target]                                          in registers -> LDS as split f16
//     acc += dS . other'                                                          [64, 256], other' = the other side's rows
// i.e. the structure of the decoder's half-layer kernel (rp_layer_tail16_kernel, st_rowpanel.h) with the ReLU replaced by
// the softmax gradient and the hidden width by the other side's row count: dS never goes to HBM (the chunked form below
// wrote and re-read 2 x 64 MB per 2048 rows).  Two launches per view: TR = false, panel = queries (ou), lse per ROW,
// result d_ou;  TR = true, panel = keys (ku), lse per COLUMN, other' = s_i ou_i, result d_ku.  v_mfma_f32_16x16x32_f16,
// W as the A operand; lane (r16 = l & 15, g = l >> 4) holds columns 4g..4g+3 (+16 ft) of panel row r16 (+16 rt).
// Column splits (blockIdx.y) write partial sums to a slab that slab_reduce_kernel adds in a fixed order.
// ------------------------------------------------------------------------------------------------
struct SlbArgs {
  const float* X;      // [n, 256] unit rows of the panel side
  int n;               // rows (multiple of 64)
  const float* w1;     // packed f16x3 image of the other side  [Ncols][256]
  const float* w2;     // packed f16x3 image of other'^T        [256][Ncols]
  int Ncols;           // multiple of 256
  int chunks_per_split;
  const float* lse;    // TR ? [Ncols] : [n]
  float scale;         // 1 / tau
  int tgt_off;         // target column of panel row R: R + tgt_off
  float* out;          // [gridDim.y][n][256] partial sums
};
#define SLB_LDS (2 * 64 * ST_LDA * 4)

template <bool TR>
__global__ __launch_bounds__(512, 2) void sl_bwd16_kernel(SlbArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* P1 = smem;
  float* P2 = smem + 64 * ST_LDA;
  const int m0 = blockIdx.x * 64;
  const int nchunk = a.Ncols >> 8;
  const int c_lo = blockIdx.y * a.chunks_per_split;
  const int c_hi = min(nchunk, c_lo + a.chunks_per_split);
  if (m0 >= a.n || c_lo >= c_hi) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int fcol = wave * 32 + 4 * g;
  const int wlane = (g >> 1) * 128 + 32 * (g & 1) + r16;
  auto W1p = [&](int c) { return (const f32x4*)a.w1 + (size_t)(c * 8 + wave) * 16 * 128 + wlane; };
  auto W2p = [&](int c) { return (const f32x4*)a.w2 + ((size_t)wave * (a.Ncols >> 4) + c * 16) * 128 + wlane; };
  f32x4 rw[2][4];
  auto wfetch = [&](const f32x4* Wl, int d, f32x4 (&dst)[4]) {
    dst[0] = Wl[d * 256];
    dst[1] = Wl[d * 256 + 64];
    dst[2] = Wl[d * 256 + 16];
    dst[3] = Wl[d * 256 + 80];
  };
  wfetch(W1p(c_lo), 0, rw[0]);
  wfetch(W1p(c_lo), 1, rw[1]);
  int bad = 0;
  {
    f32x4 xr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 512;
      xr[i] = *(const f32x4*)(a.X + (size_t)(m0 + (f >> 6)) * 256 + (f & 63) * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 512;
      panel_store4_nb16(P1 + (f >> 6) * ST_LDA, f & 63, xr[i], bad);
    }
  }
  float lrow[4];
  int tgt[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int R = m0 + 16 * rt + r16;
    lrow[rt] = TR ? 0.0f : a.lse[R];
    tgt[rt] = R + a.tgt_off;
  }
  f32x4 lcc[2], lcn[2];  // TR: lse of this lane's 8 columns, chunk being converted / next chunk
  auto lc_fetch = [&](int c, f32x4 (&dst)[2]) {
    if (TR) {
      dst[0] = *(const f32x4*)(a.lse + (c * 8 + wave) * 32 + 4 * g);
      dst[1] = *(const f32x4*)(a.lse + (c * 8 + wave) * 32 + 16 + 4 * g);
    }
  };
  lc_fetch(c_lo, lcc);
  const float* A1 = P1 + r16 * ST_LDA + rp16_frag_pos(0, g);
  const float* A2 = P2 + r16 * ST_LDA + rp16_frag_pos(0, g);
  _Float16* H2 = (_Float16*)P2;
  int cconv = c_lo;  // the chunk whose S tile is being turned into dS
  auto epi_unit = [&](const f32x4& acc, int ft, int rt) {
    const int col0 = (cconv * 8 + wave) * 32 + 16 * ft + 4 * g;
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x = fminf(fmaf(acc[q], a.scale, -(TR ? lcc[ft][q] : lrow[rt])), 10.0f);  // <= 0 up to rounding for a matching lse
      v[q] = __expf(x) - (col0 + q == tgt[rt] ? 1.0f : 0.0f);
    }
    h4 hi, lo;
    split_f16(v, hi, lo);
    _Float16* dst = H2 + (size_t)(16 * rt + r16) * (2 * ST_LDA) + 2 * rp16_quad_pos((fcol + 16 * ft) >> 2);
    *(h4*)dst = hi;
    *(h4*)(dst + 256) = lo;
  };
  auto pass = [&](const float* Pb, const f32x4* Wc, const f32x4* Wn, f32x4 (&acc)[2][4], const bool epi, const f32x4 (&o)[2][4]) {
    f32x4 pn[4];
    pn[0] = *(const f32x4*)(Pb);
    pn[1] = *(const f32x4*)(Pb + 128);
    pn[2] = *(const f32x4*)(Pb + 16 * ST_LDA);
    pn[3] = *(const f32x4*)(Pb + 16 * ST_LDA + 128);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int slot = d & 1;
      const h8 wh0 = as_h8(rw[slot][0]), wl0 = as_h8(rw[slot][1]), wh1 = as_h8(rw[slot][2]), wl1 = as_h8(rw[slot][3]);
      if (d + 2 < 8)
        wfetch(Wc, d + 2, rw[slot]);
      else
        wfetch(Wn, d + 2 - 8, rw[slot]);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const h8 p0h = as_h8(pn[0]), p0l = as_h8(pn[1]), p1h = as_h8(pn[2]), p1l = as_h8(pn[3]);
        if (!(d == 7 && hf == 1)) {
          const int nd = hf ? d + 1 : d, nrt = hf ? 0 : 2;
          const float* q = Pb + nrt * 16 * ST_LDA + 8 * nd;
          pn[0] = *(const f32x4*)(q);
          pn[1] = *(const f32x4*)(q + 128);
          pn[2] = *(const f32x4*)(q + 16 * ST_LDA);
          pn[3] = *(const f32x4*)(q + 16 * ST_LDA + 128);
        }
        const int rt = 2 * hf;
        acc[0][rt] = mfma16x32h(wh0, p0h, acc[0][rt]);
        acc[1][rt] = mfma16x32h(wh1, p0h, acc[1][rt]);
        acc[0][rt + 1] = mfma16x32h(wh0, p1h, acc[0][rt + 1]);
        acc[1][rt + 1] = mfma16x32h(wh1, p1h, acc[1][rt + 1]);
        acc[0][rt] = mfma16x32h(wl0, p0h, acc[0][rt]);
        acc[1][rt] = mfma16x32h(wl1, p0h, acc[1][rt]);
        acc[0][rt + 1] = mfma16x32h(wl0, p1h, acc[0][rt + 1]);
        acc[1][rt + 1] = mfma16x32h(wl1, p1h, acc[1][rt + 1]);
        acc[0][rt] = mfma16x32h(wh0, p0l, acc[0][rt]);
        acc[1][rt] = mfma16x32h(wh1, p0l, acc[1][rt]);
        acc[0][rt + 1] = mfma16x32h(wh0, p1l, acc[0][rt + 1]);
        acc[1][rt + 1] = mfma16x32h(wh1, p1l, acc[1][rt + 1]);
        if (epi && hf == 1) epi_unit(o[d & 1][d >> 1], d & 1, d >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto zero = [&](f32x4 (&h)[2][4]) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[ft][rt] = f32x4{0, 0, 0, 0};
  };
  f32x4 ax[2][4], ah[2][4];
  zero(ax);
  zero(ah);
  __syncthreads();  // the panel is in P1
  pass(A1, W1p(c_lo), c_lo + 1 < c_hi ? W1p(c_lo + 1) : W2p(c_lo), ah, false, ah);
  for (int c = c_lo; c + 1 < c_hi; ++c) {
    f32x4 an[2][4];
    zero(an);
    lc_fetch(c + 1, lcn);
    if (c > c_lo) __syncthreads();  // every wave is done reading the previous dS chunk
    cconv = c;
    pass(A1, W1p(c + 1), W2p(c), an, true, ah);  // S of chunk c + 1; chunk c -> dS -> P2 in the MFMAs' shadow
    __syncthreads();                             // dS chunk complete
    pass(A2, W2p(c), c + 2 < c_hi ? W1p(c + 2) : W2p(c + 1), ax, false, ax);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) ah[ft][rt] = an[ft][rt];
    if (TR) {
      lcc[0] = lcn[0];
      lcc[1] = lcn[1];
    }
  }
  {
    const int c = c_hi - 1;
    if (c > c_lo) __syncthreads();
    cconv = c;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) epi_unit(ah[ft][rt], ft, rt);
    __syncthreads();
    pass(A2, W2p(c), W2p(c), ax, false, ax);  // (the ring's last prefetch re-reads this chunk's first steps and drops them)
  }
  float* o = a.out + ((size_t)blockIdx.y * a.n + m0) * 256;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    *(f32x4*)(o + (size_t)(16 * rt + r16) * 256 + fcol) = ax[0][rt];
    *(f32x4*)(o + (size_t)(16 * rt + r16) * 256 + fcol + 16) = ax[1][rt];
  }
  (void)bad;  // unit rows lie in [-1, 1]
}

// rows: panel side (multiple of 64), cols: other side (multiple of 256); dst [rows, 256]
template <bool TR>
static int sl_bwd_fused(const float* Xrows, int rows, const float* W1packed, const float* W2packed, int cols, const float* lse,
                        float scale, int tgt_off, float* slab, size_t slab_rows, float* dst, hipStream_t st) {
  ST_ENSURE_LDS((sl_bwd16_kernel<TR>), SLB_LDS);
  const int panels = rows / 64, nchunk = cols / 256;
  const int cus = st_cu_count();
  // column splits: about one workgroup per CU (133 KiB of LDS: one resident workgroup), whole chunks, slab permitting
  int splits = std::max(1, std::min(nchunk, (cus + panels / 2) / panels));
  while (splits > 1 && (size_t)splits * rows > slab_rows) --splits;
  const int cps = (nchunk + splits - 1) / splits;
  splits = (nchunk + cps - 1) / cps;
  SlbArgs a = {Xrows, rows, W1packed, W2packed, cols, cps, lse, scale, tgt_off, splits > 1 ? slab : dst};
  hipLaunchKernelGGL(sl_bwd16_kernel<TR>, dim3(panels, splits), dim3(512), SLB_LDS, st, a);
  ST_LAUNCH_CHECK();
  if (splits > 1) {
    const size_t n4 = (size_t)rows * 64;
    hipLaunchKernelGGL(slab_reduce_kernel<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, slab, splits, n4, dst, n4);
    ST_LAUNCH_CHECK();
  }
  return ST_OK;
}

// ------------------------------------------------------------------------------------------------
// Forward on the same tiles (f16x3 format, Bq % 64 == 0, Bk % 256 == 0): S = panel . keys^T chunk by chunk as in
// sl_bwd16_kernel, and instead of dS an online log-sum-exp per panel row (one exponential per logit: e = exp(-|s - m|)
// rescales the sum or is the new term), taken from the previous chunk's accumulators in the shadow of the current
// chunk's MFMAs.  partial[row][split] = (max, sum exp(s - max), s[target] or 0, 0): lse_merge_kernel's input, as the
// 32x32x16 row-panel kernel with the LSE epilogue wrote it (135 us per 8192 x 8192 view there).
// ------------------------------------------------------------------------------------------------
struct SlfArgs {
  const float* X;   // [n, 256] unit rows of the queries
  int n;            // multiple of 64
  const float* w1;  // packed f16x3 image of the keys' unit rows [Ncols][256]
  int Ncols;        // multiple of 256
  int chunks_per_split;
  float scale;      // 1 / tau
  int tgt_off;      // target column of row R: R + tgt_off
  float4* partial;  // [n][gridDim.y]
};
#define SLF_LDS (64 * ST_LDA * 4 + 8 * 64 * 3 * 4)
__global__ __launch_bounds__(512, 2) void sl_fwd16_kernel(SlfArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* P1 = smem;
  float* red = smem + 64 * ST_LDA;  // [8 waves][64 rows][3]
  const int m0 = blockIdx.x * 64;
  const int nchunk = a.Ncols >> 8;
  const int c_lo = blockIdx.y * a.chunks_per_split;
  const int c_hi = min(nchunk, c_lo + a.chunks_per_split);
  if (m0 >= a.n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int wlane = (g >> 1) * 128 + 32 * (g & 1) + r16;
  auto W1p = [&](int c) { return (const f32x4*)a.w1 + (size_t)(c * 8 + wave) * 16 * 128 + wlane; };
  f32x4 rw[2][4];
  auto wfetch = [&](const f32x4* Wl, int d, f32x4 (&dst)[4]) {
    dst[0] = Wl[d * 256];
    dst[1] = Wl[d * 256 + 64];
    dst[2] = Wl[d * 256 + 16];
    dst[3] = Wl[d * 256 + 80];
  };
  float lm[4], ll[4], lt[4];
  int tgt[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    lm[rt] = -INFINITY;
    ll[rt] = 0.0f;
    lt[rt] = 0.0f;
    tgt[rt] = m0 + 16 * rt + r16 + a.tgt_off;
  }
  if (c_lo < c_hi) {
    wfetch(W1p(c_lo), 0, rw[0]);
    wfetch(W1p(c_lo), 1, rw[1]);
    int bad = 0;
    {
      f32x4 xr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = tid + i * 512;
        xr[i] = *(const f32x4*)(a.X + (size_t)(m0 + (f >> 6)) * 256 + (f & 63) * 4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = tid + i * 512;
        panel_store4_nb16(P1 + (f >> 6) * ST_LDA, f & 63, xr[i], bad);
      }
    }
    (void)bad;  // unit rows lie in [-1, 1]
    const float* A1 = P1 + r16 * ST_LDA + rp16_frag_pos(0, g);
    int cconv = c_lo;  // the chunk whose S tile is being folded into the running log-sum-exp
    auto epi_unit = [&](const f32x4& acc, int ft, int rt) {
      const int col0 = (cconv * 8 + wave) * 32 + 16 * ft + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float sv = acc[q] * a.scale;
        const float d = sv - lm[rt];
        const float e = __expf(-fabsf(d));
        ll[rt] = d > 0.0f ? fmaf(ll[rt], e, 1.0f) : ll[rt] + e;
        lm[rt] = fmaxf(lm[rt], sv);
        if (col0 + q == tgt[rt]) lt[rt] = sv;
      }
    };
    auto pass = [&](const f32x4* Wc, const f32x4* Wn, f32x4 (&acc)[2][4], const bool epi, const f32x4 (&o)[2][4]) {
      f32x4 pn[4];
      pn[0] = *(const f32x4*)(A1);
      pn[1] = *(const f32x4*)(A1 + 128);
      pn[2] = *(const f32x4*)(A1 + 16 * ST_LDA);
      pn[3] = *(const f32x4*)(A1 + 16 * ST_LDA + 128);
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int slot = d & 1;
        const h8 wh0 = as_h8(rw[slot][0]), wl0 = as_h8(rw[slot][1]), wh1 = as_h8(rw[slot][2]), wl1 = as_h8(rw[slot][3]);
        if (d + 2 < 8)
          wfetch(Wc, d + 2, rw[slot]);
        else
          wfetch(Wn, d + 2 - 8, rw[slot]);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const h8 p0h = as_h8(pn[0]), p0l = as_h8(pn[1]), p1h = as_h8(pn[2]), p1l = as_h8(pn[3]);
          if (!(d == 7 && hf == 1)) {
            const int nd = hf ? d + 1 : d, nrt = hf ? 0 : 2;
            const float* q = A1 + nrt * 16 * ST_LDA + 8 * nd;
            pn[0] = *(const f32x4*)(q);
            pn[1] = *(const f32x4*)(q + 128);
            pn[2] = *(const f32x4*)(q + 16 * ST_LDA);
            pn[3] = *(const f32x4*)(q + 16 * ST_LDA + 128);
          }
          const int rt = 2 * hf;
          acc[0][rt] = mfma16x32h(wh0, p0h, acc[0][rt]);
          acc[1][rt] = mfma16x32h(wh1, p0h, acc[1][rt]);
          acc[0][rt + 1] = mfma16x32h(wh0, p1h, acc[0][rt + 1]);
          acc[1][rt + 1] = mfma16x32h(wh1, p1h, acc[1][rt + 1]);
          acc[0][rt] = mfma16x32h(wl0, p0h, acc[0][rt]);
          acc[1][rt] = mfma16x32h(wl1, p0h, acc[1][rt]);
          acc[0][rt + 1] = mfma16x32h(wl0, p1h, acc[0][rt + 1]);
          acc[1][rt + 1] = mfma16x32h(wl1, p1h, acc[1][rt + 1]);
          acc[0][rt] = mfma16x32h(wh0, p0l, acc[0][rt]);
          acc[1][rt] = mfma16x32h(wh1, p0l, acc[1][rt]);
          acc[0][rt + 1] = mfma16x32h(wh0, p1l, acc[0][rt + 1]);
          acc[1][rt + 1] = mfma16x32h(wh1, p1l, acc[1][rt + 1]);
          if (epi && hf == 1) epi_unit(o[d & 1][d >> 1], d & 1, d >> 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    auto zero = [&](f32x4 (&h)[2][4]) {
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) h[ft][rt] = f32x4{0, 0, 0, 0};
    };
    f32x4 ah[2][4];
    zero(ah);
    __syncthreads();  // the panel is in P1
    pass(W1p(c_lo), W1p(min(c_lo + 1, c_hi - 1)), ah, false, ah);
    for (int c = c_lo; c + 1 < c_hi; ++c) {
      f32x4 an[2][4];
      zero(an);
      cconv = c;
      pass(W1p(c + 1), W1p(min(c + 2, c_hi - 1)), an, true, ah);  // S of chunk c + 1; chunk c folds into the sums in the MFMAs' shadow
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) ah[ft][rt] = an[ft][rt];
    }
    cconv = c_hi - 1;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) epi_unit(ah[ft][rt], ft, rt);
  }
  // rows of this lane: 16 rt + r16; the row's four lanes (g = 0 .. 3), then the eight waves through LDS, in index order
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    Lse t{lm[rt], ll[rt]};
    float tg = lt[rt];
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      Lse u;
      u.m = __shfl_xor(t.m, o, 64);
      u.l = __shfl_xor(t.l, o, 64);
      t = lse_merge(t, u);
      tg += __shfl_xor(tg, o, 64);
    }
    if (g == 0) {
      float* p = red + (wave * 64 + 16 * rt + r16) * 3;
      p[0] = t.m;
      p[1] = t.l;
      p[2] = tg;
    }
  }
  __syncthreads();
  if (tid < 64) {
    Lse t{red[tid * 3], red[tid * 3 + 1]};
    float tg = red[tid * 3 + 2];
    for (int w = 1; w < 8; ++w) {
      const float* p = red + (w * 64 + tid) * 3;
      t = lse_merge(t, Lse{p[0], p[1]});
      tg += p[2];
    }
    a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m, t.l, tg, 0.0f);
  }
}
int g_sl_fwd16 = 1;  // 0: the 32x32x16 row-panel kernel with the LSE epilogue (tests flip it)

static inline size_t sl_align(size_t v) { return (v + 255) / 256 * 256; }

struct SlWs {
  float *ou, *ku, *kup, *kut, *kutp, *on, *kn, *dou, *dku, *ds, *slab, *oup, *oust, *oustp, *rs;
  float4* partial;
  size_t total;
};
static SlWs sl_carve(int Bq, int Bk, char* base) {
  SlWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += sl_align(bytes);
    return p;
  };
  w.ou = (float*)take((size_t)Bq * 1024);
  w.ku = (float*)take((size_t)Bk * 1024);
  w.kup = (float*)take(rp_packed_floats(Bk, 256) * 4);
  w.kut = (float*)take((size_t)Bk * 1024);
  w.kutp = (float*)take((size_t)Bk * 1024);
  w.slab = (float*)take((size_t)SL_KSPLIT * SL_CHUNK * 1024);
  w.on = (float*)take((size_t)Bq * 4);
  w.kn = (float*)take((size_t)Bk * 4);
  w.dou = (float*)take((size_t)Bq * 1024);
  w.dku = (float*)take((size_t)Bk * 1024);
  w.partial = (float4*)take((size_t)Bq * SL_SPLITS * 16);
  const size_t big = Bq > Bk ? Bq : Bk;
  w.ds = (float*)take((size_t)SL_CHUNK * big * 4);
  w.oup = (float*)take(rp_packed_floats(Bq, 256) * 4);
  w.oust = (float*)take((size_t)Bq * 1024);
  w.oustp = (float*)take((size_t)Bq * 1024);
  w.rs = (float*)take((size_t)Bq * 4);
  w.total = off;
  return w;
}

extern "C" size_t st_simloss_scratch_bytes(int Bq, int Bk) {
  if (Bq <= 0 || Bk <= 0) return 0;
  return sl_carve(Bq, Bk, nullptr).total;
}

static int sl_check(const float* ori, const float* like, int Bq, int Bk, float tau, int row_offset, void* scratch,
                    size_t scratch_bytes) {
  ST_CHECK_ARG(Bq > 0 && Bk > 0, "Bq=%d Bk=%d", Bq, Bk);
  ST_CHECK_ARG(ori && like && scratch, "NULL buffer");
  ST_CHECK_ARG(tau > 0.0f, "tau=%g", (double)tau);
  ST_CHECK_ARG(row_offset >= 0 && row_offset + Bq <= Bk, "targets row_offset+[0,Bq) must lie inside [0,Bk): row_offset=%d", row_offset);
  const size_t need = st_simloss_scratch_bytes(Bq, Bk);
  if (need > scratch_bytes) {
    st_set_error("simloss scratch too small: need %zu bytes, have %zu", need, scratch_bytes);
    return ST_ENOMEM;
  }
  return ST_OK;
}

extern "C" int st_simloss_fwd(const float* ori, const float* like, int Bq, int Bk, float tau, int row_offset,
                              float* loss, float* lse, void* scratch, size_t scratch_bytes, void* stream) {
  ST_TRY(sl_check(ori, like, Bq, Bk, tau, row_offset, scratch, scratch_bytes));
  ST_CHECK_ARG(loss && lse, "NULL output");
  hipStream_t st = (hipStream_t)stream;
  if (g_sl_small && Bq <= SL_SMALL && Bk <= SL_SMALL) {
    hipLaunchKernelGGL(sl_small_fwd_kernel, dim3((Bq + 3) / 4), dim3(256), 0, st, ori, like, Bq, Bk, 1.0f / tau, row_offset, loss, lse, 0L, 0);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  SlWs w = sl_carve(Bq, Bk, (char*)scratch);
  hipLaunchKernelGGL(unit_rows_kernel, dim3((Bq + 3) / 4), dim3(256), 0, st, ori, Bq, w.ou, w.on, (float*)nullptr);
  hipLaunchKernelGGL(unit_rows_kernel, dim3((Bk + 3) / 4), dim3(256), 0, st, like, Bk, w.ku, w.kn, (float*)nullptr);
  ST_LAUNCH_CHECK();
  ST_TRY(sl_pack(w.ku, Bk, 256, w.kup, st));
  if (g_sl_fwd16 && g_sl_format == ST_GEMM_F16X3 && Bq % 64 == 0 && Bk % 256 == 0) {
    ST_ENSURE_LDS(sl_fwd16_kernel, SLF_LDS);
    const int panels = Bq / 64, nchunk = Bk / 256;
    // column splits: about two workgroups per CU, whole chunks, at most SL_SPLITS (the size of `partial`)
    int splits = std::max(1, std::min(std::min(nchunk, SL_SPLITS), (2 * st_cu_count() + panels / 2) / panels));
    const int cps = (nchunk + splits - 1) / splits;
    splits = (nchunk + cps - 1) / cps;
    SlfArgs f = {w.ou, Bq, w.kup, Bk, cps, 1.0f / tau, row_offset, w.partial};
    hipLaunchKernelGGL(sl_fwd16_kernel, dim3(panels, splits), dim3(512), SLF_LDS, st, f);
    hipLaunchKernelGGL(lse_merge_kernel, dim3((Bq + 255) / 256), dim3(256), 0, st, w.partial, splits, Bq, loss, lse);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  GemmArgs g = {};
  g.fmt = g_sl_format;
  g.X = w.ou; g.n_host = Bq; g.W = w.kup; g.N = Bk; g.scale = 1.0f / tau; g.row_offset = row_offset; g.partial = w.partial;
  ST_TRY(launch_rp<EPI_LSE>(g, Bq, SL_SPLITS, st));
  hipLaunchKernelGGL(lse_merge_kernel, dim3((Bq + 255) / 256), dim3(256), 0, st, w.partial, SL_SPLITS, Bq, loss, lse);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

extern "C" int st_simloss_bwd(const float* ori, const float* like, int Bq, int Bk, float tau, int row_offset,
                              const float* lse, const float* g, float* d_ori, float* d_like, void* scratch,
                              size_t scratch_bytes, void* stream) {
  ST_TRY(sl_check(ori, like, Bq, Bk, tau, row_offset, scratch, scratch_bytes));
  ST_CHECK_ARG(lse && g && d_ori && d_like, "NULL buffer");
  ST_CHECK_ARG(Bq % 32 == 0 && Bk % 32 == 0, "backward needs Bq and Bk to be multiples of 32 (Bq=%d Bk=%d)", Bq, Bk);
  hipStream_t st = (hipStream_t)stream;
  if (g_sl_small && Bq <= SL_SMALL && Bk <= SL_SMALL) {
    hipLaunchKernelGGL(sl_small_bwd_kernel, dim3((Bq + 3) / 4 + (Bk + 3) / 4), dim3(256), 0, st, ori, like, Bq, Bk, 1.0f / tau, row_offset,
                       lse, g, d_ori, d_like, 0L, 0, 0L);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  SlWs w = sl_carve(Bq, Bk, (char*)scratch);
  hipLaunchKernelGGL(unit_rows_kernel, dim3((Bq + 3) / 4), dim3(256), 0, st, ori, Bq, w.ou, w.on, (float*)nullptr);
  hipLaunchKernelGGL(unit_rows_kernel, dim3((Bk + 3) / 4), dim3(256), 0, st, like, Bk, w.ku, w.kn, w.kut);
  ST_LAUNCH_CHECK();
  if (g_sl_format == ST_GEMM_F16X3 && Bq % 256 == 0 && Bk % 256 == 0) {
    // Every product on the row-panel kernels in the split format.  The gradient weights g / tau leave the matrices:
    // dS' = exp(s - lse) - [target] in [-1, 1] is what gets split into f16 hi + lo (g * dS ~ 1e-8 would sit below the
    // f16 range); d_ou rows are scaled afterwards, d_ku contracts against ou rows that carry the scale.
    //   d_ou = (dS' . ku) * s_i          dS' [chunk of queries, Bk]  (EPI_DS)  then K = Bk chunked GEMM
    //   d_ku = dS'^T . (s_i ou_i)        dS'^T [chunk of keys, Bq]   (EPI_DST: the transposed block straight from ku . ou^T)
    const float scale = 1.0f / tau;
    hipLaunchKernelGGL(scale_transpose_kernel, dim3((Bq + 3) / 4), dim3(256), 0, st, w.ou, g, scale, Bq, w.oust, w.rs);
    ST_LAUNCH_CHECK();
    ST_TRY(sl_pack(w.ku, Bk, 256, w.kup, st));
    ST_TRY(sl_pack(w.ou, Bq, 256, w.oup, st));
    ST_TRY(sl_pack(w.kut, 256, Bk, w.kutp, st));
    ST_TRY(sl_pack(w.oust, 256, Bq, w.oustp, st));
    auto chunked = [&](const float* Xrows, int n_rows, const float* Wp, int Ncols, int epi_t, const float* Wt_p, float* dst,
                       int row_off0) -> int {
      for (int r0 = 0; r0 < n_rows; r0 += SL_CHUNK) {
        const int rows = (n_rows - r0 < SL_CHUNK) ? n_rows - r0 : SL_CHUNK;
        GemmArgs a = {};
        a.fmt = ST_GEMM_F16X3;
        a.X = Xrows + (size_t)r0 * 256; a.n_host = rows; a.W = Wp; a.N = Ncols; a.scale = scale; a.out = w.ds; a.ldo = Ncols;
        a.lse = epi_t ? lse : lse + r0;
        // DS: target column of row r is row_offset + r0 + r;  DST: row (key) r0 + r is the target of query column c iff r0 + r == row_offset + c
        a.row_offset = epi_t ? row_offset - r0 : row_offset + r0;
        // column splits: enough workgroups to fill the chip (a chunk is only rows / 64 panels), at least 2 blocks per wave
        const int panels = (rows + 63) / 64, nblk = (Ncols + 31) / 32;
        int sp = (Ncols + 1023) / 1024;
        while (sp * panels < 512 && sp * 2 * 8 <= nblk) sp *= 2;
        if (epi_t)
          ST_TRY(launch_rp<EPI_DST>(a, rows, sp, st));
        else
          ST_TRY(launch_rp<EPI_DS>(a, rows, sp, st));
        GemmArgs b = {};
        b.fmt = ST_GEMM_F16X3;
        b.X = w.ds; b.n_host = rows; b.K = Ncols; b.W = Wt_p; b.N = 256; b.ldo = 256;
        const int nchunk = Ncols >> 8;
        int ks = SL_KSPLIT < nchunk ? SL_KSPLIT : nchunk;
        while (ks > 1 && ((rows + 63) / 64) * ks > 1024) ks >>= 1;
        b.out = ks > 1 ? w.slab : dst + (size_t)r0 * 256;
        ST_TRY(launch_rp_kchunk<EPI_BIAS>(b, rows, st, ks));
        if (ks > 1) {
          const size_t n4 = (size_t)rows * 64;
          hipLaunchKernelGGL(slab_reduce_kernel<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.slab, ks, n4,
                             dst + (size_t)r0 * 256, n4);
          ST_LAUNCH_CHECK();
        }
      }
      return ST_OK;
    };
    if (g_sl_fused) {
      const size_t slab_rows = (size_t)SL_KSPLIT * SL_CHUNK;
      ST_TRY(sl_bwd_fused<false>(w.ou, Bq, w.kup, w.kutp, Bk, lse, scale, row_offset, w.slab, slab_rows, w.dou, st));
      ST_TRY(sl_bwd_fused<true>(w.ku, Bk, w.oup, w.oustp, Bq, lse, scale, -row_offset, w.slab, slab_rows, w.dku, st));
    } else {
      ST_TRY(chunked(w.ou, Bq, w.kup, Bk, 0, w.kutp, w.dou, 0));
      ST_TRY(chunked(w.ku, Bk, w.oup, Bq, 1, w.oustp, w.dku, 0));
    }
    hipLaunchKernelGGL(unit_rows_bwd_kernel, dim3((Bq + 3) / 4), dim3(256), 0, st, w.dou, w.ou, w.on, Bq, d_ori, 0, (const float*)w.rs);
    hipLaunchKernelGGL(unit_rows_bwd_kernel, dim3((Bk + 3) / 4), dim3(256), 0, st, w.dku, w.ku, w.kn, Bk, d_like, 1, (const float*)nullptr);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  // exact-f32 form (ST_GEMM_F32, or sizes that are not multiples of 256): f32 row-panel kernels for dS and d_ou, the
  // generic transposed-operand f32 kernel for d_ku
  hipLaunchKernelGGL(pack_weight_kernel<0>, dim3((unsigned)((rp_packed_floats(Bk, 256) / 4 + 255) / 256)), dim3(256), 0, st, w.ku, Bk, 256, w.kup);
  ST_LAUNCH_CHECK();
  const bool splitk = (Bk % 256 == 0);   // d_ou = dS . ku on the row-panel kernel, K = Bk split over workgroups
  if (splitk) {
    hipLaunchKernelGGL(pack_weight_kernel<0>, dim3((unsigned)(((size_t)8 * (Bk >> 3) * 64 + 255) / 256)), dim3(256), 0, st, w.kut, 256, Bk, w.kutp);
    ST_LAUNCH_CHECK();
  }
  ST_HIP(hipMemsetAsync(w.dku, 0, (size_t)Bk * 1024, st));
  for (int r0 = 0; r0 < Bq; r0 += SL_CHUNK) {
    const int rows = (Bq - r0 < SL_CHUNK) ? Bq - r0 : SL_CHUNK;
    // dS chunk [rows, Bk]
    GemmArgs a = {};
    a.X = w.ou + (size_t)r0 * 256; a.n_host = rows; a.W = w.kup; a.N = Bk; a.scale = 1.0f / tau;
    a.row_offset = row_offset + r0; a.lse = lse + r0; a.g = g + r0; a.out = w.ds; a.ldo = Bk;
    ST_TRY(launch_rp<EPI_DS>(a, rows, (Bk + 1023) / 1024, st));
    // d_ou[chunk] = dS . ku   (W = ku^T [256, Bk])
    GemmArgs b = {};
    b.X = w.ds; b.n_host = rows; b.K = Bk; b.W = w.kut; b.N = 256; b.out = w.dou + (size_t)r0 * 256; b.ldo = 256;
    if (splitk) {
      const int nchunk = Bk >> 8;
      int ks = SL_KSPLIT < nchunk ? SL_KSPLIT : nchunk;
      while (ks > 1 && ((rows + 63) / 64) * ks > 1024) ks >>= 1;   // no more workgroups than ~2 rounds
      b.W = w.kutp;
      b.out = ks > 1 ? w.slab : w.dou + (size_t)r0 * 256;
      ST_TRY(launch_rp_kchunk<EPI_BIAS>(b, rows, st, ks));
      if (ks > 1) {
        const size_t n4 = (size_t)rows * 64;
        hipLaunchKernelGGL(slab_reduce_kernel<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.slab, ks, n4,
                           w.dou + (size_t)r0 * 256, n4);
        ST_LAUNCH_CHECK();
      }
    } else {
      ST_TRY(launch_kloop<EPI_BIAS>(b, rows, st));
    }
    // d_ku += dS^T . ou[chunk]   (X given transposed: dS [K=rows, n=Bk]; W given transposed: ou chunk [K=rows, 256])
    GemmArgs c = {};
    c.X = w.ds; c.ldx = Bk; c.n_host = Bk; c.K = rows; c.W = w.ou + (size_t)r0 * 256; c.ldw = 256; c.N = 256;
    c.out = w.dku; c.ldo = 256; c.resid = w.dku;
    ST_TRY((launch_kloop<EPI_RESID, true, true>(c, Bk, st)));
  }
  hipLaunchKernelGGL(unit_rows_bwd_kernel, dim3((Bq + 3) / 4), dim3(256), 0, st, w.dou, w.ou, w.on, Bq, d_ori, 0, (const float*)nullptr);
  hipLaunchKernelGGL(unit_rows_bwd_kernel, dim3((Bk + 3) / 4), dim3(256), 0, st, w.dku, w.ku, w.kn, Bk, d_like, 1, (const float*)nullptr);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
