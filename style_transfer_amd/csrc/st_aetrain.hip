// One optimisation step of the autoencoder's constraint training (second half): loss and
// gradients of the loop body of train_ae_model_constraint (/root/reference/method/yelp/siamese_contrastive.py:1028-1072):
//
//   ori_latent, out = ae_model.forward(ori)                     model.py:229-250, TRAIN mode (dropout 0.1 everywhere)
//   loss_rec = LabelSmoothing(out, tgt_y) / ntokens             model.py:389-411, :1040-1041
//   like_latent = 10 `like` views + a second pass of the origins (:1049-1056)
//   constraint = mean over the 11 views of NLL(log_softmax(cos(ori_i, like_j) / tau), i)     :1060-1071
//   (bata * constraint + loss_rec).backward()                   :1072
//
// MI355X formulation: everything is LENGTH-PACKED (no padding rows): the 12 encoder passes of the reference (origins,
// 10 views, origins again; each B separate batch-1..B calls) are ONE packed batch of 12 B sentences, so every encoder
// weight gradient is one contraction over all rows; the reference's teacher-forced decodes of the views (model.py:248-249,
// outputs unused) are not computed.  The decoder's source attention sees ONE memory position (the latent), so its softmax
// is 1: the query / key projections and sublayer[1].norm receive exactly zero gradient.  f32 throughout; matrix products
// on v_mfma_f32_32x32x2_f32 (64x64 LDS tiles, arbitrary operand strides: one kernel serves Y = X W^T, dX = dY W and
// dW = dY^T X).  The B x B similarity blocks are st_simloss_fwd / st_simloss_bwd.  Dropout keep decisions are a counter
// hash of (seed, site, sentence, position, feature): regenerated in the backward pass, never stored, and reproducible
// on the host (tests replay them through the CPU restatement).
#include <math.h>

#include <algorithm>
#include "st_rowpanel.h"

#ifndef ST_TRY
#define ST_TRY(x)            \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != ST_OK) return rc_; \
  } while (0)
#endif

// Kernel-form switches of the step (ints in the library so that tests can flip them: tests/test_gpu_train.py; the
// environment variables of INTEGRATION.md section D set their start values, read once).  Every form computes the same
// function; summation orders differ.
int g_ae_fwd_h = 1;       // ST_AE_F32_FWD -> 0: forward products in exact f32
int g_ae_force_redo = 0;  // tests: the range flag starts at 1, so every split-f16 forward product is redone in exact f32
int g_ae_rp = 1;          // ST_AE_NO_RP -> 0: strided GEMM for forward and dX
int g_ae_dw = 2;          // ST_AE_NO_DW -> 0: strided GEMM + column sums; ST_AE_DW_DIRECT -> 1: operands straight from global memory; 2: through LDS
int g_ae_att16 = 1;       // ST_AE_NO_ATT16 -> 0: lane-per-row attention kernels for every sentence
int g_ae_views = 1;       // ST_AE_VIEW_LOOP -> 0: one pair of in-batch-loss launches per view (small batches too)
static void ae_env() {
  static bool done = false;
  if (done) return;
  done = true;
  if (getenv("ST_AE_F32_FWD")) g_ae_fwd_h = 0;
  if (getenv("ST_AE_NO_RP")) g_ae_rp = 0;
  if (getenv("ST_AE_DW_DIRECT")) g_ae_dw = 1;
  if (getenv("ST_AE_NO_DW")) g_ae_dw = 0;
  if (getenv("ST_AE_NO_ATT16")) g_ae_att16 = 0;
  if (getenv("ST_AE_VIEW_LOOP")) g_ae_views = 0;
}

namespace {

// ------------------------------------------------------------------------------------------------
// dropout: keep(seed, site, idx) with P(keep) = 1 - p
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t ae_mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
struct Drop {
  uint64_t seed;
  float p, scale;  // scale = 1 / (1 - p); p == 0: identity
  uint32_t site;
};
__device__ __forceinline__ float drop_factor(const Drop& d, uint64_t idx) {
  if (d.p <= 0.0f) return 1.0f;
  const uint64_t h = ae_mix(ae_mix(d.seed ^ ((uint64_t)d.site << 40)) ^ idx);
  const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
  return u >= d.p ? d.scale : 0.0f;
}

// sum over nz slices of n values each, slice z at src + z * stride: 64 values per workgroup, the slices in four contiguous
// groups (one per wave) added in index order, the four group sums added in index order: a fixed tree
__device__ __forceinline__ float slices_sum64(const float* __restrict__ src, size_t stride, int nz, size_t e, bool ok, float (&red)[4][64]) {
  const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int per = (nz + 3) / 4, z0 = g * per, z1 = min(nz, z0 + per);
  float s = 0.0f;
  if (ok) {
#pragma unroll 8
    for (int z = z0; z < z1; ++z) s += src[(size_t)z * stride + e];
  }
  red[g][l] = s;
  __syncthreads();
  return (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
// ------------------------------------------------------------------------------------------------
// C[i,j] (+)= bias[j] + sum_r A(i,r) B(r,j),  A(i,r) = A[i*sai + r*sar],  B(r,j) = B[r*sbr + j*sbj]
// ------------------------------------------------------------------------------------------------
struct GArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, R;
  long sai, sar, sbr, sbj;
  int ldc, accumulate;
};
#define AG_LDW 36  /* operand row stride in LDS: 32 contraction steps + 4 (conflict-free 16-byte fragment reads) */
// 128 x 128 tile per workgroup, 64 x 64 per wave (2 x 2 exact-f32 MFMA tiles), contraction in steps of 32 through LDS
// with the next step's operands prefetched into registers.  Operand strides are arguments (one kernel serves
// Y = X W^T, dX = dY W and dW = dY^T X); loads are scalar and coalesced along whichever index is contiguous.
// gridDim.z > 1: split over the contraction index, slice z writes its partial product to slab + z * M * N (no bias, no
// accumulate) and ae_slab_sum_kernel adds the slices in a fixed order (weight gradients contract over ~10^4 rows into
// a few hundred output tiles: without the split a handful of CUs would run the whole product).
__global__ __launch_bounds__(256) void ae_gemm_kernel(GArgs a, float* __restrict__ slab, int r_per_slice) {
  __shared__ __align__(16) float As[128 * AG_LDW], Bs[128 * AG_LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
  const int r_lo = blockIdx.z * r_per_slice, r_hi = min(a.R, r_lo + r_per_slice);
  const bool a_r_fast = a.sar == 1, b_r_fast = a.sbr == 1;
  float ra[16], rb[16];
  auto fetch = [&](int rc) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = tid + 256 * q;
      const int ar = a_r_fast ? (e & 31) : (e >> 7), ai = a_r_fast ? (e >> 5) : (e & 127);
      const int br = b_r_fast ? (e & 31) : (e >> 7), bj = b_r_fast ? (e >> 5) : (e & 127);
      ra[q] = (i0 + ai < a.M && rc + ar < r_hi) ? a.A[(long)(i0 + ai) * a.sai + (long)(rc + ar) * a.sar] : 0.0f;
      rb[q] = (j0 + bj < a.N && rc + br < r_hi) ? a.B[(long)(rc + br) * a.sbr + (long)(j0 + bj) * a.sbj] : 0.0f;
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
  if (r_lo < r_hi) fetch(r_lo);
  for (int rc = r_lo; rc < r_hi; rc += 32) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = tid + 256 * q;
      const int ar = a_r_fast ? (e & 31) : (e >> 7), ai = a_r_fast ? (e >> 5) : (e & 127);
      const int br = b_r_fast ? (e & 31) : (e >> 7), bj = b_r_fast ? (e >> 5) : (e & 127);
      As[ai * AG_LDW + ar] = ra[q];
      Bs[bj * AG_LDW + br] = rb[q];
    }
    __syncthreads();
    if (rc + 32 < r_hi) fetch(rc + 32);
    const float* Ab = As + (wr * 64 + l31) * AG_LDW + 4 * half;
    const float* Bb = Bs + (wc * 64 + l31) * AG_LDW + 4 * half;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const f32x4 a0 = *(const f32x4*)(Ab + kg * 8), a1 = *(const f32x4*)(Ab + 32 * AG_LDW + kg * 8);
      const f32x4 b0 = *(const f32x4*)(Bb + kg * 8), b1 = *(const f32x4*)(Bb + 32 * AG_LDW + kg * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0][0] = mfma32(a0[q], b0[q], acc[0][0]);
        acc[0][1] = mfma32(a0[q], b1[q], acc[0][1]);
        acc[1][0] = mfma32(a1[q], b0[q], acc[1][0]);
        acc[1][1] = mfma32(a1[q], b1[q], acc[1][1]);
      }
    }
  }
  const bool sliced = gridDim.z > 1;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int j = j0 + wc * 64 + ni * 32 + l31;
      if (j >= a.N) continue;
      const float bj = (a.bias && !sliced) ? a.bias[j] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wr * 64 + mi * 32 + mfma32_row(r, lane);
        if (i >= a.M) continue;
        if (sliced) {
          slab[((size_t)blockIdx.z * a.M + i) * a.N + j] = acc[mi][ni][r];
        } else {
          float* c = a.C + (long)i * a.ldc + j;
          *c = (a.accumulate ? *c : 0.0f) + acc[mi][ni][r] + bj;
        }
      }
    }
}
// C[i][j] = (accumulate ? C : 0) + bias[j] + sum_z slab[z][i][j], slices added in index order
__global__ __launch_bounds__(256) void ae_slab_sum_kernel(GArgs a, const float* __restrict__ slab, int nz) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)a.M * a.N) return;
  const int i = (int)(e / a.N), j = (int)(e % a.N);
  float s = 0.0f;
  for (int z = 0; z < nz; ++z) s += slab[(size_t)z * a.M * a.N + e];
  float* c = a.C + (long)i * a.ldc + j;
  *c = (a.accumulate ? *c : 0.0f) + s + (a.bias ? a.bias[j] : 0.0f);
}

// ------------------------------------------------------------------------------------------------
// Weight gradients: C[i][j] = sum_r A[r][i] B[r][j] (A = dY [R, M], B = X [R, N], both row-major) and, from the same
// loads, db[i] = sum_r A[r][i].  Both operands are contiguous along the NON-contracted index, which is exactly how
// v_mfma_f32_32x32x2_f32 wants them (lane = output index, lane half = contraction index): every operand register is one
// coalesced dword load (two rows of 32 consecutive floats per wave), no LDS, no barrier.  A wave owns a 64 x 64 tile
// (4 accumulators), a workgroup 128 x 128 (the two waves sharing an operand meet in the vector L1); operands of the next
// 8 rows are in flight under the 16 MFMAs of the current 8.  The contraction is cut into nz slices of whole 8-row
// batches, one workgroup per (tile, slice), the tiles of a slice on ONE XCD (they read the same rows: its L2 serves the
// second reader); partial products go to slab[z][i][j] and ae_dw_sum_kernel adds them in slice order (fixed order:
// results do not depend on scheduling).  Out-of-range output columns read a clamped address: their garbage only
// reaches accumulators that are never stored.  (The strided LDS kernel above ran these products at ~ 32 TFLOP/s:
// transposed operand staging with 8-way bank conflicts, two barriers per 32 rows.)
// ------------------------------------------------------------------------------------------------
struct DwArgs {
  const float* A;
  const float* B;
  float* slab;    // [nz][M][N]
  float* dbslab;  // [nz][M] or null
  int M, N, R, lda, ldb;
  int nz, per, tiles_n, tiles;
};
#ifndef DW_U
#define DW_U 4  /* contraction pairs per batch: 8 rows (8: 214 us per launch at 3 waves per SIMD, 6: 173, 4: 152 at 4 waves) */
#endif
#ifdef DW_WITH_FENCE
#define DW_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define DW_FENCE()  /* measured: pinning the whole batch of loads in front of the MFMAs is slower (303 against 213 us per launch) */
#endif
// VEC 2 (M, N even): lane l of a half holds outputs 2 l and 2 l + 1 of its 64 and loads them as one dwordx2 (half the
// load instructions: 190 against 209 us per launch); VEC 1: outputs l and l + 32, any M, N (the generator's odd vocabulary)
template <int VEC>
__global__ __launch_bounds__(256, 4) void ae_dw_kernel(DwArgs a) {
  const int nb = gridDim.x;  // a multiple of 8
  const int vid = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  if (vid >= a.tiles * a.nz) return;
  const int tile = vid % a.tiles, z = vid / a.tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int i0 = (tile / a.tiles_n) * 128 + (wave >> 1) * 64, j0 = (tile % a.tiles_n) * 128 + (wave & 1) * 64;
  const int r_lo = z * a.per, r_hi = min(a.R, r_lo + a.per);
  const bool want_db = a.dbslab && (tile % a.tiles_n) == 0 && (wave & 1) == 0;
  f32x16 acc[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][p][r] = 0.0f;
  float sa0 = 0.0f, sa1 = 0.0f;
  const long sa = 2L * a.lda, sb = 2L * a.ldb;
  const int ca = VEC == 2 ? 2 * l31 : l31, cs = VEC == 2 ? 1 : 32;  // the lane's two outputs of the wave's 64: ca, ca + cs
  const float* pa0 = a.A + (long)(r_lo + half) * a.lda + min(i0 + ca, a.M - 1 - (VEC == 2));
  const float* pb0 = a.B + (long)(r_lo + half) * a.ldb + min(j0 + ca, a.N - 1 - (VEC == 2));
  const float* pa1 = VEC == 2 ? pa0 + 1 : a.A + (long)(r_lo + half) * a.lda + min(i0 + 32 + l31, a.M - 1);
  const float* pb1 = VEC == 2 ? pb0 + 1 : a.B + (long)(r_lo + half) * a.ldb + min(j0 + 32 + l31, a.N - 1);
  float ca0[DW_U], ca1[DW_U], cb0[DW_U], cb1[DW_U], na0[DW_U], na1[DW_U], nb0[DW_U], nb1[DW_U];
  const int n_full = r_hi > r_lo ? (r_hi - r_lo) / (2 * DW_U) : 0;
  auto fetch = [&](float (&x0)[DW_U], float (&x1)[DW_U], float (&y0)[DW_U], float (&y1)[DW_U]) {
#ifdef DW_ABLATE_LOADS
#pragma unroll
    for (int u = 0; u < DW_U; ++u) { x0[u] = sa0 + u; x1[u] = sa1; y0[u] = 1.0f; y1[u] = 2.0f; }
    return;
#endif
#pragma unroll
    for (int u = 0; u < DW_U; ++u) {
      if constexpr (VEC == 2) {
        const float2 xa = *(const float2*)(pa0 + u * sa), xb = *(const float2*)(pb0 + u * sb);
        x0[u] = xa.x; x1[u] = xa.y; y0[u] = xb.x; y1[u] = xb.y;
      } else {
        x0[u] = pa0[u * sa];
        x1[u] = pa1[u * sa];
        y0[u] = pb0[u * sb];
        y1[u] = pb1[u * sb];
      }
    }
    pa0 += DW_U * sa; pa1 += DW_U * sa; pb0 += DW_U * sb; pb1 += DW_U * sb;
  };
  auto mac = [&](const float (&x0)[DW_U], const float (&x1)[DW_U], const float (&y0)[DW_U], const float (&y1)[DW_U]) {
#ifdef DW_ABLATE_MFMA
#pragma unroll
    for (int u = 0; u < DW_U; ++u) { sa0 += x0[u] * y0[u]; sa1 += x1[u] * y1[u]; }
    return;
#endif
#pragma unroll
    for (int u = 0; u < DW_U; ++u) {
      acc[0][0] = mfma32(x0[u], y0[u], acc[0][0]);
      acc[0][1] = mfma32(x0[u], y1[u], acc[0][1]);
      acc[1][0] = mfma32(x1[u], y0[u], acc[1][0]);
      acc[1][1] = mfma32(x1[u], y1[u], acc[1][1]);
    }
    if (want_db) {
#pragma unroll
      for (int u = 0; u < DW_U; ++u) {
        sa0 += x0[u];
        sa1 += x1[u];
      }
    }
  };
  // two register sets in turn, no conditional inside the loop: the wait in front of a batch's MFMAs then counts exactly
  // the 32 younger loads (a copy "current = next" makes the compiler wait for the loads it has just issued)
  int b = 0;
  if (n_full > 0) {
    fetch(ca0, ca1, cb0, cb1);
    for (; b + 2 < n_full; b += 2) {
      fetch(na0, na1, nb0, nb1);
      DW_FENCE();
      mac(ca0, ca1, cb0, cb1);
      DW_FENCE();
      fetch(ca0, ca1, cb0, cb1);
      DW_FENCE();
      mac(na0, na1, nb0, nb1);
      DW_FENCE();
    }
    mac(ca0, ca1, cb0, cb1);
    ++b;
  }
  // what is left (at most one whole batch and a partial one), one batch at a time; rows past the end contribute zeros
  for (int r0 = r_lo + b * 2 * DW_U; r0 < r_hi; r0 += 2 * DW_U) {
#pragma unroll
    for (int u = 0; u < DW_U; ++u) {
      const int r = r0 + 2 * u + half;
      const bool ok = r < r_hi;
      const long ra = (long)(ok ? r : r_hi - 1) - (r0 + half);  // row offset from the pointers' row
      const float x0 = pa0[ra * a.lda], x1 = pa1[ra * a.lda], y0 = pb0[ra * a.ldb], y1 = pb1[ra * a.ldb];
      ca0[u] = ok ? x0 : 0.0f; ca1[u] = ok ? x1 : 0.0f; cb0[u] = ok ? y0 : 0.0f; cb1[u] = ok ? y1 : 0.0f;
    }
    pa0 += DW_U * sa; pa1 += DW_U * sa; pb0 += DW_U * sb; pb1 += DW_U * sb;
    mac(ca0, ca1, cb0, cb1);
  }
  float* out = a.slab + (size_t)z * a.M * a.N;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int j = j0 + ca + cs * p;
      if (j >= a.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + (VEC == 2 ? 2 * mfma32_row(r, lane) + q : 32 * q + mfma32_row(r, lane));
        if (i < a.M) out[(size_t)i * a.N + j] = acc[q][p][r];
      }
    }
#ifdef DW_ABLATE_MFMA
  acc[0][0][0] = sa0 + sa1;
#endif
  if (want_db) {  // lane half h holds the rows of parity h: even rows + odd rows
    sa0 += __shfl_down(sa0, 32);
    sa1 += __shfl_down(sa1, 32);
    if (half == 0) {
      if (i0 + ca < a.M) a.dbslab[(size_t)z * a.M + i0 + ca] = sa0;
      if (i0 + ca + cs < a.M) a.dbslab[(size_t)z * a.M + i0 + ca + cs] = sa1;
    }
  }
}
// The same product with the workgroup's operands SHARED through LDS (M, N multiples of 128): the direct form above has
// each of the four waves load its own 64 + 64 columns, so every operand element is requested twice per workgroup, and
// measured memory-bound on exactly that (ablations at 256 x 256 x 108 k: MFMAs alone 108 us, loads alone 79 us, together
// 152 - 214 us).  Here 256 threads load a 16-row chunk of the 128 + 128 columns once (four coalesced 16-byte loads per
// thread, in flight under the MFMAs of the previous chunk), park it in one of two LDS buffers, and every wave reads its
// operand pairs from there (ds_read_b64, lane l of a half = outputs 2 l, 2 l + 1: 256 contiguous bytes per half-wave,
// conflict-free without padding).  One barrier per chunk.
#define DWL_ROWS 16
#ifndef DW_MIN_SLICE_ROWS
#define DW_MIN_SLICE_ROWS 64  /* at 13 k rows (B = 128): 64 -> 5.39 ms per step, 128 -> 5.39, 256 -> 5.59, 512 -> 6.19: workgroups count for more than the slab's bytes */
#endif
__global__ __launch_bounds__(256, 4) void ae_dw_lds_kernel(DwArgs a) {
  __shared__ __align__(16) float sA[2][DWL_ROWS][128], sB[2][DWL_ROWS][128];
  const int nb = gridDim.x;  // a multiple of 8
  const int vid = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  if (vid >= a.tiles * a.nz) return;
  const int tile = vid % a.tiles, z = vid / a.tiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int wi = wave >> 1, wc = wave & 1;
  const int i0 = (tile / a.tiles_n) * 128, j0 = (tile % a.tiles_n) * 128;
  const int r_lo = z * a.per, r_hi = min(a.R, r_lo + a.per);
  const bool want_db = a.dbslab && (tile % a.tiles_n) == 0 && wc == 0;
  f32x16 acc[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][p][r] = 0.0f;
  float sa0 = 0.0f, sa1 = 0.0f;
  // staging: float4 f = tid + 256 k of a chunk's [16][32] float4s, k = 0, 1
  const int srow = tid >> 5, scol = 4 * (tid & 31);
  const float* ga = a.A + (size_t)i0 + scol;
  const float* gb = a.B + (size_t)j0 + scol;
  f32x4 ra[2], rb[2];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = r0 + srow + 8 * k;
      ra[k] = rb[k] = f32x4{0, 0, 0, 0};
      if (r < r_hi) {
        ra[k] = *(const f32x4*)(ga + (size_t)r * a.lda);
        rb[k] = *(const f32x4*)(gb + (size_t)r * a.ldb);
      }
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      *(f32x4*)&sA[buf][srow + 8 * k][scol] = ra[k];
      *(f32x4*)&sB[buf][srow + 8 * k][scol] = rb[k];
    }
  };
  const int n_chunks = r_hi > r_lo ? (r_hi - r_lo + DWL_ROWS - 1) / DWL_ROWS : 0;
  if (n_chunks > 0) {
    fetch(r_lo);
    park(0);
  }
  __syncthreads();
  for (int b = 0; b < n_chunks; ++b) {
    const int buf = b & 1;
    if (b + 1 < n_chunks) fetch(r_lo + (b + 1) * DWL_ROWS);
#pragma unroll
    for (int u = 0; u < DWL_ROWS / 2; ++u) {
      const float2 x = *(const float2*)&sA[buf][2 * u + half][wi * 64 + 2 * l31];
      const float2 y = *(const float2*)&sB[buf][2 * u + half][wc * 64 + 2 * l31];
      acc[0][0] = mfma32(x.x, y.x, acc[0][0]);
      acc[0][1] = mfma32(x.x, y.y, acc[0][1]);
      acc[1][0] = mfma32(x.y, y.x, acc[1][0]);
      acc[1][1] = mfma32(x.y, y.y, acc[1][1]);
      if (want_db) {
        sa0 += x.x;
        sa1 += x.y;
      }
    }
    if (b + 1 < n_chunks) park(buf ^ 1);
    __syncthreads();
  }
  float* out = a.slab + (size_t)z * a.M * a.N;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wi * 64 + 2 * mfma32_row(r, lane) + q;
      const int j = j0 + wc * 64 + 2 * l31;
      *(float2*)(out + (size_t)i * a.N + j) = float2{acc[q][0][r], acc[q][1][r]};
    }
  if (want_db) {  // lane half h holds the rows of parity h: even rows + odd rows
    sa0 += __shfl_down(sa0, 32);
    sa1 += __shfl_down(sa1, 32);
    if (half == 0) {
      a.dbslab[(size_t)z * a.M + i0 + wi * 64 + 2 * l31] = sa0;
      a.dbslab[(size_t)z * a.M + i0 + wi * 64 + 2 * l31 + 1] = sa1;
    }
  }
}
// C[i][j] = sum_z slab[z][i][j]; db[i] = sum_z dbslab[z][i]
__global__ __launch_bounds__(256) void ae_dw_sum_kernel(const float* __restrict__ slab, const float* __restrict__ dbslab, int nz, int M,
                                                        int N, float* __restrict__ C, int ldc, float* __restrict__ db) {
  __shared__ float red[4][64];
  const size_t mn = (size_t)M * N, nb_c = (mn + 63) / 64;
  if (blockIdx.x < nb_c) {
    const size_t e = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = slices_sum64(slab, mn, nz, e, e < mn, red);
    if (threadIdx.x < 64 && e < mn) C[(e / N) * ldc + (e % N)] = s;
  } else {
    const size_t i = (size_t)(blockIdx.x - nb_c) * 64 + (threadIdx.x & 63);
    const float s = slices_sum64(dbslab, (size_t)M, nz, i, i < (size_t)M, red);
    if (threadIdx.x < 64 && i < (size_t)M) db[i] = s;
  }
}

// column sums: out[c] = sum_rows X[row][c], in two fixed-order stages: grid (64-column blocks, row slices) -> part[slice][c],
// then one thread per column adds the slices.  (One workgroup per 64 columns walking all rows was 45 % of the step.)
#define AE_COLSUM_SLICES 64
__global__ __launch_bounds__(256) void ae_colsum_kernel(const float* __restrict__ X, int rows, int N, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rg = threadIdx.x >> 6;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.0f;
  if (c < N)
    for (int r = r0 + rg; r < r1; r += 4) s += X[(long)r * N + c];
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < N) part[(size_t)blockIdx.y * N + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
__global__ __launch_bounds__(256) void ae_colsum_finish_kernel(const float* __restrict__ part, int nslices, int N, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float s = 0.0f;
  for (int z = 0; z < nslices; ++z) s += part[(size_t)z * N + c];
  out[c] = s;
}

// ------------------------------------------------------------------------------------------------
// x = dropout(emb[token] * sqrt(d) + pe[pos])      (model.py:126-133, :75-94)   wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ae_embed_kernel(const float* __restrict__ emb, const float* __restrict__ pe,
                                                       const int32_t* __restrict__ tok, const int32_t* __restrict__ pos,
                                                       const int32_t* __restrict__ row_sent, int n, int vocab, Drop dr,
                                                       float* __restrict__ x) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const int t = min(max(tok[r], 0), vocab - 1);
  const f32x4 e = *(const f32x4*)(emb + (size_t)t * 256 + lane * 4);
  const f32x4 p = *(const f32x4*)(pe + (size_t)pos[r] * 256 + lane * 4);
  f32x4 v = e * 16.0f + p;
  const uint64_t base = ((uint64_t)row_sent[r] * 64 + pos[r]) * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] *= drop_factor(dr, base + j);
  *(f32x4*)(x + (size_t)r * 256 + lane * 4) = v;
}
// d emb[token] += sqrt(d) * dropout'(dx), scattered over the rows of each token.  The sums are taken in 64-bit FIXED POINT
// (units of 2^-44: exact integer atomics, so the result does not depend on the order the rows arrive in -- float atomics
// made the embedding gradient the one tensor of the step that differed in its last bits from run to run); range
// +-2^19 per element, resolution 5.7e-14; ae_embed_finish_kernel converts and adds into the gradient.
#define AE_EMB_FIX 17592186044416.0 /* 2^44 */
__global__ __launch_bounds__(256) void ae_embed_bwd_kernel(const float* __restrict__ dx, const int32_t* __restrict__ tok,
                                                           const int32_t* __restrict__ pos, const int32_t* __restrict__ row_sent,
                                                           int n, int vocab, Drop dr, unsigned long long* __restrict__ acc) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const int t = min(max(tok[r], 0), vocab - 1);
  const f32x4 g = *(const f32x4*)(dx + (size_t)r * 256 + lane * 4);
  const uint64_t base = ((uint64_t)row_sent[r] * 64 + pos[r]) * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double v = fmin(fmax((double)(16.0f * g[j] * drop_factor(dr, base + j)), -524287.0), 524287.0) * AE_EMB_FIX;
    const long long q = __double2ll_rn(v);
    if (q != 0) atomicAdd(acc + (size_t)t * 256 + lane * 4 + j, (unsigned long long)q);
  }
}
__global__ __launch_bounds__(256) void ae_embed_finish_kernel(const unsigned long long* __restrict__ acc, long n, float* __restrict__ demb) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const long long q = (long long)acc[e];
  if (q != 0) demb[e] += (float)((double)q * (1.0 / AE_EMB_FIX));
}

// ------------------------------------------------------------------------------------------------
// the reference's LayerNorm (model.py:97-108): a (x - mean) / (std_unbiased + 1e-6) + b     wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ae_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                        const float* __restrict__ b, int n, float* __restrict__ y) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const f32x4 v = *(const f32x4*)(x + (size_t)r * 256 + lane * 4);
  const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
  const f32x4 c = v - mu;
  const float s = sqrtf(wave_sum(c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) * (1.0f / 255.0f));
  const float inv = 1.0f / (s + 1e-6f);
  const f32x4 ga = *(const f32x4*)(a + lane * 4), gb = *(const f32x4*)(b + lane * 4);
  *(f32x4*)(y + (size_t)r * 256 + lane * 4) = ga * (c * inv) + gb;
}
// dx (+)= LN'(dy); part[block] = (sum dy * xhat, sum dy) over the block's rows [2][256]; ae_ln_bwd_finish_kernel adds the
// blocks in index order (the first form ran on 64 workgroups with float atomics into da / db: a quarter of the chip,
// and sums whose last bits depended on the order the workgroups finished in)
__global__ __launch_bounds__(256) void ae_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ a, int n, float* __restrict__ dx,
                                                        int accumulate, float* __restrict__ part) {
  __shared__ float sa[4][256], sb[4][256];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
  const f32x4 ga = *(const f32x4*)(a + lane * 4);
  for (int r = blockIdx.x * 4 + w; r < n; r += gridDim.x * 4) {
    const f32x4 v = *(const f32x4*)(x + (size_t)r * 256 + lane * 4);
    const f32x4 g = *(const f32x4*)(dy + (size_t)r * 256 + lane * 4);
    const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
    const f32x4 c = v - mu;
    const float s = sqrtf(wave_sum(c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) * (1.0f / 255.0f));
    const float d = s + 1e-6f, inv = 1.0f / d;
    pa += g * (c * inv);
    pb += g;
    const f32x4 gy = g * ga;
    const float dot = wave_sum(gy.x * c.x + gy.y * c.y + gy.z * c.z + gy.w * c.w);
    // d/dxc of xc / (s + eps): gy / d - (gy . xc) / d^2 * xc / (255 s)
    const float k = s > 0.0f ? dot * inv * inv / (255.0f * s) : 0.0f;
    f32x4 dxc = gy * inv - k * c;
    const float m = wave_sum(dxc.x + dxc.y + dxc.z + dxc.w) * (1.0f / 256.0f);
    dxc = dxc - m;
    float* o = dx + (size_t)r * 256 + lane * 4;
    if (accumulate) dxc += *(const f32x4*)o;
    *(f32x4*)o = dxc;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sa[w][lane * 4 + j] = pa[j];
    sb[w][lane * 4 + j] = pb[j];
  }
  __syncthreads();
  const int c = threadIdx.x;
  part[(size_t)blockIdx.x * 512 + c] = (sa[0][c] + sa[1][c]) + (sa[2][c] + sa[3][c]);
  part[(size_t)blockIdx.x * 512 + 256 + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
}
__global__ __launch_bounds__(256) void ae_ln_bwd_finish_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ da,
                                                               float* __restrict__ db) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);  // 0 .. 255: da, 256 .. 511: db
  const float s = slices_sum64(part, 512, nblocks, (size_t)c, true, red);
  if (threadIdx.x < 64) {
    float* o = c < 256 ? da + c : db + (c - 256);
    *o += s;
  }
}

// ------------------------------------------------------------------------------------------------
// elementwise pieces
// ------------------------------------------------------------------------------------------------
// out = x + dropout(y)     (SublayerConnection, model.py:121-123); width 256
__global__ __launch_bounds__(256) void ae_resid_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const int32_t* __restrict__ pos, const int32_t* __restrict__ row_sent,
                                                       int n, Drop dr, float* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  f32x4 v = *(const f32x4*)(y + (size_t)r * 256 + lane * 4);
  const uint64_t base = ((uint64_t)row_sent[r] * 64 + pos[r]) * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] *= drop_factor(dr, base + j);
  *(f32x4*)(out + (size_t)r * 256 + lane * 4) = *(const f32x4*)(x + (size_t)r * 256 + lane * 4) + v;
}
// dy = dropout'(dout)
__global__ __launch_bounds__(256) void ae_resid_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ pos,
                                                           const int32_t* __restrict__ row_sent, int n, Drop dr,
                                                           float* __restrict__ dy) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  f32x4 v = *(const f32x4*)(dout + (size_t)r * 256 + lane * 4);
  const uint64_t base = ((uint64_t)row_sent[r] * 64 + pos[r]) * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] *= drop_factor(dr, base + j);
  *(f32x4*)(dy + (size_t)r * 256 + lane * 4) = v;
}
// h = dropout(relu(h)) in place, width dff  (PositionwiseFeedForward, model.py:71-72)
__global__ __launch_bounds__(256) void ae_relu_drop_kernel(float* __restrict__ h, const int32_t* __restrict__ pos,
                                                           const int32_t* __restrict__ row_sent, int n, int dff, Drop dr) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)n * dff) return;
  const int r = (int)(e / dff), c = (int)(e % dff);
  const float v = fmaxf(h[e], 0.0f);
  h[e] = v * drop_factor(dr, ((uint64_t)row_sent[r] * 64 + pos[r]) * dff + c);
}
// dh *= [h_post > 0] * scale  (h_post > 0 iff relu passed AND the unit was kept)
__global__ __launch_bounds__(256) void ae_relu_drop_bwd_kernel(float* __restrict__ dh, const float* __restrict__ hpost, long n,
                                                               float scale) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  dh[e] = hpost[e] > 0.0f ? dh[e] * scale : 0.0f;
}
__global__ __launch_bounds__(256) void ae_add_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) a[e] += b[e];
}
// a += parts[0] + parts[1] + ..., added one after the other as nv calls of ae_add_kernel would
__global__ __launch_bounds__(256) void ae_add_views_kernel(float* __restrict__ a, const float* __restrict__ parts, int nv, long n) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = a[e];
  for (int v = 0; v < nv; ++v) s += parts[(size_t)v * n + e];
  a[e] = s;
}

// latent[s] = sum over the sentence's rows of sigmoid(y[row])     (model.py:236-240)  block per sentence, thread = feature
__global__ __launch_bounds__(256) void ae_latent_kernel(const float* __restrict__ y, const int32_t* __restrict__ off, int n_sent,
                                                        float* __restrict__ lat) {
  const int s = blockIdx.x, c = threadIdx.x;
  float acc = 0.0f;
  for (int r = off[s]; r < off[s + 1]; ++r) acc += 1.0f / (1.0f + expf(-y[(size_t)r * 256 + c]));
  lat[(size_t)s * 256 + c] = acc;
}
__global__ __launch_bounds__(256) void ae_latent_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dlat,
                                                            const int32_t* __restrict__ off, int n_sent, float* __restrict__ dy) {
  const int s = blockIdx.x, c = threadIdx.x;
  const float g = dlat[(size_t)s * 256 + c];
  for (int r = off[s]; r < off[s + 1]; ++r) {
    const float sg = 1.0f / (1.0f + expf(-y[(size_t)r * 256 + c]));
    dy[(size_t)r * 256 + c] = g * sg * (1.0f - sg);
  }
}

// ------------------------------------------------------------------------------------------------
// self-attention of one (sentence, head): L <= 64 positions, d_k = 64 (model.py:18-60).  One wave; lane = query row
// (forward, dq) resp. key row (dk, dv).  P (softmax, before dropout) is saved [row][head][64] for the backward pass.
// ------------------------------------------------------------------------------------------------
#define ATT_TILE (64 * 65)
__global__ __launch_bounds__(64) void ae_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const int32_t* __restrict__ off,
                                                         int causal, Drop dr, float* __restrict__ P, float* __restrict__ out,
                                                         int short_elsewhere) {
  extern __shared__ float sm[];
  float* Ks = sm;             // [64][65]
  float* Vs = sm + ATT_TILE;
  const int s = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
  const int r0 = off[s], L = off[s + 1] - r0;
  if (L <= short_elsewhere) return;
  for (int e = i; e < L * 64; e += 64) {
    const int j = e >> 6, d = e & 63;
    Ks[j * 65 + d] = k[(size_t)(r0 + j) * 256 + h * 64 + d];
    Vs[j * 65 + d] = v[(size_t)(r0 + j) * 256 + h * 64 + d];
  }
  __syncthreads();
  if (i >= L) return;
  float qi[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) qi[d] = q[(size_t)(r0 + i) * 256 + h * 64 + d];
  float* Pi = P + ((size_t)(r0 + i) * 4 + h) * 64;
  const int nk = causal ? i + 1 : L;
  float m = -INFINITY;
  for (int j = 0; j < nk; ++j) {
    float sc = 0.0f;
#pragma unroll
    for (int d = 0; d < 64; ++d) sc = fmaf(qi[d], Ks[j * 65 + d], sc);
    sc *= 0.125f;
    Pi[j] = sc;
    m = fmaxf(m, sc);
  }
  float sum = 0.0f;
  for (int j = 0; j < nk; ++j) {
    const float e = expf(Pi[j] - m);
    Pi[j] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  float o[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.0f;
  const uint64_t base = (((uint64_t)s * 4 + h) * 64 + i) * 64;
  for (int j = 0; j < nk; ++j) {
    const float p = Pi[j] * inv;
    Pi[j] = p;
    const float pd = p * drop_factor(dr, base + j);
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = fmaf(pd, Vs[j * 65 + d], o[d]);
  }
  for (int j = nk; j < 64; ++j) Pi[j] = 0.0f;
#pragma unroll
  for (int d = 0; d < 64; ++d) out[(size_t)(r0 + i) * 256 + h * 64 + d] = o[d];
}

__global__ __launch_bounds__(64) void ae_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ P,
                                                         const float* __restrict__ dout, const int32_t* __restrict__ off,
                                                         int causal, Drop dr, float* __restrict__ dq, float* __restrict__ dk,
                                                         float* __restrict__ dv, int short_elsewhere) {
  extern __shared__ float sm[];
  float* Ks = sm;                 // [64][65]
  float* Vs = sm + ATT_TILE;
  float* Qs = sm + 2 * ATT_TILE;
  float* Gs = sm + 3 * ATT_TILE;  // d out
  float* Ss = sm + 4 * ATT_TILE;  // dS [query][key]
  float* Ds = sm + 5 * ATT_TILE;  // dropped P [query][key]
  const int s = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
  const int r0 = off[s], L = off[s + 1] - r0;
  if (L <= short_elsewhere) return;
  for (int e = i; e < L * 64; e += 64) {
    const int j = e >> 6, d = e & 63;
    const size_t g = (size_t)(r0 + j) * 256 + h * 64 + d;
    Ks[j * 65 + d] = k[g];
    Vs[j * 65 + d] = v[g];
    Qs[j * 65 + d] = q[g];
    Gs[j * 65 + d] = dout[g];
  }
  for (int e = i; e < 64 * 65; e += 64) Ss[e] = Ds[e] = 0.0f;
  __syncthreads();
  if (i < L) {
    const float* Pi = P + ((size_t)(r0 + i) * 4 + h) * 64;
    const int nk = causal ? i + 1 : L;
    const uint64_t base = (((uint64_t)s * 4 + h) * 64 + i) * 64;
    float dot = 0.0f;
    for (int j = 0; j < nk; ++j) {
      float dpd = 0.0f;
#pragma unroll
      for (int d = 0; d < 64; ++d) dpd = fmaf(Gs[i * 65 + d], Vs[j * 65 + d], dpd);
      const float f = drop_factor(dr, base + j);
      const float dp = dpd * f;
      Ds[i * 65 + j] = Pi[j] * f;
      Ss[i * 65 + j] = dp;
      dot = fmaf(dp, Pi[j], dot);
    }
    float dqi[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) dqi[d] = 0.0f;
    for (int j = 0; j < nk; ++j) {
      const float ds = Pi[j] * (Ss[i * 65 + j] - dot) * 0.125f;  // d score (the 1/sqrt(d_k) folded in)
      Ss[i * 65 + j] = ds;
#pragma unroll
      for (int d = 0; d < 64; ++d) dqi[d] = fmaf(ds, Ks[j * 65 + d], dqi[d]);
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) dq[(size_t)(r0 + i) * 256 + h * 64 + d] = dqi[d];
  }
  __syncthreads();
  if (i < L) {  // lane = key row j
    float dkj[64], dvj[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) dkj[d] = dvj[d] = 0.0f;
    for (int qi = 0; qi < L; ++qi) {
      const float ds = Ss[qi * 65 + i], pd = Ds[qi * 65 + i];
#pragma unroll
      for (int d = 0; d < 64; ++d) {
        dkj[d] = fmaf(ds, Qs[qi * 65 + d], dkj[d]);
        dvj[d] = fmaf(pd, Gs[qi * 65 + d], dvj[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) {
      dk[(size_t)(r0 + i) * 256 + h * 64 + d] = dkj[d];
      dv[(size_t)(r0 + i) * 256 + h * 64 + d] = dvj[d];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same two passes for sentences of at most 16 positions (all of yelp; the kernels above keep the longer ones and
// leave these alone, and the other way round: both are launched, each workgroup looks at its sentence's length first).
// One workgroup per sentence, one wave per head; lane = (row i = lane & 15, quarter c = lane >> 4 of the head's 64
// features), so a 9-token sentence keeps 36 lanes of every wave busy where the lane-per-row kernels keep 9, and a tile
// is 4 KB of LDS instead of 16.6 (two workgroups per CU instead of one WAVE per CU in the backward pass).  Dot products
// over the 64 features: 16 terms in the lane, then the four quarters by two xor-shuffles ((c0 + c1) + (c2 + c3), the same
// value in all four lanes).  P keeps its [row][head][64] layout; only its first 16 entries are written and read here.
// ------------------------------------------------------------------------------------------------
#define ATT16_TILE (16 * 64)
// the key loops are unrolled 16 times (their per-key results live in registers); without a fence every four keys the
// compiler hoists all 64 LDS reads of a loop to its top and spills
#define ATT16_FENCE(j) do { if (((j) & 3) == 3) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
__device__ __forceinline__ float quarter_sum(float p) {
  p += __shfl_xor(p, 16);
  p += __shfl_xor(p, 32);
  return p;
}
// dropout factors of one query row's 16 keys: quarter c hashes keys 4c .. 4c + 3, the row's four lanes exchange them
// (the counter hash is ~ 60 integer instructions per element; no LDS read shares a basic block with its branch)
__device__ __forceinline__ void att16_drop_factors(const Drop& dr, uint64_t base, int nk, int i, int c, float (&f)[16]) {
  float own[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) own[t] = 4 * c + t < nk ? drop_factor(dr, base + 4 * c + t) : 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) f[j] = __shfl(own[j & 3], i + 16 * (j >> 2));
}
// rows j = (lane >> 4) + 4 t of a [16][64] tile from the head's 64 columns of a [rows][256] matrix
__device__ __forceinline__ void att16_stage(float* __restrict__ tile, const float* __restrict__ src, int r0, int L, int h, int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = (lane >> 4) + 4 * t;
    f32x4 v = {0, 0, 0, 0};
    if (j < L) v = *(const f32x4*)(src + (size_t)(r0 + j) * 256 + h * 64 + 4 * (lane & 15));
    *(f32x4*)(tile + j * 64 + 4 * (lane & 15)) = v;
  }
}
__global__ __launch_bounds__(256, 4) void ae_attn16_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const int32_t* __restrict__ off,
                                                            int causal, Drop dr, float* __restrict__ P, float* __restrict__ out) {
  __shared__ __align__(16) float sm[4][2 * ATT16_TILE + 256];
  const int s = blockIdx.x, h = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, c = lane >> 4;
  const int r0 = off[s], L = off[s + 1] - r0;
  if (L > 16) return;
  float* Ks = sm[h];
  float* Vs = sm[h] + ATT16_TILE;
  float* Pd = Vs + ATT16_TILE;  // dropped P [query][key]
  att16_stage(Ks, k, r0, L, h, lane);
  att16_stage(Vs, v, r0, L, h, lane);
  __syncthreads();
  if (i >= L) return;
  f32x4 qi[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qi[t] = *(const f32x4*)(q + (size_t)(r0 + i) * 256 + h * 64 + 16 * c + 4 * t);
  const int nk = causal ? i + 1 : L;
  float sc[16];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float p = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 kk = *(const f32x4*)(Ks + j * 64 + 16 * c + 4 * t);
      p = fmaf(qi[t].x, kk.x, p); p = fmaf(qi[t].y, kk.y, p); p = fmaf(qi[t].z, kk.z, p); p = fmaf(qi[t].w, kk.w, p);
    }
    sc[j] = quarter_sum(p) * 0.125f;
    if (j < nk) m = fmaxf(m, sc[j]);
    ATT16_FENCE(j);
  }
  float sum = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sc[j] = j < nk ? expf(sc[j] - m) : 0.0f;
    sum += sc[j];
  }
  const float inv = 1.0f / sum;
  const uint64_t base = (((uint64_t)s * 4 + h) * 64 + i) * 64;
  float f[16];
  att16_drop_factors(dr, base, nk, i, c, f);
#pragma unroll
  for (int j = 0; j < 16; ++j) sc[j] *= inv;
  // quarter c keeps entries 4c .. 4c + 3 of the row: P to global memory, the dropped P to LDS for the rolled loop below
  // (an unrolled P.V loop gets its 64 LDS reads hoisted in front of its 256 FMAs, and spills)
  const f32x4 pw = {c == 0 ? sc[0] : c == 1 ? sc[4] : c == 2 ? sc[8] : sc[12], c == 0 ? sc[1] : c == 1 ? sc[5] : c == 2 ? sc[9] : sc[13],
                    c == 0 ? sc[2] : c == 1 ? sc[6] : c == 2 ? sc[10] : sc[14], c == 0 ? sc[3] : c == 1 ? sc[7] : c == 2 ? sc[11] : sc[15]};
  const f32x4 fw = {c == 0 ? f[0] : c == 1 ? f[4] : c == 2 ? f[8] : f[12], c == 0 ? f[1] : c == 1 ? f[5] : c == 2 ? f[9] : f[13],
                    c == 0 ? f[2] : c == 1 ? f[6] : c == 2 ? f[10] : f[14], c == 0 ? f[3] : c == 1 ? f[7] : c == 2 ? f[11] : f[15]};
  *(f32x4*)(P + ((size_t)(r0 + i) * 4 + h) * 64 + 4 * c) = pw;
  *(f32x4*)(Pd + i * 16 + 4 * c) = pw * fw;
  __syncthreads();
  f32x4 o[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 2
  for (int j = 0; j < nk; ++j) {
    const float pd = Pd[i * 16 + j];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 vv = *(const f32x4*)(Vs + j * 64 + 16 * c + 4 * t);
      o[t].x = fmaf(pd, vv.x, o[t].x); o[t].y = fmaf(pd, vv.y, o[t].y); o[t].z = fmaf(pd, vv.z, o[t].z); o[t].w = fmaf(pd, vv.w, o[t].w);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) *(f32x4*)(out + (size_t)(r0 + i) * 256 + h * 64 + 16 * c + 4 * t) = o[t];
}

__global__ __launch_bounds__(256, 4) void ae_attn16_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ P,
                                                               const float* __restrict__ dout, const int32_t* __restrict__ off,
                                                               int causal, Drop dr, float* __restrict__ dq, float* __restrict__ dk,
                                                               float* __restrict__ dv) {
  __shared__ __align__(16) float sm[4][2 * ATT16_TILE + 2 * 256];
  const int s = blockIdx.x, h = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, c = lane >> 4;
  const int r0 = off[s], L = off[s + 1] - r0;
  if (L > 16) return;
  // two tiles: K and V for the first phase (lane = query row), then Q and d out for the second (lane = key row); the
  // second pair waits in registers from the start
  float* Ks = sm[h];
  float* Vs = Ks + ATT16_TILE;
  float* Qs = Ks;
  float* Gs = Vs;
  float* Ss = Vs + ATT16_TILE;  // dS [query][key]
  float* Ds = Ss + 256;         // dropped P [query][key]
  f32x4 rq[4], rg[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = (lane >> 4) + 4 * t;
    rq[t] = rg[t] = f32x4{0, 0, 0, 0};
    if (j < L) {
      rq[t] = *(const f32x4*)(q + (size_t)(r0 + j) * 256 + h * 64 + 4 * (lane & 15));
      rg[t] = *(const f32x4*)(dout + (size_t)(r0 + j) * 256 + h * 64 + 4 * (lane & 15));
    }
  }
  att16_stage(Ks, k, r0, L, h, lane);
  att16_stage(Vs, v, r0, L, h, lane);
  __syncthreads();
  {  // lane = (query row i, quarter c)
    const bool live = i < L;
    const int nk = live ? (causal ? i + 1 : L) : 0;
    const uint64_t base = (((uint64_t)s * 4 + h) * 64 + i) * 64;
    f32x4 gi[4], pr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      gi[t] = live ? *(const f32x4*)(dout + (size_t)(r0 + i) * 256 + h * 64 + 16 * c + 4 * t) : f32x4{0, 0, 0, 0};
      pr[t] = live ? *(const f32x4*)(P + ((size_t)(r0 + i) * 4 + h) * 64 + 4 * t) : f32x4{0, 0, 0, 0};
    }
    float dp[16], dd[16];
    att16_drop_factors(dr, base, nk, i, c, dd);
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float p = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 vv = *(const f32x4*)(Vs + j * 64 + 16 * c + 4 * t);
        p = fmaf(gi[t].x, vv.x, p); p = fmaf(gi[t].y, vv.y, p); p = fmaf(gi[t].z, vv.z, p); p = fmaf(gi[t].w, vv.w, p);
      }
      const float dpd = quarter_sum(p);
      const float pij = pr[j >> 2][j & 3];
      const float f = dd[j];
      dp[j] = dpd * f;
      dd[j] = pij * f;
      if (j < nk) dot = fmaf(dp[j], pij, dot);
      ATT16_FENCE(j);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) dp[j] = j < nk ? pr[j >> 2][j & 3] * (dp[j] - dot) * 0.125f : 0.0f;  // d score (the 1/sqrt(d_k) folded in)
    // quarter c stores entries 4c .. 4c + 3 of its row of dS and of the dropped P (all four quarters hold the same values)
    const f32x4 sw = {c == 0 ? dp[0] : c == 1 ? dp[4] : c == 2 ? dp[8] : dp[12], c == 0 ? dp[1] : c == 1 ? dp[5] : c == 2 ? dp[9] : dp[13],
                      c == 0 ? dp[2] : c == 1 ? dp[6] : c == 2 ? dp[10] : dp[14], c == 0 ? dp[3] : c == 1 ? dp[7] : c == 2 ? dp[11] : dp[15]};
    const f32x4 dw = {c == 0 ? dd[0] : c == 1 ? dd[4] : c == 2 ? dd[8] : dd[12], c == 0 ? dd[1] : c == 1 ? dd[5] : c == 2 ? dd[9] : dd[13],
                      c == 0 ? dd[2] : c == 1 ? dd[6] : c == 2 ? dd[10] : dd[14], c == 0 ? dd[3] : c == 1 ? dd[7] : c == 2 ? dd[11] : dd[15]};
    *(f32x4*)(Ss + i * 16 + 4 * c) = sw;
    *(f32x4*)(Ds + i * 16 + 4 * c) = dw;
    __syncthreads();
    f32x4 dqi[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 2
    for (int j = 0; j < nk; ++j) {  // rolled, dS from LDS: unrolled, its 64 LDS reads are hoisted in front of the FMAs and spill
      const float ds = Ss[i * 16 + j];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 kk = *(const f32x4*)(Ks + j * 64 + 16 * c + 4 * t);
        dqi[t].x = fmaf(ds, kk.x, dqi[t].x); dqi[t].y = fmaf(ds, kk.y, dqi[t].y); dqi[t].z = fmaf(ds, kk.z, dqi[t].z); dqi[t].w = fmaf(ds, kk.w, dqi[t].w);
      }
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < 4; ++t) *(f32x4*)(dq + (size_t)(r0 + i) * 256 + h * 64 + 16 * c + 4 * t) = dqi[t];
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = (lane >> 4) + 4 * t;
    *(f32x4*)(Qs + j * 64 + 4 * (lane & 15)) = rq[t];
    *(f32x4*)(Gs + j * 64 + 4 * (lane & 15)) = rg[t];
  }
  __syncthreads();
  if (i < L) {  // lane = (key row j = i, quarter c): sums over the query rows in index order
    f32x4 dkj[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, dvj[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int qi = 0; qi < L; ++qi) {
      const float ds = Ss[qi * 16 + i], pd = Ds[qi * 16 + i];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 qq = *(const f32x4*)(Qs + qi * 64 + 16 * c + 4 * t), gg = *(const f32x4*)(Gs + qi * 64 + 16 * c + 4 * t);
        dkj[t].x = fmaf(ds, qq.x, dkj[t].x); dkj[t].y = fmaf(ds, qq.y, dkj[t].y); dkj[t].z = fmaf(ds, qq.z, dkj[t].z); dkj[t].w = fmaf(ds, qq.w, dkj[t].w);
        dvj[t].x = fmaf(pd, gg.x, dvj[t].x); dvj[t].y = fmaf(pd, gg.y, dvj[t].y); dvj[t].z = fmaf(pd, gg.z, dvj[t].z); dvj[t].w = fmaf(pd, gg.w, dvj[t].w);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *(f32x4*)(dk + (size_t)(r0 + i) * 256 + h * 64 + 16 * c + 4 * t) = dkj[t];
      *(f32x4*)(dv + (size_t)(r0 + i) * 256 + h * 64 + 16 * c + 4 * t) = dvj[t];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// decoder source attention over ONE memory position (model.py:195, :264-269): softmax == 1, so the attended value of
// row r is c[sentence(r)] = Wv latent + bv, times the p_attn dropout factor of (row, head)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ae_src_gather_kernel(const float* __restrict__ c, const int32_t* __restrict__ pos,
                                                            const int32_t* __restrict__ row_sent, int n, Drop dr,
                                                            float* __restrict__ u) {
  const int r = blockIdx.x, t = threadIdx.x;
  if (r >= n) return;
  const int s = row_sent[r];
  u[(size_t)r * 256 + t] = c[(size_t)s * 256 + t] * drop_factor(dr, ((uint64_t)s * 64 + pos[r]) * 4 + (t >> 6));
}
// dc[s] = sum over the sentence's rows of du[row] * factor(row, head)
__global__ __launch_bounds__(256) void ae_src_gather_bwd_kernel(const float* __restrict__ du, const int32_t* __restrict__ pos,
                                                                const int32_t* __restrict__ off, int n_sent, Drop dr,
                                                                float* __restrict__ dc) {
  const int s = blockIdx.x, t = threadIdx.x;
  float acc = 0.0f;
  for (int r = off[s]; r < off[s + 1]; ++r)
    acc += du[(size_t)r * 256 + t] * drop_factor(dr, ((uint64_t)s * 64 + pos[r]) * 4 + (t >> 6));
  dc[(size_t)s * 256 + t] = acc;
}

// ------------------------------------------------------------------------------------------------
// generator + LabelSmoothing (model.py:200-207, :389-411): per row  loss = sum_v t_v (log t_v - logp_v),
// t = confidence at the target, smoothing / (V - 2) elsewhere, 0 at PAD;  logits <- (softmax - t) * gscale  in place
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ae_xent_kernel(float* __restrict__ logits, const int32_t* __restrict__ target, int n, int V,
                                                      int pad, float smoothing, float gscale, float* __restrict__ row_loss) {
  __shared__ float red[256];
  const int r = blockIdx.x, t = threadIdx.x;
  float* x = logits + (size_t)r * V;
  float m = -INFINITY;
  for (int c = t; c < V; c += 256) m = fmaxf(m, x[c]);
  red[t] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] = fmaxf(red[t], red[t + o]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  float se = 0.0f, sx = 0.0f;
  for (int c = t; c < V; c += 256) {
    se += expf(x[c] - m);
    sx += x[c];
  }
  red[t] = se;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float lse = m + logf(red[0]);
  __syncthreads();
  red[t] = sx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float sumx = red[0];
  const int y = target[r];
  const bool live = y != pad;
  const float sv = smoothing / (float)(V - 2), conf = 1.0f - smoothing;
  if (t == 0) {
    float l = 0.0f;
    if (live) {
      // sum_v t_v log t_v  -  sum_v t_v logp_v,  logp_v = x_v - lse;  "others" = all v except the target and PAD
      const float so = sumx - x[y] - x[pad] - (float)(V - 2) * lse;
      l = conf * logf(conf) + (smoothing > 0.0f ? smoothing * logf(sv) : 0.0f) - conf * (x[y] - lse) - sv * so;
    }
    row_loss[r] = l;
  }
  __syncthreads();
  for (int c = t; c < V; c += 256) {
    float g = 0.0f;
    if (live) {
      const float tv = c == y ? conf : (c == pad ? 0.0f : sv);
      g = (expf(x[c] - lse) - tv) * gscale;
    }
    x[c] = g;
  }
}
// out[0] = scale * sum(v[0..n))   (one block; fixed order per thread, tree over threads)
__global__ __launch_bounds__(256) void ae_sum_kernel(const float* __restrict__ v, int n, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}
__global__ void ae_fill_kernel(float* p, long n, float v) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) p[e] = v;
}
__global__ void ae_row_sent_kernel(const int32_t* __restrict__ off, int n_sent, int32_t* __restrict__ row_sent) {
  const int s = blockIdx.x;
  for (int r = off[s] + threadIdx.x; r < off[s + 1]; r += blockDim.x) row_sent[r] = s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline size_t up256(size_t b) { return (b + 255) / 256 * 256; }
struct Arena {
  char* base;
  size_t off;
  float* f(size_t n) {
    float* p = base ? (float*)(base + off) : nullptr;
    off += up256(n * 4);
    return p;
  }
};
struct LayerSave {
  float *x_in, *n0, *q, *k, *v, *P, *att, *x1, *u, *x2, *n2, *h;  // u / x2: decoder only (source-attention sublayer)
};
#define AE_MAX_SLICES 16
#define AE_DW_SLAB_FLOATS (1u << 23)  /* 32 MB: 128 slices of a 256 x 256 gradient, 32 of a 256 x 1024 one */
struct Ws {
  // encoder (Re rows), decoder (Rd rows)
  int32_t *e_row_sent, *d_row_sent, *flags;
  LayerSave enc[ST_MAX_LAYERS], dec[ST_MAX_LAYERS];
  unsigned long long* emb_acc;                    // [V, 256] fixed-point sums of the embedding gradient
  float *e_x, *e_y, *lat, *dlat;                 // encoder output before / after the final norm, latents and their gradient
  float *d_x, *d_xn, *logits, *row_loss, *cc[ST_MAX_LAYERS], *dcc;
  float *ta, *tb, *tc, *td, *th;                 // scratch activations / gradients [rows, 256] x4, [rows, d_ff]
  float *sl_loss, *sl_lse, *sl_g, *sl_dori, *sl_dlike, *con_rows;
  void* sl_scratch;
  size_t sl_scratch_bytes;
  float* slab;          // split-K partial products of the weight-gradient GEMMs
  size_t slab_floats;
  float* wpack;         // fragment-packed image of the weight matrix of the product in flight (row-panel kernels)
  size_t wpack_floats;
  size_t total;
};

Ws carve(const StAeParams* p, int Re, int Se, int Rd, int B, char* base) {
  Ws w = {};
  Arena a{base, 0};
  const int L = p->n_layers, dff = p->d_ff, V = p->vocab;
  const int R = Re > Rd ? Re : Rd;
  w.flags = (int32_t*)a.f(64);
  w.e_row_sent = (int32_t*)a.f(Re);
  w.d_row_sent = (int32_t*)a.f(Rd);
  for (int l = 0; l < L; ++l) {
    LayerSave& s = w.enc[l];
    s.x_in = a.f((size_t)Re * 256); s.n0 = a.f((size_t)Re * 256); s.q = a.f((size_t)Re * 256); s.k = a.f((size_t)Re * 256);
    s.v = a.f((size_t)Re * 256); s.P = a.f((size_t)Re * 256); s.att = a.f((size_t)Re * 256); s.x1 = a.f((size_t)Re * 256);
    s.n2 = a.f((size_t)Re * 256); s.h = a.f((size_t)Re * dff);
    LayerSave& d = w.dec[l];
    d.x_in = a.f((size_t)Rd * 256); d.n0 = a.f((size_t)Rd * 256); d.q = a.f((size_t)Rd * 256); d.k = a.f((size_t)Rd * 256);
    d.v = a.f((size_t)Rd * 256); d.P = a.f((size_t)Rd * 256); d.att = a.f((size_t)Rd * 256); d.x1 = a.f((size_t)Rd * 256);
    d.u = a.f((size_t)Rd * 256); d.x2 = a.f((size_t)Rd * 256); d.n2 = a.f((size_t)Rd * 256); d.h = a.f((size_t)Rd * dff);
    w.cc[l] = a.f((size_t)B * 256);
  }
  w.emb_acc = (unsigned long long*)a.f((size_t)V * 256 * 2);
  w.e_x = a.f((size_t)Re * 256); w.e_y = a.f((size_t)Re * 256);
  w.lat = a.f((size_t)Se * 256); w.dlat = a.f((size_t)Se * 256);
  w.d_x = a.f((size_t)Rd * 256); w.d_xn = a.f((size_t)Rd * 256);
  w.logits = a.f((size_t)Rd * V); w.row_loss = a.f(Rd); w.dcc = a.f((size_t)B * 256);
  w.ta = a.f((size_t)R * 256); w.tb = a.f((size_t)R * 256); w.tc = a.f((size_t)R * 256); w.td = a.f((size_t)R * 256);
  w.th = a.f((size_t)R * dff);
  w.sl_loss = a.f(B); w.sl_lse = a.f(B); w.sl_g = a.f(B); w.sl_dori = a.f((size_t)B * 256); w.sl_dlike = a.f((size_t)B * 256);
  w.con_rows = a.f((size_t)(Se / (B > 0 ? B : 1)) * B);
  // split-contraction partial products: the strided kernel's slices, and ae_dw_kernel's [nz][M][N] + [nz][M]
  w.slab_floats = std::max((size_t)AE_MAX_SLICES * 256 * (dff > 768 ? dff : 768), (size_t)AE_DW_SLAB_FLOATS + 4 * (size_t)V * 256);
  w.slab = a.f(w.slab_floats);
  w.wpack_floats = std::max(rp_packed_floats(V, 256), std::max(rp_packed_floats(256, dff), rp_packed_floats(dff, 256)));
  w.wpack = a.f(w.wpack_floats);
  w.sl_scratch_bytes = st_simloss_scratch_bytes(B, B);
  w.sl_scratch = base ? base + a.off : nullptr;
  a.off += up256(w.sl_scratch_bytes);
  w.total = a.off;
  return w;
}

// f32 fragment image (st_rowpanel.h) of W'[n][k] = src[n * sn + k * sk]: W itself (sn = K, sk = 1) or its transpose
__global__ __launch_bounds__(256) void ae_pack_kernel(const float* __restrict__ src, long sn, long sk, int N, int K,
                                                      float* __restrict__ out) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;  // one float4 of the packed image
  const int kgs = K >> 3;
  const size_t total = (size_t)((N + 31) / 32) * kgs * 64;
  if (o >= total) return;
  const int lane = (int)(o & 63);
  const int kg = (int)((o >> 6) % kgs);
  const int blk = (int)((o >> 6) / kgs);
  const int row = blk * 32 + (lane & 31);
  f32x4 v = {0, 0, 0, 0};
  if (row < N) {
    const float* q = src + (long)row * sn + (long)(kg * 8 + 4 * (lane >> 5)) * sk;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = q[(long)j * sk];
  }
  *(f32x4*)(out + o * 4) = v;
}

// the exact-f32 redo of a forward product whose split-f16 run saw an activation outside the f16 range: packs the f32 image
// and opens the redo's row count; in range (always, in practice) it packs nothing and the redo's workgroups find 0 rows
__global__ __launch_bounds__(256) void ae_redo_pack_kernel(const int* __restrict__ flag, int rows, int* __restrict__ redo_rows,
                                                           const float* __restrict__ src, int N, int K, float* __restrict__ out) {
  const bool redo = *flag != 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *redo_rows = redo ? rows : 0;
  if (!redo) return;
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int kgs = K >> 3;
  const size_t total = (size_t)((N + 31) / 32) * kgs * 64;
  if (o >= total) return;
  const int lane = (int)(o & 63);
  const int kg = (int)((o >> 6) % kgs);
  const int blk = (int)((o >> 6) / kgs);
  const int row = blk * 32 + (lane & 31);
  f32x4 v = {0, 0, 0, 0};
  if (row < N) v = *(const f32x4*)(src + (size_t)row * K + kg * 8 + 4 * (lane >> 5));
  *(f32x4*)(out + o * 4) = v;
}

struct Ctx {
  hipStream_t st;
  int rc;
  float* slab;
  size_t slab_floats;
  float* wpack;
  size_t wpack_floats;
  int* flags;  // [0]: a forward product saw an activation outside the f16 range (sticky within the step), [1]: rows of the f32 redo
  // FORWARD products Y = X W^T + b on the split-f16 row-panel kernels (f32-grade: three f16 products per term, f32
  // accumulation, st_common.h; ~ 3x the exact-f32 rate): activations and weights are O(1) there.  The backward products
  // stay exact f32: gradients of 1e-7 .. 1e-3 sit below the f16 subnormal floor.  Each product is followed by its exact-f32
  // form gated on the device by the range flag (0 rows unless an activation left the f16 range: then every row again).
  bool h_ok(int rows, int N, int K) const {
    return g_ae_fwd_h && flags && rp_ok(rows, N, K) && K % 16 == 0;
  }
  int rp_lin_h(const float* X, const float* W, const float* bias, float* Y, int rows, int N, int K) {
    const size_t th = (size_t)((N + 31) / 32) * (K >> 4) * 64;
    hipLaunchKernelGGL(pack_weight_h_kernel<0>, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, st, W, N, K, wpack);
    ST_LAUNCH_CHECK();
    GemmArgs g = {};
    g.fmt = ST_GEMM_F16X3; g.X = X; g.n_host = rows; g.W = wpack; g.bias = bias; g.N = N; g.out = Y; g.ldo = N;
    g.range_flag = flags;
    if (K == 256) {
      const int splits = rp_splits(rows, N, 8, rp_nw(ST_GEMM_F16X3, 8));
      if (N % 32 == 0)  // the 16x16x32 projection kernel of the attack path (needs whole 32-column blocks)
        ST_TRY(launch_rp_lin_pipelined<EPI_BIAS>(g, rows, splits, st));
      else
        ST_TRY((launch_rp<EPI_BIAS, 8>(g, rows, splits, st)));
    } else {
      g.K = K;
      ST_TRY(launch_rp_kchunk<EPI_BIAS>(g, rows, st, 1));
    }
    const size_t tf = (size_t)((N + 31) / 32) * (K >> 3) * 64;
    hipLaunchKernelGGL(ae_redo_pack_kernel, dim3((unsigned)((tf + 255) / 256)), dim3(256), 0, st, (const int*)flags, rows, flags + 1, W, N, K,
                       wpack);
    ST_LAUNCH_CHECK();
    g.fmt = ST_GEMM_F32; g.range_flag = nullptr; g.n_dev = flags + 1;
    if (K == 256) return launch_rp<EPI_BIAS, 8>(g, rows, rp_splits(rows, N, 8, 8), st);
    return launch_rp_kchunk<EPI_BIAS>(g, rows, st, 1);
  }
  // Y[rows, N] (+)= X[rows, K] . W'^T + bias with W'[n][k] = Wsrc[n * sn + k * sk], on the exact-f32 ROW-PANEL kernels
  // (st_rowpanel.h: 64-row panel of X in LDS, fragment-packed W straight from L2; 110 - 130 TFLOP/s where the strided
  // kernel below reaches 45).  The weights change every step, so the image is packed in front of every product (a few
  // microseconds); contraction 256, or a multiple of 256 with a 256-wide result.
  bool rp_ok(int rows, int N, int K) const {
    return g_ae_rp && wpack && rows >= 256 && ((K == 256) || (K % 256 == 0 && N == 256)) && rp_packed_floats(N, K) <= wpack_floats;
  }
  int rp_lin(const float* X, const float* Wsrc, long sn, long sk, const float* bias, float* Y, int rows, int N, int K, int acc) {
    const size_t total = (size_t)((N + 31) / 32) * (K >> 3) * 64;
    hipLaunchKernelGGL(ae_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Wsrc, sn, sk, N, K, wpack);
    ST_LAUNCH_CHECK();
    GemmArgs g = {};
    g.fmt = ST_GEMM_F32; g.X = X; g.n_host = rows; g.W = wpack; g.bias = bias; g.N = N; g.out = Y; g.ldo = N;
    if (acc) g.resid = Y;
    if (K == 256) {
      const int splits = rp_splits(rows, N, 8, 8);
      return acc ? launch_rp<EPI_RESID, 8>(g, rows, splits, st) : launch_rp<EPI_BIAS, 8>(g, rows, splits, st);
    }
    g.K = K;
    return acc ? launch_rp_kchunk<EPI_RESID>(g, rows, st, 1) : launch_rp_kchunk<EPI_BIAS>(g, rows, st, 1);
  }
  int gemm(const float* A, long sai, long sar, const float* Bm, long sbr, long sbj, float* C, int ldc, const float* bias, int M, int N,
           int R, int accumulate) {
    if (M <= 0 || N <= 0) return ST_OK;
    GArgs a{A, Bm, C, bias, M, N, R, sai, sar, sbr, sbj, ldc, accumulate};
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    int nz = 1;
    if (tiles < 128 && R >= 2048 && slab) {  // few output tiles, long contraction: slices of >= 512 steps until ~256 workgroups
      nz = std::min(std::min(AE_MAX_SLICES, (256 + tiles - 1) / tiles), R / 512);
      while (nz > 1 && (size_t)nz * M * N > slab_floats) --nz;
    }
    const int per = nz > 1 ? (((R + nz - 1) / nz + 31) / 32) * 32 : R;
    hipLaunchKernelGGL(ae_gemm_kernel, dim3((N + 127) / 128, (M + 127) / 128, nz), dim3(256), 0, st, a, slab, per);
    if (nz > 1)
      hipLaunchKernelGGL(ae_slab_sum_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, st, a, (const float*)slab, nz);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  // dW[M, N] = dY[R, M]^T X[R, N] and db = column sums of dY, on ae_dw_kernel (slices of the contraction until ~ 4 workgroups per CU)
  bool dw_ok(int M, int N, int R) const {
    return g_ae_dw && slab && R >= 256 && (size_t)M * N + M <= slab_floats;
  }
  int dw(const float* dY, const float* X, float* dW, float* db, int M, int N, int R) {
    DwArgs a = {};
    a.A = dY; a.B = X; a.M = M; a.N = N; a.R = R; a.lda = M; a.ldb = N;
    a.tiles_n = (N + 127) / 128;
    a.tiles = a.tiles_n * ((M + 127) / 128);
    int nz = std::max(1, std::min((1024 + a.tiles - 1) / a.tiles, R / DW_MIN_SLICE_ROWS));  // ~ 4 workgroups per CU (4 waves per SIMD at 120 registers)
    while (nz > 1 && (size_t)nz * ((size_t)M * N + M) > slab_floats) --nz;
    a.per = ((R + nz - 1) / nz + 2 * DW_U - 1) / (2 * DW_U) * (2 * DW_U);
    a.nz = (R + a.per - 1) / a.per;
    a.slab = slab;
    a.dbslab = db ? slab + (size_t)a.nz * M * N : nullptr;
    const int blocks = (a.tiles * a.nz + 7) / 8 * 8;
    if (g_ae_dw == 2 && ((M | N) & 127) == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0)
      hipLaunchKernelGGL(ae_dw_lds_kernel, dim3(blocks), dim3(256), 0, st, a);
    else if (((M | N) & 1) == 0 && (((uintptr_t)dY | (uintptr_t)X) & 7) == 0)
      hipLaunchKernelGGL(ae_dw_kernel<2>, dim3(blocks), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(ae_dw_kernel<1>, dim3(blocks), dim3(256), 0, st, a);
    const size_t sum_blocks = ((size_t)M * N + 63) / 64 + (db ? (M + 63) / 64 : 0);
    hipLaunchKernelGGL(ae_dw_sum_kernel, dim3((unsigned)sum_blocks), dim3(256), 0, st, (const float*)a.slab, (const float*)a.dbslab, a.nz,
                       M, N, dW, N, db);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  // LayerNorm backward over `rows` rows: dx, and da / db += the column sums, through per-block partial sums in the slab
  int ln_bwd(const float* dy, const float* x, const float* a, int rows, float* dx, int accumulate, float* da, float* db) {
    int nb = std::max(1, std::min(1024, (rows + 3) / 4));
    while (nb > 1 && (size_t)nb * 512 > slab_floats) --nb;
    hipLaunchKernelGGL(ae_ln_bwd_kernel, dim3(nb), dim3(256), 0, st, dy, x, a, rows, dx, accumulate, slab);
    hipLaunchKernelGGL(ae_ln_bwd_finish_kernel, dim3(8), dim3(256), 0, st, (const float*)slab, nb, da, db);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  int colsum(const float* X, int rows, int N, float* out) {
    int ns = std::min(AE_COLSUM_SLICES, std::max(1, rows / 64));
    while (ns > 1 && (size_t)ns * N > slab_floats) --ns;
    if (!slab) ns = 1;
    float* part = ns > 1 ? slab : out;
    hipLaunchKernelGGL(ae_colsum_kernel, dim3((N + 63) / 64, ns), dim3(256), 0, st, X, rows, N, part);
    if (ns > 1) hipLaunchKernelGGL(ae_colsum_finish_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (const float*)part, ns, N, out);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  // Y[rows,N] = X[rows,K] W^T + b      (W row-major [N,K])
  int lin(const float* X, const float* W, const float* b, float* Y, int rows, int N, int K) {
    if (h_ok(rows, N, K)) return rp_lin_h(X, W, b, Y, rows, N, K);
    if (rp_ok(rows, N, K)) return rp_lin(X, W, K, 1, b, Y, rows, N, K, 0);
    return gemm(X, K, 1, W, 1, K, Y, N, b, rows, N, K, 0);
  }
  // dX (+)= dY W;  dW = dY^T X;  db = column sums of dY
  int lin_bwd(const float* dY, const float* X, const float* W, float* dX, int acc_dx, float* dW, float* db, int rows, int N, int K) {
    if (dX) {
      if (rp_ok(rows, K, N))   // dX[rows, K] = dY[rows, N] . W[N, K]: the "weight" of this product is W^T
        ST_TRY(rp_lin(dY, W, 1, K, nullptr, dX, rows, K, N, acc_dx));
      else
        ST_TRY(gemm(dY, N, 1, W, K, 1, dX, K, nullptr, rows, K, N, acc_dx));
    }
    if (dW && dw_ok(N, K, rows)) return dw(dY, X, dW, db, N, K, rows);
    if (dW) ST_TRY(gemm(dY, 1, N, X, K, 1, dW, K, nullptr, N, K, rows, 0));
    if (db) ST_TRY(colsum(dY, rows, N, db));
    return ST_OK;
  }
};

#define ROWS4(n) dim3(((n) + 3) / 4), dim3(256), 0, c.st
#define ELEMS(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, c.st

}  // namespace

extern "C" size_t st_ae_train_workspace_bytes(const StAeParams* p, int n_enc_rows, int n_enc_sent, int n_dec_rows, int B) {
  if (!p || n_enc_rows <= 0 || n_enc_sent <= 0 || n_dec_rows <= 0 || B <= 0) return 0;
  return carve(p, n_enc_rows, n_enc_sent, n_dec_rows, B, nullptr).total;
}

extern "C" int st_ae_train_step(const StAeParams* p, const StAeParams* g, float* grads_flat, int64_t n_params,
                                const int32_t* enc_tokens, const int32_t* enc_pos, const int32_t* enc_off, int n_enc_rows,
                                int n_enc_sent, const int32_t* dec_tokens, const int32_t* dec_pos, const int32_t* dec_targets,
                                const int32_t* dec_off, int n_dec_rows, int B, float bata, float tau, float smoothing,
                                float drop_p, uint64_t seed, float* losses_out, void* workspace, size_t workspace_bytes,
                                void* stream) {
  ae_env();
  ST_CHECK_ARG(p && g && grads_flat && n_params > 0, "NULL parameter / gradient structs");
  ST_CHECK_ARG(p->d_model == ST_D && p->n_heads == 4, "d_model=%d n_heads=%d unsupported (256, 4)", p->d_model, p->n_heads);
  ST_CHECK_ARG(p->n_layers >= 1 && p->n_layers <= ST_MAX_LAYERS && p->d_ff > 0 && p->vocab >= 4, "n_layers=%d d_ff=%d vocab=%d",
               p->n_layers, p->d_ff, p->vocab);
  ST_CHECK_ARG(B >= 1 && n_enc_sent >= 2 * B && n_enc_sent % B == 0, "n_enc_sent=%d must be (1 + views) * B with views >= 1 (B=%d)",
               n_enc_sent, B);
  ST_CHECK_ARG(B % 32 == 0, "st_ae_train_step: the similarity block's backward needs B %% 32 == 0 (B=%d)", B);
  ST_CHECK_ARG(n_enc_rows > 0 && n_dec_rows > 0, "empty batch");
  ST_CHECK_ARG(enc_tokens && enc_pos && enc_off && dec_tokens && dec_pos && dec_targets && dec_off && losses_out && workspace,
               "NULL buffer");
  ST_CHECK_ARG(drop_p >= 0.0f && drop_p < 1.0f && tau > 0.0f, "drop_p=%g tau=%g", (double)drop_p, (double)tau);
  Ws w = carve(p, n_enc_rows, n_enc_sent, n_dec_rows, B, (char*)workspace);
  if (w.total > workspace_bytes) {
    st_set_error("st_ae_train_step: workspace too small: need %zu bytes, have %zu", w.total, workspace_bytes);
    return ST_ENOMEM;
  }
  Ctx c{(hipStream_t)stream, 0, w.slab, w.slab_floats, w.wpack, w.wpack_floats, (int*)w.flags};
  ST_HIP(hipMemsetAsync(w.flags, g_ae_force_redo ? 1 : 0, 4, c.st));  // (byte fill: 0 or 0x01010101, any non-zero value flags)
  ST_HIP(hipMemsetAsync(w.flags + 1, 0, 252, c.st));
  const int Re = n_enc_rows, Se = n_enc_sent, Rd = n_dec_rows, L = p->n_layers, dff = p->d_ff, V = p->vocab;
  const int n_views = Se / B - 1;
  auto drop = [&](uint32_t site) { return Drop{seed, drop_p, drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f, site}; };
  const size_t att_fwd_lds = 2 * ATT_TILE * 4, att_bwd_lds = 6 * ATT_TILE * 4;
  const bool att16 = g_ae_att16 != 0;  // sentences of <= 16 positions on the (row, quarter) kernels
  ST_ENSURE_LDS(ae_attn_bwd_kernel, (int)att_bwd_lds);

  ST_HIP(hipMemsetAsync(grads_flat, 0, (size_t)n_params * 4, c.st));
  hipLaunchKernelGGL(ae_row_sent_kernel, dim3(Se), dim3(64), 0, c.st, enc_off, Se, w.e_row_sent);
  hipLaunchKernelGGL(ae_row_sent_kernel, dim3(B), dim3(64), 0, c.st, dec_off, B, w.d_row_sent);
  ST_LAUNCH_CHECK();

  // one pre-norm transformer layer, forward.  dec: with the source-attention sublayer between the two
  auto layer_fwd = [&](const StAeLayer& y, LayerSave& s, const float* x, float* x_out, int rows, int n_sent, const int32_t* off,
                       const int32_t* pos, const int32_t* row_sent, bool dec, uint32_t site0, const float* cconst) -> int {
    ST_HIP(hipMemcpyAsync(s.x_in, x, (size_t)rows * 1024, hipMemcpyDeviceToDevice, c.st));
    hipLaunchKernelGGL(ae_ln_fwd_kernel, ROWS4(rows), s.x_in, y.ln0_a, y.ln0_b, rows, s.n0);
    ST_TRY(c.lin(s.n0, y.wq, y.bq, s.q, rows, 256, 256));
    ST_TRY(c.lin(s.n0, y.wk, y.bk, s.k, rows, 256, 256));
    ST_TRY(c.lin(s.n0, y.wv, y.bv, s.v, rows, 256, 256));
    if (att16)
      hipLaunchKernelGGL(ae_attn16_fwd_kernel, dim3(n_sent), dim3(256), 0, c.st, s.q, s.k, s.v, off, dec ? 1 : 0, drop(site0 + 0), s.P, s.att);
    hipLaunchKernelGGL(ae_attn_fwd_kernel, dim3(n_sent, 4), dim3(64), att_fwd_lds, c.st, s.q, s.k, s.v, off, dec ? 1 : 0,
                       drop(site0 + 0), s.P, s.att, att16 ? 16 : 0);
    ST_LAUNCH_CHECK();
    ST_TRY(c.lin(s.att, y.wo, y.bo, w.ta, rows, 256, 256));
    hipLaunchKernelGGL(ae_resid_kernel, ROWS4(rows), s.x_in, w.ta, pos, row_sent, rows, drop(site0 + 1), s.x1);
    const float* xm = s.x1;
    uint32_t site = site0 + 2;
    if (dec) {
      hipLaunchKernelGGL(ae_src_gather_kernel, dim3(rows), dim3(256), 0, c.st, cconst, pos, row_sent, rows, drop(site0 + 2), s.u);
      ST_TRY(c.lin(s.u, y.xo_w, y.xo_b, w.ta, rows, 256, 256));
      hipLaunchKernelGGL(ae_resid_kernel, ROWS4(rows), s.x1, w.ta, pos, row_sent, rows, drop(site0 + 3), s.x2);
      xm = s.x2;
      site = site0 + 4;
    }
    hipLaunchKernelGGL(ae_ln_fwd_kernel, ROWS4(rows), xm, y.ln2_a, y.ln2_b, rows, s.n2);
    ST_TRY(c.lin(s.n2, y.w1, y.b1, s.h, rows, dff, 256));
    hipLaunchKernelGGL(ae_relu_drop_kernel, ELEMS((long)rows * dff), s.h, pos, row_sent, rows, dff, drop(site));
    ST_TRY(c.lin(s.h, y.w2, y.b2, w.ta, rows, 256, dff));
    hipLaunchKernelGGL(ae_resid_kernel, ROWS4(rows), xm, w.ta, pos, row_sent, rows, drop(site + 1), x_out);
    ST_LAUNCH_CHECK();
    return ST_OK;
  };
  // backward of the same layer: dx holds d(layer output) on entry and d(layer input) on exit
  auto layer_bwd = [&](const StAeLayer& y, const StAeLayer& gy, LayerSave& s, float* dx, int rows, int n_sent, const int32_t* off,
                       const int32_t* pos, const int32_t* row_sent, bool dec, uint32_t site0, const float* latent_ori,
                       float* dlat_ori) -> int {
    const float* xm = dec ? s.x2 : s.x1;
    const uint32_t site = dec ? site0 + 4 : site0 + 2;
    const float scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    // feed-forward sublayer
    hipLaunchKernelGGL(ae_resid_bwd_kernel, ROWS4(rows), dx, pos, row_sent, rows, drop(site + 1), w.ta);
    ST_TRY(c.lin_bwd(w.ta, s.h, y.w2, w.th, 0, (float*)gy.w2, (float*)gy.b2, rows, 256, dff));
    hipLaunchKernelGGL(ae_relu_drop_bwd_kernel, ELEMS((long)rows * dff), w.th, s.h, (long)rows * dff, scale);
    ST_TRY(c.lin_bwd(w.th, s.n2, y.w1, w.tb, 0, (float*)gy.w1, (float*)gy.b1, rows, dff, 256));
    ST_TRY(c.ln_bwd(w.tb, xm, y.ln2_a, rows, dx, 1, (float*)gy.ln2_a, (float*)gy.ln2_b));
    if (dec) {
      // source attention: x2 = x1 + drop(u Wo^T + bo), u = c[sentence] * head factor; no gradient reaches q / k / norm
      hipLaunchKernelGGL(ae_resid_bwd_kernel, ROWS4(rows), dx, pos, row_sent, rows, drop(site0 + 3), w.ta);
      ST_TRY(c.lin_bwd(w.ta, s.u, y.xo_w, w.tb, 0, (float*)gy.xo_w, (float*)gy.xo_b, rows, 256, 256));
      hipLaunchKernelGGL(ae_src_gather_bwd_kernel, dim3(n_sent), dim3(256), 0, c.st, w.tb, pos, off, n_sent, drop(site0 + 2), w.dcc);
      ST_TRY(c.lin_bwd(w.dcc, latent_ori, y.xv_w, dlat_ori, 1, (float*)gy.xv_w, (float*)gy.xv_b, n_sent, 256, 256));
    }
    // self-attention sublayer
    hipLaunchKernelGGL(ae_resid_bwd_kernel, ROWS4(rows), dx, pos, row_sent, rows, drop(site0 + 1), w.ta);
    ST_TRY(c.lin_bwd(w.ta, s.att, y.wo, w.tb, 0, (float*)gy.wo, (float*)gy.bo, rows, 256, 256));
    if (att16)
      hipLaunchKernelGGL(ae_attn16_bwd_kernel, dim3(n_sent), dim3(256), 0, c.st, s.q, s.k, s.v, s.P, w.tb, off, dec ? 1 : 0, drop(site0 + 0),
                         w.ta, w.tc, w.td);
    hipLaunchKernelGGL(ae_attn_bwd_kernel, dim3(n_sent, 4), dim3(64), att_bwd_lds, c.st, s.q, s.k, s.v, s.P, w.tb, off, dec ? 1 : 0,
                       drop(site0 + 0), w.ta, w.tc, w.td, att16 ? 16 : 0);
    ST_LAUNCH_CHECK();
    ST_TRY(c.lin_bwd(w.ta, s.n0, y.wq, w.tb, 0, (float*)gy.wq, (float*)gy.bq, rows, 256, 256));
    ST_TRY(c.lin_bwd(w.tc, s.n0, y.wk, w.tb, 1, (float*)gy.wk, (float*)gy.bk, rows, 256, 256));
    ST_TRY(c.lin_bwd(w.td, s.n0, y.wv, w.tb, 1, (float*)gy.wv, (float*)gy.bv, rows, 256, 256));
    ST_TRY(c.ln_bwd(w.tb, s.x_in, y.ln0_a, rows, dx, 1, (float*)gy.ln0_a, (float*)gy.ln0_b));
    ST_LAUNCH_CHECK();
    return ST_OK;
  };

  // ---------------- encoder forward over all (1 + views) * B sentences ----------------
  hipLaunchKernelGGL(ae_embed_kernel, ROWS4(Re), p->emb, p->pe, enc_tokens, enc_pos, w.e_row_sent, Re, V, drop(1), w.e_x);
  for (int l = 0; l < L; ++l)
    ST_TRY(layer_fwd(p->enc[l], w.enc[l], w.e_x, w.e_x, Re, Se, enc_off, enc_pos, w.e_row_sent, false, 10 + 4 * l, nullptr));
  hipLaunchKernelGGL(ae_ln_fwd_kernel, ROWS4(Re), w.e_x, p->enc_norm_a, p->enc_norm_b, Re, w.e_y);
  hipLaunchKernelGGL(ae_latent_kernel, dim3(Se), dim3(256), 0, c.st, w.e_y, enc_off, Se, w.lat);
  ST_LAUNCH_CHECK();
  // ---------------- decoder forward (teacher forced) on the origins' latents ----------------
  hipLaunchKernelGGL(ae_embed_kernel, ROWS4(Rd), p->emb, p->pe, dec_tokens, dec_pos, w.d_row_sent, Rd, V, drop(2), w.d_x);
  for (int l = 0; l < L; ++l) {
    ST_TRY(c.lin(w.lat, p->dec[l].xv_w, p->dec[l].xv_b, w.cc[l], B, 256, 256));
    ST_TRY(layer_fwd(p->dec[l], w.dec[l], w.d_x, w.d_x, Rd, B, dec_off, dec_pos, w.d_row_sent, true, 100 + 6 * l, w.cc[l]));
  }
  hipLaunchKernelGGL(ae_ln_fwd_kernel, ROWS4(Rd), w.d_x, p->dec_norm_a, p->dec_norm_b, Rd, w.d_xn);
  ST_TRY(c.lin(w.d_xn, p->gen_w, p->gen_b, w.logits, Rd, V, 256));
  // every packed decoder row has a non-PAD target (tokens, then EOS): ntokens = Rd  (siamese_contrastive.py:1040-1041)
  hipLaunchKernelGGL(ae_xent_kernel, dim3(Rd), dim3(256), 0, c.st, w.logits, dec_targets, Rd, V, 0, smoothing, 1.0f / (float)Rd, w.row_loss);
  hipLaunchKernelGGL(ae_sum_kernel, dim3(1), dim3(256), 0, c.st, w.row_loss, Rd, 1.0f / (float)Rd, losses_out + 0);
  ST_LAUNCH_CHECK();
  // ---------------- constraint: the (views) B x B similarity blocks ----------------
  ST_HIP(hipMemsetAsync(w.dlat, 0, (size_t)Se * 1024, c.st));
  hipLaunchKernelGGL(ae_fill_kernel, ELEMS((long)B), w.sl_g, (long)B, bata / ((float)B * (float)n_views));
  const bool views_at_once = g_ae_views && sl_small_views_ok(B) && (size_t)n_views * B * 257 <= w.slab_floats;
  if (views_at_once) {  // (the slab of the split contractions is free here: d_ori per view, then the views' log-sum-exps)
    float* parts = w.slab;
    float* lse_v = w.slab + (size_t)n_views * B * 256;
    ST_TRY(sl_small_views(w.lat, w.lat + (size_t)B * 256, n_views, B, tau, w.con_rows, lse_v, w.sl_g, parts, w.dlat + (size_t)B * 256, c.st));
    hipLaunchKernelGGL(ae_add_views_kernel, ELEMS((long)B * 256), w.dlat, (const float*)parts, n_views, (long)B * 256);
    ST_LAUNCH_CHECK();
  }
  for (int v = 0; v < n_views && !views_at_once; ++v) {
    const float* like = w.lat + (size_t)(1 + v) * B * 256;
    float* dlike = w.dlat + (size_t)(1 + v) * B * 256;
    ST_TRY(st_simloss_fwd(w.lat, like, B, B, tau, 0, w.con_rows + (size_t)v * B, w.sl_lse, w.sl_scratch, w.sl_scratch_bytes, c.st));
    ST_TRY(st_simloss_bwd(w.lat, like, B, B, tau, 0, w.sl_lse, w.sl_g, w.sl_dori, dlike, w.sl_scratch, w.sl_scratch_bytes, c.st));
    hipLaunchKernelGGL(ae_add_kernel, ELEMS((long)B * 256), w.dlat, w.sl_dori, (long)B * 256);
    ST_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(ae_sum_kernel, dim3(1), dim3(256), 0, c.st, w.con_rows, n_views * B, 1.0f / ((float)B * (float)n_views), losses_out + 1);
  ST_LAUNCH_CHECK();
  // ---------------- decoder backward ----------------
  {
    float* dx = w.d_x;  // the residual stream buffer becomes the running gradient (its value moves to td first)
    ST_TRY(c.lin_bwd(w.logits, w.d_xn, p->gen_w, w.tb, 0, (float*)g->gen_w, (float*)g->gen_b, Rd, V, 256));
    // the final norm's input is still needed by its backward: keep it in td
    ST_HIP(hipMemcpyAsync(w.td, w.d_x, (size_t)Rd * 1024, hipMemcpyDeviceToDevice, c.st));
    ST_TRY(c.ln_bwd(w.tb, w.td, p->dec_norm_a, Rd, dx, 0, (float*)g->dec_norm_a, (float*)g->dec_norm_b));
    ST_LAUNCH_CHECK();
    for (int l = L - 1; l >= 0; --l)
      ST_TRY(layer_bwd(p->dec[l], g->dec[l], w.dec[l], dx, Rd, B, dec_off, dec_pos, w.d_row_sent, true, 100 + 6 * l, w.lat, w.dlat));
    ST_HIP(hipMemsetAsync(w.emb_acc, 0, (size_t)V * 256 * 8, c.st));
    hipLaunchKernelGGL(ae_embed_bwd_kernel, ROWS4(Rd), dx, dec_tokens, dec_pos, w.d_row_sent, Rd, V, drop(2), w.emb_acc);
    ST_LAUNCH_CHECK();
  }
  // ---------------- encoder backward ----------------
  {
    float* dx = w.e_x;
    ST_HIP(hipMemcpyAsync(w.td, w.e_x, (size_t)Re * 1024, hipMemcpyDeviceToDevice, c.st));
    hipLaunchKernelGGL(ae_latent_bwd_kernel, dim3(Se), dim3(256), 0, c.st, w.e_y, w.dlat, enc_off, Se, w.tb);
    ST_TRY(c.ln_bwd(w.tb, w.td, p->enc_norm_a, Re, dx, 0, (float*)g->enc_norm_a, (float*)g->enc_norm_b));
    ST_LAUNCH_CHECK();
    for (int l = L - 1; l >= 0; --l)
      ST_TRY(layer_bwd(p->enc[l], g->enc[l], w.enc[l], dx, Re, Se, enc_off, enc_pos, w.e_row_sent, false, 10 + 4 * l, nullptr, nullptr));
    hipLaunchKernelGGL(ae_embed_bwd_kernel, ROWS4(Re), dx, enc_tokens, enc_pos, w.e_row_sent, Re, V, drop(1), w.emb_acc);
    hipLaunchKernelGGL(ae_embed_finish_kernel, ELEMS((long)V * 256), (const unsigned long long*)w.emb_acc, (long)V * 256, (float*)g->emb);
    ST_LAUNCH_CHECK();
  }
  return ST_OK;
}
