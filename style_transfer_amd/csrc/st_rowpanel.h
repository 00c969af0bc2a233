// Row-panel f32-MFMA GEMM, barrier-free main loop:  C[n,N] = op(X[n,256]) . W[N,256]^T (+ epilogue)
//
// One 256-thread workgroup owns a 64-row panel of X, resident in LDS (optionally LayerNorm'ed in
// place); LDS = 69 KiB so two workgroups share a CU.  Each of the 4 waves walks its OWN 32-column
// blocks of W (blocks wave, wave+4, ... of the workgroup's column range) and never shares W data with
// another wave, so W needs no LDS staging and the main loop needs no barrier: B fragments are loaded
// straight from global memory (L2-resident weights) into VGPRs through an 8-deep register ring.
// W is pre-packed by st_pack_weight into fragment order [block][kg][lane][4] so that every wave load
// is one fully coalesced 1 KiB read:
//     packed[((blk*(K/8) + kg)*64 + lane)*4 + j] = W[blk*32 + (lane&31)][kg*8 + 4*(lane>>5) + j]
// (the lane<->k permutation of st_gemm.h).  A fragments come from the LDS panel (conflict-free b128).
//
// H = true (gemm_format ST_GEMM_F16X3): the same kernels on v_mfma_f32_32x32x16_f16 with both operands
// split into f16 hi + lo parts and three products per step (hi.hi + hi.lo + lo.hi, st_common.h):
// f32-grade results (~2^-22 relative per term) at up to 16/3 of the f32 MFMA rate.  The LDS panel row
// keeps its 260-dword stride: dwords [0,128) hold the 256 hi halves, [128,256) the lo halves, so the
// b128 fragment reads (8 consecutive k per lane) have the bank pattern of the f32 layout.  Packed W:
//     [block][step = k/16][hi|lo][lane][8 halves],  half j of lane = W[blk*32 + (lane&31)][16*step + 8*(lane>>5) + j]
// (same size in bytes as the f32 image).
#pragma once
#include "st_gemm.h"

#include <stdlib.h>
#include <algorithm>
#include <type_traits>

// streaming outputs (written once, read by a later kernel): nontemporal stores keep them from evicting
// the L2-resident weights
#if defined(ST_ABL_NOSTORE)  /* kbench ablation: keep the value live, never store */
#define ST_STORE(ptr, v) do { if ((v) == 1.2345e-30f) *(ptr) = (v); } while (0)
#elif defined(ST_NO_NT)
#define ST_STORE(ptr, v) (*(ptr) = (v))
#else
#define ST_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))
#endif
#if defined(ST_NO_NT)
#define ST_STORE4(ptr, v) (*(ptr) = (v))
#else
#define ST_STORE4(ptr, v) __builtin_nontemporal_store((v), (ptr))
#endif

#define RP_LDS ((64 * ST_LDA + 8 * 64 * 4) * 4)

// Kernel-form switches, ONE copy per library (defined in st_decode.hip; tools/kbench.cpp defines its own and flips them
// for its A/B runs).  For same-box whole-bench comparisons (tools/ab_*.sh) they can be set from the environment, but only
// when ST_AB_ENABLE is set as well, so that a stray ST_AB_* variable in a user's environment changes nothing; read once
// per process (thread-safe static initialisation).  INTEGRATION.md lists them.  Never set in production.
#ifdef ST_RP_DEFINE_SWITCHES
#define ST_RP_SWITCH(name, value) int name = value
#else
#define ST_RP_SWITCH(name, value) extern int name
#endif
ST_RP_SWITCH(rp_lin_shape16, 1);      // ST_AB_LIN32: the 32x32x16 QKV kernel
ST_RP_SWITCH(rp_screen_shape16, 1);   // ST_AB_SCREEN32: the 32x32x16 screening pass
ST_RP_SWITCH(rp_screen_paired, 1);    // ST_AB_SCREEN_UNPAIRED: one key per logit in the 16x16x32 screening pass
ST_RP_SWITCH(rp_screen_maxnorm, 0);   // ST_AB_SCREEN_MAXNORM: one bound for all columns instead of per-column norms
ST_RP_SWITCH(rp_argmax_shape16, 1);   // ST_AB_ARGMAX32: the 32x32x16 exact argmax
ST_RP_SWITCH(rp_tail_persistent, 1);  // ST_AB_TAIL_ONE_PANEL: one workgroup per panel instead of the persistent loop
ST_RP_SWITCH(rp_tail_shape16, 1);     // ST_AB_TAIL32: the round-1 32x32x16 half-layer kernel
// Below this many live rows the unfused kernels (column splits) are used instead of the half-layer kernel.  Round 3: 64 (was
// 4096).  With a few thousand rows the unfused chain (O-projection 15 us + FFN1 38 us + FFN2 63 us per layer at 3328 rows)
// is latency-bound on every workgroup streaming a whole weight matrix; one panel per CU through the half-layer kernel takes
// ~45 us whatever the row count (batch-256 regime 12.9 k -> 18.3 k sentences/s, profiles/r03_small_batch.txt).
ST_RP_SWITCH(rp_tail_min_rows, 64);   // ST_AB_TAIL_MIN_ROWS=<n>
ST_RP_SWITCH(rp_enc_unfused, 0);      // ST_AB_ENC_UNFUSED: encoder QKV and attention as two kernels
ST_RP_SWITCH(rp_dec_fused, 0);        // ST_AB_DEC_FUSED: decode QKV + attention in one kernel (measured negative, st_decode.hip)
ST_RP_SWITCH(rp_no_emb_in_tail, 0);   // ST_AB_EMBED_KERNEL: the decode step's embed kernel writes x for the first layer's half-layer kernel
ST_RP_SWITCH(rp_kv_pos_major, 0);     // ST_AB_KV_POS_MAJOR: K / V cache as [layer][k|v][position][job] instead of [job][layer][k|v][position] (measured: no difference)
ST_RP_SWITCH(rp_enc_tab_gather, 0);   // ST_AB_ENC_TAB_GATHER: unfused encoder, first layer: the table rows are gathered into the qkv buffer first
#undef ST_RP_SWITCH
static inline void rp_ab_env() {
  static const bool done = [] {
    if (!getenv("ST_AB_ENABLE")) return true;
    if (getenv("ST_AB_TAIL32")) rp_tail_shape16 = 0;
    if (getenv("ST_AB_TAIL_ONE_PANEL")) rp_tail_persistent = 0;
    if (getenv("ST_AB_LIN32")) rp_lin_shape16 = 0;
    if (getenv("ST_AB_SCREEN32")) rp_screen_shape16 = 0;
    if (getenv("ST_AB_SCREEN_UNPAIRED")) rp_screen_paired = 0;
    if (getenv("ST_AB_SCREEN_MAXNORM")) rp_screen_maxnorm = 1;
    if (getenv("ST_AB_ARGMAX32")) rp_argmax_shape16 = 0;
    if (getenv("ST_AB_TAIL_MIN_ROWS")) rp_tail_min_rows = atoi(getenv("ST_AB_TAIL_MIN_ROWS"));
    if (getenv("ST_AB_ENC_UNFUSED")) rp_enc_unfused = 1;
    if (getenv("ST_AB_KV_POS_MAJOR")) rp_kv_pos_major = 1;
    if (getenv("ST_AB_ENC_TAB_GATHER")) rp_enc_tab_gather = 1;
    if (getenv("ST_AB_DEC_FUSED")) rp_dec_fused = 1;
    if (getenv("ST_AB_EMBED_KERNEL")) rp_no_emb_in_tail = 1;
    return true;
  }();
  (void)done;
}


// store 4 consecutive features of a panel row (feature index c = 4*c4) in the kernel's operand format.
// f16x3: a value outside the f16 range (or NaN) would turn into inf - inf = NaN inside the products and come
// out as a silent garbage argmax, so it raises *flag instead (checked by the host entry point: ST_ERANGE).
template <bool H>
__device__ __forceinline__ void panel_store4(float* row, int c4, f32x4 v, int* flag) {
  if (H) {
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    if (flag && !(m <= 65504.0f)) *flag = 1;
    h4 hi, lo;
    split_f16(v, hi, lo);
    _Float16* ph = (_Float16*)row;
    *(h4*)(ph + c4 * 4) = hi;
    *(h4*)(ph + 256 + c4 * 4) = lo;
  } else {
    *(f32x4*)(row + c4 * 4) = v;
  }
}

// The same (f16x3 only) without the branch: the range violation is OR-ed into `bad`, which the caller reports once at its
// end.  A branch per call keeps the scheduler from interleaving the independent rows of an unrolled loop.
__device__ __forceinline__ void panel_store4_nb(float* row, int c4, f32x4 v, int& bad) {
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  bad |= !(m <= 65504.0f);
  h4 hi, lo;
  split_f16(v, hi, lo);
  _Float16* ph = (_Float16*)row;
  *(h4*)(ph + c4 * 4) = hi;
  *(h4*)(ph + 256 + c4 * 4) = lo;
}

// panel_store4_nb for the K-permuted rows of the 16x16x32 kernels (rp16_quad_pos, defined with the screening kernels below)
__device__ __forceinline__ int rp16_frag_pos(int d, int g);
__device__ __forceinline__ int rp16_quad_pos(int c4);
__device__ __forceinline__ void panel_store4_nb16(float* row, int c4, f32x4 v, int& bad) {
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  bad |= !(m <= 65504.0f);
  h4 hi, lo;
  split_f16(v, hi, lo);
  *(h4*)(row + rp16_quad_pos(c4)) = hi;
  *(h4*)(row + 128 + rp16_quad_pos(c4)) = lo;
}


template <int EPI, int NW = 4, bool H = false>
__global__ __launch_bounds__(64 * NW, NW / 2) void rp_gemm_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                  // [64][ST_LDA]
  float* red = smem + 64 * ST_LDA;   // [NW waves][64 rows][4]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the W address math on the scalar unit
  const int half = lane >> 5, l31 = lane & 31;
  auto xrow = [&](int r) -> int { return a.xrows ? a.xrows[r] : r; };  // optional row gather (ARGMAX redo pass)

  // ---- X panel -> LDS (64 rows x 64 float4, 16 per thread), optional LayerNorm ----
  if (a.ln_a) {
#pragma unroll
    for (int i = 0; i < 64 / NW; ++i) {
      const int f = tid + i * 64 * NW;
      const int r = f >> 6, c4 = f & 63;
      *(f32x4*)(As + r * ST_LDA + c4 * 4) = *(const f32x4*)(a.X + (size_t)xrow(min(m0 + r, n - 1)) * 256 + c4 * 4);
    }
    __syncthreads();
    const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
    const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
#pragma unroll 4
    for (int rr = 0; rr < 64 / NW; ++rr) {
      float* row = As + (wave * (64 / NW) + rr) * ST_LDA;  // one wave owns the whole row: in-place is safe
      const f32x4 y = H ? layernorm256r(*(f32x4*)(row + lane * 4), ga, gb) : layernorm256(*(f32x4*)(row + lane * 4), ga, gb);
      panel_store4<H>(row, lane, y, a.range_flag);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 64 / NW; ++i) {
      const int f = tid + i * 64 * NW;
      const int r = f >> 6, c4 = f & 63;
      panel_store4<H>(As + r * ST_LDA, c4, *(const f32x4*)(a.X + (size_t)xrow(min(m0 + r, n - 1)) * 256 + c4 * 4), a.range_flag);
    }
  }
  __syncthreads();

  Top2 best[2][16];
  Lse run[2][16];
  float tgt[2][16];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (EPI == EPI_ARGMAX) {
        best[mi][r].m1 = -INFINITY;
        best[mi][r].i1 = 0x7fffffff;
        best[mi][r].m2 = -INFINITY;
      }
      if (EPI == EPI_LSE) {
        run[mi][r].m = -INFINITY;
        run[mi][r].l = 0.0f;
        tgt[mi][r] = 0.0f;
      }
    }

  // this workgroup's column blocks [b0, b1), this wave's: b0 + wave + NW*i
  constexpr int SPB = H ? 16 : 32;              // ring steps per 32-column block (K = 256)
  constexpr int R = 8;                          // ring depth in steps
  const int nblk = (a.N + 31) >> 5;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per;
  const int b1 = min(nblk, b0 + per);
  const int cnt = (b1 - b0 > wave) ? (b1 - b0 - wave + NW - 1) / NW : 0;
  const int total = cnt * SPB;  // (block, k-step) steps
  const f32x4* Wl = (const f32x4*)a.W + lane;
  // f32: one 1 KiB fragment per step; f16x3: a hi and a lo fragment (p = 0, 1)
  auto wload = [&](int g, int p) -> f32x4 {
    g = min(g, total - 1);
    const int blk = b0 + wave + NW * (g / SPB);
    return H ? Wl[(((size_t)blk * SPB + (g & (SPB - 1))) * 2 + p) * 64] : Wl[((size_t)blk * SPB + (g & (SPB - 1))) * 64];
  };
  f32x4 ring[R], ringl[H ? R : 1];
  if (total > 0) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      ring[j] = wload(j, 0);
      if (H) ringl[j] = wload(j, 1);
    }
  }
  const float* Ab = As + l31 * ST_LDA + 4 * half;
  // accumulators start from the column's bias (fetched one block ahead) instead of zero
  auto bias_of = [&](int i) -> float {
    const int col = ((b0 + wave + NW * i) << 5) + l31;
    if (EPI == EPI_ARGMAX && col >= a.N) return -INFINITY;  // padding columns of the last block never win
    return (a.bias && col < a.N) ? a.bias[col] : 0.0f;
  };
  float bnext = cnt > 0 ? bias_of(0) : 0.0f;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bnext;
  float pre[2][16];  // RESID: residual (+ cross-attention constant), fetched while the block's MFMAs run
  f32x4 an[H ? 4 : 1];
  if (H) {
    an[0] = *(const f32x4*)(Ab);
    an[2] = *(const f32x4*)(Ab + 32 * ST_LDA);
    an[1] = *(const f32x4*)(Ab + 128);
    an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128);
  }
  for (int g0 = 0; g0 < total; g0 += R) {
    if ((g0 & (SPB - 1)) == 0) bnext = bias_of(g0 / SPB + 1);
    if (EPI == EPI_RESID && (g0 & (SPB - 1)) == 0) {
      const int col = min(((b0 + wave + NW * (g0 / SPB)) << 5) + l31, a.N - 1);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(m0 + mi * 32 + mfma32_row(r, lane), n - 1);
          float v = a.resid[(size_t)row * a.ldo + col];
          if (a.cconst) v += a.cconst[(size_t)a.idx[row] * a.ld_cc + col];
          pre[mi][r] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int kg = (g0 + j) & (SPB - 1);
      if (H) {
        const h8 bh = as_h8(ring[j]), bl = as_h8(ringl[j]);
        ring[j] = wload(g0 + j + R, 0);
        ringl[j] = wload(g0 + j + R, 1);
        // A fragments of the NEXT step are requested before this step's MFMAs (LDS latency ~ one step)
        const h8 a0h = as_h8(an[0]), a0l = as_h8(an[1]), a1h = as_h8(an[2]), a1l = as_h8(an[3]);
        const int kn = (kg + 1) & (SPB - 1);
        an[0] = *(const f32x4*)(Ab + kn * 8);
        an[1] = *(const f32x4*)(Ab + 128 + kn * 8);
        an[2] = *(const f32x4*)(Ab + 32 * ST_LDA + kn * 8);
        an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128 + kn * 8);
        acc0 = mfma16h(a0h, bh, acc0);
        acc1 = mfma16h(a1h, bh, acc1);
        acc0 = mfma16h(a0h, bl, acc0);
        acc1 = mfma16h(a1h, bl, acc1);
        acc0 = mfma16h(a0l, bh, acc0);
        acc1 = mfma16h(a1l, bh, acc1);
        __builtin_amdgcn_sched_barrier(0);  // step structure kept (as in the half-layer kernel): 1-3 % on FFN1 / QKV / generator
      } else {
        const f32x4 b = ring[j];
        ring[j] = wload(g0 + j + R, 0);
        const f32x4 a0 = *(const f32x4*)(Ab + kg * 8);
        const f32x4 a1 = *(const f32x4*)(Ab + 32 * ST_LDA + kg * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = mfma32(a0[s], b[s], acc0);
          acc1 = mfma32(a1[s], b[s], acc1);
        }
      }
    }
    if (((g0 + R) & (SPB - 1)) == 0) {
      // ---- epilogue for this wave's column block ----
      const int col = ((b0 + wave + NW * (g0 / SPB)) << 5) + l31;
      const bool colok = col < a.N;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x16 acc = mi ? acc1 : acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + mi * 32 + mfma32_row(r, lane);
          float v = acc[r];
          if (EPI == EPI_ARGMAX) {
            // columns >= N start from a -inf bias (bias_of), so they never win; 4 VALU ops per logit:
            // second = median(best, v, second) because second <= best always holds
            Top2& t = best[mi][r];
            t.i1 = v > t.m1 ? col : t.i1;
            t.m2 = __builtin_amdgcn_fmed3f(t.m1, v, t.m2);
            t.m1 = fmaxf(t.m1, v);
          } else if (EPI == EPI_LSE) {
            if (colok) {
              // online log-sum-exp with ONE exponential per logit: e = exp(-|s - m|) rescales the sum (s > m) or is the
              // new term (s <= m)
              const float sv = v * a.scale;
              Lse& u = run[mi][r];
              const float d = sv - u.m;
              const float e = H ? __expf(-fabsf(d)) : expf(-fabsf(d));
              u.l = d > 0.0f ? fmaf(u.l, e, 1.0f) : u.l + e;
              u.m = fmaxf(u.m, sv);
              if (col == a.row_offset + row) tgt[mi][r] = sv;
            }
          } else if (row < n && colok) {
            if (EPI == EPI_BIAS) {
              ST_STORE(&a.out[(size_t)row * a.ldo + col], v);
            } else if (EPI == EPI_RELU) {
              ST_STORE(&a.out[(size_t)row * a.ldo + col], fmaxf(v, 0.0f));
            } else if (EPI == EPI_RESID) {
              ST_STORE(&a.out[(size_t)row * a.ldo + col], v + pre[mi][r]);
            } else if (EPI == EPI_DS) {
              const float sv = v * a.scale;
              const float x = sv - a.lse[row];
              const float pr = (H ? __expf(x) : expf(x)) - (col == a.row_offset + row ? 1.0f : 0.0f);
              a.out[(size_t)row * a.ldo + col] = a.g ? a.g[row] * pr * a.scale : pr;
            } else if (EPI == EPI_DST) {
              const float x = v * a.scale - a.lse[col];
              a.out[(size_t)row * a.ldo + col] = (H ? __expf(x) : expf(x)) - (row == a.row_offset + col ? 1.0f : 0.0f);
            } else if (EPI == EPI_QKV) {
              if (col < 256) {
                ST_STORE(&a.q[(size_t)row * 256 + col], v);
              } else {
                const long s = a.idx[row];
                float* base = (col < 512) ? a.kbase : a.vbase;
                ST_STORE(&base[s * a.kv_stride + ((col - 256) & 255)], v);
              }
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bnext;
    }
  }

  if (EPI == EPI_ARGMAX || EPI == EPI_LSE) {
    // reduce each (mi, reg) row over the 32 lanes of its half, then over the 4 waves through LDS
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* p = red + (wave * 64 + mi * 32 + mfma32_row(r, lane)) * 4;
        if (EPI == EPI_ARGMAX) {
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
            p[0] = t.m1;
            p[1] = __int_as_float(t.i1);
            p[2] = t.m2;
          }
        } else {
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
            p[0] = t.m;
            p[1] = t.l;
            p[2] = tg;
          }
        }
      }
    __syncthreads();
    if (tid < 64 && m0 + tid < n) {
      if (EPI == EPI_ARGMAX) {
        Top2 t = {red[tid * 4 + 0], __float_as_int(red[tid * 4 + 1]), red[tid * 4 + 2]};
        for (int w = 1; w < NW; ++w) {
          Top2 u = {red[(w * 64 + tid) * 4 + 0], __float_as_int(red[(w * 64 + tid) * 4 + 1]), red[(w * 64 + tid) * 4 + 2]};
          t = top2_merge(t, u);
        }
        a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] =
            make_float4(t.m1, __int_as_float(t.i1), t.m2, 0.0f);
      } else {
        Lse t = {red[tid * 4 + 0], red[tid * 4 + 1]};
        float tg = red[tid * 4 + 2];
        for (int w = 1; w < NW; ++w) {
          Lse u = {red[(w * 64 + tid) * 4 + 0], red[(w * 64 + tid) * 4 + 1]};
          t = lse_merge(t, u);
          tg += red[(w * 64 + tid) * 4 + 2];
        }
        a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m, t.l, tg, 0.0f);
      }
    }
  }
}

// number of column splits of a launch: maximise (fill of the last round of 512 resident workgroups) x
// (balance of 32-column blocks over the waves: a workgroup lasts as long as its busiest wave) x
// (amortisation of the ~0.25-block panel-load/LayerNorm prologue).
static inline int rp_splits(int n_rows, int N, int max_splits, int nw = 4) {
  const int panels = (n_rows + 63) / 64 > 0 ? (n_rows + 63) / 64 : 1;
  const int nblk = (N + 31) / 32;
  int cap = (nblk + nw - 1) / nw;
  if (cap > max_splits) cap = max_splits;
  int best = 1;
  double best_score = 0.0;
  for (int s = 1; s <= cap; ++s) {
    const int per = (nblk + s - 1) / s;           // blocks per workgroup
    if ((s - 1) * per >= nblk) continue;          // last split would be empty
    const int busiest = (per + nw - 1) / nw;      // blocks of the busiest wave
    const long wgs = (long)panels * s;
    const long rounds = (wgs + 511) / 512;
    const double score = ((double)wgs / (rounds * 512.0)) * ((double)nblk / ((double)s * nw * busiest)) *
                         ((double)busiest / (busiest + 0.25));
    if (score > best_score * 1.01) {
      best_score = score;
      best = s;
    }
  }
  return best;
}

template <int EPI, int NW, bool H>
static int launch_rp_t(const GemmArgs& a, int n_bound, int splits, hipStream_t st) {
  ST_ENSURE_LDS((rp_gemm_kernel<EPI, NW, H>), RP_LDS);
  if (n_bound <= 0) return ST_OK;
  dim3 grid((n_bound + 63) / 64, splits);
  hipLaunchKernelGGL((rp_gemm_kernel<EPI, NW, H>), grid, dim3(64 * NW), RP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
// ------------------------------------------------------------------------------------------------
// Screening pass, software-pipelined form (the one st_decode.hip uses; grid = panels x column splits).
// Ablations of the first form of this pass (the generic row-panel kernel with one product per step and a running
// top-2 per logit; kbench, 229 k rows): without the top-2 bookkeeping the pass takes 38 % less
// time, without the W loads 17 %, without the LDS reads 6 % - the 4 VALU operations per logit do not overlap with
// the MFMAs, because a wave's bookkeeping for a block can only start when its last MFMA is done and there are just
// two waves per SIMD.  Here the bookkeeping of block i-1 is interleaved, in one basic block, with the 32 MFMAs of
// block i (two accumulator sets), and costs 3 operations per logit: the wave-local block number is written into
// the low 8 mantissa bits of the logit (key = (v & ~0xff) | i), so the running (best, second) pair is one v_med3
// and one v_max on keys and the winning column comes out of the key.  Keys are at most 2^-15 |v| below |v|; the
// caller's margin test (finish_kernel) carries a term for that bound.
// Exact ties of keys give a zero gap: such rows are re-done by the 3-product kernel like every ambiguous row.
// 64-row panel, hi halves only in LDS (33 KiB), 4 waves, no barrier after the prologue.
// ------------------------------------------------------------------------------------------------
#define RS_LDA 132  /* dwords per panel row: 256 hi halves + 4 dwords of padding (= 4 mod 64: conflict-free b128) */
#define RS_LDS ((64 * RS_LDA + 4 * 64 * 4 + 64) * 4)
#define RS_MAX_BLOCKS 256  /* column blocks per wave that fit the 8 key bits */
#define RS_PAD_LOGIT (-3.0e38f) /* bias of padding columns: finite, so that its key is not a NaN */
template <int UNUSED = 0>
__global__ __launch_bounds__(256, 2) void rp_screen_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                  // [64][RS_LDA]
  float* red = smem + 64 * RS_LDA;   // [4 waves][64 rows][4]
  float* rownorm = red + 4 * 64 * 4; // [64]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  {
    // one wave per row, 16 rows each: global -> registers -> LayerNorm -> hi halves in LDS
    f32x4 ga = {1, 1, 1, 1}, gb = {0, 0, 0, 0};
    if (a.ln_a) {
      ga = *(const f32x4*)(a.ln_a + lane * 4);
      gb = *(const f32x4*)(a.ln_b + lane * 4);
    }
    f32x4 v[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int r = min(m0 + wave * 16 + rr, n - 1);
      v[rr] = *(const f32x4*)(a.X + (size_t)(a.xrows ? a.xrows[r] : r) * 256 + lane * 4);
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int r = wave * 16 + rr;
      f32x4 y = v[rr];
      float nrm = INFINITY;  // no LayerNorm, no norm: every row counts as ambiguous
      if (a.ln_a) {
        y = layernorm256r(y, ga, gb);
        nrm = sqrtf(wave_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w));
      }
      if (lane == 0) rownorm[r] = nrm;
      const float m = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)));
      if (a.range_flag && !(m <= 65504.0f)) *a.range_flag = 1;
      h4 hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) hi[j] = (_Float16)y[j];
      *(h4*)((_Float16*)(As + r * RS_LDA) + lane * 4) = hi;
    }
  }
  __syncthreads();

  // running (best, second) keys per (row tile, register): start below every finite logit
  float k1[2][16], k2[2][16];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) k1[mi][r] = k2[mi][r] = -INFINITY;
  constexpr int SPB = 16, R = 8;
  const int nblk = (a.N + 31) >> 5;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;  // column split blockIdx.y covers blocks [b0, b1)
  const int b0 = blockIdx.y * per;
  const int b1 = min(nblk, b0 + per);
  const int cnt = (b1 - b0 > wave) ? (b1 - b0 - wave + 3) / 4 : 0;  // this wave's blocks: b0 + wave, b0 + wave + 4, ...
  const f32x4* Wl = (const f32x4*)a.W + lane;
  const float* Ab = As + l31 * RS_LDA + 4 * half;
  auto wbase = [&](int i) -> const f32x4* {  // hi fragment of step g of block i at wbase(i)[g * 128]
    return Wl + (size_t)(b0 + wave + 4 * min(i, cnt - 1)) * SPB * 2 * 64;
  };
  auto bias_of = [&](int i) -> float {
    const int col = ((b0 + wave + 4 * i) << 5) + l31;
    return col < a.N ? (a.bias ? a.bias[col] : 0.0f) : RS_PAD_LOGIT;
  };
  auto book = [&](float v, int i, float& b1, float& b2) {  // 3 VALU operations
    const float key = __int_as_float((__float_as_int(v) & ~0xff) | i);
    b2 = __builtin_amdgcn_fmed3f(b1, key, b2);
    b1 = fmaxf(b1, key);
  };
  f32x4 ring[R];
  if (cnt > 0) {
    const f32x4* W0 = wbase(0);
#pragma unroll
    for (int j = 0; j < R; ++j) ring[j] = W0[(size_t)j * 128];
  }
  // one block: 16 k-steps into (n0, n1); meanwhile the previous block's accumulators (o0, o1; block number io) are
  // booked, two logits per k-step.  BOOK = false for the first block.
  f32x4 an0, an1;  // A fragments of the NEXT k-step (requested one step ahead: LDS latency ~ one step)
  an0 = *(const f32x4*)(Ab);
  an1 = *(const f32x4*)(Ab + 32 * RS_LDA);
  auto block = [&](f32x16& n0, f32x16& n1, int i, const f32x16& o0, const f32x16& o1, int io, bool do_book) {
    const f32x4* Wb = wbase(i);
    const f32x4* Wn = wbase(i + 1);
    const float bv = bias_of(i);
#pragma unroll
    for (int r = 0; r < 16; ++r) n0[r] = n1[r] = bv;
#pragma unroll
    for (int kg = 0; kg < SPB; ++kg) {
      const h8 bh = as_h8(ring[kg % R]);
      const h8 a0h = as_h8(an0), a1h = as_h8(an1);
      ring[kg % R] = kg + R < SPB ? Wb[(size_t)(kg + R) * 128] : Wn[(size_t)(kg + R - SPB) * 128];
      an0 = *(const f32x4*)(Ab + ((kg + 1) & (SPB - 1)) * 8);
      an1 = *(const f32x4*)(Ab + 32 * RS_LDA + ((kg + 1) & (SPB - 1)) * 8);
      n0 = mfma16h(a0h, bh, n0);
      n1 = mfma16h(a1h, bh, n1);
      if (do_book) {
        book(o0[kg], io, k1[0][kg], k2[0][kg]);
        book(o1[kg], io, k1[1][kg], k2[1][kg]);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the step structure: nothing is hoisted across k-steps (register pressure)
    }
  };
  f32x16 A0, A1, B0, B1;
  if (cnt > 0) {
    block(A0, A1, 0, A0, A1, 0, false);
    int i = 1;
    for (; i + 1 < cnt; i += 2) {
      block(B0, B1, i, A0, A1, i - 1, true);
      block(A0, A1, i + 1, B0, B1, i, true);
    }
    if (i < cnt) {  // an even number of blocks: one more into B, then book B
      block(B0, B1, i, A0, A1, i - 1, true);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        book(B0[r], i, k1[0][r], k2[0][r]);
        book(B1[r], i, k1[1][r], k2[1][r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        book(A0[r], cnt - 1, k1[0][r], k2[0][r]);
        book(A1[r], cnt - 1, k1[1][r], k2[1][r]);
      }
    }
  }
  // keys -> (value, column, second); over the 32 lanes of a half, then over the 4 waves through LDS
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Top2 t;
      t.m1 = k1[mi][r];
      t.i1 = ((b0 + wave + 4 * (__float_as_int(k1[mi][r]) & 0xff)) << 5) + l31;
      t.m2 = k2[mi][r];
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
    Top2 t = {red[tid * 4 + 0], __float_as_int(red[tid * 4 + 1]), red[tid * 4 + 2]};
    for (int w = 1; w < 4; ++w) {
      Top2 u = {red[(w * 64 + tid) * 4 + 0], __float_as_int(red[(w * 64 + tid) * 4 + 1]), red[(w * 64 + tid) * 4 + 2]};
      t = top2_merge(t, u);
    }
    // keys sit up to 2^-15 |v| off their logits: finish_kernel's margin has a term for that
    a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m1, __int_as_float(t.i1), t.m2, rownorm[tid]);
  }
}
// ------------------------------------------------------------------------------------------------
// The screening pass on v_mfma_f32_16x16x32_f16 tiles with W as the A operand (see rp_layer_tail16_kernel for the lane
// addressing of the unchanged packed image; the panel keeps its hi-halves-only layout).  Lane (r16, g) holds the logits
// of vocabulary columns 32 blk + 16 ft + 4 g + i of panel row 16 rt + r16: a row's candidates live in 8 key slots of 4
// lanes, so the final top-2 merge is local plus two lane exchanges per row tile (the 32x32x16 form: five exchanges
// for each of 32 slots).
// ------------------------------------------------------------------------------------------------
template <int UNUSED = 0>
__global__ __launch_bounds__(256, 2) void rp_screen16_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                  // [64][RS_LDA]
  float* red = smem + 64 * RS_LDA;   // [4 waves][64 rows][4]
  float* rownorm = red + 4 * 64 * 4; // [64]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  {
    // one wave per row, 16 rows each: global -> registers -> LayerNorm -> hi halves in LDS
    f32x4 ga = {1, 1, 1, 1}, gb = {0, 0, 0, 0};
    if (a.ln_a) {
      ga = *(const f32x4*)(a.ln_a + lane * 4);
      gb = *(const f32x4*)(a.ln_b + lane * 4);
    }
    f32x4 v[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int r = min(m0 + wave * 16 + rr, n - 1);
      v[rr] = *(const f32x4*)(a.X + (size_t)(a.xrows ? a.xrows[r] : r) * 256 + lane * 4);
    }
    int bad = 0;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int r = wave * 16 + rr;
      f32x4 y = v[rr];
      float nrm = INFINITY;  // no LayerNorm, no norm: every row counts as ambiguous
      if (a.ln_a) {
        y = layernorm256r(y, ga, gb);
        nrm = sqrtf(wave_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w));
      }
      if (lane == 0) rownorm[r] = nrm;
      const float m = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)));
      bad |= !(m <= 65504.0f);
      h4 hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) hi[j] = (_Float16)y[j];
      *(h4*)((_Float16*)(As + r * RS_LDA) + lane * 4) = hi;
    }
    if (bad && a.range_flag) *a.range_flag = 1;
  }
  __syncthreads();

  // With column norms (a.wnorm) the keys are UPPER BOUNDS of the exact logits, u = s + cx * ||W_v|| with
  // cx = a.scale * ||x_row|| (added to the accumulators' initial value): finish_kernel then needs only the winner's own norm for its margin (a vocabulary's largest
  // row norm can be several times the typical one: amazon 2.87 against a median of 0.71).
  const bool ukeys = a.wnorm != nullptr && a.ln_a != nullptr;
  float cx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (ukeys) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) cx[rt] = a.scale * rownorm[16 * rt + r16];
  }
  // running (best, second) keys per (feature tile, row tile, register): start below every finite logit
  float k1[2][4][4], k2[2][4][4];
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) k1[ft][rt][q] = k2[ft][rt][q] = -INFINITY;
  const int nblk = (a.N + 31) >> 5;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;  // column split blockIdx.y covers blocks [b0, b1)
  const int b0 = blockIdx.y * per;
  const int b1 = min(nblk, b0 + per);
  const int cnt = (b1 - b0 > wave) ? (b1 - b0 - wave + 3) / 4 : 0;  // this wave's blocks: b0 + wave, b0 + wave + 4, ...
  const f32x4* Wl = (const f32x4*)a.W + (g >> 1) * 128 + 32 * (g & 1) + r16;
  const float* Ab = As + r16 * RS_LDA + 4 * g;
  auto wbase = [&](int i) -> const f32x4* { return Wl + (size_t)(b0 + wave + 4 * min(i, cnt - 1)) * 16 * 128; };
  auto book = [&](float v, int i, float& b1, float& b2) {  // 3 VALU operations
    const float key = __int_as_float((__float_as_int(v) & ~0xff) | i);
    b2 = __builtin_amdgcn_fmed3f(b1, key, b2);
    b1 = fmaxf(b1, key);
  };
  f32x4 rw[4][2];  // W ring, four double steps deep: [slot][ft] (hi fragments only)
  if (cnt > 0) {
    const f32x4* W0 = wbase(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rw[j][0] = W0[j * 256];
      rw[j][1] = W0[j * 256 + 16];
    }
  }
  f32x4 pn[2];  // panel fragments (hi) of the next half-step: row tiles 2hf, 2hf + 1
  pn[0] = *(const f32x4*)(Ab);
  pn[1] = *(const f32x4*)(Ab + 16 * RS_LDA);
  // one block: 8 double k-steps into acc; meanwhile the previous block's accumulators (o; block number io) are booked,
  // two logits per half-step
  auto block = [&](f32x4 (&acc)[2][4], int i, const f32x4 (&o)[2][4], int io, bool do_book) {
    const f32x4* Wb = wbase(i);
    const f32x4* Wn = wbase(i + 1);
    {
      const int col = ((b0 + wave + 4 * i) << 5) + 4 * g;
      f32x4 bv[2], cn[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cc = col + 16 * ft + q;
          bv[ft][q] = cc < a.N ? (a.bias ? a.bias[cc] : 0.0f) : RS_PAD_LOGIT;
          cn[ft][q] = (ukeys && cc < a.N) ? a.wnorm[cc] : 0.0f;
        }
      // the accumulators start from bias + cx * ||W_v||: the bound rides along in the initial value (an FMA where the
      // plain form has a move), the bookkeeping stays at 3 operations per logit
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[0][rt][q] = fmaf(cx[rt], cn[0][q], bv[0][q]);
          acc[1][rt][q] = fmaf(cx[rt], cn[1][q], bv[1][q]);
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int sl = d & 3;
      const h8 w0 = as_h8(rw[sl][0]), w1 = as_h8(rw[sl][1]);
#if !defined(ST_ABL_S_NOW)  /* kbench ablation: the W ring is never refilled */
      {
        const f32x4* Wx = d + 4 < 8 ? Wb + (d + 4) * 256 : Wn + (d + 4 - 8) * 256;
        rw[sl][0] = Wx[0];
        rw[sl][1] = Wx[16];
      }
#endif
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const h8 p0 = as_h8(pn[0]), p1 = as_h8(pn[1]);
#if !defined(ST_ABL_S_NOLDS)  /* kbench ablation: the panel fragments are never reloaded */
        {
          const int nd = hf ? ((d + 1) & 7) : d, nrt = hf ? 0 : 2;
          const float* q = Ab + nrt * 16 * RS_LDA + 16 * nd;
          pn[0] = *(const f32x4*)(q);
          pn[1] = *(const f32x4*)(q + 16 * RS_LDA);
        }
#endif
        const int rt = 2 * hf;
        acc[0][rt] = mfma16x32h(w0, p0, acc[0][rt]);
        acc[1][rt] = mfma16x32h(w1, p0, acc[1][rt]);
        acc[0][rt + 1] = mfma16x32h(w0, p1, acc[0][rt + 1]);
        acc[1][rt + 1] = mfma16x32h(w1, p1, acc[1][rt + 1]);
#if defined(ST_ABL_S_NOBOOK)  /* kbench ablation: one of the 32 logits of a block is booked */
        if (do_book && d == 0 && hf == 0)
#else
        if (do_book)
#endif
        {  // half-step u = 2d + hf books slot (ft, rt', q-pair): 16 half-steps x 2 logits = 32
          const int u = 2 * d + hf;
          const int ft = u & 1, rtb = (u >> 1) & 3, qp = u >> 3;  // registers 2 qp, 2 qp + 1
          book(o[ft][rtb][2 * qp], io, k1[ft][rtb][2 * qp], k2[ft][rtb][2 * qp]);
          book(o[ft][rtb][2 * qp + 1], io, k1[ft][rtb][2 * qp + 1], k2[ft][rtb][2 * qp + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto book_all = [&](const f32x4 (&o)[2][4], int io) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) book(o[ft][rt][q], io, k1[ft][rt][q], k2[ft][rt][q]);
  };
  f32x4 A[2][4], B[2][4];
  if (cnt > 0) {
    block(A, 0, A, 0, false);
    int i = 1;
    for (; i + 1 < cnt; i += 2) {
      block(B, i, A, i - 1, true);
      block(A, i + 1, B, i, true);
    }
    if (i < cnt) {  // an even number of blocks: one more into B, then book B
      block(B, i, A, i - 1, true);
      book_all(B, i);
    } else {
      book_all(A, cnt - 1);
    }
  }
  // keys -> (value, column, second): the 8 slots of a lane, then the 4 lanes (g) of a row, then the 4 waves through LDS
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    Top2 t = {-INFINITY, 0x7fffffff, -INFINITY};
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Top2 u;
        u.m1 = k1[ft][rt][q];
        u.i1 = ((b0 + wave + 4 * (__float_as_int(u.m1) & 0xff)) << 5) + 16 * ft + 4 * g + q;
        u.m2 = k2[ft][rt][q];
        t = top2_merge(t, u);
      }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      Top2 u;
      u.m1 = __shfl_xor(t.m1, o, 64);
      u.i1 = __shfl_xor(t.i1, o, 64);
      u.m2 = __shfl_xor(t.m2, o, 64);
      t = top2_merge(t, u);
    }
    if (g == 0) {
      float* p = red + (wave * 64 + 16 * rt + r16) * 4;
      p[0] = t.m1;
      p[1] = __int_as_float(t.i1);
      p[2] = t.m2;
    }
  }
  __syncthreads();
  if (tid < 64 && m0 + tid < n) {
    Top2 t = {red[tid * 4 + 0], __float_as_int(red[tid * 4 + 1]), red[tid * 4 + 2]};
    for (int w = 1; w < 4; ++w) {
      Top2 u = {red[(w * 64 + tid) * 4 + 0], __float_as_int(red[(w * 64 + tid) * 4 + 1]), red[(w * 64 + tid) * 4 + 2]};
      t = top2_merge(t, u);
    }
    // keys sit up to 2^-15 |v| off their logits: finish_kernel's margin has a term for that
    a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m1, __int_as_float(t.i1), t.m2, rownorm[tid]);
  }
}
// LDS position (in dwords from the start of a panel row's hi or lo part) of the 8 halves that lane group g = lane >> 4 of a
// 16x16x32 operand contracts in double step d (k = 32 d + 8 g .. + 7).  ds_read_b128 is served in four 16-lane groups,
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS): each holds all 16 values of
// lane & 15 and TWO values of g, so with the natural order (position 16 d + 4 g, row stride = 4 mod 64) the g = 1 lanes
// sit 4 banks beside the g = 0 lanes of the next row: a 2-way conflict on every read (SQ_LDS_BANK_CONFLICT = 3.9 cycles per
// instruction on the first 16x16x32 kernels).  Here g & 1 moves a fragment by 64 dwords = the bank modulus, and a group's
// 16 lanes fall on 16 different 4-bank slots.
__device__ __forceinline__ int rp16_frag_pos(int d, int g) { return 64 * (g & 1) + 8 * d + 4 * (g >> 1); }
// the same for the writer that holds features 4 c4 .. 4 c4 + 3 of a row (2 dwords)
__device__ __forceinline__ int rp16_quad_pos(int c4) { return rp16_frag_pos(c4 >> 3, (c4 >> 1) & 3) + 2 * (c4 & 1); }

// bias and column norms of the screening pass, padded to whole 32-column blocks (out: [2][npad]): the kernel then loads
// them 4 at a time and needs no per-column range test
template <int UNUSED = 0>
__global__ void rp_screen_colpad_kernel(const float* bias, const float* wnorm, int N, int npad, float* out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= npad) return;
  out[c] = c < N ? (bias ? bias[c] : 0.0f) : RS_PAD_LOGIT;
  out[npad + c] = (c < N && wnorm) ? wnorm[c] : 0.0f;
}
static inline size_t rp_screen_colpad_floats(int N) { return 2 * (size_t)((N + 31) / 32 * 32); }
static int launch_rp_screen_colpad(const float* bias, const float* wnorm, int N, float* out, hipStream_t st) {
  const int npad = (N + 31) / 32 * 32;
  hipLaunchKernelGGL(rp_screen_colpad_kernel<0>, dim3((npad + 255) / 256), dim3(256), 0, st, bias, wnorm, N, npad, out);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
// ------------------------------------------------------------------------------------------------
// rp_screen16_kernel with cheaper bookkeeping.  The compiled form of the kernel above spends FIVE VALU operations per
// logit (v_and + v_or: a literal cannot sit in v_and_or_b32 next to a scalar; a v_max x,x,x that quiets the bit-cast
// key before fmaxf; v_med3; v_max), and with the bookkeeping ablated the pass takes 40 % less time
// (profiles/r02_screen_ablations.txt).  Here the two logits a lane holds for one (row, q) slot - feature tiles 0 and
// 1 - share ONE (best, second) pair: key = (v & ~0x1ff) | (2 block + ft) in one v_and_or_b32 each, then
//     second = max(second, med3(best, ka, kb)),  best = max3(best, ka, kb)
// i.e. 2.5 operations per logit, 32 key registers instead of 64.  Keys sit within 2^-14 |v| of their logits
// (finish_kernel's key_err covers two of them).  The W fragments are addressed as scalar base + 32-bit lane offset
// (no 64-bit VALU address arithmetic per load), and the panel fragments of a half-step are requested before the
// MFMAs of the half-step before it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rs_key(float v, unsigned keep, int code) {
  float k;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(k) : "v"(v), "v"(keep), "s"(code));
  return k;
}
__device__ __forceinline__ float rs_max(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float rs_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
#ifndef RS_PD
#define RS_PD 1
#endif
#ifndef RS_WRING
#define RS_WRING 4  /* 4 or 8 double steps */
#endif
template <bool UKEYS>  // UKEYS: a.wnorm and a.ln_a given, keys are upper bounds; a.colpad is required
__global__ __launch_bounds__(256, 2) void rp_screen16p_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                  // [64][RS_LDA]
  float* red = smem + 64 * RS_LDA;   // [4 waves][64 rows][4]
  float* rownorm = red + 4 * 64 * 4; // [64]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  {
    // one wave per row, 16 rows each: global -> registers -> LayerNorm -> hi halves in LDS
    f32x4 ga = {1, 1, 1, 1}, gb = {0, 0, 0, 0};
    if (a.ln_a) {
      ga = *(const f32x4*)(a.ln_a + lane * 4);
      gb = *(const f32x4*)(a.ln_b + lane * 4);
    }
    f32x4 v[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int r = min(m0 + wave * 16 + rr, n - 1);
      v[rr] = *(const f32x4*)(a.X + (size_t)(a.xrows ? a.xrows[r] : r) * 256 + lane * 4);
    }
    int bad = 0;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int r = wave * 16 + rr;
      f32x4 y = v[rr];
      float nrm = INFINITY;  // no LayerNorm, no norm: every row counts as ambiguous
      if (a.ln_a) {
        y = layernorm256r(y, ga, gb);
        nrm = sqrtf(wave_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w));
      }
      if (lane == 0) rownorm[r] = nrm;
      const float m = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)));
      bad |= !(m <= 65504.0f);
      h4 hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) hi[j] = (_Float16)y[j];
      *(h4*)(As + r * RS_LDA + rp16_quad_pos(lane)) = hi;
    }
    if (bad && a.range_flag) *a.range_flag = 1;
  }
  __syncthreads();

  // with column norms the keys are upper bounds of the exact logits (see rp_screen16_kernel)
  constexpr bool ukeys = UKEYS;
  float cx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (ukeys) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) cx[rt] = a.scale * rownorm[16 * rt + r16];
  }
  float k1[4][4], k2[4][4];  // running (best, second) keys per (row tile, register), both feature tiles
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) k1[rt][q] = k2[rt][q] = -INFINITY;
  const int nblk = (a.N + 31) >> 5;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;  // column split blockIdx.y covers blocks [b0, b1)
  const int b0 = blockIdx.y * per;
  const int b1 = min(nblk, b0 + per);
  const int cnt = (b1 - b0 > wave) ? (b1 - b0 - wave + 3) / 4 : 0;  // this wave's blocks: b0 + wave, b0 + wave + 4, ...
  const unsigned wl = (unsigned)(((g >> 1) * 128 + 32 * (g & 1) + r16) * 16);  // the lane's bytes inside a 4 KiB double step
  const float* Ab = As + r16 * RS_LDA + rp16_frag_pos(0, g);
  // W through a buffer descriptor: scalar block offset + the lane's 32-bit offset + immediate, no VALU address arithmetic.
  // A block of the packed image is 32 KiB (8 double steps of [hi ft0 | hi ft1 | lo | lo] x 1 KiB).
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, nblk * 32768, 0x00020000);
  auto wbase = [&](int i) -> int { return (b0 + wave + 4 * min(i, cnt - 1)) * 32768; };  // wave-uniform
#if defined(ST_SCR_GLOBAL)
  auto wfrag = [&](int p, int d, int ft) -> f32x4 { return *(const f32x4*)((const char*)a.W + (size_t)(unsigned)p + wl + d * 4096 + ft * 256); };
#else
  auto wfrag = [&](int p, int d, int ft) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + ft * 256, p + d * 4096, 0));
  };
#endif
  unsigned keep = 0xfffffe00u;
  asm volatile("" : "+v"(keep));  // one register for the whole loop (not a literal per use)
  auto book2 = [&](float va, float vb, int io, float& b1, float& b2) {  // 5 VALU operations for two logits
    const float ka = rs_key(va, keep, 2 * io), kb = rs_key(vb, keep, 2 * io + 1);
    b2 = rs_max(b2, __builtin_amdgcn_fmed3f(b1, ka, kb));
    b1 = rs_max3(b1, ka, kb);
  };
  f32x4 rw[RS_WRING][2];  // W ring, RS_WRING double steps deep: [slot][ft] (hi fragments only)
  if (cnt > 0) {
    const int W0 = wbase(0);
#pragma unroll
    for (int j = 0; j < RS_WRING; ++j) {
      rw[j][0] = wfrag(W0, j, 0);
      rw[j][1] = wfrag(W0, j, 1);
    }
  }
  // panel fragments (hi), requested three half-steps ahead: half-step h = 2d + hf uses row tiles 2hf, 2hf + 1 of double
  // step d out of pn[h & 3] (the LDS answers in more than one half-step's 4 MFMAs when two workgroups share it)
  f32x4 pn[4][2];
  auto pfrag = [&](int h, int j) -> f32x4 { return *(const f32x4*)(Ab + (2 * (h & 1) + j) * 16 * RS_LDA + 8 * ((h >> 1) & 7)); };
#pragma unroll
  for (int h = 0; h < RS_PD; ++h) {
    pn[h][0] = pfrag(h, 0);
    pn[h][1] = pfrag(h, 1);
  }
  // one block: 8 double k-steps into acc; meanwhile the previous block's accumulators (o; block number io) are booked,
  // one (row tile, register) slot = two logits per half-step
  auto block = [&](f32x4 (&acc)[2][4], int i, const f32x4 (&o)[2][4], int io, bool do_book) {
    const int Wb = wbase(i);
    const int Wn = wbase(i + 1);
    {
      const int col = ((b0 + wave + 4 * i) << 5) + 4 * g;
      f32x4 bv[2], cn[2];  // from the padded copies: two or four 16-byte loads, no column-range selects
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        bv[ft] = *(const f32x4*)(a.colpad + col + 16 * ft);
        cn[ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ukeys) cn[ft] = *(const f32x4*)(a.colpad + 32 * nblk + col + 16 * ft);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[0][rt][q] = fmaf(cx[rt], cn[0][q], bv[0][q]);
          acc[1][rt][q] = fmaf(cx[rt], cn[1][q], bv[1][q]);
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int sl = d % RS_WRING;
      const h8 w0 = as_h8(rw[sl][0]), w1 = as_h8(rw[sl][1]);
#if !defined(ST_ABL_S_NOW)  /* kbench ablation: the W ring is never refilled */
      if (d + RS_WRING < 8) {
        rw[sl][0] = wfrag(Wb, d + RS_WRING, 0);
        rw[sl][1] = wfrag(Wb, d + RS_WRING, 1);
      } else {
        rw[sl][0] = wfrag(Wn, d + RS_WRING - 8, 0);
        rw[sl][1] = wfrag(Wn, d + RS_WRING - 8, 1);
      }
#endif
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int h = 2 * d + hf;
        const h8 p0 = as_h8(pn[h & 3][0]), p1 = as_h8(pn[h & 3][1]);
#if !defined(ST_ABL_S_NOLDS)  /* kbench ablation: the panel fragments are never reloaded */
        pn[(h + RS_PD) & 3][0] = pfrag(h + RS_PD, 0);
        pn[(h + RS_PD) & 3][1] = pfrag(h + RS_PD, 1);
#endif
#if !defined(ST_SCR_NOMIDBAR)
        __builtin_amdgcn_sched_barrier(0);  // the requests go out before this half-step's MFMAs
#endif
        const int rt = 2 * hf;
        acc[0][rt] = mfma16x32h(w0, p0, acc[0][rt]);
        acc[1][rt] = mfma16x32h(w1, p0, acc[1][rt]);
        acc[0][rt + 1] = mfma16x32h(w0, p1, acc[0][rt + 1]);
        acc[1][rt + 1] = mfma16x32h(w1, p1, acc[1][rt + 1]);
#if defined(ST_ABL_S_NOBOOK)  /* kbench ablation: one of the 16 slots of a block is booked */
        if (do_book && d == 0 && hf == 0)
#else
        if (do_book)
#endif
        {  // half-step u = 2d + hf books slot (row tile u & 3, register u >> 2); row tiles 2, 3 were written last
          const int u = 2 * d + hf, rtb = u & 3, q = u >> 2;
          book2(o[0][rtb][q], o[1][rtb][q], io, k1[rtb][q], k2[rtb][q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto book_all = [&](const f32x4 (&o)[2][4], int io) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) book2(o[0][rt][q], o[1][rt][q], io, k1[rt][q], k2[rt][q]);
  };
  f32x4 A[2][4], B[2][4];
  if (cnt > 0) {
    block(A, 0, A, 0, false);
    int i = 1;
    for (; i + 1 < cnt; i += 2) {
      block(B, i, A, i - 1, true);
      block(A, i + 1, B, i, true);
    }
    if (i < cnt) {  // an even number of blocks: one more into B, then book B
      block(B, i, A, i - 1, true);
      book_all(B, i);
    } else {
      book_all(A, cnt - 1);
    }
  }
  // keys -> (value, column, second): the 4 slots of a lane, then the 4 lanes (g) of a row, then the 4 waves through LDS
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    Top2 t = {-INFINITY, 0x7fffffff, -INFINITY};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Top2 u;
      u.m1 = k1[rt][q];
      const int code = __float_as_int(u.m1) & 0x1ff;
      u.i1 = ((b0 + wave + 4 * (code >> 1)) << 5) + 16 * (code & 1) + 4 * g + q;
      u.m2 = k2[rt][q];
      t = top2_merge(t, u);
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      Top2 u;
      u.m1 = __shfl_xor(t.m1, o, 64);
      u.i1 = __shfl_xor(t.i1, o, 64);
      u.m2 = __shfl_xor(t.m2, o, 64);
      t = top2_merge(t, u);
    }
    if (g == 0) {
      float* p = red + (wave * 64 + 16 * rt + r16) * 4;
      p[0] = t.m1;
      p[1] = __int_as_float(t.i1);
      p[2] = t.m2;
    }
  }
  __syncthreads();
  if (tid < 64 && m0 + tid < n) {
    Top2 t = {red[tid * 4 + 0], __float_as_int(red[tid * 4 + 1]), red[tid * 4 + 2]};
    for (int w = 1; w < 4; ++w) {
      Top2 u = {red[(w * 64 + tid) * 4 + 0], __float_as_int(red[(w * 64 + tid) * 4 + 1]), red[(w * 64 + tid) * 4 + 2]};
      t = top2_merge(t, u);
    }
    a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m1, __int_as_float(t.i1), t.m2, rownorm[tid]);
  }
}
// grid (panels, column splits); needs ceil(N / 128) <= RS_MAX_BLOCKS (the block number must fit the 8 key bits)
static int launch_rp_screen_pipelined(const GemmArgs& a, int n_bound, int splits, hipStream_t st) {
  ST_ENSURE_LDS((rp_screen_kernel<0>), RS_LDS);
  if (n_bound <= 0) return ST_OK;
  if (a.fmt != ST_GEMM_F16X3 || (a.N + 127) / 128 > RS_MAX_BLOCKS) {
    st_set_error("rp_screen: needs the f16x3 format and N <= %d (N=%d)", RS_MAX_BLOCKS * 128, a.N);
    return ST_EINVAL;
  }
  rp_ab_env();
  if (rp_screen_shape16 && rp_screen_paired && a.colpad) {
    ST_ENSURE_LDS((rp_screen16p_kernel<true>), RS_LDS);
    ST_ENSURE_LDS((rp_screen16p_kernel<false>), RS_LDS);
    if (a.wnorm && a.ln_a)
      hipLaunchKernelGGL((rp_screen16p_kernel<true>), dim3((n_bound + 63) / 64, splits), dim3(256), RS_LDS, st, a);
    else
      hipLaunchKernelGGL((rp_screen16p_kernel<false>), dim3((n_bound + 63) / 64, splits), dim3(256), RS_LDS, st, a);
  } else if (rp_screen_shape16) {
    ST_ENSURE_LDS((rp_screen16_kernel<0>), RS_LDS);
    hipLaunchKernelGGL((rp_screen16_kernel<0>), dim3((n_bound + 63) / 64, splits), dim3(256), RS_LDS, st, a);
  } else {
    hipLaunchKernelGGL((rp_screen_kernel<0>), dim3((n_bound + 63) / 64, splits), dim3(256), RS_LDS, st, a);
  }
  ST_LAUNCH_CHECK();
  return ST_OK;
}

// ------------------------------------------------------------------------------------------------
// QKV projections, software-pipelined form (f16x3; grid = 64-row panels x column splits): the same
// arithmetic as rp_gemm_kernel<EPI_BIAS / EPI_QKV, 4, true> (LayerNorm'ed split panel in LDS, three products per
// k-step, bias-initialised accumulators), but with two accumulator sets: while block i runs its 96 MFMAs, the 32
// output values per lane of block i-1 are stored, two per k-step, inside the same basic block.  In rp_gemm_kernel the
// stores of a block sit between two blocks' MFMAs and the matrix pipe idles through them (with the stores ablated
// the QKV kernel was 25 % faster).
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void rp_lin_kernel(GemmArgs a) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_QKV, "rp_lin_kernel: BIAS or QKV");
  extern __shared__ __align__(16) float smem[];
  float* As = smem;  // [64][ST_LDA]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int bad = 0;
  auto store_split = [&](float* row, int c4, f32x4 v) {  // panel_store4<true> without the branch
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    bad |= !(m <= 65504.0f);
    h4 hi, lo;
    split_f16(v, hi, lo);
    _Float16* ph = (_Float16*)row;
    *(h4*)(ph + c4 * 4) = hi;
    *(h4*)(ph + 256 + c4 * 4) = lo;
  };
  if (a.ln_a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 6, c4 = f & 63;
      *(f32x4*)(As + r * ST_LDA + c4 * 4) = *(const f32x4*)(a.X + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4);
    }
    __syncthreads();
    const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
    const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      float* row = As + (wave * 16 + rr) * ST_LDA;  // one wave owns the whole row: in-place is safe
      store_split(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 6, c4 = f & 63;
      store_split(As + r * ST_LDA, c4, *(const f32x4*)(a.X + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4));
    }
  }
  __syncthreads();

  constexpr int SPB = 16, R = 4;
  const int nblk = (a.N + 31) >> 5;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;  // column split blockIdx.y covers blocks [b0, b1)
  const int b0 = blockIdx.y * per;
  const int b1 = min(nblk, b0 + per);
  const int cnt = (b1 - b0 > wave) ? (b1 - b0 - wave + 3) / 4 : 0;  // this wave's blocks: b0 + wave, b0 + wave + 4, ...
  const f32x4* Wl = (const f32x4*)a.W + lane;
  const float* Ab = As + l31 * ST_LDA + 4 * half;
  auto wbase = [&](int i) -> const f32x4* {  // fragments of step g of block i: hi at [g * 128], lo at [g * 128 + 64]
    return Wl + (size_t)(b0 + wave + 4 * min(i, cnt - 1)) * SPB * 2 * 64;
  };
  // this lane's 32 output rows are the same for every block: row offsets once (QKV: also the K/V cache slots)
  // QKV: the K/V cache slot (job) of each panel row, in the LDS words behind the panel (kept out of the registers)
  int* slot = (int*)(smem + 64 * ST_LDA);
  if (EPI == EPI_QKV) {
    if (tid < 64) slot[tid] = a.idx[min(m0 + tid, n - 1)];
    __syncthreads();
  }
  auto put = [&](float v, int i, int mi, int r) {
    const int blk = b0 + wave + 4 * i;  // wave-uniform: a 32-column block lies entirely in q, k or v
    const int col = (blk << 5) + l31;
    const int rl = mi * 32 + mfma32_row(r, lane);
    if (m0 + rl >= n || col >= a.N) return;
    if (EPI == EPI_BIAS) {
      ST_STORE(&a.out[(size_t)(m0 + rl) * a.ldo + col], v);
    } else if (blk < 8) {
      ST_STORE(&a.q[(size_t)(m0 + rl) * 256 + col], v);
    } else {
      float* base = (blk < 16) ? a.kbase : a.vbase;
      ST_STORE(&base[(long)slot[rl] * a.kv_stride + (col & 255)], v);
    }
  };
  f32x4 rh[R], rl[R];
  if (cnt > 0) {
    const f32x4* W0 = wbase(0);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      rh[j] = W0[(size_t)j * 128];
      rl[j] = W0[(size_t)j * 128 + 64];
    }
  }
  f32x4 an[4];
  an[0] = *(const f32x4*)(Ab);
  an[1] = *(const f32x4*)(Ab + 128);
  an[2] = *(const f32x4*)(Ab + 32 * ST_LDA);
  an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128);
  auto block = [&](f32x16& n0, f32x16& n1, int i, const f32x16& o0, const f32x16& o1, int io, bool do_put) {
    const f32x4* Wb = wbase(i);
    const f32x4* Wn = wbase(i + 1);
    const int col = ((b0 + wave + 4 * i) << 5) + l31;
    const float bv = (a.bias && col < a.N) ? a.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) n0[r] = n1[r] = bv;
#pragma unroll
    for (int kg = 0; kg < SPB; ++kg) {
      const h8 bh = as_h8(rh[kg % R]), bl = as_h8(rl[kg % R]);
      const f32x4* Wx = kg + R < SPB ? Wb + (size_t)(kg + R) * 128 : Wn + (size_t)(kg + R - SPB) * 128;
      rh[kg % R] = Wx[0];
      rl[kg % R] = Wx[64];
      const h8 a0h = as_h8(an[0]), a0l = as_h8(an[1]), a1h = as_h8(an[2]), a1l = as_h8(an[3]);
      const int kn = (kg + 1) & (SPB - 1);
      an[0] = *(const f32x4*)(Ab + kn * 8);
      an[1] = *(const f32x4*)(Ab + 128 + kn * 8);
      an[2] = *(const f32x4*)(Ab + 32 * ST_LDA + kn * 8);
      an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128 + kn * 8);
      n0 = mfma16h(a0h, bh, n0);
      n1 = mfma16h(a1h, bh, n1);
      n0 = mfma16h(a0h, bl, n0);
      n1 = mfma16h(a1h, bl, n1);
      n0 = mfma16h(a0l, bh, n0);
      n1 = mfma16h(a1l, bh, n1);
      if (do_put) {
        put(o0[kg], io, 0, kg);
        put(o1[kg], io, 1, kg);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  f32x16 A0, A1, B0, B1;
  if (cnt > 0) {
    block(A0, A1, 0, A0, A1, 0, false);
    int i = 1;
    for (; i + 1 < cnt; i += 2) {
      block(B0, B1, i, A0, A1, i - 1, true);
      block(A0, A1, i + 1, B0, B1, i, true);
    }
    if (i < cnt) {
      block(B0, B1, i, A0, A1, i - 1, true);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        put(B0[r], i, 0, r);
        put(B1[r], i, 1, r);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        put(A0[r], cnt - 1, 0, r);
        put(A1[r], cnt - 1, 1, r);
      }
    }
  }
  if (bad && a.range_flag) *a.range_flag = 1;
}
// ------------------------------------------------------------------------------------------------
// rp_lin_kernel on v_mfma_f32_16x16x32_f16 tiles with W as the A operand (see rp_layer_tail16_kernel for the lane
// addressing of the unchanged packed image and LDS panel): the shape the power-managed chip clocks higher on, and an
// epilogue of 8 16-byte stores per lane and block (4 consecutive features of one row) instead of 32 dwords.
// Needs N % 32 == 0 and 16-byte aligned output rows (ldo % 4, kv_stride % 4).
// ------------------------------------------------------------------------------------------------
#ifndef RP_LIN_RING
#define RP_LIN_RING 2  /* 2, 4 or 8; 4 measured the same (313.6 against 317.9 us at 229 k rows) for 28 more registers */
#endif
template <int EPI>
__global__ __launch_bounds__(256, 2) void rp_lin16_kernel(GemmArgs a) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_QKV, "rp_lin16_kernel: BIAS or QKV");
  extern __shared__ __align__(16) float smem[];
  float* As = smem;  // [64][ST_LDA]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  int bad = 0;
  if (a.ln_a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 6, c4 = f & 63;
      *(f32x4*)(As + r * ST_LDA + c4 * 4) = *(const f32x4*)(a.X + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4);
    }
    __syncthreads();
    const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
    const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      float* row = As + (wave * 16 + rr) * ST_LDA;  // one wave owns the whole row: in-place is safe
      panel_store4_nb16(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb), bad);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 6, c4 = f & 63;
      panel_store4_nb16(As + r * ST_LDA, c4, *(const f32x4*)(a.X + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4), bad);
    }
  }
  int* slot = (int*)(smem + 64 * ST_LDA);  // QKV: the K/V cache slot (job) of each panel row
  if (EPI == EPI_QKV && tid < 64) slot[tid] = a.idx[min(m0 + tid, n - 1)];
  __syncthreads();

  const int nblk = (a.N + 31) >> 5;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;  // column split blockIdx.y covers blocks [b0, b1)
  const int b0 = blockIdx.y * per;
  const int b1 = min(nblk, b0 + per);
  const int cnt = (b1 - b0 > wave) ? (b1 - b0 - wave + 3) / 4 : 0;  // this wave's blocks: b0 + wave, b0 + wave + 4, ...
  // W through a buffer descriptor: scalar block offset + the lane's 32-bit offset + immediate (no VALU address
  // arithmetic); a block of the packed image is 32 KiB = 8 double steps of [hi ft0 | hi ft1 | .. | lo ft0 | lo ft1] KiB
  const unsigned wl = (unsigned)(((g >> 1) * 128 + 32 * (g & 1) + r16) * 16);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, nblk * 32768, 0x00020000);
  const float* Ab = As + r16 * ST_LDA + rp16_frag_pos(0, g);  // K-permuted panel rows: conflict-free ds_read_b128
  auto wbase = [&](int i) -> int { return (b0 + wave + 4 * min(i, cnt - 1)) * 32768; };  // wave-uniform
  auto wfetch = [&](int p, int d, f32x4 (&dst)[4]) {
    dst[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl, p + d * 4096, 0));
    dst[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1024, p + d * 4096, 0));
    dst[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 256, p + d * 4096, 0));
    dst[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1280, p + d * 4096, 0));
  };
  // where the lane's four rows (one per row tile) go, computed once: element offset of (row, feature 4 g) in the plain
  // output / q, and of the row's K/V cache slot
  size_t orow[4], kvrow[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int rl = 16 * rt + r16;
    orow[rt] = (size_t)min(m0 + rl, n - 1) * (EPI == EPI_BIAS ? a.ldo : 256) + 4 * g;
    kvrow[rt] = EPI == EPI_QKV ? (size_t)((long)slot[rl] * a.kv_stride) + 4 * g : 0;
  }
  // FULL: every row of the panel exists (all panels but the last): the stores need no row test
  auto run = [&](auto FULL) {
    auto put = [&](const f32x4& v, int i, int ft, int rt) {
      const int blk = b0 + wave + 4 * i;  // wave-uniform: a 32-column block lies entirely in q, k or v
      const int colu = (blk << 5) + 16 * ft;  // uniform part of the column
      if (!decltype(FULL)::value && m0 + 16 * rt + r16 >= n) return;
#if defined(ST_ABL_NOSTORE)  /* kbench ablation: keep the value live, never store */
      if (v.x != 1.2345e-30f) return;
#endif
      if (EPI == EPI_BIAS) {
        ST_STORE4((f32x4*)(a.out + colu + orow[rt]), v);
      } else if (blk < 8) {
        ST_STORE4((f32x4*)(a.q + colu + orow[rt]), v);
      } else {
        float* base = (blk < 16) ? a.kbase : a.vbase;
        ST_STORE4((f32x4*)(base + (colu & 255) + kvrow[rt]), v);
      }
    };
    // W ring, RP_LIN_RING double steps deep.  Loads and stores share one in-order counter (vmcnt): the wait for a W
    // fragment also waits for every output store issued before that load, so the ring depth is the time a store gets to
    // be acknowledged before it stalls the MFMAs (2 steps = 0.5 us: the kernel then runs at memory time PLUS MFMA time).
    f32x4 rw[RP_LIN_RING][4];
    if (cnt > 0) {
#pragma unroll
      for (int j = 0; j < RP_LIN_RING; ++j) wfetch(wbase(0), j, rw[j]);
    }
    f32x4 pn[4];
    pn[0] = *(const f32x4*)(Ab);
    pn[1] = *(const f32x4*)(Ab + 128);
    pn[2] = *(const f32x4*)(Ab + 16 * ST_LDA);
    pn[3] = *(const f32x4*)(Ab + 16 * ST_LDA + 128);
    // one block: 8 double k-steps into acc; meanwhile the previous block's accumulators (o; block number io) are stored,
    // one 16-byte store per double step
    auto block = [&](f32x4 (&acc)[2][4], int i, const f32x4 (&o)[2][4], int io, bool do_put) {
      const int Wb = wbase(i);
      const int Wn = wbase(i + 1);
      {
        const int col = ((b0 + wave + 4 * i) << 5) + 4 * g;
        const f32x4 z = {0, 0, 0, 0};
        const f32x4 bv0 = a.bias ? *(const f32x4*)(a.bias + col) : z, bv1 = a.bias ? *(const f32x4*)(a.bias + col + 16) : z;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          acc[0][rt] = bv0;
          acc[1][rt] = bv1;
        }
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int sl = d % RP_LIN_RING;
        const h8 wh0 = as_h8(rw[sl][0]), wl0 = as_h8(rw[sl][1]), wh1 = as_h8(rw[sl][2]), wl1 = as_h8(rw[sl][3]);
        if (d + RP_LIN_RING < 8)
          wfetch(Wb, d + RP_LIN_RING, rw[sl]);
        else
          wfetch(Wn, d + RP_LIN_RING - 8, rw[sl]);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const h8 p0h = as_h8(pn[0]), p0l = as_h8(pn[1]), p1h = as_h8(pn[2]), p1l = as_h8(pn[3]);
          {
            const int nd = hf ? ((d + 1) & 7) : d, nrt = hf ? 0 : 2;
            const float* q = Ab + nrt * 16 * ST_LDA + 8 * nd;
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
          if (do_put && hf == 1) put(o[d & 1][d >> 1], io, d & 1, d >> 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    f32x4 A[2][4], B[2][4];
    if (cnt > 0) {
      block(A, 0, A, 0, false);
      int i = 1;
      for (; i + 1 < cnt; i += 2) {
        block(B, i, A, i - 1, true);
        block(A, i + 1, B, i, true);
      }
      if (i < cnt) {
        block(B, i, A, i - 1, true);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) put(B[ft][rt], i, ft, rt);
      } else {
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) put(A[ft][rt], cnt - 1, ft, rt);
      }
    }
  };
  if (m0 + 64 <= n)
    run(std::true_type{});
  else
    run(std::false_type{});
  if (bad && a.range_flag) *a.range_flag = 1;
}
// ------------------------------------------------------------------------------------------------
// Exact argmax (three products per term) on 16x16x32 tiles: the redo pass of the screened argmax and the generator of
// small or unscreened batches in the f16x3 format.  rp_lin16_kernel's loop (W as the A operand, K-permuted panel rows,
// buffer-descriptor W loads) with rp_screen16p_kernel's slot structure: the two logits a lane holds for one (row tile,
// register) slot - feature tiles 0 and 1 - share one exact (best, code, second) triple, code = 2 block + ft; columns
// reach a slot in increasing order and `>` is strict, so exact ties keep the lowest column.  Bookkeeping of block i - 1
// (4 operations per logit) sits inside the MFMAs of block i; the final merge is local over a lane's 4 slots plus two
// lane exchanges per row tile (rp_gemm_kernel<EPI_ARGMAX>: five exchanges for each of 32 slots, which with 16 column
// splits cost as much as the workgroup's MFMAs).  Needs a.colpad (padded bias: columns >= N never win).
// ------------------------------------------------------------------------------------------------
template <int UNUSED = 0>
__global__ __launch_bounds__(256, 2) void rp_argmax16_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                 // [64][ST_LDA]
  float* red = smem + 64 * ST_LDA;  // [4 waves][64 rows][4]
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  auto xrow = [&](int r) -> int { return a.xrows ? a.xrows[r] : r; };  // optional row gather (redo pass)
  int bad = 0;
  if (a.ln_a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 6, c4 = f & 63;
      *(f32x4*)(As + r * ST_LDA + c4 * 4) = *(const f32x4*)(a.X + (size_t)xrow(min(m0 + r, n - 1)) * 256 + c4 * 4);
    }
    __syncthreads();
    const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
    const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      float* row = As + (wave * 16 + rr) * ST_LDA;  // one wave owns the whole row: in-place is safe
      panel_store4_nb16(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb), bad);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 6, c4 = f & 63;
      panel_store4_nb16(As + r * ST_LDA, c4, *(const f32x4*)(a.X + (size_t)xrow(min(m0 + r, n - 1)) * 256 + c4 * 4), bad);
    }
  }
  __syncthreads();

  const int nblk = (a.N + 31) >> 5;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;  // column split blockIdx.y covers blocks [b0, b1)
  const int b0 = blockIdx.y * per;
  const int b1 = min(nblk, b0 + per);
  const int cnt = (b1 - b0 > wave) ? (b1 - b0 - wave + 3) / 4 : 0;  // this wave's blocks: b0 + wave, b0 + wave + 4, ...
  const unsigned wl = (unsigned)(((g >> 1) * 128 + 32 * (g & 1) + r16) * 16);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, nblk * 32768, 0x00020000);
  const float* Ab = As + r16 * ST_LDA + rp16_frag_pos(0, g);
  auto wbase = [&](int i) -> int { return (b0 + wave + 4 * min(i, cnt - 1)) * 32768; };  // wave-uniform
  auto wfetch = [&](int p, int d, f32x4 (&dst)[4]) {
    dst[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl, p + d * 4096, 0));
    dst[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1024, p + d * 4096, 0));
    dst[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 256, p + d * 4096, 0));
    dst[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1280, p + d * 4096, 0));
  };
  float m1[4][4], m2[4][4];  // exact (best, second) per (row tile, register) slot, both feature tiles
  int bc[4][4];              // code of the best: 2 block + feature tile
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      m1[rt][q] = m2[rt][q] = -INFINITY;
      bc[rt][q] = 0;
    }
  auto book = [&](float v, int code, float& b1, int& c1, float& b2) {  // 4 VALU operations
    c1 = v > b1 ? code : c1;
    b2 = __builtin_amdgcn_fmed3f(b1, v, b2);
    b1 = rs_max(b1, v);
  };
  f32x4 rw[2][4];
  if (cnt > 0) {
    wfetch(wbase(0), 0, rw[0]);
    wfetch(wbase(0), 1, rw[1]);
  }
  f32x4 pn[4];
  pn[0] = *(const f32x4*)(Ab);
  pn[1] = *(const f32x4*)(Ab + 128);
  pn[2] = *(const f32x4*)(Ab + 16 * ST_LDA);
  pn[3] = *(const f32x4*)(Ab + 16 * ST_LDA + 128);
  auto block = [&](f32x4 (&acc)[2][4], int i, const f32x4 (&o)[2][4], int io, bool do_book) {
    const int Wb = wbase(i);
    const int Wn = wbase(i + 1);
    {
      const int col = ((b0 + wave + 4 * i) << 5) + 4 * g;
      const f32x4 bv0 = *(const f32x4*)(a.colpad + col), bv1 = *(const f32x4*)(a.colpad + col + 16);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        acc[0][rt] = bv0;
        acc[1][rt] = bv1;
      }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int sl = d & 1;
      const h8 wh0 = as_h8(rw[sl][0]), wl0 = as_h8(rw[sl][1]), wh1 = as_h8(rw[sl][2]), wl1 = as_h8(rw[sl][3]);
      if (d + 2 < 8)
        wfetch(Wb, d + 2, rw[sl]);
      else
        wfetch(Wn, d + 2 - 8, rw[sl]);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const h8 p0h = as_h8(pn[0]), p0l = as_h8(pn[1]), p1h = as_h8(pn[2]), p1l = as_h8(pn[3]);
        {
          const int nd = hf ? ((d + 1) & 7) : d, nrt = hf ? 0 : 2;
          const float* q = Ab + nrt * 16 * ST_LDA + 8 * nd;
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
        if (do_book) {  // half-step u = 2d + hf books slot (row tile u & 3, register u >> 2): feature tile 0, then 1
          const int u = 2 * d + hf, rtb = u & 3, q = u >> 2;
          book(o[0][rtb][q], 2 * io, m1[rtb][q], bc[rtb][q], m2[rtb][q]);
          book(o[1][rtb][q], 2 * io + 1, m1[rtb][q], bc[rtb][q], m2[rtb][q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto book_all = [&](const f32x4 (&o)[2][4], int io) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        book(o[0][rt][q], 2 * io, m1[rt][q], bc[rt][q], m2[rt][q]);
        book(o[1][rt][q], 2 * io + 1, m1[rt][q], bc[rt][q], m2[rt][q]);
      }
  };
  f32x4 A[2][4], B[2][4];
  if (cnt > 0) {
    block(A, 0, A, 0, false);
    int i = 1;
    for (; i + 1 < cnt; i += 2) {
      block(B, i, A, i - 1, true);
      block(A, i + 1, B, i, true);
    }
    if (i < cnt) {
      block(B, i, A, i - 1, true);
      book_all(B, i);
    } else {
      book_all(A, cnt - 1);
    }
  }
  if (bad && a.range_flag) *a.range_flag = 1;
  // (best, column, second): the 4 slots of a lane, then the 4 lanes (g) of a row, then the 4 waves through LDS
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    Top2 t = {-INFINITY, 0x7fffffff, -INFINITY};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Top2 u;
      u.m1 = m1[rt][q];
      const int code = bc[rt][q];
      u.i1 = u.m1 == -INFINITY ? 0x7fffffff : ((b0 + wave + 4 * (code >> 1)) << 5) + 16 * (code & 1) + 4 * g + q;
      u.m2 = m2[rt][q];
      t = top2_merge(t, u);
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      Top2 u;
      u.m1 = __shfl_xor(t.m1, o, 64);
      u.i1 = __shfl_xor(t.i1, o, 64);
      u.m2 = __shfl_xor(t.m2, o, 64);
      t = top2_merge(t, u);
    }
    if (g == 0) {
      float* p = red + (wave * 64 + 16 * rt + r16) * 4;
      p[0] = t.m1;
      p[1] = __int_as_float(t.i1);
      p[2] = t.m2;
    }
  }
  __syncthreads();
  if (tid < 64 && m0 + tid < n) {
    Top2 t = {red[tid * 4 + 0], __float_as_int(red[tid * 4 + 1]), red[tid * 4 + 2]};
    for (int w = 1; w < 4; ++w) {
      Top2 u = {red[(w * 64 + tid) * 4 + 0], __float_as_int(red[(w * 64 + tid) * 4 + 1]), red[(w * 64 + tid) * 4 + 2]};
      t = top2_merge(t, u);
    }
    a.partial[(size_t)(m0 + tid) * gridDim.y + blockIdx.y] = make_float4(t.m1, __int_as_float(t.i1), t.m2, 0.0f);
  }
}
static int launch_rp_argmax16(const GemmArgs& a, int n_bound, int splits, hipStream_t st) {
  ST_ENSURE_LDS((rp_argmax16_kernel<0>), RP_LDS);
  if (n_bound <= 0) return ST_OK;
  hipLaunchKernelGGL((rp_argmax16_kernel<0>), dim3((n_bound + 63) / 64, splits), dim3(256), RP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
template <int EPI>
static int launch_rp_lin_pipelined(const GemmArgs& a, int n_bound, int splits, hipStream_t st) {
  ST_ENSURE_LDS((rp_lin_kernel<EPI>), RP_LDS);
  ST_ENSURE_LDS((rp_lin16_kernel<EPI>), RP_LDS);
  rp_ab_env();
  if (n_bound <= 0) return ST_OK;
  const bool aligned = a.N % 32 == 0 && (EPI == EPI_BIAS ? (a.ldo % 4 == 0 && ((size_t)a.out & 15) == 0)
                                                        : (a.kv_stride % 4 == 0 && ((size_t)a.q & 15) == 0 && ((size_t)a.kbase & 15) == 0 &&
                                                           ((size_t)a.vbase & 15) == 0)) &&
                       (!a.bias || ((size_t)a.bias & 15) == 0);
  if (rp_lin_shape16 && aligned)
    hipLaunchKernelGGL((rp_lin16_kernel<EPI>), dim3((n_bound + 63) / 64, splits), dim3(256), RP_LDS, st, a);
  else
    hipLaunchKernelGGL((rp_lin_kernel<EPI>), dim3((n_bound + 63) / 64, splits), dim3(256), RP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

// a.fmt selects the operand format the packed W image was built for (ST_GEMM_F32 / ST_GEMM_F16X3)
// NW is the wave count of the f32 kernel; the f16x3 kernels always run 4 waves (2 workgroups x 4 waves with
// 256 registers each measured faster than 8-wave workgroups at 128: QKV 199 vs 242 us at 131072 rows)
static inline int rp_nw(int fmt, int nw_f32) { return fmt == ST_GEMM_F16X3 ? 4 : nw_f32; }
template <int EPI, int NW = 4>
static int launch_rp(const GemmArgs& a, int n_bound, int splits, hipStream_t st) {
  if (EPI == EPI_ARGMAX && a.fmt == ST_GEMM_F16X3 && a.colpad) {
    rp_ab_env();
    if (rp_argmax_shape16) return launch_rp_argmax16(a, n_bound, splits, st);
  }
  return a.fmt == ST_GEMM_F16X3 ? launch_rp_t<EPI, 4, true>(a, n_bound, splits, st)
                                : launch_rp_t<EPI, NW, false>(a, n_bound, splits, st);
}

// ------------------------------------------------------------------------------------------------
// K > 256 with N == 256 (feed_forward.w_2: K = d_ff): the X panel is walked in 256-wide K chunks
// (reloaded into LDS between two barriers), the whole [64,256] output stays in accumulators
// (wave w owns column blocks w and w+4: 2 blocks x 2 row tiles = 64 registers).  Epilogue: BIAS/RESID.
// ------------------------------------------------------------------------------------------------
template <int EPI, bool H = false>
__global__ __launch_bounds__(256, 2) void rp_kchunk_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the W address math on the scalar unit
  const int half = lane >> 5, l31 = lane & 31;
  constexpr int SPC = H ? 16 : 32;  // ring steps per 256-wide K chunk
  constexpr int F = H ? 2 : 1;      // fragments per step
  const int K = a.K, nchunk = K >> 8, spb = nchunk * SPC;  // steps per column block
  // split-K (gridDim.y > 1): this workgroup sums chunks [c_begin, c_end) into slab blockIdx.y of `out`
  // ([gridDim.y][n_host][ldo]; summed by slab_reduce_kernel / slab_reduce_rows_kernel); bias and residual enter
  // through slab 0 only
  const bool first = blockIdx.y == 0;
  const int cps = (nchunk + gridDim.y - 1) / gridDim.y;
  const int c_begin = blockIdx.y * cps, c_end = min(nchunk, c_begin + cps);
  float* outp = a.out + (size_t)blockIdx.y * a.n_host * a.ldo;
  const f32x4* W0 = (const f32x4*)a.W + (size_t)wave * spb * F * 64 + lane;        // block `wave`
  const f32x4* W1 = (const f32x4*)a.W + (size_t)(wave + 4) * spb * F * 64 + lane;  // block `wave + 4`
  const int total = max(c_end * SPC, 1);
  f32x4 r0[4], r1[4], r0l[H ? 4 : 1], r1l[H ? 4 : 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t o = (size_t)min(c_begin * SPC + j, total - 1) * F * 64;
    r0[j] = W0[o];
    r1[j] = W1[o];
    if (H) {
      r0l[j] = W0[o + 64];
      r1l[j] = W1[o + 64];
    }
  }
  // accumulators start from bias (+ residual + cross constant): the fetch overlaps the whole K loop
  f32x16 acc[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int col = (wave + 4 * nb) * 32 + l31;
    const float bv = (a.bias && first) ? a.bias[col] : 0.0f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = bv;
        if (EPI == EPI_RESID && first) {
          const int row = min(m0 + mi * 32 + mfma32_row(r, lane), n - 1);
          v += a.resid[(size_t)row * a.ldo + col];
          if (a.cconst) v += a.cconst[(size_t)a.idx[row] * a.ld_cc + col];
        }
        acc[mi][nb][r] = v;
      }
  }
  const float* Ab = As + l31 * ST_LDA + 4 * half;
  for (int c = c_begin; c < c_end; ++c) {
    if (c > c_begin) __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 6, c4 = f & 63;
      panel_store4<H>(As + r * ST_LDA, c4, *(const f32x4*)(a.X + (size_t)min(m0 + r, n - 1) * K + c * 256 + c4 * 4), a.range_flag);
    }
    __syncthreads();
    for (int g0 = 0; g0 < SPC; g0 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kg = g0 + j;
        const size_t nx = (size_t)min(c * SPC + kg + 4, total - 1) * F * 64;
        if (H) {
          const h8 b0h = as_h8(r0[j]), b1h = as_h8(r1[j]), b0l = as_h8(r0l[j]), b1l = as_h8(r1l[j]);
          r0[j] = W0[nx];
          r1[j] = W1[nx];
          r0l[j] = W0[nx + 64];
          r1l[j] = W1[nx + 64];
          const h8 a0h = as_h8(*(const f32x4*)(Ab + kg * 8)), a0l = as_h8(*(const f32x4*)(Ab + 128 + kg * 8));
          const h8 a1h = as_h8(*(const f32x4*)(Ab + 32 * ST_LDA + kg * 8));
          const h8 a1l = as_h8(*(const f32x4*)(Ab + 32 * ST_LDA + 128 + kg * 8));
          acc[0][0] = mfma16h(a0h, b0h, acc[0][0]);
          acc[1][0] = mfma16h(a1h, b0h, acc[1][0]);
          acc[0][1] = mfma16h(a0h, b1h, acc[0][1]);
          acc[1][1] = mfma16h(a1h, b1h, acc[1][1]);
          acc[0][0] = mfma16h(a0h, b0l, acc[0][0]);
          acc[1][0] = mfma16h(a1h, b0l, acc[1][0]);
          acc[0][1] = mfma16h(a0h, b1l, acc[0][1]);
          acc[1][1] = mfma16h(a1h, b1l, acc[1][1]);
          acc[0][0] = mfma16h(a0l, b0h, acc[0][0]);
          acc[1][0] = mfma16h(a1l, b0h, acc[1][0]);
          acc[0][1] = mfma16h(a0l, b1h, acc[0][1]);
          acc[1][1] = mfma16h(a1l, b1h, acc[1][1]);
        } else {
          const f32x4 b0 = r0[j], b1 = r1[j];
          r0[j] = W0[nx];
          r1[j] = W1[nx];
          const f32x4 a0 = *(const f32x4*)(Ab + kg * 8);
          const f32x4 a1 = *(const f32x4*)(Ab + 32 * ST_LDA + kg * 8);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc[0][0] = mfma32(a0[s], b0[s], acc[0][0]);
            acc[1][0] = mfma32(a1[s], b0[s], acc[1][0]);
            acc[0][1] = mfma32(a0[s], b1[s], acc[0][1]);
            acc[1][1] = mfma32(a1[s], b1[s], acc[1][1]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int col = (wave + 4 * nb) * 32 + l31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + mi * 32 + mfma32_row(r, lane);
        if (row < n) ST_STORE(&outp[(size_t)row * a.ldo + col], acc[mi][nb][r]);
      }
  }
}

// out[i] (+)= sum_y slab[y][i]   (split-K combine; float4 per thread)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, size_t stride4,
                                                          float* __restrict__ out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = *(const f32x4*)(slab + i * 4);
  for (int y = 1; y < nsplit; ++y) s += *(const f32x4*)(slab + ((size_t)y * stride4 + i) * 4);
  *(f32x4*)(out + i * 4) = s;
}

// split-K combine over the LIVE rows only (n read from device memory): out[row][c] = sum_y slab[y][row][c]
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void slab_reduce_rows_kernel(const float* __restrict__ slab, int nsplit, size_t stride,
                                                               float* __restrict__ out, const int* __restrict__ n_dev, int ldo) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (size_t)*n_dev * ldo) return;
  f32x4 s = *(const f32x4*)(slab + i);
  for (int y = 1; y < nsplit; ++y) s += *(const f32x4*)(slab + (size_t)y * stride + i);
  *(f32x4*)(out + i) = s;
}

template <int EPI, bool H>
static int launch_rp_kchunk_t(const GemmArgs& a, int n_bound, hipStream_t st, int ksplit) {
  ST_ENSURE_LDS((rp_kchunk_kernel<EPI, H>), RP_LDS);
  if (n_bound <= 0) return ST_OK;
  if (a.N != 256 || a.K % 256 != 0 || a.K <= 0) {
    st_set_error("rp_kchunk: needs N == 256 and K a multiple of 256 (N=%d K=%d)", a.N, a.K);
    return ST_EINVAL;
  }
  hipLaunchKernelGGL((rp_kchunk_kernel<EPI, H>), dim3((n_bound + 63) / 64, ksplit), dim3(256), RP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
template <int EPI>
static int launch_rp_kchunk(const GemmArgs& a, int n_bound, hipStream_t st, int ksplit = 1) {
  return a.fmt == ST_GEMM_F16X3 ? launch_rp_kchunk_t<EPI, true>(a, n_bound, st, ksplit)
                                : launch_rp_kchunk_t<EPI, false>(a, n_bound, st, ksplit);
}

// ------------------------------------------------------------------------------------------------
// Fused post-attention half-layer:  x1 = att.Wo^T + bo + x (+ cross-attention constant);  x <- x1;
// h = relu(LayerNorm(x1).W1^T + b1).   One 64-row panel: the attention output is the LDS panel for the
// O-projection (wave w owns column blocks w, w+4), x1 replaces it in LDS, is normalised in place and
// becomes the panel for feed_forward.w_1 (N = d_ff, blocks w, w+4, ...).  Saves a launch, one panel
// load and the x round trip of the unfused pair rp_gemm<RESID> + rp_gemm<RELU>.
// ------------------------------------------------------------------------------------------------
struct OFArgs {
  const float* att;    // [n,256]
  const int* n_dev;
  int n_host;
  const float *wo, *bo;          // packed [256,256], [256]
  float* x;                      // [n,256] residual in, x1 out
  const float* cconst;           // optional [S,256]
  const int* idx;
  const float *ln_a, *ln_b;      // LayerNorm before the FFN
  const float *w1, *b1;          // packed [d_ff,256], [d_ff]
  float* h;                      // [n,d_ff]   (rp_oproj_ffn1_kernel only)
  int d_ff;
  int fmt;                       // operand format of wo / w1 / w2 (ST_GEMM_F32 / ST_GEMM_F16X3)
  const float *w2, *b2;          // packed [256,d_ff], [256]   (rp_layer_tail_kernel only)
  int* range_flag;               // f16x3: set to 1 when an operand leaves the f16 range (null = unchecked)
  // rp_layer_tail16_kernel<SITE, true>: the residual is the decode step's EMBEDDING, computed here from the token the row's
  // job emitted last (emb[token] * 16 + pe[pos], model.py:126-133) instead of read from x (x is still written)
  const float* emb;              // [vocab,256]
  const float* pe_row;           // pe + pos * 256
  const int32_t* ys;             // [jobs,T] emitted ids
  int T, pos, bos, vocab;
  // rp_layer_tail16_kernel<SITE, 2>: the same for the encoder's first layer, whose rows carry their own token and position
  // (every position < RP_TAIL_PE_ROWS: the first rows of pe are kept in LDS)
  const int32_t* tokens;         // [n]
  const int32_t* tok_pos;        // [n]
  const float* pe;               // [>= RP_TAIL_PE_ROWS, 256]
};
#define RP_TAIL_PE_ROWS 16

template <bool H = false>
__global__ __launch_bounds__(256, 2) void rp_oproj_ffn1_kernel(OFArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the W address math on the scalar unit
  const int half = lane >> 5, l31 = lane & 31;
  constexpr int SPB = H ? 16 : 32;  // ring steps per 32-column block (K = 256)
  constexpr int F = H ? 2 : 1;      // fragments per step
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = tid + i * 256;
    const int r = f >> 6, c4 = f & 63;
    panel_store4<H>(As + r * ST_LDA, c4, *(const f32x4*)(a.att + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4), a.range_flag);
  }
  // ---- phase 1: O-projection, accumulators start from bias + residual (+ constant) ----
  const f32x4* W0 = (const f32x4*)a.wo + (size_t)wave * SPB * F * 64 + lane;
  const f32x4* W1 = (const f32x4*)a.wo + (size_t)(wave + 4) * SPB * F * 64 + lane;
  f32x4 r0[4], r1[4], r0l[H ? 4 : 1], r1l[H ? 4 : 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r0[j] = W0[(size_t)j * F * 64];
    r1[j] = W1[(size_t)j * F * 64];
    if (H) {
      r0l[j] = W0[(size_t)j * F * 64 + 64];
      r1l[j] = W1[(size_t)j * F * 64 + 64];
    }
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int col = (wave + 4 * nb) * 32 + l31;
    const float bv = a.bo[col];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(m0 + mi * 32 + mfma32_row(r, lane), n - 1);
        float v = bv + a.x[(size_t)row * 256 + col];
        if (a.cconst) v += a.cconst[(size_t)a.idx[row] * 256 + col];
        acc[mi][nb][r] = v;
      }
  }
  __syncthreads();
  const float* Ab = As + l31 * ST_LDA + 4 * half;
  for (int g0 = 0; g0 < SPB; g0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kg = g0 + j;
      const size_t nx = (size_t)min(kg + 4, SPB - 1) * F * 64;
      if (H) {
        const h8 b0h = as_h8(r0[j]), b1h = as_h8(r1[j]), b0l = as_h8(r0l[j]), b1l = as_h8(r1l[j]);
        r0[j] = W0[nx];
        r1[j] = W1[nx];
        r0l[j] = W0[nx + 64];
        r1l[j] = W1[nx + 64];
        const h8 a0h = as_h8(*(const f32x4*)(Ab + kg * 8)), a0l = as_h8(*(const f32x4*)(Ab + 128 + kg * 8));
        const h8 a1h = as_h8(*(const f32x4*)(Ab + 32 * ST_LDA + kg * 8));
        const h8 a1l = as_h8(*(const f32x4*)(Ab + 32 * ST_LDA + 128 + kg * 8));
        acc[0][0] = mfma16h(a0h, b0h, acc[0][0]);
        acc[1][0] = mfma16h(a1h, b0h, acc[1][0]);
        acc[0][1] = mfma16h(a0h, b1h, acc[0][1]);
        acc[1][1] = mfma16h(a1h, b1h, acc[1][1]);
        acc[0][0] = mfma16h(a0h, b0l, acc[0][0]);
        acc[1][0] = mfma16h(a1h, b0l, acc[1][0]);
        acc[0][1] = mfma16h(a0h, b1l, acc[0][1]);
        acc[1][1] = mfma16h(a1h, b1l, acc[1][1]);
        acc[0][0] = mfma16h(a0l, b0h, acc[0][0]);
        acc[1][0] = mfma16h(a1l, b0h, acc[1][0]);
        acc[0][1] = mfma16h(a0l, b1h, acc[0][1]);
        acc[1][1] = mfma16h(a1l, b1h, acc[1][1]);
      } else {
        const f32x4 b0 = r0[j], b1 = r1[j];
        r0[j] = W0[nx];
        r1[j] = W1[nx];
        const f32x4 a0 = *(const f32x4*)(Ab + kg * 8);
        const f32x4 a1 = *(const f32x4*)(Ab + 32 * ST_LDA + kg * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0][0] = mfma32(a0[s], b0[s], acc[0][0]);
          acc[1][0] = mfma32(a1[s], b0[s], acc[1][0]);
          acc[0][1] = mfma32(a0[s], b1[s], acc[0][1]);
          acc[1][1] = mfma32(a1[s], b1[s], acc[1][1]);
        }
      }
    }
  }
  __syncthreads();  // every wave is done reading the attention panel
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int col = (wave + 4 * nb) * 32 + l31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = mi * 32 + mfma32_row(r, lane);
        const float v = acc[mi][nb][r];
        As[lr * ST_LDA + col] = v;
        if (m0 + lr < n) ST_STORE(&a.x[(size_t)(m0 + lr) * 256 + col], v);
      }
  }
  __syncthreads();
  {
    const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
    const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      float* row = As + (wave * 16 + rr) * ST_LDA;
      panel_store4<H>(row, lane, layernorm256(*(f32x4*)(row + lane * 4), ga, gb), a.range_flag);
    }
  }
  __syncthreads();
  // ---- phase 2: h = relu(xn . W1^T + b1), this wave's blocks wave, wave+4, ... ----
  constexpr int R = 8;
  const int nblk = a.d_ff >> 5;
  const int cnt = (nblk - wave + 3) >> 2;
  const int total = cnt * SPB;
  const f32x4* Wl = (const f32x4*)a.w1 + lane;
  auto wload = [&](int g, int p) -> f32x4 {
    g = min(g, total - 1);
    return Wl[(((size_t)(wave + 4 * (g / SPB)) * SPB + (g & (SPB - 1))) * F + p) * 64];
  };
  f32x4 ring[R], ringl[H ? R : 1];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    ring[j] = wload(j, 0);
    if (H) ringl[j] = wload(j, 1);
  }
  f32x16 acc0 = {0}, acc1 = {0};
  f32x4 an[H ? 4 : 1];
  if (H) {
    an[0] = *(const f32x4*)(Ab);
    an[1] = *(const f32x4*)(Ab + 128);
    an[2] = *(const f32x4*)(Ab + 32 * ST_LDA);
    an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128);
  }
  for (int g0 = 0; g0 < total; g0 += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int kg = (g0 + j) & (SPB - 1);
      if (H) {
        const h8 bh = as_h8(ring[j]), bl = as_h8(ringl[j]);
        ring[j] = wload(g0 + j + R, 0);
        ringl[j] = wload(g0 + j + R, 1);
        // A fragments of the NEXT step are requested before this step's MFMAs (LDS latency ~ one step)
        const h8 a0h = as_h8(an[0]), a0l = as_h8(an[1]), a1h = as_h8(an[2]), a1l = as_h8(an[3]);
        const int kn = (kg + 1) & (SPB - 1);
        an[0] = *(const f32x4*)(Ab + kn * 8);
        an[1] = *(const f32x4*)(Ab + 128 + kn * 8);
        an[2] = *(const f32x4*)(Ab + 32 * ST_LDA + kn * 8);
        an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128 + kn * 8);
        acc0 = mfma16h(a0h, bh, acc0);
        acc1 = mfma16h(a1h, bh, acc1);
        acc0 = mfma16h(a0h, bl, acc0);
        acc1 = mfma16h(a1h, bl, acc1);
        acc0 = mfma16h(a0l, bh, acc0);
        acc1 = mfma16h(a1l, bh, acc1);
      } else {
        const f32x4 b = ring[j];
        ring[j] = wload(g0 + j + R, 0);
        const f32x4 a0 = *(const f32x4*)(Ab + kg * 8);
        const f32x4 a1 = *(const f32x4*)(Ab + 32 * ST_LDA + kg * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = mfma32(a0[s], b[s], acc0);
          acc1 = mfma32(a1[s], b[s], acc1);
        }
      }
    }
    if (((g0 + R) & (SPB - 1)) == 0) {
      const int col = ((wave + 4 * (g0 / SPB)) << 5) + l31;
      const float bv = a.b1[col];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x16 ac = mi ? acc1 : acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + mi * 32 + mfma32_row(r, lane);
          if (row < n) ST_STORE(&a.h[(size_t)row * a.d_ff + col], fmaxf(ac[r] + bv, 0.0f));
        }
      }
      acc0 = (f32x16){0};
      acc1 = (f32x16){0};
    }
  }
}

#define RP_FUSE_MIN_ROWS 65536  /* below this the unfused pair (with column splits) fills the chip better */
template <bool H>
static int launch_oproj_ffn1_t(const OFArgs& a, int n_bound, hipStream_t st) {
  ST_ENSURE_LDS((rp_oproj_ffn1_kernel<H>), RP_LDS);
  if (n_bound <= 0) return ST_OK;
  hipLaunchKernelGGL(rp_oproj_ffn1_kernel<H>, dim3((n_bound + 63) / 64), dim3(256), RP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
static int launch_oproj_ffn1(const OFArgs& a, int n_bound, hipStream_t st) {
  return a.fmt == ST_GEMM_F16X3 ? launch_oproj_ffn1_t<true>(a, n_bound, st) : launch_oproj_ffn1_t<false>(a, n_bound, st);
}

// ------------------------------------------------------------------------------------------------
// Whole post-attention half-layer in one kernel (f16x3 format):
//   x1 = att.Wo^T + bo + x (+ cross-attention constant);   x <- x1 + relu(LayerNorm(x1).W1^T + b1).W2^T + b2
// One 64-row panel per workgroup of 8 waves; the hidden activations never leave the CU.  LDS holds two
// split-f16 panels (133 KiB, one workgroup per CU): P1 = attention output, then x1, then LayerNorm(x1);
// P2 = one 256-wide chunk of the hidden layer.  Wave w owns output column block w: x1 lives in its
// accumulators (32 registers) from the O-projection to the final store and receives the FFN2 partial
// sums chunk by chunk.  FFN1 is computed TRANSPOSED (W1 fragment as the A operand, x fragment as B), so a
// lane ends up holding 4 consecutive hidden units of ONE row per accumulator group and writes them to P2
// as packed halves (ds_write_b64) in exactly the layout the FFN2 A fragments are read from.
// HBM traffic per row: att + x in, x out (3 KiB) instead of 13 KiB for the unfused chain.
// ------------------------------------------------------------------------------------------------
#define RP_TAIL_LDS ((2 * 64 * ST_LDA + 128) * 4)  /* two panels + the cross-attention index and the token of the next panel's rows */
#define RP_TAIL_LDS_ENC (RP_TAIL_LDS + (128 + RP_TAIL_PE_ROWS * ST_LDA) * 4)  /* + the rows' positions and the first rows of pe */
#define RP_TAIL_MIN_ROWS rp_tail_min_rows
#define RP_TAIL_MMA6(ACC0, ACC1, A0H, A0L, A1H, A1L, BH, BL) \
  ACC0 = mfma16h(A0H, BH, ACC0);                             \
  ACC1 = mfma16h(A1H, BH, ACC1);                             \
  ACC0 = mfma16h(A0H, BL, ACC0);                             \
  ACC1 = mfma16h(A1H, BL, ACC1);                             \
  ACC0 = mfma16h(A0L, BH, ACC0);                             \
  ACC1 = mfma16h(A1L, BH, ACC1);

#ifdef ST_TAIL_STAMPS
__device__ unsigned long long g_tail_stamps[64];
#define TSTAMP(i) do { if (blockIdx.x == 100 && threadIdx.x == 0) g_tail_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(i)
#endif
#ifndef RP_TAIL_RING
#define RP_TAIL_RING 4  /* W fragment ring depth in k-steps (16 steps per pass) */
#endif
template <int UNUSED = 0>
__global__ __launch_bounds__(512, RP_TAIL_RING > 4 ? 1 : 2) void rp_layer_tail_kernel(OFArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* P1 = smem;
  float* P2 = smem + 64 * ST_LDA;
  int* sIdx = (int*)(smem + 2 * 64 * ST_LDA);
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  // PERSISTENT over panels (grid = one workgroup per CU): panel p + gridDim.x of this workgroup follows panel p, and its
  // attention panel, residual rows and cross-attention constants are requested BEFORE the last FFN2 pass of panel p, so
  // the entry latency of a panel (10.7 k of 93.7 k cycles in the one-panel-per-workgroup form, plus ~7 us of workgroup
  // turnaround per round) is paid once per workgroup.  A grid of one workgroup per panel runs the loop once (kbench A/B).
  const int mstride = gridDim.x * 64;
  int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int half = lane >> 5, l31 = lane & 31;
  int col = wave * 32 + l31;  // this lane's output column (O-projection and FFN2)
  const int nchunk = a.d_ff >> 8;
  TSTAMP(31);

  // 16 k-steps of one 32-column block: fragments (hi, lo) through a register ring RG steps deep
  constexpr int RG = RP_TAIL_RING;
  f32x4 rh[RG], rl[RG];
  auto ring_fill = [&](const f32x4* Wb) {
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      rh[j] = Wb[(size_t)j * 128];
      rl[j] = Wb[(size_t)j * 128 + 64];
    }
  };

  int bad = 0;    // an operand left the f16 range (or is NaN): reported once at the end
  f32x4 attr[8];  // this thread's share of an attention panel on its way from HBM to P1
  auto att_load = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 512;
      attr[i] = *(const f32x4*)(a.att + (size_t)min(mb + (f >> 6), n - 1) * 256 + (f & 63) * 4);
    }
  };
  auto att_store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 512;
      panel_store4_nb(P1 + (f >> 6) * ST_LDA, f & 63, attr[i], bad);
    }
  };
  // The residual (+ cross-attention constant: a dependent gather through idx, staged in LDS one phase earlier) joins the
  // accumulators AFTER the O-projection pass, so only the attention panel's latency stands before the first MFMA.
  f32x16 ax0, ax1, px0, px1, pc0, pc1;
  auto idx_stage = [&](int mb) {
    if (a.cconst && tid < 64) sIdx[tid] = a.idx[min(mb + tid, n - 1)];
  };
  auto resid_load = [&](int mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int r0 = mfma32_row(r, lane);
      const int row0 = min(mb + r0, n - 1), row1 = min(mb + 32 + r0, n - 1);
      px0[r] = a.x[(size_t)row0 * 256 + col];
      px1[r] = a.x[(size_t)row1 * 256 + col];
    }
    if (a.cconst) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int r0 = mfma32_row(r, lane);
        pc0[r] = a.cconst[(size_t)sIdx[r0] * 256 + col];
        pc1[r] = a.cconst[(size_t)sIdx[32 + r0] * 256 + col];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) pc0[r] = pc1[r] = 0.0f;
    }
  };
  att_load(m0);
  idx_stage(m0);
  // parameters used after barriers are fetched now (their latency would otherwise be exposed there)
  const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
  const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
  const float b2v = a.b2[col];
  const float bov = a.bo[col];
  f32x4 b1n[4];  // FFN1 bias of this lane's 16 hidden units in the coming chunk (transposed tile: 8g + 4*half + q)
  auto b1_fetch = [&](int c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) b1n[g] = *(const f32x4*)(a.b1 + (c * 8 + wave) * 32 + 8 * g + 4 * half);
  };
  att_store();
  __syncthreads();  // sIdx of the first panel
  resid_load(m0);
  bool first = true;
  for (;;) {
  // The lane coordinates are re-derived from an opaque copy of the thread index in every iteration: otherwise the
  // compiler hoists ~120 registers of loop-invariant addresses out of the panel loop and spills them.
  TSTAMP(35);
  asm volatile("" : "+v"(tid));
  lane = tid & 63;
  half = lane >> 5;
  l31 = lane & 31;
  col = wave * 32 + l31;
  // ---- O-projection; accumulators start from the bias ----
  const f32x4* Wo = (const f32x4*)a.wo + (size_t)wave * 16 * 128 + lane;
  auto W1p = [&](int c) { return (const f32x4*)a.w1 + (size_t)(c * 8 + wave) * 16 * 128 + lane; };                  // hidden block 8c + wave
  auto W2p = [&](int c) { return (const f32x4*)a.w2 + ((size_t)wave * (a.d_ff >> 4) + c * 16) * 128 + lane; };  // k-steps 16c .. 16c+15 of output block `wave`
  if (first) ring_fill(Wo);
  first = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) ax0[r] = ax1[r] = bov;
  TSTAMP(0);
  __syncthreads();
  TSTAMP(1);
  const float* A1 = P1 + l31 * ST_LDA + 4 * half;  // fragment base in P1 (rows l31 and l31 + 32)
  const float* A2 = P2 + l31 * ST_LDA + 4 * half;
  // one 16-step pass: standard orientation (panel rows x W columns) or transposed (W rows x panel rows)
  // (the W ring never drains: the last RG steps of a pass request the first RG steps of the NEXT pass, Wn)
  auto pass = [&](const float* Ab, const f32x4* Wb, const f32x4* Wn, f32x16& c0, f32x16& c1, bool transposed) {
    f32x4 an[4];  // A fragments of the next k-step (requested one step ahead)
    an[0] = *(const f32x4*)(Ab);
    an[1] = *(const f32x4*)(Ab + 128);
    an[2] = *(const f32x4*)(Ab + 32 * ST_LDA);
    an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128);
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += RG) {
#pragma unroll
      for (int j = 0; j < RG; ++j) {
        const int kg = g0 + j;
        const h8 bh = as_h8(rh[j]), bl = as_h8(rl[j]);
        if (kg + RG < 16) {  // static after unrolling
          rh[j] = Wb[(size_t)(kg + RG) * 128];
          rl[j] = Wb[(size_t)(kg + RG) * 128 + 64];
        } else {
          rh[j] = Wn[(size_t)(kg + RG - 16) * 128];
          rl[j] = Wn[(size_t)(kg + RG - 16) * 128 + 64];
        }
        const h8 a0h = as_h8(an[0]), a0l = as_h8(an[1]), a1h = as_h8(an[2]), a1l = as_h8(an[3]);
        if (kg + 1 < 16) {
          an[0] = *(const f32x4*)(Ab + (kg + 1) * 8);
          an[1] = *(const f32x4*)(Ab + 128 + (kg + 1) * 8);
          an[2] = *(const f32x4*)(Ab + 32 * ST_LDA + (kg + 1) * 8);
          an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128 + (kg + 1) * 8);
        }
        if (transposed) {
          c0 = mfma16h(bh, a0h, c0);
          c1 = mfma16h(bh, a1h, c1);
          c0 = mfma16h(bl, a0h, c0);
          c1 = mfma16h(bl, a1h, c1);
          c0 = mfma16h(bh, a0l, c0);
          c1 = mfma16h(bh, a1l, c1);
        } else {
          RP_TAIL_MMA6(c0, c1, a0h, a0l, a1h, a1l, bh, bl)
        }
        // keep the step structure: without this the scheduler hoists the A-fragment reads of many steps and the kernel
        // loses ~4 % (kbench 600 -> 572 us at 131072 rows)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  pass(A1, Wo, W1p(0), ax0, ax1, false);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    ax0[r] += px0[r] + pc0[r];
    ax1[r] += px1[r] + pc1[r];
  }
  TSTAMP(2);
  b1_fetch(0);
  if (m0 + mstride < n) idx_stage(m0 + mstride);  // read by resid_load before the last FFN2 pass, many barriers from here
  __syncthreads();  // every wave is done reading the attention panel
  TSTAMP(3);
  // ---- x1 -> P1 (f32), LayerNorm + split in place ----
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    P1[(mfma32_row(r, lane)) * ST_LDA + col] = ax0[r];
    P1[(32 + mfma32_row(r, lane)) * ST_LDA + col] = ax1[r];
  }
  TSTAMP(32);
  __syncthreads();
  TSTAMP(33);
  {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float* row = P1 + (wave * 8 + rr) * ST_LDA;
      panel_store4_nb(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb), bad);
    }
  }
  {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ax0[r] += b2v;
      ax1[r] += b2v;
    }
  }
  TSTAMP(34);
  __syncthreads();
  _Float16* H2 = (_Float16*)P2;
  TSTAMP(4);
  // ReLU + f16 split of one accumulator group (4 consecutive hidden units of one panel row) -> hidden chunk in P2
  auto epi_unit = [&](const f32x16& acc, int g, int rowbase) {
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[4 * g + q], 0.0f);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    bad |= !(mx <= 65504.0f);  // hidden unit outside the f16 range (or NaN); reported once at the end (no branch here)
    h4 hi, lo;
    split_f16(v, hi, lo);
    const int jl = wave * 32 + 8 * g + 4 * half;
    *(h4*)(H2 + (size_t)(rowbase + l31) * (2 * ST_LDA) + jl) = hi;
    *(h4*)(H2 + (size_t)(rowbase + l31) * (2 * ST_LDA) + 256 + jl) = lo;
  };
  // FFN1 pass of the NEXT chunk (transposed, into c0 / c1) with the conversion of the CURRENT chunk's accumulators
  // (o0 / o1) spread over its k-steps: one basic block, so the VALU / LDS-store work runs in the MFMAs' shadow
  auto pass_epi = [&](const float* Ab, const f32x4* Wb, const f32x4* Wn, f32x16& c0, f32x16& c1, const f32x16& o0, const f32x16& o1) {
    f32x4 an[4];  // A fragments of the next k-step, requested one step ahead (LDS latency ~ one step)
    an[0] = *(const f32x4*)(Ab);
    an[1] = *(const f32x4*)(Ab + 128);
    an[2] = *(const f32x4*)(Ab + 32 * ST_LDA);
    an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128);
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
      const int j = kg % RG;
      const h8 bh = as_h8(rh[j]), bl = as_h8(rl[j]);
      if (kg + RG < 16) {
        rh[j] = Wb[(size_t)(kg + RG) * 128];
        rl[j] = Wb[(size_t)(kg + RG) * 128 + 64];
      } else {
        rh[j] = Wn[(size_t)(kg + RG - 16) * 128];
        rl[j] = Wn[(size_t)(kg + RG - 16) * 128 + 64];
      }
      const h8 a0h = as_h8(an[0]), a0l = as_h8(an[1]), a1h = as_h8(an[2]), a1l = as_h8(an[3]);
      if (kg + 1 < 16) {
        an[0] = *(const f32x4*)(Ab + (kg + 1) * 8);
        an[1] = *(const f32x4*)(Ab + 128 + (kg + 1) * 8);
        an[2] = *(const f32x4*)(Ab + 32 * ST_LDA + (kg + 1) * 8);
        an[3] = *(const f32x4*)(Ab + 32 * ST_LDA + 128 + (kg + 1) * 8);
      }
      c0 = mfma16h(bh, a0h, c0);
      c1 = mfma16h(bh, a1h, c1);
      c0 = mfma16h(bl, a0h, c0);
      c1 = mfma16h(bl, a1h, c1);
      c0 = mfma16h(bh, a0l, c0);
      c1 = mfma16h(bh, a1l, c1);
#if defined(ST_ABL_NOCONV)   /* kbench ablation: FFN1 pass without the conversion of the previous chunk */
      asm volatile("" ::"v"(o0[kg]), "v"(o1[kg]));
#else
      if (kg & 1) {  // 8 conversion units over the 16 steps
        const int u = kg >> 1;
        if (u & 1)
          epi_unit(o1, u >> 1, 32);
        else
          epi_unit(o0, u >> 1, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);  // nothing is hoisted across k-steps (the 32 A-fragment reads would spill)
    }
  };
  auto ah_init = [&](f32x16& h0, f32x16& h1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int q = 0; q < 4; ++q) h0[4 * g + q] = h1[4 * g + q] = b1n[g][q];
    }
  };
  // ---- FFN1 of hidden chunk 0 (block jb = 8c + wave, transposed: lane = panel row, registers = hidden units) ----
  f32x16 ah0, ah1;
  ah_init(ah0, ah1);
  if (nchunk > 1) b1_fetch(1);  // consumed one chunk later
  TSTAMP(5);
  pass(A1, W1p(0), nchunk > 1 ? W1p(1) : W2p(0), ah0, ah1, true);
  TSTAMP(6);
  for (int c = 0; c + 1 < nchunk; ++c) {
    f32x16 an0, an1;
    ah_init(an0, an1);
    if (c + 2 < nchunk) b1_fetch(c + 2);
    if (c > 0) __syncthreads();  // every wave is done reading the previous hidden chunk
    // ---- hidden chunk c -> P2, overlapped with FFN1 of chunk c + 1 ----
    pass_epi(A1, W1p(c + 1), W2p(c), an0, an1, ah0, ah1);
    TSTAMP(7 + c * 5);
    __syncthreads();  // hidden chunk complete
    TSTAMP(8 + c * 5);
    // ---- FFN2 partial: x += h_c . W2[:, chunk]^T ----
    pass(A2, W2p(c), c + 2 < nchunk ? W1p(c + 2) : W2p(c + 1), ax0, ax1, false);
    TSTAMP(9 + c * 5);
    ah0 = an0;
    ah1 = an1;
  }
  {  // last chunk (peeled: the prefetch below must not be a conditional definition inside the chunk loop)
    const int c = nchunk - 1;
    // The next panel of this workgroup: its attention rows, residual rows and constants are requested here and arrive
    // under the conversion, the two barriers and the last pass (memory returns in order: W fragments requested after
    // these loads wait for them, so they are issued as early as the registers allow).  Unconditional, with clamped
    // rows, so that the registers are plain definitions (a workgroup's last panel requests rows it never uses).
    att_load(m0 + mstride);
    resid_load(m0 + mstride);
    if (c > 0) __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      epi_unit(ah0, g, 0);
      epi_unit(ah1, g, 32);
    }
    TSTAMP(7 + c * 5);
    __syncthreads();  // hidden chunk complete
    TSTAMP(8 + c * 5);
    pass(A2, W2p(c), Wo, ax0, ax1, false);  // (a workgroup's last panel requests Wo once more and drops it)
    TSTAMP(9 + c * 5);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row0 = m0 + mfma32_row(r, lane), row1 = row0 + 32;
    if (row0 < n) ST_STORE(&a.x[(size_t)row0 * 256 + col], ax0[r]);
    if (row1 < n) ST_STORE(&a.x[(size_t)row1 * 256 + col], ax1[r]);
  }
  TSTAMP(30);
  m0 += mstride;
  if (m0 >= n) break;
  att_store();
  TSTAMP(36);
  }  // panels of this workgroup
  if (bad && a.range_flag) *a.range_flag = 1;
}

// ------------------------------------------------------------------------------------------------
// The same half-layer on 32-ROW panels with 4 waves: 70.6 KiB of LDS and <= 256 registers, so TWO workgroups share
// a CU and one group's memory-bound phases (attention-panel load, residual gather, LayerNorm, final store, the
// barriers, the workgroup turnaround between panels) run under the other group's MFMA passes.  With one 64-row
// workgroup per CU those phases leave the matrix pipe idle for ~40 % of a panel (kbench stamps: 93.7 k cycles per
// panel of which the nine passes need 55 k at two waves per SIMD).
// Wave w owns output column blocks w and w + 4 (O-projection, FFN2) and hidden blocks 8c + w, 8c + w + 4 of chunk c:
// the two accumulators of a pass are two COLUMN blocks against one A fragment (the 64-row kernel: two row tiles
// against one W fragment), so a W fragment now feeds three MFMAs instead of six: twice the L2 -> CU weight traffic.
// ------------------------------------------------------------------------------------------------
#define RP_TAIL32_LDS ((2 * 32 * ST_LDA + 1024) * 4)
template <int UNUSED = 0>
__global__ __launch_bounds__(256, 2) void rp_layer_tail32_kernel(OFArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* P1 = smem;
  float* P2 = smem + 32 * ST_LDA;
  float* B1s = smem + 2 * 32 * ST_LDA;  // feed_forward.w_1 bias, staged once (d_ff <= 1024 floats)
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 32;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int col0 = wave * 32 + l31, col1 = col0 + 128;  // this lane's two output columns
  const int nchunk = a.d_ff >> 8;
  constexpr int RG = 4;
  f32x4 rh0[RG], rl0[RG], rh1[RG], rl1[RG];
  auto ring_fill = [&](const f32x4* W0, const f32x4* W1) {
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      rh0[j] = W0[(size_t)j * 128];
      rl0[j] = W0[(size_t)j * 128 + 64];
      rh1[j] = W1[(size_t)j * 128];
      rl1[j] = W1[(size_t)j * 128 + 64];
    }
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = tid + i * 256;
    const int r = f >> 6, c4 = f & 63;
    panel_store4<true>(P1 + r * ST_LDA, c4, *(const f32x4*)(a.att + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4), a.range_flag);
  }
  for (int i = tid; i < (a.d_ff >> 2); i += 256) *(f32x4*)(B1s + i * 4) = *(const f32x4*)(a.b1 + i * 4);
  const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
  const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
  const float b2v0 = a.b2[col0], b2v1 = a.b2[col1];
  const f32x4* Wo0 = (const f32x4*)a.wo + (size_t)wave * 16 * 128 + lane;
  const f32x4* Wo1 = (const f32x4*)a.wo + (size_t)(wave + 4) * 16 * 128 + lane;
  ring_fill(Wo0, Wo1);
  f32x16 ax0, ax1, px0, px1;
  {
    const float bv0 = a.bo[col0], bv1 = a.bo[col1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(m0 + mfma32_row(r, lane), n - 1);
      float v0 = a.x[(size_t)row * 256 + col0], v1 = a.x[(size_t)row * 256 + col1];
      if (a.cconst) {
        const size_t s = (size_t)a.idx[row] * 256;
        v0 += a.cconst[s + col0];
        v1 += a.cconst[s + col1];
      }
      px0[r] = v0;
      px1[r] = v1;
      ax0[r] = bv0;
      ax1[r] = bv1;
    }
  }
  __syncthreads();
  const float* A1 = P1 + l31 * ST_LDA + 4 * half;
  const float* A2 = P2 + l31 * ST_LDA + 4 * half;
  // one 16-step pass over two W blocks (W0 -> c0, W1 -> c1) against one A fragment stream
  auto pass = [&](const float* Ab, const f32x4* W0, const f32x4* W1, f32x16& c0, f32x16& c1, bool transposed) {
    f32x4 an[2];
    an[0] = *(const f32x4*)(Ab);
    an[1] = *(const f32x4*)(Ab + 128);
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
      const int j = kg % RG;
      const h8 b0h = as_h8(rh0[j]), b0l = as_h8(rl0[j]), b1h = as_h8(rh1[j]), b1l = as_h8(rl1[j]);
      if (kg + RG < 16) {
        rh0[j] = W0[(size_t)(kg + RG) * 128];
        rl0[j] = W0[(size_t)(kg + RG) * 128 + 64];
        rh1[j] = W1[(size_t)(kg + RG) * 128];
        rl1[j] = W1[(size_t)(kg + RG) * 128 + 64];
      }
      const h8 ah = as_h8(an[0]), al = as_h8(an[1]);
      if (kg + 1 < 16) {
        an[0] = *(const f32x4*)(Ab + (kg + 1) * 8);
        an[1] = *(const f32x4*)(Ab + 128 + (kg + 1) * 8);
      }
      if (transposed) {
        c0 = mfma16h(b0h, ah, c0);
        c1 = mfma16h(b1h, ah, c1);
        c0 = mfma16h(b0l, ah, c0);
        c1 = mfma16h(b1l, ah, c1);
        c0 = mfma16h(b0h, al, c0);
        c1 = mfma16h(b1h, al, c1);
      } else {
        c0 = mfma16h(ah, b0h, c0);
        c1 = mfma16h(ah, b1h, c1);
        c0 = mfma16h(ah, b0l, c0);
        c1 = mfma16h(ah, b1l, c1);
        c0 = mfma16h(al, b0h, c0);
        c1 = mfma16h(al, b1h, c1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  pass(A1, Wo0, Wo1, ax0, ax1, false);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    ax0[r] += px0[r];
    ax1[r] += px1[r];
  }
  __syncthreads();  // every wave is done reading the attention panel
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    P1[mfma32_row(r, lane) * ST_LDA + col0] = ax0[r];
    P1[mfma32_row(r, lane) * ST_LDA + col1] = ax1[r];
  }
  const f32x4* W1a = (const f32x4*)a.w1 + (size_t)wave * 16 * 128 + lane;        // hidden block 8c + wave
  const f32x4* W1b = (const f32x4*)a.w1 + (size_t)(wave + 4) * 16 * 128 + lane;  // hidden block 8c + wave + 4
  ring_fill(W1a, W1b);
  __syncthreads();
#pragma unroll 4
  for (int rr = 0; rr < 8; ++rr) {
    float* row = P1 + (wave * 8 + rr) * ST_LDA;
    panel_store4<true>(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb), a.range_flag);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    ax0[r] += b2v0;
    ax1[r] += b2v1;
  }
  __syncthreads();
  _Float16* H2 = (_Float16*)P2;
  int bad = 0;
  auto epi_unit = [&](const f32x16& acc, int g, int blk) {  // blk: position of the hidden block inside the chunk
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[4 * g + q], 0.0f);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    bad |= !(mx <= 65504.0f);
    h4 hi, lo;
    split_f16(v, hi, lo);
    const int jl = blk * 32 + 8 * g + 4 * half;
    *(h4*)(H2 + (size_t)l31 * (2 * ST_LDA) + jl) = hi;
    *(h4*)(H2 + (size_t)l31 * (2 * ST_LDA) + 256 + jl) = lo;
  };
  auto pass_epi = [&](const float* Ab, const f32x4* W0, const f32x4* W1, f32x16& c0, f32x16& c1, const f32x16& o0,
                      const f32x16& o1) {
    f32x4 an[2];
    an[0] = *(const f32x4*)(Ab);
    an[1] = *(const f32x4*)(Ab + 128);
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
      const int j = kg % RG;
      const h8 b0h = as_h8(rh0[j]), b0l = as_h8(rl0[j]), b1h = as_h8(rh1[j]), b1l = as_h8(rl1[j]);
      if (kg + RG < 16) {
        rh0[j] = W0[(size_t)(kg + RG) * 128];
        rl0[j] = W0[(size_t)(kg + RG) * 128 + 64];
        rh1[j] = W1[(size_t)(kg + RG) * 128];
        rl1[j] = W1[(size_t)(kg + RG) * 128 + 64];
      }
      const h8 ah = as_h8(an[0]), al = as_h8(an[1]);
      if (kg + 1 < 16) {
        an[0] = *(const f32x4*)(Ab + (kg + 1) * 8);
        an[1] = *(const f32x4*)(Ab + 128 + (kg + 1) * 8);
      }
      c0 = mfma16h(b0h, ah, c0);
      c1 = mfma16h(b1h, ah, c1);
      c0 = mfma16h(b0l, ah, c0);
      c1 = mfma16h(b1l, ah, c1);
      c0 = mfma16h(b0h, al, c0);
      c1 = mfma16h(b1h, al, c1);
      if (kg & 1) {  // 8 conversion units over the 16 steps
        const int u = kg >> 1;
        if (u & 1)
          epi_unit(o1, u >> 1, wave + 4);
        else
          epi_unit(o0, u >> 1, wave);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto ah_init = [&](f32x16& h0, f32x16& h1, int c) {  // FFN1 bias of this lane's hidden units (LDS copy)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 u0 = *(const f32x4*)(B1s + (c * 8 + wave) * 32 + 8 * g + 4 * half);
      const f32x4 u1 = *(const f32x4*)(B1s + (c * 8 + wave + 4) * 32 + 8 * g + 4 * half);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h0[4 * g + q] = u0[q];
        h1[4 * g + q] = u1[q];
      }
    }
  };
  f32x16 ah0, ah1;
  ah_init(ah0, ah1, 0);
  pass(A1, W1a, W1b, ah0, ah1, true);
  for (int c = 0; c < nchunk; ++c) {
    const bool more = c + 1 < nchunk;
    f32x16 an0, an1;
    if (more) {
      W1a = (const f32x4*)a.w1 + (size_t)((c + 1) * 8 + wave) * 16 * 128 + lane;
      W1b = (const f32x4*)a.w1 + (size_t)((c + 1) * 8 + wave + 4) * 16 * 128 + lane;
      ring_fill(W1a, W1b);
      ah_init(an0, an1, c + 1);
    }
    if (c > 0) __syncthreads();  // every wave is done reading the previous hidden chunk
    if (more) {
      pass_epi(A1, W1a, W1b, an0, an1, ah0, ah1);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        epi_unit(ah0, g, wave);
        epi_unit(ah1, g, wave + 4);
      }
    }
    const f32x4* W2a = (const f32x4*)a.w2 + ((size_t)wave * (a.d_ff >> 4) + c * 16) * 128 + lane;
    const f32x4* W2b = (const f32x4*)a.w2 + ((size_t)(wave + 4) * (a.d_ff >> 4) + c * 16) * 128 + lane;
    ring_fill(W2a, W2b);
    __syncthreads();  // hidden chunk complete
    pass(A2, W2a, W2b, ax0, ax1, false);
    if (more) {
      ah0 = an0;
      ah1 = an1;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + mfma32_row(r, lane);
    if (row < n) {
      ST_STORE(&a.x[(size_t)row * 256 + col0], ax0[r]);
      ST_STORE(&a.x[(size_t)row * 256 + col1], ax1[r]);
    }
  }
  if (bad && a.range_flag) *a.range_flag = 1;
}

static int launch_layer_tail32(const OFArgs& a, int n_bound, hipStream_t st) {
  ST_ENSURE_LDS((rp_layer_tail32_kernel<0>), RP_TAIL32_LDS);
  if (n_bound <= 0) return ST_OK;
  if (a.fmt != ST_GEMM_F16X3 || a.d_ff % 256 != 0 || a.d_ff > 1024 || !a.w2 || !a.b2) {
    st_set_error("rp_layer_tail32: needs the f16x3 format, d_ff %% 256 == 0, d_ff <= 1024 and w2/b2 (fmt=%d d_ff=%d)", a.fmt, a.d_ff);
    return ST_EINVAL;
  }
  hipLaunchKernelGGL(rp_layer_tail32_kernel<0>, dim3((n_bound + 31) / 32), dim3(256), RP_TAIL32_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

// ------------------------------------------------------------------------------------------------
// The same half-layer on v_mfma_f32_16x16x32_f16 tiles.  The chip lowers its clock under matrix load and the clock it
// holds depends on the MFMA shape: a bare loop of the 32x32x16 form on random operands sustains 1.61 PFLOP/s here, the
// 16x16x32 form 1.82 (profiles/r02_mfma_sustained_rate.txt).  Same LDS panels, same packed weights (read through
// different lane addresses), same persistent panel loop; what changes:
//  * EVERY product is computed in the orientation the 64-row kernel uses for FFN1 only: the W fragment is the A
//    operand (16 output features x 32 k), the panel fragment the B operand (32 k x 16 panel rows).  Lane (r16 = l & 15,
//    g = l >> 4) then holds features 4g .. 4g+3 of panel row r16 in one accumulator: the residual loads, the x1 -> LDS
//    writes, the hidden-chunk writes and the final stores are 16-byte accesses (8 per lane instead of 32 dwords).
//  * A pass is 8 double k-steps (K = 32) of two half-steps (row tiles {0,1}, {2,3}); a half-step issues 12 MFMAs (192
//    cycles, as one k-step of the 32x32x16 form) from 4 panel fragments and the double step's 4 W fragments.
// W fragment of k-steps (2d, 2d+1), feature tile ft, lane (r16, g): k = 32d + 8g .. +7 lives in the packed image at
// k-step 2d + (g >> 1), lane 16 ft + r16 + 32 (g & 1): four 256-byte runs per wave load.
// ------------------------------------------------------------------------------------------------
// SITE only names the instantiation (0 = decoder, i.e. the edit loop; 1 = encoder, i.e. set-up), so that profiles and
// counter passes can tell the two apart (tools/summarize_pmc.py): the code is the same.
// EMB != 0: the residual rows are embeddings computed in place (OFArgs.emb ..), so that no embed kernel writes x for this
// launch: 1 = decode step (one position, token = the job's last id), 2 = encoder (a token and a position per row)
template <int SITE = 0, int EMB = 0>
__global__ __launch_bounds__(512, 2) void rp_layer_tail16_kernel(OFArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* P1 = smem;
  float* P2 = smem + 64 * ST_LDA;
  int* sIdx = (int*)(smem + 2 * 64 * ST_LDA);
  int* sTok = sIdx + 64;
  int* sPos = sTok + 64;                   // EMB == 2 only (RP_TAIL_LDS_ENC): [2][64], by panel parity - the positions are
                                           // read where the residual is ADDED, when wave 0 may already be staging the next panel
  float* sPe = (float*)(sPos + 128);       // [RP_TAIL_PE_ROWS][ST_LDA]
  int par = 0;                             // parity of this workgroup's current panel
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int mstride = gridDim.x * 64;
  int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int r16 = lane & 15, g = lane >> 4;
  int fcol = wave * 32 + 4 * g;  // this lane's first output feature of feature tile 0 (tile 1: + 16)
  const int nchunk = a.d_ff >> 8;
  TSTAMP(31);

  f32x4 rw[2][4];  // W ring, two double steps deep: [slot][2 * ft + lo]
  auto wfetch = [&](const f32x4* Wl, int d, f32x4 (&dst)[4]) {
    dst[0] = Wl[d * 256];
    dst[1] = Wl[d * 256 + 64];
    dst[2] = Wl[d * 256 + 16];
    dst[3] = Wl[d * 256 + 80];
  };

  int bad = 0;    // an operand left the f16 range (or is NaN): reported once at the end
  f32x4 attr[8];  // this thread's share of an attention panel on its way from HBM to P1
  auto att_load = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 512;
      attr[i] = *(const f32x4*)(a.att + (size_t)min(mb + (f >> 6), n - 1) * 256 + (f & 63) * 4);
    }
  };
  auto att_store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 512;
      panel_store4_nb16(P1 + (f >> 6) * ST_LDA, f & 63, attr[i], bad);  // K-permuted rows: conflict-free fragment reads
    }
  };
  f32x4 ax[2][4], px[2][4], pc[2][4];  // [feature tile][row tile]
  auto idx_stage = [&](int mb, int stage_par) {
    if (EMB == 2) {
      if (tid < 64) {
        const int r = min(mb + tid, n - 1);
        sTok[tid] = min(max(a.tokens[r], 0), a.vocab - 1);
        sPos[64 * stage_par + tid] = min(max(a.tok_pos[r], 0), RP_TAIL_PE_ROWS - 1);
      }
    } else if ((a.cconst || EMB) && tid < 64) {
      const int s = a.idx[min(mb + tid, n - 1)];
      sIdx[tid] = s;
      if (EMB) {
        const int tok = a.pos == 0 ? a.bos : a.ys[(size_t)s * a.T + a.pos - 1];
        sTok[tid] = min(max(tok, 0), a.vocab - 1);
      }
    }
  };
  auto resid_load = [&](int mb) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const float* xr = EMB ? a.emb + (size_t)sTok[16 * rt + r16] * 256 + fcol
                            : a.x + (size_t)min(mb + 16 * rt + r16, n - 1) * 256 + fcol;
      px[0][rt] = *(const f32x4*)xr;
      px[1][rt] = *(const f32x4*)(xr + 16);
    }
    if (a.cconst) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const float* cr = a.cconst + (size_t)sIdx[16 * rt + r16] * 256 + fcol;
        pc[0][rt] = *(const f32x4*)cr;
        pc[1][rt] = *(const f32x4*)(cr + 16);
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) pc[0][rt] = pc[1][rt] = f32x4{0, 0, 0, 0};
    }
  };
  att_load(m0);
  idx_stage(m0, 0);
  if (EMB == 2) {
#pragma unroll
    for (int i = 0; i < RP_TAIL_PE_ROWS * 64 / 512; ++i) {
      const int f = tid + i * 512;
      *(f32x4*)(sPe + (f >> 6) * ST_LDA + (f & 63) * 4) = *(const f32x4*)(a.pe + (size_t)(f >> 6) * 256 + (f & 63) * 4);
    }
  }
  f32x4 b1n[2];  // FFN1 bias of this lane's 8 hidden units in the coming chunk
  auto b1_fetch = [&](int c) {
    b1n[0] = *(const f32x4*)(a.b1 + (c * 8 + wave) * 32 + 4 * g);
    b1n[1] = *(const f32x4*)(a.b1 + (c * 8 + wave) * 32 + 16 + 4 * g);
  };
  att_store();
  __syncthreads();  // sIdx of the first panel
  resid_load(m0);
  bool first = true;
  for (;;) {
  // (lane coordinates from an opaque copy of the thread index: see rp_layer_tail_kernel)
  TSTAMP(35);
  asm volatile("" : "+v"(tid));
  lane = tid & 63;
  r16 = lane & 15;
  g = lane >> 4;
  fcol = wave * 32 + 4 * g;
  const int wlane = (g >> 1) * 128 + 32 * (g & 1) + r16;  // this lane's place inside a double step of a packed block
  const f32x4* Wo = (const f32x4*)a.wo + (size_t)wave * 16 * 128 + wlane;
  auto W1p = [&](int c) { return (const f32x4*)a.w1 + (size_t)(c * 8 + wave) * 16 * 128 + wlane; };                  // hidden block 8c + wave
  auto W2p = [&](int c) { return (const f32x4*)a.w2 + ((size_t)wave * (a.d_ff >> 4) + c * 16) * 128 + wlane; };  // k-steps 16c .. 16c+15 of output block `wave`
  if (first) {
    wfetch(Wo, 0, rw[0]);
    wfetch(Wo, 1, rw[1]);
  }
  first = false;
  {
    const f32x4 bo0 = *(const f32x4*)(a.bo + fcol), bo1 = *(const f32x4*)(a.bo + fcol + 16);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      ax[0][rt] = bo0;
      ax[1][rt] = bo1;
    }
  }
  TSTAMP(0);
  __syncthreads();
  TSTAMP(1);
  const float* A1 = P1 + r16 * ST_LDA + rp16_frag_pos(0, g);  // panel fragment base (row tile 0, double step 0, hi)
  const float* A2 = P2 + r16 * ST_LDA + rp16_frag_pos(0, g);
  // One pass over K = 256: acc[ft][rt] += W[features of tile ft] . panel[rows of tile rt]^T.  With EPI the accumulators
  // o of the previous hidden chunk are converted (ReLU, f16 split) and written to P2 in the MFMAs' shadow.
  _Float16* H2 = (_Float16*)P2;
  auto epi_unit = [&](const f32x4& acc, int ft, int rt) {
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[q], 0.0f);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    bad |= !(mx <= 65504.0f);
    h4 hi, lo;
    split_f16(v, hi, lo);
    _Float16* dst = H2 + (size_t)(16 * rt + r16) * (2 * ST_LDA) + 2 * rp16_quad_pos((fcol + 16 * ft) >> 2);  // hidden unit = k of FFN2
    *(h4*)dst = hi;
    *(h4*)(dst + 256) = lo;
  };
  auto pass = [&](const float* Pb, const f32x4* Wc, const f32x4* Wn, f32x4 (&acc)[2][4], const bool epi, const f32x4 (&o)[2][4]) {
    f32x4 pn[4];  // panel fragments of the next half-step: row tile 2hf (hi, lo), row tile 2hf + 1 (hi, lo)
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
        wfetch(Wn, d + 2 - 8, rw[slot]);  // the ring never drains: the next pass's first double steps
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
        if (epi && hf == 1) epi_unit(o[d & 1][d >> 1], d & 1, d >> 1);  // 8 conversion units over the 8 double steps
        __builtin_amdgcn_sched_barrier(0);  // keep the half-step structure (see rp_layer_tail_kernel)
      }
    }
  };
  pass(A1, Wo, W1p(0), ax, false, ax);
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      if (EMB) {  // x = emb[token] * 16 + pe[pos], the embed kernel's arithmetic
        const f32x4 pev = EMB == 2 ? *(const f32x4*)(sPe + sPos[64 * par + 16 * rt + r16] * ST_LDA + fcol + 16 * ft)
                                   : *(const f32x4*)(a.pe_row + fcol + 16 * ft);
        f32x4 xv;
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = fmaf(px[ft][rt][q], 16.0f, pev[q]);
        ax[ft][rt] += xv + pc[ft][rt];
      } else {
        ax[ft][rt] += px[ft][rt] + pc[ft][rt];
      }
    }
  TSTAMP(2);
  b1_fetch(0);
  if (m0 + mstride < n) idx_stage(m0 + mstride, par ^ 1);  // read by resid_load in the last chunk, many barriers from here
  __syncthreads();  // every wave is done reading the attention panel
  TSTAMP(3);
  // (the LayerNorm gains are fetched per panel - L2 hits, the address hangs on the opaque lane - instead of living in 8
  // registers through the whole loop: the kernel sits at its 256-register limit)
  const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
  const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
  // ---- x1 -> P1 (f32), LayerNorm + split in place ----
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) *(f32x4*)(P1 + (16 * rt + r16) * ST_LDA + fcol + 16 * ft) = ax[ft][rt];
  TSTAMP(32);
  __syncthreads();
  TSTAMP(33);
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    float* row = P1 + (wave * 8 + rr) * ST_LDA;
    panel_store4_nb16(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb), bad);
  }
  {
    const f32x4 b20 = *(const f32x4*)(a.b2 + fcol), b21 = *(const f32x4*)(a.b2 + fcol + 16);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      ax[0][rt] += b20;
      ax[1][rt] += b21;
    }
  }
  TSTAMP(34);
  __syncthreads();
  TSTAMP(4);
  auto ah_init = [&](f32x4 (&h)[2][4]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      h[0][rt] = b1n[0];
      h[1][rt] = b1n[1];
    }
  };
  // ---- FFN1 of hidden chunk 0 (hidden block 8c + wave) ----
  f32x4 ah[2][4];
  ah_init(ah);
  if (nchunk > 1) b1_fetch(1);  // consumed one chunk later
  TSTAMP(5);
  pass(A1, W1p(0), nchunk > 1 ? W1p(1) : W2p(0), ah, false, ah);
  TSTAMP(6);
  for (int c = 0; c + 1 < nchunk; ++c) {
    f32x4 an[2][4];
    ah_init(an);
    if (c + 2 < nchunk) b1_fetch(c + 2);
    if (c > 0) __syncthreads();  // every wave is done reading the previous hidden chunk
    // ---- hidden chunk c -> P2, overlapped with FFN1 of chunk c + 1 ----
    pass(A1, W1p(c + 1), W2p(c), an, true, ah);
    TSTAMP(7 + c * 5);
    __syncthreads();  // hidden chunk complete
    TSTAMP(8 + c * 5);
    // ---- FFN2 partial: x += h_c . W2[:, chunk]^T ----
    pass(A2, W2p(c), c + 2 < nchunk ? W1p(c + 2) : W2p(c + 1), ax, false, ax);
    TSTAMP(9 + c * 5);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) ah[ft][rt] = an[ft][rt];
  }
  {  // last chunk (peeled, with the next panel's loads: see rp_layer_tail_kernel)
    const int c = nchunk - 1;
    att_load(m0 + mstride);
    resid_load(m0 + mstride);
    if (c > 0) __syncthreads();
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) epi_unit(ah[ft][rt], ft, rt);
    TSTAMP(7 + c * 5);
    __syncthreads();  // hidden chunk complete
    TSTAMP(8 + c * 5);
    pass(A2, W2p(c), Wo, ax, false, ax);  // (a workgroup's last panel requests Wo once more and drops it)
    TSTAMP(9 + c * 5);
  }
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int row = m0 + 16 * rt + r16;
    if (row < n) {
      __builtin_nontemporal_store(ax[0][rt], (f32x4*)(a.x + (size_t)row * 256 + fcol));
      __builtin_nontemporal_store(ax[1][rt], (f32x4*)(a.x + (size_t)row * 256 + fcol + 16));
    }
  }
  TSTAMP(30);
  m0 += mstride;
  par ^= 1;
  if (m0 >= n) break;
  att_store();
  TSTAMP(36);
  }  // panels of this workgroup
  if (bad && a.range_flag) *a.range_flag = 1;
}

static inline int st_cu_count() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cus[dev];
}
static int launch_layer_tail64(const OFArgs& a, int n_bound, hipStream_t st) {
  ST_ENSURE_LDS((rp_layer_tail_kernel<0>), RP_TAIL_LDS);
  if (n_bound <= 0) return ST_OK;
  if (a.fmt != ST_GEMM_F16X3 || a.d_ff % 256 != 0 || !a.w2 || !a.b2) {
    st_set_error("rp_layer_tail: needs the f16x3 format, d_ff %% 256 == 0 and w2/b2 (fmt=%d d_ff=%d)", a.fmt, a.d_ff);
    return ST_EINVAL;
  }
  int grid = (n_bound + 63) / 64;
  if (rp_tail_persistent) grid = std::min(grid, st_cu_count());  // 133 KiB of LDS: one workgroup per CU, each loops over its panels
  hipLaunchKernelGGL(rp_layer_tail_kernel<0>, dim3(grid), dim3(512), RP_TAIL_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

template <int SITE = 0, int EMB = 0>
static int launch_layer_tail16(const OFArgs& a, int n_bound, hipStream_t st) {
  constexpr int LDS = EMB == 2 ? RP_TAIL_LDS_ENC : RP_TAIL_LDS;
  ST_ENSURE_LDS((rp_layer_tail16_kernel<SITE, EMB>), LDS);
  if (n_bound <= 0) return ST_OK;
  if (a.fmt != ST_GEMM_F16X3 || a.d_ff % 256 != 0 || !a.w2 || !a.b2) {
    st_set_error("rp_layer_tail16: needs the f16x3 format, d_ff %% 256 == 0 and w2/b2 (fmt=%d d_ff=%d)", a.fmt, a.d_ff);
    return ST_EINVAL;
  }
  int grid = (n_bound + 63) / 64;
  if (rp_tail_persistent) grid = std::min(grid, st_cu_count());
  hipLaunchKernelGGL((rp_layer_tail16_kernel<SITE, EMB>), dim3(grid), dim3(512), LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

// The library uses the 64-row kernel: the 32-row / two-workgroups-per-CU form measured 8 % SLOWER (kbench A/B at 131072
// and 229376 rows, bit-identical outputs; profiles/r02_kbench_tail_ab.txt): it hides the panel load, LayerNorm, store and
// turnaround phases as intended, but every W fragment feeds three MFMAs instead of six and the doubled L2 -> CU weight
// stream costs more than the overlap gains.  launch_layer_tail32 stays reachable from tools/kbench.cpp only.
template <int SITE = 0>
static int launch_layer_tail(const OFArgs& a, int n_bound, hipStream_t st) {
  rp_ab_env();
  if (a.emb) {
    if (!rp_tail_shape16) {
      st_set_error("rp_layer_tail: the embedding residual needs the 16x16x32 form");
      return ST_EINVAL;
    }
    return a.tokens ? launch_layer_tail16<SITE, 2>(a, n_bound, st) : launch_layer_tail16<SITE, 1>(a, n_bound, st);
  }
  return rp_tail_shape16 ? launch_layer_tail16<SITE>(a, n_bound, st) : launch_layer_tail64(a, n_bound, st);
}

// packed-weight helpers -------------------------------------------------------------------------
static inline size_t rp_packed_floats(int N, int K) { return (size_t)((N + 31) / 32) * 32 * K; }

// f16x3 image: one thread per 8 elements of W -> a hi and a lo h8
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void pack_weight_h_kernel(const float* __restrict__ W, int N, int K,
                                                            float* __restrict__ out) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;  // (block, step, lane)
  const int steps = K >> 4;
  const size_t total = (size_t)((N + 31) / 32) * steps * 64;
  if (o >= total) return;
  const int lane = (int)(o & 63);
  const int step = (int)((o >> 6) % steps);
  const int blk = (int)((o >> 6) / steps);
  const int row = blk * 32 + (lane & 31);
  f32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
  if (row < N) {
    const float* src = W + (size_t)row * K + step * 16 + 8 * (lane >> 5);
    v0 = *(const f32x4*)src;
    v1 = *(const f32x4*)(src + 4);
  }
  h4 h0, l0, h1, l1;
  split_f16(v0, h0, l0);
  split_f16(v1, h1, l1);
  h8 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = h0[j];
    hi[4 + j] = h1[j];
    lo[j] = l0[j];
    lo[4 + j] = l1[j];
  }
  h8* dst = (h8*)out + (((size_t)blk * steps + step) * 2) * 64 + lane;
  dst[0] = hi;
  dst[64] = lo;
}

template <int UNUSED = 0>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ W, int N, int K,
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
  if (row < N) v = *(const f32x4*)(W + (size_t)row * K + kg * 8 + 4 * (lane >> 5));
  *(f32x4*)(out + o * 4) = v;
}
