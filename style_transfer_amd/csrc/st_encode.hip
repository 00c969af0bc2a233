// Batched, length-packed Transformer encoder -> latent (replaces EncoderDecoder.forward's encoder
// half / get_latent, /root/reference/method/yelp/model.py:229-240, :253-257, Encoder :137-163, for the
// 1 + 10 + 10 sentences the reference encodes one at a time, siamese_contrastive.py:706-711).
// Rows = non-pad tokens of all sentences, packed back to back: padding never enters a product
// (pad keys are masked to exp(-1e9)=0 in the reference and pad positions are dropped from the
// latent sum by src_mask, model.py:240), so the packed result equals the padded one.
#include "st_rowpanel.h"

__global__ __launch_bounds__(256) void enc_embed_kernel(const float* __restrict__ emb, const float* __restrict__ pe,
                                                        const int32_t* __restrict__ tok, const int32_t* __restrict__ pos,
                                                        int R, int vocab, float* __restrict__ x) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const int t = min(max(tok[r], 0), vocab - 1);
  const f32x4 e = *(const f32x4*)(emb + (size_t)t * 256 + lane * 4);
  const f32x4 p = *(const f32x4*)(pe + (size_t)pos[r] * 256 + lane * 4);
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = fmaf(e[q], 16.0f, p[q]);  // (rp_layer_tail16_kernel<1, 2> repeats exactly this)
  *(f32x4*)(x + (size_t)r * 256 + lane * 4) = v;
}

// Full (non-causal) self-attention inside each sentence of 17..64 tokens (amazon): block = sentence, wave = head, on
// the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), operands straight from the qkv rows in HBM / L2 (no LDS):
//   S^T tile [32 keys x 32 queries] = K . Q^T : lane (l31, half) supplies row l31 of K (A operand) and of Q (B operand),
//     dims half*32 .. half*32+31 of the head (8 coalesced float4 loads each); step kk contracts the dim pair
//     {kk, 32 + kk}, so 32 steps cover the 64 dims.  The lane then holds 16 keys of ONE query (column l31): the
//     softmax over keys is lane-local plus one exchange with lane ^ 32 for the maximum and one for the sum.
//   O tile [32 queries x 32 dims] = P . V : the accumulator register r of the S^T tile IS the A operand of step r
//     (keys {row(r, 0), row(r, 1)} of the tile), B = V[key][dim l31] is one dword load per step.
// P is normalised before the product, as the reference multiplies softmax(scores) by V (model.py attention()).
// qkv rows: [q | k | v] (768 floats).  The previous VALU kernel (keys in registers, one query at a time, 12 LDS-crossbar
// shuffles per query) ran the amazon shape at 1.43 ms per launch; this one is bound by the qkv row traffic.
// Where the [q | k | v] rows of a launch live: the projected rows (qkv, row r at r * 768), or - first layer with the
// (token, position) table - the table itself, row r at (tok[r] * tab_len + pos[r]) * 768: the attention kernels read the
// table rows where they lie (a gather into qkv first wrote and re-read 3 KB per token: 3.6 % of the amazon-shaped step).
struct EncRowSrc {
  const float* qkv;
  const float* tab;
  int tab_len, vocab;
  const int32_t* tok;
  const int32_t* pos;
};
__device__ __forceinline__ int enc_row_index(const EncRowSrc& s, int r) {
  if (!s.tab) return r;
  const int t = min(max(s.tok[r], 0), s.vocab - 1), p = min(max(s.pos[r], 0), s.tab_len - 1);
  return t * s.tab_len + p;  // (< 2^31: the table is at most 8 GB of 3 KB rows)
}
template <bool FAST>  // as in enc_attn16_kernel below
__global__ __launch_bounds__(256) void enc_attn_kernel(EncRowSrc src, const int32_t* __restrict__ off, int n_rows, float* __restrict__ out) {
  __shared__ int rowix[64];
  const int s = blockIdx.x;
  const int r0 = off[s], len = off[s + 1] - r0;
  if (len <= 16 || len > 64) return;  // short sentences: enc_attn16_kernel
  if (r0 < 0 || r0 + len > n_rows) return;  // bad offsets (flagged by enc_check_len_kernel): touch nothing
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  if (threadIdx.x < 64) rowix[threadIdx.x] = enc_row_index(src, r0 + min((int)threadIdx.x, len - 1));
  __syncthreads();
  const float* base = (src.tab ? src.tab : src.qkv) + h * 64;  // row j of the sentence: base + rowix[j] * 768
  const int nt = (len + 31) >> 5;  // 32-row tiles of queries and of keys: 1 or 2
  f32x4 kf[2][8];
#pragma unroll
  for (int kj = 0; kj < 2; ++kj)
    if (kj < nt) {
      const float* kp = base + (size_t)rowix[min(kj * 32 + l31, len - 1)] * 768 + 256 + half * 32;
#pragma unroll
      for (int c = 0; c < 8; ++c) kf[kj][c] = *(const f32x4*)(kp + c * 4);
    }
  for (int qi = 0; qi < nt; ++qi) {
    f32x4 qf[8];
    {
      const float* qp = base + (size_t)rowix[min(qi * 32 + l31, len - 1)] * 768 + half * 32;
#pragma unroll
      for (int c = 0; c < 8; ++c) qf[c] = *(const f32x4*)(qp + c * 4);
    }
    f32x16 sacc[2];
    float m = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
      if (kj < nt) {
        f32x16 acc = {0};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = mfma32(kf[kj][c][e], qf[c][e], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kj * 32 + mfma32_row(r, lane);
          acc[r] = key < len ? acc[r] * 0.125f : -INFINITY;  // / sqrt(d_k); keys beyond the sentence do not exist
          m = fmaxf(m, acc[r]);
        }
        sacc[kj] = acc;
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
      if (kj < nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = FAST ? __expf(sacc[kj][r] - m) : expf(sacc[kj][r] - m);
          sacc[kj][r] = e;
          sum += e;
        }
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x16 oacc = {0};
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
        if (kj < nt) {
          float vv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            vv[r] = base[(size_t)rowix[min(kj * 32 + mfma32_row(r, lane), len - 1)] * 768 + 512 + dt * 32 + l31];
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc = mfma32(sacc[kj][r] * inv, vv[r], oacc);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qi * 32 + mfma32_row(r, lane);
        if (q < len) out[(size_t)(r0 + q) * 256 + h * 64 + dt * 32 + l31] = oacc[r];
      }
    }
  }
}

// Sentences of <= 16 tokens (all of yelp): ONE wave per sentence handles the four heads.  Lane l holds dims
// 4l..4l+3 (head l >> 4) of every K and V row of the sentence in registers (coalesced 1 KiB row loads); for each
// query the q.k partial sums are reduced over the 16 lanes of a head with DPP row rotations, so the softmax and
// the PV sum need no further lane traffic (same scheme as attn_step16_kernel in st_decode.hip).
// FAST (f16x3 format): exp through v_exp_f32 (x * log2 e, ~1 ulp + |x| 2^-24) instead of the library's correctly
// rounded expf, whose ~20 instructions per call were a quarter of this kernel (encoder setup 87 -> 84 ms per step);
// the exact-f32 format keeps expf.
template <bool FAST>
__global__ __launch_bounds__(256) void enc_attn16_kernel(EncRowSrc src, const int32_t* __restrict__ off, int n_sent, int n_rows,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_sent) return;
  const int r0 = off[s], len = off[s + 1] - r0;
  if (len <= 0 || len > 16) return;
  if (r0 < 0 || r0 + len > n_rows) return;  // bad offsets (flagged by enc_check_len_kernel): touch nothing
  const float* base = (src.tab ? src.tab : src.qkv) + lane * 4;
  auto row = [&](int j) { return base + (size_t)enc_row_index(src, r0 + j) * 768; };  // (wave-uniform index)
  f32x4 kr[16], vr[16];
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < len) {  // uniform
      const float* rj = row(j);
      kr[j] = *(const f32x4*)(rj + 256);
      vr[j] = *(const f32x4*)(rj + 512);
    }
  f32x4 qn = *(const f32x4*)row(0);
  for (int i = 0; i < len; ++i) {
    const f32x4 qv = qn;
    if (i + 1 < len) qn = *(const f32x4*)row(i + 1);
    float sc[16];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      sc[j] = -INFINITY;
      if (j < len) {
        const float part = fmaf(qv.x, kr[j].x, fmaf(qv.y, kr[j].y, fmaf(qv.z, kr[j].z, qv.w * kr[j].w)));
        sc[j] = row16_sum(part) * 0.125f;  // / sqrt(d_k)
        m = fmaxf(m, sc[j]);
      }
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < len) {
        sc[j] = FAST ? __expf(sc[j] - m) : expf(sc[j] - m);
        sum += sc[j];
      }
    const float inv = 1.0f / sum;
    f32x4 o4 = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < len) o4 += (sc[j] * inv) * vr[j];
    *(f32x4*)(out + (size_t)(r0 + i) * 256 + lane * 4) = o4;
  }
}

// ------------------------------------------------------------------------------------------------
// QKV projection + self-attention of one encoder layer in ONE kernel (f16x3 format, sentences of <= 16 tokens, i.e.
// all of yelp): q, k, v never go to HBM.  The two-kernel form (rp_lin16_kernel<EPI_BIAS> + enc_attn16_kernel) writes
// 3 KB per token and reads them back: 28 + 18 ms of a 354 ms bench step, with the QKV kernel running at memory time PLUS
// MFMA time.
// A workgroup OWNS the sentences whose first row lies in its window of `win` = 65 - max_sent_len rows (>= 49): their rows
// [r0, r1) are at most win - 1 + max_sent_len = 64, one panel, no sentence is split and none is computed twice
// (enc_window_kernel gives every window its first sentence).  Per head h: the 192 columns [q_h | k_h | v_h] of the
// stacked projection go through the MFMAs (wave w: 48 columns = 3 feature tiles, row tiles {0,1} or {2,3}; W as the A
// operand, K-permuted panel rows, three products per term, as rp_lin16_kernel) into an f32 LDS buffer, then the eight
// waves run the attention of 4 query rows at a time: lane (qs, j) scores key j of query qs (64-term dot product from
// LDS; rows of one sentence broadcast), softmax over the 16 lanes of a DPP row, the probabilities pass through a small
// LDS buffer and lane (qs, t) accumulates dims 4t..4t+3 of the output.  The next head's MFMAs start before the
// slower waves have finished their queries (the barrier sits in front of the epilogue stores, not in front of the loop).
// ------------------------------------------------------------------------------------------------
#define EQA_LD 72  /* dwords per row of the head buffers: 64 hi halves | 64 lo halves | 8 of padding (= 8 mod 64: conflict-free b128 fragment reads) */
#define EQA_LDS ((64 * ST_LDA + 3 * 64 * EQA_LD + 2 * 64) * 4)

// first_sent[w] = first sentence whose first row is >= w * win, for w = 0 .. n_win (first_sent[n_win] = n_sent)
__global__ __launch_bounds__(256) void enc_window_kernel(const int32_t* __restrict__ off, int n_sent, int win, int n_win,
                                                         int32_t* __restrict__ first_sent) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s > n_sent) return;
  // (offsets are the caller's and enc_check_len_kernel's verdict is read only at the end of the call: clamp, so that a
  // negative or non-monotonic offset can never index first_sent[] out of range)
  const int wa = s < n_sent ? min(max(off[s], 0) / win, n_win) : n_win;        // window of this sentence's first row (sentinel: n_win)
  const int wb = s > 0 ? min(max(off[s - 1], 0) / win, n_win) : -1;            // window of the previous sentence's first row
  for (int w = wb + 1; w <= wa; ++w) first_sent[w] = s;                // windows (wb, wa] start with sentence s
}

__device__ __forceinline__ float row16_max(float v) {
#define ST_ROW_MAX(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false)))
  ST_ROW_MAX(0x121);  // row_ror:1
  ST_ROW_MAX(0x122);
  ST_ROW_MAX(0x124);
  ST_ROW_MAX(0x128);
#undef ST_ROW_MAX
  return v;
}

struct EqaArgs {
  const float* x;          // [R,256]
  const int32_t* off;      // [n_sent + 1]
  const int32_t* first_sent;  // [n_win + 1]
  const float *ln_a, *ln_b;
  const float* W;          // packed [768,256], f16x3
  const float* bias;       // [768]
  float* att;              // [R,256]
  int* range_flag;
  int R, n_sent, n_win;
};

template <int UNUSED = 0>
__global__ __launch_bounds__(512) void enc_qkv_attn16_kernel(EqaArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                      // [64][ST_LDA] LayerNorm'ed split panel
  float* Qs = smem + 64 * ST_LDA;        // [64 rows][EQA_LD] q of the current head, f16 hi | lo
  float* Ks = Qs + 64 * EQA_LD;          // [64 rows][EQA_LD] k
  float* Vt = Ks + 64 * EQA_LD;          // [64 dims][EQA_LD] v TRANSPOSED: row = dim, column = panel row (key)
  int* slist = (int*)(Vt + 64 * EQA_LD);  // [64] the window's non-empty sentences: first panel row | length << 8
  int* scount = slist + 64;              // their number
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // persistent over windows (one workgroup per CU: 117 KiB of LDS): the next window's rows travel from HBM into
  // registers while this window computes
  struct Win { int s_lo, s_hi, r0, nr; };
  auto window = [&](int p) -> Win {
    Win w = {0, 0, 0, 0};
    if (p < a.n_win) {
      w.s_lo = a.first_sent[p];
      w.s_hi = a.first_sent[p + 1];
      if (w.s_lo < w.s_hi) {
        w.r0 = a.off[w.s_lo];
        w.nr = a.off[w.s_hi] - w.r0;
        // (enc_check_len_kernel flags bad offsets; until the host reads that flag nothing may be touched through them)
        if (w.nr < 0 || w.nr > 64 || w.r0 < 0 || w.r0 + w.nr > a.R || w.s_lo < 0 || w.s_hi > a.n_sent) w.nr = 0;
      }
    }
    return w;
  };
  f32x4 xr[8];
  auto xload = [&](const Win& w) {
    if (w.nr > 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = tid + i * 512;
        xr[i] = *(const f32x4*)(a.x + (size_t)(w.r0 + min(f >> 6, w.nr - 1)) * 256 + (f & 63) * 4);
      }
    }
  };
  const int w4 = wave & 3, rh = wave >> 2;  // 48 columns of the head's 192, row tiles 2rh and 2rh + 1
  const unsigned wl = (unsigned)(((g >> 1) * 128 + 32 * (g & 1) + r16) * 16);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, 24 * 32768, 0x00020000);
  // feature tile f of head h: global column, and the byte offset of its hi fragments inside the packed image
  auto gcol = [&](int h, int f) -> int {
    const int c = 48 * w4 + 16 * f;
    return 256 * (c >> 6) + 64 * h + (c & 63);
  };
  auto wofs = [&](int h, int f) -> int {
    const int c = gcol(h & 3, f);
    return (c >> 5) * 32768 + ((c >> 4) & 1) * 256;
  };
  auto wfetch = [&](int h, int d, f32x4 (&dst)[6]) {
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const int o = wofs(h, f) + d * 4096;
      dst[2 * f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl, o, 0));
      dst[2 * f + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1024, o, 0));
    }
  };
  const float* Pb = As + (32 * rh + r16) * ST_LDA + rp16_frag_pos(0, g);
  const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
  const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
  int bad = 0;
  f32x4 rw[2][6];
  wfetch(0, 0, rw[0]);
  wfetch(0, 1, rw[1]);
  Win cur = window(blockIdx.x);
  xload(cur);
  for (int p = blockIdx.x; p < a.n_win; p += gridDim.x) {
    const Win nxt = window(p + gridDim.x);
    const int r0 = cur.r0, nr = cur.nr;
    if (nr > 0) {  // (uniform)
      // ---- panel rows -> LDS, sentence bounds, LayerNorm, f16 split (K-permuted rows) ----
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = tid + i * 512;
        *(f32x4*)(As + (f >> 6) * ST_LDA + (f & 63) * 4) = xr[i];
      }
      if (tid == 0) *scount = 0;
      __syncthreads();
      xload(nxt);  // in flight during this window's products
      for (int s = cur.s_lo + tid; s < cur.s_hi; s += 512) {
        const int b = a.off[s] - r0, e = a.off[s + 1] - r0;
        if (e > b && e <= 64) slist[atomicAdd(scount, 1)] = b | ((e - b) << 8);  // (at most 64: every one has a row)
      }
#pragma unroll 4
      for (int rr = 0; rr < 8; ++rr) {
        float* row = As + (wave * 8 + rr) * ST_LDA;  // one wave owns the whole row: in-place is safe
        panel_store4_nb16(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb), bad);
      }
      __syncthreads();
      for (int h = 0; h < 4; ++h) {
        f32x4 acc[3][2];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          const f32x4 b4 = *(const f32x4*)(a.bias + gcol(h, f) + 4 * g);
          acc[f][0] = b4;
          acc[f][1] = b4;
        }
        f32x4 pn[4];
        pn[0] = *(const f32x4*)(Pb);
        pn[1] = *(const f32x4*)(Pb + 128);
        pn[2] = *(const f32x4*)(Pb + 16 * ST_LDA);
        pn[3] = *(const f32x4*)(Pb + 16 * ST_LDA + 128);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const int sl = d & 1;
          h8 wh[3], wo[3];
#pragma unroll
          for (int f = 0; f < 3; ++f) {
            wh[f] = as_h8(rw[sl][2 * f]);
            wo[f] = as_h8(rw[sl][2 * f + 1]);
          }
          if (d + 2 < 8)
            wfetch(h, d + 2, rw[sl]);
          else
            wfetch(h + 1, d + 2 - 8, rw[sl]);  // the ring never drains: head 4 = head 0 of the next window
          const h8 p0h = as_h8(pn[0]), p0l = as_h8(pn[1]), p1h = as_h8(pn[2]), p1l = as_h8(pn[3]);
          if (d < 7) {
            const float* q = Pb + 8 * (d + 1);
            pn[0] = *(const f32x4*)(q);
            pn[1] = *(const f32x4*)(q + 128);
            pn[2] = *(const f32x4*)(q + 16 * ST_LDA);
            pn[3] = *(const f32x4*)(q + 16 * ST_LDA + 128);
          }
#pragma unroll
          for (int f = 0; f < 3; ++f) {
            acc[f][0] = mfma16x32h(wh[f], p0h, acc[f][0]);
            acc[f][1] = mfma16x32h(wh[f], p1h, acc[f][1]);
          }
#pragma unroll
          for (int f = 0; f < 3; ++f) {
            acc[f][0] = mfma16x32h(wo[f], p0h, acc[f][0]);
            acc[f][1] = mfma16x32h(wo[f], p1h, acc[f][1]);
          }
#pragma unroll
          for (int f = 0; f < 3; ++f) {
            acc[f][0] = mfma16x32h(wh[f], p0l, acc[f][0]);
            acc[f][1] = mfma16x32h(wh[f], p1l, acc[f][1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (h > 0) __syncthreads();  // every wave is done with the previous head's attention: the head buffers may be overwritten
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          const int c = 48 * w4 + 16 * f;  // (wave-uniform) column inside [q | k | v]; the lane adds 4 g
          const int d0 = (c & 63) + 4 * g;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = 32 * rh + 16 * j + r16;
            h4 hi, lo;
            split_f16(acc[f][j], hi, lo);
            if ((c >> 6) < 2) {
              _Float16* dst = (_Float16*)(((c >> 6) ? Ks : Qs) + row * EQA_LD) + d0;
              *(h4*)dst = hi;
              *(h4*)(dst + 64) = lo;
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                _Float16* dst = (_Float16*)(Vt + (d0 + i) * EQA_LD) + row;
                dst[0] = hi[i];
                dst[64] = lo[i];
              }
            }
          }
        }
        __syncthreads();  // q | k | v^T of head h complete
        // ---- attention of head h on the matrix cores, one sentence per wave and pass, the sentence's rows at CANONICAL
        // tile positions (key j = row j of the A tile, whatever the sentence's place in the panel: a sentence's result
        // does not depend on its neighbours, bit for bit).
        //   S^T tile = K . Q^T (A = the sentence's K rows, B = its Q rows: same fragment form as a panel): lane (r16, g)
        //   gets keys 4g .. 4g+3 of query r16; softmax over the lane's 4 values and the 4 lanes of the query;
        //   O^T = V^T . P: the lane's own probabilities ARE its B fragment (k = 4g .. 4g+3, upper half of the k-step 0),
        //   A = V^T rows read at the sentence's columns. ----
        {
          const int ns = *scount;
          for (int si = wave; si < ns; si += 8) {
            const int ent = slist[si], first = ent & 255, len = ent >> 8;  // (wave-uniform)
            const int row = min(first + r16, 63);                          // rows beyond the sentence: masked below
            const float* qb = Qs + row * EQA_LD + 4 * g;
            const h8 qh0 = as_h8(*(const f32x4*)(qb)), qh1 = as_h8(*(const f32x4*)(qb + 16));
            const h8 ql0 = as_h8(*(const f32x4*)(qb + 32)), ql1 = as_h8(*(const f32x4*)(qb + 48));
            const float* kb = Ks + row * EQA_LD + 4 * g;
            const h8 kh0 = as_h8(*(const f32x4*)(kb)), kh1 = as_h8(*(const f32x4*)(kb + 16));
            const h8 kl0 = as_h8(*(const f32x4*)(kb + 32)), kl1 = as_h8(*(const f32x4*)(kb + 48));
            f32x4 S = {0, 0, 0, 0};
            S = mfma16x32h(kh0, qh0, S);
            S = mfma16x32h(kh1, qh1, S);
            S = mfma16x32h(kl0, qh0, S);
            S = mfma16x32h(kl1, qh1, S);
            S = mfma16x32h(kh0, ql0, S);
            S = mfma16x32h(kh1, ql1, S);
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              S[i] = 4 * g + i < len ? S[i] * 0.125f : -INFINITY;  // / sqrt(d_k); keys beyond the sentence
              m = fmaxf(m, S[i]);
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));  // (key 0 always exists: m is finite)
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              S[i] = 4 * g + i < len ? __expf(S[i] - m) : 0.0f;
              sum += S[i];
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            h4 ph, pl;
            split_f16(S * (1.0f / sum), ph, pl);
            const h4 z4 = {0, 0, 0, 0};
            const h8 bh = __builtin_shufflevector(ph, z4, 0, 1, 2, 3, 4, 5, 6, 7);
            const h8 bl = __builtin_shufflevector(pl, z4, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              const _Float16* vb = (const _Float16*)(Vt + (16 * dt + r16) * EQA_LD);
              h8 ah, al;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int key = min(first + 4 * g + i, 63);
                ah[i] = vb[key];
                al[i] = vb[64 + key];
                ah[4 + i] = (_Float16)0.0f;
                al[4 + i] = (_Float16)0.0f;
              }
              f32x4 o = {0, 0, 0, 0};
              o = mfma16x32h(ah, bh, o);
              o = mfma16x32h(al, bh, o);
              o = mfma16x32h(ah, bl, o);
              if (r16 < len) *(f32x4*)(a.att + (size_t)(r0 + first + r16) * 256 + 64 * h + 16 * dt + 4 * g) = o;
            }
          }
        }
      }
      __syncthreads();  // the last head's attention is done before the next window overwrites the panel and hb
    } else {
      xload(nxt);
    }
    cur = nxt;
  }
  if (bad && a.range_flag) *a.range_flag = 1;
}

// ------------------------------------------------------------------------------------------------
// First encoder layer with the (token, position) table (StEncoderWeights.l0_table): the layer's input is
// LayerNorm(emb[token] * 16 + pe[pos]), so its q | k | v rows depend on (token, position) only and are computed once per
// checkpoint (st_encode_build_l0_table, with rp_lin16_kernel: the values enc_qkv_attn16_kernel's projection produces, bit
// for bit).  enc_tab_attn16_kernel is enc_qkv_attn16_kernel with the projection replaced by a gather: per head the
// window's q | k | v rows travel from the table into the SAME split-f16 head buffers, and the attention block below is
// that kernel's, instruction for instruction: identical latents.  No panel, no LayerNorm, no MFMA projection; 56 KiB of
// LDS, two workgroups per CU; the table is shared by all sentences (frequent rows come from L2 / Infinity Cache).
// ------------------------------------------------------------------------------------------------
#define ETA_LDS ((3 * 64 * EQA_LD + 2 * 64) * 4)
struct EtaArgs {
  const float* tab;        // [vocab][tab_len][768]
  int tab_len, vocab;
  const int32_t* tokens;   // [R]
  const int32_t* tok_pos;  // [R]
  const int32_t* off;      // [n_sent + 1]
  const int32_t* first_sent;
  float* att;              // [R,256]
  int R, n_sent, n_win;
};
__global__ __launch_bounds__(512, 4) void enc_tab_attn16_kernel(EtaArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;                      // [64 rows][EQA_LD] q of the current head, f16 hi | lo
  float* Ks = Qs + 64 * EQA_LD;          // [64 rows][EQA_LD] k
  float* Vt = Ks + 64 * EQA_LD;          // [64 dims][EQA_LD] v TRANSPOSED
  int* slist = (int*)(Vt + 64 * EQA_LD);
  int* scount = slist + 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // gather role of this thread: panel row tid >> 3, six float4 of the head's 48 (q: 0..15, k: 16..31, v: 32..47)
  const int grow = tid >> 3, gpart = tid & 7;
  for (int p = blockIdx.x; p < a.n_win; p += gridDim.x) {
    const int s_lo = a.first_sent[p], s_hi = a.first_sent[p + 1];
    if (s_lo >= s_hi) continue;  // (uniform)
    const int r0 = a.off[s_lo];
    int nr = a.off[s_hi] - r0;
    if (nr <= 0 || nr > 64 || r0 < 0 || r0 + nr > a.R || s_lo < 0 || s_hi > a.n_sent) continue;  // (bad offsets are flagged elsewhere)
    if (tid == 0) *scount = 0;
    const int rr = r0 + min(grow, nr - 1);
    const int tk = min(max(a.tokens[rr], 0), a.vocab - 1);
    const int ps = min(max(a.tok_pos[rr], 0), a.tab_len - 1);
    const float* trow = a.tab + ((size_t)tk * a.tab_len + ps) * 768;
    __syncthreads();  // the previous window's attention is done (head buffers, slist), scount is reset
    for (int s = s_lo + tid; s < s_hi; s += 512) {
      const int b = a.off[s] - r0, e = a.off[s + 1] - r0;
      if (e > b && e <= 64) slist[atomicAdd(scount, 1)] = b | ((e - b) << 8);
    }
    f32x4 tv[6];
    auto tload = [&](int h) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int c4 = gpart * 6 + i;
        tv[i] = *(const f32x4*)(trow + 256 * (c4 >> 4) + 64 * h + 4 * (c4 & 15));
      }
    };
    tload(0);
    for (int h = 0; h < 4; ++h) {
      if (h > 0) __syncthreads();  // every wave is done with the previous head's attention
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int c4 = gpart * 6 + i;
        const int d0 = 4 * (c4 & 15);
        h4 hi, lo;
        split_f16(tv[i], hi, lo);
        if ((c4 >> 4) < 2) {
          _Float16* dst = (_Float16*)(((c4 >> 4) ? Ks : Qs) + grow * EQA_LD) + d0;
          *(h4*)dst = hi;
          *(h4*)(dst + 64) = lo;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            _Float16* dst = (_Float16*)(Vt + (d0 + j) * EQA_LD) + grow;
            dst[0] = hi[j];
            dst[64] = lo[j];
          }
        }
      }
      if (h < 3) tload(h + 1);  // in flight during this head's attention
      __syncthreads();  // q | k | v^T of head h complete
      {
        const int ns = *scount;
        for (int si = wave; si < ns; si += 8) {
          const int ent = slist[si], first = ent & 255, len = ent >> 8;  // (wave-uniform)
          const int row = min(first + r16, 63);
          const float* qb = Qs + row * EQA_LD + 4 * g;
          const h8 qh0 = as_h8(*(const f32x4*)(qb)), qh1 = as_h8(*(const f32x4*)(qb + 16));
          const h8 ql0 = as_h8(*(const f32x4*)(qb + 32)), ql1 = as_h8(*(const f32x4*)(qb + 48));
          const float* kb = Ks + row * EQA_LD + 4 * g;
          const h8 kh0 = as_h8(*(const f32x4*)(kb)), kh1 = as_h8(*(const f32x4*)(kb + 16));
          const h8 kl0 = as_h8(*(const f32x4*)(kb + 32)), kl1 = as_h8(*(const f32x4*)(kb + 48));
          f32x4 S = {0, 0, 0, 0};
          S = mfma16x32h(kh0, qh0, S);
          S = mfma16x32h(kh1, qh1, S);
          S = mfma16x32h(kl0, qh0, S);
          S = mfma16x32h(kl1, qh1, S);
          S = mfma16x32h(kh0, ql0, S);
          S = mfma16x32h(kh1, ql1, S);
          float m = -INFINITY;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            S[i] = 4 * g + i < len ? S[i] * 0.125f : -INFINITY;
            m = fmaxf(m, S[i]);
          }
          m = fmaxf(m, __shfl_xor(m, 16, 64));
          m = fmaxf(m, __shfl_xor(m, 32, 64));
          float sum = 0.0f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            S[i] = 4 * g + i < len ? __expf(S[i] - m) : 0.0f;
            sum += S[i];
          }
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          h4 ph, pl;
          split_f16(S * (1.0f / sum), ph, pl);
          const h4 z4 = {0, 0, 0, 0};
          const h8 bh = __builtin_shufflevector(ph, z4, 0, 1, 2, 3, 4, 5, 6, 7);
          const h8 bl = __builtin_shufflevector(pl, z4, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const _Float16* vb = (const _Float16*)(Vt + (16 * dt + r16) * EQA_LD);
            h8 ah, al;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int key = min(first + 4 * g + i, 63);
              ah[i] = vb[key];
              al[i] = vb[64 + key];
              ah[4 + i] = (_Float16)0.0f;
              al[4 + i] = (_Float16)0.0f;
            }
            f32x4 o = {0, 0, 0, 0};
            o = mfma16x32h(ah, bh, o);
            o = mfma16x32h(al, bh, o);
            o = mfma16x32h(ah, bl, o);
            if (r16 < len) *(f32x4*)(a.att + (size_t)(r0 + first + r16) * 256 + 64 * h + 16 * dt + 4 * g) = o;
          }
        }
      }
    }
  }
}

// qkv[r] = table row of (token[r], pos[r]): the first layer's projection for sentences of 17..64 tokens (amazon), whose
// attention kernels read plain qkv rows
__global__ __launch_bounds__(256) void enc_tab_gather_kernel(const float* __restrict__ tab, int tab_len, int vocab,
                                                             const int32_t* __restrict__ tok, const int32_t* __restrict__ pos,
                                                             int R, float* __restrict__ qkv) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const int t = min(max(tok[r], 0), vocab - 1), p = min(max(pos[r], 0), tab_len - 1);
  const float* src = tab + ((size_t)t * tab_len + p) * 768 + lane * 4;
  float* dst = qkv + (size_t)r * 768 + lane * 4;
#pragma unroll
  for (int i = 0; i < 3; ++i) *(f32x4*)(dst + 256 * i) = *(const f32x4*)(src + 256 * i);
}

// latent[s] = sum over the sentence's rows of sigmoid(LayerNorm_f(x[row]))   (model.py:236-240)
__global__ __launch_bounds__(256) void enc_latent_kernel(const float* __restrict__ x, const int32_t* __restrict__ off,
                                                         int n_sent, int n_rows, const float* __restrict__ ln_a,
                                                         const float* __restrict__ ln_b, float* __restrict__ latent) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_sent) return;
  const int lane = threadIdx.x & 63;
  const f32x4 ga = *(const f32x4*)(ln_a + lane * 4), gb = *(const f32x4*)(ln_b + lane * 4);
  f32x4 acc = {0, 0, 0, 0};
  const int r_lo = max(off[s], 0), r_hi = min(off[s + 1], n_rows);  // (bad offsets are flagged by enc_check_len_kernel)
  for (int r = r_lo; r < r_hi; ++r) {
    const f32x4 y = layernorm256(*(const f32x4*)(x + (size_t)r * 256 + lane * 4), ga, gb);
    acc.x += 1.0f / (1.0f + expf(-y.x));
    acc.y += 1.0f / (1.0f + expf(-y.y));
    acc.z += 1.0f / (1.0f + expf(-y.z));
    acc.w += 1.0f / (1.0f + expf(-y.w));
  }
  *(f32x4*)(latent + (size_t)s * 256 + lane * 4) = acc;
}

// A sentence longer than the caller's max_sent_len (or than 64) would be skipped by both attention kernels and leave
// stale workspace rows behind: found here and reported as ST_EINVAL (second flag word; the first = f16 range).
__global__ __launch_bounds__(256) void enc_check_len_kernel(const int32_t* __restrict__ off, int n_sent, int max_len, int n_rows,
                                                            int* __restrict__ flag) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n_sent) return;
  const int len = off[s + 1] - off[s];
  if (len < 0 || len > max_len || off[s] < 0 || off[s + 1] > n_rows) flag[1] = 1;
}

static inline size_t enc_align(size_t v) { return (v + 255) / 256 * 256; }

extern "C" size_t st_encode_workspace_bytes(const StEncoderWeights* w, int n_rows, int n_sent) {
  if (!w || n_rows <= 0) return 0;
  (void)n_sent;
  return enc_align((size_t)n_rows * 256 * 4) * 2 + enc_align((size_t)n_rows * 768 * 4) +
         enc_align((size_t)n_rows * w->d_ff * 4) + 256 +  // + the f16-range flag
         enc_align(((size_t)n_rows / 49 + 2) * 4);       // + first sentence of every ownership window (fused QKV + attention)
}

extern "C" int st_encode_latent(const StEncoderWeights* w, const int32_t* tokens, const int32_t* tok_pos,
                                const int32_t* sent_off, int n_rows, int n_sent, int max_sent_len, float* latent,
                                void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(w != nullptr, "weights struct is NULL");
  ST_CHECK_ARG(w->d_model == ST_D && w->n_heads == 4, "d_model=%d n_heads=%d unsupported (256, 4)", w->d_model, w->n_heads);
  ST_CHECK_ARG(w->d_ff > 0 && w->d_ff % 256 == 0, "d_ff=%d must be a positive multiple of 256", w->d_ff);
  ST_CHECK_ARG(w->n_layers >= 1 && w->n_layers <= ST_MAX_LAYERS, "n_layers=%d", w->n_layers);
  ST_CHECK_ARG(n_rows >= 0 && n_sent >= 0, "n_rows=%d n_sent=%d", n_rows, n_sent);
  ST_CHECK_ARG(max_sent_len <= 64 && max_sent_len <= w->pe_rows, "max_sent_len=%d unsupported (<= 64 and <= pe_rows)", max_sent_len);
  if (n_sent == 0) return ST_OK;
  ST_CHECK_ARG(sent_off && latent, "NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  if (n_rows == 0) {
    ST_HIP(hipMemsetAsync(latent, 0, (size_t)n_sent * 256 * 4, st));
    return ST_OK;
  }
  ST_CHECK_ARG(tokens && tok_pos && workspace, "NULL buffer");
  ST_CHECK_ARG(w->gemm_format == ST_GEMM_F32 || w->gemm_format == ST_GEMM_F16X3, "gemm_format=%d unknown", w->gemm_format);
  const size_t need = st_encode_workspace_bytes(w, n_rows, n_sent);
  if (need > workspace_bytes) {
    st_set_error("encode workspace too small: need %zu bytes, have %zu", need, workspace_bytes);
    return ST_ENOMEM;
  }
  char* base = (char*)workspace;
  float* x = (float*)base; base += enc_align((size_t)n_rows * 256 * 4);
  float* att = (float*)base; base += enc_align((size_t)n_rows * 256 * 4);
  float* qkv = (float*)base; base += enc_align((size_t)n_rows * 768 * 4);
  float* h = (float*)base; base += enc_align((size_t)n_rows * w->d_ff * 4);
  int* range_flag = (int*)base; base += 256;
  int32_t* first_sent = (int32_t*)base;
  ST_HIP(hipMemsetAsync(range_flag, 0, 2 * sizeof(int), st));
  const int R = n_rows;
  rp_ab_env();
  // fused QKV + attention (f16x3, every sentence <= 16 tokens): ownership windows of 65 - max_sent_len >= 49 rows
  const bool fused = w->gemm_format == ST_GEMM_F16X3 && max_sent_len <= 16 && max_sent_len >= 1 && !rp_enc_unfused;
  const int win = 65 - (max_sent_len < 1 ? 1 : max_sent_len), n_win = (R + win - 1) / win;
  if (fused) {
    ST_ENSURE_LDS((enc_qkv_attn16_kernel<0>), EQA_LDS);
    hipLaunchKernelGGL(enc_window_kernel, dim3((n_sent + 1 + 255) / 256), dim3(256), 0, st, sent_off, n_sent, win, n_win, first_sent);
  }
  hipLaunchKernelGGL(enc_check_len_kernel, dim3((n_sent + 255) / 256), dim3(256), 0, st, sent_off, n_sent, max_sent_len, n_rows, range_flag);
  // With the first layer on the (token, position) table nothing reads the embedded rows but that layer's half-layer kernel
  // (its residual), which then computes them itself: no embed kernel, 1 KB per row less written and read.
  const bool emb_in_tail = fused && w->l0_table != nullptr && w->l0_table_len >= max_sent_len && w->gemm_format == ST_GEMM_F16X3 &&
                           max_sent_len <= RP_TAIL_PE_ROWS && w->pe_rows >= RP_TAIL_PE_ROWS && R >= RP_TAIL_MIN_ROWS && rp_tail_shape16 &&
                           !rp_no_emb_in_tail;
  if (!emb_in_tail) {
    hipLaunchKernelGGL(enc_embed_kernel, dim3((R + 3) / 4), dim3(256), 0, st, w->emb, w->pe, tokens, tok_pos, R, w->vocab, x);
    ST_LAUNCH_CHECK();
  }
  for (int l = 0; l < w->n_layers; ++l) {
    const StEncLayer& y = w->layer[l];
    ST_CHECK_ARG(y.ln0_a && y.ln0_b && y.wqkv && y.bqkv && y.wo && y.bo && y.ln1_a && y.ln1_b && y.w1 && y.b1 && y.w2 && y.b2,
                 "NULL weight pointer in encoder layer %d", l);
    const bool tab0 = l == 0 && w->l0_table != nullptr && w->l0_table_len >= max_sent_len && w->gemm_format == ST_GEMM_F16X3;
    if (tab0 && fused) {
      ST_ENSURE_LDS(enc_tab_attn16_kernel, ETA_LDS);
      EtaArgs e = {};
      e.tab = w->l0_table; e.tab_len = w->l0_table_len; e.vocab = w->vocab; e.tokens = tokens; e.tok_pos = tok_pos;
      e.off = sent_off; e.first_sent = first_sent; e.att = att; e.R = R; e.n_sent = n_sent; e.n_win = n_win;
      hipLaunchKernelGGL(enc_tab_attn16_kernel, dim3(std::min(n_win, 2 * st_cu_count())), dim3(512), ETA_LDS, st, e);
      ST_LAUNCH_CHECK();
    } else if (fused) {
      EqaArgs e = {};
      e.x = x; e.off = sent_off; e.first_sent = first_sent; e.ln_a = y.ln0_a; e.ln_b = y.ln0_b; e.W = y.wqkv; e.bias = y.bqkv;
      e.att = att; e.range_flag = range_flag; e.R = R; e.n_sent = n_sent; e.n_win = n_win;
      hipLaunchKernelGGL((enc_qkv_attn16_kernel<0>), dim3(std::min(n_win, st_cu_count())), dim3(512), EQA_LDS, st, e);
      ST_LAUNCH_CHECK();
    } else {
      GemmArgs g = {};
      g.fmt = w->gemm_format; g.range_flag = range_flag;
      g.X = x; g.n_host = R; g.ln_a = y.ln0_a; g.ln_b = y.ln0_b; g.W = y.wqkv; g.bias = y.bqkv; g.N = 768;
      g.out = qkv; g.ldo = 768;
      const int qsplits = rp_splits(R, 768, 3, rp_nw(w->gemm_format, 8));
      EncRowSrc src = {qkv, nullptr, 0, 0, nullptr, nullptr};
      if (tab0 && !rp_enc_tab_gather) {
        src.tab = w->l0_table; src.tab_len = w->l0_table_len; src.vocab = w->vocab; src.tok = tokens; src.pos = tok_pos;
      } else if (tab0) {
        hipLaunchKernelGGL(enc_tab_gather_kernel, dim3((R + 3) / 4), dim3(256), 0, st, w->l0_table, w->l0_table_len, w->vocab, tokens,
                           tok_pos, R, qkv);
        ST_LAUNCH_CHECK();
      } else if (w->gemm_format == ST_GEMM_F16X3)
        ST_TRY(launch_rp_lin_pipelined<EPI_BIAS>(g, R, qsplits, st));
      else
        ST_TRY((launch_rp<EPI_BIAS, 8>(g, R, qsplits, st)));
      if (w->gemm_format == ST_GEMM_F16X3)
        hipLaunchKernelGGL(enc_attn16_kernel<true>, dim3((n_sent + 3) / 4), dim3(256), 0, st, src, sent_off, n_sent, R, att);
      else
        hipLaunchKernelGGL(enc_attn16_kernel<false>, dim3((n_sent + 3) / 4), dim3(256), 0, st, src, sent_off, n_sent, R, att);
      if (max_sent_len > 16) {
        if (w->gemm_format == ST_GEMM_F16X3)
          hipLaunchKernelGGL(enc_attn_kernel<true>, dim3(n_sent), dim3(256), 0, st, src, sent_off, R, att);
        else
          hipLaunchKernelGGL(enc_attn_kernel<false>, dim3(n_sent), dim3(256), 0, st, src, sent_off, R, att);
      }
      ST_LAUNCH_CHECK();
    }
    OFArgs of = {};
    of.fmt = w->gemm_format; of.range_flag = range_flag;
    of.att = att; of.n_host = R; of.wo = y.wo; of.bo = y.bo; of.x = x; of.ln_a = y.ln1_a; of.ln_b = y.ln1_b;
    of.w1 = y.w1; of.b1 = y.b1; of.h = h; of.d_ff = w->d_ff; of.w2 = y.w2; of.b2 = y.b2;
    if (l == 0 && emb_in_tail) {
      of.emb = w->emb; of.pe = w->pe; of.tokens = tokens; of.tok_pos = tok_pos; of.vocab = w->vocab;
    }
    if (w->gemm_format == ST_GEMM_F16X3 && R >= RP_TAIL_MIN_ROWS) {
      ProfPair pt;
      if (g_prof_on == ST_PROF_HALF_LAYER) ST_TRY(prof_begin(&pt, 2.0 * 256.0 * (256.0 + 2.0 * (double)w->d_ff), nullptr, R, st));
      ST_TRY(launch_layer_tail<1>(of, R, st));  // O-projection + FFN in one kernel
      if (g_prof_on == ST_PROF_HALF_LAYER) ST_TRY(prof_end(&pt, st));
      continue;
    }
    if (R >= RP_FUSE_MIN_ROWS) {
      ST_TRY(launch_oproj_ffn1(of, R, st));
    } else {
      GemmArgs o = {};
      o.fmt = w->gemm_format; o.range_flag = range_flag;
      o.X = att; o.n_host = R; o.W = y.wo; o.bias = y.bo; o.N = 256; o.out = x; o.ldo = 256; o.resid = x;
      ST_TRY(launch_rp<EPI_RESID>(o, R, rp_splits(R, 256, 2), st));
      GemmArgs f1 = {};
      f1.fmt = w->gemm_format; f1.range_flag = range_flag;
      f1.X = x; f1.n_host = R; f1.ln_a = y.ln1_a; f1.ln_b = y.ln1_b; f1.W = y.w1; f1.bias = y.b1; f1.N = w->d_ff;
      f1.out = h; f1.ldo = w->d_ff;
      ST_TRY((launch_rp<EPI_RELU, 8>(f1, R, rp_splits(R, w->d_ff, 4, rp_nw(w->gemm_format, 8)), st)));
    }
    GemmArgs f2 = {};
    f2.fmt = w->gemm_format; f2.range_flag = range_flag;
    f2.X = h; f2.n_host = R; f2.K = w->d_ff; f2.W = y.w2; f2.bias = y.b2; f2.N = 256; f2.out = x; f2.ldo = 256; f2.resid = x;
    ST_TRY(launch_rp_kchunk<EPI_RESID>(f2, R, st));
  }
  hipLaunchKernelGGL(enc_latent_kernel, dim3((n_sent + 3) / 4), dim3(256), 0, st, x, sent_off, n_sent, R, w->lnf_a, w->lnf_b, latent);
  ST_LAUNCH_CHECK();
  {  // the only synchronisation of this entry point: the flag word
    int flag[2] = {0, 0};
    ST_HIP(hipMemcpyAsync(flag, range_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    if (flag[1]) {
      st_set_error("st_encode_latent: sent_off holds a sentence longer than max_sent_len=%d (or offsets outside [0, n_rows]); "
                   "the latents are invalid", max_sent_len);
      return ST_EINVAL;
    }
    if (flag[0]) {
      st_set_error("st_encode_latent: an activation left the f16 range (|x| >= 65504 or NaN) in the f16x3 format; "
                   "the latents are invalid - pack the weights with ST_GEMM_F32");
      return ST_ERANGE;
    }
  }
  return ST_OK;
}

extern "C" size_t st_encode_l0_table_bytes(const StEncoderWeights* w, int positions) {
  if (!w || positions <= 0 || w->vocab <= 0) return 0;
  return (size_t)w->vocab * positions * 768 * 4;
}

__global__ __launch_bounds__(256) void enc_l0_embed_kernel(const float* __restrict__ emb, const float* __restrict__ pe, int vocab,
                                                           int positions, float* __restrict__ x) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)vocab * positions) return;
  const int lane = threadIdx.x & 63;
  const int tok = (int)(r / positions), pos = (int)(r % positions);
  const f32x4 e = *(const f32x4*)(emb + (size_t)tok * 256 + lane * 4);
  const f32x4 p = *(const f32x4*)(pe + (size_t)pos * 256 + lane * 4);
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = fmaf(e[q], 16.0f, p[q]);   // exactly enc_embed_kernel's expression
  *(f32x4*)(x + (size_t)r * 256 + lane * 4) = v;
}

extern "C" int st_encode_build_l0_table(const StEncoderWeights* w, int positions, float* table, void* scratch,
                                        size_t scratch_bytes, void* stream) {
  ST_CHECK_ARG(w != nullptr && table && scratch, "NULL argument");
  ST_CHECK_ARG(w->d_model == ST_D && w->n_heads == 4 && w->n_layers >= 1, "encoder shape unsupported");
  ST_CHECK_ARG(w->gemm_format == ST_GEMM_F16X3, "the first-layer table is used by the f16x3 format only");
  ST_CHECK_ARG(positions >= 1 && positions <= 64 && positions <= w->pe_rows, "positions=%d", positions);
  const long rows = (long)w->vocab * positions;
  ST_CHECK_ARG(rows < (1L << 31), "vocab * positions = %ld rows: too large", rows);
  ST_CHECK_ARG(scratch_bytes >= (size_t)rows * 1024, "scratch too small: need %zu bytes", (size_t)rows * 1024);
  const StEncLayer& y = w->layer[0];
  ST_CHECK_ARG(w->emb && w->pe && y.ln0_a && y.ln0_b && y.wqkv && y.bqkv, "NULL weight pointer");
  hipStream_t st = (hipStream_t)stream;
  float* x = (float*)scratch;
  int* flag = nullptr;
  ST_HIP(hipMalloc(&flag, sizeof(int)));
  ST_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
  hipLaunchKernelGGL(enc_l0_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, w->emb, w->pe, w->vocab, positions, x);
  ST_LAUNCH_CHECK();
  int rc = ST_OK;
  for (long r0 = 0; r0 < rows && rc == ST_OK; r0 += 1 << 20) {
    const int n = (int)((rows - r0 < (1L << 20)) ? rows - r0 : (1L << 20));
    GemmArgs g = {};
    g.fmt = w->gemm_format; g.range_flag = flag;
    g.X = x + (size_t)r0 * 256; g.n_host = n; g.ln_a = y.ln0_a; g.ln_b = y.ln0_b; g.W = y.wqkv; g.bias = y.bqkv; g.N = 768;
    g.out = table + (size_t)r0 * 768; g.ldo = 768;
    rc = launch_rp_lin_pipelined<EPI_BIAS>(g, n, 3, st);
  }
  int bad = 0;
  if (rc == ST_OK) {
    hipError_t e = hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = ST_EHIP;
  }
  (void)hipFree(flag);
  if (rc != ST_OK) return rc;
  if (bad) {
    st_set_error("st_encode_build_l0_table: an activation left the f16 range; pack the weights with ST_GEMM_F32");
    return ST_ERANGE;
  }
  return ST_OK;
}
