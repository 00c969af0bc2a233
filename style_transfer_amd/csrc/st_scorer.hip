// Fused style-scorer step: SiameseNet (+Predictor) forward, contrastive score against the like /
// unlike unit vectors, backward to the latent, Adam update — one launch, one 32-sentence tile per
// workgroup, every intermediate in LDS.  Replaces get_contrastive_loss + loss.backward() +
// Adam.step() (/root/reference/method/yelp/siamese_contrastive.py:367-453, :796-810).
//
// Only d(loss)/d(latent) is needed, so the backward pass needs no stored activations — just the
// ReLU / LeakyReLU sign masks (kept as ballot bit-words) and the transposed weights.
// Products run on v_mfma_f32_32x32x2_f32 (exact f32): M = 32 sentences, A from LDS, B (weights)
// straight from L2, fragment-packed by st_pack_weight so each wave load is one coalesced 1 KiB read
// (weights are 2.3 MB, read once per tile).
#include "st_common.h"
#include <stdlib.h>

#define SC_LD 516        // activation row stride in floats (512 + 4; = 4 mod 64 -> conflict-free b128)
#define SC_ROWS 32
#define SC_NW 8          // waves per workgroup; two workgroups per CU (79 KiB of LDS each) = 4 waves per SIMD
#define SC_RPW (SC_ROWS / SC_NW)  // rows per wave in the row-wise phases
#define SC_BUF (SC_ROWS * SC_LD)
#ifndef SC_ZCH
#define SC_ZCH 10     // like / unlike unit vectors requested together in the score phase
#endif

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };
enum { BWD_NONE = 0, BWD_RELU = 1, BWD_LRELU = 2 };

// Y[32,N] = X[32,K] . W[N,K]^T, wave w takes 32-column chunks w, w+4, ...
//  forward  (FWD=true):  y = act(acc + bias[col]); sign bits of the pre-activation -> bits[row][chunk]
//  backward (FWD=false): y = acc * act'(bits[row][chunk])
template <int K, int N, bool FWD, int ACT>
__device__ __forceinline__ void tile_linear(const float* Xs, float* Ys,
                                            const float* __restrict__ W, const float* __restrict__ bias,
                                            uint32_t* __restrict__ bits, int bits_ld) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // W fragments (fragment-packed by st_pack_weight, 1 KiB per wave load) run through an 8-deep register ring that
  // continues into the wave's next 32-column chunk, so only the first chunk of a layer waits for L2
  constexpr int KS = K / 8, RW = 8;
  static_assert(KS % RW == 0, "K must be a multiple of 64");
  const int nch = (N / 32 > wave) ? (N / 32 - wave + SC_NW - 1) / SC_NW : 0;
  auto wfrag = [&](int ci, int kg) -> f32x4 {
    const int c = wave + SC_NW * min(ci, nch - 1);
    return *(const f32x4*)(W + ((size_t)c * KS + kg) * 256 + (size_t)lane * 4);
  };
  f32x4 ring[RW];
  if (nch > 0) {
#pragma unroll
    for (int j = 0; j < RW; ++j) ring[j] = wfrag(0, j);
  }
  // A wave owns at most two 32-column chunks (N <= 512, 8 waves): both are accumulated in registers, and the outputs
  // are written only after EVERY wave has finished reading Xs, so Ys may be the same buffer as Xs.  One activation
  // buffer instead of two brings the workgroup to 79 KiB of LDS: two workgroups share a CU and one's row-wise phases
  // (scores, Adam, layer-start weight latency) run under the other's MFMAs.
  static_assert(N / 32 <= 2 * SC_NW, "at most two chunks per wave");
  auto compute = [&](int ci, f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* xr = Xs + l31 * SC_LD + 4 * half;
    for (int kg0 = 0; kg0 < KS; kg0 += RW) {
#pragma unroll
      for (int j = 0; j < RW; ++j) {
        const int kg = kg0 + j;
        const f32x4 b = ring[j];
        ring[j] = kg + RW < KS ? wfrag(ci, kg + RW) : wfrag(ci + 1, kg + RW - KS);
        const f32x4 a = *(const f32x4*)(xr + kg * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma32(a[s], b[s], acc);
      }
    }
  };
  auto store = [&](int ci, const f32x16& acc) {
    const int c = wave + SC_NW * ci;
    const int col = c * 32 + l31;
    const float bv = FWD ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mfma32_row(r, lane);
      float v = acc[r] + bv;
      if (FWD) {
        if (ACT != ACT_NONE) {
          const bool on = v > 0.0f;
          const unsigned long long bal = __ballot(on);
          if (lane == 0) {
            const int r0 = (r & 3) + 8 * (r >> 2);
            bits[r0 * bits_ld + c] = (uint32_t)bal;
            bits[(r0 + 4) * bits_ld + c] = (uint32_t)(bal >> 32);
          }
          v = on ? v : (ACT == ACT_RELU ? 0.0f : 0.2f * v);
        }
      } else {
        if (ACT != BWD_NONE) {
          const bool on = (bits[row * bits_ld + c] >> l31) & 1u;
          v = on ? v : (ACT == BWD_RELU ? 0.0f : 0.2f * v);
        }
      }
      Ys[row * SC_LD + col] = v;
    }
  };
  f32x16 acc0, acc1;
  if (nch > 0) compute(0, acc0);
  if (N / 32 > SC_NW && nch > 1) compute(1, acc1);
  __syncthreads();
  if (nch > 0) store(0, acc0);
  if (N / 32 > SC_NW && nch > 1) store(1, acc1);
}

struct ScorerArgs {
  StScorerWeights w;
  // embed mode
  const float* x;   // [n,256]
  float* z;         // [n,256]
  int n;
  int unit;
  // edit mode
  float* latent;
  float* adam_m;
  float* adam_v;
  const float* like_z;
  const float* unlike_z;
  int n_pos, n_neg;
  const float* lr;
  int* step;
  const uint8_t* active;
  float tau;
  int flags;
  float* loss_out;
  float* grad_out;
  float* scores_out;  // [B, n_pos + n_neg] cos / tau (optional)
};

// The optimizer update of one sentence row by one wave (lane = 4 coordinates): torch.optim.Adam's single-tensor path
// (betas (0.9, 0.999), eps 1e-8) or one of the other rules of --victim_optimizer (flags bits 8..11).
__device__ __forceinline__ void update_row(const ScorerArgs& a, int b, f32x4 g) {
  const int lane = threadIdx.x & 63;
  const int t = a.step[b] + 1;
  const int opt = (a.flags & ST_F_OPT_MASK) >> ST_F_OPT_SHIFT;
  f32x4 m = *(const f32x4*)(a.adam_m + (size_t)b * 256 + lane * 4);
  f32x4 v = *(const f32x4*)(a.adam_v + (size_t)b * 256 + lane * 4);
  f32x4 x = *(const f32x4*)(a.latent + (size_t)b * 256 + lane * 4);
  const float lr = a.lr[b];
  if (opt == ST_OPT_ADAM || opt == ST_OPT_ADAMW) {
    const double bc1 = 1.0 - pow(0.9, (double)t);
    const double bc2 = 1.0 - pow(0.999, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    if (opt == ST_OPT_ADAMW) x = x * (1.0f - lr * 0.01f);  // torch.optim.AdamW: param.mul_(1 - lr * weight_decay) first
    m = m + 0.1f * (g - m);
    v = v * 0.999f + (0.001f * g) * g;
    f32x4 den;
    den.x = sqrtf(v.x) / bc2s + 1e-8f;
    den.y = sqrtf(v.y) / bc2s + 1e-8f;
    den.z = sqrtf(v.z) / bc2s + 1e-8f;
    den.w = sqrtf(v.w) / bc2s + 1e-8f;
    x = x - step_size * (m / den);
  } else if (opt == ST_OPT_SGD) {
    x = x - lr * g;
  } else if (opt == ST_OPT_ADAMAX) {
    m = m + 0.1f * (g - m);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q] * 0.999f, fabsf(g[q]) + 1e-8f);
    const float clr = (float)((double)lr / (1.0 - pow(0.9, (double)t)));
    x = x - clr * (m / v);
  } else if (opt == ST_OPT_RMSPROP) {
    v = v * 0.99f + (0.01f * g) * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] -= lr * (g[q] / (sqrtf(v[q]) + 1e-8f));
  } else if (opt == ST_OPT_ADAGRAD) {
    v = v + g * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] -= lr * (g[q] / (sqrtf(v[q]) + 1e-10f));
  } else if (opt == ST_OPT_NADAM) {
    // torch.optim.NAdam (momentum_decay 4e-3): mu_t = 0.9 (1 - 0.5 * 0.96^(t * 0.004)), mu_product = prod_{i <= t} mu_i
    const double mu = 0.9 * (1.0 - 0.5 * pow(0.96, (double)t * 0.004));
    const double mu_next = 0.9 * (1.0 - 0.5 * pow(0.96, (double)(t + 1) * 0.004));
    double mu_prod = 1.0;
    for (int i = 1; i <= t; ++i) mu_prod *= 0.9 * (1.0 - 0.5 * pow(0.96, (double)i * 0.004));
    const float bc2 = (float)(1.0 - pow(0.999, (double)t));
    m = m + 0.1f * (g - m);
    v = v * 0.999f + (0.001f * g) * g;
    const float cg = (float)(-(double)lr * (1.0 - mu) / (1.0 - mu_prod));
    const float cm = (float)(-(double)lr * mu_next / (1.0 - mu_prod * mu_next));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float den = sqrtf(v[q] / bc2) + 1e-8f;
      x[q] += cg * (g[q] / den);
      x[q] += cm * (m[q] / den);
    }
  } else if (opt == ST_OPT_RADAM) {
    // torch.optim.RAdam: the variance rectification term switches on when rho_t > 5 (t >= 6 with beta2 = 0.999)
    const double b2t = pow(0.999, (double)t);
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - b2t;
    const double rho_inf = 2.0 / (1.0 - 0.999) - 1.0;
    const double rho_t = rho_inf - 2.0 * (double)t * b2t / bc2;
    m = m + 0.1f * (g - m);
    v = v * 0.999f + (0.001f * g) * g;
    const float bc1f = (float)bc1;
    if (rho_t > 5.0) {
      const float rect = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
      const float bc2s = (float)sqrt(bc2);
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] -= (m[q] / bc1f) * lr * (bc2s / (sqrtf(v[q]) + 1e-8f)) * rect;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] -= (m[q] / bc1f) * lr;
    }
  } else if (opt == ST_OPT_ADADELTA) {
    // torch.optim.Adadelta (rho 0.9, eps 1e-6): v = square_avg, m = acc_delta
    v = v * 0.9f + (0.1f * g) * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float delta = sqrtf(m[q] + 1e-6f) / sqrtf(v[q] + 1e-6f) * g[q];
      m[q] = m[q] * 0.9f + 0.1f * delta * delta;
      x[q] -= lr * delta;
    }
  } else if (opt == ST_OPT_ASGD) {
    // torch.optim.ASGD (lambd 1e-4, alpha 0.75, t0 1e6): eta_t = lr / (1 + lambd lr (t - 1))^alpha; the averaged iterate `ax`
    // is never read by the edit loop
    const float eta = (float)((double)lr / pow(1.0 + 1e-4 * (double)lr * (double)(t - 1), 0.75));
    x = x * (1.0f - 1e-4f * eta);
    x = x - eta * g;
  } else {  // ST_OPT_RPROP (etas (0.5, 1.2), step sizes in [1e-6, 50]): m = previous gradient, v = step size (lr at the first step)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float ss = t == 1 ? lr : v[q];
      const float sg = g[q] * m[q];
      float gq = g[q];
      if (sg > 0.0f) ss = fminf(ss * 1.2f, 50.0f);
      else if (sg < 0.0f) { ss = fmaxf(ss * 0.5f, 1e-6f); gq = 0.0f; }
      const float sgn = gq > 0.0f ? 1.0f : (gq < 0.0f ? -1.0f : 0.0f);
      x[q] -= sgn * ss;
      v[q] = ss;
      m[q] = gq;
    }
  }
  *(f32x4*)(a.adam_m + (size_t)b * 256 + lane * 4) = m;
  *(f32x4*)(a.adam_v + (size_t)b * 256 + lane * 4) = v;
  *(f32x4*)(a.latent + (size_t)b * 256 + lane * 4) = x;
  if (lane == 0) a.step[b] = t;
}

// Scores, loss and d(-loss)/dp of ONE sentence row, by one wave (lane = 4 of the 256 features of p):
// pos / neg cosine scores against the like / unlike unit vectors (siamese_contrastive.py:409-419), either objective
// (:424-453), loss_out / scores_out.  Returns the lane's 4 components of dP.
__device__ __forceinline__ f32x4 score_row(const ScorerArgs& a, int m0, int n, int r, int b, f32x4 p, float* cosv, float* coef,
                                           int np, int nn, int nt, float inv_tau) {
  const int lane = threadIdx.x & 63;
  const float pn = fmaxf(sqrtf(wave_sum(p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w)), 1e-8f);
  // The unit vectors come from HBM (20 KB per sentence): SC_ZCH rows are requested together, so a row costs nt / SC_ZCH
  // memory round trips instead of nt (one wave walks its rows alone: with one load in flight the score phase was a fifth
  // of the kernel's time).  Same operations in the same order as the one-at-a-time loop.
  auto zrow = [&](int j) {
#ifdef SC_ABL_NOZ
    return p + (float)j;
#endif
    const float* zp = j < np ? a.like_z + ((size_t)b * np + j) * 256 : a.unlike_z + ((size_t)b * nn + (j - np)) * 256;
    return *(const f32x4*)(zp + lane * 4);
  };
  // --not_nll: the coefficients g_j = +-1 / (tau nt) do not depend on the scores, so sum_j g_j z_j is taken in THIS pass over
  // the unit vectors (the same products and additions in the same order as the second pass below made them) and the
  // vectors are read once per row instead of twice: 20 KB of the 44 KB a sentence costs the kernel per iteration.
  const bool one_pass = (a.flags & ST_F_NOT_NLL) != 0;
  const float c_nn = inv_tau / (float)nt;
  f32x4 gs = {0, 0, 0, 0};
  for (int j0 = 0; j0 < nt; j0 += SC_ZCH) {
    f32x4 zc[SC_ZCH];
#pragma unroll
    for (int u = 0; u < SC_ZCH; ++u) zc[u] = zrow(min(j0 + u, nt - 1));
#pragma unroll
    for (int u = 0; u < SC_ZCH; ++u)
      if (j0 + u < nt) {
        const f32x4 zj = zc[u];
        const float d = wave_sum(p.x * zj.x + p.y * zj.y + p.z * zj.z + p.w * zj.w);
        if (lane == 0) cosv[r * 32 + j0 + u] = d / pn;
        if (one_pass) gs += (j0 + u < np ? c_nn : -c_nn) * zj;
      }
  }
  __builtin_amdgcn_wave_barrier();
  if (a.scores_out && lane < nt && m0 + r < n) a.scores_out[(size_t)(m0 + r) * nt + lane] = cosv[r * 32 + lane] * inv_tau;
  // per-lane coefficient g_j = d(-loss_ref)/d cos_j ; loss_ref = what get_contrastive_loss returns
  float g = 0.0f, lossv = 0.0f;
  if (a.flags & ST_F_NOT_NLL) {
    // loss_ref = (sum neg - sum pos) / (np + nn)      (siamese_contrastive.py:424-436)
    const float c = c_nn;
    if (lane < nt) g = lane < np ? c : -c;
    float sacc = 0.0f;
    for (int j = 0; j < nt; ++j) sacc += (j < np ? -1.0f : 1.0f) * cosv[r * 32 + j] * inv_tau;
    lossv = sacc / (float)nt;
  } else {
    // loss_i = -log_softmax([pos_i, neg_0..])[0]; loss_ref = mean_i (or min_i with --just_use_one)  (:438-453)
    float negmax = -INFINITY;
    for (int k = 0; k < nn; ++k) negmax = fmaxf(negmax, cosv[r * 32 + np + k] * inv_tau);
    int best_i = 0;
    float best_l = INFINITY, mean_l = 0.0f;
    for (int i = 0; i < np; ++i) {
      const float si = cosv[r * 32 + i] * inv_tau;
      const float M = fmaxf(si, negmax);
      float den = expf(si - M);
      for (int k = 0; k < nn; ++k) den += expf(cosv[r * 32 + np + k] * inv_tau - M);
      const float li = M + logf(den) - si;
      mean_l += li;
      if (li < best_l) { best_l = li; best_i = i; }
    }
    const bool one = a.flags & ST_F_JUST_USE_ONE;
    lossv = one ? best_l : mean_l / (float)np;
    if (lane < nt) {
      float cj = 0.0f;  // d loss_ref / d s_lane
      const float sl = cosv[r * 32 + lane] * inv_tau;
      for (int i = 0; i < np; ++i) {
        const float wi = one ? (i == best_i ? 1.0f : 0.0f) : 1.0f / (float)np;
        if (wi == 0.0f) continue;
        const float si = cosv[r * 32 + i] * inv_tau;
        const float M = fmaxf(si, negmax);
        float den = expf(si - M);
        for (int k = 0; k < nn; ++k) den += expf(cosv[r * 32 + np + k] * inv_tau - M);
        if (lane < np) {
          if (i == lane) cj += wi * (expf(si - M) / den - 1.0f);
        } else {
          cj += wi * expf(sl - M) / den;
        }
      }
      g = -cj * inv_tau;
    }
  }
  if (lane < 32) coef[r * 32 + lane] = lane < nt ? g : 0.0f;
  __builtin_amdgcn_wave_barrier();
  float gc = 0.0f;
  if (one_pass) {
    for (int j = 0; j < nt; ++j) gc += coef[r * 32 + j] * cosv[r * 32 + j];
  } else {
    for (int j0 = 0; j0 < nt; j0 += SC_ZCH) {
      f32x4 zc[SC_ZCH];
#pragma unroll
      for (int u = 0; u < SC_ZCH; ++u) zc[u] = zrow(min(j0 + u, nt - 1));
#pragma unroll
      for (int u = 0; u < SC_ZCH; ++u)
        if (j0 + u < nt) {
          const float gj = coef[r * 32 + j0 + u];
          gs += gj * zc[u];
          gc += gj * cosv[r * 32 + j0 + u];
        }
    }
  }
  const f32x4 dp = gs / pn - (gc / (pn * pn)) * p;
  if (lane == 0 && a.loss_out && m0 + r < n) a.loss_out[m0 + r] = lossv;
  return dp;
}

template <bool EDIT>
__global__ __launch_bounds__(64 * SC_NW, 4) void scorer_kernel(ScorerArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* bufA = smem;  // the ONE activation buffer: every layer runs in place (tile_linear)
  float* bufB = smem;
  uint32_t* bits1 = (uint32_t*)(smem + SC_BUF);  // [32][16]
  uint32_t* bits2 = bits1 + 32 * 16;                 // [32][16]
  uint32_t* bitsq = bits2 + 32 * 16;                 // [32][4]
  float* cosv = (float*)(bitsq + 32 * 4);            // [32][32]
  float* coef = cosv + 32 * 32;                      // [32][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.n;
  const int m0 = blockIdx.x * SC_ROWS;
  if (m0 >= n) return;
  const float* xin = EDIT ? a.latent : a.x;

  // latent tile -> bufA (32 rows x 64 float4 = 2048)
#pragma unroll
  for (int i = 0; i < 2048 / (64 * SC_NW); ++i) {
    const int f = tid + i * 64 * SC_NW;
    const int r = f >> 6, c4 = f & 63;
    *(f32x4*)(bufA + r * SC_LD + c4 * 4) = *(const f32x4*)(xin + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4);
  }
  __syncthreads();
  tile_linear<256, 512, true, ACT_RELU>(bufA, bufB, a.w.s1_w, a.w.s1_b, bits1, 16);
  __syncthreads();
  tile_linear<512, 512, true, ACT_RELU>(bufB, bufA, a.w.s2_w, a.w.s2_b, bits2, 16);
  __syncthreads();
  tile_linear<512, 256, true, ACT_NONE>(bufA, bufB, a.w.s3_w, a.w.s3_b, nullptr, 0);
  __syncthreads();

  if (!EDIT) {
    // z (optionally unit-normalised) -> global; wave handles 8 rows
    for (int rr = 0; rr < SC_RPW; ++rr) {
      const int r = wave * SC_RPW + rr;
      if (m0 + r >= n) continue;
      f32x4 v = *(const f32x4*)(bufB + r * SC_LD + lane * 4);
      if (a.unit) {
        const float nn = sqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
        v = v / fmaxf(nn, 1e-8f);
      }
      *(f32x4*)(a.z + (size_t)(m0 + r) * 256 + lane * 4) = v;
    }
    return;
  } else {
    const bool simplest = a.flags & ST_F_SIMPLEST;
    if (!simplest) {
      tile_linear<256, 128, true, ACT_LRELU>(bufB, bufA, a.w.p1_w, a.w.p1_b, bitsq, 4);
      __syncthreads();
      tile_linear<128, 256, true, ACT_NONE>(bufA, bufB, a.w.p2_w, a.w.p2_b, nullptr, 0);
      __syncthreads();
    }
    // ---- scores, loss, dL/dp  (p in bufB[.,0:256]; dP -> bufA[.,0:256]) ----
    const int np = a.n_pos, nn = a.n_neg, nt = np + nn;
    const float inv_tau = 1.0f / a.tau;
    for (int rr = 0; rr < SC_RPW; ++rr) {
      const int r = wave * SC_RPW + rr;
      const int b = min(m0 + r, n - 1);
      const f32x4 p = *(const f32x4*)(bufB + r * SC_LD + lane * 4);
      const f32x4 dp = score_row(a, m0, n, r, b, p, cosv, coef, np, nn, nt, inv_tau);
      *(f32x4*)(bufA + r * SC_LD + lane * 4) = dp;
    }
    __syncthreads();
    // ---- backward to the latent ----
    if (!simplest) {
      tile_linear<256, 128, false, BWD_LRELU>(bufA, bufB, a.w.p2_wt, nullptr, bitsq, 4);
      __syncthreads();
      tile_linear<128, 256, false, BWD_NONE>(bufB, bufA, a.w.p1_wt, nullptr, nullptr, 0);
      __syncthreads();
    }
    tile_linear<256, 512, false, BWD_RELU>(bufA, bufB, a.w.s3_wt, nullptr, bits2, 16);
    __syncthreads();
    tile_linear<512, 512, false, BWD_RELU>(bufB, bufA, a.w.s2_wt, nullptr, bits1, 16);
    __syncthreads();
    tile_linear<512, 256, false, BWD_NONE>(bufA, bufB, a.w.s1_wt, nullptr, nullptr, 0);
    __syncthreads();
    // ---- optimizer step on the latent ----
    for (int rr = 0; rr < SC_RPW; ++rr) {
      const int r = wave * SC_RPW + rr;
      const int b = m0 + r;
      if (b >= n) continue;
      const f32x4 g = *(const f32x4*)(bufB + r * SC_LD + lane * 4);
      if (a.grad_out) *(f32x4*)(a.grad_out + (size_t)b * 256 + lane * 4) = g;
      if (a.active && !a.active[b]) continue;
      update_row(a, b, g);
    }
  }
}

// ================================================================================================
// f16x3 form (StScorerWeights.gemm_format == ST_GEMM_F16X3): the same step on v_mfma_f32_16x16x32_f16 with split
// operands x = hi + lo, three products per term (hi.hi + lo.hi + hi.lo, f32 accumulate), i.e. 16/3 of the exact-f32
// MFMA rate: the f32 kernel above is MFMA-bound (75 MFLOP per 32-sentence tile = 123 us of one CU's f32 matrix pipe).
//  * Weights: the packed f16x3 image of st_pack_weight (the 32x32x16 layout read through the 16x16x32 lane addresses,
//    as in st_rowpanel.h), of W PRE-SCALED by a power of two so that max |W| lies in [2^13, 2^14): BatchNorm-folded
//    weights are small (Predictor fc1: rms 0.003) and an unscaled lo part would be an f16 subnormal with a handful of
//    bits.  The result is multiplied by unscale = 1 / scale (exact).  W is the A operand: a lane ends up with 4
//    consecutive output features of ONE sentence, stored to LDS as packed halves (two 8-byte stores).
//  * Activations: [32 rows][SC_LD dwords], hi halves of feature k at dwords [0, K/2), lo halves at [256, 256 + K/2).
//    Forward activations are O(1) and go in unscaled; the backward pass starts from dP, which is 1e-5 .. 1e-2: every
//    row of dP is scaled by a power of two that brings its largest element to [512, 1024) and the final gradient row
//    is multiplied by the inverse (rowinv), so small gradients keep f32-grade relative precision.
//  * ReLU / LeakyReLU masks: the lane that computes (sentence, feature) in the forward layer is the lane that needs
//    that mask bit in the backward layer of the same width, so the masks live in three registers per lane, not in LDS.
//  * A value beyond the f16 range (it would turn into inf - inf inside the products) raises a sticky device flag that
//    st_scorer_status reports.
// ================================================================================================
#define SH_LO 256
#ifndef SC16_RING
#define SC16_RING 2
#endif
static __device__ int g_scorer_range_flag;

#ifndef SC16_ROWS
#define SC16_ROWS 32                  // sentences per workgroup of the f16x3 kernel (64 builds too: see the note at the kernel)
#endif
#define SC16_RT (SC16_ROWS / 16)      // 16-row tiles
#define SC16_RPW (SC16_ROWS / SC_NW)  // rows per wave in the row-wise phases
#define SC16_BUF (SC16_ROWS * SC_LD)
template <int K, int N, bool FWD, int ACT, bool OUT_F32>
__device__ __forceinline__ void tile_linear16(float* Xs, const float* __restrict__ W, const float* __restrict__ bias,
                                              float unscale, uint32_t (&mask)[SC16_RT / 2], const float* rowinv, int& bad) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  constexpr int DS = K / 32;            // double k-steps
  constexpr int NFT = N / 16 / SC_NW;   // 16-feature tiles per wave: tile t = wave + 8 i
  constexpr int RT = SC16_RT;
  static_assert(NFT >= 1 && N % (16 * SC_NW) == 0 && K % 32 == 0, "layer shape");
  constexpr int R = SC16_RING;          // W ring depth in double steps
  static_assert(DS % R == 0, "ring");
  const unsigned wl = (unsigned)(((g >> 1) * 128 + 32 * (g & 1) + r16) * 16);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, N * K * 4, 0x00020000);
  // A wave's NFT feature tiles go through the matrix cores G at a time (all their accumulators stay in registers until the
  // layer's in-place write): the W ring and the fragments in use are those of G tiles, not of NFT (64 rows x 4 tiles would
  // not fit 256 registers).  The ring runs on from one group into the next, so only a layer's first fragments wait for L2.
  constexpr int G = NFT > 2 ? 2 : NFT, NG = NFT / G, TOT = NG * DS;
  static_assert(NFT % G == 0, "groups");
  auto wfetch = [&](int s, f32x4 (&dst)[G][2]) {   // s = group * DS + double step
    const int ig = s / DS, d = s % DS;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int t = wave + SC_NW * (ig * G + i);
      const int off = (t >> 1) * (K / 16) * 2048 + d * 4096 + (t & 1) * 256;  // wave-uniform
      dst[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl, off, 0));
      dst[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1024, off, 0));
    }
  };
  f32x4 ring[R][G][2];
#pragma unroll
  for (int j = 0; j < R; ++j) wfetch(j, ring[j]);
  f32x4 acc[NFT][RT];
#pragma unroll
  for (int i = 0; i < NFT; ++i)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[i][rt] = (f32x4){0, 0, 0, 0};
  const float* xb = Xs + r16 * SC_LD + 4 * g;
#pragma unroll
  for (int s = 0; s < TOT; ++s) {
    const int ig = s / DS, d = s % DS;
    f32x4 (&wf)[G][2] = ring[s % R];
    h8 xh[RT], xl[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      xh[rt] = as_h8(*(const f32x4*)(xb + 16 * rt * SC_LD + 16 * d));
      xl[rt] = as_h8(*(const f32x4*)(xb + 16 * rt * SC_LD + SH_LO + 16 * d));
    }
    // (per accumulator the order is hi.hi, lo.hi, hi.lo, as in the 32-row form: a row's result does not depend on the tile height)
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[ig * G + i][rt] = mfma16x32h(as_h8(wf[i][0]), xh[rt], acc[ig * G + i][rt]);
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[ig * G + i][rt] = mfma16x32h(as_h8(wf[i][1]), xh[rt], acc[ig * G + i][rt]);
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[ig * G + i][rt] = mfma16x32h(as_h8(wf[i][0]), xl[rt], acc[ig * G + i][rt]);
    // the slot is refilled behind its own MFMAs (no second copy of the fragments)
    if (s + R < TOT) wfetch(s + R, ring[s % R]);
#ifndef SC16_NOSB
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  __syncthreads();  // every wave has read its last input fragment: the layer is written IN PLACE
#pragma unroll
  for (int i = 0; i < NFT; ++i)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int f0 = 16 * (wave + SC_NW * i) + 4 * g;
      const int row = 16 * rt + r16;
      float* yrow = Xs + row * SC_LD;
      f32x4 v = acc[i][rt] * unscale;
      if (FWD) v += *(const f32x4*)(bias + f0);
      if (ACT != ACT_NONE) {
        // mask word rt >> 1 holds the nibbles of row tiles 2 (rt >> 1), 2 (rt >> 1) + 1: (2 i + (rt & 1)) * 4, NFT <= 4
        const int sh = (2 * i + (rt & 1)) * 4;
        uint32_t& mk = mask[rt >> 1];
        if (FWD) {
          uint32_t nib = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool on = v[j] > 0.0f;
            nib |= (on ? 1u : 0u) << j;
            v[j] = on ? v[j] : (ACT == ACT_RELU ? 0.0f : 0.2f * v[j]);
          }
          mk |= nib << sh;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool on = (mk >> (sh + j)) & 1u;
            v[j] = on ? v[j] : (ACT == ACT_RELU ? 0.0f : 0.2f * v[j]);
          }
        }
      }
      if (OUT_F32) {
        if (!FWD) v = v * rowinv[row];
        *(f32x4*)(yrow + f0) = v;
      } else {
        const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        bad |= !(m <= 65504.0f);
        h4 hi, lo;
        split_f16(v, hi, lo);
        *(h4*)(yrow + (f0 >> 1)) = hi;
        *(h4*)(yrow + SH_LO + (f0 >> 1)) = lo;
      }
    }
  // (the range test is evaluated HERE: left alone, the compiler sinks these maxima to the kernel's end, where `bad` is
  // read, and keeps every layer's outputs alive for them - in scratch memory)
  asm volatile("" : "+v"(bad));
}

#define SC16_LDS_BYTES ((SC16_BUF + SC16_ROWS * 32 * 2 + SC16_ROWS) * 4)

// Registers.  The first build of this kernel for two workgroups per CU (128 registers) spilled 8 vector registers to scratch
// and returned WRONG gradients on ~1 % of the rows (lanes 48..63 of one component of dP, only in workgroups that started as
// the second one on a CU, different from run to run: tools/check_scorer_determinism.py), so it ran as one workgroup per CU
// at 159 registers.  The cause of the pressure was not the products: every layer's epilogue tests its outputs against the
// f16 range (`bad`), and the compiler SANK those maxima to the end of the kernel, where `bad` is read, keeping every layer's
// outputs alive for them.  With the test pinned behind each layer (the asm statement at the end of tile_linear16) the
// forward instantiation needs 90 registers and the edit step 109 - no spill at the 128 of two workgroups per CU, which
// hide each other's layer-start and score-phase latencies: 1472 -> 1198 us per launch at 98 304 sentences, same box
// (tools/bench_scorer_step.py).  A 64-row tile (SC16_ROWS 64: one workgroup per CU, half the weight stream out of L2, 224 -
// 255 registers) measured the same 1.21 ms, and streaming the unit vectors through register buffers one row ahead bought it
// nothing more; 64 rows on 16 waves (the registers of 32 x 8) measured 1.40 ms: the two-workgroup form is the simplest and
// the fastest of the four.  tests/test_gpu_kernels.py::
// test_scorer_edit_step_is_bit_reproducible guards the result.
#ifndef SC16_WAVES_PER_EU
#define SC16_WAVES_PER_EU (SC16_ROWS > 32 ? 2 : 4)
#endif
template <bool EDIT>
__global__ __launch_bounds__(64 * SC_NW, SC16_WAVES_PER_EU) void scorer16_kernel(ScorerArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* buf = smem;                       // [SC16_ROWS][SC_LD]: the one activation buffer
  float* cosv = smem + SC16_BUF;           // [SC16_ROWS][32]
  float* coef = cosv + SC16_ROWS * 32;     // [SC16_ROWS][32]
  float* rowinv = coef + SC16_ROWS * 32;   // [SC16_ROWS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.n;
  const int m0 = blockIdx.x * SC16_ROWS;
  if (m0 >= n) return;
  const float* xin = EDIT ? a.latent : a.x;
  int bad = 0;
  // latent tile -> split halves (SC16_ROWS rows x 64 float4)
#pragma unroll
  for (int i = 0; i < SC16_ROWS * 64 / (64 * SC_NW); ++i) {
    const int f = tid + i * 64 * SC_NW;
    const int r = f >> 6, c4 = f & 63;
    const f32x4 v = *(const f32x4*)(xin + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4);
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    bad |= !(m <= 65504.0f);
    h4 hi, lo;
    split_f16(v, hi, lo);
    *(h4*)(buf + r * SC_LD + c4 * 2) = hi;
    *(h4*)(buf + r * SC_LD + SH_LO + c4 * 2) = lo;
  }
  __syncthreads();
  const float* us = a.w.unscale;
  uint32_t m1[SC16_RT / 2] = {}, m2[SC16_RT / 2] = {}, mq[SC16_RT / 2] = {}, none[SC16_RT / 2] = {};
  const bool simplest = EDIT && (a.flags & ST_F_SIMPLEST);
  tile_linear16<256, 512, true, ACT_RELU, false>(buf, a.w.s1_w, a.w.s1_b, us[0], m1, nullptr, bad);
  __syncthreads();
  tile_linear16<512, 512, true, ACT_RELU, false>(buf, a.w.s2_w, a.w.s2_b, us[1], m2, nullptr, bad);
  __syncthreads();
  if (!EDIT || simplest)
    tile_linear16<512, 256, true, ACT_NONE, true>(buf, a.w.s3_w, a.w.s3_b, us[2], none, nullptr, bad);
  else
    tile_linear16<512, 256, true, ACT_NONE, false>(buf, a.w.s3_w, a.w.s3_b, us[2], none, nullptr, bad);
  __syncthreads();

  if (!EDIT) {
    for (int rr = 0; rr < SC16_RPW; ++rr) {
      const int r = wave * SC16_RPW + rr;
      if (m0 + r >= n) continue;
      f32x4 v = *(const f32x4*)(buf + r * SC_LD + lane * 4);
      if (a.unit) {
        const float nn = sqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
        v = v / fmaxf(nn, 1e-8f);
      }
      *(f32x4*)(a.z + (size_t)(m0 + r) * 256 + lane * 4) = v;
    }
    if (bad) g_scorer_range_flag = 1;
    return;
  } else {
    if (!simplest) {
      tile_linear16<256, 128, true, ACT_LRELU, false>(buf, a.w.p1_w, a.w.p1_b, us[3], mq, nullptr, bad);
      __syncthreads();
      tile_linear16<128, 256, true, ACT_NONE, true>(buf, a.w.p2_w, a.w.p2_b, us[4], none, nullptr, bad);
      __syncthreads();
    }
    // ---- scores, loss, dL/dp: p as f32 in buf[., 0:256]; dP goes back as scaled split halves ----
    const int np = a.n_pos, nn = a.n_neg, nt = np + nn;
    const float inv_tau = 1.0f / a.tau;
    // dP row -> scaled split halves in the activation buffer
    auto finish_row = [&](int r, f32x4 dp) {
      const float mx = wave_max(fmaxf(fmaxf(fabsf(dp.x), fabsf(dp.y)), fmaxf(fabsf(dp.z), fabsf(dp.w))));
      // power of two that brings the row's largest |dP| into [512, 1024); 1 for a zero / non-finite row
      const int ex = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;
      const bool ok = mx > 0.0f && ex > -100 && ex < 100;
      const float sc = ok ? __uint_as_float((uint32_t)(127 + 9 - ex) << 23) : 1.0f;
      if (lane == 0) rowinv[r] = ok ? __uint_as_float((uint32_t)(127 - 9 + ex) << 23) : 1.0f;
      const f32x4 v = dp * sc;
      const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
      bad |= !(m <= 65504.0f);
      h4 hi, lo;
      split_f16(v, hi, lo);
      *(h4*)(buf + r * SC_LD + lane * 2) = hi;
      *(h4*)(buf + r * SC_LD + SH_LO + lane * 2) = lo;
    };
    for (int rr = 0; rr < SC16_RPW; ++rr) {
      const int r = wave * SC16_RPW + rr;
      const int b = min(m0 + r, n - 1);
      const f32x4 p = *(const f32x4*)(buf + r * SC_LD + lane * 4);
      finish_row(r, score_row(a, m0, n, r, b, p, cosv, coef, np, nn, nt, inv_tau));
    }
    __syncthreads();
    // ---- backward to the latent ----
    if (!simplest) {
      tile_linear16<256, 128, false, BWD_LRELU, false>(buf, a.w.p2_wt, nullptr, us[4], mq, nullptr, bad);
      __syncthreads();
      tile_linear16<128, 256, false, BWD_NONE, false>(buf, a.w.p1_wt, nullptr, us[3], none, nullptr, bad);
      __syncthreads();
    }
    tile_linear16<256, 512, false, BWD_RELU, false>(buf, a.w.s3_wt, nullptr, us[2], m2, nullptr, bad);
    __syncthreads();
    tile_linear16<512, 512, false, BWD_RELU, false>(buf, a.w.s2_wt, nullptr, us[1], m1, nullptr, bad);
    __syncthreads();
    tile_linear16<512, 256, false, BWD_NONE, true>(buf, a.w.s1_wt, nullptr, us[0], none, rowinv, bad);
    __syncthreads();
    for (int rr = 0; rr < SC16_RPW; ++rr) {
      const int r = wave * SC16_RPW + rr;
      const int b = m0 + r;
      if (b >= n) continue;
      const f32x4 g = *(const f32x4*)(buf + r * SC_LD + lane * 4);
      if (a.grad_out) *(f32x4*)(a.grad_out + (size_t)b * 256 + lane * 4) = g;
      if (a.active && !a.active[b]) continue;
      update_row(a, b, g);
    }
    if (bad) g_scorer_range_flag = 1;
  }
}

#define SC_LDS_BYTES ((SC_BUF + 32 * 16 * 2 + 32 * 4 + 32 * 32 * 2) * 4)

static int check_scorer(const StScorerWeights* w) {
  ST_CHECK_ARG(w != nullptr, "scorer weights struct is NULL");
  ST_CHECK_ARG(w->s1_w && w->s1_wt && w->s1_b && w->s2_w && w->s2_wt && w->s2_b && w->s3_w && w->s3_wt && w->s3_b,
               "NULL SiameseNet weight pointer");
  ST_CHECK_ARG(w->gemm_format == ST_GEMM_F32 || w->gemm_format == ST_GEMM_F16X3, "scorer gemm_format=%d unknown", w->gemm_format);
  if (w->gemm_format == ST_GEMM_F16X3)
    for (int i = 0; i < 5; ++i) ST_CHECK_ARG(w->unscale[i] > 0.0f, "scorer unscale[%d]=%g must be > 0 in the f16x3 format", i, (double)w->unscale[i]);
  return ST_OK;
}

template <bool EDIT>
static int launch_scorer(const ScorerArgs& a, hipStream_t st) {
  if (a.w.gemm_format == ST_GEMM_F16X3) {
    ST_ENSURE_LDS((scorer16_kernel<EDIT>), SC16_LDS_BYTES);
    hipLaunchKernelGGL(scorer16_kernel<EDIT>, dim3((a.n + SC16_ROWS - 1) / SC16_ROWS), dim3(64 * SC_NW), SC16_LDS_BYTES, st, a);
    ST_LAUNCH_CHECK();
    return ST_OK;
  }
  ST_ENSURE_LDS((scorer_kernel<EDIT>), SC_LDS_BYTES);
  hipLaunchKernelGGL(scorer_kernel<EDIT>, dim3((a.n + SC_ROWS - 1) / SC_ROWS), dim3(64 * SC_NW), SC_LDS_BYTES, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

extern "C" int st_scorer_embed(const StScorerWeights* w, const float* x, float* z, int n, int unit, void* stream) {
  int rc = check_scorer(w);
  if (rc) return rc;
  ST_CHECK_ARG(n >= 0, "n=%d", n);
  if (n == 0) return ST_OK;
  ST_CHECK_ARG(x && z, "NULL buffer");
  ScorerArgs a = {};
  a.w = *w; a.x = x; a.z = z; a.n = n; a.unit = unit;
  return launch_scorer<false>(a, (hipStream_t)stream);
}

extern "C" int st_scorer_edit_step(const StScorerWeights* w, float* latent, float* adam_m, float* adam_v,
                                   const float* like_z, const float* unlike_z, int n_pos, int n_neg, const float* lr,
                                   int* step, const uint8_t* active, int B, float tau, int flags, float* loss_out,
                                   float* grad_out, float* scores_out, void* stream) {
  int rc = check_scorer(w);
  if (rc) return rc;
  ST_CHECK_ARG(B >= 0, "B=%d", B);
  if (B == 0) return ST_OK;
  ST_CHECK_ARG(latent && adam_m && adam_v && lr && step, "NULL buffer");
  ST_CHECK_ARG(n_pos >= 0 && n_pos <= ST_MAX_REFS && n_neg >= 0 && n_neg <= ST_MAX_REFS, "n_pos=%d n_neg=%d (max %d each)",
               n_pos, n_neg, ST_MAX_REFS);
  ST_CHECK_ARG(n_pos + n_neg > 0, "need at least one like or unlike sentence");
  ST_CHECK_ARG((n_pos == 0 || like_z) && (n_neg == 0 || unlike_z), "NULL like_z/unlike_z");
  ST_CHECK_ARG((flags & ST_F_NOT_NLL) || n_pos > 0, "the NLL objective needs n_pos > 0");
  ST_CHECK_ARG(tau > 0.0f, "tau=%g", (double)tau);
  ST_CHECK_ARG(((flags & ST_F_OPT_MASK) >> ST_F_OPT_SHIFT) <= ST_OPT_RPROP, "unknown optimizer code %d in flags",
               (flags & ST_F_OPT_MASK) >> ST_F_OPT_SHIFT);
  if (!(flags & ST_F_SIMPLEST))
    ST_CHECK_ARG(w->p1_w && w->p1_wt && w->p1_b && w->p2_w && w->p2_wt && w->p2_b, "NULL Predictor weight pointer");
  ScorerArgs a = {};
  a.w = *w; a.n = B; a.latent = latent; a.adam_m = adam_m; a.adam_v = adam_v; a.like_z = like_z; a.unlike_z = unlike_z;
  a.n_pos = n_pos; a.n_neg = n_neg; a.lr = lr; a.step = step; a.active = active; a.tau = tau; a.flags = flags;
  a.loss_out = loss_out; a.grad_out = grad_out; a.scores_out = scores_out;
  return launch_scorer<true>(a, (hipStream_t)stream);
}

/* sticky f16-range flag of the f16x3 scorer kernels (see st_kernels.h) */
extern "C" int st_scorer_status(int* out_of_range, int reset) {
  ST_CHECK_ARG(out_of_range != nullptr, "st_scorer_status: NULL output");
  ST_HIP(hipMemcpyFromSymbol(out_of_range, HIP_SYMBOL(g_scorer_range_flag), sizeof(int)));
  if (reset && *out_of_range) {
    const int zero = 0;
    ST_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scorer_range_flag), &zero, sizeof(int)));
  }
  return ST_OK;
}
