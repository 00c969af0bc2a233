// Legacy FGIM edit: Classifier (model.py:414-433: 256 -> 100 -> 50 -> 1, LeakyReLU(0.2),
// sigmoid) forward, BCE loss against the target label, backward to the latent, and the perturbed latents
// x - eps_k * dL/dx for a list of step sizes (fgim_attack, model.py:454-516: every epsilon of its list takes
// exactly ONE step from the original latent, because `it` starts at 4 and the loop ends at it >= 5).
// One wave per sentence; the classifier is 31 k parameters (L1/L2 resident), so plain VALU dot products
// with DPP wave sums are enough: this path is a few microseconds next to the 9 greedy decodes that follow it.
#include "st_common.h"

#define FG_H1 100
#define FG_H2 50

__global__ __launch_bounds__(256) void fgim_kernel(StClassifierWeights w, const float* __restrict__ latent,
                                                   const float* __restrict__ target, const float* __restrict__ eps,
                                                   int n_eps, int B, float* __restrict__ pred, float* __restrict__ grad,
                                                   float* __restrict__ out) {
  __shared__ float h1s[4][FG_H1 + 4], d1s[4][FG_H1 + 4], h2s[4][FG_H2 + 2], d2s[4][FG_H2 + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  float* h1 = h1s[wave];
  float* d1 = d1s[wave];
  float* h2 = h2s[wave];
  float* d2 = d2s[wave];
  const f32x4 x = *(const f32x4*)(latent + (size_t)b * 256 + lane * 4);
  // fc1 + LeakyReLU: pre-activations z1 (kept for the backward sign)
  for (int j = 0; j < FG_H1; ++j) {
    const f32x4 wj = *(const f32x4*)(w.w1 + (size_t)j * 256 + lane * 4);
    const float z = wave_sum(fmaf(x.x, wj.x, fmaf(x.y, wj.y, fmaf(x.z, wj.z, x.w * wj.w)))) + w.b1[j];
    if (lane == 0) h1[j] = z;
  }
  __builtin_amdgcn_wave_barrier();
  // fc2 + LeakyReLU (lane = output unit)
  float z2 = 0.0f;
  if (lane < FG_H2) {
    float acc = w.b2[lane];
    for (int j = 0; j < FG_H1; ++j) {
      const float z = h1[j];
      acc = fmaf(w.w2[lane * FG_H1 + j], z > 0.0f ? z : 0.2f * z, acc);
    }
    z2 = acc;
    h2[lane] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  // fc3 + sigmoid
  const float a2 = z2 > 0.0f ? z2 : 0.2f * z2;
  const float z3 = wave_sum(lane < FG_H2 ? w.w3[lane] * a2 : 0.0f) + w.b3[0];
  const float o = 1.0f / (1.0f + expf(-z3));
  if (lane == 0 && pred) pred[b] = o;
  // BCELoss(size_average) over this sentence's single output: dL/do = (o - t) / (o (1 - o)); d o / d z3 = o (1 - o)
  // (torch.nn.BCELoss clamps log() at -100, which only matters when o rounds to exactly 0 or 1 in f32)
  const float t = target[b];
  const float oo = fmaxf(o * (1.0f - o), 1e-12f);
  const float dz3 = (o - t) / oo * (o * (1.0f - o));
  // back through fc3, LeakyReLU, fc2
  if (lane < FG_H2) d2[lane] = dz3 * w.w3[lane] * (z2 > 0.0f ? 1.0f : 0.2f);
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < FG_H1; j += 64) {
    float acc = 0.0f;
    for (int k = 0; k < FG_H2; ++k) acc = fmaf(d2[k], w.w2[k * FG_H1 + j], acc);
    d1[j] = acc * (h1[j] > 0.0f ? 1.0f : 0.2f);
  }
  __builtin_amdgcn_wave_barrier();
  // back through fc1: lane owns 4 latent coordinates
  f32x4 g = {0, 0, 0, 0};
  for (int j = 0; j < FG_H1; ++j) g += d1[j] * *(const f32x4*)(w.w1 + (size_t)j * 256 + lane * 4);
  if (grad) *(f32x4*)(grad + (size_t)b * 256 + lane * 4) = g;
  for (int k = 0; k < n_eps; ++k) *(f32x4*)(out + ((size_t)k * B + b) * 256 + lane * 4) = x - eps[k] * g;
}

extern "C" int st_fgim_perturb(const StClassifierWeights* w, const float* latent, const float* target, const float* eps,
                               int n_eps, int B, float* pred, float* grad, float* out, void* stream) {
  ST_CHECK_ARG(w && w->w1 && w->b1 && w->w2 && w->b2 && w->w3 && w->b3, "NULL classifier weight pointer");
  ST_CHECK_ARG(B >= 0 && n_eps >= 0 && n_eps <= 64, "B=%d n_eps=%d (n_eps <= 64)", B, n_eps);
  if (B == 0) return ST_OK;
  ST_CHECK_ARG(latent && target && (n_eps == 0 || (eps && out)), "NULL buffer");
  hipLaunchKernelGGL(fgim_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, *w, latent, target, eps, n_eps, B, pred,
                     grad, out);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
