// KV-cached greedy decode of the Transformer autoencoder's decoder (replaces
// EncoderDecoder.greedy_decode, /root/reference/method/yelp/model.py:271-299, which re-runs the whole
// decoder over the growing prefix at every step) + generator GEMM with fused argmax (model.py:200-207,
// :290-292).  One launch sequence per decode step over the COMPACTED list of live sentences.
#define ST_RP_DEFINE_SWITCHES  /* this translation unit owns the kernel-form switches of st_rowpanel.h */
#include "st_rowpanel.h"

#include <stdarg.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";
void st_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* st_last_error(void) { return g_err; }
extern "C" int st_abi_version(void) { return 8; }

// ---- optional in-library timing of the dominant kernel (generator GEMM + argmax) -----------------
#include <vector>
int g_prof_on = 0;  // ST_PROF_*: which kernel's launches are bracketed (also read by st_encode.hip)
static std::vector<ProfPair> g_prof;       // recorded pairs
static std::vector<ProfPair> g_prof_free;  // recycled events
static double* g_prof_acc = nullptr;       // device: algorithmic flops of the bracketed launches (actual live rows)
__global__ void prof_tap_kernel(const int* n_dev, int n_host, double per_row, double* acc) {
  *acc += (double)(n_dev ? *n_dev : n_host) * per_row;  // one thread, stream-ordered: no atomic needed
}

extern "C" int st_profile_enable(int which) {
  ST_CHECK_ARG(which >= ST_PROF_OFF && which <= ST_PROF_HALF_LAYER, "st_profile_enable: which=%d", which);
  if (which != ST_PROF_OFF && !g_prof_acc) {
    ST_HIP(hipMalloc(&g_prof_acc, sizeof(double)));
    ST_HIP(hipMemset(g_prof_acc, 0, sizeof(double)));
  }
  g_prof_on = which;
  return ST_OK;
}
extern "C" int st_profile_read(double* gen_ms, double* gen_flops, long long* gen_launches, int reset) {
  double ms = 0.0, fl = 0.0;
  for (auto& p : g_prof) {
    ST_HIP(hipEventSynchronize(p.b));
    float t = 0.0f;
    ST_HIP(hipEventElapsedTime(&t, p.a, p.b));
    ms += t;
  }
  if (g_prof_acc) ST_HIP(hipMemcpy(&fl, g_prof_acc, sizeof(double), hipMemcpyDeviceToHost));
  if (gen_ms) *gen_ms = ms;
  if (gen_flops) *gen_flops = fl;
  if (gen_launches) *gen_launches = (long long)g_prof.size();
  if (reset) {
    for (auto& p : g_prof) g_prof_free.push_back(p);
    g_prof.clear();
    if (g_prof_acc) ST_HIP(hipMemset(g_prof_acc, 0, sizeof(double)));
  }
  return ST_OK;
}
int prof_begin(ProfPair* p, double per_row, const int* n_dev, int n_host, hipStream_t st) {
  if (!g_prof_free.empty()) {
    *p = g_prof_free.back();
    g_prof_free.pop_back();
  } else {
    ST_HIP(hipEventCreate(&p->a));
    ST_HIP(hipEventCreate(&p->b));
  }
  p->flops = per_row * n_host;  // host-side bound, informational
  hipLaunchKernelGGL(prof_tap_kernel, dim3(1), dim3(1), 0, st, n_dev, n_host, per_row, g_prof_acc);
  ST_LAUNCH_CHECK();
  ST_HIP(hipEventRecord(p->a, st));
  return ST_OK;
}
int prof_end(ProfPair* p, hipStream_t st) {
  ST_HIP(hipEventRecord(p->b, st));
  g_prof.push_back(*p);
  return ST_OK;
}

// ---- executed-work counters (bench.py's whole-loop roofline): a file-scope device array per GPU -------
// [0] decode token-steps (live rows summed over decode steps)   [1] keys attended (sum of prefix lengths)
// [2] decode jobs (rows at step 0)   [3] reserved.   Each GPU holds its own copy; its address is looked up once per device.
static __device__ __align__(256) unsigned long long g_work_counters[4];
static unsigned long long* g_work_ptr[ST_MAX_DEVICES] = {};
int st_work_counters_ptr(unsigned long long** out) {
  int dev = 0;
  ST_HIP(hipGetDevice(&dev));
  ST_CHECK_ARG(dev >= 0 && dev < ST_MAX_DEVICES, "device ordinal %d out of range", dev);
  if (!g_work_ptr[dev]) {
    void* p = nullptr;
    ST_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_work_counters)));
    g_work_ptr[dev] = (unsigned long long*)p;
  }
  *out = g_work_ptr[dev];
  return ST_OK;
}
extern "C" int st_work_counters_read(unsigned long long* out4_host, int reset) {
  ST_CHECK_ARG(out4_host != nullptr, "st_work_counters_read: NULL output");
  unsigned long long* p = nullptr;
  ST_TRY(st_work_counters_ptr(&p));
  ST_HIP(hipMemcpy(out4_host, p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) ST_HIP(hipMemset(p, 0, 4 * sizeof(unsigned long long)));
  return ST_OK;
}

// ------------------------------------------------------------------------------------------------
// x[r] = emb[token(s)] * sqrt(d) + pe[pos]   (model.py:126-133 Embeddings, :75-94 PositionalEncoding)
// one wave per live row; also resets the *next* live counter (ping-pong, see st_decode_greedy).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ emb, const float* __restrict__ pe,
                                                    const int* __restrict__ idx, const int* __restrict__ n_dev,
                                                    int* __restrict__ n_next, int* __restrict__ n_amb,
                                                    const int32_t* __restrict__ ys, int T, int pos, int bos, int vocab,
                                                    float* __restrict__ x, unsigned long long* __restrict__ work) {
  const int n = *n_dev;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *n_next = 0;
    *n_amb = 0;
    // executed work of this process (st_work_counters_read): token-steps and attended keys, exact live counts
    atomicAdd(work + 0, (unsigned long long)n);
    atomicAdd(work + 1, (unsigned long long)n * (unsigned long long)(pos + 1));
    if (pos == 0) atomicAdd(work + 2, (unsigned long long)n);
  }
  if (!x) return;  // the first layer's half-layer kernel computes the rows itself (OFArgs.emb)
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const int s = idx[r];
  int tok = pos == 0 ? bos : ys[(size_t)s * T + pos - 1];
  tok = min(max(tok, 0), vocab - 1);
  const f32x4 e = *(const f32x4*)(emb + (size_t)tok * 256 + lane * 4);
  const f32x4 p = *(const f32x4*)(pe + (size_t)pos * 256 + lane * 4);
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = fmaf(e[q], 16.0f, p[q]);  // (the half-layer kernel's EMB form repeats exactly this)
  *(f32x4*)(x + (size_t)r * 256 + lane * 4) = v;
}

// ------------------------------------------------------------------------------------------------
// single-query causal self-attention over the cache (model.py:18-27 attention, :30-60 MHA; h=4, d_k=64)
// block = one live row, wave = head, lane j = key position j (pos+1 <= 64 keys), then lane d = out dim.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_step_kernel(const float* __restrict__ q, const float* __restrict__ kbase,
                                                        const float* __restrict__ vbase, long kv_stride, long pos_stride,
                                                        const int* __restrict__ idx, const int* __restrict__ n_dev,
                                                        int pos, float* __restrict__ out) {
  __shared__ __align__(16) float qs[256];
  const int r = blockIdx.x;
  if (r >= *n_dev) return;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  qs[tid] = q[(size_t)r * 256 + tid];
  __syncthreads();
  const long s = idx[r];
  const float* K = kbase + s * kv_stride + h * 64;
  const float* V = vbase + s * kv_stride + h * 64;
  float sc = -INFINITY;
  if (lane <= pos) {
    const float* kr = K + (size_t)lane * pos_stride;
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      const f32x4 kv = *(const f32x4*)(kr + d);
      const f32x4 qv = *(const f32x4*)(qs + h * 64 + d);
      acc = fmaf(qv.x, kv.x, acc);
      acc = fmaf(qv.y, kv.y, acc);
      acc = fmaf(qv.z, kv.z, acc);
      acc = fmaf(qv.w, kv.w, acc);
    }
    sc = acc * 0.125f;  // / sqrt(d_k)
  }
  const float m = wave_max(sc);
  const float e = lane <= pos ? expf(sc - m) : 0.0f;
  const float p = e / wave_sum(e);
  float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;   // four independent chains keep four V loads in flight
  int j = 0;
  for (; j + 3 <= pos; j += 4) {
    const float v0 = V[(size_t)j * pos_stride + lane], v1 = V[(size_t)(j + 1) * pos_stride + lane];
    const float v2 = V[(size_t)(j + 2) * pos_stride + lane], v3 = V[(size_t)(j + 3) * pos_stride + lane];
    o0 = fmaf(__shfl(p, j, 64), v0, o0);
    o1 = fmaf(__shfl(p, j + 1, 64), v1, o1);
    o2 = fmaf(__shfl(p, j + 2, 64), v2, o2);
    o3 = fmaf(__shfl(p, j + 3, 64), v3, o3);
  }
  for (; j <= pos; ++j) o0 = fmaf(__shfl(p, j, 64), V[(size_t)j * pos_stride + lane], o0);
  out[(size_t)r * 256 + h * 64 + lane] = (o0 + o1) + (o2 + o3);
}


// Short-prefix variant (pos < 16, i.e. almost every step of a yelp decode): ONE wave per row handles all four
// heads.  Every K and V row is read as one coalesced 1 KiB wave load (lane l holds dims 4l..4l+3, which belong
// to head l >> 4); q.k partial sums are reduced over the 16 lanes of a head with DPP row shifts, so every lane
// of a head ends up with all of that head's scores and the softmax needs no further lane traffic; the PV sum
// is lane-local.  All 2(pos+1) loads of a row are independent and issued up front.
// FAST (f16x3 format): softmax exponential through v_exp_f32 (as in the encoder's attention, st_encode.hip).
// NK = pos + 1, the number of keys, is a TEMPLATE parameter (the step is the same for every row of a launch): with a
// run-time pos the unrolled 16-key body kept its softmax and P.V parts for all 16 keys behind selects (96 v_cndmask,
// 16 v_exp per row whatever pos was) and the kernel's floor -- its time with K / V served from L2 -- was ~ 0.6 of its
// time: VALU issue, not HBM.  Same operations in the same order on the keys that exist: identical results.
template <bool FAST, int NK>
__global__ __launch_bounds__(256) void attn_step16_kernel(const float* __restrict__ q, const float* __restrict__ kbase,
                                                          const float* __restrict__ vbase, long kv_stride, long pos_stride,
                                                          const int* __restrict__ idx, const int* __restrict__ n_dev,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= *n_dev) return;
  const long s = idx[r];
  const float* K = kbase + s * kv_stride + lane * 4;
  const float* V = vbase + s * kv_stride + lane * 4;
  const f32x4 qv = *(const f32x4*)(q + (size_t)r * 256 + lane * 4);
  f32x4 kr[NK], vr[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    kr[j] = *(const f32x4*)(K + (size_t)j * pos_stride);
    vr[j] = *(const f32x4*)(V + (size_t)j * pos_stride);
  }
  float sc[NK];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const float part = fmaf(qv.x, kr[j].x, fmaf(qv.y, kr[j].y, fmaf(qv.z, kr[j].z, qv.w * kr[j].w)));
    sc[j] = row16_sum(part) * 0.125f;  // / sqrt(d_k)
    m = fmaxf(m, sc[j]);
  }
  float sum = 0.0f;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    sc[j] = FAST ? __expf(sc[j] - m) : expf(sc[j] - m);
    sum += sc[j];
  }
  const float inv = 1.0f / sum;
  f32x4 o4 = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NK; ++j) o4 += (sc[j] * inv) * vr[j];
  *(f32x4*)(out + (size_t)r * 256 + lane * 4) = o4;
}
#define ST_NK_CASES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
template <bool FAST>
static void launch_attn_step16(int nk, int n_bound, hipStream_t st, const float* q, const float* kb, const float* vb, long kv_stride,
                               long pos_stride, const int* idx, const int* n_dev, float* out) {
  switch (nk) {
#define ST_NK_CASE(N)                                                                                                       \
  case N:                                                                                                                   \
    hipLaunchKernelGGL((attn_step16_kernel<FAST, N>), dim3((n_bound + 3) / 4), dim3(256), 0, st, q, kb, vb, kv_stride, pos_stride, \
                       idx, n_dev, out);                                                                                    \
    break;
    ST_NK_CASES(ST_NK_CASE)
#undef ST_NK_CASE
  }
}

// ------------------------------------------------------------------------------------------------
// First decoder layer with the (token, position) table (StDecoderWeights.l0_table): the layer's q | k | v rows depend
// on (token, position) only, so a decode step reads them from the table instead of projecting them and caching K / V per
// job.  tok(j) = BOS for j = 0, else the token the job emitted at step j - 1.  Same arithmetic as attn_step16_kernel /
// attn_step_kernel on the same values: identical outputs.  The table is shared by all jobs, so frequent (token, position)
// rows are served from L2 / Infinity Cache instead of HBM.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l0_embed_kernel(const float* __restrict__ emb, const float* __restrict__ pe, int vocab,
                                                       int positions, float* __restrict__ x) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)vocab * positions) return;
  const int lane = threadIdx.x & 63;
  const int tok = (int)(r / positions), pos = (int)(r % positions);
  const f32x4 e = *(const f32x4*)(emb + (size_t)tok * 256 + lane * 4);
  const f32x4 p = *(const f32x4*)(pe + (size_t)pos * 256 + lane * 4);
  *(f32x4*)(x + (size_t)r * 256 + lane * 4) = e * 16.0f + p;   // exactly embed_kernel's expression
}

template <bool FAST, int NK>
__global__ __launch_bounds__(256) void attn_step16_tab_kernel(const float* __restrict__ tab, int tab_len,
                                                              const int32_t* __restrict__ ys, int T, int bos, int vocab,
                                                              const int* __restrict__ idx, const int* __restrict__ n_dev,
                                                              float* __restrict__ out) {
  constexpr int pos = NK - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= *n_dev) return;
  const long s = idx[r];
  // lane j holds the token of key position j (j <= pos < 16)
  int tokj = bos;
  if (lane >= 1 && lane <= pos) tokj = ys[(size_t)s * T + lane - 1];
  tokj = min(max(tokj, 0), vocab - 1);
  f32x4 kr[NK], vr[NK];
  f32x4 qv;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const int t = __builtin_amdgcn_readlane(tokj, j);
    const float* row = tab + ((size_t)t * tab_len + j) * 768 + lane * 4;
    kr[j] = *(const f32x4*)(row + 256);
    vr[j] = *(const f32x4*)(row + 512);
    if (j == pos) qv = *(const f32x4*)row;
  }
  float sc[NK];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const float part = fmaf(qv.x, kr[j].x, fmaf(qv.y, kr[j].y, fmaf(qv.z, kr[j].z, qv.w * kr[j].w)));
    sc[j] = row16_sum(part) * 0.125f;  // / sqrt(d_k)
    m = fmaxf(m, sc[j]);
  }
  float sum = 0.0f;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    sc[j] = FAST ? __expf(sc[j] - m) : expf(sc[j] - m);
    sum += sc[j];
  }
  const float inv = 1.0f / sum;
  f32x4 o4 = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NK; ++j) o4 += (sc[j] * inv) * vr[j];
  *(f32x4*)(out + (size_t)r * 256 + lane * 4) = o4;
}
template <bool FAST>
static void launch_attn_step16_tab(int nk, int n_bound, hipStream_t st, const float* tab, int tab_len, const int32_t* ys, int T, int bos,
                                   int vocab, const int* idx, const int* n_dev, float* out) {
  switch (nk) {
#define ST_NK_CASE(N)                                                                                                          \
  case N:                                                                                                                      \
    hipLaunchKernelGGL((attn_step16_tab_kernel<FAST, N>), dim3((n_bound + 3) / 4), dim3(256), 0, st, tab, tab_len, ys, T, bos, vocab, \
                       idx, n_dev, out);                                                                                       \
    break;
    ST_NK_CASES(ST_NK_CASE)
#undef ST_NK_CASE
  }
}

// general form (16 <= pos < 64): attn_step_kernel with the K / V rows of the emitted tokens
__global__ __launch_bounds__(256) void attn_step_tab_kernel(const float* __restrict__ tab, int tab_len,
                                                            const int32_t* __restrict__ ys, int T, int bos, int vocab,
                                                            const int* __restrict__ idx, const int* __restrict__ n_dev,
                                                            int pos, float* __restrict__ out) {
  __shared__ __align__(16) float qs[256];
  __shared__ int toks[64];
  const int r = blockIdx.x;
  if (r >= *n_dev) return;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const long s = idx[r];
  if (tid < 64) {
    int t = bos;
    if (tid >= 1 && tid <= pos) t = ys[(size_t)s * T + tid - 1];
    toks[tid] = min(max(t, 0), vocab - 1);
  }
  __syncthreads();
  qs[tid] = tab[((size_t)toks[pos] * tab_len + pos) * 768 + tid];
  __syncthreads();
  float sc = -INFINITY;
  if (lane <= pos) {
    const float* kr = tab + ((size_t)toks[lane] * tab_len + lane) * 768 + 256 + h * 64;
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      const f32x4 kv = *(const f32x4*)(kr + d);
      const f32x4 qv = *(const f32x4*)(qs + h * 64 + d);
      acc = fmaf(qv.x, kv.x, acc);
      acc = fmaf(qv.y, kv.y, acc);
      acc = fmaf(qv.z, kv.z, acc);
      acc = fmaf(qv.w, kv.w, acc);
    }
    sc = acc * 0.125f;  // / sqrt(d_k)
  }
  const float m = wave_max(sc);
  const float e = lane <= pos ? expf(sc - m) : 0.0f;
  const float p = e / wave_sum(e);
  auto vrow = [&](int j) -> float { return tab[((size_t)toks[j] * tab_len + j) * 768 + 512 + h * 64 + lane]; };
  float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;   // four independent chains, as attn_step_kernel
  int j = 0;
  for (; j + 3 <= pos; j += 4) {
    const float v0 = vrow(j), v1 = vrow(j + 1), v2 = vrow(j + 2), v3 = vrow(j + 3);
    o0 = fmaf(__shfl(p, j, 64), v0, o0);
    o1 = fmaf(__shfl(p, j + 1, 64), v1, o1);
    o2 = fmaf(__shfl(p, j + 2, 64), v2, o2);
    o3 = fmaf(__shfl(p, j + 3, 64), v3, o3);
  }
  for (; j <= pos; ++j) o0 = fmaf(__shfl(p, j, 64), vrow(j), o0);
  out[(size_t)r * 256 + h * 64 + lane] = (o0 + o1) + (o2 + o3);
}

// ------------------------------------------------------------------------------------------------
// Decode QKV projection AND the single-query self-attention of one layer in one kernel (f16x3 format, pos < 16).
// Phase 1 is rp_lin16_kernel<EPI_QKV> without a column split: a workgroup owns a 64-row panel and all 24 column
// blocks; wave w takes its blocks in the order k, k, v, v, q, q.  k and v go to the K/V cache from the accumulators
// (the stores of block i - 1 inside the MFMAs of block i); the two q blocks stay in accumulators until every wave has
// finished reading the panel, then they are written over it in LDS as f32 rows: q never goes to HBM.
// Phase 2 is attn_step16_kernel<true>, one wave per row and 16 rows per wave, q from LDS, K / V rows 0 .. pos from the
// cache (row pos was written in phase 1 by this workgroup: __threadfence + barrier; its lines were never read before
// by this CU, so the L1 holds no stale copy).  Rows are taken RIF at a time so that the loads of several rows are in
// flight together.  Every value is computed by the same instruction sequence as in the two-kernel form: identical
// results.  Two workgroups share a CU (66.8 KiB of LDS each): one's K / V stream runs under the other's MFMAs, which
// is the point - the two kernels it replaces ran at HBM time PLUS MFMA time.
// ------------------------------------------------------------------------------------------------
template <int NK>
__global__ __launch_bounds__(256, 2) void dec_qkv_attn16_kernel(GemmArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* As = smem;  // [64][ST_LDA]
  int* slot = (int*)(smem + 64 * ST_LDA);
  const int n = a.n_dev ? *a.n_dev : a.n_host;
  const int m0 = blockIdx.x * 64;
  if (m0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int pos = a.pos;
  int bad = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = tid + i * 256;
    const int r = f >> 6, c4 = f & 63;
    *(f32x4*)(As + r * ST_LDA + c4 * 4) = *(const f32x4*)(a.X + (size_t)min(m0 + r, n - 1) * 256 + c4 * 4);
  }
  if (tid < 64) slot[tid] = a.idx[min(m0 + tid, n - 1)];
  __syncthreads();
  {
    const f32x4 ga = *(const f32x4*)(a.ln_a + lane * 4);
    const f32x4 gb = *(const f32x4*)(a.ln_b + lane * 4);
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      float* row = As + (wave * 16 + rr) * ST_LDA;  // one wave owns the whole row: in-place is safe
      panel_store4_nb16(row, lane, layernorm256r(*(f32x4*)(row + lane * 4), ga, gb), bad);
    }
  }
  __syncthreads();

  // this wave's six 32-column blocks: k (8 + w, 12 + w), v (16 + w, 20 + w), q (w, 4 + w)
  auto blk_of = [&](int i) -> int { i = min(i, 5); return wave + 4 * (i < 4 ? i + 2 : i - 4); };
  const unsigned wl = (unsigned)(((g >> 1) * 128 + 32 * (g & 1) + r16) * 16);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, 24 * 32768, 0x00020000);
  const float* Ab = As + r16 * ST_LDA + rp16_frag_pos(0, g);
  auto wbase = [&](int i) -> int { return blk_of(i) * 32768; };
  auto wfetch = [&](int p, int d, f32x4 (&dst)[4]) {
    dst[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl, p + d * 4096, 0));
    dst[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1024, p + d * 4096, 0));
    dst[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 256, p + d * 4096, 0));
    dst[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wl + 1280, p + d * 4096, 0));
  };
  size_t kvrow[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) kvrow[rt] = (size_t)((long)slot[16 * rt + r16] * a.kv_stride) + (size_t)pos * 256 + 4 * g;
  const bool full = m0 + 64 <= n;
  auto put = [&](const f32x4& v, int i, int ft, int rt) {  // k / v blocks only (i < 4)
    const int blk = blk_of(i);
    if (!full && m0 + 16 * rt + r16 >= n) return;
    float* base = (blk < 16) ? a.kbase : a.vbase;
    *(f32x4*)(base + (((blk << 5) + 16 * ft) & 255) + kvrow[rt]) = v;
  };
  f32x4 rw[RP_LIN_RING][4];
#pragma unroll
  for (int j = 0; j < RP_LIN_RING; ++j) wfetch(wbase(0), j, rw[j]);
  f32x4 pn[4];
  pn[0] = *(const f32x4*)(Ab);
  pn[1] = *(const f32x4*)(Ab + 128);
  pn[2] = *(const f32x4*)(Ab + 16 * ST_LDA);
  pn[3] = *(const f32x4*)(Ab + 16 * ST_LDA + 128);
  auto block = [&](f32x4 (&acc)[2][4], int i, const f32x4 (&o)[2][4], int io, bool do_put) {
    const int Wb = wbase(i);
    const int Wn = wbase(i + 1);
    {
      const int col = (blk_of(i) << 5) + 4 * g;
      const f32x4 bv0 = *(const f32x4*)(a.bias + col), bv1 = *(const f32x4*)(a.bias + col + 16);
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
  block(A, 0, A, 0, false);
  block(B, 1, A, 0, true);
  block(A, 2, B, 1, true);
  block(B, 3, A, 2, true);
  block(A, 4, B, 3, true);   // A: q block `wave`
  block(B, 5, B, 5, false);  // B: q block `wave + 4`
  if (bad && a.range_flag) *a.range_flag = 1;
  __syncthreads();  // every wave has read its last panel fragment: the panel becomes the q tile (f32 rows)
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      float* qrow = As + (16 * rt + r16) * ST_LDA + 16 * ft + 4 * g;
      *(f32x4*)(qrow + 32 * wave) = A[ft][rt];
      *(f32x4*)(qrow + 32 * (wave + 4)) = B[ft][rt];
    }
  __threadfence();  // this workgroup's new K / V rows are read back below
  __syncthreads();

  // ---- phase 2: attention, one wave per row ----
  constexpr int RIF = NK <= 4 ? 4 : (NK <= 8 ? 2 : 1);
  for (int r0 = 0; r0 < 16; r0 += RIF) {
    f32x4 kr[RIF][NK], vr[RIF][NK];
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int row = wave * 16 + r0 + u;
      const long s = slot[min(row, 63)];
      const float* K = a.kbase + s * a.kv_stride + lane * 4;
      const float* V = a.vbase + s * a.kv_stride + lane * 4;
#pragma unroll
      for (int j = 0; j < NK; ++j)
        if (j <= pos) {  // uniform
          kr[u][j] = *(const f32x4*)(K + (size_t)j * 256);
          vr[u][j] = *(const f32x4*)(V + (size_t)j * 256);
        }
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int row = wave * 16 + r0 + u;
      if (m0 + row >= n) continue;  // uniform
      const f32x4 qv = *(const f32x4*)(As + row * ST_LDA + lane * 4);
      float sc[NK];
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        sc[j] = -INFINITY;
        if (j <= pos) {
          const float part = fmaf(qv.x, kr[u][j].x, fmaf(qv.y, kr[u][j].y, fmaf(qv.z, kr[u][j].z, qv.w * kr[u][j].w)));
          sc[j] = row16_sum(part) * 0.125f;  // / sqrt(d_k)
          m = fmaxf(m, sc[j]);
        }
      }
      float sum = 0.0f;
#pragma unroll
      for (int j = 0; j < NK; ++j)
        if (j <= pos) {
          sc[j] = __expf(sc[j] - m);
          sum += sc[j];
        }
      const float inv = 1.0f / sum;
      f32x4 o4 = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < NK; ++j)
        if (j <= pos) o4 += (sc[j] * inv) * vr[u][j];
      *(f32x4*)(a.out + (size_t)(m0 + row) * 256 + lane * 4) = o4;
    }
  }
}

template <int NK>
static int launch_dec_qkv_attn16_t(const GemmArgs& a, int n_bound, hipStream_t st) {
  ST_ENSURE_LDS((dec_qkv_attn16_kernel<NK>), RP_LDS);
  hipLaunchKernelGGL((dec_qkv_attn16_kernel<NK>), dim3((n_bound + 63) / 64), dim3(256), RP_LDS, st, a);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
static int launch_dec_qkv_attn16(const GemmArgs& a, int n_bound, hipStream_t st) {
  if (n_bound <= 0) return ST_OK;
  if (a.pos < 4) return launch_dec_qkv_attn16_t<4>(a, n_bound, st);
  if (a.pos < 8) return launch_dec_qkv_attn16_t<8>(a, n_bound, st);
  return launch_dec_qkv_attn16_t<16>(a, n_bound, st);
}

// ------------------------------------------------------------------------------------------------
// merge the generator's per-split (best, index, second) partials; emit the token; keep the sentence
// live unless it produced EOS/PAD (early exit) or this was the last step.
//   rows != null: partial row i belongs to live row rows[i] (the redo pass over ambiguous rows).
//   margin_c >= 0: the partials come from the one-product screening pass; partial.w is the 2-norm of the
//   row.  Every screened logit is within E = margin_c/2 * norm (+ slack) of its exact value, so the best
//   column is certainly the exact argmax when it leads the runner-up by more than 2E; otherwise the row
//   is appended to amb_idx and decided by the 3-product kernel.
// ------------------------------------------------------------------------------------------------
// slot in an append-only list for every thread with pred set: ONE atomic per 1024-thread block (all threads must
// call).  With one atomic per wave the ~7 k same-address atomics of a 230 k-row step took most of the kernel's 29 us.
#define FIN_THREADS 1024
__device__ __forceinline__ int block_append(int* counter, bool pred, int* lds /* [FIN_THREADS / 64 + 1] */) {
  const unsigned long long mask = __ballot(pred);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // lds may still be read from a previous call
  if (lane == 0) lds[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < FIN_THREADS / 64; ++w) {
      const int c = lds[w];
      lds[w] = tot;
      tot += c;
    }
    lds[FIN_THREADS / 64] = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  return lds[FIN_THREADS / 64] + lds[wave] + __popcll(mask & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(FIN_THREADS) void finish_kernel(const float4* __restrict__ partial, int nsplit,
                                                     const int* __restrict__ idx, const int* __restrict__ n_dev,
                                                     int32_t* __restrict__ ys, float* __restrict__ gap, int T, int pos,
                                                     int eos, int pad, int early, int last, int* __restrict__ idx_next,
                                                     int* __restrict__ n_next, const int* __restrict__ rows,
                                                     float margin_c, int* __restrict__ amb_idx, int* __restrict__ n_amb,
                                                     const float* __restrict__ wnorm, float cx_c) {
  __shared__ int app[FIN_THREADS / 64 + 1];
  const int i = blockIdx.x * FIN_THREADS + threadIdx.x;
  const bool valid = i < *n_dev;
  Top2 t = {0.0f, 0, 0.0f};
  float norm = 0.0f;
  int r = 0;
  if (valid) {
    const float4 p0 = partial[(size_t)i * nsplit];
    t.m1 = p0.x; t.i1 = __float_as_int(p0.y); t.m2 = p0.z;
    norm = p0.w;
    for (int k = 1; k < nsplit; ++k) {
      const float4 pk = partial[(size_t)i * nsplit + k];
      Top2 u = {pk.x, __float_as_int(pk.y), pk.z};
      t = top2_merge(t, u);
    }
    r = rows ? rows[i] : i;
  }
  // the screening kernels report keys: logits with their low 9 (rp_screen16p_kernel; the other forms: 8) mantissa bits
  // replaced, i.e. each within 2^-14 of its magnitude; the third term covers both values of any pair, whichever split
  // they came from
  const float key_err = (1.01f / 8192.0f) * fmaxf(fabsf(t.m1), fabsf(t.m2));
  // wnorm: the partials are UPPER bounds u_v = s_v + cx ||W_v|| (cx = cx_c ||x||) of the exact logits: the winner's lower
  // bound is u_1 - 2 cx ||W_i1||, and every other column's exact logit is below its own upper bound <= u_2
  const float margin = wnorm ? (valid ? 2.0f * cx_c * norm * wnorm[t.i1] : 0.0f) : margin_c * norm;
  const bool amb = valid && margin_c >= 0.0f && !(t.m1 - t.m2 > margin + 1e-4f + key_err);
  if (margin_c >= 0.0f) {  // uniform
    const int slot = block_append(n_amb, amb, app);
    if (amb) amb_idx[slot] = r;
  }
  const bool emit = valid && !amb;
  int s = 0;
  if (emit) {
    s = idx[r];
    ys[(size_t)s * T + pos] = t.i1;
    if (gap) gap[(size_t)s * T + pos] = t.m1 - t.m2;
  }
  const bool live = emit && !(last || (early && (t.i1 == eos || t.i1 == pad)));
  const int slot = block_append(n_next, live, app);
  if (live) idx_next[slot] = s;
}

__global__ void iota_kernel(int* idx, int* n_dev, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = i;
  if (i == 0) *n_dev = n;
}

// standalone merge for st_gemm_argmax
__global__ __launch_bounds__(256) void argmax_merge_kernel(const float4* __restrict__ partial, int nsplit, int n,
                                                           int32_t* __restrict__ ids, float* __restrict__ best,
                                                           float* __restrict__ gap) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  float4 p0 = partial[(size_t)r * nsplit];
  Top2 t = {p0.x, __float_as_int(p0.y), p0.z};
  for (int k = 1; k < nsplit; ++k) {
    const float4 pk = partial[(size_t)r * nsplit + k];
    Top2 u = {pk.x, __float_as_int(pk.y), pk.z};
    t = top2_merge(t, u);
  }
  ids[r] = t.i1;
  if (best) best[r] = t.m1;
  if (gap) gap[r] = t.m1 - t.m2;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DecodeWs {
  float* kv;       // [B][L][2][T][256]
  float* x;        // [B][256]
  float* q;        // [B][256]
  float* att;      // [B][256]
  float* h;        // [B][d_ff]
  float* cc;       // [L][B][256] cross-attention constants
  float4* partial; // [B][GEN_SPLITS]
  int* idx[2];     // [B]
  int* amb;        // [B] live rows the screening pass could not decide
  int* cnt;        // [0],[1]: live counts (ping-pong), [2]: ambiguous rows of the current step
  float* slab;     // [d_ff/256][64][256] split-K partials of feed_forward.w_2 when one panel of rows is left
  float* colpad;   // [2][32 ceil(vocab / 32)] padded generator bias and column norms (screening pass)
  size_t total;
};
#define ST_GEN_SPLITS 16        /* max column splits of the generator (partials per row) */
#define ST_SCREEN_MIN_ROWS 2048 /* below this the two extra launches of the screened argmax cost more than they save */
#define ST_SYNC_RELAX_ROWS 1024  /* below this many live rows the host reads the live count every 4th step only */
#define ST_GEN_SPLITS_SMALL 96  /* ... when few rows are live: the late steps are latency-bound on streaming W */

// kv_layers: decoder layers that keep a K / V cache (all of them, or all but the first when it reads the (token, position) table)
static DecodeWs carve(const StDecoderWeights* w, int B, int max_len, char* base, int kv_layers) {
  DecodeWs ws;
  const int T = max_len - 1, L = w->n_layers;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  ws.kv = (float*)take((size_t)B * (kv_layers > 0 ? kv_layers : 1) * 2 * T * 256 * 4);
  ws.x = (float*)take((size_t)B * 256 * 4);
  ws.q = (float*)take((size_t)B * 256 * 4);
  ws.att = (float*)take((size_t)B * 256 * 4);
  ws.h = (float*)take((size_t)B * w->d_ff * 4);
  ws.cc = (float*)take((size_t)L * B * 256 * 4);
  ws.partial = (float4*)take(((size_t)B * ST_GEN_SPLITS + ST_GEN_SPLITS_SMALL * 64) * 16);
  ws.idx[0] = (int*)take((size_t)B * 4);
  ws.idx[1] = (int*)take((size_t)B * 4);
  ws.amb = (int*)take((size_t)B * 4);
  ws.cnt = (int*)take(256);
  ws.slab = (float*)take((size_t)(w->d_ff / 256) * 64 * 256 * 4);
  ws.colpad = (float*)take(rp_screen_colpad_floats(w->vocab) * 4);
  ws.total = off;
  return ws;
}

static int check_weights(const StDecoderWeights* w, int max_len) {
  ST_CHECK_ARG(w != nullptr, "weights struct is NULL");
  ST_CHECK_ARG(w->d_model == ST_D, "d_model=%d unsupported (kernels are specialised for %d)", w->d_model, ST_D);
  ST_CHECK_ARG(w->n_heads == 4, "n_heads=%d unsupported (4)", w->n_heads);
  ST_CHECK_ARG(w->d_ff > 0 && w->d_ff % 256 == 0, "d_ff=%d must be a positive multiple of 256", w->d_ff);
  ST_CHECK_ARG(w->n_layers >= 1 && w->n_layers <= ST_MAX_LAYERS, "n_layers=%d out of range", w->n_layers);
  ST_CHECK_ARG(w->vocab >= 4, "vocab=%d too small", w->vocab);
  ST_CHECK_ARG(w->gemm_format == ST_GEMM_F32 || w->gemm_format == ST_GEMM_F16X3, "gemm_format=%d unknown", w->gemm_format);
  ST_CHECK_ARG(max_len >= 2 && max_len - 1 <= 64, "max_len=%d unsupported (decode length max_len-1 must be <= 64)", max_len);
  ST_CHECK_ARG(w->pe_rows >= max_len - 1, "pe_rows=%d < decode length %d", w->pe_rows, max_len - 1);
  ST_CHECK_ARG(w->emb && w->pe && w->lnf_a && w->lnf_b && w->gen_w && w->gen_b, "NULL weight pointer");
  for (int l = 0; l < w->n_layers; ++l) {
    const StDecLayer& y = w->layer[l];
    ST_CHECK_ARG(y.ln0_a && y.ln0_b && y.wqkv && y.bqkv && y.wo && y.bo && y.xv_w && y.xv_b && y.xo_w && y.xo_b &&
                     y.ln2_a && y.ln2_b && y.w1 && y.b1 && y.w2 && y.b2,
                 "NULL weight pointer in layer %d", l);
  }
  return ST_OK;
}

static bool dec_uses_l0_table(const StDecoderWeights* w, int max_len, int options) {
  return w->l0_table != nullptr && w->l0_table_len >= max_len - 1 && w->gemm_format == ST_GEMM_F16X3 && !(options & ST_DEC_NO_L0_TABLE);
}
extern "C" size_t st_decode_workspace_bytes_opt(const StDecoderWeights* w, int B, int max_len, int options) {
  if (!w || B <= 0 || max_len < 2) return 0;
  return carve(w, B, max_len, nullptr, w->n_layers - (dec_uses_l0_table(w, max_len, options) ? 1 : 0)).total;
}
extern "C" size_t st_decode_workspace_bytes(const StDecoderWeights* w, int B, int max_len) {
  return st_decode_workspace_bytes_opt(w, B, max_len, ST_DEC_NO_L0_TABLE);   // enough for every option
}

extern "C" int st_decode_greedy(const StDecoderWeights* w, const float* latent, int B, int max_len, int bos, int eos,
                                int pad, int options, int sync_every, int32_t* out_ids, float* top2gap,
                                void* workspace, size_t workspace_bytes, int* n_steps_host, void* stream) {
  ST_TRY(check_weights(w, max_len));
  ST_CHECK_ARG(B >= 0, "B=%d", B);
  rp_ab_env();
  if (n_steps_host) *n_steps_host = 0;
  if (B == 0) return ST_OK;
  ST_CHECK_ARG(latent && out_ids && workspace, "NULL buffer");
  ST_CHECK_ARG(sync_every >= 1, "sync_every=%d must be >= 1", sync_every);
  hipStream_t st = (hipStream_t)stream;
  const int T = max_len - 1, L = w->n_layers;
  const bool use_tab = dec_uses_l0_table(w, max_len, options);
  const int kv_first = use_tab ? 1 : 0, kv_layers = L - kv_first;   // the first layer keeps no K / V cache when it reads the table
  DecodeWs ws = carve(w, B, max_len, (char*)workspace, kv_layers);
  if (ws.total > workspace_bytes) {
    st_set_error("workspace too small: need %zu bytes, have %zu", ws.total, workspace_bytes);
    return ST_ENOMEM;
  }
  const int early = (options & ST_DEC_EARLY_EXIT) ? 1 : 0;
  // K / V cache layout.  Job-major (default): [job][layer][k | v][position][256]: a decode step writes 1 KB per job, layer and
  // side and reads (pos + 1) KB, at a stride of ~ 120 KB across up to 140 GB.  Position-major (ST_AB_KV_POS_MAJOR):
  // [layer][k | v][position][job][256]: a step writes one contiguous slab of job rows and reads pos + 1 such slabs front to
  // back.  Built to test whether page (TLB) locality holds the attention and QKV kernels back: it does not -- same-box
  // 184.9 / 185.5 k against 184.8 / 185.5 k sentences/s, identical results (profiles/r03_round_end_ab.txt).
  const bool kv_pm = rp_kv_pos_major != 0;
  const long kv_stride = kv_pm ? 256 : (long)(kv_layers > 0 ? kv_layers : 1) * 2 * T * 256;   // floats between two jobs
  const long kv_pos_stride = kv_pm ? (long)B * 256 : 256;                                       // floats between two positions
  unsigned long long* work = nullptr;
  ST_TRY(st_work_counters_ptr(&work));

  ST_HIP(hipMemsetAsync(out_ids, 0, (size_t)B * T * 4, st));  // PAD after an early exit
  if (top2gap) ST_HIP(hipMemsetAsync(top2gap, 0, (size_t)B * T * 4, st));
  ST_HIP(hipMemsetAsync(ws.cnt, 0, 256, st));  // counters and the f16-range flag (cnt[3])
  hipLaunchKernelGGL(iota_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ws.idx[0], ws.cnt + 0, B);
  ST_LAUNCH_CHECK();

  // cross-attention constants: cc[l] = (latent Wv^T + bv) Wo^T + bo   (one key => softmax == 1)
  for (int l = 0; l < L; ++l) {
    GemmArgs g = {};
    g.fmt = w->gemm_format; g.range_flag = ws.cnt + 3;
    g.X = latent; g.n_host = B; g.W = w->layer[l].xv_w; g.bias = w->layer[l].xv_b; g.N = 256;
    g.out = ws.att; g.ldo = 256;
    ST_TRY(launch_rp<EPI_BIAS>(g, B, 1, st));
    g.X = ws.att; g.W = w->layer[l].xo_w; g.bias = w->layer[l].xo_b; g.out = ws.cc + (size_t)l * B * 256;
    ST_TRY(launch_rp<EPI_BIAS>(g, B, 1, st));
  }

  // Screened argmax (f16x3 only; not when the caller wants the exact top-2 gaps): every screened logit is
  // within (2^-10 + 2^-16) * ||x|| * max_v ||W_v|| of the exact one (f16 rounding of both operands + f32
  // accumulation); 25 % slack on top.  margin_c multiplies the row norm in finish_kernel.
  const bool screen_ok = w->gemm_format == ST_GEMM_F16X3 && top2gap == nullptr && w->gen_w_maxnorm > 0.0f &&
                         !(options & ST_DEC_NO_SCREEN);
  const float margin_c = 2.0f * 1.25f * (1.0f / 1024.0f + 1.0f / 65536.0f) * w->gen_w_maxnorm;
  // padded generator bias / column norms: the 16x16x32 generator kernels (screening pass, exact argmax) load them 4 at a time
  if (w->gemm_format == ST_GEMM_F16X3) ST_TRY(launch_rp_screen_colpad(w->gen_b, w->gen_w_norm, w->vocab, ws.colpad, st));
  // One kernel for a layer's QKV projection + attention (f16x3, pos < 16): MEASURED NEGATIVE, off unless ST_AB_DEC_FUSED is
  // set (same-box A/B, profiles/r03_dec_fused_negative.txt: identical results, 130 k against 145 k sentences/s).  Its
  // attention phase has 8 waves x 1-4 rows in flight per CU and each wave walks its 16 rows in sequential HBM round trips;
  // attn_step16_kernel keeps 12 independent rows per CU in flight all the time.
  const bool fuse_qkv_attn = w->gemm_format == ST_GEMM_F16X3 && rp_dec_fused && !kv_pm && kv_stride % 4 == 0 && ((size_t)ws.kv & 15) == 0;
  const int fuse_min_rows = 1;
  int n_bound = B;
  int cur = 0;
  int steps = 0;
  int since_sync = 0;
  static const bool trace_steps = getenv("ST_TRACE_STEPS") != nullptr;  // debug: live rows per step on stderr
  if (trace_steps) fprintf(stderr, "[st_decode] B=%d live:", B);
  for (int pos = 0; pos < T && n_bound > 0; ++pos) {
    const int* idx = ws.idx[cur];
    const int* n_dev = ws.cnt + cur;
    // With the first layer on the table nothing reads the embedded rows but that layer's half-layer kernel (its residual):
    // it computes them itself from the tokens, and the embed kernel only keeps the step's counters (1 KB per row less
    // written and read).  Below RP_TAIL_MIN_ROWS the unfused chain runs and x is written as before.
    const bool emb_in_tail = use_tab && w->gemm_format == ST_GEMM_F16X3 && n_bound >= RP_TAIL_MIN_ROWS && rp_tail_shape16 && !rp_no_emb_in_tail;
    hipLaunchKernelGGL(embed_kernel, dim3(emb_in_tail ? 1 : (n_bound + 3) / 4), dim3(256), 0, st, w->emb, w->pe, idx, n_dev,
                       ws.cnt + (cur ^ 1), ws.cnt + 2, out_ids, T, pos, bos, w->vocab, emb_in_tail ? (float*)nullptr : ws.x, work);
    ST_LAUNCH_CHECK();
    for (int l = 0; l < L; ++l) {
      const StDecLayer& y = w->layer[l];
      const int lk = l - kv_first;   // this layer's place in the K / V cache (-1: the table layer keeps none)
      const size_t side = (size_t)T * 256 * (kv_pm ? (size_t)B : 1);   // floats of one layer's K (or V) block
      float* kb = lk >= 0 ? ws.kv + (size_t)(lk * 2 + 0) * side : nullptr;
      float* vb = lk >= 0 ? ws.kv + (size_t)(lk * 2 + 1) * side : nullptr;
      if (l == 0 && use_tab) {
        // first layer: q | k | v of (token, position) from the table; no projection, no K / V cache rows
        if (pos < 16)
          launch_attn_step16_tab<true>(pos + 1, n_bound, st, w->l0_table, w->l0_table_len, out_ids, T, bos, w->vocab, idx, n_dev, ws.att);
        else
          hipLaunchKernelGGL(attn_step_tab_kernel, dim3(n_bound), dim3(256), 0, st, w->l0_table, w->l0_table_len, out_ids, T, bos,
                             w->vocab, idx, n_dev, pos, ws.att);
        ST_LAUNCH_CHECK();
      } else {
      GemmArgs g = {};
      g.fmt = w->gemm_format; g.range_flag = ws.cnt + 3;
      g.X = ws.x; g.n_dev = n_dev; g.n_host = n_bound; g.ln_a = y.ln0_a; g.ln_b = y.ln0_b;
      g.W = y.wqkv; g.bias = y.bqkv; g.N = 768; g.idx = idx; g.q = ws.q;
      g.kbase = kb + (size_t)pos * kv_pos_stride; g.vbase = vb + (size_t)pos * kv_pos_stride; g.kv_stride = kv_stride;
      const int qsplits = rp_splits(n_bound, 768, n_bound <= 2048 ? 8 : 3, rp_nw(w->gemm_format, 8));
      if (fuse_qkv_attn && pos < 16 && n_bound >= fuse_min_rows) {
        g.kbase = kb; g.vbase = vb; g.pos = pos; g.out = ws.att;   // (bases without the pos offset)
        ST_TRY(launch_dec_qkv_attn16(g, n_bound, st));
      } else {
      if (w->gemm_format == ST_GEMM_F16X3)
        ST_TRY(launch_rp_lin_pipelined<EPI_QKV>(g, n_bound, qsplits, st));  // stores of block i-1 inside the MFMAs of block i
      else
        ST_TRY((launch_rp<EPI_QKV, 8>(g, n_bound, qsplits, st)));
      if (pos < 16 && w->gemm_format == ST_GEMM_F16X3)
        launch_attn_step16<true>(pos + 1, n_bound, st, ws.q, kb, vb, kv_stride, kv_pos_stride, idx, n_dev, ws.att);
      else if (pos < 16)
        launch_attn_step16<false>(pos + 1, n_bound, st, ws.q, kb, vb, kv_stride, kv_pos_stride, idx, n_dev, ws.att);
      else
        hipLaunchKernelGGL(attn_step_kernel, dim3(n_bound), dim3(256), 0, st, ws.q, kb, vb, kv_stride, kv_pos_stride, idx, n_dev, pos,
                           ws.att);
      ST_LAUNCH_CHECK();
      }
      }
      OFArgs of = {};
      of.fmt = w->gemm_format; of.range_flag = ws.cnt + 3;
      of.att = ws.att; of.n_dev = n_dev; of.n_host = n_bound; of.wo = y.wo; of.bo = y.bo; of.x = ws.x;
      of.cconst = ws.cc + (size_t)l * B * 256; of.idx = idx; of.ln_a = y.ln2_a; of.ln_b = y.ln2_b;
      of.w1 = y.w1; of.b1 = y.b1; of.h = ws.h; of.d_ff = w->d_ff; of.w2 = y.w2; of.b2 = y.b2;
      if (l == 0 && emb_in_tail) {
        of.emb = w->emb; of.pe_row = w->pe + (size_t)pos * 256; of.ys = out_ids; of.T = T; of.pos = pos; of.bos = bos; of.vocab = w->vocab;
      }
      if (w->gemm_format == ST_GEMM_F16X3 && n_bound >= RP_TAIL_MIN_ROWS) {
        ProfPair pt;
        if (g_prof_on == ST_PROF_HALF_LAYER)
          ST_TRY(prof_begin(&pt, 2.0 * 256.0 * (256.0 + 2.0 * (double)w->d_ff), n_dev, n_bound, st));
        ST_TRY(launch_layer_tail(of, n_bound, st));  // O-projection + FFN in one kernel
        if (g_prof_on == ST_PROF_HALF_LAYER) ST_TRY(prof_end(&pt, st));
        continue;
      }
      if (n_bound >= RP_FUSE_MIN_ROWS) {
        ST_TRY(launch_oproj_ffn1(of, n_bound, st));
      } else {
        GemmArgs o = {};
        o.fmt = w->gemm_format; o.range_flag = ws.cnt + 3;
        o.X = ws.att; o.n_dev = n_dev; o.n_host = n_bound; o.W = y.wo; o.bias = y.bo; o.N = 256;
        o.out = ws.x; o.ldo = 256; o.resid = ws.x; o.cconst = ws.cc + (size_t)l * B * 256; o.ld_cc = 256; o.idx = idx;
        ST_TRY(launch_rp<EPI_RESID>(o, n_bound, rp_splits(n_bound, 256, 2), st));
        GemmArgs f1 = {};
        f1.fmt = w->gemm_format; f1.range_flag = ws.cnt + 3;
        f1.X = ws.x; f1.n_dev = n_dev; f1.n_host = n_bound; f1.ln_a = y.ln2_a; f1.ln_b = y.ln2_b;
        f1.W = y.w1; f1.bias = y.b1; f1.N = w->d_ff; f1.out = ws.h; f1.ldo = w->d_ff;
        ST_TRY((launch_rp<EPI_RELU, 8>(f1, n_bound, rp_splits(n_bound, w->d_ff, n_bound <= 2048 ? 8 : 4, rp_nw(w->gemm_format, 8)), st)));
      }
      GemmArgs f2 = {};
      f2.fmt = w->gemm_format; f2.range_flag = ws.cnt + 3;
      f2.X = ws.h; f2.n_dev = n_dev; f2.n_host = n_bound; f2.K = w->d_ff; f2.W = y.w2; f2.bias = y.b2; f2.N = 256;
      f2.out = ws.x; f2.ldo = 256; f2.resid = ws.x;
      if (n_bound <= 64 && w->d_ff > 256) {
        // one panel left (the straggler tail): a single workgroup would stream all of W2 (1 MiB) by itself, ~44 us;
        // one workgroup per 256-wide K chunk + a combine over the live rows takes a third of that
        const int ks = w->d_ff / 256;
        f2.out = ws.slab;
        ST_TRY(launch_rp_kchunk<EPI_RESID>(f2, n_bound, st, ks));
        hipLaunchKernelGGL(slab_reduce_rows_kernel<0>, dim3((n_bound * 64 + 255) / 256), dim3(256), 0, st, ws.slab, ks,
                           (size_t)n_bound * 256, ws.x, n_dev, 256);
        ST_LAUNCH_CHECK();
      } else {
        ST_TRY(launch_rp_kchunk<EPI_RESID>(f2, n_bound, st));
      }
    }
    GemmArgs gg = {};
    gg.fmt = w->gemm_format; gg.range_flag = ws.cnt + 3;
    gg.X = ws.x; gg.n_dev = n_dev; gg.n_host = n_bound; gg.ln_a = w->lnf_a; gg.ln_b = w->lnf_b;
    gg.W = w->gen_w; gg.bias = w->gen_b; gg.N = w->vocab; gg.partial = ws.partial;
    gg.colpad = w->gemm_format == ST_GEMM_F16X3 ? ws.colpad : nullptr;
    // partial buffer: B * ST_GEN_SPLITS + 64 * ST_GEN_SPLITS_SMALL entries
    const long room = ((long)B * ST_GEN_SPLITS + 64L * ST_GEN_SPLITS_SMALL) / n_bound;
    const int gsplits = rp_splits(n_bound, w->vocab, (int)(room < ST_GEN_SPLITS_SMALL ? (room < ST_GEN_SPLITS ? ST_GEN_SPLITS : room) : ST_GEN_SPLITS_SMALL));
    const int last = pos == T - 1 ? 1 : 0;
    // (a vocabulary beyond 32768 would not fit the block number into the screening kernel's 8 key bits: no screening)
    const bool screen = screen_ok && n_bound >= ST_SCREEN_MIN_ROWS && (w->vocab + 127) / 128 <= RS_MAX_BLOCKS;
    ProfPair pp;
    if (g_prof_on == ST_PROF_GENERATOR) ST_TRY(prof_begin(&pp, 2.0 * 256.0 * (double)w->vocab, n_dev, n_bound, st));
    if (screen) {
      // pass 1: one-product logits for every live row; rows with a clear winner are finished
      // per-column bound (gen_w_norm): the screening kernel books upper bounds, finish_kernel uses the winner's own norm
      const float cx_c = 1.25f * (1.0f / 1024.0f + 1.0f / 65536.0f);
      const float* wnorm = (rp_screen_shape16 && w->gen_w_norm && !rp_screen_maxnorm) ? w->gen_w_norm : nullptr;
      gg.wnorm = wnorm; gg.scale = cx_c;
      ST_TRY(launch_rp_screen_pipelined(gg, n_bound, gsplits, st));  // bookkeeping interleaved with the next block's MFMAs
      gg.wnorm = nullptr;
      hipLaunchKernelGGL(finish_kernel, dim3((n_bound + FIN_THREADS - 1) / FIN_THREADS), dim3(FIN_THREADS), 0, st, ws.partial, gsplits, idx, n_dev,
                         out_ids, (float*)nullptr, T, pos, eos, pad, early, last, ws.idx[cur ^ 1], ws.cnt + (cur ^ 1),
                         (const int*)nullptr, margin_c, ws.amb, ws.cnt + 2, wnorm, cx_c);
      ST_LAUNCH_CHECK();
      // pass 2: the undecided rows again with the 3-product kernel (row gather through ws.amb; the grid covers
      // the worst case, workgroups beyond the ambiguous count exit at once)
      gg.xrows = ws.amb; gg.n_dev = ws.cnt + 2;
      const int rsplits = rp_splits((n_bound + 15) / 16, w->vocab, ST_GEN_SPLITS);
      ST_TRY(launch_rp<EPI_ARGMAX>(gg, n_bound, rsplits, st));
      hipLaunchKernelGGL(finish_kernel, dim3((n_bound + FIN_THREADS - 1) / FIN_THREADS), dim3(FIN_THREADS), 0, st, ws.partial, rsplits, idx, ws.cnt + 2,
                         out_ids, (float*)nullptr, T, pos, eos, pad, early, last, ws.idx[cur ^ 1], ws.cnt + (cur ^ 1),
                         (const int*)ws.amb, -1.0f, (int*)nullptr, (int*)nullptr, (const float*)nullptr, 0.0f);
      ST_LAUNCH_CHECK();
    } else {
      ST_TRY(launch_rp<EPI_ARGMAX>(gg, n_bound, gsplits, st));
      hipLaunchKernelGGL(finish_kernel, dim3((n_bound + FIN_THREADS - 1) / FIN_THREADS), dim3(FIN_THREADS), 0, st, ws.partial, gsplits, idx, n_dev,
                         out_ids, top2gap, T, pos, eos, pad, early, last, ws.idx[cur ^ 1], ws.cnt + (cur ^ 1),
                         (const int*)nullptr, -1.0f, (int*)nullptr, (int*)nullptr, (const float*)nullptr, 0.0f);
      ST_LAUNCH_CHECK();
    }
    if (g_prof_on == ST_PROF_GENERATOR) ST_TRY(prof_end(&pp, st));
    cur ^= 1;
    ++steps;
    // The host reads the live count to size the next grids.  With few rows left (the straggler tail: a handful of
    // decodes that never emit EOS) a step is ~0.25 ms and the read-back costs a ~50 us bubble, so it is done every
    // 4th step there: the kernels take n from n_dev and workgroups past it exit at once, so a stale (larger) bound
    // changes no result; the price is at most 3 empty steps after the last row has finished.
    ++since_sync;
    const int every = n_bound < ST_SYNC_RELAX_ROWS ? (sync_every > 4 ? sync_every : 4) : sync_every;
    if (early && since_sync >= every && pos + 1 < T) {
      since_sync = 0;
      int live = 0;
      ST_HIP(hipMemcpyAsync(&live, ws.cnt + cur, sizeof(int), hipMemcpyDeviceToHost, st));
      ST_HIP(hipStreamSynchronize(st));
      n_bound = live;
      if (trace_steps) fprintf(stderr, " %d", live);
    }
  }
  if (trace_steps) fprintf(stderr, "\n");
  if (n_steps_host) *n_steps_host = steps;
  if (w->gemm_format == ST_GEMM_F16X3) {
    int flag = 0;
    ST_HIP(hipMemcpyAsync(&flag, ws.cnt + 3, sizeof(int), hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    if (flag) {
      st_set_error("st_decode_greedy: an activation left the f16 range (|x| >= 65504 or NaN) in the f16x3 format; "
                   "the decoded ids are invalid - pack the weights with ST_GEMM_F32");
      return ST_ERANGE;
    }
  }
  return ST_OK;
}

extern "C" int st_gemm_argmax(const float* h, const float* ln_a, const float* ln_b, const float* W, const float* b,
                              int n, int V, int gemm_format, int32_t* ids, float* best, float* gap, void* scratch,
                              void* stream) {
  ST_CHECK_ARG(n >= 0 && V >= 1, "n=%d V=%d", n, V);
  ST_CHECK_ARG(gemm_format == ST_GEMM_F32 || gemm_format == ST_GEMM_F16X3, "gemm_format=%d unknown", gemm_format);
  if (n == 0) return ST_OK;
  ST_CHECK_ARG(h && W && b && ids && scratch, "NULL buffer");
  ST_CHECK_ARG((ln_a == nullptr) == (ln_b == nullptr), "ln_a and ln_b must both be given or both NULL");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs gg = {};
  gg.fmt = gemm_format;
  gg.X = h; gg.n_host = n; gg.ln_a = ln_a; gg.ln_b = ln_b; gg.W = W; gg.bias = b; gg.N = V;
  gg.partial = (float4*)scratch;
  // a vocabulary of whole 32-column blocks needs no padded bias copy: the 16x16x32 kernel can read b itself
  if (gemm_format == ST_GEMM_F16X3 && V % 32 == 0 && ((size_t)b & 15) == 0) gg.colpad = b;
  ST_TRY(launch_rp<EPI_ARGMAX>(gg, n, ST_ARGMAX_SPLITS, st));
  hipLaunchKernelGGL(argmax_merge_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float4*)scratch,
                     ST_ARGMAX_SPLITS, n, ids, best, gap);
  ST_LAUNCH_CHECK();
  return ST_OK;
}

// ------------------------------------------------------------------------------------------------
// st_gemm_argmax without any allocation and without a caller-supplied scratch buffer: the per-split
// partials live in a file-scope device array.  Every GPU holds its own copy of the array at its own
// address, so the address is looked up once per device (hipGetSymbolAddress with that device current)
// and host code only ever uses that pointer, never the array's name.
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[(size_t)ST_NOALLOC_MAX_ROWS * ST_ARGMAX_SPLITS * 16];
static void* g_scratch_ptr[ST_MAX_DEVICES] = {};

static int scratch_for_current_device(void** out) {
  int dev = 0;
  ST_HIP(hipGetDevice(&dev));
  ST_CHECK_ARG(dev >= 0 && dev < ST_MAX_DEVICES, "device ordinal %d out of range", dev);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    ST_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    ST_CHECK_ARG(p != nullptr, "hipGetSymbolAddress returned a null address for the scratch array");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return ST_OK;
}

extern "C" int st_gemm_argmax_noalloc(const float* h, const float* ln_a, const float* ln_b, const float* W,
                                      const float* b, int n, int V, int gemm_format, int32_t* ids, float* best,
                                      float* gap, void* stream) {
  ST_CHECK_ARG(n >= 0 && n <= ST_NOALLOC_MAX_ROWS,
               "st_gemm_argmax_noalloc: n=%d exceeds the static scratch (ST_NOALLOC_MAX_ROWS=%d rows); "
               "split the call or use st_gemm_argmax with a caller-owned scratch buffer",
               n, ST_NOALLOC_MAX_ROWS);
  void* sp = nullptr;
  ST_TRY(scratch_for_current_device(&sp));
  // every (row < n, split) partial is rewritten by the GEMM before the merge reads it: nothing has to start at zero
  return st_gemm_argmax(h, ln_a, ln_b, W, b, n, V, gemm_format, ids, best, gap, sp, stream);
}

extern "C" size_t st_packed_weight_floats(int N, int K) { return (N > 0 && K > 0 && K % 8 == 0) ? rp_packed_floats(N, K) : 0; }

extern "C" int st_pack_weight(const float* W, int N, int K, int gemm_format, float* out, void* stream) {
  ST_CHECK_ARG(W && out, "NULL buffer");
  if (gemm_format == ST_GEMM_F16X3) {
    ST_CHECK_ARG(N > 0 && K > 0 && K % 16 == 0, "N=%d K=%d (K must be a positive multiple of 16)", N, K);
    const size_t total = (size_t)((N + 31) / 32) * (K >> 4) * 64;
    hipLaunchKernelGGL(pack_weight_h_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, N, K,
                       out);
  } else {
    ST_CHECK_ARG(gemm_format == ST_GEMM_F32, "gemm_format=%d unknown", gemm_format);
    ST_CHECK_ARG(N > 0 && K > 0 && K % 8 == 0, "N=%d K=%d (K must be a positive multiple of 8)", N, K);
    const size_t total = (size_t)((N + 31) / 32) * (K >> 3) * 64;
    hipLaunchKernelGGL(pack_weight_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, N, K,
                       out);
  }
  ST_LAUNCH_CHECK();
  return ST_OK;
}

extern "C" size_t st_decode_l0_table_bytes(const StDecoderWeights* w, int positions) {
  if (!w || positions <= 0 || w->vocab <= 0) return 0;
  return (size_t)w->vocab * positions * 768 * 4;
}

extern "C" int st_decode_build_l0_table(const StDecoderWeights* w, int positions, float* table, void* scratch,
                                        size_t scratch_bytes, void* stream) {
  ST_TRY(check_weights(w, positions + 1));
  ST_CHECK_ARG(table && scratch, "NULL buffer");
  ST_CHECK_ARG(w->gemm_format == ST_GEMM_F16X3, "the first-layer table is used by the f16x3 format only");
  const long rows = (long)w->vocab * positions;
  ST_CHECK_ARG(rows < (1L << 31), "vocab * positions = %ld rows: too large", rows);
  ST_CHECK_ARG(scratch_bytes >= (size_t)rows * 1024, "scratch too small: need %zu bytes", (size_t)rows * 1024);
  hipStream_t st = (hipStream_t)stream;
  float* x = (float*)scratch;
  int* flag = nullptr;
  ST_HIP(hipMalloc(&flag, sizeof(int)));
  ST_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
  hipLaunchKernelGGL(l0_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, w->emb, w->pe, w->vocab, positions, x);
  ST_LAUNCH_CHECK();
  // the decode step's own kernel (rp_lin16_kernel, LayerNorm prologue, bias in the accumulator): same values bit for bit
  const StDecLayer& y = w->layer[0];
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
    st_set_error("st_decode_build_l0_table: an activation left the f16 range; pack the weights with ST_GEMM_F32");
    return ST_ERANGE;
  }
  return ST_OK;
}
