// Text-change detection on token ids (replaces id2text_sentence + the string compare and the
// Change/iteration bookkeeping of contrastive_attack_one,
// /root/reference/method/yelp/data.py:43-52 and siamese_contrastive.py:816-839).
//
// Two decodes give the same TEXT iff their id prefixes up to the first EOS(3)/PAD(0) agree after
// (a) ids >= vocab -> UNK and (b) mapping ids that spell the same word to one canonical id.
// One WAVE per sentence, lane = position (rows of <= 64 ids: every maximum sequence length the reference is run with); a
// thread-per-sentence form walks longer rows.
#include "st_common.h"

int g_accept_thread_form = 0;  // 1: the thread-per-sentence kernel for every row length (tests compare the two forms)

__device__ __forceinline__ int canon_id(int t, int vocab, int unk, const int32_t* canon) {
  if (t >= vocab || t < 0) t = unk;
  return canon ? canon[t] : t;
}

__global__ __launch_bounds__(256) void accept_kernel(const int32_t* __restrict__ new_ids, int32_t* __restrict__ acc_ids,
                                                     int T, int B, int vocab, int unk, int eos, int pad,
                                                     const int32_t* __restrict__ canon, int max_change,
                                                     int32_t* __restrict__ n_changes, uint8_t* __restrict__ changed,
                                                     uint8_t* __restrict__ active, int32_t* __restrict__ rec_ids,
                                                     const int32_t* __restrict__ confirm_ids) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B) return;
  const bool live = active[s] != 0;
  const int32_t* nw = new_ids + (size_t)s * T;
  int32_t* ac = acc_ids + (size_t)s * T;
  bool diff = false;
  if (live) {
    for (int i = 0; i < T; ++i) {
      const int a = nw[i], b = ac[i];
      const bool ea = (a == eos || a == pad), eb = (b == eos || b == pad);
      if (ea || eb) {
        diff = ea != eb;
        break;
      }
      if (canon_id(a, vocab, unk, canon) != canon_id(b, vocab, unk, canon)) {
        diff = true;
        break;
      }
    }
    if (diff && confirm_ids) {
      // --retext (siamese_contrastive.py:819-825): the new text counts only if re-encoding and re-decoding
      // it reproduces it (re_generator_text == generator_text)
      const int32_t* cf = confirm_ids + (size_t)s * T;
      for (int i = 0; i < T; ++i) {
        const int a = nw[i], b = cf[i];
        const bool ea = (a == eos || a == pad), eb = (b == eos || b == pad);
        if (ea || eb) {
          if (ea != eb) diff = false;
          break;
        }
        if (canon_id(a, vocab, unk, canon) != canon_id(b, vocab, unk, canon)) {
          diff = false;
          break;
        }
      }
    }
    if (diff) {
      for (int i = 0; i < T; ++i) ac[i] = nw[i];
      const int k = n_changes[s] + 1;
      n_changes[s] = k;
      if (k >= max_change) active[s] = 0;
    }
  }
  changed[s] = diff ? 1 : 0;
  if (rec_ids && live)
    for (int i = 0; i < T; ++i) rec_ids[(size_t)s * T + i] = ac[i];
}

// The same decisions with a wave per sentence (T <= 64).  The loop above stops at the first position that holds a
// terminator on either side (texts differ iff only one side ends there) or, before that, at the first position whose
// canonical ids differ: both "firsts" are the lowest set bits of two ballots.  Round 3: the thread-per-sentence walk is a
// chain of up to 59 dependent loads at a 236-byte stride (74 us per call at 98 304 sentences, 22 us at 256).
__device__ __forceinline__ bool texts_differ_wave(int a, int b, bool in_row, int vocab, int unk, int eos, int pad, const int32_t* canon) {
  const bool ea = in_row && (a == eos || a == pad), eb = in_row && (b == eos || b == pad);
  const bool term = ea || eb;
  const bool neq = in_row && !term && canon_id(a, vocab, unk, canon) != canon_id(b, vocab, unk, canon);
  const unsigned long long tm = __ballot(term), nm = __ballot(neq), om = __ballot(ea != eb);
  const unsigned long long before = tm ? ((tm & (~tm + 1)) - 1) : ~0ull;  // positions in front of the first terminator
  if (nm & before) return true;
  return tm ? ((om & (tm & (~tm + 1))) != 0) : false;
}
__global__ __launch_bounds__(256) void accept_wave_kernel(const int32_t* __restrict__ new_ids, int32_t* __restrict__ acc_ids,
                                                          int T, int B, int vocab, int unk, int eos, int pad,
                                                          const int32_t* __restrict__ canon, int max_change,
                                                          int32_t* __restrict__ n_changes, uint8_t* __restrict__ changed,
                                                          uint8_t* __restrict__ active, int32_t* __restrict__ rec_ids,
                                                          const int32_t* __restrict__ confirm_ids) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= B) return;
  const bool live = active[s] != 0, in_row = lane < T;
  bool diff = false;
  int a = 0, b = 0;
  if (live) {  // wave-uniform
    a = in_row ? new_ids[(size_t)s * T + lane] : 0;
    b = in_row ? acc_ids[(size_t)s * T + lane] : 0;
    diff = texts_differ_wave(a, b, in_row, vocab, unk, eos, pad, canon);
    if (diff && confirm_ids) {
      const int c = in_row ? confirm_ids[(size_t)s * T + lane] : 0;
      if (texts_differ_wave(a, c, in_row, vocab, unk, eos, pad, canon)) diff = false;
    }
    if (diff) {
      if (in_row) acc_ids[(size_t)s * T + lane] = a;
      if (lane == 0) {
        const int k = n_changes[s] + 1;
        n_changes[s] = k;
        if (k >= max_change) active[s] = 0;
      }
    }
    if (rec_ids && in_row) rec_ids[(size_t)s * T + lane] = diff ? a : b;
  }
  if (lane == 0) changed[s] = diff ? 1 : 0;
}

extern "C" int st_accept_changes(const int32_t* new_ids, int32_t* acc_ids, int T, int B, int vocab, int unk, int eos,
                                 int pad, const int32_t* canon, int max_change, int32_t* n_changes, uint8_t* changed,
                                 uint8_t* active, int32_t* rec_ids, const int32_t* confirm_ids, void* stream) {
  ST_CHECK_ARG(B >= 0 && T >= 1, "B=%d T=%d", B, T);
  if (B == 0) return ST_OK;
  ST_CHECK_ARG(new_ids && acc_ids && n_changes && changed && active, "NULL buffer");
  if (T <= 64 && !g_accept_thread_form)
    hipLaunchKernelGGL(accept_wave_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, new_ids, acc_ids, T, B,
                       vocab, unk, eos, pad, canon, max_change, n_changes, changed, active, rec_ids, confirm_ids);
  else
    hipLaunchKernelGGL(accept_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, new_ids, acc_ids, T, B,
                       vocab, unk, eos, pad, canon, max_change, n_changes, changed, active, rec_ids, confirm_ids);
  ST_LAUNCH_CHECK();
  return ST_OK;
}
