// Host side of the output boundary: the per-sentence block of the reference's log, formatted from the
// compact per-sentence records at memory speed (no GPU code in this file).
//
// Replaces the print_and_log calls of attack_contrastive / contrastive_attack_one
// (/root/reference/method/yelp/siamese_contrastive.py:702-732, :789, :825-836) together with
// id2text_sentence (data.py:43-52): ids -> words up to the first EOS / PAD, ids >= vocab -> UNK, ' '.join.
// At 10^5 sentences/s the log is ~200 MB/s of text; a Python loop over sentences formats ~2 k sentences/s.
#include <stdint.h>
#include <string.h>

#include "../../include/st_kernels.h"

namespace {
struct Out {
  char* p;
  long long cap, len;
  void put(const char* s, long long n) {
    if (len + n <= cap) memcpy(p + len, s, (size_t)n);
    len += n;  // keeps counting past the capacity: the caller learns the size it needs
  }
  void lit(const char* s) { put(s, (long long)strlen(s)); }
  void num(long long v) {
    char b[24];
    int n = 0;
    if (v == 0) b[n++] = '0';
    bool neg = v < 0;
    unsigned long long u = neg ? (unsigned long long)(-v) : (unsigned long long)v;
    while (u) {
      b[n++] = (char)('0' + u % 10);
      u /= 10;
    }
    if (neg) b[n++] = '-';
    for (int i = 0; i < n / 2; ++i) {
      char t = b[i];
      b[i] = b[n - 1 - i];
      b[n - 1 - i] = t;
    }
    put(b, n);
  }
};

struct Vocab {
  const char* bytes;
  const int64_t* off;
  int n, eos, pad, unk;
};

// id2text_sentence: words up to the first EOS / PAD, joined by single spaces
void text(Out& o, const Vocab& v, const int32_t* ids, int n_ids) {
  bool first = true;
  for (int i = 0; i < n_ids; ++i) {
    int t = ids[i];
    if (t == v.eos || t == v.pad) break;
    if (t >= v.n || t < 0) t = v.unk;
    if (!first) o.put(" ", 1);
    first = false;
    o.put(v.bytes + v.off[t], v.off[t + 1] - v.off[t]);
  }
}
}  // namespace

extern "C" long long st_format_attack_log(const char* vocab_bytes, const int64_t* vocab_off, int vocab, int eos, int pad,
                                           int unk, const char* prefix, int S, int T, int I, long long first_index,
                                           const int32_t* src_tokens, const int64_t* src_off, const int32_t* init_ids,
                                           const int32_t* like_ids, int kp, const int32_t* unlike_ids, int kn,
                                           const int32_t* n_iter, const uint8_t* changed, const int32_t* rows,
                                           char* out, long long out_cap) {
  if (!vocab_bytes || !vocab_off || vocab < 1 || S < 0 || T < 1 || I < 0 || !prefix) return -1;
  if (S > 0 && (!src_tokens || !src_off || !init_ids || !n_iter || !changed)) return -1;
  if (!out) out_cap = 0;
  Out o = {out, out_cap, 0};
  const Vocab v = {vocab_bytes, vocab_off, vocab, eos, pad, unk};
  const long long plen = (long long)strlen(prefix);
  static const char stars[] = "********************";
  static const char dashes[] = "----------------------------------------";
  long long row = 0;  // next changed row
  for (int s = 0; s < S; ++s) {
    o.put(prefix, plen); o.lit(stars); o.lit("Sentence Pair."); o.num(first_index + s); o.lit(stars); o.put("\n", 1);
    o.put(prefix, plen); o.lit("Src: ");
    text(o, v, src_tokens + src_off[s], (int)(src_off[s + 1] - src_off[s]));
    o.put("\n", 1);
    o.put(prefix, plen); o.lit("Rec: ");
    text(o, v, init_ids + (size_t)s * T, T);
    o.put("\n", 1);
    if (like_ids)
      for (int j = 0; j < kp; ++j) {
        o.put(prefix, plen); o.lit("Sim"); o.num(j); o.lit(": ");
        text(o, v, like_ids + ((size_t)s * kp + j) * T, T);
        o.put("\n", 1);
      }
    if (unlike_ids)
      for (int j = 0; j < kn; ++j) {
        o.put(prefix, plen); o.lit("Ref"); o.num(j); o.lit(": ");
        text(o, v, unlike_ids + ((size_t)s * kn + j) * T, T);
        o.put("\n", 1);
      }
    o.put(prefix, plen); o.lit(dashes); o.put("\n", 1);
    // the accepted text is formatted once per change and copied for the following 'iteration.' lines
    const int32_t* cur = init_ids + (size_t)s * T;
    long long tbeg = -1, tlen = 0;  // span of the current accepted text inside `out` (valid while it fits)
    int k = 0;
    const int ni = n_iter[s] < I ? n_iter[s] : I;
    for (int it = 0; it < ni; ++it) {
      if (changed[(size_t)s * I + it]) {
        if (!rows) return -1;
        cur = rows + (size_t)row * T;
        ++row;
        o.put(prefix, plen); o.lit("Change."); o.num(k++); o.lit(", ");
        tbeg = o.len;
        text(o, v, cur, T);
        tlen = o.len - tbeg;
        o.put("\n", 1);
        o.put(prefix, plen); o.lit("iteration."); o.num(it); o.lit(", ");
        if (o.len + tlen <= o.cap) memmove(o.p + o.len, o.p + tbeg, (size_t)tlen);
        o.len += tlen;
        o.put("\n", 1);
      } else {
        o.put(prefix, plen); o.lit("iteration."); o.num(it); o.lit(", ");
        if (tbeg >= 0) {
          if (o.len + tlen <= o.cap) memmove(o.p + o.len, o.p + tbeg, (size_t)tlen);
          o.len += tlen;
        } else {
          tbeg = o.len;
          text(o, v, cur, T);
          tlen = o.len - tbeg;
        }
        o.put("\n", 1);
      }
    }
  }
  return o.len;
}
