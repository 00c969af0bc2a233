/*
 * st_kernels.h — C-ABI of libst_kernels.so: the MI355X (gfx950) kernels under the gradient-guided
 * `--attack` loop of Facico/style-transfer.
 *
 * The reference has no FFI of its own for this path (it is in-process PyTorch); each
 * entry point below names the reference Python it replaces (file:line under
 * /root/reference/method/yelp/).  All pointers are DEVICE pointers unless the parameter is named
 * *_host or is a weights struct (host struct holding device pointers).  Every function returns 0 on
 * success and a negative ST_E* code on failure (st_last_error() gives the text); nothing throws
 * across the boundary; the caller owns every buffer; `stream` is a hipStream_t (0 = default
 * stream).  Kernels are launched asynchronously on `stream`; st_decode_greedy synchronises it (to
 * read back the live-sentence count, see below), and in the f16x3 format st_decode_greedy and
 * st_encode_latent synchronise once more at their end to read the f16-range flag (ST_ERANGE).
 *
 * Arithmetic type everywhere: IEEE f32 (the reference's dtype) in memory and in every non-matrix
 * operation.  Matrix products against packed weights run in one of two operand formats, chosen when
 * the weights are packed (ST_GEMM_*, below): exact f32 on v_mfma_f32_32x32x2_f32 (bit-identical to an
 * fmaf chain), or the 3-product f16 split on v_mfma_f32_32x32x16_f16 (f32 accumulate, ~2^-22 relative
 * error per term).  The style scorer and the in-batch loss always use the exact form.  Token ids are int32.
 */
#ifndef ST_KERNELS_H
#define ST_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_OK 0
#define ST_EINVAL (-1)   /* bad argument / unsupported shape          */
#define ST_EHIP (-2)     /* a HIP runtime call or kernel launch failed */
#define ST_ENOMEM (-3)   /* workspace too small                        */
#define ST_ERANGE (-4)   /* f16x3 format: an activation left the f16 range (|x| >= 65504 or NaN); the
                            outputs of that call are invalid — repack with ST_GEMM_F32             */

#define ST_D 256          /* latent/model width the kernels are specialised for (main.py:70,73) */
#define ST_MAX_LAYERS 4
#define ST_MAX_REFS 16    /* max like / unlike sentences per origin (reference data: 10 + 10)   */

/* library identification / error text (no reference counterpart: ABI plumbing) */
int st_abi_version(void);
const char* st_last_error(void);

/* Timing of one kernel of st_decode_greedy with HIP events on the caller's stream, for bench.py's
 * roofline line (no reference counterpart).  While enabled, every launch of the selected kernel is
 * bracketed with an event pair; st_profile_read waits for them and returns the summed kernel
 * milliseconds, the summed algorithmic flops and the launch count; reset != 0 clears the record.
 *   ST_PROF_GENERATOR  : generator GEMM + argmax (both passes when screened); 2 * live_rows * 256 * V flops
 *   ST_PROF_HALF_LAYER : the fused post-attention half-layer (f16x3 format) of st_decode_greedy AND
 *                        st_encode_latent; 2 * rows * 256 * (256 + 2 d_ff)
 * (live_rows is exact when sync_every == 1).                                                      */
#define ST_PROF_OFF 0
#define ST_PROF_GENERATOR 1
#define ST_PROF_HALF_LAYER 2
int st_profile_enable(int which);
int st_profile_read(double* ms, double* flops, long long* launches, int reset);
/* Executed work on the current GPU since the last reset, counted on the device from the exact live counts (no
 * reference counterpart; bench.py's whole-loop roofline): out4_host[0] = decode token-steps (live rows summed over
 * the decode steps of every st_decode_greedy call), [1] = keys attended (sum of prefix lengths over those
 * token-steps), [2] = decode jobs started, [3] reserved.  Synchronises the device.                                            */
int st_work_counters_read(unsigned long long* out4_host, int reset);

/* Weight packing (no reference counterpart).  The GEMM kernels read nn.Linear weight matrices
 * W [N,K] row-major in MFMA-fragment order so that every wave load is one coalesced 1 KiB read.
 *   ST_GEMM_F32   : packed[((blk*(K/8) + kg)*64 + lane)*4 + j] = W[blk*32 + (lane&31)][kg*8 + 4*(lane>>5) + j]
 *   ST_GEMM_F16X3 : W = hi + lo with hi = f16(W), lo = f16(W - hi); 16-byte fragments of 8 halves,
 *                   fragment[((blk*(K/16) + step)*2 + part)*64 + lane][j] = part(W[blk*32 + (lane&31)][16*step + 8*(lane>>5) + j]),
 *                   part 0 = hi, 1 = lo.  Requires |W| < 65504 (f16 range); so must the activations it
 *                   multiplies (LayerNorm outputs, attention outputs, ReLU hidden units).
 * (rows >= N are zero).  Both images take st_packed_weight_floats(N, K) floats.  Fields documented as
 * "packed" below must point at such an image; a weights struct's `gemm_format` says which one ALL of
 * its packed fields use.  K % 16 == 0.                                                           */
#define ST_GEMM_F32 0
#define ST_GEMM_F16X3 1
size_t st_packed_weight_floats(int N, int K);
int st_pack_weight(const float* W, int N, int K, int gemm_format, float* out_packed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (1) Style scorer: SiameseNet + Predictor + contrastive score + backward-to-latent + Adam.
 * ------------------------------------------------------------------------------------------- */

/* SiameseNet (siamese_contrastive.py:29-63) and Predictor (:66-86) in eval mode.  BatchNorm1d
 * (eval, running stats) is folded into the preceding Linear by the host:  W' = s*W, b' = s*b + t
 * with s = gamma/sqrt(var+eps), t = beta - mean*s.   Each matrix is given as [out,in] (`w`) AND as its
 * transpose [in,out] (`wt`, used by the backward pass), both PACKED with st_pack_weight.  The Predictor hidden width (100) is
 * zero-padded to 128 by the host.                                                              */
typedef struct StScorerWeights {
  const float *s1_w, *s1_wt, *s1_b;   /* 256 -> 512 (+BN1 folded), ReLU          */
  const float *s2_w, *s2_wt, *s2_b;   /* 512 -> 512 (+BN2 folded), ReLU          */
  const float *s3_w, *s3_wt, *s3_b;   /* 512 -> 256                              */
  const float *p1_w, *p1_wt, *p1_b;   /* 256 -> 128 (100 padded; +BN folded), LeakyReLU(0.2) */
  const float *p2_w, *p2_wt, *p2_b;   /* 128 -> 256                              */
  int32_t gemm_format;                /* ST_GEMM_F32: exact f32 MFMA.  ST_GEMM_F16X3: every `w` / `wt` image is packed in the
                                         f16x3 format from the matrix MULTIPLIED by a power of two (one per layer, chosen so
                                         that max |W| lies in [2^13, 2^14): BatchNorm-folded weights are small, and an f16
                                         lo part needs magnitude); the kernels multiply the products by unscale[]        */
  float unscale[5];                   /* f16x3: 1 / (that power of two) for s1, s2, s3, p1, p2 (exact)                  */
} StScorerWeights;

/* z = SiameseNet(x) for n rows (replaces batch_forward(1, siamese_net, like/unlike_latent),
 * siamese_contrastive.py:381-382, which the reference recomputes every iteration although it is
 * loop-invariant).  If unit != 0 each row is divided by max(||z||, 1e-8) (F.cosine_similarity's
 * normalisation) so the edit step can use dot products.                                          */
int st_scorer_embed(const StScorerWeights* w, const float* x, float* z, int n, int unit, void* stream);

/* The f16x3 scorer kernels raise a sticky per-process flag on the device when a value leaves the f16 range (|x| >= 65504
 * or NaN: the split products would then return garbage).  Reading it synchronises the device, so it is not done per step:
 * call this once after a batch of steps; out_of_range != 0 means the results since the last reset are invalid (use
 * ST_GEMM_F32).  No counterpart in the reference.                                                                        */
int st_scorer_status(int* out_of_range, int reset);

/* flags for st_scorer_edit_step */
#define ST_F_NOT_NLL 1        /* --not_nll       (siamese_contrastive.py:424-436) */
#define ST_F_SIMPLEST 2       /* --simplest_siamese (:384-400): no Predictor       */
#define ST_F_JUST_USE_ONE 4   /* --just_use_one  (:449-452)                        */
/* --victim_optimizer (siamese_contrastive.py:780: getattr(optim, name)([ori_latent], lr=lr), torch defaults otherwise),
 * bits 8..11 of flags; adam_m / adam_v are the rule's two state arrays (both zero at the start).                      */
#define ST_F_OPT_SHIFT 8
#define ST_F_OPT_MASK (15 << ST_F_OPT_SHIFT)
#define ST_OPT_ADAM 0      /* betas (0.9, 0.999), eps 1e-8                                         */
#define ST_OPT_SGD 1       /* x -= lr g                                                            */
#define ST_OPT_ADAMW 2     /* Adam with decoupled weight decay 0.01                                */
#define ST_OPT_ADAMAX 3    /* m, u = max(0.999 u, |g| + 1e-8); x -= lr / (1 - 0.9^t) m / u          */
#define ST_OPT_RMSPROP 4   /* v = 0.99 v + 0.01 g^2; x -= lr g / (sqrt(v) + 1e-8)                   */
#define ST_OPT_ADAGRAD 5   /* s += g^2; x -= lr g / (sqrt(s) + 1e-10)                               */
#define ST_OPT_NADAM 6     /* Adam with Nesterov momentum, momentum_decay 4e-3                     */
#define ST_OPT_RADAM 7     /* rectified Adam (plain bias-corrected momentum while rho_t <= 5)      */
#define ST_OPT_ADADELTA 8  /* rho 0.9, eps 1e-6; adam_v = square_avg, adam_m = acc_delta           */
#define ST_OPT_ASGD 9      /* lambd 1e-4, alpha 0.75, t0 1e6 (the averaged iterate is not kept)    */
#define ST_OPT_RPROP 10    /* etas (0.5, 1.2), step sizes [1e-6, 50]; adam_m = previous gradient, adam_v = step size */

/* One edit iteration for B sentences (replaces get_contrastive_loss + `loss=-(loss); zero_grad();
 * loss.backward(); ori_optimizer.step()`, siamese_contrastive.py:796-810, with torch.optim.Adam
 * defaults beta=(0.9,0.999), eps=1e-8, built per sentence at :780-781).
 *   latent, adam_m, adam_v : [B,256] in/out      like_z / unlike_z : [B,n_pos,256] / [B,n_neg,256]
 *   unit vectors from st_scorer_embed(unit=1)     lr : [B] (pos_victim_lr / neg_victim_lr, :740)
 *   step : [B] Adam step count BEFORE this call (incremented here)   active : [B] 0 = skip row
 *   loss_out : [B] the value get_contrastive_loss returns (before the sign flip), may be NULL
 *   grad_out : [B,256] d(-loss)/d latent, may be NULL (parity tests)
 *   scores_out : [B, n_pos + n_neg] the pos_score / neg_score values (cosine / tau, :410-418), may be NULL;
 *   --dance prints a digest of them (:833-835)                                                     */
int st_scorer_edit_step(const StScorerWeights* w, float* latent, float* adam_m, float* adam_v,
                        const float* like_z, const float* unlike_z, int n_pos, int n_neg,
                        const float* lr, int* step, const uint8_t* active, int B, float tau,
                        int flags, float* loss_out, float* grad_out, float* scores_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (2) Greedy decode with the Transformer autoencoder's decoder (KV-cached).
 * ------------------------------------------------------------------------------------------- */
typedef struct StDecLayer {
  const float *ln0_a, *ln0_b;          /* sublayer[0].norm  (model.py:97-108 LayerNorm)         */
  const float *wqkv, *bqkv;            /* self_attn.linears[0..2] stacked: [768,256], [768]     */
  const float *wo, *bo;                /* self_attn.linears[3]                                  */
  const float *xv_w, *xv_b;            /* src_attn.linears[2] (value projection)                */
  const float *xo_w, *xo_b;            /* src_attn.linears[3]; q/k projections and sublayer[1].norm
                                          cannot influence the output: memory has ONE position so
                                          softmax == 1 (model.py:264-269)         */
  const float *ln2_a, *ln2_b;          /* sublayer[2].norm                                      */
  const float *w1, *b1, *w2, *b2;      /* feed_forward.w_1 [d_ff,256], w_2 [256,d_ff]           */
} StDecLayer;

typedef struct StDecoderWeights {
  int32_t d_model, n_heads, d_ff, n_layers, vocab, pe_rows;
  int32_t gemm_format;                 /* ST_GEMM_*: format of every PACKED field below          */
  float gen_w_maxnorm;                 /* max_v ||generator.proj.weight[v]||_2 (host-computed); > 0 enables the
                                          screened argmax of the f16x3 format, 0 = always the full product */
  const float* emb;                    /* tgt_embed.0.lut.weight [V,256]                        */
  const float* pe;                     /* tgt_embed.1.pe [pe_rows,256]                          */
  const float *lnf_a, *lnf_b;          /* decoder.norm                                          */
  const float *gen_w, *gen_b;          /* generator.proj [V,256] PACKED (st_pack_weight), [V]   */
  const float* gen_w_norm;             /* [V] ||generator.proj.weight[v]||_2, each rounded UP, or NULL: with it the screened
                                          argmax bounds every column by its own norm instead of gen_w_maxnorm           */
  StDecLayer layer[ST_MAX_LAYERS];
  const float* l0_table;               /* NULL, or the table st_decode_build_l0_table filled: [vocab][l0_table_len][768] f32.
                                          The first decoder layer's self-attention input is LayerNorm(emb[token] * 16 + pe[pos]):
                                          its q | k | v depend on (token, position) alone, not on the sentence, so they are
                                          computed once per checkpoint; decode steps then skip that layer's QKV projection and
                                          its K/V cache and read the rows of the tokens already emitted (f16x3 format)        */
  int32_t l0_table_len;                /* positions per token in l0_table (>= max_len - 1 of every decode call)            */
  int32_t reserved1;
} StDecoderWeights;

/* Bytes of the first-layer q | k | v table for `positions` decode positions, and its construction: rows = vocab * positions
 * through the layer's own LayerNorm + stacked projection kernel (the values a decode step would compute, bit for bit).
 * scratch: vocab * positions * 1024 bytes (the embedded inputs).  No counterpart in the reference (model.py:191-196 recomputes
 * them for every sentence and every step).                                                                                */
size_t st_decode_l0_table_bytes(const StDecoderWeights* w, int positions);
int st_decode_build_l0_table(const StDecoderWeights* w, int positions, float* table, void* scratch, size_t scratch_bytes,
                             void* stream);

/* bytes of device workspace st_decode_greedy needs for B sentences */
size_t st_decode_workspace_bytes(const StDecoderWeights* w, int B, int max_len);
/* the same for the `options` the call will pass to st_decode_greedy (ST_DEC_*): when the first layer reads l0_table it keeps no
 * K / V cache, and the workspace is about half the size st_decode_workspace_bytes (which covers every option) returns */
size_t st_decode_workspace_bytes_opt(const StDecoderWeights* w, int B, int max_len, int options);

/* options for st_decode_greedy */
#define ST_DEC_EARLY_EXIT 1   /* stop a sentence at its first EOS/PAD token (output-equivalent:
                                 data.py:43-52 truncates there); off = the reference's fixed
                                 max_len-1 steps                                                 */
#define ST_DEC_NO_L0_TABLE 4  /* ignore StDecoderWeights.l0_table: project the first layer's q | k | v per step (tests, A/B) */
#define ST_DEC_NO_SCREEN 2    /* f16x3 format: always run the 3-product generator (no one-product
                                 screening pass; see gen_w_maxnorm)                              */

/* ys = EncoderDecoder.greedy_decode(latent, max_len, start_id) (model.py:271-299) for B sentences.
 *   latent [B,256]   out_ids [B,max_len-1] int32 (positions after an early exit hold 0 = PAD)
 *   top2gap [B,max_len-1] f32 or NULL: best minus second-best logit at each emitted token (the
 *   record that lets a divergence be traced to an argmax near-tie)
 *   n_steps_host (host int*, may be NULL): number of decode steps actually run.
 * Synchronises `stream` once per `sync_every` steps to read the live count (>=1).
 * With gen_w_maxnorm > 0 in the f16x3 format and top2gap == NULL, steps with >= 2048 live rows use
 * the screened argmax (one-product pass, then the full product on the undecided rows): same ids.  */
int st_decode_greedy(const StDecoderWeights* w, const float* latent, int B, int max_len, int bos,
                     int eos, int pad, int options, int sync_every, int32_t* out_ids,
                     float* top2gap, void* workspace, size_t workspace_bytes, int* n_steps_host,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * (3) Generator GEMM + argmax: ids[r] = argmax_v( LN(h[r]) . W[v] + b[v] )  (model.py:200-207 +
 * torch.max(prob, dim=1), :290-292; log_softmax is argmax-invariant and skipped; ties -> lowest
 * index).  W is PACKED (st_pack_weight) in `gemm_format`.  ln_a/ln_b may be NULL.
 * scratch: n*ST_ARGMAX_SPLITS*16 bytes. */
#define ST_ARGMAX_SPLITS 4
int st_gemm_argmax(const float* h, const float* ln_a, const float* ln_b, const float* W,
                   const float* b, int n, int V, int gemm_format, int32_t* ids, float* best, float* gap,
                   void* scratch, void* stream);

/* (3b) st_gemm_argmax for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that may not call hipMalloc or hand over a workspace — code under HIP graph capture, tight serving
 * loops, harnesses that forbid allocation.  Same arguments and results as st_gemm_argmax minus `scratch`: the
 * per-split partials go to a static device array inside the library (ST_NOALLOC_MAX_ROWS * ST_ARGMAX_SPLITS * 16
 * bytes = 4 MiB per GPU the library runs on, held until the process ends).  n > ST_NOALLOC_MAX_ROWS returns
 * ST_EINVAL.  The call allocates nothing (no hipMalloc / hipFree / allocating library routine); its first use on
 * a device resolves that device's copy of the array with hipGetSymbolAddress.  All calls on one GPU share the
 * array: calls that overlap on one GPU (two streams, two threads) race — serialise them.  Nothing in the array
 * needs clearing between calls (every partial that is read was written by the same call).                  */
#define ST_NOALLOC_MAX_ROWS 65536
int st_gemm_argmax_noalloc(const float* h, const float* ln_a, const float* ln_b, const float* W,
                           const float* b, int n, int V, int gemm_format, int32_t* ids, float* best, float* gap,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * (4) Text-change detection (siamese_contrastive.py:816-839 + data.py:43-52), on ids.
 * For each active sentence: the decoded row `new_ids` and the accepted row `acc_ids` are compared
 * as texts (prefix up to the first EOS/PAD, ids >= vocab -> UNK, ids mapped through `canon`
 * (id -> first id with the same word) when canon != NULL).  If different: acc_ids <- new_ids,
 * changed[s] = 1, n_changes[s] += 1, and active[s] <- 0 once n_changes[s] >= max_change.
 * rec_ids (may be NULL) [B,T] receives the accepted row after this iteration.
 * confirm_ids (may be NULL; --retext, siamese_contrastive.py:819-825): decode of the RE-ENCODED new text;
 * when given, a change is accepted only if confirm_ids spells the same text as new_ids.          */
int st_accept_changes(const int32_t* new_ids, int32_t* acc_ids, int T, int B, int vocab, int unk,
                      int eos, int pad, const int32_t* canon, int max_change, int32_t* n_changes,
                      uint8_t* changed, uint8_t* active, int32_t* rec_ids, const int32_t* confirm_ids,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * (5) In-batch InfoNCE similarity loss (train_ae_model_constraint, siamese_contrastive.py:1060-1069):
 *   sim = cos(ori[i], like[j]) / tau  for i in [0,Bq), j in [0,Bk);  loss[i] = -log_softmax(sim[i])[row_offset+i]
 * The [Bq,Bk] matrix is never written to memory (online softmax).  lse[i] = logsumexp(sim[i]) is
 * kept for the backward.  bwd: given g[i] = dL/dloss[i], accumulates d_ori [Bq,256], d_like [Bk,256]
 * (d_like is ADDED to: caller zeroes it; rank-partial sums are reduce-scattered by the host).   */
int st_simloss_fwd(const float* ori, const float* like, int Bq, int Bk, float tau, int row_offset,
                   float* loss, float* lse, void* scratch, size_t scratch_bytes, void* stream);
int st_simloss_bwd(const float* ori, const float* like, int Bq, int Bk, float tau, int row_offset,
                   const float* lse, const float* g, float* d_ori, float* d_like, void* scratch,
                   size_t scratch_bytes, void* stream);
size_t st_simloss_scratch_bytes(int Bq, int Bk);
/* Operand format of the similarity products (ST_GEMM_*; returns the previous one; unknown values change nothing).
 * ST_GEMM_F16X3 (default): unit rows and dS' = exp(s - lse) - [target] lie in [-1, 1], so every product of the block
 * (forward, and in the backward the dS', dS'^T, d_ori and d_like products when Bq and Bk are multiples of 256) runs as
 * three f16 MFMA products of hi + lo split operands (f32-grade).  ST_GEMM_F32: exact f32 MFMA throughout.           */
int st_simloss_format(int gemm_format);

/* ---------------------------------------------------------------------------------------------
 * (6) Encoder -> latent, batched and length-packed (EncoderDecoder.forward's encoder half /
 * get_latent, model.py:229-240, :253-257; Encoder :137-163).  latent[s] = sum over the sentence's
 * non-pad tokens of sigmoid(encoder(x)).  The reference's unused teacher-forced decode inside
 * forward() (model.py:248-249) is not computed.
 *   tokens/tok_pos [n_rows]: packed non-pad token ids and their position in the sentence
 *   sent_off [n_sent+1]: first row of each sentence (sent_off[n_sent] = n_rows); max_sent_len <= 64 */
typedef struct StEncLayer {
  const float *ln0_a, *ln0_b, *wqkv, *bqkv, *wo, *bo, *ln1_a, *ln1_b, *w1, *b1, *w2, *b2;
} StEncLayer;
typedef struct StEncoderWeights {
  int32_t d_model, n_heads, d_ff, n_layers, vocab, pe_rows;
  int32_t gemm_format;                 /* ST_GEMM_*: format of wqkv / wo / w1 / w2 (all PACKED)  */
  int32_t l0_table_len;                /* positions per token in l0_table                        */
  const float* emb;                    /* src_embed.0.lut.weight [V,256] (shared with tgt)      */
  const float* pe;                     /* src_embed.1.pe                                        */
  const float *lnf_a, *lnf_b;          /* encoder.norm                                          */
  StEncLayer layer[ST_MAX_LAYERS];
  const float* l0_table;               /* NULL, or st_encode_build_l0_table's [vocab][l0_table_len][768]: q | k | v of the first
                                          encoder layer per (token, position) - its input LayerNorm(emb[token] * 16 + pe[pos])
                                          does not depend on the sentence.  With it (f16x3 format, every sentence no longer than
                                          l0_table_len) the first layer runs no projection: identical latents                 */
} StEncoderWeights;
/* the encoder's first-layer table: see StDecoderWeights.l0_table / st_decode_build_l0_table (same construction, scratch =
 * vocab * positions * 1024 bytes).  No counterpart in the reference.                                                        */
size_t st_encode_l0_table_bytes(const StEncoderWeights* w, int positions);
int st_encode_build_l0_table(const StEncoderWeights* w, int positions, float* table, void* scratch, size_t scratch_bytes,
                             void* stream);
size_t st_encode_workspace_bytes(const StEncoderWeights* w, int n_rows, int n_sent);
int st_encode_latent(const StEncoderWeights* w, const int32_t* tokens, const int32_t* tok_pos,
                     const int32_t* sent_off, int n_rows, int n_sent, int max_sent_len, float* latent,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (7) Legacy FGIM edit (fgim_attack, model.py:454-516, with Classifier, model.py:414-433; SURVEY §8f
 * rank 4 — the reference's call sites, main.py:481/:511, are commented out).  For each sentence:
 *   o = sigmoid(fc3(lrelu(fc2(lrelu(fc1(x))))));  g = d BCELoss(o, target) / d x  (one sentence per loss,
 *   as in the reference's batch of 1);  out[k][b] = x[b] - eps[k] * g[b]  for the n_eps step sizes
 * (every epsilon of fgim_attack's list takes exactly one step from the ORIGINAL latent: `it` starts at 4
 * and the loop ends at it >= 5).  The caller decodes out[] with st_decode_greedy.
 *   weights: plain row-major nn.Linear tensors (fc1 [100,256], fc2 [50,100], fc3 [1,50]) and biases
 *   latent [B,256]  target [B] (0/1)  eps [n_eps] (device)  pred [B] or NULL  grad [B,256] or NULL
 *   out [n_eps,B,256]                                                                              */
typedef struct StClassifierWeights {
  const float *w1, *b1, *w2, *b2, *w3, *b3;
} StClassifierWeights;
int st_fgim_perturb(const StClassifierWeights* w, const float* latent, const float* target, const float* eps,
                    int n_eps, int B, float* pred, float* grad, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (8) One optimisation step of the style scorer: the loop body of train_contrastive
 * (siamese_contrastive.py:918-931) — SiameseNet (:29-62) and Predictor (:66-86) in TRAIN mode
 * (BatchNorm batch statistics per call, Dropout 0.2), get_contrastive_loss in its NLL form (:367-453), gradients of
 * every parameter, BatchNorm running statistics, and torch.optim.Adam (:881-883, :929-931) as a second call so a
 * data-parallel caller can all-reduce `grads` in between.
 *
 *   params / grads / adam_m / adam_v   flat f32 [ST_TP_COUNT], tensors in the order of
 *        list(siamese_net.parameters()) + list(predictor_net.parameters()), nn.Linear weights row-major [out,in]
 *   bn   flat f32 [ST_TB_COUNT]: running_mean / running_var of SiameseNet.norm1, .norm2, Predictor.norm1 (in/out)
 *   x    [1+Kp+Kn, B, 256] latents of the calls in the reference's order: ori, like_0.., unlike_0.. (:380-382)
 *   keep_masks  u8 [1+Kp+Kn][2][B][512] (1 = keep) for SiameseNet.d1 / .d2 of every call, or NULL: the keep decision
 *        is then a counter hash of (seed, element) — pass a fresh seed every step
 *   flags  ST_F_SIMPLEST (--simplest_siamese: no Predictor, like / unlike outputs NOT detached, :384-400),
 *          ST_F_JUST_USE_ONE (:449-452)
 *   loss_out [1] (device)   choice_out [Kp,B] i32 or NULL: argmax of each log-softmax row (0 = the positive won; the
 *        reference's accuracy counter, :926-933)
 * Requires B >= 2, 1 <= Kp, 1 <= Kn, Kp + Kn <= 64.  In ST_F_SIMPLEST mode the Predictor entries of grads are 0.     */
#define ST_TP_S_FC1_W 0
#define ST_TP_S_FC1_B 131072
#define ST_TP_S_BN1_G 131584
#define ST_TP_S_BN1_B 132096
#define ST_TP_S_FC2_W 132608
#define ST_TP_S_FC2_B 394752
#define ST_TP_S_BN2_G 395264
#define ST_TP_S_BN2_B 395776
#define ST_TP_S_FC3_W 396288
#define ST_TP_S_FC3_B 527360
#define ST_TP_SIAMESE_COUNT 527616   /* the optimizer of --simplest_siamese covers only these (:877-879) */
#define ST_TP_P_FC1_W 527616
#define ST_TP_P_FC1_B 553216
#define ST_TP_P_BN_G 553316
#define ST_TP_P_BN_B 553416
#define ST_TP_P_FC2_W 553516
#define ST_TP_P_FC2_B 579116
#define ST_TP_COUNT 579372
#define ST_TB_S_RM1 0
#define ST_TB_S_RV1 512
#define ST_TB_S_RM2 1024
#define ST_TB_S_RV2 1536
#define ST_TB_P_RM 2048
#define ST_TB_P_RV 2148
#define ST_TB_COUNT 2248
int st_scorer_train_workspace_bytes(int B, int Kp, int Kn, int flags, size_t* bytes);
int st_scorer_train_step(const float* params, float* bn, const float* x, int B, int Kp, int Kn, float tau, int flags,
                         float drop_p, const uint8_t* keep_masks, uint64_t seed, float* grads, float* loss_out,
                         int32_t* choice_out, void* workspace, size_t workspace_bytes, void* stream);
/* torch.optim.Adam with default options (no weight decay, no amsgrad) over n elements; step is 1-based.  */
int st_adam_step(float* params, const float* grads, float* adam_m, float* adam_v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (8b) One optimisation step of the AUTOENCODER under the in-batch constraint: loss and gradients of the loop body of
 * train_ae_model_constraint (siamese_contrastive.py:1028-1072; SURVEY §8f rank 3, second half) — EncoderDecoder.forward
 * in TRAIN mode (model.py:229-250), LabelSmoothing / ntokens (model.py:389-411, :1040-1041), the B x B similarity loss
 * over the `like` views plus a second pass of the origins (:1049-1071), (bata * constraint + rec).backward() (:1072).
 * The optimizer (NoamOpt + Adam(betas 0.9 / 0.98, eps 1e-9), :1009-1010, model.py:355-381) is st_adam_step on the flat
 * vectors with lr = the Noam rate, so a data-parallel caller can all-reduce `grads_flat` in between.
 *   p / g        parameters and gradients: host structs of DEVICE pointers, plain row-major nn.Linear tensors; g points
 *                into grads_flat [n_params] (zeroed here; tensors that receive no gradient — the source attention's
 *                query / key projections and its LayerNorm, whose softmax runs over ONE memory position — stay zero)
 *   encoder input, length-packed: (1 + views) * B sentences in the order  origins | view 0 (B sentences) | ... |
 *                origins again (the reference's ori_latent_cp pass); enc_tokens / enc_pos [n_enc_rows], enc_off [n_enc_sent + 1]
 *   decoder input, length-packed teacher forcing of the B origins: rows BOS, t0, .. (dec_tokens), targets t0, .., EOS
 *                (dec_targets), positions dec_pos, dec_off [B + 1]; sentences <= 64 rows
 *   drop_p, seed  Dropout probability (reference: 0.1) and the seed of the counter-hash keep decisions
 *   losses_out [2] (device): reconstruction loss (sum / ntokens), constraint loss (before * bata)
 * Requires B % 32 == 0 (similarity block backward).                                                              */
typedef struct StAeLayer {
  const float *ln0_a, *ln0_b;                          /* sublayer[0].norm                                  */
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;  /* self_attn.linears[0..3]                           */
  const float *xv_w, *xv_b, *xo_w, *xo_b;              /* decoder: src_attn.linears[2], [3]                 */
  const float *ln2_a, *ln2_b;                          /* norm before the feed-forward (encoder: sublayer[1], decoder: sublayer[2]) */
  const float *w1, *b1, *w2, *b2;                      /* feed_forward                                      */
} StAeLayer;
typedef struct StAeParams {
  int32_t d_model, n_heads, d_ff, n_layers, vocab, pe_rows;
  const float *emb, *pe;                               /* shared src / tgt embedding [V,256]; positional table */
  const float *enc_norm_a, *enc_norm_b, *dec_norm_a, *dec_norm_b;
  const float *gen_w, *gen_b;                          /* generator.proj [V,256], [V]                        */
  StAeLayer enc[ST_MAX_LAYERS], dec[ST_MAX_LAYERS];
} StAeParams;
size_t st_ae_train_workspace_bytes(const StAeParams* p, int n_enc_rows, int n_enc_sent, int n_dec_rows, int B);
int st_ae_train_step(const StAeParams* p, const StAeParams* g, float* grads_flat, int64_t n_params,
                     const int32_t* enc_tokens, const int32_t* enc_pos, const int32_t* enc_off, int n_enc_rows,
                     int n_enc_sent, const int32_t* dec_tokens, const int32_t* dec_pos, const int32_t* dec_targets,
                     const int32_t* dec_off, int n_dec_rows, int B, float bata, float tau, float smoothing, float drop_p,
                     uint64_t seed, float* losses_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (9) HOST function (no GPU work, all pointers are host pointers): the per-sentence block of the log that
 * attack_contrastive / contrastive_attack_one print (siamese_contrastive.py:702-732, :789, :825-836) with
 * id2text_sentence (data.py:43-52), from the compact per-sentence records:
 *   vocab_bytes / vocab_off [vocab+1]: the words of word_to_id.txt, concatenated (UTF-8)
 *   prefix: what precedes every record in the log file ("INFO:siamese_contrastive.py:print_and_log:1592 - ")
 *   src_tokens / src_off [S+1]: the origin sentences ('Src:' line)     init_ids [S,T]: 'Rec:' line
 *   like_ids [S,kp,T] / unlike_ids [S,kn,T] or NULL: 'Sim{j}:' / 'Ref{j}:' lines
 *   n_iter [S], changed [S,I] (1 = a 'Change.k' line precedes 'iteration.it'), rows [sum(changed),T]: the accepted
 *   ids after each change, sentence-major; the text of 'iteration.it' is the last accepted one (init_ids before
 *   the first change).   first_index: global index of sentence 0 ('Sentence Pair.{i}').
 * Returns the number of bytes the block takes; it is written to `out` only if that is <= out_cap (call with
 * out = NULL to size the buffer); -1 on a bad argument.  --dance blocks are formatted by the Python writer. */
long long st_format_attack_log(const char* vocab_bytes, const int64_t* vocab_off, int vocab, int eos, int pad, int unk,
                               const char* prefix, int S, int T, int I, long long first_index,
                               const int32_t* src_tokens, const int64_t* src_off, const int32_t* init_ids,
                               const int32_t* like_ids, int kp, const int32_t* unlike_ids, int kn,
                               const int32_t* n_iter, const uint8_t* changed, const int32_t* rows, char* out,
                               long long out_cap);

#ifdef __cplusplus
}
#endif
#endif /* ST_KERNELS_H */
