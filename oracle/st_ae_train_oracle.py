"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, float64) of ONE optimisation step of the autoencoder's
constraint training: the loop body of train_ae_model_constraint
(/root/reference/method/yelp/siamese_contrastive.py:1028-1078) with

  EncoderDecoder.forward in TRAIN mode      model.py:229-250 (encoder :137-163, decoder :167-196, MHA :18-60,
                                            FFN :63-72, LayerNorm :97-108, SublayerConnection :111-123,
                                            Embeddings + PositionalEncoding :75-94,126-133, Generator :200-207)
  LabelSmoothing (KLDivLoss, sum) / ntokens  model.py:389-411, siamese_contrastive.py:1040-1041
  the B x B constraint over 11 views         :1049-1071  (10 `like` views + a second pass of the origins)
  (bata * constraint + rec).backward()       :1072
  NoamOpt(d_model, 1, 2000, Adam(lr=0, betas=(0.9, 0.98), eps=1e-9)).step()   :1009-1010, model.py:355-381

Gradients come from a minimal reverse-mode tape over numpy arrays (every primitive's backward is written out below);
the padded [B, L] batch layout and the ORDER of the nn.Dropout calls follow the reference, so the keep-masks that
oracle/make_ae_train_golden.py captured from the reference can be replayed call by call.

Pinned by tests/test_oracle_golden.py against tests/golden/ae_train_{nodrop,drop}.npz (the reference's own run).
Only tests/ may import this file.
"""
import math

import numpy as np

import st_oracle as O

F = np.float64


# ------------------------------------------------------------------------------------------------
# a minimal tape
# ------------------------------------------------------------------------------------------------
class T:
    __slots__ = ("v", "g", "parents", "back")

    def __init__(self, v, parents=(), back=None):
        self.v = np.asarray(v, F)
        self.g = None
        self.parents = parents
        self.back = back

    def acc(self, g):
        self.g = g.copy() if self.g is None else self.g + g


def backward(roots):
    """roots: [(tensor, seed gradient)].  ONE reverse sweep over the union graph (a second sweep would push the
    gradients already accumulated in shared nodes through their ancestors again)."""
    import sys
    sys.setrecursionlimit(100000)
    order, seen = [], set()

    def visit(t):
        if id(t) in seen:
            return
        seen.add(id(t))
        for p in t.parents:
            visit(p)
        order.append(t)
    for r, _ in roots:
        visit(r)
    for r, seed in roots:
        r.acc(np.asarray(seed, F) * np.ones_like(r.v))
    for t in reversed(order):
        if t.back is not None and t.g is not None:
            t.back(t.g)


def _unbroadcast(g, shape):
    while g.ndim > len(shape):
        g = g.sum(0)
    for i, s in enumerate(shape):
        if s == 1 and g.shape[i] != 1:
            g = g.sum(i, keepdims=True)
    return g


def add(a, b):
    out = T(a.v + b.v, (a, b))
    out.back = lambda g: (a.acc(_unbroadcast(g, a.v.shape)), b.acc(_unbroadcast(g, b.v.shape)))
    return out


def scale(a, c):
    out = T(a.v * c, (a,))
    out.back = lambda g: a.acc(g * c)
    return out


def mul_const(a, m):
    """elementwise product with a constant array (dropout masks, attention masks)"""
    out = T(a.v * m, (a,))
    out.back = lambda g: a.acc(_unbroadcast(g * m, a.v.shape))
    return out


def linear(x, W, b):
    """nn.Linear: y = x W^T + b"""
    out = T(x.v @ W.v.T + b.v, (x, W, b))

    def back(g):
        x2, g2 = x.v.reshape(-1, x.v.shape[-1]), g.reshape(-1, g.shape[-1])
        x.acc(g @ W.v)
        W.acc(g2.T @ x2)
        b.acc(g2.sum(0))
    out.back = back
    return out


def matmul(a, b):
    out = T(a.v @ b.v, (a, b))
    out.back = lambda g: (a.acc(_unbroadcast(g @ np.swapaxes(b.v, -1, -2), a.v.shape)),
                          b.acc(_unbroadcast(np.swapaxes(a.v, -1, -2) @ g, b.v.shape)))
    return out


def reshape(a, shape):
    out = T(a.v.reshape(shape), (a,))
    out.back = lambda g: a.acc(g.reshape(a.v.shape))
    return out


def transpose(a, i, j):
    out = T(np.swapaxes(a.v, i, j), (a,))
    out.back = lambda g: a.acc(np.swapaxes(g, i, j))
    return out


def relu(a):
    out = T(np.maximum(a.v, 0.0), (a,))
    out.back = lambda g: a.acc(g * (a.v > 0))
    return out


def sigmoid(a):
    s = 1.0 / (1.0 + np.exp(-a.v))
    out = T(s, (a,))
    out.back = lambda g: a.acc(g * s * (1.0 - s))
    return out


def masked_fill(a, keep, value):
    """scores.masked_fill(mask == 0, value)"""
    out = T(np.where(keep, a.v, value), (a,))
    out.back = lambda g: a.acc(_unbroadcast(g * keep, a.v.shape))
    return out


def softmax(a):
    e = np.exp(a.v - a.v.max(-1, keepdims=True))
    p = e / e.sum(-1, keepdims=True)
    out = T(p, (a,))
    out.back = lambda g: a.acc(p * (g - (g * p).sum(-1, keepdims=True)))
    return out


def log_softmax(a):
    m = a.v.max(-1, keepdims=True)
    lse = m + np.log(np.exp(a.v - m).sum(-1, keepdims=True))
    out = T(a.v - lse, (a,))
    out.back = lambda g: a.acc(g - np.exp(a.v - lse) * g.sum(-1, keepdims=True))
    return out


def layer_norm(x, a2, b2, eps=1e-6):
    """model.py:97-108: a_2 * (x - mean) / (std_unbiased + eps) + b_2"""
    n = x.v.shape[-1]
    mu = x.v.mean(-1, keepdims=True)
    xc = x.v - mu
    s = np.sqrt((xc * xc).sum(-1, keepdims=True) / (n - 1))
    d = s + eps
    out = T(a2.v * xc / d + b2.v, (x, a2, b2))

    def back(g):
        gy = g * a2.v
        a2.acc((g * xc / d).reshape(-1, n).sum(0))
        b2.acc(g.reshape(-1, n).sum(0))
        # y = xc / d, d = s(xc) + eps, ds/dxc = xc / ((n - 1) s)
        dxc = gy / d - (gy * xc).sum(-1, keepdims=True) / (d * d) * xc / ((n - 1) * s)
        x.acc(dxc - dxc.mean(-1, keepdims=True))
    out.back = back
    return out


def embedding(lut, ids, mult):
    out = T(lut.v[ids] * mult, (lut,))

    def back(g):
        d = np.zeros_like(lut.v)
        np.add.at(d, ids.reshape(-1), g.reshape(-1, g.shape[-1]) * mult)
        lut.acc(d)
    out.back = back
    return out


# ------------------------------------------------------------------------------------------------
# the model (parameter names = the reference's state-dict keys)
# ------------------------------------------------------------------------------------------------
class Drop:
    """nn.Dropout in call order.  masks = None: p = 0 (identity).  Otherwise an iterator over the reference's keep
    masks (bool arrays, flattened) in the order its Dropout modules fired; skip(n) discards n calls (forward passes
    whose outputs the step never uses: the teacher-forced decodes of the `like` views, model.py:248-249)."""

    def __init__(self, masks=None, p=0.0):
        self.masks, self.p, self.k = masks, p, 0

    def __call__(self, x):
        if self.masks is None:
            return x
        m = self.masks[self.k]
        self.k += 1
        assert m.size >= x.v.size, (self.k, m.size, x.v.shape)
        keep = m[: x.v.size].reshape(x.v.shape)
        return mul_const(x, keep / (1.0 - self.p))

    def skip(self, n):
        if self.masks is not None:
            self.k += n


def mha(P, prefix, q_in, k_in, v_in, keep, drop, h=4):
    B, d = q_in.v.shape[0], q_in.v.shape[-1]
    dk = d // h
    heads = lambda t: transpose(reshape(t, (B, -1, h, dk)), 1, 2)
    q = heads(linear(q_in, P[prefix + ".linears.0.weight"], P[prefix + ".linears.0.bias"]))
    k = heads(linear(k_in, P[prefix + ".linears.1.weight"], P[prefix + ".linears.1.bias"]))
    v = heads(linear(v_in, P[prefix + ".linears.2.weight"], P[prefix + ".linears.2.bias"]))
    sc = scale(matmul(q, transpose(k, -1, -2)), 1.0 / math.sqrt(dk))
    sc = masked_fill(sc, keep[:, None], -1e9)
    p = drop(softmax(sc))
    x = reshape(transpose(matmul(p, v), 1, 2), (B, -1, d))
    return linear(x, P[prefix + ".linears.3.weight"], P[prefix + ".linears.3.bias"])


def ffn(P, prefix, x, drop):
    hdn = drop(relu(linear(x, P[prefix + ".w_1.weight"], P[prefix + ".w_1.bias"])))
    return linear(hdn, P[prefix + ".w_2.weight"], P[prefix + ".w_2.bias"])


def sublayer(P, prefix, x, fn, drop):
    return add(x, drop(fn(layer_norm(x, P[prefix + ".norm.a_2"], P[prefix + ".norm.b_2"]))))


def n_layers(P, which):
    return 1 + max(int(k.split(".")[2]) for k in P if k.startswith(which + ".layers."))


def encode_latent(P, src, drop, pe):
    """EncoderDecoder.forward's first half (model.py:236-240): latent = sum over non-pad positions of sigmoid(encoder)"""
    keep = (src != 0)[:, None, :]                       # src_mask [B,1,L]
    x = drop(add(embedding(P["src_embed.0.lut.weight"], src, 16.0), T(pe[None, : src.shape[1]])))
    for l in range(n_layers(P, "encoder")):
        p = "encoder.layers.%d" % l
        x = sublayer(P, p + ".sublayer.0", x, lambda y: mha(P, p + ".self_attn", y, y, y, keep, drop), drop)
        x = sublayer(P, p + ".sublayer.1", x, lambda y: ffn(P, p + ".feed_forward", y, drop), drop)
    x = sigmoid(layer_norm(x, P["encoder.norm.a_2"], P["encoder.norm.b_2"]))
    return reshape(matmul(T(keep.astype(F)), x), (src.shape[0], -1))     # bmm(src_mask, latent).squeeze(1)


def decode_logp(P, latent, tgt, drop, pe):
    """decode + generator (model.py:248-249, :264-269): memory = latent.unsqueeze(1) with an all-ones source mask"""
    B, Tn = tgt.shape
    keep = ((tgt != 0)[:, None, :] & np.tril(np.ones((Tn, Tn), bool))[None])        # make_std_mask
    ones = np.ones((B, 1, 1), bool)
    m = reshape(latent, (B, 1, -1))
    x = drop(add(embedding(P["src_embed.0.lut.weight"], tgt, 16.0), T(pe[None, :Tn])))
    for l in range(n_layers(P, "decoder")):
        p = "decoder.layers.%d" % l
        x = sublayer(P, p + ".sublayer.0", x, lambda y: mha(P, p + ".self_attn", y, y, y, keep, drop), drop)
        x = sublayer(P, p + ".sublayer.1", x, lambda y: mha(P, p + ".src_attn", y, m, m, ones, drop), drop)
        x = sublayer(P, p + ".sublayer.2", x, lambda y: ffn(P, p + ".feed_forward", y, drop), drop)
    x = layer_norm(x, P["decoder.norm.a_2"], P["decoder.norm.b_2"])
    return log_softmax(linear(x, P["generator.proj.weight"], P["generator.proj.bias"]))


DROPS_ENC, DROPS_DEC = 9, 13      # nn.Dropout calls of one forward(): encoder half, decoder half (N = 2 layers)


def label_smoothing_loss(logp, tgt_y, V, pad=0, smoothing=0.1):
    """LabelSmoothing.forward (model.py:401-411) with KLDivLoss(size_average=False): sum t (log t - x)"""
    x = reshape(logp, (-1, V))
    y = tgt_y.reshape(-1)
    t = np.full(x.v.shape, smoothing / (V - 2), F)
    t[np.arange(y.size), y] = 1.0 - smoothing
    t[:, pad] = 0.0
    t[y == pad] = 0.0
    with np.errstate(divide="ignore", invalid="ignore"):
        ent = np.where(t > 0, t * np.log(t), 0.0).sum()
    out = T(ent - (t * x.v).sum(), (x,))
    out.back = lambda g: x.acc(-t * g)
    return out


def pad_batch(seqs, V, bos=2, eos=3, unk=1, max_seq_length=60, not_for_max=None):
    """pad_batch_seuqences (data.py:119-152): src [n, max_l-1], tgt (decoder input) [n, max_l], tgt_y [n, max_l]"""
    max_l = max(len(s) for s in seqs)
    if not_for_max is not None:
        max_l = max(max_l, not_for_max)
    max_l = min(max_seq_length, max_l + 1)
    src = np.zeros((len(seqs), max_l - 1), np.int64)
    tgt = np.zeros((len(seqs), max_l), np.int64)
    tgt_y = np.zeros((len(seqs), max_l), np.int64)
    for i, s in enumerate(seqs):
        tgt[i, 0] = bos
        n = min(max_l - 1, len(s))
        for j in range(n):
            t = s[j] if s[j] < V else unk
            src[i, j] = t
            tgt[i, j + 1] = t
            tgt_y[i, j] = t
        tgt_y[i, min(max_l, len(s) + 1) - 1] = eos
    return src, tgt, tgt_y


def train_step_loss_and_grads(sd, origin, like_views, bata=0.1, tau=0.1, masks=None, p_drop=0.0):
    """One loop body (:1030-1072).  origin: B id lists; like_views: K lists of B id lists (view-major, as
    like_tensor_src [like_num, batch, seq]).  Returns (rec_loss, constraint_loss, grads by parameter name).
    masks: the reference's Dropout keep-masks of this step in call order (None = dropout 0)."""
    V = sd["generator.proj.weight"].shape[0]
    names = [k for k in sd if not k.endswith(".pe") and k != "tgt_embed.0.lut.weight"]
    P = {k: T(sd[k]) for k in names}
    pe = O.positional_encoding(5000).astype(F)
    drop = Drop(masks, p_drop)
    src, tgt, tgt_y = pad_batch(origin, V)
    B = len(origin)
    # (1) ae_model.forward(ori): latent + teacher-forced reconstruction
    ori_latent = encode_latent(P, src, drop, pe)
    logp = decode_logp(P, ori_latent, tgt, drop, pe)
    ntokens = float((tgt_y != 0).sum())
    rec = scale(label_smoothing_loss(logp, tgt_y, V), 1.0 / ntokens)
    # (2) the like views (batch_forward: one forward() per view, padded to the longest sentence of ALL views) and the
    #     second pass over the origins; their decoder halves produce nothing the loss uses
    nfm = max(len(s) for view in like_views for s in view)
    lats = []
    for view in like_views:
        vsrc, _, _ = pad_batch(view, V, not_for_max=nfm)
        lats.append(encode_latent(P, vsrc, drop, pe))
        drop.skip(DROPS_DEC)
    lats.append(encode_latent(P, src, drop, pe))
    drop.skip(DROPS_DEC)
    # (3) constraint: mean over views of NLL(log_softmax(cos / tau), arange(B))   (:1060-1071)
    ori_v = ori_latent.v
    losses, d_ori = [], np.zeros_like(ori_v)
    gw = np.full(B, 1.0 / (B * len(lats)), F)
    for lt in lats:
        l, _ = O.simloss(ori_v, lt.v, tau, 0)
        losses.append(float(np.mean(l)))
    con = float(np.mean(losses))
    # backward of bata * con + rec: the similarity block's gradients enter the tape at the latents
    roots = [(rec, 1.0)]
    for lt in lats:
        do, dl = O.simloss_grads(ori_v, lt.v, gw * bata, tau, 0)
        d_ori += do
        roots.append((lt, dl))
    roots.append((ori_latent, d_ori))
    backward(roots)
    grads = {k: (P[k].g if P[k].g is not None else np.zeros_like(P[k].v)) for k in names}
    return float(rec.v), con, grads


def noam_rate(step, d_model=256, factor=1, warmup=2000):
    return factor * (d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.98, eps=1e-9):
    """torch.optim.Adam (no weight decay / amsgrad), step is 1-based"""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    den = np.sqrt(v) / math.sqrt(1 - b2 ** step) + eps
    return p - (lr / (1 - b1 ** step)) * m / den, m, v
