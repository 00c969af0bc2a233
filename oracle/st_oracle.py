"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (style_transfer_amd/*).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

A CPU (numpy, float32) restatement of the reference algorithm on the `--attack` path of
Facico/style-transfer, each function citing the reference file:line it follows
(paths relative to /root/reference/method/yelp/).  Pinned against golden vectors produced by
importing the reference itself in the build container (oracle/make_goldens.py ->
tests/golden/*.npz); see tests/test_oracle_golden.py.

Weights are passed as dict name -> numpy array with the reference's state-dict key names.
"""
import math

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------------------------------------
# Transformer pieces (model.py)
# ------------------------------------------------------------------------------------------------
def positional_encoding(rows, d_model=256):
    """model.py:77-89 (float32 like torch)."""
    pe = np.zeros((rows, d_model), F32)
    position = np.arange(0, rows, dtype=F32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2, dtype=F32) * F32(-(math.log(10000.0) / d_model))).astype(F32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe


def layer_norm(x, a, b, eps=1e-6):
    """model.py:97-108: a*(x-mean)/(std_unbiased+eps)+b."""
    mean = x.mean(-1, keepdims=True, dtype=F32)
    std = x.std(-1, keepdims=True, ddof=1, dtype=F32)
    return (a * (x - mean) / (std + F32(eps)) + b).astype(F32)


def linear(x, sd, prefix):
    return (x @ sd[prefix + ".weight"].T + sd[prefix + ".bias"]).astype(F32)


def softmax(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m, dtype=F32)
    return (e / e.sum(-1, keepdims=True, dtype=F32)).astype(F32)


def mha(sd, prefix, q_in, k_in, v_in, mask, h=4):
    """model.py:30-60 + attention :18-27.  q_in [B,Tq,d], k_in/v_in [B,Tk,d], mask [B,Tq|1,Tk] bool."""
    B, Tq, d = q_in.shape
    dk = d // h
    q = linear(q_in, sd, prefix + ".linears.0").reshape(B, Tq, h, dk).transpose(0, 2, 1, 3)
    k = linear(k_in, sd, prefix + ".linears.1").reshape(B, -1, h, dk).transpose(0, 2, 1, 3)
    v = linear(v_in, sd, prefix + ".linears.2").reshape(B, -1, h, dk).transpose(0, 2, 1, 3)
    scores = (q @ k.transpose(0, 1, 3, 2) / F32(math.sqrt(dk))).astype(F32)
    if mask is not None:
        scores = np.where(mask[:, None, :, :], scores, F32(-1e9))
    p = softmax(scores)
    x = (p @ v).transpose(0, 2, 1, 3).reshape(B, Tq, d)
    return linear(x, sd, prefix + ".linears.3")


def ffn(sd, prefix, x):
    """model.py:63-72."""
    return linear(np.maximum(linear(x, sd, prefix + ".w_1"), 0), sd, prefix + ".w_2")


def embed(sd, ids, pe):
    """model.py:126-133 Embeddings (*sqrt(d)) + :91-94 PositionalEncoding."""
    x = sd["src_embed.0.lut.weight"][ids] * F32(math.sqrt(256.0))
    return (x + pe[None, : ids.shape[1]]).astype(F32)


def n_layers(sd, which):
    return 1 + max(int(k.split(".")[2]) for k in sd if k.startswith(which + ".layers."))


def encode_latent(sd, src, pe=None):
    """EncoderDecoder.get_latent (model.py:253-257) = encoder (:137-163) -> sigmoid -> masked sum.
    src [B,L] int (0 = pad)."""
    pe = positional_encoding(64) if pe is None else pe
    mask = (src != 0)[:, None, :]
    x = embed(sd, src, pe)
    for l in range(n_layers(sd, "encoder")):
        p = "encoder.layers.%d." % l
        xn = layer_norm(x, sd[p + "sublayer.0.norm.a_2"], sd[p + "sublayer.0.norm.b_2"])
        x = x + mha(sd, p + "self_attn", xn, xn, xn, mask)
        xn = layer_norm(x, sd[p + "sublayer.1.norm.a_2"], sd[p + "sublayer.1.norm.b_2"])
        x = x + ffn(sd, p + "feed_forward", xn)
    x = layer_norm(x, sd["encoder.norm.a_2"], sd["encoder.norm.b_2"])
    sig = (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)
    return (mask.astype(F32) @ sig)[:, 0, :].astype(F32)


def decode_full(sd, latent, ys, pe):
    """EncoderDecoder.decode (model.py:264-269) over the whole prefix ys [B,t] (literal form)."""
    B, t = ys.shape
    mem = latent[:, None, :]
    tgt_mask = np.tril(np.ones((t, t), bool))[None]
    x = embed(sd, ys, pe)
    for l in range(n_layers(sd, "decoder")):
        p = "decoder.layers.%d." % l
        xn = layer_norm(x, sd[p + "sublayer.0.norm.a_2"], sd[p + "sublayer.0.norm.b_2"])
        x = x + mha(sd, p + "self_attn", xn, xn, xn, tgt_mask)
        xn = layer_norm(x, sd[p + "sublayer.1.norm.a_2"], sd[p + "sublayer.1.norm.b_2"])
        x = x + mha(sd, p + "src_attn", xn, mem, mem, np.ones((B, 1, 1), bool))
        xn = layer_norm(x, sd[p + "sublayer.2.norm.a_2"], sd[p + "sublayer.2.norm.b_2"])
        x = x + ffn(sd, p + "feed_forward", xn)
    return layer_norm(x, sd["decoder.norm.a_2"], sd["decoder.norm.b_2"])


def greedy_decode_literal(sd, latent, max_len=60, start_id=2, want_gap=False):
    """model.py:271-299 line by line: whole decoder over the growing prefix each step, no early stop.
    (log_softmax is kept out: argmax-invariant.)  -> ids [B,max_len-1] (and top-2 logit gaps)."""
    pe = positional_encoding(max_len)
    B = latent.shape[0]
    ys = np.full((B, 1), start_id, np.int64)
    gaps = np.zeros((B, max_len - 1), F32)
    for i in range(max_len - 1):
        out = decode_full(sd, latent, ys, pe)
        logits = linear(out[:, -1], sd, "generator.proj")
        nxt = logits.argmax(-1)  # first max = lowest index, like torch.max on CPU
        if want_gap:
            part = np.partition(logits, -2, axis=-1)
            gaps[:, i] = part[:, -1] - part[:, -2]
        ys = np.concatenate([ys, nxt[:, None]], 1)
    return (ys[:, 1:], gaps) if want_gap else ys[:, 1:]


def greedy_decode(sd, latent, max_len=60, start_id=2, eos=3, pad=0, early_exit=False, want_gap=False):
    """Same function with a KV cache and the single-key cross-attention folded to a constant
     — the formulation the HIP kernels use; checked against the literal form in
    tests.  early_exit stops a sentence after its first EOS/PAD (later ids = 0)."""
    B = latent.shape[0]
    T = max_len - 1
    L = n_layers(sd, "decoder")
    pe = positional_encoding(max_len)
    emb = sd["src_embed.0.lut.weight"]
    cc = []
    for l in range(L):
        p = "decoder.layers.%d.src_attn" % l
        cc.append(linear(linear(latent, sd, p + ".linears.2"), sd, p + ".linears.3"))
    kc = np.zeros((L, B, T, 256), F32)
    vc = np.zeros((L, B, T, 256), F32)
    ids = np.zeros((B, T), np.int64)
    gaps = np.zeros((B, T), F32)
    tok = np.full((B,), start_id, np.int64)
    live = np.arange(B)
    for pos in range(T):
        if live.size == 0:
            break
        x = (emb[tok[live]] * F32(16.0) + pe[pos]).astype(F32)
        for l in range(L):
            p = "decoder.layers.%d." % l
            xn = layer_norm(x, sd[p + "sublayer.0.norm.a_2"], sd[p + "sublayer.0.norm.b_2"])
            q = linear(xn, sd, p + "self_attn.linears.0")
            kc[l, live, pos] = linear(xn, sd, p + "self_attn.linears.1")
            vc[l, live, pos] = linear(xn, sd, p + "self_attn.linears.2")
            n = live.size
            qh = q.reshape(n, 4, 1, 64)
            kh = kc[l, live, : pos + 1].reshape(n, pos + 1, 4, 64).transpose(0, 2, 1, 3)
            vh = vc[l, live, : pos + 1].reshape(n, pos + 1, 4, 64).transpose(0, 2, 1, 3)
            sc = (qh @ kh.transpose(0, 1, 3, 2) / F32(8.0)).astype(F32)
            att = (softmax(sc) @ vh).reshape(n, 256)
            x = x + linear(att, sd, p + "self_attn.linears.3")
            x = x + cc[l][live]
            xn = layer_norm(x, sd[p + "sublayer.2.norm.a_2"], sd[p + "sublayer.2.norm.b_2"])
            x = x + ffn(sd, p + "feed_forward", xn)
        out = layer_norm(x, sd["decoder.norm.a_2"], sd["decoder.norm.b_2"])
        logits = linear(out, sd, "generator.proj")
        nxt = logits.argmax(-1)
        if want_gap:
            part = np.partition(logits, -2, axis=-1)
            gaps[live, pos] = part[:, -1] - part[:, -2]
        ids[live, pos] = nxt
        tok[live] = nxt
        if early_exit:
            live = live[(nxt != eos) & (nxt != pad)]
    return (ids, gaps) if want_gap else ids


# ------------------------------------------------------------------------------------------------
# Style scorer (siamese_contrastive.py)
# ------------------------------------------------------------------------------------------------
def batchnorm_eval(x, sd, prefix, eps=1e-5):
    return ((x - sd[prefix + ".running_mean"]) / np.sqrt(sd[prefix + ".running_var"] + F32(eps)) * sd[prefix + ".weight"]
            + sd[prefix + ".bias"]).astype(F32)


def siamese_forward(ssd, x, cache=None):
    """SiameseNet.forward, eval mode (siamese_contrastive.py:47-62)."""
    a1 = batchnorm_eval(linear(x, ssd, "fc1"), ssd, "norm1")
    h1 = np.maximum(a1, 0)
    a2 = batchnorm_eval(linear(h1, ssd, "fc2"), ssd, "norm2")
    h2 = np.maximum(a2, 0)
    z = linear(h2, ssd, "fc3")
    if cache is not None:
        cache.update(a1=a1, a2=a2)
    return z


def predictor_forward(psd, z, cache=None):
    """Predictor.forward (siamese_contrastive.py:77-86)."""
    a = batchnorm_eval(linear(z, psd, "fc1"), psd, "norm1")
    q = np.where(a > 0, a, F32(0.2) * a).astype(F32)
    if cache is not None:
        cache.update(aq=a)
    return linear(q, psd, "fc2")


def _bn_scale(sd, prefix, eps=1e-5):
    return (sd[prefix + ".weight"] / np.sqrt(sd[prefix + ".running_var"] + F32(eps))).astype(F32)


def contrastive_loss_and_grad(ssd, psd, ori_latent, like_latent, unlike_latent, tau=0.1, not_nll=False,
                              simplest=False, just_use_one=False):
    """get_contrastive_loss (siamese_contrastive.py:367-453) for ONE sentence + the gradient of
    `-(loss)` w.r.t. ori_latent that `loss.backward()` (:806-809) produces (manual backprop).
    ori_latent [1,256]; like_latent [Kp,256]; unlike_latent [Kn,256] (None -> absent).
    -> (loss as returned by get_contrastive_loss, grad [1,256])."""
    cache = {}
    z = siamese_forward(ssd, ori_latent, cache)
    like_z = None if like_latent is None else siamese_forward(ssd, like_latent)
    unlike_z = None if unlike_latent is None else siamese_forward(ssd, unlike_latent)
    p = z if simplest else predictor_forward(psd, z, cache)
    refs = [r for r in (like_z, unlike_z) if r is not None]
    Z = np.concatenate(refs, 0)
    kp = 0 if like_z is None else like_z.shape[0]
    kn = 0 if unlike_z is None else unlike_z.shape[0]
    pn = max(float(np.sqrt((p * p).sum(dtype=F32))), 1e-8)
    zn = np.maximum(np.sqrt((Z * Z).sum(-1, dtype=F32)), F32(1e-8))
    Zu = (Z / zn[:, None]).astype(F32)
    cos = (Zu @ p[0] / F32(pn)).astype(F32)  # F.cosine_similarity(ori_p, like_z, dim=-1)
    s = (cos / F32(tau)).astype(F32)
    pos, neg = s[:kp], s[kp:]
    if not_nll:
        loss = (neg.sum(dtype=F32) - pos.sum(dtype=F32)) / F32(kp + kn)       # :424-436
        dl_ds = np.concatenate([np.full(kp, -1.0, F32), np.full(kn, 1.0, F32)]) / F32(kp + kn)
    else:
        li, grads = [], []
        for i in range(kp):                                                   # :438-446
            sc = np.concatenate([pos[i:i + 1], neg])
            m = sc.max()
            lse = m + np.log(np.exp(sc - m, dtype=F32).sum(dtype=F32))
            li.append(lse - sc[0])
            sm = np.exp(sc - lse, dtype=F32)
            g = np.zeros(kp + kn, F32)
            g[i] = sm[0] - 1
            g[kp:] = sm[1:]
            grads.append(g)
        if just_use_one:                                                      # :449-452
            k = int(np.argmin(li))
            loss, dl_ds = li[k], grads[k]
        else:
            loss, dl_ds = sum(li) / F32(kp), sum(grads) / F32(kp)
    # d(-loss)/dp through cos = p.zu/|p|
    g_cos = (-dl_ds / F32(tau)).astype(F32)
    dp = ((g_cos @ Zu) / F32(pn) - (g_cos @ cos) * p[0] / F32(pn * pn)).astype(F32)[None]
    if not simplest:
        dq = dp @ psd["fc2.weight"]
        da = np.where(cache["aq"] > 0, dq, F32(0.2) * dq) * _bn_scale(psd, "norm1")
        dz = da @ psd["fc1.weight"]
    else:
        dz = dp
    dh2 = dz @ ssd["fc3.weight"]
    da2 = np.where(cache["a2"] > 0, dh2, 0) * _bn_scale(ssd, "norm2")
    dh1 = da2 @ ssd["fc2.weight"]
    da1 = np.where(cache["a1"] > 0, dh1, 0) * _bn_scale(ssd, "norm1")
    dx = da1 @ ssd["fc1.weight"]
    return F32(loss), dx.astype(F32)


class Adam:
    """torch.optim.Adam, single tensor, defaults (siamese_contrastive.py:780-781)."""

    def __init__(self, shape, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.m = np.zeros(shape, F32)
        self.v = np.zeros(shape, F32)
        self.t = 0
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps

    def step(self, x, g):
        self.t += 1
        self.m = (self.m + F32(1 - self.b1) * (g - self.m)).astype(F32)
        self.v = (self.v * F32(self.b2) + F32(1 - self.b2) * g * g).astype(F32)
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        denom = (np.sqrt(self.v) / F32(math.sqrt(bc2)) + F32(self.eps)).astype(F32)
        return (x - F32(self.lr / bc1) * (self.m / denom)).astype(F32)


class _Rule:
    """the other torch.optim rules --victim_optimizer can name (single tensor, torch defaults apart from lr)"""

    def __init__(self, shape, lr):
        self.m = np.zeros(shape, F32)
        self.v = np.zeros(shape, F32)
        self.t = 0
        self.lr = lr


class SGD(_Rule):
    def step(self, x, g):
        return (x - F32(self.lr) * g).astype(F32)


class AdamW(Adam):
    def step(self, x, g):          # torch.optim.AdamW: param.mul_(1 - lr * weight_decay), weight_decay 0.01, then Adam
        return Adam.step(self, (x * F32(1.0 - self.lr * 0.01)).astype(F32), g)


class Adamax(_Rule):
    def step(self, x, g):          # betas (0.9, 0.999), eps 1e-8
        self.t += 1
        self.m = (self.m + F32(0.1) * (g - self.m)).astype(F32)
        self.v = np.maximum(self.v * F32(0.999), np.abs(g) + F32(1e-8)).astype(F32)
        clr = F32(self.lr / (1 - 0.9 ** self.t))
        return (x - clr * (self.m / self.v)).astype(F32)


class RMSprop(_Rule):
    def step(self, x, g):          # alpha 0.99, eps 1e-8, no momentum
        self.v = (self.v * F32(0.99) + F32(0.01) * g * g).astype(F32)
        return (x - F32(self.lr) * (g / (np.sqrt(self.v) + F32(1e-8)))).astype(F32)


class Adagrad(_Rule):
    def step(self, x, g):          # eps 1e-10, no lr decay
        self.v = (self.v + g * g).astype(F32)
        return (x - F32(self.lr) * (g / (np.sqrt(self.v) + F32(1e-10)))).astype(F32)


OPTIMIZERS = {"Adam": Adam, "SGD": SGD, "AdamW": AdamW, "Adamax": Adamax, "RMSprop": RMSprop, "Adagrad": Adagrad}


# ------------------------------------------------------------------------------------------------
# data.py / attack driver
# ------------------------------------------------------------------------------------------------
def id2text_sentence(ids, id_to_word):
    """data.py:43-52."""
    out = []
    for i in ids:
        i = int(i)
        if i == 3 or i == 0:
            break
        if i >= len(id_to_word):
            i = 1
        out.append(id_to_word[i])
    return " ".join(out)


def pad_sequences(seqs, vocab_size, unk=1, max_seq_length=60, not_for_max=None):
    """encoder inputs of pad_batch_seuqences (data.py:119-152): [n, max_l-1], ids >= V -> UNK."""
    max_l = max(len(s) for s in seqs)
    if not_for_max is not None:
        max_l = max(max_l, not_for_max)
    max_l = min(max_seq_length, max_l + 1)
    out = np.zeros((len(seqs), max_l - 1), np.int64)
    for i, s in enumerate(seqs):
        for j in range(min(max_l - 1, len(s))):
            out[i, j] = s[j] if s[j] < vocab_size else unk
    return out


def headline(s, tok="*"):
    """utils.py:8-10."""
    return tok * 20 + s + tok * 20


def attack_one(ae_sd, ssd, psd, ori_latent, like_latent, unlike_latent, lr, id_to_word, log, tau=0.1,
               not_nll=False, simplest=False, just_use_one=False, max_iter=18, max_change=7, max_len=60,
               decode=None, trace=None, optimizer="Adam"):
    """contrastive_attack_one (siamese_contrastive.py:754-839) without --retext/--dance."""
    decode = decode or (lambda lat: greedy_decode(ae_sd, lat, max_len, early_exit=True))
    x = ori_latent.copy()
    opt = OPTIMIZERS[optimizer](x.shape, lr)
    cur_text = id2text_sentence(decode(x)[0], id_to_word)
    log(headline("", "-"))
    k = 0
    for it in range(max_iter):
        loss, g = contrastive_loss_and_grad(ssd, psd, x, like_latent, unlike_latent, tau, not_nll, simplest, just_use_one)
        x = opt.step(x, g)
        ids = decode(x)[0]
        text = id2text_sentence(ids, id_to_word)
        changed = text != cur_text
        if changed:
            cur_text = text
            log("Change.{}, {}".format(k, cur_text))
            k += 1
        log("iteration.{}, {}".format(it, cur_text))
        if trace is not None:
            trace.append(dict(loss=loss, grad=g.copy(), latent=x.copy(), ids=np.asarray(ids).copy(), changed=changed))
        if k >= max_change:
            break
    return x


# ------------------------------------------------------------------------------------------------
# In-batch InfoNCE (train_ae_model_constraint, siamese_contrastive.py:1060-1069)
# ------------------------------------------------------------------------------------------------
def simloss(ori, like, tau=0.1, row_offset=0):
    """loss[i] = -log_softmax(cos(ori[i], like[:]) / tau)[row_offset + i]; float64 accumulate for use as
    a reference.  -> (loss [Bq], lse [Bq], d_ori, d_like for g = 1/Bq each)."""
    o = ori.astype(np.float64)
    k = like.astype(np.float64)
    on = np.maximum(np.linalg.norm(o, axis=1, keepdims=True), 1e-8)
    kn = np.maximum(np.linalg.norm(k, axis=1, keepdims=True), 1e-8)
    ou, ku = o / on, k / kn
    s = ou @ ku.T / tau
    m = s.max(1, keepdims=True)
    lse = (m + np.log(np.exp(s - m).sum(1, keepdims=True)))[:, 0]
    tgt = row_offset + np.arange(o.shape[0])
    loss = lse - s[np.arange(o.shape[0]), tgt]
    return loss, lse


def simloss_grads(ori, like, g, tau=0.1, row_offset=0):
    o = ori.astype(np.float64)
    k = like.astype(np.float64)
    on = np.maximum(np.linalg.norm(o, axis=1, keepdims=True), 1e-8)
    kn = np.maximum(np.linalg.norm(k, axis=1, keepdims=True), 1e-8)
    ou, ku = o / on, k / kn
    s = ou @ ku.T / tau
    p = np.exp(s - s.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    p[np.arange(o.shape[0]), row_offset + np.arange(o.shape[0])] -= 1.0
    ds = p * g[:, None] / tau
    dou = ds @ ku
    dku = ds.T @ ou
    d_ori = (dou - (dou * ou).sum(1, keepdims=True) * ou) / on
    d_like = (dku - (dku * ku).sum(1, keepdims=True) * ku) / kn
    return d_ori, d_like


# ------------------------------------------------------------------------------------------------
# Legacy FGIM path (model.py:414-433 Classifier, :454-516 fgim_attack)
# ------------------------------------------------------------------------------------------------
def classifier_forward(csd, x):
    """Classifier.forward: sigmoid(fc3(lrelu(fc2(lrelu(fc1(x)))))) -> (out [B,1], pre-activations)."""
    z1 = (x @ csd["fc1.weight"].T + csd["fc1.bias"]).astype(F32)
    a1 = np.where(z1 > 0, z1, F32(0.2) * z1).astype(F32)
    z2 = (a1 @ csd["fc2.weight"].T + csd["fc2.bias"]).astype(F32)
    a2 = np.where(z2 > 0, z2, F32(0.2) * z2).astype(F32)
    z3 = (a2 @ csd["fc3.weight"].T + csd["fc3.bias"]).astype(F32)
    return (F32(1) / (F32(1) + np.exp(-z3, dtype=F32))).astype(F32), (z1, z2)


def fgim_grad(csd, x, target):
    """d BCELoss(Classifier(x), target) / d x, one sentence per loss (the reference's batch of 1):
    dL/dz3 = out - target for sigmoid + BCE."""
    out, (z1, z2) = classifier_forward(csd, x)
    dz3 = (out - target.reshape(-1, 1)).astype(F32)
    d2 = (dz3 @ csd["fc3.weight"]) * np.where(z2 > 0, F32(1), F32(0.2))
    d1 = (d2 @ csd["fc2.weight"]) * np.where(z1 > 0, F32(1), F32(0.2))
    return out[:, 0], (d1 @ csd["fc1.weight"]).astype(F32)


def fgim_perturb(csd, x, target, epsilons):
    """fgim_attack: every epsilon takes ONE step from the original latent (it starts at 4, loop ends at it >= 5)."""
    pred, g = fgim_grad(csd, x, target)
    return pred, g, np.stack([(x - F32(e) * g).astype(F32) for e in epsilons])
