"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, fp32) of ONE optimisation step of the reference's
`train_contrastive` (siamese_contrastive.py:858-947): SiameseNet / Predictor in TRAIN mode (batch-statistics
BatchNorm, Dropout 0.2), `get_contrastive_loss` in its default NLL form (:367-453), manual back-propagation to
every parameter, the BatchNorm running-statistics updates, and torch.optim.Adam over the parameter list of :881-883.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
Pinned by tests/golden/scorer_train_*.npz, which oracle/make_train_golden.py produced by running the imported,
unmodified reference modules (dropout masks captured with forward hooks, so the step is deterministic).

Conventions
  ssd / psd     SiameseNet / Predictor state dicts (numpy fp32), reference key names
  ori [B,256]; like [Kp,B,256]; unlike [Kn,B,256]   (latents, already detached: :915-916)
  masks         bool [G,2,B,512] keep-masks of d1 / d2 for the G = 1+Kp+Kn SiameseNet calls in the reference's
                order (ori, like_0.., unlike_0..: :380-382); None = no dropout
"""
import math

import numpy as np

F32 = np.float32

# list(siamese_net.parameters()) + list(predictor_net.parameters())  (:881-882) in nn.Module registration order
PARAM_ORDER = [("s", "fc1.weight"), ("s", "fc1.bias"), ("s", "norm1.weight"), ("s", "norm1.bias"),
               ("s", "fc2.weight"), ("s", "fc2.bias"), ("s", "norm2.weight"), ("s", "norm2.bias"),
               ("s", "fc3.weight"), ("s", "fc3.bias"),
               ("p", "fc1.weight"), ("p", "fc1.bias"), ("p", "norm1.weight"), ("p", "norm1.bias"),
               ("p", "fc2.weight"), ("p", "fc2.bias")]
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _lin(x, sd, p):
    return (x @ sd[p + ".weight"].T + sd[p + ".bias"]).astype(F32)


def bn_train(x, sd, p, stats):
    """nn.BatchNorm1d in train mode: normalise with the batch's mean / biased variance; running statistics move by
    momentum 0.1 towards the batch mean / UNBIASED variance.  `stats` (dict) is updated in place."""
    n = x.shape[0]
    mean = x.mean(0, dtype=F32)
    var = ((x - mean) ** 2).mean(0, dtype=F32)
    inv = (1.0 / np.sqrt(var + F32(BN_EPS))).astype(F32)
    xhat = ((x - mean) * inv).astype(F32)
    stats[p + ".running_mean"] = ((1 - BN_MOMENTUM) * stats[p + ".running_mean"] + BN_MOMENTUM * mean).astype(F32)
    stats[p + ".running_var"] = ((1 - BN_MOMENTUM) * stats[p + ".running_var"]
                                 + BN_MOMENTUM * var * (n / (n - 1.0))).astype(F32)
    stats[p + ".num_batches_tracked"] = stats.get(p + ".num_batches_tracked", 0) + 1
    return (xhat * sd[p + ".weight"] + sd[p + ".bias"]).astype(F32), (xhat, inv)


def bn_train_bwd(dy, cache, gamma):
    xhat, inv = cache
    n = dy.shape[0]
    dgamma = (dy * xhat).sum(0, dtype=F32)
    dbeta = dy.sum(0, dtype=F32)
    dxh = dy * gamma
    dx = (inv / F32(n)) * (F32(n) * dxh - dxh.sum(0, dtype=F32) - xhat * (dxh * xhat).sum(0, dtype=F32))
    return dx.astype(F32), dgamma, dbeta


def siamese_train_forward(ssd, x, keep, stats, drop_p=0.2):
    """SiameseNet.forward in train mode (:47-62) on one [B,256] call; keep: bool [2,B,512] or None."""
    sc = F32(1.0 / (1.0 - drop_p))
    h1 = _lin(x, ssd, "fc1")
    y1, c1 = bn_train(h1, ssd, "norm1", stats)
    a1 = np.maximum(y1, 0)
    if keep is not None:
        a1 = (a1 * keep[0] * sc).astype(F32)
    h2 = _lin(a1, ssd, "fc2")
    y2, c2 = bn_train(h2, ssd, "norm2", stats)
    a2 = np.maximum(y2, 0)
    if keep is not None:
        a2 = (a2 * keep[1] * sc).astype(F32)
    z = _lin(a2, ssd, "fc3")
    return z, dict(x=x, y1=y1, c1=c1, a1=a1, y2=y2, c2=c2, a2=a2, keep=keep, sc=sc)


def siamese_train_backward(ssd, dz, c, grads):
    def acc(k, v):
        grads[k] = grads.get(k, 0) + v.astype(F32)
    acc("s.fc3.weight", dz.T @ c["a2"])
    acc("s.fc3.bias", dz.sum(0))
    da2 = dz @ ssd["fc3.weight"]
    if c["keep"] is not None:
        da2 = da2 * c["keep"][1] * c["sc"]
    dy2 = np.where(c["y2"] > 0, da2, 0).astype(F32)
    dh2, dg, db = bn_train_bwd(dy2, c["c2"], ssd["norm2.weight"])
    acc("s.norm2.weight", dg)
    acc("s.norm2.bias", db)
    acc("s.fc2.weight", dh2.T @ c["a1"])
    acc("s.fc2.bias", dh2.sum(0))
    da1 = dh2 @ ssd["fc2.weight"]
    if c["keep"] is not None:
        da1 = da1 * c["keep"][0] * c["sc"]
    dy1 = np.where(c["y1"] > 0, da1, 0).astype(F32)
    dh1, dg, db = bn_train_bwd(dy1, c["c1"], ssd["norm1.weight"])
    acc("s.norm1.weight", dg)
    acc("s.norm1.bias", db)
    acc("s.fc1.weight", dh1.T @ c["x"])
    acc("s.fc1.bias", dh1.sum(0))


def _cos_scores(p, Z, tau):
    """F.cosine_similarity(p [B,d], Z [K,B,d], dim=-1).transpose(0,1) / tau  -> [B,K] (+ pieces for the backward)."""
    pn = np.maximum(np.sqrt((p * p).sum(-1, dtype=F32)), F32(1e-8))          # [B]
    zn = np.maximum(np.sqrt((Z * Z).sum(-1, dtype=F32)), F32(1e-8))          # [K,B]
    pu = (p / pn[:, None]).astype(F32)
    zu = (Z / zn[..., None]).astype(F32)
    cos = np.einsum("bd,kbd->bk", pu, zu).astype(F32)
    return (cos / F32(tau)).astype(F32), (pn, zn, pu, zu, cos)


def scorer_train_loss_and_grads(ssd, psd, ori, like, unlike, masks=None, tau=0.1, simplest=False, just_use_one=False,
                                drop_p=0.2):
    """-> (loss, choice [Kp,B] int, grads {"s.fc1.weight": ...}, new BatchNorm statistics {"s.norm1.running_mean": ...})"""
    kp, kn, B = like.shape[0], unlike.shape[0], ori.shape[0]
    groups = [ori] + [like[i] for i in range(kp)] + [unlike[i] for i in range(kn)]
    sstats = {k: v.copy() for k, v in ssd.items() if "running" in k}
    pstats = {k: v.copy() for k, v in psd.items() if "running" in k}
    zs, caches = [], []
    for g, x in enumerate(groups):                                            # :380-382
        z, c = siamese_train_forward(ssd, x, None if masks is None else masks[g], sstats, drop_p)
        zs.append(z)
        caches.append(c)
    Z = np.stack(zs[1:])                                                      # [Kp+Kn,B,256]
    if simplest:
        p = zs[0]
    else:                                                                     # Predictor.forward, train mode (:77-86)
        hp = _lin(zs[0], psd, "fc1")
        yp, cp = bn_train(hp, psd, "norm1", pstats)
        q = np.where(yp > 0, yp, F32(0.2) * yp).astype(F32)
        p = _lin(q, psd, "fc2")
    s, (pn, zn, pu, zu, cos) = _cos_scores(p, Z, tau)                          # [B,Kp+Kn]
    pos, neg = s[:, :kp], s[:, kp:]
    losses, dls, choice = [], [], []
    for i in range(kp):                                                       # :438-446
        sc = np.concatenate([pos[:, i:i + 1], neg], 1)
        m = sc.max(1, keepdims=True)
        lse = m + np.log(np.exp(sc - m, dtype=F32).sum(1, keepdims=True, dtype=F32))
        logp = (sc - lse).astype(F32)
        choice.append(logp.argmax(1))
        losses.append(F32(-logp[:, 0].mean(dtype=F32)))
        sm = np.exp(logp, dtype=F32)
        d = np.zeros((B, kp + kn), F32)
        d[:, i] = (sm[:, 0] - 1) / B
        d[:, kp:] = sm[:, 1:] / B
        dls.append(d)
    if just_use_one:                                                          # :449-452
        k = int(np.argmin(losses))
        loss, ds = losses[k], dls[k]
    else:
        loss, ds = F32(sum(losses) / F32(kp)), (sum(dls) / F32(kp)).astype(F32)
    dcos = (ds / F32(tau)).astype(F32)                                        # [B,K]
    # cos = pu . zu ;  d/dp = (zu - cos pu)/|p| ;  d/dz = (pu - cos zu)/|z|
    dp = ((np.einsum("bk,kbd->bd", dcos, zu) - (dcos * cos).sum(1, keepdims=True) * pu) / pn[:, None]).astype(F32)
    grads = {}
    if simplest:
        dz0 = dp
        dZ = ((dcos.T[..., None] * (pu[None] - cos.T[..., None] * zu)) / zn[..., None]).astype(F32)  # like/unlike not detached (:390-400)
    else:
        grads["p.fc2.weight"] = (dp.T @ q).astype(F32)
        grads["p.fc2.bias"] = dp.sum(0).astype(F32)
        dq = dp @ psd["fc2.weight"]
        dyp = np.where(yp > 0, dq, F32(0.2) * dq).astype(F32)
        dhp, dg, db = bn_train_bwd(dyp, cp, psd["norm1.weight"])
        grads["p.norm1.weight"], grads["p.norm1.bias"] = dg, db
        grads["p.fc1.weight"] = (dhp.T @ zs[0]).astype(F32)
        grads["p.fc1.bias"] = dhp.sum(0).astype(F32)
        dz0 = (dhp @ psd["fc1.weight"]).astype(F32)
        dZ = None                                                             # detached (:406-407)
    siamese_train_backward(ssd, dz0, caches[0], grads)
    if dZ is not None:
        for g in range(kp + kn):
            siamese_train_backward(ssd, dZ[g], caches[g + 1], grads)
    stats = {"s." + k: v for k, v in sstats.items()}
    stats.update({"p." + k: v for k, v in pstats.items()})
    return loss, np.stack(choice), grads, stats


def adam_update(param, grad, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (defaults, no weight decay), step number t (1-based) -> (param, m, v)."""
    m = (m + F32(1 - b1) * (grad - m)).astype(F32)
    v = (v * F32(b2) + F32(1 - b2) * grad * grad).astype(F32)
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = (np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)).astype(F32)
    return (param - F32(lr / bc1) * (m / denom)).astype(F32), m, v
