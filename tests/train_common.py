"""Shared helpers of the scorer-training tests (CPU and GPU)."""
import os

import numpy as np

import st_train_oracle as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["%s.%s" % kv for kv in T.PARAM_ORDER]
STRIDE = 7   # oracle/make_train_golden.py keeps every 7th element of the flat vectors
MODES = [("nll", {}), ("simplest", dict(simplest=True)), ("nll_one", dict(just_use_one=True))]


def load_init():
    from safetensors.numpy import load_file
    return (load_file(os.path.join(GOLD, "scorer_train_init_siamese.safetensors")),
            load_file(os.path.join(GOLD, "scorer_train_init_predictor.safetensors")))


def golden_latents():
    g = np.load(os.path.join(GOLD, "attack_notnll.npz"))
    return (g["init_latent"], g["like_latent"].transpose(1, 0, 2).copy(), g["unlike_latent"].transpose(1, 0, 2).copy())


def golden_masks(gd, step):
    return np.unpackbits(gd["masks"][step], axis=-1).astype(bool)


def flat(sds_or_grads, like):
    """dict keyed 's.fc1.weight' (missing -> zeros shaped like `like[name]`) -> flat vector in PARAM_ORDER"""
    return np.concatenate([np.asarray(sds_or_grads.get(k, np.zeros_like(like[k]))).ravel() for k in NAMES])


def named(ssd, psd):
    d = {"s." + k: v for k, v in ssd.items()}
    d.update({"p." + k: v for k, v in psd.items()})
    return d


def tensor_slices(shapes):
    off, out = 0, {}
    for k in NAMES:
        n = int(np.prod(shapes[k].shape))
        out[k] = slice(off, off + n)
        off += n
    return out


def noise_grad_tensors(gd, step):
    """Tensors whose true gradient is zero (a bias in front of a BatchNorm: the mean subtraction cancels it): the
    reference's values there are rounding noise ~1e-9 that Adam turns into +-lr-sized moves, so neither their gradient
    nor their parameter values are comparable between two fp32 implementations."""
    return {k for k, nrm in zip(NAMES, gd["grad_norms"][step]) if nrm < 1e-6}


def oracle_steps(tag_kw, gd, n_steps, lr):
    """Run the numpy restatement for n_steps from the golden initial weights, replaying the golden masks."""
    ssd, psd = load_init()
    ori, like, unlike = golden_latents()
    sds = {"s": ssd, "p": psd}
    m = {k: np.zeros_like(sds[k[0]][k[2:]]) for k in NAMES}
    v = {k: np.zeros_like(sds[k[0]][k[2:]]) for k in NAMES}
    out = []
    for step in range(n_steps):
        loss, choice, grads, stats = T.scorer_train_loss_and_grads(ssd, psd, ori, like, unlike, golden_masks(gd, step), **tag_kw)
        for k in NAMES:
            if k in grads:
                sd = sds[k[0]]
                sd[k[2:]], m[k], v[k] = T.adam_update(sd[k[2:]], grads[k], m[k], v[k], step + 1, lr)
        for k2, val in stats.items():
            sds[k2[0]][k2[2:]] = val
        out.append(dict(loss=loss, choice=choice, grads=grads, params={k: sds[k[0]][k[2:]].copy() for k in NAMES},
                        stats={k: np.array(val) for k, val in stats.items()}))
    return out
