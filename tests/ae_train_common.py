"""Shared by the CPU and GPU tests of the AE constraint-training step: the golden batches and the dropout masks."""
import json
import os

import numpy as np

from conftest import GOLD

PER_FILE = 32


def records():
    recs = []
    for i in (0, 1):
        d = json.load(open(os.path.join(GOLD, "yelp", "neg10_pos10_test%d.json" % i)))
        recs += d["data"][:PER_FILE]
    return recs


def golden_batch(g, k):
    """batch k of the reference run: (origin id lists, 10 view-major lists of id lists)"""
    recs = records()
    B = PER_FILE
    batch = [recs[i] for i in g["order"][k * B:(k + 1) * B]]
    parse = lambda s: [int(x) for x in s.split()]
    return [parse(r["origin"]) for r in batch], [[parse(r["sim%d" % j]) for r in batch] for j in range(10)]


def reference_masks(g, step=0):
    sizes, flat = g["mask_sizes_%d" % step], g["masks_%d" % step]
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [np.unpackbits(flat[off[i]:off[i + 1]]).astype(bool) for i in range(len(sizes))]


def hashed_masks(origin, views, seed, p, n_layers=2, d_ff=1024):
    """The keep-masks the HIP step draws (train_ae.keep_mask), laid out as the reference's nn.Dropout calls see them:
    padded [B, L] batches, one forward() per pass, call order of model.py (encoder half of every pass; decoder half of
    the first pass only, the other decoder halves are placeholders the restatement skips)."""
    from style_transfer_amd.train_ae import SITE_DEC_PE, SITE_ENC_PE, dec_site, enc_site, keep_mask
    B = len(origin)
    masks = []
    rowwise = lambda site, s0, L, w: keep_mask(seed, site, ((np.arange(s0, s0 + B, dtype=np.uint64)[:, None, None] * 64
                                                            + np.arange(L, dtype=np.uint64)[None, :, None]) * w
                                                           + np.arange(w, dtype=np.uint64)[None, None, :]), p).ravel()
    attn = lambda site, s0, L: keep_mask(seed, site, (((np.arange(s0, s0 + B, dtype=np.uint64)[:, None, None, None] * 4
                                                        + np.arange(4, dtype=np.uint64)[None, :, None, None]) * 64
                                                       + np.arange(L, dtype=np.uint64)[None, None, :, None]) * 64
                                                      + np.arange(L, dtype=np.uint64)[None, None, None, :]), p).ravel()

    def enc_pass(s0, L):
        masks.append(rowwise(SITE_ENC_PE, s0, L, 256))
        for l in range(n_layers):
            masks.append(attn(enc_site(l, 0), s0, L))
            masks.append(rowwise(enc_site(l, 1), s0, L, 256))
            masks.append(rowwise(enc_site(l, 2), s0, L, d_ff))
            masks.append(rowwise(enc_site(l, 3), s0, L, 256))

    Lo = max(len(s) for s in origin)
    T = Lo + 1
    enc_pass(0, Lo)
    masks.append(rowwise(SITE_DEC_PE, 0, T, 256))
    for l in range(n_layers):
        masks.append(attn(dec_site(l, 0), 0, T))
        masks.append(rowwise(dec_site(l, 1), 0, T, 256))
        masks.append(rowwise(dec_site(l, 2), 0, T, 4).reshape(B, T, 4).transpose(0, 2, 1).ravel())     # p_attn [B, h, T, 1]
        masks.append(rowwise(dec_site(l, 3), 0, T, 256))
        masks.append(rowwise(dec_site(l, 4), 0, T, d_ff))
        masks.append(rowwise(dec_site(l, 5), 0, T, 256))
    Lv = max(len(s) for view in views for s in view)
    for v in range(len(views)):
        enc_pass((1 + v) * B, Lv)
        masks += [None] * 13
    enc_pass((1 + len(views)) * B, Lo)
    masks += [None] * 13
    return masks


NOISE_TENSORS = ("self_attn.linears.1.bias",)      # key-projection biases: the true gradient is 0 (softmax shift invariance)
ZERO_TENSORS = ("src_attn.linears.0.", "src_attn.linears.1.", "sublayer.1.norm.")   # decoder: softmax over ONE key


def is_noise(name):
    return name.endswith(NOISE_TENSORS)


def is_zero(name):
    return name.startswith("decoder.") and any(z in name for z in ZERO_TENSORS)
