#!/usr/bin/env python3
"""Debug aid: are a sentence's edited latents bit-identical (a) between two runs of the same batch, (b) between a batch
of 32768 and a batch of 1024 that contains it, (c) with and without the trace options?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
from style_transfer_amd import weights as W
from style_transfer_amd.attack import AttackConfig, AttackRunner
from style_transfer_amd.data import AttackSet, load_word_counts

dev = torch.device("cuda", 0)
fmt = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
ae = W.AEWeights(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_yelp.safetensors")), dev)
sc = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                     W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev, gemm_format=fmt)
counts = load_word_counts(os.path.join(GOLD, "yelp", "word_to_id.txt"), ae.vocab)
data = AttackSet.synthetic(32768, ae.vocab, "yelp", seed=2024, word_counts=counts)
cfg = AttackConfig(not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2, log_decodes=False)
run = lambda d, g, c=cfg: AttackRunner(ae, sc, c, dev).run_batch(d, g)
a = run(data, list(range(32768)))
b = run(data, list(range(32768)))
print("scorer", fmt)
print("(a) same batch twice: latents equal", np.array_equal(a.final_latent, b.final_latent), "ids equal", np.array_equal(a.iter_ids, b.iter_ids), "n_iter equal", np.array_equal(a.n_iter, b.n_iter))
lo = 5000
c = run(data.span(lo, lo + 1024), list(range(lo, lo + 1024)))
same_n = a.n_iter[lo:lo + 1024] == c.n_iter
dl = np.abs(a.final_latent[lo:lo + 1024] - c.final_latent).max(1)
print("(b) 1024 of them in their own batch: n_iter equal %d of 1024; latents bit-equal %d; max |d latent| %.3g" % (same_n.sum(), (dl == 0).sum(), dl.max()))
tc = AttackConfig(not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2, log_decodes=False, want_trace=True, want_gap=True)
d = run(data.span(lo, lo + 1024), list(range(lo, lo + 1024)), tc)
dl2 = np.abs(c.final_latent - d.final_latent).max(1)
print("(c) with trace options: n_iter equal %d of 1024; latents bit-equal %d; max %.3g" % ((c.n_iter == d.n_iter).sum(), (dl2 == 0).sum(), dl2.max()))
# the latent after iteration n_iter-1 from the trace must be the final latent
