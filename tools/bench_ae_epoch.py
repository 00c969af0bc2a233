"""Wall time per batch of the AE constraint-training EPOCH LOOP (train_ae.train_ae_model_constraint: host packing, H2D, step,
optimizer, loss bookkeeping) against the bare step on a pre-packed batch, and against the round-2 form of the loop
(per-sentence Python packing, losses fetched every step).   python tools/bench_ae_epoch.py [batch_size] [n_batches]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from style_transfer_amd import weights as W  # noqa: E402
from style_transfer_amd.data import AttackSet  # noqa: E402
from style_transfer_amd.train_ae import AETrainConfig, AETrainer, train_ae_model_constraint  # noqa: E402
from style_transfer_amd.train_scorer import epoch_order  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
sd = W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(ROOT, "tests", "golden", "surrogate_ae_yelp.safetensors")))
data = AttackSet.synthetic(B * NB, 9339, "yelp", n_pos=10, n_neg=0, seed=3)


def old_loop(tr):
    for recs in epoch_order(len(data), B):
        origin = [data.origin[i] for i in recs]
        views = [[data.like[i][j] for i in recs] for j in range(data.n_pos)]
        tr.loss_and_grads(tr.pack(origin, views))
        tr.step()


out = {"B": B, "batches": NB}
for name, fn in (("epoch_loop", lambda tr: train_ae_model_constraint(tr, data, B, 1, log=lambda *a: None)), ("round2_loop", old_loop)):
    tr = AETrainer(sd, AETrainConfig(), dev)
    fn(tr)                      # warm: workspace allocation, first-call costs
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(tr)
    torch.cuda.synchronize()
    out[name + "_ms_per_batch"] = (time.perf_counter() - t) / NB * 1e3
tr = AETrainer(sd, AETrainConfig(), dev)
recs = epoch_order(len(data), B)[0]
batch = tr.pack([data.origin[i] for i in recs], [[data.like[i][j] for i in recs] for j in range(10)])
for _ in range(3):
    tr.loss_and_grads(batch)
    tr.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    tr.loss_and_grads(batch, fetch=False)
    tr.step()
torch.cuda.synchronize()
out["bare_step_ms"] = (time.perf_counter() - t) / 20 * 1e3
print(json.dumps(out))
