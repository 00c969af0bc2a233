"""Time st_ae_train_step + st_adam_step at the reference's batch (128 records, 10 views) on synthetic yelp-shaped data."""
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
from style_transfer_amd.train_ae import AETrainConfig, AETrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
sd = W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(ROOT, "tests", "golden", "surrogate_ae_yelp.safetensors")))
tr = AETrainer(sd, AETrainConfig(), dev)
data = AttackSet.synthetic(B, tr.vocab, "yelp", n_pos=10, n_neg=0, seed=3)
origin = [data.origin[i] for i in range(B)]
views = [[data.like[i][j] for i in range(B)] for j in range(10)]
batch = tr.pack(origin, views)
for _ in range(2):
    tr.loss_and_grads(batch)
    tr.step()
torch.cuda.synchronize()
t = time.perf_counter()
n = 10
for _ in range(n):
    rec, con = tr.loss_and_grads(batch)
    tr.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
rows = batch["enc"][0].numel()
print(json.dumps({"B": B, "encoder_rows": rows, "decoder_rows": batch["dec"][0].numel(), "ms_per_step": dt * 1e3,
                  "records_per_s": B / dt, "rec": rec, "con": con}))
