"""AE constraint-training step with the in-batch-loss views in one launch pair up to different block sizes
(g_sl_small_views_max), against the per-view GEMM path:  python tools/ae_views_limit_probe.py"""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from style_transfer_amd import _lib, weights as W  # noqa: E402
from style_transfer_amd.data import AttackSet  # noqa: E402
from style_transfer_amd.train_ae import AETrainConfig, AETrainer  # noqa: E402

dev = torch.device("cuda", 0)
sd = W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(ROOT, "tests", "golden", "surrogate_ae_yelp.safetensors")))
sw = C.c_int.in_dll(_lib.lib(), "g_sl_small_views_max")
for B in (128, 256, 384, 512, 768):
    tr = AETrainer(sd, AETrainConfig(), dev)
    data = AttackSet.synthetic(B, tr.vocab, "yelp", n_pos=10, n_neg=0, seed=3)
    batch = tr.pack([data.origin[i] for i in range(B)], [[data.like[i][j] for i in range(B)] for j in range(10)])
    out = {}
    for lim in (128, 1024):
        sw.value = lim
        for _ in range(2):
            tr.loss_and_grads(batch)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            rec, con = tr.loss_and_grads(batch)
        torch.cuda.synchronize()
        out[lim] = ((time.perf_counter() - t) / 10 * 1e3, con)
    sw.value = 384
    print(json.dumps({"B": B, "ms_limit_128": out[128][0], "ms_limit_1024": out[1024][0], "con_128": out[128][1], "con_1024": out[1024][1]}), flush=True)
