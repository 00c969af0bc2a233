"""Debug aid: the forced exact-f32 redo of the AE step's forward against the exact-f32 forward."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from style_transfer_amd import _lib, weights as W  # noqa: E402
from style_transfer_amd.train_ae import AETrainConfig, AETrainer  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
sd = W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_yelp.safetensors")))
L = _lib.lib()
tr = AETrainer(sd, AETrainConfig(dropout=0.1, seed=1234), torch.device("cuda", 0))
rng = np.random.default_rng(5)
sent = lambda: [int(t) for t in rng.integers(4, tr.vocab, int(rng.integers(17, 31) if rng.random() < 0.2 else rng.integers(3, 16)))]
origin = [sent() for _ in range(64)]
views = [[sent() for _ in range(64)] for _ in range(10)]
batch = tr.pack(origin, views)
res = {}
for name, sw, v in (("f32", "g_ae_fwd_h", 0), ("redo", "g_ae_force_redo", 1), ("f32 again", "g_ae_fwd_h", 0), ("redo again", "g_ae_force_redo", 1)):
    s = C.c_int.in_dll(L, sw)
    old = s.value
    s.value = v
    rec, con = tr.loss_and_grads(batch, seed=99)
    s.value = old
    res[name] = (rec, con, tr.grads.double().cpu().numpy())
    print(name, repr(rec), repr(con))
for a, b in (("f32", "f32 again"), ("redo", "redo again"), ("f32", "redo")):
    d = res[a][2] - res[b][2]
    print(a, "vs", b, ": differing elements", int((d != 0).sum()), "of", d.size, " max |d| %.3g" % np.abs(d).max())
    if (d != 0).any():
        for i, n in enumerate(tr.names):
            dd = d[tr.offsets[i]:tr.offsets[i + 1]]
            if (dd != 0).any():
                print("   %-50s %8d differ, |d| / |g| = %.3g" % (n, int((dd != 0).sum()), np.linalg.norm(dd) / max(np.linalg.norm(res[a][2][tr.offsets[i]:tr.offsets[i + 1]]), 1e-30)))
