"""Debug aid: gradients of one AE training step with the forward products in split f16 against exact f32 (golden batch 0, dropout 0)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from style_transfer_amd import _lib, weights as W  # noqa: E402
from style_transfer_amd.train_ae import AETrainConfig, AETrainer  # noqa: E402
import ae_train_common as AC  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
sd = W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_yelp.safetensors")))
g = np.load(os.path.join(GOLD, "ae_train_nodrop.npz"))
tr = AETrainer(sd, AETrainConfig(dropout=0.0, seed=1234), torch.device("cuda", 0))
origin, views = AC.golden_batch(g, 0)
batch = tr.pack(origin, views)
L = _lib.lib()
out = {}
for v in (1, 0):
    C.c_int.in_dll(L, "g_ae_fwd_h").value = v
    rec, con = tr.loss_and_grads(batch, seed=1)
    out[v] = (rec, con, tr.grads.double().cpu().numpy())
print("losses  h:", out[1][:2], " f32:", out[0][:2], " ref:", float(g["rec_loss"][0]), float(g["con_loss"][0]))
gs = g["grad"][0].astype(np.float64)
for v in (1, 0):
    print("fwd_h=%d: strided elements vs reference: max |d| / max |ref| = %.3g" % (v, np.abs(out[v][2][::37] - gs).max() / np.abs(gs).max()))
rows = []
for i, n in enumerate(tr.names):
    a, b = out[1][2][tr.offsets[i]:tr.offsets[i + 1]], out[0][2][tr.offsets[i]:tr.offsets[i + 1]]
    nb = np.linalg.norm(b)
    rows.append((np.linalg.norm(a - b) / max(nb, 1e-30), n, nb, abs(np.linalg.norm(a) - g["grad_norms"][0][i]) / max(g["grad_norms"][0][i], 1e-30),
                 abs(nb - g["grad_norms"][0][i]) / max(g["grad_norms"][0][i], 1e-30)))
rows.sort(reverse=True)
for r in rows[:14]:
    print("%-50s |h - f32| / |f32| = %.3g   |f32| = %.3g   norm error vs reference: h %.3g  f32 %.3g" % (r[1], r[0], r[2], r[3], r[4]))
