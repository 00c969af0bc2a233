"""Debug aid: as ae_fwd_format_probe.py on the synthetic mixed-length batch of test_ae_train_step_kernel_forms_agree."""
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
for drop in (0.0, 0.1):
    for longs in (0.0, 0.2):
        tr = AETrainer(sd, AETrainConfig(dropout=drop, seed=1234), torch.device("cuda", 0))
        rng = np.random.default_rng(5)
        sent = lambda: [int(t) for t in rng.integers(4, tr.vocab, int(rng.integers(17, 31) if rng.random() < longs else rng.integers(3, 16)))]
        B = 64
        origin = [sent() for _ in range(B)]
        views = [[sent() for _ in range(B)] for _ in range(10)]
        batch = tr.pack(origin, views)
        out = {}
        for v in (1, 0):
            C.c_int.in_dll(L, "g_ae_fwd_h").value = v
            rec, con = tr.loss_and_grads(batch, seed=99)
            out[v] = (rec, con, tr.grads.double().cpu().numpy())
        C.c_int.in_dll(L, "g_ae_fwd_h").value = 1
        rows = []
        for i, n in enumerate(tr.names):
            a, b = out[1][2][tr.offsets[i]:tr.offsets[i + 1]], out[0][2][tr.offsets[i]:tr.offsets[i + 1]]
            nb = np.linalg.norm(b)
            if nb > 1e-6:
                rows.append((np.linalg.norm(a - b) / nb, n, nb))
        rows.sort(reverse=True)
        print("dropout %.1f long share %.1f: losses h %r f32 %r; worst tensors:" % (drop, longs, out[1][:2], out[0][:2]))
        for r in rows[:4]:
            print("   %-50s |h - f32| / |f32| = %.3g   |f32| = %.3g" % (r[1], r[0], r[2]))

# sensitivity of the exact-f32 step itself: every parameter perturbed by 3e-7 relative (the size of the split-f16 forward's rounding)
print("exact-f32 forward, parameters perturbed by 3e-7 relative:")
C.c_int.in_dll(L, "g_ae_fwd_h").value = 0
tr = AETrainer(sd, AETrainConfig(dropout=0.1, seed=1234), torch.device("cuda", 0))
rng = np.random.default_rng(5)
sent = lambda: [int(t) for t in rng.integers(4, tr.vocab, int(rng.integers(3, 16)))]
origin = [sent() for _ in range(64)]
views = [[sent() for _ in range(64)] for _ in range(10)]
batch = tr.pack(origin, views)
tr.loss_and_grads(batch, seed=99)
g0 = tr.grads.double().cpu().numpy()
gen = torch.Generator(device="cuda").manual_seed(1)
tr.params.mul_(1.0 + 3e-7 * (2 * torch.rand(tr.params.shape, device="cuda", generator=gen) - 1))
tr.loss_and_grads(batch, seed=99)
g1 = tr.grads.double().cpu().numpy()
rows = []
for i, n in enumerate(tr.names):
    a, b = g1[tr.offsets[i]:tr.offsets[i + 1]], g0[tr.offsets[i]:tr.offsets[i + 1]]
    nb = np.linalg.norm(b)
    if nb > 1e-6:
        rows.append((np.linalg.norm(a - b) / nb, n, nb))
rows.sort(reverse=True)
for r in rows[:5]:
    print("   %-50s |perturbed - base| / |base| = %.3g   |base| = %.3g" % (r[1], r[0], r[2]))
C.c_int.in_dll(L, "g_ae_fwd_h").value = 1
