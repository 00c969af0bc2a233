"""Time of one decode step in the straggler regime (a handful of live rows): st_decode_greedy on B rows without early exit."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from style_transfer_amd import ops, weights as W  # noqa: E402

dev = torch.device("cuda", 0)
sd = W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(ROOT, "tests", "golden", "surrogate_ae_yelp.safetensors")))
ae = W.AEWeights({k: torch.as_tensor(v) for k, v in sd.items()}, dev, gemm_format="f16x3")
ws = ops.DecodeWorkspace(dev)
for B in (4, 8, 64, 256):
    lat = torch.rand(B, 256, device=dev) * 8
    for _ in range(2):
        ids, _, steps = ops.greedy_decode(ae, lat, early_exit=False, sync_every=4, ws=ws)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 5
    for _ in range(n):
        ids, _, steps = ops.greedy_decode(ae, lat, early_exit=False, sync_every=4, ws=ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("B=%d: %d steps in %.2f ms = %.1f us per step" % (B, steps, dt * 1e3, dt * 1e6 / steps))
