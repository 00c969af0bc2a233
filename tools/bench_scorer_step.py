"""Stand-alone timing of st_scorer_edit_step (the edit loop's scorer kernel) on synthetic latents.

    python tools/bench_scorer_step.py [--batch 98304] [--format f16x3] [--reps 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=98304)
    ap.add_argument("--format", default="f16x3")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from style_transfer_amd import ops, weights as W
    dev = torch.device("cuda:0")
    sw = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                         W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev, gemm_format=a.format)
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(1)
    lat = torch.rand(B, 256, device=dev, generator=g) * 8
    like = ops.scorer_embed(sw, (lat[:, None, :] + torch.randn(B, 10, 256, device=dev, generator=g)).reshape(-1, 256)).view(B, 10, 256)
    unlike = ops.scorer_embed(sw, torch.rand(B * 10, 256, device=dev, generator=g) * 8).view(B, 10, 256)
    m, v = torch.zeros_like(lat), torch.zeros_like(lat)
    lr = torch.full((B,), 0.2, device=dev)
    step = torch.zeros(B, dtype=torch.int32, device=dev)
    act = torch.ones(B, dtype=torch.uint8, device=dev)
    for _ in range(3):
        ops.scorer_edit_step(sw, lat, m, v, like, unlike, lr, step, act, not_nll=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        ops.scorer_edit_step(sw, lat, m, v, like, unlike, lr, step, act, not_nll=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    fl = B * 2.0 * 2.0 * 575488
    print("scorer edit step %s B=%d: %.1f us per launch, %.1f TFLOP/s algorithmic, unit vectors %.2f TB/s" %
          (a.format, B, us, fl / us / 1e6, B * 20 * 1024 / us / 1e6))


if __name__ == "__main__":
    main()
