"""Throughput of the scorer training step (st_scorer_train_step + st_adam_step) on one MI355X.

    python tools/bench_scorer_train.py [--batch 128] [--steps 200] [--cpu-steps 3]

Prints one JSON line: steps/s and records/s (a record = one origin with its Kp + Kn reference sentences) with inputs
resident in HBM, the achieved f32 MFMA rate of the step's matrix products against the 157.3 TFLOP/s peak, and the numpy
restatement (oracle/st_train_oracle.py) timed on the host cores as the CPU baseline.  Latents are synthetic
(U(0,8) origins, like = origin + N(0,1), unlike ~ U(0,8)); weights are freshly initialised.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

F32_MFMA_PEAK_TFLOPS = 157.3


def step_flops(B, Kp, Kn, simplest):
    G = 1 + Kp + Kn
    fwd = 2.0 * G * B * (256 * 512 + 512 * 512 + 512 * 256)
    pred = 0.0 if simplest else 2.0 * B * (256 * 100 + 100 * 256)
    Gb = G if simplest else 1
    bwd = 2.0 * 2.0 * Gb * B * (256 * 512 + 512 * 512 + 512 * 256) - 2.0 * Gb * B * 256 * 512 + 2.0 * pred
    return fwd + pred + bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--kp", type=int, default=10)
    ap.add_argument("--kn", type=int, default=10)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--simplest", action="store_true")
    a = ap.parse_args()
    from style_transfer_amd.train_scorer import ScorerTrainer, TrainConfig, fresh_state_dicts
    dev = torch.device("cuda:0")
    ssd, psd = fresh_state_dicts(seed=1)
    tr = ScorerTrainer(ssd, psd, TrainConfig(simplest=a.simplest), dev)
    rng = np.random.default_rng(1234)
    B, Kp, Kn = a.batch, a.kp, a.kn
    ori = (rng.random((B, 256)) * 8).astype(np.float32)
    like = (ori[None] + rng.standard_normal((Kp, B, 256))).astype(np.float32)
    unlike = (rng.random((Kn, B, 256)) * 8).astype(np.float32)
    d = [torch.from_numpy(x).to(dev) for x in (ori, like, unlike)]
    for _ in range(a.warmup):
        tr.step(*d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step(*d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"metric": "scorer training steps/sec (train_contrastive body)", "value": 1.0 / dt, "unit": "steps/s",
           "records_per_s": B / dt, "ms_per_step": dt * 1e3, "dtype": "f32",
           "config": {"workload": "B=%d origins x (1 + %d like + %d unlike) latents, %s" % (B, Kp, Kn, "simplest_siamese" if a.simplest else "SiameseNet + Predictor"),
                      "final_loss": float(tr.loss.item())},
           "roofline": {"bound": "mfma" if B >= 2048 else "launch", "achieved": step_flops(B, Kp, Kn, a.simplest) / dt / 1e12,
                        "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": step_flops(B, Kp, Kn, a.simplest) / dt / 1e12 / F32_MFMA_PEAK_TFLOPS}}
    if a.cpu_steps > 0:
        import st_train_oracle as T
        s_np = {k: v.numpy() for k, v in ssd.items()}
        p_np = {k: v.numpy() for k, v in psd.items()}
        masks = rng.random((1 + Kp + Kn, 2, B, 512)) >= 0.2
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            T.scorer_train_loss_and_grads(s_np, p_np, ori, like, unlike, masks, simplest=a.simplest)
        cdt = (time.perf_counter() - t0) / a.cpu_steps
        out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "steps/s", "cores": min(16, os.cpu_count() or 1), "kind": "port",
                               "sample": "%d steps of the numpy restatement (forward + backward, no Adam)" % a.cpu_steps}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
