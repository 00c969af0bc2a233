#!/usr/bin/env python3
"""BASELINE config 2's regime: the 1000 real yelp test sentences in batches of --batch (default 256) through
AttackRunner.run_prepared, timed on the host.  Under rocprofv3 --kernel-trace --stats this gives the per-kernel picture
of the latency-bound small-batch path (tools/profile_small.sh)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--log-decodes", action="store_true")
    ap.add_argument("--ipd", default="12,6")
    ap.add_argument("--scorer-format", default="f16x3")
    ap.add_argument("--single-max", type=int, default=None, help="AttackConfig.single_round_max_batch")
    ap.add_argument("--n", type=int, default=1000, help="sentences in all (the 1000 real ones tiled)")
    a = ap.parse_args()
    from style_transfer_amd import weights as W
    from style_transfer_amd.attack import AttackConfig, AttackRunner
    from style_transfer_amd.data import AttackSet
    dev = torch.device("cuda", 0)
    ae = W.AEWeights(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_yelp.safetensors")), dev)
    scorer = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                             W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev, gemm_format=a.scorer_format)
    data = AttackSet.from_json([os.path.join(GOLD, "yelp", "neg10_pos10_test%d.json" % i) for i in (0, 1)], ae.vocab)
    if a.n != 1000:
        data = data.subset([i % 1000 for i in range(a.n)])
    ipd = tuple(int(v) for v in a.ipd.split(","))
    cfg = AttackConfig(not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2, log_decodes=a.log_decodes, iters_per_decode=ipd)
    if a.single_max is not None:
        cfg.single_round_max_batch = a.single_max
    runner = AttackRunner(ae, scorer, cfg, dev)
    spans = [(b0, min(len(data), b0 + a.batch)) for b0 in range(0, len(data), a.batch)]
    preps = [runner.prepare(data.span(b0, b1)) for b0, b1 in spans]
    gidx = [torch.arange(b0, b1, device=dev) for b0, b1 in spans]
    for rep in range(a.reps + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        setup = loop = 0.0
        for p, g in zip(preps, gidx):
            r = runner.run_prepared(p, g, fetch=False)
            setup += r.timing["setup_s"]
            loop += r.timing["loop_s"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if rep:
            print("batch %d: %d sentences in %.1f ms = %.0f sentences/s (setup %.1f ms, loop %.1f ms)" %
                  (a.batch, len(data), dt * 1e3, len(data) / dt, setup * 1e3, loop * 1e3), flush=True)


if __name__ == "__main__":
    main()
