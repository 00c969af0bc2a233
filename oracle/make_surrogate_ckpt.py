"""Fixture generator (runs only in the build container, needs /root/reference).

The reference's autoencoder checkpoint save/origin/constraint_ae_model0_params.pkl is not shipped
(.MISSING_LARGE_BLOBS:1), so parity "on the same checkpoint" is defined on a SURROGATE checkpoint:
the reference's own make_model (model.py:302-324) trained here for reconstruction with the
reference's LabelSmoothing (model.py:389-411) + NoamOpt (model.py:355-381) on the sentences found
in the two shipped test JSONs (origin + ref* + sim*), following train_ae_model_constraint's
optimiser settings (siamese_contrastive.py:1009-1012) without the constraint term.

Output: tests/golden/surrogate_ae_<task>.safetensors  (unique tensors only; the three `pe` buffers
are deterministic — model.py:77-89 — and are regenerated at load time; src/tgt embeddings are one
shared tensor, model.py:308-313).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="yelp")
    ap.add_argument("--steps", type=int, default=2500)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default=None)
    ap.add_argument("--fp16-round", action="store_true",
                    help="round the trained weights to fp16-representable values and store them as fp16 (half the "
                         "file; the checkpoint IS the rounded weights, loaded back as exact f32 by every consumer)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ref_model, ref_data, _ = ref_harness.import_reference()
    args = ref_harness.make_args(a.task)
    V = args.vocab_size

    sents = []
    for f in ("neg10_pos10_test0.json", "neg10_pos10_test1.json"):
        d = json.load(open(args.data_path + f))
        for rec in d["data"]:
            for k, v in rec.items():
                if k == "origin" or k.startswith("ref") or k.startswith("sim"):
                    sents.append(tuple(int(x) for x in v.strip().split()))
    sents = sorted(set(s for s in sents if len(s) > 0))
    print("unique sentences:", len(sents), "V =", V, flush=True)

    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    model = ref_model.make_model(V, 2, 256, 256, 1024)
    model.train()
    opt = ref_model.NoamOpt(256, 1, 2000, torch.optim.Adam(model.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9))
    crit = ref_model.LabelSmoothing(size=V, padding_idx=0, smoothing=0.1)
    loader = ref_data.AttackDataLoader(a.batch, 2, 3, 1, 60, V)
    t0 = time.time()
    for step in range(a.steps):
        idx = rng.choice(len(sents), a.batch, replace=False)
        lines = [list(sents[i]) for i in idx]
        src, tgt, tgt_y, _, _ = ref_data.pad_batch_seuqences(lines, 2, 3, 1, 60, V)
        src = torch.tensor(src, dtype=torch.long)
        tgt = torch.tensor(tgt, dtype=torch.long)
        tgt_y = torch.tensor(tgt_y, dtype=torch.long)
        src_mask = (src != 0).unsqueeze(-2)
        tgt_mask = loader.make_std_mask(tgt, 0)
        ntok = (tgt_y != 0).sum().float()
        _, out = model.forward(src, tgt, src_mask, tgt_mask)
        loss = crit(out.contiguous().view(-1, out.size(-1)), tgt_y.contiguous().view(-1)) / ntok
        opt.optimizer.zero_grad()
        loss.backward()
        opt.step()
        if step % 50 == 0 or step == a.steps - 1:
            print("step %d loss %.4f  %.1fs" % (step, loss.item(), time.time() - t0), flush=True)
    model.eval()
    sd = model.state_dict()
    keep = {}
    for k, v in sd.items():
        if k.endswith(".pe"):
            continue
        if k == "tgt_embed.0.lut.weight":
            assert torch.equal(v, sd["src_embed.0.lut.weight"])
            continue
        keep[k] = v.detach().clone().contiguous()
        if a.fp16_round:
            keep[k] = keep[k].half()
    from safetensors.torch import save_file
    out = a.out or os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "surrogate_ae_%s.safetensors" % a.task)
    save_file(keep, out, metadata={"task": a.task, "steps": str(a.steps), "V": str(V)})
    print("saved", out, sum(v.numel() for v in keep.values()), "params")


if __name__ == "__main__":
    main()
