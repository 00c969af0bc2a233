"""TEST INFRASTRUCTURE ONLY — golden vectors for the AE constraint-training step (second half).

Runs the UNMODIFIED reference function train_ae_model_constraint (siamese_contrastive.py:1000-1088) — its own data
loaders (create_ae_data_loader :950-997, AttackDataLoader data.py:298-505, shuffle seed 234), EncoderDecoder.forward,
batch_forward, LabelSmoothing, NoamOpt + Adam — for one epoch of TWO batches of 32 records on the surrogate AE
checkpoint.  The records are the first 32 of each yelp test JSON (origin + sim0..9), written to a temporary data
directory under the file names the function reads (ae_train_constraint{0,1}.json).  Nothing is restated: NoamOpt.step
and Accumulator.inc are wrapped only to RECORD gradients, parameters and the two losses of every step.

Two runs:  dropout 0 (make_model(..., dropout=0.0): every number is deterministic), and the reference's dropout 0.1
with every nn.Dropout keep-mask captured by forward hooks in call order (replayed by the restatement).

  python oracle/make_ae_train_golden.py [--threads 6]   ->  tests/golden/ae_train_{nodrop,drop}.npz
Gradients / parameters are stored as per-tensor 2-norms, every 37th element of the flat vector (parameters() order)
and in full for the small tensors.
"""
import argparse
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
from make_goldens import GOLD, load_ae  # noqa: E402

STRIDE = 37
PER_FILE = 32
BATCH = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ref_model, ref_data, sc = ref_harness.import_reference()
    ckpt = os.path.join(GOLD, "surrogate_ae_yelp.safetensors")
    src_dir = os.path.join(GOLD, "yelp")
    for tag, drop in (("nodrop", 0.0), ("drop", 0.1)):
        with tempfile.TemporaryDirectory() as td:
            for i in (0, 1):
                d = json.load(open(os.path.join(src_dir, "neg10_pos10_test%d.json" % i)))
                d["data"] = d["data"][:PER_FILE]
                d["sent_num"] = PER_FILE
                json.dump(d, open(os.path.join(td, "ae_train_constraint%d.json" % i), "w"))
            args = ref_harness.make_args(batch_size=BATCH, data_path=td + "/", current_save_path=td + "/")
            ae = load_ae(ref_model, args, ckpt)
            for m in ae.modules():                       # make_model(..., dropout=p): the same modules, p set afterwards
                if isinstance(m, torch.nn.Dropout):
                    m.p = drop
            named = dict(ae.named_parameters())
            names = list(named)
            rec = dict(rec_loss=[], con_loss=[], grad=[], grad_norms=[], param=[], param_norms=[], small_grads=[], lr=[])
            masks = []
            hooks = []
            if drop > 0:
                for m in ae.modules():
                    if isinstance(m, torch.nn.Dropout):
                        hooks.append(m.register_forward_hook(
                            lambda mod, inp, out: masks.append(np.packbits((out != 0).numpy().ravel())) if mod.training else None))
            small = [n for n in names if named[n].numel() <= 1024]
            orig_step = ref_model.NoamOpt.step
            mask_steps = []

            def step(self):
                gz = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in named.items()}
                rec["grad"].append(np.concatenate([gz[k].numpy().ravel() for k in names])[::STRIDE].copy())
                rec["grad_norms"].append(np.array([float(gz[k].double().norm()) for k in names], np.float64))
                rec["small_grads"].append(np.concatenate([gz[k].numpy().ravel() for k in small]).copy())
                orig_step(self)
                rec["lr"].append(self._rate)
                rec["param"].append(np.concatenate([named[k].detach().numpy().ravel() for k in names])[::STRIDE].copy())
                rec["param_norms"].append(np.array([float(named[k].detach().double().norm()) for k in names], np.float64))
                mask_steps.append(list(masks))
                masks.clear()

            orig_inc = sc.Accumulator.inc

            def inc(self, v=1, n=1):
                (rec["rec_loss"] if self.name == "rec loss" else rec["con_loss"]).append(float(v))
                orig_inc(self, v, n)

            ref_model.NoamOpt.step = step
            sc.NoamOpt.step = step
            sc.Accumulator.inc = inc
            sc.print_and_log = lambda s: None
            torch.manual_seed(7)
            try:
                sc.train_ae_model_constraint(ae, args, 1)
            finally:
                ref_model.NoamOpt.step = orig_step
                sc.NoamOpt.step = orig_step
                sc.Accumulator.inc = orig_inc
                for h in hooks:
                    h.remove()
            # the order in which the (shuffled) loaders served the 24 records: random.seed(234); shuffle (data.py:403-408)
            order = list(range(2 * PER_FILE))
            random.seed(234)
            random.shuffle(order)
            out = {k: np.stack(v) for k, v in rec.items()}
            out["order"] = np.asarray(order, np.int32)       # record k of the run = record order[k] of [file0[:32] + file1[:32]]
            out["names"] = np.array(names)
            out["small_names"] = np.array(small)
            out["sizes"] = np.array([named[n].numel() for n in names], np.int64)
            out["dropout"] = np.float32(drop)
            out["bata"], out["tau"] = np.float32(args.bata), np.float32(args.tau)
            if drop > 0:
                # per step: the keep-masks of every nn.Dropout call in call order, bit-packed and concatenated; sizes beside
                for s, ms in enumerate(mask_steps[:1]):      # (step 0 only: 5 MB of bits per step)
                    out["mask_sizes_%d" % s] = np.array([m.size for m in ms], np.int64)
                    out["masks_%d" % s] = np.concatenate(ms)
            np.savez_compressed(os.path.join(GOLD, "ae_train_%s.npz" % tag), **out)
            print(tag, "rec loss", rec["rec_loss"], "constraint", rec["con_loss"], "lr", rec["lr"],
                  "dropout calls per step", [len(m) for m in mask_steps])


if __name__ == "__main__":
    main()
