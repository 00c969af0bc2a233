"""TEST INFRASTRUCTURE ONLY — golden vectors for the legacy FGIM path.

Runs the UNMODIFIED reference `fgim_attack` (model.py:454-516) and `Classifier` (model.py:414-433) on the
surrogate yelp AE for a few latents of the attack goldens.  The reference's classifier checkpoint
(dis_model_params.pkl) is not shipped, so a seeded random Classifier is used and stored with the golden
(tests/golden/fgim_classifier.safetensors); its stdout (`epsilon:` / `| It .. | dis model pred .. |` / text
lines) is captured verbatim.

  python oracle/make_fgim_golden.py [--threads 6]
"""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
from make_goldens import load_ae, GOLD  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ref_model, ref_data, _ = ref_harness.import_reference()
    args = ref_harness.make_args(retext=False)
    ae = load_ae(ref_model, args, os.path.join(GOLD, "surrogate_ae_yelp.safetensors"))
    ae.eval()
    torch.manual_seed(3)
    cls = ref_model.Classifier(256, 1)
    with torch.no_grad():                      # spread the predictions: a fresh init predicts ~0.5 everywhere
        cls.fc3.weight.mul_(8.0)
        cls.fc1.weight.mul_(0.5)
    cls.eval()
    g = np.load(os.path.join(GOLD, "attack_notnll.npz"))
    pick = [0, 1, 2, 3, 32, 33, 34, 35]
    lat = g["init_latent"][pick]
    targets = np.array([1.0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0], np.float32)
    eps = [0.01, 0.012, 0.014, 0.015, 0.016, 0.017, 0.018, 0.019, 0.02]   # model.py:467
    logs, last = [], []
    for s in range(len(pick)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            txt = ref_model.fgim_attack(cls, torch.from_numpy(lat[s:s + 1].copy()), torch.tensor([[float(targets[s])]]), ae,
                                        args.max_sequence_length, args.id_bos, ref_data.id2text_sentence, args.id_to_word,
                                        g["init_ids"][pick[s]].tolist(), args)
        logs.append(buf.getvalue())
        last.append(txt)
    # direct values for the numeric checks: prediction, gradient of BCE wrt the latent
    x = torch.from_numpy(lat.copy()).requires_grad_(True)
    out = cls(x)
    grads = []
    for s in range(len(pick)):
        xs = torch.from_numpy(lat[s:s + 1].copy()).requires_grad_(True)
        loss = torch.nn.BCELoss()(cls(xs), torch.tensor([[float(targets[s])]]))
        loss.backward()
        grads.append(xs.grad.numpy()[0])
    from safetensors.numpy import save_file
    save_file({k: v.detach().numpy() for k, v in cls.state_dict().items()}, os.path.join(GOLD, "fgim_classifier.safetensors"))
    np.savez_compressed(os.path.join(GOLD, "fgim.npz"), latent=lat, target=targets, eps=np.asarray(eps, np.float32),
                        pred=out.detach().numpy()[:, 0], grad=np.stack(grads))
    with open(os.path.join(GOLD, "fgim_log.txt"), "w") as f:
        for s, (lg, tx) in enumerate(zip(logs, last)):
            f.write("=====sentence %d=====\n" % s)
            f.write(lg)
            f.write("=====returned=====\n%s\n" % tx)
    print("fgim golden written: preds", out.detach().numpy()[:, 0])


if __name__ == "__main__":
    main()
