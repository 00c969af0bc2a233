"""TEST INFRASTRUCTURE ONLY — golden vectors for the scorer training step (first half).

Runs the UNMODIFIED reference SiameseNet / Predictor / get_contrastive_loss (siamese_contrastive.py:29-86,367-453)
in train mode with torch.optim.Adam exactly as train_contrastive does (:871-883, :918-931) for three steps on the
64 golden latents (ori [64,256], like / unlike [10,64,256]) and records what each step produced.  Dropout keep-masks
are captured with forward hooks on the two nn.Dropout modules so the restatement can replay them.

  python oracle/make_train_golden.py [--threads 6]

Writes tests/golden/scorer_train_init_{siamese,predictor}.safetensors (the seeded initial weights) and
tests/golden/scorer_train_{nll,simplest,nll_one}.npz.  Gradients / parameters are stored as every 7th element of
the flat parameter vector (order: oracle/st_train_oracle.PARAM_ORDER) plus per-tensor 2-norms.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
from make_goldens import GOLD  # noqa: E402
from st_train_oracle import PARAM_ORDER  # noqa: E402

STRIDE = 7
STEPS = 3
LR = 3e-3  # main.py:82


def flat(named, what):
    return np.concatenate([what(named[k]).detach().numpy().ravel() for k in named])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    _, _, sc = ref_harness.import_reference()
    g = np.load(os.path.join(GOLD, "attack_notnll.npz"))
    ori = torch.from_numpy(g["init_latent"].copy())
    like = torch.from_numpy(g["like_latent"].transpose(1, 0, 2).copy())
    unlike = torch.from_numpy(g["unlike_latent"].transpose(1, 0, 2).copy())
    from safetensors.numpy import save_file
    torch.manual_seed(11)
    s0 = sc.SiameseNet(256)
    p0 = sc.Predictor(256)
    save_file({k: v.numpy() for k, v in s0.state_dict().items()}, os.path.join(GOLD, "scorer_train_init_siamese.safetensors"))
    save_file({k: v.numpy() for k, v in p0.state_dict().items()}, os.path.join(GOLD, "scorer_train_init_predictor.safetensors"))
    for tag, over in (("nll", {}), ("simplest", dict(simplest_siamese=True)), ("nll_one", dict(just_use_one=True))):
        args = ref_harness.make_args(**over)
        net, pred = sc.SiameseNet(256), sc.Predictor(256)
        net.load_state_dict(s0.state_dict())
        pred.load_state_dict(p0.state_dict())
        net.train()
        pred.train()
        params = list(net.parameters()) if args.simplest_siamese else list(net.parameters()) + list(pred.parameters())
        opt = torch.optim.Adam(params, lr=LR)
        named = {"s." + k: v for k, v in net.named_parameters()}
        named.update({"p." + k: v for k, v in pred.named_parameters()})
        assert list(named) == ["%s.%s" % kv for kv in PARAM_ORDER]
        masks = []
        hooks = [m.register_forward_hook(lambda mod, inp, out: masks.append((out != 0).numpy())) for m in (net.d1, net.d2)]
        crit = torch.nn.NLLLoss()
        torch.manual_seed(5)
        rec = dict(loss=[], choice=[], masks=[], grad=[], grad_norms=[], param=[], param_norms=[])
        for step in range(STEPS):
            masks.clear()
            loss, choice = sc.get_contrastive_loss(net, pred, crit, ori, like, unlike, args)
            opt.zero_grad()
            loss.backward()
            opt.step()
            mk = np.stack(masks).reshape(21, 2, 64, 512)             # calls alternate d1, d2 per SiameseNet call
            rec["masks"].append(np.packbits(mk, axis=-1))
            rec["loss"].append(loss.item())
            rec["choice"].append(np.asarray(choice, np.int32))
            gz = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in named.items()}
            rec["grad"].append(np.concatenate([gz[k].numpy().ravel() for k in named])[::STRIDE])
            rec["grad_norms"].append(np.array([float(gz[k].norm()) for k in named], np.float32))
            rec["param"].append(np.concatenate([named[k].detach().numpy().ravel() for k in named])[::STRIDE])
            rec["param_norms"].append(np.array([float(named[k].detach().norm()) for k in named], np.float32))
        for h in hooks:
            h.remove()
        out = {k: np.stack(v) for k, v in rec.items()}
        for k, v in net.state_dict().items():
            if "running" in k or "tracked" in k:
                out["bn.s." + k] = v.numpy()
        for k, v in pred.state_dict().items():
            if "running" in k or "tracked" in k:
                out["bn.p." + k] = v.numpy()
        np.savez_compressed(os.path.join(GOLD, "scorer_train_%s.npz" % tag), lr=np.float32(LR), **out)
        print(tag, "losses", rec["loss"], "acc", [float((c == 0).mean()) for c in rec["choice"]])


if __name__ == "__main__":
    main()
