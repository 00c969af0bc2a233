"""Fixture generator (build container only; needs /root/reference).  Runs the UNMODIFIED reference
functions on the surrogate AE checkpoint + the shipped SiameseNet/Predictor weights and records
golden vectors under tests/golden/.

What runs: siamese_contrastive.attack_contrastive (siamese_contrastive.py:672-743) itself, with
  * create_train_data_loader_contrastive wrapped so the three loaders hold a SUBSET of the 1000 test
    sentences (the function walks loader.num_batch sentences; the full set takes ~3 h on 8 cores),
  * print_and_log wrapped to capture the log lines,
  * get_contrastive_loss / ae_model.greedy_decode / ae_model.forward wrapped to RECORD their inputs
    and outputs (latent before/after each Adam step, gradient, loss, decoded ids, top-2 logit gap).
No arithmetic is restated here: every number comes out of the reference's own code.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_ae(ref_model, args, ckpt):
    from safetensors.torch import load_file
    sd = load_file(ckpt)
    ae = ref_model.make_model(args.vocab_size, 2, 256, 256, 1024)
    full = ae.state_dict()
    for k in full:
        if k.endswith(".pe"):
            continue
        src = "src_embed.0.lut.weight" if k == "tgt_embed.0.lut.weight" else k
        full[k] = sd[src].float()
    ae.load_state_dict(full)
    ae.eval()
    return ae


def run_attack(subset, flags, ckpt, tag, threads, compact_to=None):
    """compact_to: path prefix; when given, a COMPACT record (no encoder goldens, no gradients, ids as int16) is
    written there instead of tests/golden/attack_<tag>.npz (the whole-split goldens: yelp_full / amazon_full)."""
    torch.set_num_threads(threads)
    ref_model, ref_data, ref_sc = ref_harness.import_reference()
    flags = dict(flags)
    task = flags.pop("task", "yelp")
    args = ref_harness.make_args(task, **flags)   # (amazon ships no Siamese/Predictor: the yelp ones are used)
    ae = load_ae(ref_model, args, ckpt)
    siamese, predictor = ref_sc.SiameseNet(256), ref_sc.Predictor(256)
    ref_sc.load_siamese(siamese, predictor, args)

    if not hasattr(ref_sc, "_st_originals"):  # wrappers are re-installed per run around the ORIGINAL functions
        ref_sc._st_originals = (ref_sc.create_train_data_loader_contrastive, ref_sc.get_contrastive_loss,
                                ref_sc.contrastive_attack_one)
    (ref_sc.create_train_data_loader_contrastive, ref_sc.get_contrastive_loss,
     ref_sc.contrastive_attack_one) = ref_sc._st_originals

    lines = []
    ref_sc.print_and_log = lambda s: lines.append(s)

    orig_create = ref_sc.create_train_data_loader_contrastive

    def create_subset(*a, **k):
        loaders = orig_create(*a, **k)
        for ld in loaders:
            if ld.src_batches == []:
                continue
            for name in ("src_batches", "tgt_batches", "tgt_y_batches", "src_mask_batches", "tgt_mask_batches", "ntokens_batches"):
                full = getattr(ld, name)
                setattr(ld, name, [full[i] for i in subset])
            ld.num_batch = len(subset)
        return loaders

    ref_sc.create_train_data_loader_contrastive = create_subset

    rec = dict(enc_src=[], enc_latent=[], sent=[])
    cur = {}

    orig_forward = ae.forward

    def forward(src, tgt, src_mask, tgt_mask):
        latent, prob = orig_forward(src, tgt, src_mask, tgt_mask)
        if src is not None:
            rec["enc_src"].append(src.numpy().copy())
            rec["enc_latent"].append(latent.detach().numpy().copy())
        return latent, prob

    ae.forward = forward

    gaps_step = []

    def gen_hook(mod, inp, out):
        if out.dim() == 2:
            top = out.detach().topk(2, dim=-1).values
            gaps_step.append((top[:, 0] - top[:, 1]).numpy().copy())

    ae.generator.register_forward_hook(gen_hook)

    orig_decode = ae.greedy_decode
    decodes = []

    def greedy_decode(latent, max_len, start_id):
        gaps_step.clear()
        ys = orig_decode(latent, max_len, start_id)
        if latent is not None:
            decodes.append(dict(latent=latent.detach().numpy().copy(), ids=ys.numpy().copy(),
                                gap=np.stack(gaps_step, 1).copy(),
                                grad=None if latent.grad is None else latent.grad.numpy().copy()))
        return ys

    ae.greedy_decode = greedy_decode

    orig_loss = ref_sc.get_contrastive_loss
    losses = []

    def get_loss(*a, **k):
        out = orig_loss(*a, **k)
        losses.append(float(out[0].item()))
        return out

    ref_sc.get_contrastive_loss = get_loss

    orig_attack_one = ref_sc.contrastive_attack_one
    per_sentence = []

    def attack_one(sn, pn, aem, ori_latent, like_latent, unlike_latent, a):
        decodes.clear()
        losses.clear()
        orig_attack_one(sn, pn, aem, ori_latent, like_latent, unlike_latent, a)
        per_sentence.append(dict(
            like_latent=None if like_latent is None else like_latent.numpy().copy(),
            unlike_latent=None if unlike_latent is None else unlike_latent.numpy().copy(),
            decodes=[dict(d) for d in decodes], losses=list(losses), lr=a.victim_lr))

    ref_sc.contrastive_attack_one = attack_one

    t0 = time.time()
    ref_sc.attack_contrastive(siamese, predictor, ae, args)
    dt = time.time() - t0
    print("[%s] reference attack_contrastive on %d sentences: %.1f s (%.2f s/sentence, %d threads)" %
          (tag, len(subset), dt, dt / len(subset), threads), flush=True)

    S, I, T = len(subset), args.attack_max_iter, args.max_sequence_length - 1
    n_iter = np.zeros(S, np.int32)
    init_latent = np.zeros((S, 256), np.float32)
    init_ids = np.zeros((S, T), np.int32)
    init_gap = np.zeros((S, T), np.float32)
    lat = np.zeros((S, I, 256), np.float32)
    grad = np.zeros((S, I, 256), np.float32)
    ids = np.zeros((S, I, T), np.int32)
    gap = np.zeros((S, I, T), np.float32)
    loss = np.zeros((S, I), np.float32)
    re_ids = np.zeros((S, I, T), np.int32)
    re_latent = np.zeros((S, I, 256), np.float32)
    kp = per_sentence[0]["like_latent"].shape[0] if per_sentence[0]["like_latent"] is not None else 0
    kn = per_sentence[0]["unlike_latent"].shape[0] if per_sentence[0]["unlike_latent"] is not None else 0
    like = np.zeros((S, kp, 256), np.float32)
    unlike = np.zeros((S, kn, 256), np.float32)
    lr = np.zeros(S, np.float32)
    for s, ps in enumerate(per_sentence):
        d0 = ps["decodes"][0]
        init_latent[s], init_ids[s], init_gap[s] = d0["latent"][0], d0["ids"][0], d0["gap"][0]
        main = ps["decodes"][1:]
        if flags.get("retext"):
            # retext() decodes the re-encoded text right after every main decode (siamese_contrastive.py:849)
            re_dec, main = main[1::2], main[0::2]
            for it, d in enumerate(re_dec):
                re_ids[s, it], re_latent[s, it] = d["ids"][0], d["latent"][0]
        n_iter[s] = len(main)
        for it, d in enumerate(main):
            lat[s, it], ids[s, it], gap[s, it], grad[s, it] = d["latent"][0], d["ids"][0], d["gap"][0], d["grad"][0]
            loss[s, it] = ps["losses"][it]
        if kp:
            like[s] = ps["like_latent"][:, 0]
        if kn:
            unlike[s] = ps["unlike_latent"][:, 0]
        lr[s] = ps["lr"]
    if compact_to is not None:
        np.savez_compressed(compact_to + ".npz", subset=np.asarray(subset, np.int32), n_iter=n_iter,
                            init_latent=init_latent, init_ids=init_ids.astype(np.int16), init_gap=init_gap,
                            latent=lat, ids=ids.astype(np.int16), gap=gap.astype(np.float16), gap_min=gap.min(-1),
                            loss=loss, lr=lr, ref_seconds=np.float32(dt), ref_threads=np.int32(threads))
        with open(compact_to + "_log.txt", "w") as f:
            for ln in lines:
                f.write("INFO:siamese_contrastive.py:print_and_log:1592 - " + ln + "\n")
        return dt
    # encoder goldens: every ae_model.forward call in order (1 + kp + kn per sentence)
    Lmax = max(x.shape[1] for x in rec["enc_src"])
    enc_src = np.zeros((len(rec["enc_src"]), Lmax), np.int32)
    for i, x in enumerate(rec["enc_src"]):
        enc_src[i, : x.shape[1]] = x[0]
    enc_latent = np.concatenate(rec["enc_latent"], 0).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "attack_%s.npz" % tag), subset=np.asarray(subset, np.int32), n_iter=n_iter,
                        init_latent=init_latent, init_ids=init_ids, init_gap=init_gap, latent=lat, grad=grad, ids=ids,
                        gap=gap, loss=loss, like_latent=like, unlike_latent=unlike, lr=lr, enc_src=enc_src,
                        enc_latent=enc_latent, re_ids=re_ids, re_latent=re_latent, ref_seconds=np.float32(dt), ref_threads=np.int32(threads))
    with open(os.path.join(GOLD, "attack_%s_log.txt" % tag), "w") as f:
        for ln in lines:
            f.write("INFO:siamese_contrastive.py:print_and_log:1592 - " + ln + "\n")
    return dt


def run_simloss():
    """The B x B in-batch InfoNCE block of train_ae_model_constraint (siamese_contrastive.py:1060-1069),
    same torch calls, B = 128 (main.py:67), 3 views, with autograd gradients."""
    import torch.nn.functional as F
    torch.manual_seed(7)
    B, tau = 128, 0.1
    ori = (torch.rand(B, 256) * 10).requires_grad_(True)
    like = (torch.rand(3, B, 256) * 10).requires_grad_(True)
    crit = torch.nn.NLLLoss()
    losses, per_row = [], []
    for i in range(3):
        sim = F.cosine_similarity(ori.unsqueeze(1), like[i], dim=-1) / tau
        targets = torch.tensor(list(range(B)))
        sim_softmax = F.log_softmax(sim, dim=-1)
        losses.append(crit(sim_softmax, targets))
        per_row.append(-sim_softmax[torch.arange(B), targets].detach().numpy())
    total = sum(losses) / len(losses)
    total.backward()
    np.savez_compressed(os.path.join(GOLD, "simloss_b128.npz"), ori=ori.detach().numpy(), like=like.detach().numpy(),
                        loss_rows=np.stack(per_row), loss=np.float32(total.item()), d_ori=ori.grad.numpy(),
                        d_like=like.grad.numpy(), tau=np.float32(tau))
    print("simloss golden written, loss", total.item())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="main")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--n", type=int, default=32, help="sentences per label for the main run")
    ap.add_argument("--shard", type=int, default=0)
    ap.add_argument("--nshards", type=int, default=1)
    ap.add_argument("--scratch", default="/tmp/st_full_goldens")
    ap.add_argument("--count", type=int, default=1000, help="sentences of the split to run (amazon_full: first n/2 of each label)")
    a = ap.parse_args()
    ckpt = os.path.join(GOLD, "surrogate_ae_yelp.safetensors")
    if a.what == "simloss":
        run_simloss()
    elif a.what in ("yelp_full", "amazon_full"):
        # The WHOLE test split (yelp: all 1000; amazon: --count of them), sharded over single-thread processes:
        #   for k in 0..7: python oracle/make_goldens.py --what yelp_full --shard k --nshards 8 --threads 1 &
        #   python oracle/make_goldens.py --what yelp_full --nshards 8 --shard -1      (merge)
        task = a.what.split("_")[0]
        if task == "amazon":
            ckpt = os.path.join(GOLD, "surrogate_ae_amazon.safetensors")
        half = a.count // 2
        everything = list(range(half)) + list(range(500, 500 + half))
        os.makedirs(a.scratch, exist_ok=True)
        if a.shard >= 0:
            mine = everything[a.shard::a.nshards]   # strided: every shard gets both labels and equal work
            run_attack(mine, dict(task=task, not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2), ckpt,
                       "%s_s%d" % (a.what, a.shard), a.threads, compact_to=os.path.join(a.scratch, "%s_s%d" % (a.what, a.shard)))
        else:
            parts = [np.load(os.path.join(a.scratch, "%s_s%d.npz" % (a.what, k))) for k in range(a.nshards)]
            subset = np.concatenate([p["subset"] for p in parts])
            order = np.argsort(subset, kind="stable")
            assert np.array_equal(subset[order], np.asarray(everything))
            out = {k: np.concatenate([p[k] for p in parts])[order] for k in parts[0].files if parts[0][k].ndim >= 1}
            out["ref_seconds"] = np.float32(sum(float(p["ref_seconds"]) for p in parts))
            out["ref_threads"] = np.int32(1)
            np.savez_compressed(os.path.join(GOLD, "attack_%s.npz" % a.what), **out)
            # log blocks in global order: a block starts at its 'Sentence Pair.' line; the run header lines (before the
            # first block) are taken from shard 0
            blocks, header = {}, None
            for k in range(a.nshards):
                text = open(os.path.join(a.scratch, "%s_s%d_log.txt" % (a.what, k))).read().split("\n")
                starts = [i for i, ln in enumerate(text) if "Sentence Pair." in ln]
                if header is None:
                    header = text[: starts[0]]
                ids_k = parts[k]["subset"]
                assert len(starts) == len(ids_k)
                for j, st in enumerate(starts):
                    en = starts[j + 1] if j + 1 < len(starts) else len(text)
                    blk = [ln for ln in text[st:en] if ln != ""]
                    # (the shard numbered its sentences 0..; the global number is restored when writing)
                    blocks[int(ids_k[j])] = (j, blk)
            with open(os.path.join(GOLD, "attack_%s_log.txt" % a.what), "w") as f:
                f.write("\n".join(header) + "\n")
                for pos, g in enumerate(everything):
                    j, blk = blocks[g]
                    blk = list(blk)
                    blk[0] = blk[0].replace("Sentence Pair.%d*" % j, "Sentence Pair.%d*" % pos)
                    f.write("\n".join(blk) + "\n")
            print("merged %d sentences; reference time %.0f thread-seconds" % (len(everything), out["ref_seconds"]))
    elif a.what == "main":
        subset = list(range(a.n)) + list(range(500, 500 + a.n))
        run_attack(subset, dict(not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2), ckpt, "notnll", a.threads)
    elif a.what == "amazon":
        ckpt = os.path.join(GOLD, "surrogate_ae_amazon.safetensors")
        sub = list(range(8)) + list(range(500, 508))
        run_attack(sub, dict(task="amazon", not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2), ckpt, "amazon", a.threads)
    elif a.what == "retext":
        run_attack([0, 1, 2, 3, 500, 501, 502, 503], dict(not_nll=True, retext=True, pos_victim_lr=0.2, neg_victim_lr=0.2), ckpt, "retext", a.threads)
    elif a.what == "dance":
        # --dance: two iterations, no Change lines, 'Siamese Loss score' / 'Siamese attack tensors' lines (:792-793, :833-835)
        sub = [0, 1, 2, 3, 500, 501, 502, 503]
        run_attack(sub, dict(dance=True, not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2), ckpt, "dance_notnll", a.threads)
        run_attack(sub, dict(dance=True), ckpt, "dance_nll", a.threads)
    elif a.what == "optimizers":
        # --victim_optimizer: the reference takes any torch.optim name with its defaults (siamese_contrastive.py:780).
        # Learning rates chosen so that each rule moves the latent visibly within 18 iterations.
        sub = [0, 1, 500, 501]
        for name, lr in (("SGD", 200.0), ("AdamW", 0.2), ("Adamax", 0.2), ("RMSprop", 0.05), ("Adagrad", 0.5)):
            run_attack(sub, dict(not_nll=True, victim_optimizer=name, pos_victim_lr=lr, neg_victim_lr=lr), ckpt, "opt_" + name.lower(), a.threads)
    elif a.what == "optimizers2":
        # the remaining first-order torch.optim rules (NAdam, RAdam, Adadelta, ASGD, Rprop), torch defaults apart from lr
        sub = [0, 1, 500, 501]
        for name, lr in (("NAdam", 0.2), ("RAdam", 0.2), ("Adadelta", 50.0), ("ASGD", 200.0), ("Rprop", 0.05)):
            run_attack(sub, dict(not_nll=True, victim_optimizer=name, pos_victim_lr=lr, neg_victim_lr=lr), ckpt, "opt_" + name.lower(), a.threads)
    elif a.what == "variants":
        sub = [0, 1, 2, 500, 501, 502]
        if not os.path.exists(os.path.join(GOLD, "attack_nll.npz")):
            run_attack(sub, dict(), ckpt, "nll", a.threads)
        if not os.path.exists(os.path.join(GOLD, "attack_nll_one.npz")):
            run_attack(sub, dict(just_use_one=True), ckpt, "nll_one", a.threads)
        if not os.path.exists(os.path.join(GOLD, "attack_simplest.npz")):
            run_attack(sub, dict(not_nll=True, simplest_siamese=True, pos_victim_lr=0.2, neg_victim_lr=0.3), ckpt, "simplest", a.threads)
        # (--not_nll with pos_num != neg_num raises in the reference: pos_loss [K+] + neg_loss [K-], :429)
        run_attack(sub, dict(pos_num=3, neg_num=5), ckpt, "p3n5", a.threads)
        run_attack(sub, dict(not_nll=True, pos_num=4, neg_num=4, pos_victim_lr=0.2, neg_victim_lr=0.2), ckpt, "p4n4", a.threads)
