"""CPU: pins the oracle (oracle/st_oracle.py) to golden vectors recorded from the reference itself
(tests/golden/attack_*.npz, simloss_b128.npz; generator: oracle/make_goldens.py)."""
import os

import numpy as np
import pytest

from conftest import GOLD, gold

import st_oracle as O  # noqa: E402


def test_encoder_latents(ae_sd, gold_notnll):
    g = gold_notnll
    idx = np.r_[0:42, 600:642]
    lat = O.encode_latent(ae_sd, g["enc_src"][idx].astype(np.int64))
    assert np.abs(lat - g["enc_latent"][idx]).max() < 2e-5 * np.abs(g["enc_latent"]).max()


@pytest.mark.parametrize("tag,kw", [("notnll", dict(not_nll=True)), ("nll", dict()), ("nll_one", dict(just_use_one=True)),
                                    ("simplest", dict(not_nll=True, simplest=True)), ("p3n5", dict()), ("p4n4", dict(not_nll=True))])  # (retext goldens: latent jumps on accept, GPU test only)
def test_scorer_loss_grad_adam(scorer_sd, tag, kw):
    path = os.path.join(GOLD, "attack_%s.npz" % tag)
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    g = gold("attack_%s.npz" % tag)
    ssd, psd = scorer_sd
    for s in range(min(6, len(g["subset"]))):
        x = g["init_latent"][s:s + 1]
        opt = O.Adam((1, 256), float(g["lr"][s]))
        for it in range(int(g["n_iter"][s])):
            loss, grad = O.contrastive_loss_and_grad(ssd, psd, x, g["like_latent"][s], g["unlike_latent"][s], 0.1, **kw)
            assert abs(loss - g["loss"][s, it]) < 2e-5 * max(1, abs(g["loss"][s, it]))
            assert np.abs(grad[0] - g["grad"][s, it]).max() < 2e-4 * np.abs(g["grad"][s, it]).max()
            x = opt.step(x, g["grad"][s:s + 1, it])     # reference gradient: isolates the Adam restatement
            assert np.abs(x[0] - g["latent"][s, it]).max() < 1e-5 * np.abs(g["latent"][s, it]).max()


OPT_LR = {"SGD": 200.0, "AdamW": 0.2, "Adamax": 0.2, "RMSprop": 0.05, "Adagrad": 0.5}


@pytest.mark.parametrize("name", list(OPT_LR))
def test_other_victim_optimizers_match_torch(scorer_sd, name):
    """--victim_optimizer: the reference builds getattr(optim, name)([latent], lr=lr) (siamese_contrastive.py:780); goldens are its
    own attack_contrastive runs with each rule (oracle/make_goldens.py --what optimizers).  Fed the reference's gradients, the
    numpy rules must reproduce the reference's latents."""
    g = gold("attack_opt_%s.npz" % name.lower())
    ssd, psd = scorer_sd
    for s in range(len(g["subset"])):
        x = g["init_latent"][s:s + 1]
        assert abs(float(g["lr"][s]) - OPT_LR[name]) < 1e-6
        opt = O.OPTIMIZERS[name]((1, 256), float(g["lr"][s]))
        for it in range(int(g["n_iter"][s])):
            loss, grad = O.contrastive_loss_and_grad(ssd, psd, x, g["like_latent"][s], g["unlike_latent"][s], 0.1, not_nll=True)
            assert abs(loss - g["loss"][s, it]) < 2e-5 * max(1, abs(g["loss"][s, it]))
            x = opt.step(x, g["grad"][s:s + 1, it])
            assert np.abs(x[0] - g["latent"][s, it]).max() < 1e-5 * np.abs(g["latent"][s, it]).max(), (name, s, it)
        assert np.abs(g["latent"][s, int(g["n_iter"][s]) - 1] - g["init_latent"][s]).max() > 1e-3      # the rule really moved it


def test_greedy_decode_kv_cache_matches_reference(ae_sd, gold_notnll):
    g = gold_notnll
    lat = np.concatenate([g["init_latent"][:12], g["latent"][0, :int(g["n_iter"][0])], g["latent"][40, :int(g["n_iter"][40])]])
    want = np.concatenate([g["init_ids"][:12], g["ids"][0, :int(g["n_iter"][0])], g["ids"][40, :int(g["n_iter"][40])]])
    wgap = np.concatenate([g["init_gap"][:12], g["gap"][0, :int(g["n_iter"][0])], g["gap"][40, :int(g["n_iter"][40])]])
    ids, gap = O.greedy_decode(ae_sd, lat, 60, want_gap=True)
    for b in range(len(lat)):
        if not np.array_equal(ids[b], want[b]):
            first = int(np.nonzero(ids[b] != want[b])[0][0])
            assert wgap[b, first] < 1e-3
        else:
            assert np.abs(gap[b] - wgap[b]).max() < 1e-3
    ids_e = O.greedy_decode(ae_sd, lat[:6], 60, early_exit=True)
    for b in range(6):
        e = next(i for i, t in enumerate(want[b]) if t in (0, 3)) if any(t in (0, 3) for t in want[b]) else 58
        assert np.array_equal(ids_e[b, :e + 1], want[b, :e + 1]) and (ids_e[b, e + 1:] == 0).all()


def test_literal_decode_matches_reference_prefix(ae_sd, gold_notnll):
    """model.py:271-299 restated line by line (no KV cache), one sentence, first 12 steps"""
    g = gold_notnll
    ids = O.greedy_decode_literal(ae_sd, g["init_latent"][1:2], max_len=13)
    assert np.array_equal(ids[0], g["init_ids"][1, :12])


def test_attack_one_reproduces_reference_log(ae_sd, scorer_sd, gold_notnll, id_to_word):
    """contrastive_attack_one restated: Change./iteration. lines of two sentences equal the reference log"""
    g = gold_notnll
    ref = open(os.path.join(GOLD, "attack_notnll_log.txt")).read().split("Sentence Pair.")[1:]
    ssd, psd = scorer_sd
    for s in (2, 33):
        lines = []
        O.attack_one(ae_sd, ssd, psd, g["init_latent"][s:s + 1], g["like_latent"][s], g["unlike_latent"][s], float(g["lr"][s]),
                     id_to_word, lines.append, not_nll=True)
        want = [l.split(" - ", 1)[1] for l in ref[s].splitlines()[1:] if " - " in l]
        want = [w for w in want[want.index("-" * 40):] if w != "*" * 20]  # drop the next block's headline prefix
        assert lines == want


def test_simloss_oracle_vs_reference_torch():
    g = gold("simloss_b128.npz")
    tot = 0
    for v in range(3):
        loss, lse = O.simloss(g["ori"], g["like"][v], float(g["tau"]))
        assert np.abs(loss - g["loss_rows"][v]).max() < 1e-4
        tot += loss.mean() / 3
        gw = np.full(128, 1.0 / (3 * 128))
        do, dl = O.simloss_grads(g["ori"], g["like"][v], gw, float(g["tau"]))
        assert np.abs(dl - g["d_like"][v]).max() < 1e-3 * np.abs(g["d_like"][v]).max()
    assert abs(tot - float(g["loss"])) < 1e-5


def _fgim_reference_lines():
    """tests/golden/fgim_log.txt: per sentence the reference's stdout and its returned text."""
    blocks = open(os.path.join(GOLD, "fgim_log.txt")).read().split("=====sentence ")[1:]
    out = []
    for b in blocks:
        body, ret = b.split("=====returned=====\n")
        out.append((body.split("\n", 1)[1].splitlines(), ret.rstrip("\n")))
    return out


def test_fgim_oracle_vs_reference(ae_sd, id_to_word):
    """Legacy FGIM path (model.py:414-433, :454-516): prediction, BCE gradient wrt the latent, and the texts the
    reference prints for its nine step sizes, with a seeded random Classifier (the shipped one is missing)."""
    from safetensors.numpy import load_file
    csd = load_file(os.path.join(GOLD, "fgim_classifier.safetensors"))
    g = gold("fgim.npz")
    pred, grad, out = O.fgim_perturb(csd, g["latent"], g["target"], g["eps"])
    assert np.abs(pred - g["pred"]).max() < 1e-6
    assert np.abs(grad - g["grad"]).max() < 1e-5 * np.abs(g["grad"]).max()
    ref = _fgim_reference_lines()
    for s in range(len(ref)):
        ids = O.greedy_decode(ae_sd, out[:, s], early_exit=True)
        texts = [O.id2text_sentence(ids[k], id_to_word) for k in range(len(g["eps"]))]
        lines, ret = ref[s]
        ref_texts = [lines[i] for i in range(3, len(lines), 3)]        # gold / (epsilon, It, text) x 9
        assert texts == ref_texts and texts[-1] == ret
        assert lines[2] == "| It {:2d} | dis model pred {:5.4f} |".format(5, float(pred[s]))


# ---- scorer training step: numpy restatement vs the reference's torch autograd + Adam ----
@pytest.mark.parametrize("tag,kw", [("nll", {}), ("simplest", dict(simplest=True)), ("nll_one", dict(just_use_one=True))])
def test_scorer_train_step_oracle_vs_reference(tag, kw):
    import train_common as TC
    gd = np.load(os.path.join(GOLD, "scorer_train_%s.npz" % tag))
    steps = TC.oracle_steps(kw, gd, 3, float(gd["lr"]))
    shapes = TC.named(*TC.load_init())
    sl = TC.tensor_slices(shapes)
    idx = np.arange(0, sum(int(np.prod(shapes[k].shape)) for k in TC.NAMES), TC.STRIDE)
    for s, st in enumerate(steps):
        assert abs(st["loss"] - gd["loss"][s]) <= (2e-6 if s == 0 else 2e-4) * abs(gd["loss"][s])
        assert (st["choice"] == gd["choice"][s]).mean() >= (1.0 if s == 0 else 0.995)
        gf = TC.flat(st["grads"], shapes)
        noise = TC.noise_grad_tensors(gd, s)
        for i, k in enumerate(TC.NAMES):
            nrm = np.sqrt((gf[sl[k]].astype(np.float64) ** 2).sum())
            if k in noise:
                assert nrm < 1e-6                      # ours is rounding noise as well
                continue
            # per-tensor norm: every step (later steps see Adam's amplification of ~1e-8 gradients, see train_common)
            assert abs(nrm - gd["grad_norms"][s][i]) <= (1e-5 if s == 0 else 0.1) * gd["grad_norms"][s][i], (s, k)
            if s == 0:
                sel = (idx >= sl[k].start) & (idx < sl[k].stop)
                np.testing.assert_allclose(gf[idx[sel]], gd["grad"][0][sel], rtol=1e-4, atol=2e-7, err_msg=k)
                # one Adam step: +-lr moves; elements with |g| ~ eps = 1e-8 depend on the last bit of g
                pf = st["params"][k].ravel()[idx[sel] - sl[k].start]
                ok = np.abs(pf - gd["param"][0][sel]) <= 1e-6
                assert ok.mean() >= 0.999, (k, ok.mean())
    # BatchNorm statistics after 3 x 21 (resp. 3 x 1) updates; the noise-driven bias drift enters running_mean
    for k in gd.files:
        if k.startswith("bn.") and "tracked" not in k:
            np.testing.assert_allclose(steps[-1]["stats"][k[3:]], gd[k], rtol=0, atol=5e-3, err_msg=k)
        elif k.startswith("bn."):
            assert int(gd[k]) in (0, 3, 63)


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_ae_train_step_oracle_vs_reference(tag, ae_sd):
    """The numpy restatement of the AE constraint-training step (oracle/st_ae_train_oracle.py) against the reference's
    own train_ae_model_constraint run (tests/golden/ae_train_*.npz): both losses, every parameter's gradient (2-norm
    per tensor + every 37th element), with dropout 0 and with the reference's 264 captured nn.Dropout masks replayed
    in call order; parameter order; NoamOpt / Adam's first step."""
    import st_ae_train_oracle as A
    import ae_train_common as AC
    from style_transfer_amd.train_ae import ae_param_names, noam_rate
    g = gold("ae_train_%s.npz" % tag)
    names = [str(n) for n in g["names"]]
    assert names == ae_param_names(2)
    sd = {k: v.astype(np.float64) for k, v in ae_sd.items()}
    origin, views = AC.golden_batch(g, 0)
    masks = AC.reference_masks(g) if tag == "drop" else None
    rec, con, grads = A.train_step_loss_and_grads(sd, origin, views, float(g["bata"]), float(g["tau"]), masks, float(g["dropout"]))
    assert abs(rec - g["rec_loss"][0]) < 1e-5 * abs(g["rec_loss"][0]) and abs(con - g["con_loss"][0]) < 1e-5 * g["con_loss"][0]
    for i, n in enumerate(names):
        ref = float(g["grad_norms"][0][i])
        mine = float(np.linalg.norm(grads[n]))
        if AC.is_noise(n) or AC.is_zero(n):
            assert mine < 1e-6 and ref < 1e-6, (n, mine, ref)      # true gradient 0: the reference holds rounding noise
        else:
            assert abs(mine - ref) < 1e-4 * ref, (n, mine, ref)
    flat = np.concatenate([grads[n].ravel() for n in names])
    assert np.abs(flat[::37] - g["grad"][0]).max() < 5e-5 * np.abs(g["grad"][0]).max()   # the reference is f32
    # NoamOpt: lr(1) = 256^-0.5 * 2000^-1.5; Adam's first step moves every parameter by lr * g / (|g| + eps)
    lr = noam_rate(1)
    assert abs(lr - float(g["lr"][0])) < 1e-12
    p0 = np.concatenate([sd[n].ravel() for n in names])
    p1, _, _ = A.adam_step(p0, flat, np.zeros_like(p0), np.zeros_like(p0), 1, lr)
    sel = np.abs(flat[::37]) > 1e-6
    assert np.abs(p1[::37][sel] - g["param"][0][sel]).max() < 2e-7
