"""GPU parity tests, kernel by kernel, through the C-ABI (style_transfer_amd.ops -> libst_kernels.so),
against the CPU oracle (oracle/st_oracle.py) on the same seeded inputs.

Tolerances (floating point; the reference computes in f32):
  * matrix products / activations: 2e-4 relative to the row's max magnitude (f32 accumulation order)
  * argmax ids: exact wherever the oracle's top-2 logit gap exceeds 1e-3 (near-ties are reported)
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, gold

pytestmark = pytest.mark.gpu

import st_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda", 0)


FORMATS = ["f16x3", "f32"]  # every GEMM-carrying kernel is checked in both operand formats


@pytest.fixture(scope="module", params=FORMATS)
def ae(request, ae_sd_torch, dev):
    from style_transfer_amd.weights import AEWeights
    return AEWeights(ae_sd_torch, dev, gemm_format=request.param)


@pytest.fixture(scope="module", params=FORMATS)
def scorer(request, scorer_sd_torch, dev):
    from style_transfer_amd.weights import ScorerWeights
    return ScorerWeights(scorer_sd_torch[0], scorer_sd_torch[1], dev, gemm_format=request.param)


def some_sentences(n, seed=0):
    d = json.load(open(os.path.join(GOLD, "yelp", "neg10_pos10_test0.json")))["data"]
    rng = np.random.default_rng(seed)
    out = []
    for i in rng.choice(len(d), n, replace=False):
        key = ["origin", "ref3", "sim7"][int(i) % 3]
        out.append([int(x) for x in d[int(i)][key].split()])
    return out


def relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


@pytest.mark.parametrize("fmt", FORMATS)
# V % 32 == 0 with the f16x3 format: rp_argmax16_kernel (the decoder's exact-argmax kernel), otherwise rp_gemm_kernel<EPI_ARGMAX>
@pytest.mark.parametrize("n,V,ln", [(1, 9339, True), (63, 9339, False), (65, 19724, True), (300, 100, True), (128, 256, False),
                                    (777, 9344, True), (5, 32, True), (200, 1024, False)])
def test_gemm_argmax(dev, n, V, ln, fmt):
    from style_transfer_amd import ops
    rng = np.random.default_rng(n + V)
    h = rng.standard_normal((n, 256)).astype(np.float32) * 3
    W = (rng.standard_normal((V, 256)) * 0.1).astype(np.float32)
    b = rng.standard_normal(V).astype(np.float32)
    a_, b_ = (rng.standard_normal(256).astype(np.float32), rng.standard_normal(256).astype(np.float32))
    x = O.layer_norm(h, a_, b_) if ln else h
    logits = x.astype(np.float64) @ W.T.astype(np.float64) + b
    t = lambda v: torch.from_numpy(v).to(dev)
    ids, best, gap = ops.gemm_argmax(t(h), t(W), t(b), t(a_) if ln else None, t(b_) if ln else None, gemm_format=fmt)
    ids, best, gap = ids.cpu().numpy(), best.cpu().numpy(), gap.cpu().numpy()
    srt = np.sort(logits, -1)
    ref_gap = srt[:, -1] - (srt[:, -2] if V > 1 else 0)
    clear = ref_gap > 1e-3
    assert (ids[clear] == logits.argmax(-1)[clear]).all()
    # f32-grade in BOTH formats: the split products carry ~2^-22 per term, the f32 chain ~2^-24 per add
    err = np.abs(best - srt[:, -1]).max()
    print("gemm_argmax %s n=%d V=%d: max |best - f64| = %.3g (|logit| max %.3g)" % (fmt, n, V, err, np.abs(srt[:, -1]).max()))
    assert err < 2e-5 * max(1.0, np.abs(srt[:, -1]).max())
    assert np.abs(gap - ref_gap).max() < 1e-4


@pytest.mark.parametrize("V", [1000, 1024])  # 1024: rp_argmax16_kernel in the f16x3 format
def test_gemm_argmax_ties_pick_lowest_index(dev, V):
    """torch.max on CPU returns the first maximal index (model.py:292); exact ties must do the same."""
    from style_transfer_amd import ops
    h = torch.zeros(70, 256, device=dev)
    h[:, 0] = 1.0
    W = torch.zeros(V, 256, device=dev)
    b = torch.zeros(V, device=dev)
    for c in (3, 257, 600, 999):
        W[c, 0] = 2.0
    for fmt in FORMATS:
        ids, best, gap = ops.gemm_argmax(h, W, b, gemm_format=fmt)
        assert (ids.cpu().numpy() == 3).all() and (gap.cpu().numpy() == 0).all()


def test_scorer_embed(dev, scorer, scorer_sd):
    from style_transfer_amd import ops
    rng = np.random.default_rng(1)
    x = (rng.random((77, 256)) * 8).astype(np.float32)
    z = O.siamese_forward(scorer_sd[0], x)
    got = ops.scorer_embed(scorer, torch.from_numpy(x).to(dev), unit=False).cpu().numpy()
    assert relerr(got, z) < 2e-5
    zu = z / np.maximum(np.linalg.norm(z, axis=1, keepdims=True), 1e-8)
    got = ops.scorer_embed(scorer, torch.from_numpy(x).to(dev), unit=True).cpu().numpy()
    assert relerr(got, zu) < 2e-5


@pytest.mark.parametrize("mode", [dict(not_nll=True), dict(), dict(just_use_one=True), dict(not_nll=True, simplest=True),
                                  dict(not_nll=True, kp=3, kn=5), dict(kp=1, kn=10)])
def test_scorer_edit_step(dev, scorer, scorer_sd, mode):
    """three consecutive iterations of loss / gradient / Adam vs the oracle, per sentence"""
    from style_transfer_amd import ops
    mode = dict(mode)
    kp, kn = mode.pop("kp", 10), mode.pop("kn", 10)
    ssd, psd = scorer_sd
    B = 37
    rng = np.random.default_rng(5)
    lat = (rng.random((B, 256)) * 8).astype(np.float32)
    like = (rng.random((B, kp, 256)) * 8).astype(np.float32)
    unlike = (rng.random((B, kn, 256)) * 8).astype(np.float32)
    lr = np.where(np.arange(B) % 2 == 0, 0.2, 0.3).astype(np.float32)
    t = lambda v: torch.from_numpy(v).to(dev)
    like_z = ops.scorer_embed(scorer, t(like.reshape(-1, 256))).view(B, kp, 256)
    unlike_z = ops.scorer_embed(scorer, t(unlike.reshape(-1, 256))).view(B, kn, 256)
    x, m, v = t(lat.copy()), torch.zeros(B, 256, device=dev), torch.zeros(B, 256, device=dev)
    step = torch.zeros(B, dtype=torch.int32, device=dev)
    active = torch.ones(B, dtype=torch.uint8, device=dev)
    active[5] = 0
    opts = [O.Adam((1, 256), float(lr[b])) for b in range(B)]
    xo = [lat[b:b + 1].copy() for b in range(B)]
    # a second oracle trajectory that is NEVER re-synchronised to the GPU state: a compounding error would show here
    opts_free = [O.Adam((1, 256), float(lr[b])) for b in range(B)]
    xfree = [lat[b:b + 1].copy() for b in range(B)]
    sign_mismatch = 0
    for it in range(3):
        loss, grad = ops.scorer_edit_step(scorer, x, m, v, like_z, unlike_z, t(lr), step, active, tau=0.1,
                                          want_loss=True, want_grad=True, **mode)
        loss, grad, xg = loss.cpu().numpy(), grad.cpu().numpy(), x.cpu().numpy()
        for b in range(B):
            if b == 5:
                assert np.array_equal(xg[b], lat[b])
                continue
            lo, go = O.contrastive_loss_and_grad(ssd, psd, xo[b], like[b], unlike[b], 0.1, not_nll=mode.get("not_nll", False),
                                                 simplest=mode.get("simplest", False), just_use_one=mode.get("just_use_one", False))
            assert abs(loss[b] - lo) < 1e-4 * max(1, abs(lo)), (it, b, loss[b], lo)
            assert relerr(grad[b], go[0]) < 1e-3, (it, b, relerr(grad[b], go[0]))
            sign_mismatch += int((np.sign(grad[b]) != np.sign(go[0])).sum())
            xo[b] = opts[b].step(xo[b], go)
            # first Adam step is lr*sign(g): a sign flip of a near-zero gradient moves that coordinate by 2*lr
            bad = np.abs(xg[b] - xo[b][0]) > 1e-3 * np.abs(xo[b][0]).max()
            assert bad.sum() <= 1, (it, b, int(bad.sum()))
            xo[b] = xg[b:b + 1].copy()  # re-sync so a tolerated flip does not compound
            _, gf = O.contrastive_loss_and_grad(ssd, psd, xfree[b], like[b], unlike[b], 0.1, not_nll=mode.get("not_nll", False),
                                                simplest=mode.get("simplest", False), just_use_one=mode.get("just_use_one", False))
            xfree[b] = opts_free[b].step(xfree[b], gf)
    assert sign_mismatch <= 2
    # free-running oracle vs GPU after all three steps: only the coordinates behind those sign flips may differ
    drift = sum(int((np.abs(xg[b] - xfree[b][0]) > 1e-3 * np.abs(xfree[b][0]).max()).sum()) for b in range(B) if b != 5)
    assert drift <= sign_mismatch, (drift, sign_mismatch)
    assert step.cpu().numpy()[0] == 3 and step.cpu().numpy()[5] == 0


def test_encode_latent(dev, ae, ae_sd):
    from style_transfer_amd import ops
    from style_transfer_amd.data import pack_sentences
    sents = some_sentences(40) + [[], [5]]
    sents += [sents[0] + sents[1], (sents[2] * 8)[:59], sents[3][:16], (sents[4] * 3)[:17]]   # > 16 tokens: general kernel
    sents += [(sents[5] * 8)[:32], (sents[6] * 8)[:33], (sents[7] * 9)[:64], (sents[8] * 8)[:48]]   # tile boundaries of the MFMA form
    tok, pos, off, mx = pack_sentences(sents)
    t = lambda v: torch.from_numpy(v).to(dev)
    got = ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), mx).cpu().numpy()
    for i, s in enumerate(sents):
        if not s:
            assert np.abs(got[i]).max() == 0
            continue
        # padded to a longer length on purpose: padding must not change the latent
        src = np.zeros((1, min(len(s) + 3, 64)), np.int64)
        src[0, :len(s)] = s
        ref = O.encode_latent(ae_sd, src)[0]
        assert relerr(got[i], ref) < 2e-4, (i, relerr(got[i], ref))


@pytest.mark.parametrize("max_len", [16, 9])
def test_encode_latent_short_sentences_fused_kernel(dev, ae, ae_sd, max_len):
    """every sentence <= 16 tokens (all of yelp): in the f16x3 format the QKV projection and the attention of a layer run in
    ONE kernel whose workgroups own the sentences that start in their window of 65 - max_len rows.  Many windows, sentences
    of every length 1 .. max_len (also empty ones) in a random order, each latent against the oracle."""
    from style_transfer_amd import ops
    from style_transfer_amd.data import pack_sentences
    rng = np.random.default_rng(max_len)
    pool = [w for s_ in some_sentences(40) for w in s_]
    sents = []
    for i in range(260):
        n = int(rng.integers(0, max_len + 1)) if i % 7 else max_len
        sents.append([int(pool[int(k)]) for k in rng.integers(0, len(pool), n)])
    tok, pos, off, mx = pack_sentences(sents)
    assert mx == max_len and int(off[-1]) > 10 * (65 - max_len)
    t = lambda v: torch.from_numpy(v).to(dev)
    got = ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), mx).cpu().numpy()
    worst = 0.0
    for i, s_ in enumerate(sents):
        if not s_:
            assert np.abs(got[i]).max() == 0
            continue
        ref = O.encode_latent(ae_sd, np.asarray([s_], np.int64))[0]
        worst = max(worst, relerr(got[i], ref))
        assert relerr(got[i], ref) < 2e-4, (i, len(s_), relerr(got[i], ref))
    print("fused encoder kernel, max_len %d: %d sentences, %d rows, worst relative latent error %.3g" % (max_len, len(sents), int(off[-1]), worst))


def test_encode_latent_rejects_a_sentence_longer_than_declared(dev, ae):
    """st_encode_latent must not return ST_OK with stale attention rows when sent_off holds a sentence longer than the
    caller's max_sent_len (the attention kernels skip it): ST_EINVAL."""
    from style_transfer_amd import ops
    from style_transfer_amd.data import pack_sentences
    sents = some_sentences(6) + [(some_sentences(1)[0] * 9)[:40]]
    tok, pos, off, mx = pack_sentences(sents)
    t = lambda v: torch.from_numpy(v).to(dev)
    ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), mx)      # the true maximum: fine
    with pytest.raises(Exception, match="max_sent_len"):
        ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), 16)


@pytest.mark.parametrize("early", [True, False])
def test_greedy_decode(dev, ae, ae_sd, early):
    from style_transfer_amd import ops
    sents = some_sentences(24, seed=3)
    L = max(len(s) for s in sents)
    src = np.zeros((len(sents), L), np.int64)
    for i, s in enumerate(sents):
        src[i, :len(s)] = s
    lat = O.encode_latent(ae_sd, src)
    lat[3] += 0.7  # an edited latent
    ref, ref_gap = O.greedy_decode(ae_sd, lat, 60, early_exit=early, want_gap=True)
    ids, gap, steps = ops.greedy_decode(ae, torch.from_numpy(lat).to(dev), early_exit=early, want_gap=True)
    ids, gap = ids.cpu().numpy(), gap.cpu().numpy()
    assert steps == 59 if not early else steps <= 59
    n_bad = 0
    for b in range(len(sents)):
        end = 59
        for i in range(59):
            if ref[b, i] in (0, 3):
                end = i + 1
                break
        upto = end if early else 59
        if not np.array_equal(ids[b, :upto], ref[b, :upto]):
            first = int(np.nonzero(ids[b, :upto] != ref[b, :upto])[0][0])
            assert ref_gap[b, first] < 1e-3, "sentence %d diverges at %d with a clear gap %g" % (b, first, ref_gap[b, first])
            n_bad += 1
        else:
            assert np.abs(gap[b, :upto] - ref_gap[b, :upto]).max() < 2e-3
        if early:
            assert (ids[b, end:] == 0).all()
    assert n_bad <= 1


def test_decode_matches_input_sentence(dev, ae, ae_sd):
    """size-independent property: the surrogate autoencoder reconstructs most of its inputs"""
    from style_transfer_amd import ops
    from style_transfer_amd.data import pack_sentences
    sents = some_sentences(64, seed=9)
    tok, pos, off, mx = pack_sentences(sents)
    t = lambda v: torch.from_numpy(v).to(dev)
    lat = ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), mx)
    ids, _, _ = ops.greedy_decode(ae, lat)
    ids = ids.cpu().numpy()
    ok = sum(1 for i, s in enumerate(sents) if list(ids[i, :len(s)]) == s and ids[i, len(s)] == 3)
    assert ok >= 16, ok  # the CPU oracle reconstructs 23 of these 64


def test_accept_changes(dev):
    from style_transfer_amd import ops
    T, B = 59, 6
    acc = np.zeros((B, T), np.int32)
    new = np.zeros((B, T), np.int32)
    acc[:, :4] = [10, 11, 12, 3]
    new[0, :4] = [10, 11, 12, 3]                    # same
    new[1, :5] = [10, 11, 12, 13, 3]                # longer
    new[2, :4] = [10, 11, 3, 99]                    # shorter (junk after EOS ignored)
    new[3, :4] = [10, 11, 12, 0]; new[3, 5] = 77    # PAD ends like EOS -> same text
    new[4, :4] = [10, 50000, 12, 3]                 # id >= vocab -> UNK, differs from 11
    new[5, :4] = [10, 11, 13, 3]                    # inactive row: must be ignored
    t = lambda v: torch.from_numpy(v).to(dev)
    acc_d, new_d = t(acc.copy()), t(new)
    n_changes = torch.tensor([0, 6, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    changed = torch.zeros(B, dtype=torch.uint8, device=dev)
    active = torch.tensor([1, 1, 1, 1, 1, 0], dtype=torch.uint8, device=dev)
    rec = torch.zeros(B, T, dtype=torch.int32, device=dev)
    ops.accept_changes(new_d, acc_d, 9339, n_changes, changed, active, max_change=7, rec_ids=rec)
    assert changed.cpu().tolist() == [0, 1, 1, 0, 1, 0]
    assert n_changes.cpu().tolist() == [0, 7, 1, 0, 1, 0]
    assert active.cpu().tolist() == [1, 0, 1, 1, 1, 0]
    a = acc_d.cpu().numpy()
    assert np.array_equal(a[1], new[1]) and np.array_equal(a[0], acc[0]) and np.array_equal(a[5], acc[5])
    assert np.array_equal(rec.cpu().numpy()[:5], a[:5])


@pytest.mark.parametrize("T", [59, 64, 7, 80])
def test_accept_changes_wave_and_thread_forms_agree(dev, T):
    """The wave-per-sentence kernel (rows of <= 64 ids) against the thread-per-sentence walk on random rows built to hit
    every branch: equal prefixes with junk behind the terminator, EOS against PAD, one side longer, ids beyond the
    vocabulary, ids that spell the same word (canon map), rows without a terminator, inactive rows, the --retext
    confirmation and the change cap.  T = 80 takes the thread form in both runs (longer rows than a wave holds)."""
    import ctypes as C
    from style_transfer_amd import _lib, ops
    rng = np.random.default_rng(T)
    B, V = 4000, 50
    base = rng.integers(4, V + 6, (B, T)).astype(np.int32)            # some ids >= V -> UNK

    def with_end(rows):
        rows = rows.copy()
        n = rng.integers(0, T + 6, B)                                   # terminator position (>= T: none)
        for i in range(B):
            if n[i] < T:
                rows[i, n[i]] = 3 if rng.random() < 0.6 else 0
        return rows
    acc = with_end(base)
    new = acc.copy()
    mode = rng.integers(0, 6, B)
    for i in range(B):
        if mode[i] == 1:                                              # change one id somewhere
            new[i, rng.integers(0, T)] = rng.integers(4, V)
        elif mode[i] == 2:                                            # move the terminator
            new[i] = with_end(base)[i]
        elif mode[i] == 3:                                            # junk behind the first terminator only
            e = np.nonzero((acc[i] == 3) | (acc[i] == 0))[0]
            if e.size:
                new[i, e[0] + 1:] = rng.integers(0, V, T - e[0] - 1)
        elif mode[i] == 4:                                            # swap EOS and PAD
            new[i] = np.where(acc[i] == 3, 0, np.where(acc[i] == 0, 3, acc[i]))
    canon = np.arange(V, dtype=np.int32)
    canon[10:20] = canon[20:30]                                       # ten pairs of ids spell the same word
    conf = np.where(rng.random((B, 1)) < 0.5, new, with_end(base)).astype(np.int32)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    sw = C.c_int.in_dll(_lib.lib(), "g_accept_thread_form")
    out = {}
    for form in (0, 1):
        acc_d, n_ch = t(acc), t((np.arange(B) % 7).astype(np.int32))
        changed = torch.zeros(B, dtype=torch.uint8, device=dev)
        active = t((np.arange(B) % 5 != 0).astype(np.uint8))
        rec = torch.full((B, T), -1, dtype=torch.int32, device=dev)
        sw.value = form
        try:
            ops.accept_changes(t(new), acc_d, V, n_ch, changed, active, max_change=7, canon=t(canon), rec_ids=rec, confirm_ids=t(conf))
            torch.cuda.synchronize()
        finally:
            sw.value = 0
        out[form] = [x.cpu().numpy() for x in (acc_d, n_ch, changed, active, rec)]
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert 0.03 < out[0][2].mean() < 0.9                              # both outcomes occur


@pytest.fixture(params=["f16x3", "f32"])
def sl_fmt(request):
    """both operand formats of the similarity products meet the same tolerances"""
    from style_transfer_amd import ops
    prev = ops.simloss_format(request.param)
    yield request.param
    ops.simloss_format(prev)


def test_simloss_golden_b128(dev, sl_fmt):
    """reference torch code of siamese_contrastive.py:1060-1069 at B = 128 (fixture simloss_b128.npz)"""
    from style_transfer_amd import ops
    g = gold("simloss_b128.npz")
    ori = torch.from_numpy(g["ori"]).to(dev)
    d_ori = np.zeros_like(g["ori"])
    tot = 0.0
    for v in range(3):
        like = torch.from_numpy(np.ascontiguousarray(g["like"][v])).to(dev)
        loss, lse = ops.simloss_fwd(ori, like, tau=float(g["tau"]))
        assert np.abs(loss.cpu().numpy() - g["loss_rows"][v]).max() < 2e-4
        tot += float(loss.mean().item()) / 3
        gr = torch.full((128,), 1.0 / (3 * 128), device=dev)
        do, dl = ops.simloss_bwd(ori, like, lse, gr, tau=float(g["tau"]))
        d_ori += do.cpu().numpy()
        assert relerr(dl.cpu().numpy(), g["d_like"][v]) < 1e-3
    assert abs(tot - float(g["loss"])) < 1e-4
    assert relerr(d_ori, g["d_ori"]) < 1e-3


@pytest.mark.parametrize("Bq,Bk,off", [(64, 320, 32), (1088, 1088, 0), (96, 2048, 1900), (256, 1536, 1024), (1280, 1280, 0),
                                       (1024, 8192, 3072)])   # the last: one rank's row block of config 5 on 8 GPUs
def test_simloss_vs_oracle(dev, Bq, Bk, off, sl_fmt):
    from style_transfer_amd import ops
    rng = np.random.default_rng(Bq)
    ori = (rng.random((Bq, 256)) * 10).astype(np.float32)
    like = (rng.random((Bk, 256)) * 10).astype(np.float32)
    like[off:off + Bq] += 0.5 * ori
    gw = rng.random(Bq).astype(np.float32)
    loss_ref, lse_ref = O.simloss(ori, like, 0.1, off)
    do_ref, dl_ref = O.simloss_grads(ori, like, gw.astype(np.float64), 0.1, off)
    t = lambda v: torch.from_numpy(v).to(dev)
    loss, lse = ops.simloss_fwd(t(ori), t(like), 0.1, off)
    assert np.abs(loss.cpu().numpy() - loss_ref).max() < 3e-4
    do, dl = ops.simloss_bwd(t(ori), t(like), lse, t(gw), 0.1, off)
    assert relerr(do.cpu().numpy(), do_ref) < 1e-3
    assert relerr(dl.cpu().numpy(), dl_ref) < 1e-3


def test_simloss_full_size_b8192(dev, sl_fmt):
    """BASELINE config 5 at its full size (B = 8192 rows x 8192 in-batch negatives): the [B,B] matrix is never
    materialised by the kernels; checked against a chunked float64 torch evaluation of the same formula
    (siamese_contrastive.py:1060-1069) and through the size-independent identities sum_j dS[i,j] = 0
    (=> d_ori is orthogonal to ori after the cosine Jacobian) and loss >= 0."""
    from style_transfer_amd import ops
    B, tau = 8192, 0.1
    g = torch.Generator(device="cpu").manual_seed(1234)
    ori = (torch.rand(B, 256, generator=g) * 10).to(dev)
    like = (torch.rand(B, 256, generator=g) * 10).to(dev)
    like += 0.3 * ori
    gw = torch.rand(B, generator=g).to(dev)
    loss, lse = ops.simloss_fwd(ori, like, tau, 0)
    d_ori, d_like = ops.simloss_bwd(ori, like, lse, gw, tau, 0)
    o64 = ori.double().requires_grad_(True)
    l64 = like.double().requires_grad_(True)
    ou = o64 / o64.norm(dim=1, keepdim=True).clamp_min(1e-8)
    lu = l64 / l64.norm(dim=1, keepdim=True).clamp_min(1e-8)
    rows = []
    for c in range(0, B, 1024):                      # 1024 x 8192 float64 blocks
        sim = ou[c:c + 1024] @ lu.t() / tau
        rows.append(torch.logsumexp(sim, 1) - sim[torch.arange(sim.shape[0]), torch.arange(c, c + sim.shape[0])])
    ref = torch.cat(rows)
    (ref * gw.double()).sum().backward()
    assert float((loss.double() - ref.detach()).abs().max()) < 3e-4
    assert float(loss.min()) >= 0.0
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    assert rel(d_ori, o64.grad) < 1e-3 and rel(d_like, l64.grad) < 1e-3
    # cosine similarity is scale-invariant in each argument: every gradient row is orthogonal to its input row
    assert float(((d_ori * ori).sum(1).abs() / (d_ori.norm(dim=1) * ori.norm(dim=1) + 1e-30)).max()) < 1e-3
    assert float(((d_like * like).sum(1).abs() / (d_like.norm(dim=1) * like.norm(dim=1) + 1e-30)).max()) < 1e-3


def test_errors_are_reported_not_thrown_across_abi(dev, ae):
    from style_transfer_amd import ops, _lib
    with pytest.raises(_lib.StError, match="max_len"):
        ops.greedy_decode(ae, torch.zeros(2, 256, device=dev), max_len=100)


@pytest.mark.parametrize("n_rows", [48, 6000])   # small: unfused kernels; large: half-layer kernel
def test_f16_range_violation_is_reported(dev, ae_sd_torch, n_rows):
    """The f16x3 format needs every multiplied activation inside the f16 range.  A checkpoint that breaks this
    (here: feed_forward.w_1 of one layer scaled by 3e4 -> hidden units ~1e6) must raise ST_ERANGE instead of
    decoding garbage from inf - inf; the f32 format takes the same weights without complaint."""
    from style_transfer_amd import ops, _lib
    from style_transfer_amd.weights import AEWeights
    from style_transfer_amd.data import pack_sentences
    sd = {k: v.clone() for k, v in ae_sd_torch.items()}
    sd["decoder.layers.1.feed_forward.w_1.weight"] *= 3e4
    sd["encoder.layers.0.feed_forward.w_1.weight"] *= 3e4
    lat = torch.rand(n_rows, 256, device=dev) * 8
    bad = AEWeights(sd, dev, gemm_format="f16x3")
    with pytest.raises(_lib.StError, match="f16 range"):
        ops.greedy_decode(bad, lat)
    sents = some_sentences(40, seed=3) * (1 + n_rows // 100)
    tok, pos, off, mx = pack_sentences(sents)
    t = lambda v: torch.from_numpy(v).to(dev)
    with pytest.raises(_lib.StError, match="f16 range"):
        ops.encode_latent(bad, t(tok), t(pos), t(off), len(sents), mx)
    ok = AEWeights(sd, dev, gemm_format="f32")
    ids, _, _ = ops.greedy_decode(ok, lat)
    assert ids.shape == (n_rows, 59)


def test_large_batch_paths_equal_small_batch(dev, ae, ae_sd):
    """Full-size property: above 65536 rows the decoder / encoder switch to the fused O-projection+FFN1
    kernel and multi-round grids; every replica of a sentence must produce exactly the ids / (nearly) the
    latent of the small-batch run that the oracle checks above."""
    from style_transfer_amd import ops
    from style_transfer_amd.data import pack_sentences
    sents = some_sentences(40, seed=11)
    t = lambda v: torch.from_numpy(v).to(dev)
    tok, pos, off, mx = pack_sentences(sents)
    lat_small = ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), mx)
    reps = 240                                            # 2400 sentences, > 16384 token rows
    tok, pos, off, mx = pack_sentences(sents * reps)
    assert tok.size >= 65536
    lat_big = ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents) * reps, mx)
    d = (lat_big.view(reps, len(sents), 256) - lat_small[None]).abs().max().item()
    assert d < 1e-4 * lat_small.abs().max().item(), d
    ids_small, _, _ = ops.greedy_decode(ae, lat_small)
    big = lat_small.repeat(1700, 1).contiguous()          # 68000 decode jobs
    ids_big, _, _ = ops.greedy_decode(ae, big)
    same = (ids_big.view(1700, len(sents), 59) == ids_small[None]).all(dim=2).all(dim=0)
    assert int(same.sum().item()) >= len(sents) - 1, same.cpu().tolist()


@pytest.mark.parametrize("copies", [1, 6])
def test_screened_argmax_equals_full_product(dev, ae, gold_notnll, copies):
    """f16x3 format, >= 2048 live rows: the generator first runs a ONE-product (hi.hi) pass and redoes only the
    rows whose top-2 gap is inside the error bound of that pass with the 3-product kernel.  The emitted ids
    must equal those of the full product on every row (bit-exact integer work), on distinct latents.
    copies = 1: 6912 rows (column-split screening kernel); copies = 6: 41472 rows (>= 512 panels: the software-
    pipelined screening kernel with key-packed bookkeeping, rp_screen_kernel)."""
    from style_transfer_amd import ops
    if ae.gemm_format != "f16x3":
        pytest.skip("the screening pass exists only in the f16x3 format")
    rng = np.random.default_rng(5)
    base = gold_notnll["latent"].reshape(-1, 256)                       # 1152 latents of real edit trajectories
    lat = np.concatenate([base + rng.standard_normal(base.shape).astype(np.float32) * s
                          for _ in range(copies) for s in (0, 0.05, 0.1, 0.2, 0.4, 0.8)])
    if copies > 1:
        lat[:1152 * 6 * (copies - 1):6] += 1e-3                          # the noise-free copies are made distinct too
    lat = torch.from_numpy(lat.astype(np.float32)).to(dev)              # 6912 x copies distinct rows
    ids_full, _, steps_full = ops.greedy_decode(ae, lat, screen=False)
    ids_scr, _, steps_scr = ops.greedy_decode(ae, lat, screen=True)
    assert steps_full == steps_scr
    assert torch.equal(ids_full, ids_scr)
    ids_fixed, _, _ = ops.greedy_decode(ae, lat, screen=True, early_exit=False)
    ends = ((ids_full == 3) | (ids_full == 0)).int().argmax(dim=1)
    has_end = ((ids_full == 3) | (ids_full == 0)).any(dim=1)
    col = torch.arange(59, device=dev)[None]
    upto = torch.where(has_end, ends, torch.full_like(ends, 58))[:, None]
    assert torch.equal(torch.where(col <= upto, ids_fixed, torch.zeros_like(ids_fixed)),
                       torch.where(col <= upto, ids_full, torch.zeros_like(ids_full)))


def test_constraint_loss_wrapper_single_rank(dev):
    """contrastive.constraint_loss_and_grads on one rank with the HIP kernels == reference torch autograd"""
    from style_transfer_amd.contrastive import constraint_loss_and_grads
    g = gold("simloss_b128.npz")
    ori = torch.from_numpy(g["ori"]).to(dev)
    like = torch.from_numpy(g["like"]).to(dev)
    loss, d_ori, d_like = constraint_loss_and_grads(ori, like, float(g["tau"]))
    assert abs(loss - float(g["loss"])) < 1e-4
    assert relerr(d_ori.cpu().numpy(), g["d_ori"]) < 1e-3
    assert relerr(d_like.cpu().numpy(), g["d_like"]) < 1e-3


def test_fgim_matches_reference_and_oracle(dev, ae, ae_sd, id_to_word):
    """st_fgim_perturb (Classifier forward, BCE backward to the latent, x - eps * grad) and the
    host driver style_transfer_amd.fgim.fgim_attack against the reference's own run (golden: prediction, gradient,
    stdout lines, returned text) and, with step sizes large enough to change the decoded text, against the oracle."""
    from safetensors.torch import load_file
    from style_transfer_amd import ops
    from style_transfer_amd.fgim import fgim_attack, EPSILONS
    from style_transfer_amd.weights import ClassifierWeights
    from test_oracle_golden import _fgim_reference_lines
    csd_t = load_file(os.path.join(GOLD, "fgim_classifier.safetensors"))
    cw = ClassifierWeights(csd_t, dev)
    g = gold("fgim.npz")
    lat, tgt = torch.from_numpy(g["latent"]).to(dev), torch.from_numpy(g["target"]).to(dev)
    pred, grad, out = ops.fgim_perturb(cw, lat, tgt, g["eps"].tolist(), want_grad=True)
    assert np.abs(pred.cpu().numpy() - g["pred"]).max() < 1e-6
    assert np.abs(grad.cpu().numpy() - g["grad"]).max() < 1e-5 * np.abs(g["grad"]).max()
    ref_out = g["latent"][None] - g["eps"][:, None, None] * g["grad"][None]
    assert np.abs(out.cpu().numpy() - ref_out).max() < 1e-6
    lines = []
    texts, _ = fgim_attack(ae, cw, lat, tgt, id_to_word, gold_ids=gold("attack_notnll.npz")["init_ids"][[0, 1, 2, 3, 32, 33, 34, 35]],
                           log=lines.append)
    ref = _fgim_reference_lines()
    per = len(lines) // len(ref)
    for s, (ref_lines, ret) in enumerate(ref):
        assert lines[s * per:(s + 1) * per] == ref_lines
        assert texts[s][-1] == ret
    # large steps (the reference keeps such a list commented out, model.py:466): texts do change; oracle parity
    csd = {k: v.numpy() for k, v in csd_t.items()}
    big = [2.0, 4.0, 8.0, 16.0]
    _, _, out_big = ops.fgim_perturb(cw, lat, tgt, big)
    _, _, ref_big = O.fgim_perturb(csd, g["latent"], g["target"], big)
    assert np.abs(out_big.cpu().numpy() - ref_big).max() < 1e-5
    ids, _, _ = ops.greedy_decode(ae, out_big.view(-1, 256))
    ref_ids, ref_gap = O.greedy_decode(ae_sd, ref_big.reshape(-1, 256), early_exit=True, want_gap=True)
    ids = ids.cpu().numpy()
    for r in range(ids.shape[0]):
        if not np.array_equal(ids[r], ref_ids[r]):
            first = int(np.nonzero(ids[r] != ref_ids[r])[0][0])
            assert ref_gap[r, first] < 1e-3, (r, first, float(ref_gap[r, first]))
    assert (ids.reshape(len(big), -1, 59)[-1] != ids.reshape(len(big), -1, 59)[0]).any()   # the texts do change


@pytest.mark.parametrize("kind", ["negative", "non_monotonic", "beyond_rows"])
def test_encoder_rejects_bad_sentence_offsets_without_a_fault(dev, ae, kind):
    """Offsets come from the caller.  The kernels run before the host reads the verdict of the offset check, so every one
    of them must stay inside its buffers on offsets that are negative, go backwards or point past the rows; the call then
    returns ST_EINVAL.  A good call afterwards still gives the right latents (nothing was clobbered)."""
    from style_transfer_amd import ops, _lib
    from style_transfer_amd.data import pack_sentences
    sents = some_sentences(300, seed=21)
    tok, pos, off, mx = pack_sentences(sents)
    t = lambda v: torch.from_numpy(v).to(dev)
    good = ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), mx)
    bad = off.copy()
    if kind == "negative":
        bad[5] = -7000
    elif kind == "non_monotonic":
        bad[100], bad[101] = bad[101] + 3, bad[100]
    else:
        bad[200:] += 50000
    with pytest.raises(_lib.StError):
        ops.encode_latent(ae, t(tok), t(pos), t(bad), len(sents), mx)
    torch.cuda.synchronize()
    again = ops.encode_latent(ae, t(tok), t(pos), t(off), len(sents), mx)
    assert torch.equal(good, again)


def test_scorer_gradient_accuracy_against_float64(dev, scorer_sd_torch, scorer, ae_sd_torch):
    """Both scorer formats are f32-grade: d(-loss)/d(latent) of the --not_nll objective against a float64 torch evaluation
    (eval-mode BatchNorm, siamese_contrastive.py:367-436) on the encoder latents of 256 real test sentences.  The error is
    measured against the row's largest gradient component; its median must be below 5e-7 and its 99th percentile below
    2e-5 (observed: 1e-7 / 2.5e-6 in both formats; the tail are coordinates next to a ReLU kink)."""
    from style_transfer_amd import ops
    from style_transfer_amd.attack import AttackConfig, AttackRunner
    from style_transfer_amd.data import AttackSet
    from style_transfer_amd.weights import AEWeights
    s, p = scorer_sd_torch
    ae = AEWeights(ae_sd_torch, dev)
    data = AttackSet.from_json([os.path.join(GOLD, "yelp", "neg10_pos10_test%d.json" % i) for i in (0, 1)], ae.vocab).span(0, 256)
    B = len(data)
    r = AttackRunner(ae, None, AttackConfig(), dev)
    lat = r.encode_prepared(r.prepare(data))
    x, like, unlike = lat[:B].clone(), lat[B:B + 10 * B], lat[B + 10 * B:]
    d = lambda t: t.double()
    bn = lambda h, sd, k: (h - d(sd[k + ".running_mean"])) / torch.sqrt(d(sd[k + ".running_var"]) + 1e-5) * d(sd[k + ".weight"]) + d(sd[k + ".bias"])

    def siamese64(xx):
        h = torch.relu(bn(xx @ d(s["fc1.weight"]).t() + d(s["fc1.bias"]), s, "norm1"))
        h = torch.relu(bn(h @ d(s["fc2.weight"]).t() + d(s["fc2.bias"]), s, "norm2"))
        return h @ d(s["fc3.weight"]).t() + d(s["fc3.bias"])
    unit = lambda z: z / z.norm(dim=1, keepdim=True)
    lz64, uz64 = unit(siamese64(d(like.cpu()))).view(B, 10, 256), unit(siamese64(d(unlike.cpu()))).view(B, 10, 256)
    x64 = d(x.cpu()).requires_grad_(True)
    z = siamese64(x64)
    q = torch.nn.functional.leaky_relu(bn(z @ d(p["fc1.weight"]).t() + d(p["fc1.bias"]), p, "norm1"), 0.2)
    pp = unit(q @ d(p["fc2.weight"]).t() + d(p["fc2.bias"]))
    loss = ((((pp[:, None, :] * uz64).sum(-1) - (pp[:, None, :] * lz64).sum(-1)) / 0.1).sum(1) / 20).sum()
    (-loss).backward()
    g64 = x64.grad
    lz = ops.scorer_embed(scorer, like).view(B, 10, 256)
    uz = ops.scorer_embed(scorer, unlike).view(B, 10, 256)
    assert (lz.cpu().double() - lz64).abs().max() < 2e-6
    xx = x.clone()
    m, v = torch.zeros_like(xx), torch.zeros_like(xx)
    step = torch.zeros(B, dtype=torch.int32, device=dev)
    _, g = ops.scorer_edit_step(scorer, xx, m, v, lz, uz, torch.full((B,), 0.2, device=dev), step,
                                torch.ones(B, dtype=torch.uint8, device=dev), not_nll=True, want_grad=True)
    err = ((g.cpu().double() - g64).abs() / g64.abs().max(1, keepdim=True).values).flatten()
    med, p99 = float(err.median()), float(err.kthvalue(int(0.99 * err.numel())).values)
    print("scorer %s: gradient error / row max: median %.2e, 99 %% %.2e, max %.2e" % (scorer.gemm_format, med, p99, float(err.max())))
    assert med < 5e-7 and p99 < 2e-5
    scorer.check_range()          # nothing left the f16 range


def test_scorer_f16_range_violation_is_reported(dev, scorer_sd_torch):
    """f16x3 scorer: a latent beyond the f16 range (it would become inf - inf inside the split products) raises the
    library's sticky flag, which ScorerWeights.check_range (st_scorer_status) turns into an error; the flag is reset by the
    read, and the exact-f32 format takes the same input without complaint."""
    from style_transfer_amd import ops, _lib
    from style_transfer_amd.weights import ScorerWeights
    sw = ScorerWeights(scorer_sd_torch[0], scorer_sd_torch[1], dev, gemm_format="f16x3")
    x = torch.rand(64, 256, device=dev) * 8
    ops.scorer_embed(sw, x)
    sw.check_range()
    x[3, 7] = 1e6
    ops.scorer_embed(sw, x)
    with pytest.raises(_lib.StError, match="f16 range"):
        sw.check_range()
    sw.check_range()              # reset by the read
    z = ops.scorer_embed(ScorerWeights(scorer_sd_torch[0], scorer_sd_torch[1], dev, gemm_format="f32"), x)
    assert torch.isfinite(z).all()


def test_scorer_edit_step_is_bit_reproducible(dev, scorer):
    """The same edit steps from the same state must return the same bits, also with more workgroups than fit the chip at
    once (1024 tiles of 32 sentences).  Guards a real failure: the first f16x3 scorer kernel was built for two workgroups per
    CU, spilled registers, and returned different gradients from run to run on ~1 % of the rows (st_scorer.hip)."""
    from style_transfer_amd import ops
    torch.manual_seed(0)
    B = 32768
    x0 = torch.rand(B, 256, device=dev) * 8
    lz = ops.scorer_embed(scorer, torch.rand(B * 10, 256, device=dev) * 8).view(B, 10, 256)
    uz = ops.scorer_embed(scorer, torch.rand(B * 10, 256, device=dev) * 8).view(B, 10, 256)
    like_again = torch.rand(B * 10, 256, device=dev) * 8
    assert torch.equal(ops.scorer_embed(scorer, like_again), ops.scorer_embed(scorer, like_again))
    outs = []
    for rep in range(3):
        x = x0.clone()
        m, v = torch.zeros_like(x), torch.zeros_like(x)
        step = torch.zeros(B, dtype=torch.int32, device=dev)
        gs = []
        for it in range(4):
            _, g = ops.scorer_edit_step(scorer, x, m, v, lz, uz, torch.full((B,), 0.2, device=dev), step,
                                        torch.ones(B, dtype=torch.uint8, device=dev), not_nll=True, want_grad=True)
            gs.append(g)
        outs.append((x, torch.stack(gs)))
    for rep in (1, 2):
        assert torch.equal(outs[0][1], outs[rep][1]), "gradients differ between identical runs"
        assert torch.equal(outs[0][0], outs[rep][0]), "latents differ between identical runs"


def test_first_layer_table_gives_the_same_decodes(dev, ae_sd_torch, gold_notnll):
    """The first decoder layer's q | k | v depend on (token, position) only; with StDecoderWeights.l0_table the decode steps
    read them from a table built once per checkpoint instead of projecting them and caching K / V per job.  Same kernel,
    same values: the decoded ids must be identical, with early exit and with the reference's fixed 59 steps (prefixes beyond
    16 keys take the general attention kernel)."""
    from style_transfer_amd import ops
    from style_transfer_amd.weights import AEWeights
    ae = AEWeights(ae_sd_torch, dev, gemm_format="f16x3")
    assert ae.dec.l0_table and ae.dec.l0_table_len == 59
    rng = np.random.default_rng(8)
    base = gold_notnll["latent"].reshape(-1, 256)[:1152]
    lat = torch.from_numpy(np.concatenate([base, base + rng.standard_normal(base.shape).astype(np.float32) * 0.3]).astype(np.float32)).to(dev)
    for early in (True, False):
        a, _, sa = ops.greedy_decode(ae, lat, early_exit=early, l0_table=True)
        b, _, sb = ops.greedy_decode(ae, lat, early_exit=early, l0_table=False)
        assert sa == sb and torch.equal(a, b), early
    small = lat[:40].contiguous()            # a handful of rows: the small-grid kernel forms
    a, ga, _ = ops.greedy_decode(ae, small, want_gap=True, l0_table=True)
    b, gb, _ = ops.greedy_decode(ae, small, want_gap=True, l0_table=False)
    assert torch.equal(a, b) and torch.equal(ga, gb)


@pytest.mark.gpu
def test_mixed_length_chunks_encoded_apart_give_the_same_latents(dev, ae):
    """AttackRunner.prepare splits a chunk that mixes sentences of <= 16 tokens with longer ones into two launches
    (short: the fused QKV + attention kernels and the first-layer table) and sends the latents to their rows by index.
    Against the unsplit chunk: every row filled, the long sentences bit-identical (same kernels, other neighbours), the
    short ones within 2e-6 of the row's norm (another kernel family computes them)."""
    from style_transfer_amd.attack import AttackConfig, AttackRunner
    from style_transfer_amd.data import AttackSet, Ragged
    rng = np.random.default_rng(3)
    n = 6000
    lens = np.where(rng.random(n) < 0.6, rng.integers(1, 17, n), rng.integers(17, 32, n)).astype(np.int64)
    rag = Ragged.from_lengths(rng.integers(4, ae.vocab, int(lens.sum()), dtype=np.int32), lens)
    none = Ragged(np.zeros(0, np.int32), np.zeros(1, np.int64))
    data = AttackSet(rag, none, none, None, 0, 0)
    lat = {}
    for split in (True, False):
        r = AttackRunner(ae, None, AttackConfig(split_by_length=split), dev)
        r.ENC_CHUNK = 4096                       # two chunks: one split call each, then the rows of both
        prep = r.prepare(data)
        assert len(prep["chunks"]) == (4 if split and ae.gemm_format == "f16x3" else 2)   # (the fused kernels are f16x3 only)
        lat[split] = r.encode_prepared(prep).cpu().numpy()
    a, b = lat[True], lat[False]
    assert a.shape == b.shape == (n, 256) and np.isfinite(a).all()
    if ae.gemm_format == "f16x3":
        # the unsplit chunks' first layer read the (token, position) table rows where they lie; gathered into the q / k / v
        # buffer first (the form behind ST_AB_ENC_TAB_GATHER) the same values reach the same kernels
        import ctypes as C
        from style_transfer_amd import _lib
        sw = C.c_int.in_dll(_lib.lib(), "rp_enc_tab_gather")
        sw.value = 1
        try:
            r = AttackRunner(ae, None, AttackConfig(split_by_length=False), dev)
            r.ENC_CHUNK = 4096
            g = r.encode_prepared(r.prepare(data)).cpu().numpy()
        finally:
            sw.value = 0
        assert np.array_equal(g, b)
    long_ = lens > 16
    assert np.array_equal(a[long_], b[long_])
    err = np.abs(a[~long_] - b[~long_]).max(1) / np.linalg.norm(b[~long_], axis=1)
    print("short sentences, split against unsplit: worst |d latent| / |latent| = %.2e" % err.max())
    assert err.max() < 2e-6


def test_first_encoder_layer_table_gives_the_same_latents(dev, ae_sd_torch):
    """StEncoderWeights.l0_table: the first encoder layer's q | k | v per (token, position) from a table instead of a
    projection per row.  Short sentences (<= 16 tokens: the attention block of the fused kernel on gathered rows) and long
    ones (17..59 tokens: gathered qkv rows + the general attention kernels) must give bit-identical latents."""
    from style_transfer_amd import ops
    from style_transfer_amd.data import pack_sentences
    from style_transfer_amd.weights import AEWeights
    with_tab = AEWeights(ae_sd_torch, dev, gemm_format="f16x3")
    without = AEWeights(ae_sd_torch, dev, gemm_format="f16x3", l0_table=False)
    assert with_tab.enc.l0_table and not without.enc.l0_table
    t = lambda v: torch.from_numpy(v).to(dev)
    rng = np.random.default_rng(4)
    V = with_tab.vocab
    short = some_sentences(400, seed=13) + [[5], [7, 9]] + [[int(x) for x in rng.integers(4, V, 16)] for _ in range(5)]
    long_ = [[int(x) for x in rng.integers(4, V, n)] for n in rng.integers(1, 60, 300)]
    for sents in (short, long_, short * 40):           # (the last one: ~250 k rows, persistent grids)
        tok, pos, off, mx = pack_sentences(sents)
        a = ops.encode_latent(with_tab, t(tok), t(pos), t(off), len(sents), mx)
        b = ops.encode_latent(without, t(tok), t(pos), t(off), len(sents), mx)
        assert torch.equal(a, b), (len(sents), mx, float((a - b).abs().max()))


# kernel-form switches kept in the tree for same-box A/B runs: each is a library-level int
# (st_rowpanel.h, st_simloss.hip) that the environment sets once per process; the test flips them directly
AB_SWITCHES = [("rp_tail_shape16", 0), ("rp_tail_persistent", 0), ("rp_lin_shape16", 0), ("rp_screen_shape16", 0),
               ("rp_screen_paired", 0), ("rp_screen_maxnorm", 1), ("rp_argmax_shape16", 0), ("rp_enc_unfused", 1),
               ("rp_dec_fused", 1), ("rp_tail_min_rows", 4096), ("rp_no_emb_in_tail", 1), ("rp_kv_pos_major", 1)]


def test_every_ab_kernel_form_computes_the_same(dev, ae_sd_torch, gold_notnll):
    """Every kernel form kept behind an ST_AB_* switch must decode the same ids and encode the same latents as the
    default form (f16x3 format; 6912 distinct latents, so the screening pass, the exact-argmax redo, the half-layer
    kernel, the QKV kernel and the straggler tail all run; 260 packed sentences through the encoder)."""
    import ctypes as C
    from style_transfer_amd import _lib, ops, weights as W
    from style_transfer_amd.data import pack_sentences, Ragged
    L = _lib.lib()
    ae = W.AEWeights(ae_sd_torch, dev, gemm_format="f16x3")
    rng = np.random.default_rng(11)
    base = gold_notnll["latent"].reshape(-1, 256)
    lat = np.concatenate([base + rng.standard_normal(base.shape).astype(np.float32) * s for s in (0, 0.05, 0.1, 0.2, 0.4, 0.8)])
    lat = torch.from_numpy(lat.astype(np.float32)).to(dev)
    sents = some_sentences(260, seed=3)
    tokens, tok_pos, sent_off, max_len = pack_sentences(Ragged.from_lists(sents))
    enc_in = [torch.from_numpy(x).to(dev) for x in (tokens, tok_pos, sent_off)]

    def run():
        ids, _, _ = ops.greedy_decode(ae, lat)
        z = ops.encode_latent(ae, *enc_in, len(sents), max_len)
        return ids.cpu(), z.cpu()
    L.st_decode_workspace_bytes(None, 1, 60)   # (any call: the library has read its environment by now)
    ids0, z0 = run()
    for name, value in AB_SWITCHES:
        sw = C.c_int.in_dll(L, name)
        old = sw.value
        sw.value = value
        try:
            ids, z = run()
        finally:
            sw.value = old
        assert torch.equal(ids, ids0), "%s=%d decodes differently" % (name, value)
        # the encoder's unfused / fused forms share the product order: identical latents; the 32x32x16 forms
        # differ in the MFMA-internal summation order only
        assert float((z - z0).abs().max() / z0.abs().max()) < 2e-6, "%s=%d encodes differently" % (name, value)
    ids1, z1 = run()
    assert torch.equal(ids1, ids0) and torch.equal(z1, z0)


def test_simloss_forward_forms_agree(dev):
    """The 16x16x32 forward of the in-batch loss (online log-sum-exp behind the S tiles) against the 32x32x16 row-panel
    kernel with the LSE epilogue, same inputs: loss and lse of every row within 2e-6 relative (summation orders differ)."""
    import ctypes as C
    from style_transfer_amd import _lib, ops
    g = torch.Generator(device=dev).manual_seed(4)
    sw = C.c_int.in_dll(_lib.lib(), "g_sl_fwd16")
    for Bq, Bk, off in ((512, 1024, 256), (8192, 8192, 0), (64, 256, 100)):
        ori = torch.rand(Bq, 256, device=dev, generator=g) * 10 - 3
        like = torch.rand(Bk, 256, device=dev, generator=g) * 10 - 3
        outs = []
        for v in (1, 0):
            sw.value = v
            try:
                outs.append([t.double().cpu() for t in ops.simloss_fwd(ori, like, 0.1, off)])
            finally:
                sw.value = 1
        for a, b in zip(*outs):
            assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max()), (Bq, Bk, float((a - b).abs().max()))


def test_simloss_small_block_kernels_agree_with_the_gemm_path(dev):
    """Blocks of at most 128 x 128 run in one plain-f32 kernel per pass; the GEMM path (either operand format) must give
    the same loss, lse and gradients (1e-5 of each tensor's largest entry; d_like is added to what the buffer holds)."""
    import ctypes as C
    from style_transfer_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(6)
    sw = C.c_int.in_dll(L, "g_sl_small")
    for Bq, Bk, off in ((128, 128, 0), (32, 32, 0), (64, 128, 64), (32, 96, 10)):
        ori = torch.rand(Bq, 256, device=dev, generator=g) * 10 - 3
        like = torch.rand(Bk, 256, device=dev, generator=g) * 10 - 3
        gw = torch.rand(Bq, device=dev, generator=g) / Bq
        outs = []
        for v in (1, 0):
            sw.value = v
            try:
                loss, lse = ops.simloss_fwd(ori, like, 0.1, off)
                d_ori, d_like = ops.simloss_bwd(ori, like, lse, gw, 0.1, off)
                outs.append([t.double().cpu() for t in (loss, lse, d_ori, d_like)])
            finally:
                sw.value = 1
        for name, a, b in zip(("loss", "lse", "d_ori", "d_like"), *outs):
            assert float((a - b).abs().max()) < 1e-5 * float(b.abs().max()), (Bq, Bk, name, float((a - b).abs().max()), float(b.abs().max()))


def test_simloss_chunked_backward_form_equals_fused(dev):
    """ST_AB_SIMLOSS_CHUNKED: the round-2 backward (dS through HBM) against the fused kernel, same inputs"""
    import ctypes as C
    from style_transfer_amd import _lib, ops
    g = torch.Generator(device=dev).manual_seed(3)
    ori = torch.rand(512, 256, device=dev, generator=g) * 10
    like = torch.rand(1024, 256, device=dev, generator=g) * 10
    gw = torch.rand(512, device=dev, generator=g)
    _, lse = ops.simloss_fwd(ori, like, 0.1, 256)
    sw = C.c_int.in_dll(_lib.lib(), "g_sl_fused")
    outs = []
    for v in (1, 0):
        sw.value = v
        try:
            outs.append([t.cpu() for t in ops.simloss_bwd(ori, like, lse, gw, 0.1, 256)])
        finally:
            sw.value = 1
    for a, b in zip(*outs):
        assert float((a - b).abs().max() / b.abs().max()) < 1e-5
