"""CPU: host-side logic — vocabulary / data loading, packing, log grammar, CLI flags, sharding and the
world_size-2 gather (gloo)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT


def test_vocab_and_attack_set(id_to_word):
    from style_transfer_amd.data import AttackSet, canonical_ids
    assert len(id_to_word) == 9339 and id_to_word[:4] == ["<PAD>", "<UNK>", "<BOS>", "<EOS>"]
    assert canonical_ids(id_to_word) is None
    c = canonical_ids(["a", "b", "a", "c"])
    assert list(c) == [0, 1, 0, 3]
    files = [os.path.join(GOLD, "yelp", "neg10_pos10_test%d.json" % i) for i in (0, 1)]
    d = AttackSet.from_json(files, 9339)
    assert len(d) == 1000 and d.n_pos == 10 and d.n_neg == 10
    assert d.labels[:500] == [0] * 500 and d.labels[500:] == [1] * 500
    assert d.origin[0] == [78, 429, 3797, 83, 975, 527, 14, 34, 57, 1167, 418, 6, 418, 4]
    d2 = AttackSet.from_json(files, 9339, pos_num=3, neg_num=0)
    assert d2.n_pos == 3 and d2.n_neg == 0
    d3 = AttackSet.from_json(files, 100)          # ids >= V -> UNK (data.py:142-143)
    assert max(max(s) for s in d3.origin) < 100 and 1 in d3.origin[0]


def test_encoder_inputs_match_reference_golden(gold_notnll):
    """the packed encoder inputs equal what the reference's AttackDataLoader fed ae_model.forward"""
    from style_transfer_amd.data import AttackSet
    g = gold_notnll
    files = [os.path.join(GOLD, "yelp", "neg10_pos10_test%d.json" % i) for i in (0, 1)]
    d = AttackSet.from_json(files, 9339).subset(list(g["subset"]))
    for s in range(len(d)):
        rows = g["enc_src"][s * 21:(s + 1) * 21]
        strip = lambda r: [int(t) for t in r if t != 0]
        assert strip(rows[0]) == d.origin[s]
        for j in range(10):
            assert strip(rows[1 + j]) == d.like[s][j]
            assert strip(rows[11 + j]) == d.unlike[s][j]


def test_pack_sentences():
    from style_transfer_amd.data import pack_sentences
    tok, pos, off, mx = pack_sentences([[5, 6, 7], [], [9], [4, 0, 8]])
    assert tok.tolist() == [5, 6, 7, 9, 4, 8]
    assert pos.tolist() == [0, 1, 2, 0, 0, 2]       # id 0 inside a sentence: masked position, index kept
    assert off.tolist() == [0, 3, 3, 4, 6] and mx == 3


def test_truncation_to_max_sequence_length():
    from style_transfer_amd.data import _clip
    assert len(_clip(list(range(4, 104)), 9339, 1, 60)) == 59


def test_id2text_matches_reference_semantics(id_to_word):
    from style_transfer_amd.data import id2text_sentence
    assert id2text_sentence([4, 5, 3, 6], id_to_word) == ". the"
    assert id2text_sentence([4, 0, 6], id_to_word) == "."
    assert id2text_sentence([4, 99999], id_to_word) == ". <UNK>"
    assert id2text_sentence([2, 1], id_to_word) == "<BOS> <UNK>"


def test_extract_out_equals_get_i_iter_semantics():
    from style_transfer_amd.logfmt import LOG_PREFIX, extract_out, headline
    lines = []
    for s in range(3):
        lines.append(LOG_PREFIX + headline("Sentence Pair.%d" % s))
        lines.append(LOG_PREFIX + "Src: x")
        if s != 1:
            lines.append(LOG_PREFIX + "Change.0, a%d" % s)
        lines.append(LOG_PREFIX + "iteration.0, b%d" % s)
        if s == 2:
            lines += [LOG_PREFIX + "Change.1, c", LOG_PREFIX + "iteration.1, c", LOG_PREFIX + "Change.2, d", LOG_PREFIX + "Change.3, e"]
    assert extract_out(lines, pos_c=2, neg_c=2, limit=3) == ["a0\n", "b1\n", "d\n"]
    assert extract_out(lines, pos_c=0, neg_c=0, limit=3) == ["a0\n", "b1\n", "a2\n"]
    assert extract_out(lines, it=0, limit=3) == ["b0\n", "b1\n", "b2\n"]


@pytest.mark.skipif(not os.path.exists("/root/reference/outputs/get_i_iter.py"), reason="reference script not on this box")
def test_reference_get_i_iter_runs_unchanged_on_our_log_format(tmp_path):
    """outputs/get_i_iter.py (with an inert `fasttext` stub module on the path — it imports it but never
    uses it) parses the golden log and our extractor agrees with it."""
    from style_transfer_amd.logfmt import extract_out
    (tmp_path / "fasttext.py").write_text("")
    log = os.path.join(GOLD, "attack_notnll_log.txt")
    out = tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=str(tmp_path))
    subprocess.check_call([sys.executable, "/root/reference/outputs/get_i_iter.py", "--file", log, "--out", str(out),
                           "--pos_c", "2", "--neg_c", "2", "--partial"], env=env, stdout=subprocess.DEVNULL)
    ref = open(out).readlines()
    mine = extract_out(open(log).read().splitlines(), 2, 2, limit=50)
    assert ref == mine


def test_cli_flags_and_defaults_match_main_py():
    from style_transfer_amd.main import build_parser
    a = build_parser().parse_args(["--attack", "--not_nll", "--pos_victim_lr=0.2", "--neg_victim_lr=0.2"])
    assert (a.id_pad, a.id_unk, a.id_bos, a.id_eos) == (0, 1, 2, 3)
    assert a.task == "yelp" and a.word_dict_max_num == 5 and a.batch_size == 128 and a.max_sequence_length == 60
    assert (a.num_layers_AE, a.transformer_model_size, a.transformer_ff_size, a.latent_size) == (2, 256, 1024, 256)
    assert a.tau == 0.1 and a.victim_optimizer == "Adam" and a.victim_lr == 0.25
    assert a.attack_max_iter == 18 and a.attack_max_change == 7
    assert a.pos_num is None and a.neg_num is None and a.pos_victim_lr == 0.2 and a.neg_victim_lr == 0.2
    assert a.attack and a.not_nll and not (a.simplest_siamese or a.reverse_order or a.just_use_one or a.retext or a.dance)


def test_shard_range_covers_everything():
    from style_transfer_amd.main import shard_range
    for n in (0, 1, 7, 1000, 1001):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


GATHER_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from style_transfer_amd.main import gather_dense_to_rank0
from style_transfer_amd.pipeline import gather_to_rank0, shard_range
from style_transfer_amd.attack import AttackConfig, AttackResult, LeanRecord, empty_result
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
I, T = 18, 59
rng = np.random.default_rng(0)
cfg = AttackConfig(log_decodes=True)
for S in (11, 1):          # S = 1: rank 1 holds an empty shard (more ranks than sentences)
    n_iter = rng.integers(1, 19, S).astype(np.int32)
    changed = (rng.integers(0, 3, (S, I)) == 0).astype(np.uint8) * (np.arange(I)[None, :] < n_iter[:, None])
    rows = rng.integers(0, 9000, (int(changed.sum()), T)).astype(np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    full = LeanRecord(t(n_iter), t(rng.integers(0, 9000, (S, T)).astype(np.int32)), t(changed.astype(np.uint8)), t(rows),
                      t(rng.integers(0, 9000, (S, 10, T)).astype(np.int32)), None, t(rng.standard_normal((S, 256)).astype(np.float32)))
    lo, hi = shard_range(S, rank, world)
    r0, r1 = int(changed[:lo].sum()), int(changed[:hi].sum())
    mine = LeanRecord(full.n_iter[lo:hi], full.init_ids[lo:hi], full.changed[lo:hi], full.rows[r0:r1], full.like_ids[lo:hi], None,
                      full.final_latent[lo:hi])
    if lo == hi:
        mine = LeanRecord.empty(AttackConfig(log_decodes=True), 10, 0, want_latent=True)
    got = gather_to_rank0(mine, rank, world, "cpu")
    if rank == 0:
        for f in ("n_iter", "init_ids", "changed", "rows", "like_ids", "final_latent"):
            assert torch.equal(getattr(got, f), getattr(full, f)), f
        assert got.unlike_ids is None
        dense = got.expand()                      # dense table from the changed rows
        k = 0
        for s in range(S):
            cur = full.init_ids[s].numpy()
            for it in range(I):
                if changed[s, it]:
                    cur = rows[k]; k += 1
                want = cur if it < n_iter[s] else np.zeros(T, np.int32)
                assert np.array_equal(dense.iter_ids[s, it], want), (s, it)
    else:
        assert got is None
    # --dance keeps the dense records, with the loss / score tables and the final latents (ADVICE r1: these were dropped)
    dfull = AttackResult(n_iter, full.init_ids.numpy(), rng.integers(0, 9000, (S, 2, T)).astype(np.int32),
                         np.zeros((S, 2), np.uint8), full.like_ids.numpy(), None, full.final_latent.numpy())
    dfull.dance = dict(loss=rng.standard_normal((S, 2)).astype(np.float32), scores=rng.standard_normal((S, 2, 20)).astype(np.float32),
                       n_pos=10, not_nll=True)
    dm = AttackResult(dfull.n_iter[lo:hi], dfull.init_ids[lo:hi], dfull.iter_ids[lo:hi], dfull.changed[lo:hi], dfull.like_ids[lo:hi],
                      None, dfull.final_latent[lo:hi])
    dm.dance = dict(dfull.dance, loss=dfull.dance["loss"][lo:hi], scores=dfull.dance["scores"][lo:hi])
    dg = gather_dense_to_rank0(dm, rank, world, "cpu")
    if rank == 0:
        for f in ("n_iter", "init_ids", "iter_ids", "changed", "like_ids", "final_latent"):
            assert np.array_equal(getattr(dg, f), getattr(dfull, f)), f
        assert np.array_equal(dg.dance["loss"], dfull.dance["loss"]) and np.array_equal(dg.dance["scores"], dfull.dance["scores"])
        assert dg.dance["n_pos"] == 10 and dg.dance["not_nll"] is True
dist.destroy_process_group()
open(os.path.join(%(out)r, "ok%%d" %% rank), "w").write("ok")
"""


def test_two_rank_gather_reassembles_global_order(tmp_path):
    """N > 1 path: contiguous shards, the compact records (changed rows only) sent to rank 0 with exact sizes, global
    order restored, dense table rebuilt; --dance's dense records incl. loss / scores / latents (gloo)."""
    import socket
    with socket.socket() as so:          # a free rendezvous port (a fixed one can collide with another run)
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(GATHER_WORKER % dict(root=ROOT, out=str(tmp_path)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()   # (the ranks' stdout lines interleave)


SIMLOSS_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np, torch, torch.distributed as dist
import st_oracle as O
from style_transfer_amd.contrastive import constraint_loss_and_grads
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
g = dict(np.load(os.path.join(%(root)r, "tests", "golden", "simloss_b128.npz")))
B, b = 128, 128 // world
def fwd(ori, like, tau, off):
    l, lse = O.simloss(ori.numpy(), like.numpy(), tau, off)
    return torch.from_numpy(l.astype(np.float32)), torch.from_numpy(lse.astype(np.float32))
def bwd(ori, like, lse, gw, tau, off, d_like):
    do, dl = O.simloss_grads(ori.numpy(), like.numpy(), gw.numpy().astype(np.float64), tau, off)
    d_like += torch.from_numpy(dl.astype(np.float32))
    return torch.from_numpy(do.astype(np.float32)), d_like
ori = torch.from_numpy(g["ori"][rank * b:(rank + 1) * b].copy())
like = torch.from_numpy(g["like"][:, rank * b:(rank + 1) * b].copy())
def gloo_reduce_scatter(out, inp):          # gloo has no reduce_scatter: all_reduce + this rank's slice
    dist.all_reduce(inp)
    out.copy_(inp.view(world, -1)[rank])
loss, d_ori, d_like = constraint_loss_and_grads(ori, like, float(g["tau"]), fwd, bwd, gloo_reduce_scatter)
assert abs(loss - float(g["loss"])) < 1e-5, (loss, float(g["loss"]))
assert np.abs(d_ori.numpy() - g["d_ori"][rank * b:(rank + 1) * b]).max() < 1e-3 * np.abs(g["d_ori"]).max()
assert np.abs(d_like.numpy() - g["d_like"][:, rank * b:(rank + 1) * b]).max() < 1e-3 * np.abs(g["d_like"]).max()
dist.destroy_process_group()
open(os.path.join(%(out)r, "ok%%d" %% rank), "w").write("ok")
"""


def test_two_rank_global_negatives_loss_matches_reference(tmp_path):
    """config 5 wiring: rows sharded over 2 ranks, all_gather of `like`, row_offset targets, reduce-scatter of
    d_like; the result must equal the reference's single-process torch loss/gradients (simloss_b128.npz)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(SIMLOSS_WORKER % dict(root=ROOT, out=str(tmp_path)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_dance_digest_matches_reference_definition():
    """--dance prints the second value of get_contrastive_loss (siamese_contrastive.py:433, :443-444)."""
    from style_transfer_amd.attack import dance_digest
    scores = np.array([0.5, -0.2, 0.1, 3.0, -1.0, 0.2], np.float32)   # 3 pos, 3 neg
    # --not_nll: #(pos_loss < 0) + #(neg_loss < 0) with pos_loss = -pos_score, neg_loss = neg_score
    assert dance_digest(scores, 3, True) == 2 + 1
    # NLL: per positive i the argmax of [pos_i, neg_0, neg_1, neg_2]
    assert dance_digest(scores, 3, False) == [[1], [1], [1]]
    assert dance_digest(np.array([5.0, 0.0, 1.0, 2.0], np.float32), 1, False) == [[0]]


TRAIN_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import st_train_oracle as T
import train_common as TC
from style_transfer_amd.train_scorer import ScorerTrainer, TrainConfig, train_contrastive, epoch_order, PARAM_ORDER, PARAM_SHAPES
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")

class OracleTrainer(ScorerTrainer):
    # the HIP calls replaced by the numpy restatement: this test is about the host loop and the collectives
    def _sds(self):
        s, p = self.state_dicts()
        return {k: v.numpy() for k, v in s.items()}, {k: v.numpy() for k, v in p.items()}
    def loss_and_grads(self, ori, like, unlike, keep_masks=None, seed=None):
        ssd, psd = self._sds()
        loss, choice, grads, stats = T.scorer_train_loss_and_grads(ssd, psd, ori.numpy(), like.numpy(), unlike.numpy(), None)
        shapes = TC.named(ssd, psd)
        self.grads.copy_(torch.from_numpy(TC.flat(grads, shapes)))
        self.loss[0] = float(loss)
        bn = [stats[n + "." + k] for n, k, _ in __import__("style_transfer_amd.train_scorer", fromlist=["BN_ORDER"]).BN_ORDER]
        self.bn.copy_(torch.from_numpy(np.concatenate(bn)))
        return torch.from_numpy(choice.astype(np.int32))
    def adam(self):
        self.step_no += 1
        p, m, v = T.adam_update(self.params.numpy(), self.grads.numpy(), self.m.numpy(), self.v.numpy(), self.step_no, self.cfg.lr)
        self.params.copy_(torch.from_numpy(p)); self.m.copy_(torch.from_numpy(m)); self.v.copy_(torch.from_numpy(v))

ssd, psd = TC.load_init()
g = np.load(os.path.join(TC.GOLD, "attack_notnll.npz"))
ori, like, unlike = (torch.from_numpy(g[k].copy()) for k in ("init_latent", "like_latent", "unlike_latent"))
tr = OracleTrainer(ssd, psd, TrainConfig(drop_p=0.0), "cpu")
lines = []
hist = train_contrastive(tr, ori, like, unlike, batch_size=32, epochs=2, log=lines.append, rank=rank, world=world)
# every rank holds the same parameters after the all-reduced steps
gathered = [torch.zeros_like(tr.params) for _ in range(world)]
dist.all_gather(gathered, tr.params)
assert all(torch.equal(gathered[0], t) for t in gathered)
# single-process emulation of the same data-parallel schedule: per-half-batch gradients averaged
ref = OracleTrainer(ssd, psd, TrainConfig(drop_p=0.0), "cpu")
for epoch in range(2):
    for rec in epoch_order(64, 32):
        gs = []
        for r in range(world):
            sel = torch.as_tensor(rec[r * 16:(r + 1) * 16])
            ref.loss_and_grads(ori[sel], like[sel].transpose(0, 1), unlike[sel].transpose(0, 1))
            gs.append(ref.grads.clone())
        ref.grads.copy_(sum(gs) / world)
        ref.adam()
# elementwise Adam turns last-bit differences of ~1e-8 gradients into +-lr moves (train_common.noise_grad_tensors):
# compare the tensors that carry real gradient
off = 0
for (net, name), shape in zip(PARAM_ORDER, PARAM_SHAPES):
    n = int(np.prod(shape))
    if not (name.endswith("bias") and name.startswith("fc")):
        d = (tr.params[off:off + n] - ref.params[off:off + n]).abs()
        assert float((d <= 1e-5).float().mean()) >= 0.999, (net, name, float(d.max()))
    off += n
assert hist[1][0] < hist[0][0] and lines[-1].startswith("Epoch Loss=")
dist.destroy_process_group()
open(os.path.join(%(out)r, "ok%%d" %% rank), "w").write("ok")
"""


def test_two_rank_data_parallel_scorer_training(tmp_path):
    """train_contrastive on 2 ranks (gloo): each rank takes half of every batch, gradients are averaged with one
    all-reduce of the flat vector, both ranks end with identical parameters that match a single-process emulation."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(TRAIN_WORKER % dict(root=ROOT, out=str(tmp_path)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_epoch_order_is_the_reference_shuffle():
    """random.seed(234); random.shuffle(list) (siamese_contrastive.py:276-290, data.py:403-408): same permutation for
    any list of that length, batches of int(n / batch_size) * batch_size records, remainder dropped."""
    import random
    from style_transfer_amd.train_scorer import epoch_order
    items = ["r%d" % i for i in range(1000)]
    random.seed(234)
    random.shuffle(items)
    batches = epoch_order(1000, 128)
    assert len(batches) == 7 and all(len(b) == 128 for b in batches)
    assert ["r%d" % i for b in batches for i in b] == items[:896]


def test_ragged_sets_behave_like_the_lists_they_replace():
    from style_transfer_amd.data import AttackSet, Ragged, pack_sentences
    sents = [[5, 6, 7], [], [9], [4, 8]]
    r = Ragged.from_lists(sents)
    assert len(r) == 4 and list(r) == sents and r[-1] == [4, 8] and r.lens.tolist() == [3, 0, 1, 2]
    assert list(r.take([3, 0, 3])) == [[4, 8], [5, 6, 7], [4, 8]] and list(r.span(1, 3)) == [[], [9]]
    assert list(Ragged.concat([r.span(0, 1), r.span(2, 4)])) == [[5, 6, 7], [9], [4, 8]]
    d = AttackSet([[4, 5], [6]], [[[7], [8, 9]], [[10], []]], [[[11]], [[12, 13]]])
    assert (len(d), d.n_pos, d.n_neg) == (2, 2, 1) and d.like[1] == [[10], []] and d.unlike[0] == [[11]]
    assert list(d.all_sentences()) == [[4, 5], [6], [7], [8, 9], [10], [], [11], [12, 13]]
    sub = d.subset([1, 1, 0])
    assert sub.origin[0] == [6] and sub.like[2] == [[7], [8, 9]] and sub.unlike[1] == [[12, 13]]
    sp = d.span(1, 2)
    assert len(sp) == 1 and sp.origin[0] == [6] and sp.like[0] == [[10], []]
    # the vectorised synthetic generator: shapes of SURVEY 8d, a million-scale call must not loop in Python
    syn = AttackSet.synthetic(5000, 9339, "yelp", seed=3)
    ol, rl = syn.origin_r.lens, syn.like_r.lens
    assert ol.min() >= 6 and ol.max() <= 15 and abs(ol.mean() - 10.3) < 0.3 and rl.min() == 1 and rl.max() == 16
    assert syn.origin_r.tokens.min() >= 4 and syn.origin_r.tokens.max() < 9339 and len(syn.unlike_r) == 50000
    tok, pos, off, mx = pack_sentences(syn.all_sentences())
    assert off[-1] == tok.size and mx == 16 and pos.max() == 15
    am = AttackSet.synthetic(100, 19724, "amazon", seed=3)
    assert am.origin_r.lens.min() >= 8 and am.like_r.lens.min() >= 8 and am.like_r.lens.max() == 31


def test_log_formatter_matches_the_python_writer(id_to_word):
    """st_format_attack_log (host C function; compact records) writes the same bytes as the per-sentence Python
    writer on the dense records: Src / Rec / Sim / Ref lines, Change numbering, repeated iteration texts, ids >= V
    -> <UNK>, texts cut at EOS / PAD, empty texts, sentences that stop early."""
    import torch
    from style_transfer_amd.attack import AttackConfig, LeanRecord, write_attack_log
    from style_transfer_amd.data import AttackSet
    from style_transfer_amd.logfmt import LOG_PREFIX, LogWriter, VocabBytes, format_attack_block
    rng = np.random.default_rng(11)
    S, I, T, V = 40, 18, 59, len(id_to_word)

    def ids_row():
        n = int(rng.integers(0, 14))
        row = rng.integers(4, V + 50, T).astype(np.int32)      # some ids beyond the vocabulary
        row[n] = 3 if rng.random() < 0.7 else 0
        return row
    n_iter = rng.integers(1, I + 1, S).astype(np.int32)
    changed = ((rng.random((S, I)) < 0.35) & (np.arange(I)[None, :] < n_iter[:, None])).astype(np.uint8)
    rows = np.stack([ids_row() for _ in range(int(changed.sum()))]) if changed.sum() else np.zeros((0, T), np.int32)
    init = np.stack([ids_row() for _ in range(S)])
    data = AttackSet.synthetic(S, V, "yelp", n_pos=3, n_neg=2, seed=5)
    for with_refs in (False, True):
        like = np.stack([ids_row() for _ in range(S * 3)]).reshape(S, 3, T) if with_refs else None
        unlike = np.stack([ids_row() for _ in range(S * 2)]).reshape(S, 2, T) if with_refs else None
        t = lambda a: None if a is None else torch.from_numpy(a)
        rec = LeanRecord(t(n_iter), t(init), t(changed), t(rows), t(like), t(unlike))
        block = format_attack_block(VocabBytes(id_to_word), data, rec, first_index=500)
        w = LogWriter(None, echo=False)
        write_attack_log(w, data, rec.expand(), id_to_word, first_index=500)
        want = "".join(LOG_PREFIX + m + "\n" for m in w.lines)
        assert block.tobytes().decode("utf-8") == want
        w2 = LogWriter(None, echo=False)
        w2.write_block(block)
        assert w2.lines == w.lines
        # to a file: the byte array itself (no copy), a strided view of it and plain bytes all land as the same text,
        # between lines written the ordinary way
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "log")
            w3 = LogWriter(path, echo=False)
            w3("head")
            w3.write_block(block)
            w3.write_block(np.repeat(block, 2)[::2])
            w3.write_block(block.tobytes())
            w3("tail")
            w3.close()
            assert open(path, "rb").read().decode("utf-8") == LOG_PREFIX + "head\n" + want * 3 + LOG_PREFIX + "tail\n"


def test_bench_self_launch_control_flow_without_a_gpu():
    """`python bench.py --gpus 2 --dry-run`, no launcher: two worker processes, gloo rendezvous on 127.0.0.1, barrier,
    max-over-ranks timing, exactly one JSON line (rank 0).  The same code path starts the real N-GPU run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["dry_run"] is True and j["ms_per_step"] >= 20.0   # the slower rank's time


def test_bench_launcher_returns_promptly_when_a_worker_dies():
    """A worker that dies before the rendezvous must not leave its sibling (waiting in the store) and the launcher hanging:
    the launcher polls its children, stops the others and passes the failing return code on."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ST_BENCH_TEST_DIE_RANK"] = "1"
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=120)
    dt = time.monotonic() - t0
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert "rank 1 exited with 3" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]      # no result line from a broken run
    assert dt < 60, dt


def test_parallel_log_formatter_equals_the_single_block(id_to_word):
    """format_attack_blocks (rank 0 after the gather: chunks on host threads, in order) == format_attack_block"""
    import torch
    from style_transfer_amd.attack import LeanRecord
    from style_transfer_amd.data import AttackSet
    from style_transfer_amd.logfmt import VocabBytes, format_attack_block, format_attack_blocks
    rng = np.random.default_rng(3)
    S, I, T, V = 3000, 18, 59, len(id_to_word)
    n_iter = rng.integers(1, I + 1, S).astype(np.int32)
    changed = ((rng.random((S, I)) < 0.3) & (np.arange(I)[None, :] < n_iter[:, None])).astype(np.uint8)
    mk = lambda n: np.where(rng.random((n, T)) < 0.1, 3, rng.integers(4, V, (n, T))).astype(np.int32)
    rec = LeanRecord(torch.from_numpy(n_iter), torch.from_numpy(mk(S)), torch.from_numpy(changed), torch.from_numpy(mk(int(changed.sum()))))
    full = AttackSet.synthetic(S, V, "yelp", seed=9)
    only = AttackSet.synthetic(S, V, "yelp", seed=9, origins_only=True)
    assert np.array_equal(full.origin_r.tokens, only.origin_r.tokens) and len(only) == S and only.n_pos == 0
    vocab = VocabBytes(id_to_word)
    one = format_attack_block(vocab, full, rec, 7).tobytes()
    many = [b.tobytes() for b in format_attack_blocks(vocab, only, rec, 7, threads=4, min_chunk=100)]
    assert len(many) > 4 and b"".join(many) == one
    sub = rec.span(100, 250)
    assert len(sub) == 150 and int(sub.changed.sum()) == sub.rows.shape[0]
    assert format_attack_block(vocab, full.span(100, 250), sub, 107).tobytes() in one



def test_ae_training_batches_pack_the_same_from_lists_and_from_flat_arrays():
    """AETrainer.pack_arrays (the epoch loop's vectorised form) against pack_lists (the per-sentence form the golden tests
    use): the same encoder and decoder arrays, including sentences longer than max_seq_length - 1 and ids outside the vocabulary."""
    from style_transfer_amd.data import Ragged
    from style_transfer_amd.train_ae import AETrainer
    rng = np.random.default_rng(8)
    V, B, K = 500, 32, 10
    sent = lambda: [int(t) for t in rng.integers(4, V + 40, int(rng.integers(1, 70) if rng.random() < 0.1 else rng.integers(1, 16)))]
    origin = [sent() for _ in range(B)]
    views = [[sent() for _ in range(B)] for _ in range(K)]
    enc_l, dec_l, enc_sents, _ = AETrainer.pack_lists(origin, views, V)
    enc_a, dec_a, n_sent = AETrainer.pack_arrays(Ragged.from_lists(origin), Ragged.from_lists([s for v in views for s in v]), V)
    assert n_sent == len(enc_sents) == (K + 2) * B
    for a, b in zip(enc_l + dec_l, enc_a + dec_a):
        assert np.array_equal(np.asarray(a, np.int64), np.asarray(b, np.int64))
