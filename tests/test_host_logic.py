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
from style_transfer_amd.main import gather_results, shard_range
from style_transfer_amd.attack import AttackResult
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
from style_transfer_amd.attack import AttackConfig, empty_result
I, T = 18, 59
rng = np.random.default_rng(0)
for S in (11, 1):          # S = 1: rank 1 holds an empty shard (more ranks than sentences)
    full = AttackResult(rng.integers(0, 19, S).astype(np.int32), rng.integers(0, 9000, (S, T)).astype(np.int32),
                        rng.integers(0, 9000, (S, I, T)).astype(np.int32), rng.integers(0, 2, (S, I)).astype(np.uint8),
                        rng.integers(0, 9000, (S, 10, T)).astype(np.int32), None)
    lo, hi = shard_range(S, rank, world)
    mine = AttackResult(full.n_iter[lo:hi], full.init_ids[lo:hi], full.iter_ids[lo:hi], full.changed[lo:hi], full.like_ids[lo:hi], None)
    if lo == hi:
        mine = empty_result(AttackConfig(log_decodes=True), 10, 0)
    got = gather_results(mine, world, torch.device("cpu"))
    for f in ("n_iter", "init_ids", "iter_ids", "changed", "like_ids"):
        assert np.array_equal(getattr(got, f), getattr(full, f)), f
    assert got.unlike_ids is None
dist.destroy_process_group()
open(os.path.join(%(out)r, "ok%%d" %% rank), "w").write("ok")
"""


def test_two_rank_gather_reassembles_global_order(tmp_path):
    """N > 1 path: contiguous shards, all_gather of ragged per-shard records, global order restored (gloo)."""
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
loss, d_ori, d_like = constraint_loss_and_grads(ori, like, float(g["tau"]), fwd, bwd)
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
