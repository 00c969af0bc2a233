"""Input side of the boundary: vocabulary and the neg10_pos10_test{0,1}.json attack set.

Mirrors data.py of the reference (load_word_dict_info :67-82, pad_batch_seuqences :119-152,
AttackDataLoader :298-505) but batch-first and length-packed: instead of one padded [1,L] tensor per
sentence, a batch is (tokens, tok_pos, sent_off) over all non-pad tokens.
"""
import json
import os

import numpy as np


def load_word_dict_info(word_dict_file: str, max_num: int):
    """data.py:67-82 — `word[\\tcount]` lines, stop at the first count < max_num."""
    id_to_word = []
    with open(word_dict_file, "r") as f:
        for line in f.readlines():
            item_list = line.strip().split("\t")
            if len(item_list) > 1 and int(item_list[1]) < max_num:
                break
            id_to_word.append(item_list[0])
    print("Load word-dict with %d size and %d max_num." % (len(id_to_word), max_num))
    return id_to_word, len(id_to_word)


def load_word_counts(word_dict_file: str, vocab_size: int):
    """counts column of word_to_id.txt for the first vocab_size entries (specials without a count -> 0)"""
    counts = []
    with open(word_dict_file, "r") as f:
        for line in f:
            it = line.strip().split("\t")
            counts.append(int(it[1]) if len(it) > 1 else 0)
            if len(counts) == vocab_size:
                break
    return np.asarray(counts, np.int64)


def canonical_ids(id_to_word):
    """id -> first id spelling the same word (text equality == canonical-id equality); None if identity."""
    first = {}
    canon = np.arange(len(id_to_word), dtype=np.int32)
    ident = True
    for i, w in enumerate(id_to_word):
        j = first.setdefault(w, i)
        if j != i:
            canon[i] = j
            ident = False
    return None if ident else canon


def id2text_sentence(sen_id, id_to_word):
    """data.py:43-52."""
    sen_text = []
    max_i = len(id_to_word)
    for i in sen_id:
        i = int(i)
        if i == 3 or i == 0:
            break
        if i >= max_i:
            i = 1
        sen_text.append(id_to_word[i])
    return " ".join(sen_text)


def _clip(ids, vocab_size, unk, max_seq_length):
    """pad_batch_seuqences (data.py:119-152): ids >= V -> UNK, at most max_seq_length-1 tokens."""
    return [t if t < vocab_size else unk for t in ids[: max_seq_length - 1]]


class AttackSet:
    """All sentences of the attack run in the reference's order: file0 (label 0) then file1 (label 1)
    (siamese_contrastive.py:249-259; --reverse_order flips the files)."""

    def __init__(self, origin, like, unlike, labels=None):
        self.origin = origin      # list[list[int]]
        self.like = like          # list[list[list[int]]]  (K+ per sentence; [] when pos_num == 0)
        self.unlike = unlike
        self.labels = labels
        self.n_pos = len(like[0]) if like else 0
        self.n_neg = len(unlike[0]) if unlike else 0

    def __len__(self):
        return len(self.origin)

    @classmethod
    def from_json(cls, files, vocab_size, unk=1, max_seq_length=60, pos_num=None, neg_num=None):
        origin, like, unlike, labels = [], [], [], []
        for fn in files:
            with open(fn, "r") as fin:
                d = json.load(fin)
            ref_num = d["ref_num"] if neg_num is None else neg_num   # data.py:342-343
            sim_num = d["sim_num"] if pos_num is None else pos_num
            for rec in d["data"][: d["sent_num"]]:
                parse = lambda s: _clip([int(x) for x in s.strip().split()], vocab_size, unk, max_seq_length)
                origin.append(parse(rec["origin"]))
                like.append([parse(rec["sim%d" % j]) for j in range(sim_num)])
                unlike.append([parse(rec["ref%d" % j]) for j in range(ref_num)])
                labels.append(rec.get("label", -1))
        return cls(origin, like, unlike, labels)

    @classmethod
    def synthetic(cls, n, vocab_size, shape="yelp", n_pos=10, n_neg=10, seed=1234, word_counts=None):
        """SURVEY.md §8d synthetic sets: yelp-shaped origins 6-15 tokens (mean ~10.3), refs/sims 1-16;
        amazon-shaped origins 8-15, refs/sims 8-31; ids uniform in [4, V), or — when word_counts (the
        counts column of word_to_id.txt) is given — drawn from that unigram (Zipfian) distribution."""
        rng = np.random.default_rng(seed)
        if word_counts is not None:
            pr = np.asarray(word_counts[4:vocab_size], np.float64)
            pr = pr / pr.sum()
            pool = (rng.choice(len(pr), size=1 << 22, p=pr) + 4).astype(np.int64)
            cursor = [0]

            class _Z:
                @staticmethod
                def integers(lo, hi, size=None):
                    if lo == 4 and hi == vocab_size:
                        k = int(size)
                        if cursor[0] + k > len(pool):
                            cursor[0] = 0
                        out = pool[cursor[0]:cursor[0] + k]
                        cursor[0] += k
                        return out
                    return rng.integers(lo, hi, size)

                normal = rng.normal
            tok = _Z
        else:
            tok = rng
        if shape == "yelp":
            olen = np.clip(rng.normal(10.3, 2.6, n).round().astype(int), 6, 15)
            rlo, rhi = 1, 16
        else:
            olen = rng.integers(8, 16, n)
            rlo, rhi = 8, 31
        origin = [tok.integers(4, vocab_size, l).tolist() for l in olen]
        mk = lambda k: [[tok.integers(4, vocab_size, rng.integers(rlo, rhi + 1)).tolist() for _ in range(k)] for _ in range(n)]
        return cls(origin, mk(n_pos), mk(n_neg))

    def subset(self, idx):
        return AttackSet([self.origin[i] for i in idx], [self.like[i] for i in idx], [self.unlike[i] for i in idx],
                         None if self.labels is None else [self.labels[i] for i in idx])


def pack_sentences(sents):
    """list of id lists -> (tokens, tok_pos, sent_off, max_len) int32 arrays; id 0 inside a sentence is
    a masked position in the reference (src_mask = src != 0) and is dropped here with positions kept."""
    n = len(sents)
    lens = np.fromiter((len(s) for s in sents), np.int64, n)
    total = int(lens.sum())
    tokens = np.fromiter((t for s in sents for t in s), np.int32, total)
    tok_pos = (np.arange(total) - np.repeat(np.cumsum(lens) - lens, lens)).astype(np.int32)
    sent_id = np.repeat(np.arange(n), lens)
    keep = tokens != 0
    if not keep.all():
        tokens, tok_pos, sent_id = tokens[keep], tok_pos[keep], sent_id[keep]
        lens = np.bincount(sent_id, minlength=n)
    sent_off = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=sent_off[1:])
    return tokens, tok_pos, sent_off, int(lens.max()) if n else 0
