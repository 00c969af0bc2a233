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


def last_ids(id_to_word):
    """id -> LAST id spelling the same word: what text -> `word_to_id[w]` gives when the dictionary was filled in id
    order (main.py:194-197; --retext re-encodes through it, siamese_contrastive.py:746-751).  None if identity."""
    last = {w: i for i, w in enumerate(id_to_word)}
    m = np.fromiter((last[w] for w in id_to_word), np.int32, len(id_to_word))
    return None if (m == np.arange(len(id_to_word))).all() else m


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


class Ragged:
    """n sentences as one flat int32 token array + offsets: the host-side form of every sentence list
    (a million origins with 20 references each are 21 M sentences: Python lists of lists do not scale)."""

    __slots__ = ("tokens", "off")

    def __init__(self, tokens, off):
        self.tokens = np.ascontiguousarray(tokens, np.int32)
        self.off = np.ascontiguousarray(off, np.int64)
        assert self.off.ndim == 1 and self.off.size >= 1 and self.off[0] == 0 and self.off[-1] == self.tokens.size

    @classmethod
    def from_lists(cls, sents):
        n = len(sents)
        lens = np.fromiter((len(x) for x in sents), np.int64, n)
        off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        tokens = np.fromiter((t for x in sents for t in x), np.int32, int(off[-1]))
        return cls(tokens, off)

    @classmethod
    def from_lengths(cls, tokens, lens):
        off = np.zeros(len(lens) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return cls(tokens, off)

    def __len__(self):
        return self.off.size - 1

    @property
    def lens(self):
        return np.diff(self.off)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self.tokens[self.off[i]:self.off[i + 1]].tolist()

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def take(self, idx):
        """the sentences idx (any order, repeats allowed) as a new Ragged, without a Python loop"""
        idx = np.asarray(idx, np.int64).reshape(-1)
        lens = self.off[idx + 1] - self.off[idx] if idx.size else np.zeros(0, np.int64)
        off = np.zeros(idx.size + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        src = np.repeat(self.off[idx] - off[:-1], lens) + np.arange(int(off[-1]), dtype=np.int64)
        return Ragged(self.tokens[src], off)

    def span(self, lo, hi):
        """sentences [lo, hi) (contiguous: a view of the tokens)"""
        a, b = self.off[lo], self.off[hi]
        return Ragged(self.tokens[a:b], self.off[lo:hi + 1] - a)

    @staticmethod
    def concat(parts):
        parts = [p for p in parts if len(p)]
        if not parts:
            return Ragged(np.zeros(0, np.int32), np.zeros(1, np.int64))
        tokens = np.concatenate([p.tokens for p in parts])
        offs, base = [np.zeros(1, np.int64)], 0
        for p in parts:
            offs.append(p.off[1:] + base)
            base += int(p.off[-1])
        return Ragged(tokens, np.concatenate(offs))


class _Groups:
    """like / unlike as the reference indexes them: groups[s] = the K sentences of origin s (list of id lists)"""

    def __init__(self, rag, k):
        self.rag, self.k = rag, k

    def __len__(self):
        return len(self.rag) // self.k if self.k else 0

    def __getitem__(self, s):
        if s < 0:
            s += len(self)
        return [self.rag[s * self.k + j] for j in range(self.k)]

    def __iter__(self):
        return (self[s] for s in range(len(self)))


def _as_ragged(x):
    return x if isinstance(x, Ragged) else Ragged.from_lists(x)


class AttackSet:
    """All sentences of the attack run in the reference's order: file0 (label 0) then file1 (label 1)
    (siamese_contrastive.py:249-259; --reverse_order flips the files).  Stored as three Ragged arrays:
    origins [S], like [S * K+] and unlike [S * K-] (sentence j of origin s at s * K + j)."""

    def __init__(self, origin, like, unlike, labels=None, n_pos=None, n_neg=None):
        """origin: list of id lists or Ragged; like / unlike: list (per origin) of K id lists, or a Ragged of
        S * K sentences together with n_pos / n_neg."""
        self.origin_r = _as_ragged(origin)
        if isinstance(like, Ragged):
            self.n_pos, self.like_r = int(n_pos), like
        else:
            self.n_pos = len(like[0]) if len(like) else 0
            self.like_r = Ragged.from_lists([x for grp in like for x in grp])
        if isinstance(unlike, Ragged):
            self.n_neg, self.unlike_r = int(n_neg), unlike
        else:
            self.n_neg = len(unlike[0]) if len(unlike) else 0
            self.unlike_r = Ragged.from_lists([x for grp in unlike for x in grp])
        assert len(self.like_r) == len(self) * self.n_pos and len(self.unlike_r) == len(self) * self.n_neg
        self.labels = labels

    def __len__(self):
        return len(self.origin_r)

    # the reference's view of the data (lists of id lists); only small cases and the tests walk these
    @property
    def origin(self):
        return self.origin_r

    @property
    def like(self):
        return _Groups(self.like_r, self.n_pos)

    @property
    def unlike(self):
        return _Groups(self.unlike_r, self.n_neg)

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
    def synthetic(cls, n, vocab_size, shape="yelp", n_pos=10, n_neg=10, seed=1234, word_counts=None, origins_only=False):
        """SURVEY.md §8d synthetic sets: yelp-shaped origins 6-15 tokens (mean ~10.3), refs/sims 1-16;
        amazon-shaped origins 8-15, refs/sims 8-31; ids uniform in [4, V), or — when word_counts (the
        counts column of word_to_id.txt) is given — drawn from that unigram (Zipfian) distribution.
        Fully vectorised: a million origins (21 M sentences, ~200 M tokens) take seconds."""
        rng = np.random.default_rng(seed)
        if shape == "yelp":
            olen = np.clip(rng.normal(10.3, 2.6, n).round().astype(np.int64), 6, 15)
            rlo, rhi = 1, 16
        else:
            olen = rng.integers(8, 16, n).astype(np.int64)
            rlo, rhi = 8, 31
        plen = rng.integers(rlo, rhi + 1, n * n_pos).astype(np.int64)
        nlen = rng.integers(rlo, rhi + 1, n * n_neg).astype(np.int64)
        if word_counts is not None:
            pr = np.asarray(word_counts[4:vocab_size], np.float64)
            cdf = np.cumsum(pr / pr.sum())
            cdf[-1] = 1.0
            # a 4 M-token pool drawn exactly from the unigram distribution, resampled by index (a searchsorted over
            # 200 M uniforms would take 20 s; the resample of an i.i.d. pool is i.i.d. from its empirical distribution)
            pool = (np.searchsorted(cdf, rng.random(1 << 22), side="right") + 4).astype(np.int32)
            draw = lambda k: pool[rng.integers(0, pool.size, k, dtype=np.int32)]
        else:
            draw = lambda k: rng.integers(4, vocab_size, k, dtype=np.int32)
        mk = lambda lens: Ragged.from_lengths(draw(int(lens.sum())), lens)
        if origins_only:     # the same origins as the full set (they are drawn first), without the 20 references each
            return cls(mk(olen), Ragged(np.zeros(0, np.int32), np.zeros(1, np.int64)), Ragged(np.zeros(0, np.int32), np.zeros(1, np.int64)), None, 0, 0)
        return cls(mk(olen), mk(plen), mk(nlen), None, n_pos, n_neg)

    def subset(self, idx):
        idx = np.asarray(idx, np.int64).reshape(-1)
        grp = lambda k: (idx[:, None] * k + np.arange(k)[None, :]).reshape(-1)
        return AttackSet(self.origin_r.take(idx), self.like_r.take(grp(self.n_pos)), self.unlike_r.take(grp(self.n_neg)),
                         None if self.labels is None else [self.labels[i] for i in idx], self.n_pos, self.n_neg)

    def span(self, lo, hi):
        """origins [lo, hi) with their references (contiguous shard or batch: views, no copies)"""
        return AttackSet(self.origin_r.span(lo, hi), self.like_r.span(lo * self.n_pos, hi * self.n_pos),
                         self.unlike_r.span(lo * self.n_neg, hi * self.n_neg),
                         None if self.labels is None else self.labels[lo:hi], self.n_pos, self.n_neg)

    def all_sentences(self):
        """every sentence in encoder order: all origins, all like groups, all unlike groups"""
        return Ragged.concat([self.origin_r, self.like_r, self.unlike_r])


def pack_sentences(sents):
    """id lists (or a Ragged) -> (tokens, tok_pos, sent_off, max_len) int32 arrays; id 0 inside a sentence is
    a masked position in the reference (src_mask = src != 0) and is dropped here with positions kept."""
    rag = _as_ragged(sents)
    n = len(rag)
    lens = rag.lens
    tokens = rag.tokens
    total = tokens.size
    tok_pos = (np.arange(total, dtype=np.int64) - np.repeat(rag.off[:-1], lens)).astype(np.int32)
    keep = tokens != 0
    if not keep.all():
        sent_id = np.repeat(np.arange(n), lens)
        tokens, tok_pos, sent_id = tokens[keep], tok_pos[keep], sent_id[keep]
        lens = np.bincount(sent_id, minlength=n)
    sent_off = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=sent_off[1:])
    return np.ascontiguousarray(tokens, np.int32), tok_pos, sent_off, int(lens.max()) if n else 0
