"""The gradient-guided attack driver, batch-first (replaces attack_contrastive +
contrastive_attack_one, /root/reference/method/yelp/siamese_contrastive.py:672-839).

The reference walks the 1000 test sentences one at a time; every sentence is independent (nets in
eval mode, fresh Adam per sentence, :689-691, :780), so here a batch of B sentences moves through
the same steps together, each with its own Adam state, change counter and stop flag:

  encode ori/like/unlike      -> st_encode_latent    (one packed launch sequence for all 21*B sentences)
  like_z / unlike_z           -> st_scorer_embed     (once; the reference recomputes them every iteration)
  [Rec/Sim/Ref log decodes]   -> st_decode_greedy    (optional, only needed for the log lines)
  for it in range(max_iter):
      score, backward, Adam   -> st_scorer_edit_step (all B rows, inactive rows masked)
      greedy decode           -> st_decode_greedy    (live rows only, compacted)
      text change bookkeeping -> st_accept_changes
"""
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import ops
from .data import AttackSet, id2text_sentence, pack_sentences
from .logfmt import headline


@dataclass
class AttackConfig:
    """The attack-relevant flags of main.py:35-126 (same names and defaults)."""
    id_pad: int = 0
    id_unk: int = 1
    id_bos: int = 2
    id_eos: int = 3
    max_sequence_length: int = 60
    tau: float = 0.1
    victim_optimizer: str = "Adam"
    victim_lr: float = 0.25
    attack_max_iter: int = 18
    attack_max_change: int = 7
    simplest_siamese: bool = False
    just_use_one: bool = False
    not_nll: bool = False
    retext: bool = False
    dance: bool = False
    pos_victim_lr: float = 0.25
    neg_victim_lr: float = 0.25
    # build-side knobs (no reference counterpart)
    early_exit: bool = True        # stop decoding a sentence at its first EOS/PAD (output-equivalent)
    sync_every: int = 3            # decode steps between two host reads of the live count (kernels read it on the
                                   # device every step; a stale, larger bound only launches workgroups that exit at once)
    log_decodes: bool = True       # the 1+10+10 decodes behind the Rec:/Sim:/Ref: log lines
    want_gap: bool = False
    want_trace: bool = False       # keep latents / grads / losses per iteration (parity tests)
    # edit iterations whose latents are decoded together as one batch: an int (every round that many) or a
    # schedule of round sizes, e.g. (9, 3, 3, 3): later rounds are shorter because sentences that reach their 7th
    # change inside a round have the rest of that round's decodes discarded (44 % of yelp-shaped sentences run all
    # 18 iterations, the others stop anywhere from iteration 7 on)
    iters_per_decode: object = (12, 6)
    # Small batches are latency-bound (a decode step costs the same for 300 rows as for 5000), so halving the number of
    # decode steps beats not decoding iterations that may be discarded: up to this many sentences ALL iterations are
    # decoded in one round (batch 256: 12.9 k -> 20.8 k sentences/s with this alone; results identical by construction)
    single_round_max_batch: int = 8192
    merge_log_decodes_max_bytes: float = 150e9   # decode workspace up to which the log decodes join round 1 (121 / 241 KB per job)
    defer_stragglers: bool = True  # see run_prepared: sentences whose decodes do not end are finished in an extra round
    split_by_length: bool = True   # see prepare: short and long sentences of a mixed chunk are encoded apart
    count_units: bool = False

    @classmethod
    def from_namespace(cls, ns, **over):
        kw = {k: getattr(ns, k) for k in cls.__dataclass_fields__ if hasattr(ns, k)}
        kw.update(over)
        return cls(**kw)


def sched_first(c, I, B):
    """edit iterations decoded in the first round (AttackRunner.run_prepared's schedule)"""
    if c.retext:
        return 1
    if isinstance(c.iters_per_decode, int):
        return max(1, min(c.iters_per_decode, I))
    if B <= c.single_round_max_batch:
        return I
    return max(1, min(int(c.iters_per_decode[0]), I))


STRAGGLER_LEN = 32   # a decode job with no EOS/PAD among its first 32 tokens marks its sentence as a straggler


@dataclass
class AttackResult:
    """Per-sentence records, all numpy on the host; enough to write the reference's log."""
    n_iter: np.ndarray            # [S] iterations run (<= max_iter)
    init_ids: np.ndarray          # [S,T] decode of the un-edited latent (the 'Rec:' line)
    iter_ids: np.ndarray          # [S,I,T] accepted text after each iteration
    changed: np.ndarray           # [S,I] 1 where a 'Change.k' line precedes 'iteration.it'
    like_ids: np.ndarray = None   # [S,K+,T] 'Sim{i}:' decodes (log_decodes)
    unlike_ids: np.ndarray = None
    final_latent: np.ndarray = None
    trace: dict = field(default_factory=dict)
    timing: dict = field(default_factory=dict)
    device: dict = None           # run_prepared(fetch=False): the same records as device tensors
    dance: dict = None            # --dance: loss [S,I], scores [S,I,K+ + K-], n_pos, not_nll


@dataclass
class LeanRecord:
    """The per-sentence records in the compact form that leaves the GPU and crosses ranks: the accepted text only
    changes where `changed` is set, so instead of the dense [S,I,T] table (4.2 KB per sentence) only the rows of the
    changes travel (<= attack_max_change rows of 236 B per sentence).  Fields are torch tensors on one device
    (GPU after a run, CPU after .cpu()); expand() rebuilds the dense AttackResult."""
    n_iter: torch.Tensor           # [S] int32
    init_ids: torch.Tensor         # [S,T] int32
    changed: torch.Tensor          # [S,I] uint8
    rows: torch.Tensor             # [sum(changed),T] int32, sentence-major then iteration
    like_ids: torch.Tensor = None  # [S,K+,T] (log_decodes)
    unlike_ids: torch.Tensor = None
    final_latent: torch.Tensor = None

    FIELDS = ("n_iter", "init_ids", "changed", "rows", "like_ids", "unlike_ids", "final_latent")

    def __len__(self):
        return int(self.n_iter.shape[0])

    @classmethod
    def from_device(cls, d, want_latent=False):
        """d: AttackResult.device of run_prepared(fetch=False)"""
        n_iter = d["was_active"].sum(0, dtype=torch.int32)
        I = d["changed"].shape[0]
        changed = d["changed"].t().contiguous()
        changed = changed * (torch.arange(I, device=changed.device)[None, :] < n_iter[:, None]).to(torch.uint8)
        rows = d["iter_ids"].permute(1, 0, 2)[changed.bool()].contiguous()
        return cls(n_iter, d["init_ids"], changed, rows, d.get("like_ids"), d.get("unlike_ids"),
                   d["latent"] if want_latent else None)

    def to(self, device, pinned=False):
        def mv(t):
            if t is None:
                return None
            if pinned and device == "cpu" and t.is_cuda:
                buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                buf.copy_(t, non_blocking=True)
                return buf
            return t.to(device)
        out = LeanRecord(*[mv(getattr(self, f)) for f in self.FIELDS])
        if pinned and device == "cpu":
            torch.cuda.current_stream().synchronize()
        return out

    def cpu(self):
        return self.to("cpu")

    def span(self, lo, hi, row_start=None):
        """sentences [lo, hi) of this record (views); row_start [S+1]: prefix sums of the changed counts, computed
        here when not given (pass it when cutting one record into many spans)"""
        if row_start is None:
            row_start = torch.zeros(len(self) + 1, dtype=torch.int64)
            row_start[1:] = torch.cumsum(self.changed.sum(1, dtype=torch.int64).cpu(), 0)
        cut = lambda t: None if t is None else t[lo:hi]
        return LeanRecord(self.n_iter[lo:hi], self.init_ids[lo:hi], self.changed[lo:hi],
                          self.rows[int(row_start[lo]):int(row_start[hi])], cut(self.like_ids), cut(self.unlike_ids),
                          cut(self.final_latent))

    @staticmethod
    def cat(parts):
        cat = lambda xs: None if any(x is None for x in xs) else torch.cat(xs, 0)
        return LeanRecord(*[cat([getattr(p, f) for p in parts]) for f in LeanRecord.FIELDS])

    @classmethod
    def empty(cls, cfg, kp, kn, device="cpu", want_latent=False):
        T = cfg.max_sequence_length - 1
        I = 2 if cfg.dance else cfg.attack_max_iter
        z = lambda *sh, dt=torch.int32: torch.zeros(sh, dtype=dt, device=device)
        ids = lambda k: z(0, k, T) if (cfg.log_decodes and k) else None
        return cls(z(0), z(0, T), z(0, I, dt=torch.uint8), z(0, T), ids(kp), ids(kn),
                   z(0, 256, dt=torch.float32) if want_latent else None)

    def expand(self) -> AttackResult:
        """the dense records (numpy): iter_ids[s,it] = the last changed row at or before it, else init_ids[s]"""
        r = self.cpu()
        n_iter, init, ch, rows = r.n_iter.numpy(), r.init_ids.numpy(), r.changed.numpy(), r.rows.numpy()
        S, I = ch.shape
        T = init.shape[1]
        cs = np.cumsum(ch.astype(np.int64), 1)
        start = np.zeros(S, np.int64)
        if S:
            np.cumsum(cs[:-1, -1], out=start[1:])
        if rows.shape[0]:
            picked = rows[np.clip(start[:, None] + cs - 1, 0, rows.shape[0] - 1)]
        else:
            picked = np.zeros((S, I, T), np.int32)
        it_ids = np.where((cs > 0)[:, :, None], picked, init[:, None, :])
        it_ids = np.where((np.arange(I)[None, :] < n_iter[:, None])[:, :, None], it_ids, 0).astype(np.int32)
        npy = lambda t: None if t is None else t.numpy()
        return AttackResult(n_iter, init, it_ids, ch, npy(r.like_ids), npy(r.unlike_ids), npy(r.final_latent))


class AttackRunner:
    def __init__(self, ae, scorer, cfg: AttackConfig, device):
        from ._lib import ST_OPTIMIZERS
        if cfg.victim_optimizer not in ST_OPTIMIZERS:     # the reference: getattr(optim, name), any torch.optim class (:780)
            raise NotImplementedError("--victim_optimizer %s: the edit-step kernel has %s (torch defaults apart from lr)"
                                      % (cfg.victim_optimizer, ", ".join(ST_OPTIMIZERS)))
        self.ae, self.scorer, self.cfg, self.device = ae, scorer, cfg, device
        self.ws = ops.DecodeWorkspace(device)
        self.canon = None

    def set_vocab(self, id_to_word):
        from .data import canonical_ids, last_ids
        c = canonical_ids(id_to_word)
        self.canon = None if c is None else torch.from_numpy(c).to(self.device)
        m = last_ids(id_to_word)      # --retext: a word spelled by several ids re-encodes as the last of them
        self.retext_map = None if m is None else torch.from_numpy(m).to(self.device)

    # -- pieces -----------------------------------------------------------------------------------
    ENC_CHUNK = 32768  # sentences per encoder launch sequence (bounds the [rows, d_ff] workspace)
    # The fused encoder kernels (QKV projection + attention in one launch, first layer from the (token, position) table:
    # csrc/st_encode.hip) take a launch whose sentences all have <= 16 tokens.  A chunk that mixes lengths (amazon:
    # references of up to 31 tokens, 65 % of them <= 16) is encoded as two launches, short and long sentences apart, and
    # the latents go to their rows by index; a sentence's latent does not depend on what shares its launch.
    SHORT_LEN = 16
    SPLIT_MIN = 1024   # short sentences a mixed chunk must hold before it is split

    def prepare(self, batch: AttackSet):
        """Host -> device: pack the 1 + K+ + K- sentences of every origin (order: all origins, all
        like groups, all unlike groups) into device token arrays, in chunks of ENC_CHUNK sentences.
        A chunk is (tokens, tok_pos, sent_off, n_sent, max_len, row0, dest): latents go to rows row0 .. row0 + n_sent - 1
        of the batch's latent array, or to rows `dest` (int64 device tensor) when the chunk is one length class of a
        mixed span."""
        sents = batch.all_sentences()            # one flat token array + offsets (data.Ragged): no per-sentence Python
        dev, chunks, host = self.device, [], []

        def h2d(x):
            """numpy -> device through a PINNED staging buffer that the returned dict keeps alive: the copy is asynchronous
            on the current stream and must not depend on how the runtime stages pageable memory"""
            src = torch.from_numpy(np.ascontiguousarray(x))
            pin = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
            pin.copy_(src)
            host.append(pin)
            return pin.to(dev, non_blocking=True)
        split_ok = self.cfg.split_by_length and self.ae.gemm_format == "f16x3"
        for c0 in range(0, len(sents), self.ENC_CHUNK):
            c1 = min(len(sents), c0 + self.ENC_CHUNK)
            span = sents.span(c0, c1)
            lens = span.lens
            short = lens <= self.SHORT_LEN
            n_short = int(short.sum())
            if split_ok and self.SPLIT_MIN <= n_short < c1 - c0:
                for idx in (np.nonzero(short)[0], np.nonzero(~short)[0]):
                    tokens, tok_pos, sent_off, max_len = pack_sentences(span.take(idx))
                    chunks.append((h2d(tokens), h2d(tok_pos), h2d(sent_off), int(idx.size), max_len, c0, h2d(idx.astype(np.int64) + c0)))
            else:
                tokens, tok_pos, sent_off, max_len = pack_sentences(span)
                chunks.append((h2d(tokens), h2d(tok_pos), h2d(sent_off), c1 - c0, max_len, c0, None))
        return dict(chunks=chunks, n_sent=len(sents), B=len(batch), kp=batch.n_pos, kn=batch.n_neg, _host=host)

    def encode_prepared(self, prep):
        cur = torch.cuda.current_stream(self.device)
        chunks = prep["chunks"]
        for c in chunks:      # a prefetch thread may have allocated them on its side stream
            for x in (c[0], c[1], c[2], c[6]):
                if x is not None:
                    x.record_stream(cur)
        if len(chunks) == 1:
            t, p, o, n, mx, _, _ = chunks[0]
            return ops.encode_latent(self.ae, t, p, o, n, mx, ws=self.ws)
        lat = torch.empty(prep["n_sent"], 256, dtype=torch.float32, device=self.device)   # (no torch.cat: 2 GB per 98 304-sentence batch)
        for (t, p, o, n, mx, row0, dest) in chunks:
            if dest is None:
                ops.encode_latent(self.ae, t, p, o, n, mx, ws=self.ws, out=lat[row0:row0 + n])
            else:
                lat.index_copy_(0, dest, ops.encode_latent(self.ae, t, p, o, n, mx, ws=self.ws))
        return lat

    def encode(self, sents):
        return self.encode_prepared(self.prepare(AttackSet(sents, [], [])))

    def decode(self, latent, want_gap=False):
        c = self.cfg
        return ops.greedy_decode(self.ae, latent, c.max_sequence_length, c.id_bos, c.id_eos, c.id_pad,
                                 early_exit=c.early_exit, sync_every=c.sync_every, want_gap=want_gap, ws=self.ws)

    def reencode(self, ids):
        """--retext (siamese_contrastive.py:842-855): decoded ids [n,T] -> text -> ids -> latent.  The text
        is the id prefix before the first EOS/PAD (data.py:43-52); ids go through `retext_map` (identity for the
        shipped vocabularies, which have no duplicate words); packing happens on the device (torch indexing = plumbing)."""
        c = self.cfg
        n, T = ids.shape
        term = (ids == c.id_eos) | (ids == c.id_pad)
        lens = torch.where(term.any(1), term.int().argmax(1), torch.full((n,), T, device=ids.device, dtype=torch.int64))
        ar = torch.arange(T, device=ids.device)
        mask = ar[None, :] < lens[:, None]
        tokens = ids[mask].to(torch.int32).contiguous()
        if getattr(self, "retext_map", None) is not None:
            tokens = self.retext_map[tokens.long().clamp_(0, self.retext_map.numel() - 1)].contiguous()
        tok_pos = ar.to(torch.int32)[None, :].expand(n, T)[mask].contiguous()
        off = torch.zeros(n + 1, dtype=torch.int32, device=ids.device)
        off[1:] = torch.cumsum(lens, 0).to(torch.int32)
        return ops.encode_latent(self.ae, tokens, tok_pos, off, n, int(lens.max().item()) if n else 0, ws=self.ws)

    # -- one batch --------------------------------------------------------------------------------
    def run_batch(self, batch: AttackSet, global_index) -> AttackResult:
        return self.run_prepared(self.prepare(batch), global_index)

    def run_prepared(self, prep, global_index, fetch=True) -> AttackResult:
        """global_index [B]: position of each sentence in the full run (lr switch at 500, :740).
        fetch=False leaves the records on the device (bench: no D2H inside the timed region)."""
        c, dev = self.cfg, self.device
        B, kp, kn = prep["B"], prep["kp"], prep["kn"]
        T = c.max_sequence_length - 1
        I = 2 if c.dance else c.attack_max_iter
        if c.not_nll and kp != kn and kp and kn:
            # the reference adds pos_loss [K+] to neg_loss [K-] (siamese_contrastive.py:429): a size-mismatch error
            raise ValueError("--not_nll needs pos_num == neg_num (got %d and %d), as in the reference" % (kp, kn))
        if B == 0:
            return empty_result(c, kp, kn)
        tm = {}
        t0 = time.perf_counter()
        lat_all = self.encode_prepared(prep)
        ori = lat_all[:B].clone()
        like_lat = lat_all[B:B + B * kp]
        unlike_lat = lat_all[B + B * kp:]
        like_z = ops.scorer_embed(self.scorer, like_lat).view(B, kp, 256) if kp else None
        unlike_z = ops.scorer_embed(self.scorer, unlike_lat).view(B, kn, 256) if kn else None
        res = AttackResult(None, None, None, None)
        like_ids_dev = unlike_ids_dev = None
        # The Rec / Sim / Ref log lines need the decodes of the 1 + K+ + K- encoder latents (siamese_contrastive.py:713-732).
        # They ride along in the FIRST decode round of the edit loop (one call of (rounds + 1 + K+ + K-) B jobs instead of
        # two calls with a straggler tail each) unless that would not fit the K/V cache budget.
        n_first = (sched_first(c, I, B) + 1 + kp + kn) * B
        merge_log = (c.log_decodes and (kp or kn) and not (c.retext or c.want_trace)
                     and ops.decode_workspace_bytes(self.ae, n_first, c.max_sequence_length) <= c.merge_log_decodes_max_bytes)
        if c.log_decodes and (kp or kn) and not merge_log:
            ref_ids, _, _ = self.decode(lat_all[B:])
            like_ids_dev = ref_ids[:B * kp].view(B, kp, T) if kp else None
            unlike_ids_dev = ref_ids[B * kp:].view(B, kn, T) if kn else None
        gidx = global_index if torch.is_tensor(global_index) else torch.as_tensor(np.asarray(global_index), device=dev)
        lr = torch.where(gidx < 500, torch.tensor(c.pos_victim_lr, device=dev), torch.tensor(c.neg_victim_lr, device=dev)).float().contiguous()
        m = torch.zeros(B, 256, device=dev)
        v = torch.zeros(B, 256, device=dev)
        step = torch.zeros(B, dtype=torch.int32, device=dev)
        active = torch.ones(B, dtype=torch.uint8, device=dev)
        n_changes = torch.zeros(B, dtype=torch.int32, device=dev)
        changed = torch.zeros(I, B, dtype=torch.uint8, device=dev)
        was_active = torch.zeros(I, B, dtype=torch.uint8, device=dev)
        iter_ids = torch.zeros(I, B, T, dtype=torch.int32, device=dev)
        new_ids = torch.zeros(B, T, dtype=torch.int32, device=dev)
        final_latent = ori.clone()
        init_ids = acc_ids = init_gap = None
        want_gap = c.want_gap or c.want_trace
        tr = {k: [] for k in ("latent", "grad", "loss", "ids", "gap")} if c.want_trace else None
        max_change = (1 << 30) if c.dance else c.attack_max_change
        dance_rec = []   # --dance: per iteration (loss [B], scores [B, kp + kn])
        if c.retext:
            sched = [1] * I                   # --retext: the latent itself depends on the decodes
        elif isinstance(c.iters_per_decode, int):
            sched = [max(1, min(c.iters_per_decode, I))] * I
        else:
            sched = [max(1, int(v)) for v in c.iters_per_decode] + [1] * I
            if B <= c.single_round_max_batch:
                sched = [I]
        torch.cuda.synchronize(dev)
        tm["setup_s"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        n_units = 0
        # The latent trajectory x_0 -> x_1 -> ... does not depend on the decodes (they only feed the change
        # counter and the stop rule, siamese_contrastive.py:812-839), so C edit steps are taken first and
        # their C latents (plus x_0 in the first round) are decoded as ONE batch; the change bookkeeping
        # then replays the iterations in order.  Iterations past a sentence's stop point are discarded.
        state = dict(init_ids=None, acc_ids=None, init_gap=None, n_units=0, deferred=None)
        # Straggler deferral: a handful of decode jobs never emit EOS and keep ~40 almost-empty decode steps alive at the
        # end of EVERY round.  After the first round the sentences that own such jobs are set aside (sentences are
        # independent), the other rounds run without those tails, and the set-aside sentences finish their remaining
        # iterations in one extra round at the end.  Not used where the per-iteration records are taken in lockstep.
        defer_ok = c.defer_stragglers and not (c.retext or c.dance or c.want_trace) and len([v for v in sched[:I] if v > 1]) > 1

        def rounds(sel, it0, sched):
            """Run decode rounds for the sentences selected by `sel` (bool [B], None = all) from iteration it0 on."""
            for C in sched:
                if it0 >= I:
                    break
                mask = active if sel is None else (active & sel.to(torch.uint8))
                live = torch.nonzero(mask, as_tuple=False).flatten()
                n_live = int(live.numel())
                if n_live == 0:
                    break
                cn = min(C, I - it0)
                snaps = [ori.index_select(0, live)] if it0 == 0 else []
                for j in range(cn):
                    out = ops.scorer_edit_step(self.scorer, ori, m, v, like_z, unlike_z, lr, step, mask, tau=c.tau,
                                               not_nll=c.not_nll, simplest=c.simplest_siamese,
                                               just_use_one=c.just_use_one, want_loss=c.want_trace or c.dance,
                                               want_grad=c.want_trace, want_scores=c.dance, optimizer=c.victim_optimizer)
                    loss, grad = out[0], out[1]
                    if c.dance:
                        dance_rec.append((loss.cpu().numpy(), out[2].cpu().numpy()))
                    snaps.append(ori.index_select(0, live))
                    if tr is not None:
                        tr["latent"].append(ori.detach().cpu().numpy().copy())
                        tr["grad"].append(grad.cpu().numpy())
                        tr["loss"].append(loss.cpu().numpy())
                extra = lat_all[B:] if (merge_log and it0 == 0) else None
                ids, gap, _ = self.decode(torch.cat(snaps + ([extra] if extra is not None else []), 0), want_gap=want_gap)
                if extra is not None:
                    ref_ids = ids[len(snaps) * n_live:]
                    state["like_ids"] = ref_ids[:B * kp].view(B, kp, T) if kp else None
                    state["unlike_ids"] = ref_ids[B * kp:].view(B, kn, T) if kn else None
                    ids = ids[:len(snaps) * n_live]
                    gap = gap[:len(snaps) * n_live] if gap is not None else None
                ids = ids.view(len(snaps), n_live, T)
                gap = gap.view(len(snaps), n_live, T) if gap is not None else None
                k0 = 0
                if it0 == 0:
                    state["init_ids"] = torch.zeros(B, T, dtype=torch.int32, device=dev)
                    state["init_ids"].index_copy_(0, live, ids[0])
                    state["acc_ids"] = state["init_ids"].clone()
                    if gap is not None:
                        state["init_gap"] = torch.zeros(B, T, device=dev).index_copy_(0, live, gap[0])
                    k0 = 1
                    if defer_ok:
                        ended = ((ids == c.id_eos) | (ids == c.id_pad))[:, :, :STRAGGLER_LEN].any(-1)   # [jobs of a sentence, n_live]
                        strag = (~ended).any(0)
                        if 0 < int(strag.sum().item()) <= max(64, B // 64):
                            state["deferred"] = torch.zeros(B, dtype=torch.bool, device=dev)
                            state["deferred"][live] = strag
                acc_ids = state["acc_ids"]
                re_ids = re_lat = None
                if c.retext:
                    re_lat = self.reencode(ids[k0])
                    re_ids_live, _, _ = self.decode(re_lat)
                    re_ids = torch.zeros(B, T, dtype=torch.int32, device=dev)
                    re_ids.index_copy_(0, live, re_ids_live)
                for j in range(cn):
                    it = it0 + j
                    was_active[it] |= mask
                    state["n_units"] += int(mask.sum().item()) if c.count_units else 0
                    new_ids.index_copy_(0, live, ids[k0 + j])
                    still = mask.bool()
                    final_latent[live] = torch.where(still[live, None], snaps[k0 + j], final_latent[live])
                    ch = changed[it] if sel is None else torch.zeros(B, dtype=torch.uint8, device=dev)
                    rec = iter_ids[it] if sel is None else torch.zeros(B, T, dtype=torch.int32, device=dev)
                    ops.accept_changes(new_ids, acc_ids, self.ae.vocab, n_changes, ch, mask, max_change=max_change,
                                       unk=c.id_unk, eos=c.id_eos, pad=c.id_pad, canon=self.canon, rec_ids=rec,
                                       confirm_ids=re_ids)
                    if sel is not None:   # fold this pass's rows back into the per-iteration records
                        changed[it] = torch.where(sel, ch, changed[it])
                        iter_ids[it] = torch.where(sel[:, None], rec, iter_ids[it])
                        active.copy_(torch.where(sel, mask, active))
                    if c.retext:
                        # accepted: ori_latent.data = re_latent.data (siamese_contrastive.py:826); Adam state carries on
                        acc_live = changed[it].bool()[live]
                        ori[live] = torch.where(acc_live[:, None], re_lat, ori[live])
                        final_latent[live] = torch.where(acc_live[:, None], re_lat, final_latent[live])
                    if tr is not None:
                        tr["ids"].append(new_ids.cpu().numpy().copy())
                        tr["gap"].append(torch.zeros(B, T, device=dev).index_copy_(0, live, gap[k0 + j]).cpu().numpy())
                it0 += cn
                if it0 == cn and state["deferred"] is not None:
                    return it0   # first round done and stragglers found: the caller splits the batch
            return it0

        it_done = rounds(None, 0, sched)
        if state["deferred"] is not None:
            rounds(~state["deferred"], it_done, sched[1:])
            rounds(state["deferred"], it_done, [I])
        init_ids, acc_ids, init_gap, n_units = state["init_ids"], state["acc_ids"], state["init_gap"], state["n_units"]
        if merge_log:
            like_ids_dev, unlike_ids_dev = state.get("like_ids"), state.get("unlike_ids")
        tm["deferred_sentences"] = 0 if state["deferred"] is None else int(state["deferred"].sum().item())
        if not c.count_units:
            n_units = int(was_active.sum().item())
        torch.cuda.synchronize(dev)
        if self.scorer is not None and self.scorer.gemm_format == "f16x3":
            self.scorer.check_range()
        tm["loop_s"] = time.perf_counter() - t1
        tm["units"] = n_units
        res.timing = tm
        if not fetch:
            res.device = dict(was_active=was_active, iter_ids=iter_ids, changed=changed, init_ids=init_ids, latent=final_latent,
                              like_ids=like_ids_dev, unlike_ids=unlike_ids_dev)
            return res
        res.n_iter = was_active.sum(0).cpu().numpy().astype(np.int32)
        res.init_ids = init_ids.cpu().numpy()
        res.iter_ids = iter_ids.permute(1, 0, 2).contiguous().cpu().numpy()
        res.changed = changed.t().contiguous().cpu().numpy()
        res.final_latent = final_latent.cpu().numpy()
        res.like_ids = None if like_ids_dev is None else like_ids_dev.cpu().numpy()
        res.unlike_ids = None if unlike_ids_dev is None else unlike_ids_dev.cpu().numpy()
        if c.dance:
            res.dance = dict(loss=np.stack([d[0] for d in dance_rec], 1), scores=np.stack([d[1] for d in dance_rec], 1),
                             n_pos=kp, not_nll=c.not_nll)
        if tr is not None:
            res.trace = {k: np.stack(vv, 1) if vv else None for k, vv in tr.items()}
            res.trace["init_gap"] = None if init_gap is None else init_gap.cpu().numpy()
            res.trace["init_latent"] = lat_all[:B].cpu().numpy()
            res.trace["like_latent"] = like_lat.view(B, kp, 256).cpu().numpy() if kp else None
            res.trace["unlike_latent"] = unlike_lat.view(B, kn, 256).cpu().numpy() if kn else None
        res.timing = tm
        return res


def empty_result(cfg: AttackConfig, kp: int, kn: int) -> AttackResult:
    """Records of zero sentences with the shapes a non-empty run would have (an empty shard of a multi-GPU run:
    every rank must present the same fields to the gather)."""
    T = cfg.max_sequence_length - 1
    I = 2 if cfg.dance else cfg.attack_max_iter
    ids = lambda k: np.zeros((0, k, T), np.int32) if (cfg.log_decodes and k) else None
    return AttackResult(np.zeros(0, np.int32), np.zeros((0, T), np.int32), np.zeros((0, I, T), np.int32),
                        np.zeros((0, I), np.uint8), ids(kp), ids(kn), final_latent=np.zeros((0, 256), np.float32),
                        timing=dict(units=0, setup_s=0.0, loop_s=0.0))


def concat_results(parts):
    cat = lambda xs: None if any(x is None for x in xs) else np.concatenate(xs, 0)
    r = AttackResult(cat([p.n_iter for p in parts]), cat([p.init_ids for p in parts]), cat([p.iter_ids for p in parts]),
                     cat([p.changed for p in parts]), cat([p.like_ids for p in parts]), cat([p.unlike_ids for p in parts]),
                     cat([p.final_latent for p in parts]))
    r.timing = {k: sum(p.timing.get(k, 0) for p in parts) for k in ("setup_s", "loop_s", "units", "deferred_sentences")}
    if parts and getattr(parts[0], "dance", None):
        r.dance = dict(parts[0].dance, loss=cat([p.dance["loss"] for p in parts]), scores=cat([p.dance["scores"] for p in parts]))
    if parts and parts[0].trace:
        r.trace = {k: cat([p.trace[k] for p in parts]) for k in parts[0].trace}
    return r


def dance_digest(scores, n_pos, not_nll):
    """The second value get_contrastive_loss returns (printed by --dance, siamese_contrastive.py:834):
    --not_nll: #(pos_score > 0) + #(neg_score < 0) (:433); otherwise, per positive i, the argmax of
    [pos_i, neg_0, ...] as a one-element list (:443-444)."""
    pos, neg = scores[:n_pos], scores[n_pos:]
    if not_nll:
        return int((-pos < 0).sum() + (neg < 0).sum())
    return [[int(np.argmax(np.concatenate([pos[i:i + 1], neg])))] for i in range(n_pos)]


def write_attack_log(log, data: AttackSet, res: AttackResult, id_to_word, first_index=0, dance=False):
    """The per-sentence block of the reference's log, in global order (SURVEY.md Appendix A item 8;
    siamese_contrastive.py:702-732, :789, :825-836)."""
    for s in range(len(data)):
        log(headline("Sentence Pair.{}".format(first_index + s)))
        log("Src: " + id2text_sentence(data.origin[s], id_to_word))
        log("Rec: " + id2text_sentence(res.init_ids[s], id_to_word))
        if res.like_ids is not None:
            for i in range(res.like_ids.shape[1]):
                log("Sim{}: ".format(i) + id2text_sentence(res.like_ids[s, i], id_to_word))
        if res.unlike_ids is not None:
            for i in range(res.unlike_ids.shape[1]):
                log("Ref{}: ".format(i) + id2text_sentence(res.unlike_ids[s, i], id_to_word))
        log(headline("", note_token="-"))
        k = 0
        for it in range(int(res.n_iter[s])):
            text = id2text_sentence(res.iter_ids[s, it], id_to_word)
            if res.changed[s, it] and not dance:
                log("Change.{}, {}".format(k, text))
                k += 1
            if dance:
                d = res.dance
                # `loss = -(loss)` before the print (:806); .item() of an f32 tensor prints as a Python float
                log("Siamese Loss score:{}".format(float(-np.float32(d["loss"][s, it]))))
                log("Siamese attack tensors:{}".format(dance_digest(d["scores"][s, it], d["n_pos"], d["not_nll"])))
            log("iteration.{}, {}".format(it, text))
