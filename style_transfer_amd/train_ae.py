"""Host side of the autoencoder's constraint training (second half): what
train_ae_model_constraint (/root/reference/method/yelp/siamese_contrastive.py:1000-1088) does per batch, on the HIP
step st_ae_train_step + st_adam_step:

    ori_latent, out = ae_model.forward(ori);  loss_rec = LabelSmoothing(out, tgt_y) / ntokens          :1037-1041
    like_latent = 10 views + second pass of the origins;  constraint = mean_v NLL(log_softmax(cos / tau))  :1049-1071
    (bata * constraint + loss_rec).backward();  NoamOpt(256, 1, 2000, Adam(betas=(0.9, 0.98), eps=1e-9)).step()   :1072-1078

All parameters live in ONE flat f32 vector in the order of ae_model.parameters() (the shared embedding once), so the
optimizer is one elementwise kernel and a data-parallel job all-reduces the gradient as one RCCL bucket between
loss_and_grads() and step().  torch only owns the device memory.
"""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .data import Ragged, pack_sentences


def ae_param_names(n_layers=2):
    """names of ae_model.parameters() in order (make_model, model.py:302-324: encoder, decoder, src_embed, generator;
    tgt_embed shares src_embed's table)"""
    names = []
    lin = lambda p: [p + ".weight", p + ".bias"]
    for l in range(n_layers):
        p = "encoder.layers.%d." % l
        for i in range(4):
            names += lin(p + "self_attn.linears.%d" % i)
        names += lin(p + "feed_forward.w_1") + lin(p + "feed_forward.w_2")
        for i in range(2):
            names += [p + "sublayer.%d.norm.a_2" % i, p + "sublayer.%d.norm.b_2" % i]
    names += ["encoder.norm.a_2", "encoder.norm.b_2"]
    for l in range(n_layers):
        p = "decoder.layers.%d." % l
        for a in ("self_attn", "src_attn"):
            for i in range(4):
                names += lin(p + "%s.linears.%d" % (a, i))
        names += lin(p + "feed_forward.w_1") + lin(p + "feed_forward.w_2")
        for i in range(3):
            names += [p + "sublayer.%d.norm.a_2" % i, p + "sublayer.%d.norm.b_2" % i]
    names += ["decoder.norm.a_2", "decoder.norm.b_2", "src_embed.0.lut.weight", "generator.proj.weight", "generator.proj.bias"]
    return names


def noam_rate(step, d_model=256, factor=1, warmup=2000):
    """NoamOpt.rate (model.py:375-381)"""
    return factor * (d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


@dataclass
class AETrainConfig:
    bata: float = 0.1          # main.py:92
    tau: float = 0.1           # main.py:91
    smoothing: float = 0.1     # siamese_contrastive.py:1011-1012
    dropout: float = 0.1       # make_model default (model.py:302)
    warmup: int = 2000         # :1009
    factor: float = 1.0
    seed: int = 1234


# ---- the kernels' dropout decisions, reproduced on the host (tests replay them through the CPU restatement) ----
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def keep_mask(seed, site, idx, p):
    """bool array: the keep decision of csrc/st_aetrain.hip's drop_factor for element indices `idx` of dropout `site`"""
    with np.errstate(over="ignore"):
        idx = np.asarray(idx, np.uint64)
        h = _mix(_mix(np.uint64(seed) ^ (np.uint64(site) << np.uint64(40))) ^ idx)
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(p)


SITE_ENC_PE, SITE_DEC_PE = 1, 2
enc_site = lambda layer, k: 10 + 4 * layer + k      # k: 0 attention, 1 sublayer[0], 2 feed-forward, 3 sublayer[1]
dec_site = lambda layer, k: 100 + 6 * layer + k     # k: 0 self-attention, 1 sublayer[0], 2 source attention, 3 sublayer[1], 4 ffn, 5 sublayer[2]


class AETrainer:
    def __init__(self, state_dict, cfg: AETrainConfig, device):
        """state_dict: the AE's state-dict (torch tensors; reference key names; `pe` buffers optional)"""
        self.cfg, self.device = cfg, device
        sd = state_dict
        self.n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.layers."))
        self.names = ae_param_names(self.n_layers)
        self.shapes = [tuple(sd[n].shape) for n in self.names]
        sizes = [int(np.prod(s)) for s in self.shapes]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.n_params = int(self.offsets[-1])
        self.params = torch.cat([sd[n].detach().float().reshape(-1) for n in self.names]).to(device).contiguous()
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.vocab, d_model = self.shapes[self.names.index("src_embed.0.lut.weight")]
        assert d_model == 256
        self.d_ff = self.shapes[self.names.index("encoder.layers.0.feed_forward.w_1.weight")][0]
        from .weights import positional_encoding
        self.pe = positional_encoding(64, 256).to(device).contiguous()
        self.losses = torch.zeros(2, dtype=torch.float32, device=device)
        self.step_no = 0
        self._p = self._struct(self.params)
        self._g = self._struct(self.grads)
        self._ws = None

    def _struct(self, flat):
        s = _lib.StAeParams()
        s.d_model, s.n_heads, s.d_ff, s.n_layers, s.vocab, s.pe_rows = 256, 4, self.d_ff, self.n_layers, self.vocab, 64
        ptr = lambda name: flat.data_ptr() + 4 * int(self.offsets[self.names.index(name)])
        s.emb, s.pe = ptr("src_embed.0.lut.weight"), self.pe.data_ptr()
        s.enc_norm_a, s.enc_norm_b = ptr("encoder.norm.a_2"), ptr("encoder.norm.b_2")
        s.dec_norm_a, s.dec_norm_b = ptr("decoder.norm.a_2"), ptr("decoder.norm.b_2")
        s.gen_w, s.gen_b = ptr("generator.proj.weight"), ptr("generator.proj.bias")
        for l in range(self.n_layers):
            for which, arr, ffn_norm in (("encoder", s.enc, 1), ("decoder", s.dec, 2)):
                p, y = "%s.layers.%d." % (which, l), arr[l]
                y.ln0_a, y.ln0_b = ptr(p + "sublayer.0.norm.a_2"), ptr(p + "sublayer.0.norm.b_2")
                for i, (wn, bn) in enumerate((("wq", "bq"), ("wk", "bk"), ("wv", "bv"), ("wo", "bo"))):
                    setattr(y, wn, ptr(p + "self_attn.linears.%d.weight" % i))
                    setattr(y, bn, ptr(p + "self_attn.linears.%d.bias" % i))
                if which == "decoder":
                    y.xv_w, y.xv_b = ptr(p + "src_attn.linears.2.weight"), ptr(p + "src_attn.linears.2.bias")
                    y.xo_w, y.xo_b = ptr(p + "src_attn.linears.3.weight"), ptr(p + "src_attn.linears.3.bias")
                y.ln2_a, y.ln2_b = ptr(p + "sublayer.%d.norm.a_2" % ffn_norm), ptr(p + "sublayer.%d.norm.b_2" % ffn_norm)
                y.w1, y.b1 = ptr(p + "feed_forward.w_1.weight"), ptr(p + "feed_forward.w_1.bias")
                y.w2, y.b2 = ptr(p + "feed_forward.w_2.weight"), ptr(p + "feed_forward.w_2.bias")
        return s

    # -- data: one batch in the reference's roles --------------------------------------------------
    @staticmethod
    def pack_lists(origin, views, vocab, bos=2, eos=3, unk=1, max_seq_length=60):
        """origin: B id lists; views: K lists of B id lists (view-major, like_tensor_src [like_num, batch, seq]).
        -> the int32 numpy arrays of the length-packed encoder batch (origins | views | origins) and decoder batch"""
        clip = lambda s: [t if t < vocab else unk for t in s[: max_seq_length - 1]]
        origin = [clip(s) for s in origin]
        enc_sents = origin + [clip(s) for view in views for s in view] + origin
        tok, pos, off, mx = pack_sentences(enc_sents)
        assert mx <= 64 and tok.size == sum(len(s) for s in enc_sents), "training sentences: <= 64 tokens, no PAD ids inside"
        dec_in = [[bos] + s for s in origin]
        dec_tgt = [s + [eos] for s in origin]
        dtok, dpos, doff, dmx = pack_sentences(dec_in)
        dtgt = np.fromiter((t for s in dec_tgt for t in s), np.int32, dtok.size)
        assert dmx <= 64
        return (tok, pos, off), (dtok, dpos, dtgt, doff), enc_sents, origin

    def pack(self, origin, views, bos=2, eos=3, unk=1, max_seq_length=60):
        """pack_lists -> device int32 arrays"""
        enc, dec, enc_sents, origin = self.pack_lists(origin, views, self.vocab, bos, eos, unk, max_seq_length)
        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(dev)
        return dict(B=len(origin), n_sent=len(enc_sents), enc=tuple(t(a) for a in enc), dec=tuple(t(a) for a in dec),
                    enc_sents=enc_sents, origin=origin)

    @staticmethod
    def pack_arrays(origin_r, views_r, vocab, bos=2, eos=3, unk=1, max_seq_length=60):
        """pack() on flat arrays, without a per-sentence Python loop (the epoch loop's form: at the reference's batch of
        128 records the list version costs ~ 4 ms of host time per 5 ms step).  origin_r: Ragged of B sentences; views_r:
        Ragged of K * B sentences, view-major.  -> (enc tokens, positions, offsets), (dec tokens, positions, targets,
        offsets), n_sent: int32 numpy arrays equal to what pack() sends to the device."""
        def clip(r):
            lens = r.lens
            tok = np.where(r.tokens < vocab, r.tokens, unk).astype(np.int32)
            limit = max_seq_length - 1
            if lens.size and int(lens.max()) > limit:
                pos = np.arange(tok.size, dtype=np.int64) - np.repeat(r.off[:-1], lens)
                tok = tok[pos < limit]
                lens = np.minimum(lens, limit)
            return Ragged.from_lengths(tok, lens)
        o = clip(origin_r)
        enc = Ragged.concat([o, clip(views_r), o])
        tok, pos, off, mx = pack_sentences(enc)
        assert mx <= 64 and tok.size == enc.tokens.size, "training sentences: <= 64 tokens, no PAD ids inside"
        B, lens = len(o), o.lens
        doff = np.zeros(B + 1, np.int64)
        np.cumsum(lens + 1, out=doff[1:])
        n = int(doff[-1])
        first = np.zeros(n, bool)
        first[doff[:-1]] = True
        last = np.zeros(n, bool)
        last[doff[1:] - 1] = True
        dtok = np.empty(n, np.int32)
        dtok[first] = bos
        dtok[~first] = o.tokens
        dtgt = np.empty(n, np.int32)
        dtgt[last] = eos
        dtgt[~last] = o.tokens
        dpos = (np.arange(n, dtype=np.int64) - np.repeat(doff[:-1], lens + 1)).astype(np.int32)
        assert int(lens.max()) + 1 <= 64
        return (tok, pos, off), (dtok, dpos, dtgt, doff.astype(np.int32)), len(enc)

    def pack_ragged(self, origin_r, views_r):
        """pack() for the epoch loop: Ragged arrays in, pinned staging, asynchronous copies (the buffers stay alive in the
        returned dict until the step that reads them has been queued)"""
        enc, dec, n_sent = self.pack_arrays(origin_r, views_r, self.vocab)
        host = []

        def t(a):
            src = torch.from_numpy(np.ascontiguousarray(a, np.int32))
            pin = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
            pin.copy_(src)
            host.append(pin)
            return pin.to(self.device, non_blocking=True)
        return dict(B=len(origin_r), n_sent=n_sent, enc=tuple(t(a) for a in enc), dec=tuple(t(a) for a in dec), _host=host)

    def loss_and_grads(self, batch, seed=None, fetch=True):
        """-> (reconstruction loss, constraint loss) as floats; self.grads holds d(bata * constraint + rec) / d params.
        fetch=False: no host synchronisation, the two losses stay in self.losses (device) and None is returned"""
        c = self.cfg
        et, ep, eo = batch["enc"]
        dt, dp, dy, do = batch["dec"]
        L = lib()
        need = L.st_ae_train_workspace_bytes(C.byref(self._p), et.numel(), batch["n_sent"], dt.numel(), batch["B"])
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        seed = (c.seed + 7919 * self.step_no) if seed is None else seed
        check(L.st_ae_train_step(C.byref(self._p), C.byref(self._g), self.grads.data_ptr(), self.n_params,
                                 et.data_ptr(), ep.data_ptr(), eo.data_ptr(), et.numel(), batch["n_sent"],
                                 dt.data_ptr(), dp.data_ptr(), dy.data_ptr(), do.data_ptr(), dt.numel(), batch["B"],
                                 float(c.bata), float(c.tau), float(c.smoothing), float(c.dropout), int(seed) & (2 ** 64 - 1),
                                 self.losses.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                 torch.cuda.current_stream().cuda_stream), "st_ae_train_step")
        if not fetch:
            return None
        rec, con = self.losses.tolist()
        return rec, con

    def step(self):
        """NoamOpt.step (model.py:366-373) around Adam(betas=(0.9, 0.98), eps=1e-9)"""
        self.step_no += 1
        lr = noam_rate(self.step_no, 256, self.cfg.factor, self.cfg.warmup)
        check(lib().st_adam_step(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                 self.n_params, float(lr), 0.9, 0.98, 1e-9, self.step_no,
                                 torch.cuda.current_stream().cuda_stream), "st_adam_step")
        return lr

    def tensor(self, flat, name):
        i = self.names.index(name)
        return flat[int(self.offsets[i]):int(self.offsets[i + 1])].view(self.shapes[i])

    def state_dict(self):
        sd = {n: self.tensor(self.params, n).detach().cpu().clone() for n in self.names}
        sd["tgt_embed.0.lut.weight"] = sd["src_embed.0.lut.weight"]
        return sd


def fresh_ae_state_dict(vocab, n_layers=2, d_model=256, d_ff=1024, seed=None):
    """make_model's initialisation (model.py:302-324): nn.Linear / nn.Embedding defaults, then xavier_uniform_ on every
    parameter with more than one dimension; LayerNorm a_2 = 1, b_2 = 0."""
    g = torch.Generator().manual_seed(0 if seed is None else seed)
    sd = {}
    shapes = dict()
    for n in ae_param_names(n_layers):
        if n.endswith("lut.weight") or n == "generator.proj.weight":
            shapes[n] = (vocab, d_model)
        elif n == "generator.proj.bias":
            shapes[n] = (vocab,)
        elif "w_1.weight" in n:
            shapes[n] = (d_ff, d_model)
        elif "w_1.bias" in n:
            shapes[n] = (d_ff,)
        elif "w_2.weight" in n:
            shapes[n] = (d_model, d_ff)
        elif n.endswith(".weight"):
            shapes[n] = (d_model, d_model)
        else:
            shapes[n] = (d_model,)
    for n, sh in shapes.items():
        if len(sh) == 2:
            bound = math.sqrt(6.0 / (sh[0] + sh[1]))
            sd[n] = (torch.rand(sh, generator=g) * 2 - 1) * bound
        elif n.endswith("a_2"):
            sd[n] = torch.ones(sh)
        elif n.endswith("b_2"):
            sd[n] = torch.zeros(sh)
        else:                                   # nn.Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            fan_in = d_ff if "w_2.bias" in n else d_model
            sd[n] = (torch.rand(sh, generator=g) * 2 - 1) / math.sqrt(fan_in)
    sd["tgt_embed.0.lut.weight"] = sd["src_embed.0.lut.weight"]
    return sd


def train_ae_model_constraint(trainer: AETrainer, data, batch_size=128, epochs=1, log=print, save_path=None):
    """The epoch loop of train_ae_model_constraint (siamese_contrastive.py:1020-1088) over an AttackSet read from
    ae_train_constraint{0,1}.json (origin + sim0..): the reference's shuffle (seed 234, the same permutation for the
    origin and the like loader), int(n / batch_size) batches, one checkpoint per epoch under the reference's name."""
    import os
    from .logfmt import headline
    from .train_scorer import epoch_order
    log(headline("Training ae_model with Constraint", note_token="+"))
    hist = []
    for epoch in range(epochs):
        log(headline("Start Running Epoch {}".format(epoch)))
        # the loss sums live on the device and are read when a line is logged (every 200 batches, as the reference prints)
        # and at the end of the epoch: no host synchronisation per step, so packing the next batch (flat arrays, no
        # per-sentence Python) and queueing its ~ 400 launches overlap the current step on the GPU
        sums = torch.zeros(2, dtype=torch.float64, device=trainer.device)
        batches = epoch_order(len(data), batch_size)
        K = data.n_pos
        keep = None
        for it, recs in enumerate(batches):
            idx = np.asarray(recs, np.int64)
            views = data.like_r.take((idx[None, :] * K + np.arange(K, dtype=np.int64)[:, None]).reshape(-1))   # view-major
            batch = trainer.pack_ragged(data.origin_r.take(idx), views)
            trainer.loss_and_grads(batch, fetch=False)
            trainer.step()
            sums += trainer.losses
            keep = (batch, keep[0] if keep else None)     # the pinned staging of the last two batches stays alive
            if it % 200 == 0:
                rec_sum, con_sum = sums.tolist()
                log("epoch {}, batch {}, rec loss={}, constraint loss={}".format(epoch, it, rec_sum / (it + 1), con_sum / (it + 1)))
        rec_sum, con_sum = sums.tolist()
        hist.append((rec_sum / max(1, len(batches)), con_sum / max(1, len(batches))))
        if save_path is not None:
            torch.save(trainer.state_dict(), os.path.join(save_path, "constraint_ae_model{:d}_params.pkl".format(epoch)))
    return hist
