"""Checkpoint -> device weight buffers + C structs.

Reads the three state-dicts the reference loads (main.py:454-469, siamese_contrastive.py:1565-1587)
with the key names of make_model / SiameseNet / Predictor, and lays them out for the kernels:
decoder/encoder Q,K,V projections stacked to one [768,256] matrix, BatchNorm (eval) folded into
the preceding Linear, Predictor hidden 100 zero-padded to 128, transposed copies for the backward.
"""
import math
import os

import numpy as np
import torch

from . import _lib


def positional_encoding(rows: int, d_model: int = 256) -> torch.Tensor:
    """The `pe` buffer of PositionalEncoding (model.py:77-89), same torch ops, first `rows` rows."""
    pe = torch.zeros(rows, d_model)
    position = torch.arange(0, rows).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def load_state_dict_file(path: str) -> dict:
    """torch.save(state_dict) files (.pkl) or the compact .safetensors fixture.  Executes nothing
    from the file (weights_only=True)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def expand_ae_state_dict(sd: dict) -> dict:
    """Compact fixture -> full make_model key set (shared embedding under both names, model.py:308-313)."""
    sd = {k: (v.float() if v.dtype in (torch.float16, torch.bfloat16) else v) for k, v in sd.items()}  # fp16-stored fixtures are exact in f32
    if "tgt_embed.0.lut.weight" not in sd:
        sd["tgt_embed.0.lut.weight"] = sd["src_embed.0.lut.weight"]
    return sd


class _Packed:
    """Keeps the device tensors alive next to the ctypes struct that points at them."""

    def __init__(self, device, gemm_format="f32"):
        self.device = device
        self.gemm_format = gemm_format
        self._keep = []

    def dev(self, t: torch.Tensor) -> int:
        d = t.detach().to(dtype=torch.float32).contiguous().to(self.device)
        self._keep.append(d)
        return d.data_ptr()

    def packed(self, t: torch.Tensor) -> int:
        """[N,K] Linear weight -> fragment-ordered device image (st_pack_weight)."""
        from . import ops
        d = ops.pack_weight(t.detach().to(dtype=torch.float32).contiguous().to(self.device), self.gemm_format)
        self._keep.append(d)
        return d.data_ptr()


class AEWeights(_Packed):
    """Decoder + encoder weights of the Transformer autoencoder (model.py:302-324).
    gemm_format: "f16x3" (default: 3-product f16 split on the f16 MFMA, f32-grade results) or "f32"
    (exact f32 MFMA, ~3x slower)."""

    def __init__(self, sd: dict, device, max_len: int = 60, gemm_format: str = "f16x3", l0_table: bool = True,
                 l0_table_max_bytes: int = 8 << 30):
        super().__init__(device, gemm_format)
        if gemm_format not in _lib.GEMM_FORMATS:
            raise ValueError("gemm_format %r: expected one of %s" % (gemm_format, sorted(_lib.GEMM_FORMATS)))
        if gemm_format == "f16x3":
            big = max(float(v.abs().max()) for k, v in sd.items() if k.endswith(".weight") and v.dim() == 2)
            if not big < 6.0e4:
                raise _lib.StError("gemm_format f16x3 needs |W| < 65504 (max |W| = %g): use gemm_format='f32'" % big)
        sd = expand_ae_state_dict(sd)
        emb = sd["tgt_embed.0.lut.weight"]
        self.vocab, self.d_model = emb.shape
        if self.d_model != 256:
            raise _lib.StError("transformer_model_size=%d unsupported: kernels are specialised for 256" % self.d_model)
        self.n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.layers."))
        self.d_ff = sd["decoder.layers.0.feed_forward.w_1.weight"].shape[0]
        self.max_len = max_len
        pe = positional_encoding(max(max_len, 64), self.d_model)
        self.emb_ptr = self.dev(emb)
        self.pe_ptr = self.dev(pe)
        self.dec = _lib.StDecoderWeights()
        for s in (self.dec,):
            s.d_model, s.n_heads, s.d_ff, s.n_layers, s.vocab, s.pe_rows = self.d_model, 4, self.d_ff, self.n_layers, self.vocab, pe.shape[0]
        self.dec.gemm_format = _lib.GEMM_FORMATS[gemm_format]
        # bound used by the screened argmax (st_kernels.h: gen_w_maxnorm), rounded up
        gnorm = sd["generator.proj.weight"].double().norm(dim=1) * (1.0 + 1e-6)
        self.dec.gen_w_maxnorm = float(gnorm.max())
        self.dec.gen_w_norm = self.dev(torch.nextafter(gnorm.float(), torch.tensor(float("inf"))))   # per-column bound, rounded up
        self.dec.emb, self.dec.pe = self.emb_ptr, self.pe_ptr
        self.dec.lnf_a, self.dec.lnf_b = self.dev(sd["decoder.norm.a_2"]), self.dev(sd["decoder.norm.b_2"])
        self.dec.gen_w, self.dec.gen_b = self.packed(sd["generator.proj.weight"]), self.dev(sd["generator.proj.bias"])
        for l in range(self.n_layers):
            p = "decoder.layers.%d." % l
            y = self.dec.layer[l]
            y.ln0_a, y.ln0_b = self.dev(sd[p + "sublayer.0.norm.a_2"]), self.dev(sd[p + "sublayer.0.norm.b_2"])
            y.wqkv = self.packed(torch.cat([sd[p + "self_attn.linears.%d.weight" % i] for i in range(3)], 0))
            y.bqkv = self.dev(torch.cat([sd[p + "self_attn.linears.%d.bias" % i] for i in range(3)], 0))
            y.wo, y.bo = self.packed(sd[p + "self_attn.linears.3.weight"]), self.dev(sd[p + "self_attn.linears.3.bias"])
            y.xv_w, y.xv_b = self.packed(sd[p + "src_attn.linears.2.weight"]), self.dev(sd[p + "src_attn.linears.2.bias"])
            y.xo_w, y.xo_b = self.packed(sd[p + "src_attn.linears.3.weight"]), self.dev(sd[p + "src_attn.linears.3.bias"])
            y.ln2_a, y.ln2_b = self.dev(sd[p + "sublayer.2.norm.a_2"]), self.dev(sd[p + "sublayer.2.norm.b_2"])
            y.w1, y.b1 = self.packed(sd[p + "feed_forward.w_1.weight"]), self.dev(sd[p + "feed_forward.w_1.bias"])
            y.w2, y.b2 = self.packed(sd[p + "feed_forward.w_2.weight"]), self.dev(sd[p + "feed_forward.w_2.bias"])
        # first decoder layer: q | k | v per (token, position), computed once (st_kernels.h: l0_table); skipped for
        # vocabularies whose table would exceed l0_table_max_bytes
        self.dec.l0_table, self.dec.l0_table_len = None, 0
        if gemm_format == "f16x3" and l0_table:
            T = max_len - 1
            L = _lib.lib()
            import ctypes as C
            nbytes = L.st_decode_l0_table_bytes(C.byref(self.dec), T)
            if 0 < nbytes <= l0_table_max_bytes:
                tab = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
                scratch = torch.empty(self.vocab * T * 256, dtype=torch.float32, device=device)
                with torch.cuda.device(device):
                    _lib.check(L.st_decode_build_l0_table(C.byref(self.dec), T, tab.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                                          torch.cuda.current_stream().cuda_stream), "st_decode_build_l0_table")
                del scratch
                self._keep.append(tab)
                self.dec.l0_table, self.dec.l0_table_len = tab.data_ptr(), T
        self.enc = _lib.StEncoderWeights()
        s = self.enc
        n_enc = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
        s.d_model, s.n_heads, s.d_ff, s.n_layers, s.vocab, s.pe_rows = self.d_model, 4, self.d_ff, n_enc, self.vocab, pe.shape[0]
        s.gemm_format = _lib.GEMM_FORMATS[gemm_format]
        s.emb, s.pe = self.emb_ptr, self.pe_ptr  # shared embedding; identical pe buffers
        s.lnf_a, s.lnf_b = self.dev(sd["encoder.norm.a_2"]), self.dev(sd["encoder.norm.b_2"])
        for l in range(n_enc):
            p = "encoder.layers.%d." % l
            y = s.layer[l]
            y.ln0_a, y.ln0_b = self.dev(sd[p + "sublayer.0.norm.a_2"]), self.dev(sd[p + "sublayer.0.norm.b_2"])
            y.wqkv = self.packed(torch.cat([sd[p + "self_attn.linears.%d.weight" % i] for i in range(3)], 0))
            y.bqkv = self.dev(torch.cat([sd[p + "self_attn.linears.%d.bias" % i] for i in range(3)], 0))
            y.wo, y.bo = self.packed(sd[p + "self_attn.linears.3.weight"]), self.dev(sd[p + "self_attn.linears.3.bias"])
            y.ln1_a, y.ln1_b = self.dev(sd[p + "sublayer.1.norm.a_2"]), self.dev(sd[p + "sublayer.1.norm.b_2"])
            y.w1, y.b1 = self.packed(sd[p + "feed_forward.w_1.weight"]), self.dev(sd[p + "feed_forward.w_1.bias"])
            y.w2, y.b2 = self.packed(sd[p + "feed_forward.w_2.weight"]), self.dev(sd[p + "feed_forward.w_2.bias"])
        # the encoder's first layer has the same property (st_kernels.h: StEncoderWeights.l0_table)
        s.l0_table, s.l0_table_len = None, 0
        if gemm_format == "f16x3" and l0_table:
            import ctypes as C
            T = max_len - 1
            L = _lib.lib()
            nbytes = L.st_encode_l0_table_bytes(C.byref(s), T)
            if 0 < nbytes <= l0_table_max_bytes:
                tab = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
                scratch = torch.empty(self.vocab * T * 256, dtype=torch.float32, device=device)
                with torch.cuda.device(device):
                    _lib.check(L.st_encode_build_l0_table(C.byref(s), T, tab.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                                          torch.cuda.current_stream().cuda_stream), "st_encode_build_l0_table")
                del scratch
                self._keep.append(tab)
                s.l0_table, s.l0_table_len = tab.data_ptr(), T


def fold_bn(w, b, bn_w, bn_b, mean, var, eps=1e-5):
    """Linear followed by eval-mode BatchNorm1d == Linear with W' = s W, b' = s b + t (float64 fold)."""
    s = bn_w.double() / torch.sqrt(var.double() + eps)
    t = bn_b.double() - mean.double() * s
    return (w.double() * s[:, None]).float(), (b.double() * s + t).float()


class ScorerWeights(_Packed):
    """SiameseNet + Predictor (siamese_contrastive.py:29-86) from their two state-dicts.
    gemm_format "f16x3" (default): split-f16 operands on the f16 matrix cores, three products per term; every matrix is
    pre-scaled by a power of two (exact) so that its largest element lies in [2^13, 2^14) and the kernel undoes the scale
    (StScorerWeights.unscale).  "f32": exact f32 MFMA."""

    def __init__(self, siamese_sd: dict, predictor_sd, device, gemm_format: str = "f16x3"):
        super().__init__(device, gemm_format)
        if gemm_format not in _lib.GEMM_FORMATS:
            raise ValueError("gemm_format %r: expected one of %s" % (gemm_format, sorted(_lib.GEMM_FORMATS)))
        if siamese_sd["fc1.weight"].shape != (512, 256):
            raise _lib.StError("latent_size %d unsupported: kernels are specialised for 256" % siamese_sd["fc1.weight"].shape[1])
        self.c = _lib.StScorerWeights()
        c = self.c
        c.gemm_format = _lib.GEMM_FORMATS[gemm_format]
        w1, b1 = fold_bn(siamese_sd["fc1.weight"], siamese_sd["fc1.bias"], siamese_sd["norm1.weight"],
                         siamese_sd["norm1.bias"], siamese_sd["norm1.running_mean"], siamese_sd["norm1.running_var"])
        w2, b2 = fold_bn(siamese_sd["fc2.weight"], siamese_sd["fc2.bias"], siamese_sd["norm2.weight"],
                         siamese_sd["norm2.bias"], siamese_sd["norm2.running_mean"], siamese_sd["norm2.running_var"])
        w3, b3 = siamese_sd["fc3.weight"], siamese_sd["fc3.bias"]

        def pair(i, w):
            """-> (packed W, packed W^T); in the f16x3 format of W * 2^e with max |W| 2^e in [2^13, 2^14)"""
            w = w.float()
            if gemm_format == "f16x3":
                big = float(w.abs().max())
                if not (0.0 < big < float("inf")):
                    raise _lib.StError("scorer layer %d has no finite non-zero weight" % i)
                e = 13 - math.floor(math.log2(big))
                w = w * (2.0 ** e)
                c.unscale[i] = 2.0 ** -e
            else:
                c.unscale[i] = 1.0
            return self.packed(w), self.packed(w.t())
        (c.s1_w, c.s1_wt), c.s1_b = pair(0, w1), self.dev(b1)
        (c.s2_w, c.s2_wt), c.s2_b = pair(1, w2), self.dev(b2)
        (c.s3_w, c.s3_wt), c.s3_b = pair(2, w3), self.dev(b3)
        if predictor_sd is not None:
            u1, c1 = fold_bn(predictor_sd["fc1.weight"], predictor_sd["fc1.bias"], predictor_sd["norm1.weight"],
                             predictor_sd["norm1.bias"], predictor_sd["norm1.running_mean"], predictor_sd["norm1.running_var"])
            hid = u1.shape[0]
            if hid > 128:
                raise _lib.StError("Predictor hidden width %d > 128 unsupported" % hid)
            u1p = torch.zeros(128, 256); u1p[:hid] = u1
            c1p = torch.zeros(128); c1p[:hid] = c1
            u2p = torch.zeros(256, 128); u2p[:, :hid] = predictor_sd["fc2.weight"]
            (c.p1_w, c.p1_wt), c.p1_b = pair(3, u1p), self.dev(c1p)
            (c.p2_w, c.p2_wt), c.p2_b = pair(4, u2p), self.dev(predictor_sd["fc2.bias"])

    def check_range(self):
        """raises when an f16x3 scorer kernel saw a value outside the f16 range since the last check (synchronises)"""
        import ctypes as C
        flag = C.c_int(0)
        _lib.check(_lib.lib().st_scorer_status(C.byref(flag), 1), "st_scorer_status")
        if flag.value:
            raise _lib.StError("scorer (f16x3 format): an activation or gradient left the f16 range; the edited latents are "
                               "invalid - build ScorerWeights with gemm_format='f32'")


class ClassifierWeights(_Packed):
    """The legacy FGIM style classifier (model.py:414-433: 256 -> 100 -> 50 -> 1) from its state-dict."""

    def __init__(self, sd: dict, device):
        super().__init__(device)
        if tuple(sd["fc1.weight"].shape) != (100, 256) or tuple(sd["fc2.weight"].shape) != (50, 100) or \
                tuple(sd["fc3.weight"].shape) != (1, 50):
            raise _lib.StError("Classifier shape unsupported: expected fc1 [100,256], fc2 [50,100], fc3 [1,50]")
        self.c = _lib.StClassifierWeights()
        for i in (1, 2, 3):
            setattr(self.c, "w%d" % i, self.dev(sd["fc%d.weight" % i]))
            setattr(self.c, "b%d" % i, self.dev(sd["fc%d.bias" % i]))
