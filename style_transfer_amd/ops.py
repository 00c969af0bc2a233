"""Python callables over the C-ABI, on torch device tensors (torch = device memory + streams only).

Names mirror the reference functions they replace:
  greedy_decode         <- EncoderDecoder.greedy_decode          (model.py:271-299)
  encode_latent         <- EncoderDecoder.get_latent / forward   (model.py:229-257)
  scorer_embed          <- SiameseNet.forward on like/unlike     (siamese_contrastive.py:381-382)
  scorer_edit_step      <- get_contrastive_loss + backward + Adam.step (:796-810)
  gemm_argmax           <- Generator + torch.max                 (model.py:200-207, :290-292)
  accept_changes        <- id2text_sentence compare + Change bookkeeping (:816-839)
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(t, shape=None):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "need contiguous f32 CUDA tensor"
    if shape is not None:
        assert tuple(t.shape) == tuple(shape), "shape %s != %s" % (tuple(t.shape), tuple(shape))
    return t


class DecodeWorkspace:
    """Grow-only device workspace shared by decode and encode calls."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = None
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self.buf


def decode_workspace_bytes(ae, B, max_len=60, l0_table=True):
    """device workspace st_decode_greedy needs for B decode jobs (K/V cache: 121 KB per job when the first layer reads
    the (token, position) table, 241 KB otherwise)"""
    return int(lib().st_decode_workspace_bytes_opt(C.byref(ae.dec), int(B), max_len, 0 if l0_table else _lib.ST_DEC_NO_L0_TABLE))


def greedy_decode(ae, latent, max_len=60, start_id=2, eos=3, pad=0, early_exit=True, sync_every=1,
                  want_gap=False, ws: DecodeWorkspace = None, screen=True, l0_table=True):
    """-> (ids [B,max_len-1] int32, gap or None, steps_run).
    screen=False (or want_gap=True) forces the full 3-product generator of the f16x3 format on every row."""
    _f32(latent)
    B = latent.shape[0]
    assert latent.shape[1] == 256
    T = max_len - 1
    ids = torch.empty(B, T, dtype=torch.int32, device=latent.device)
    gap = torch.empty(B, T, dtype=torch.float32, device=latent.device) if want_gap else None
    if B == 0:
        return ids, gap, 0
    options = ((_lib.ST_DEC_EARLY_EXIT if early_exit else 0) | (0 if screen else _lib.ST_DEC_NO_SCREEN) |
               (0 if l0_table else _lib.ST_DEC_NO_L0_TABLE))
    need = lib().st_decode_workspace_bytes_opt(C.byref(ae.dec), B, max_len, options)
    ws = ws or DecodeWorkspace(latent.device)
    buf = ws.get(need)
    steps = C.c_int(0)
    check(lib().st_decode_greedy(C.byref(ae.dec), latent.data_ptr(), B, max_len, start_id, eos, pad, options,
                                 sync_every, ids.data_ptr(), _p(gap),
                                 buf.data_ptr(), buf.numel(), C.byref(steps), _stream()), "st_decode_greedy")
    return ids, gap, steps.value


def encode_latent(ae, tokens, tok_pos, sent_off, n_sent, max_sent_len, ws: DecodeWorkspace = None, out=None):
    """Packed tokens (int32 device tensors) -> latent [n_sent,256] (written into `out`, a contiguous [n_sent,256] f32
    tensor, when given: the chunks of a large batch go straight into their rows of one buffer)."""
    R = tokens.numel()
    if out is not None:
        assert out.shape == (n_sent, 256) and out.dtype == torch.float32 and out.is_contiguous() and out.device == sent_off.device
    latent = out if out is not None else torch.empty(n_sent, 256, dtype=torch.float32, device=sent_off.device)
    if n_sent == 0:
        return latent
    need = lib().st_encode_workspace_bytes(C.byref(ae.enc), R, n_sent)
    ws = ws or DecodeWorkspace(latent.device)
    buf = ws.get(max(need, 256))
    check(lib().st_encode_latent(C.byref(ae.enc), _p(tokens), _p(tok_pos), sent_off.data_ptr(), R, n_sent,
                                 max_sent_len, latent.data_ptr(), buf.data_ptr(), buf.numel(), _stream()),
          "st_encode_latent")
    return latent


def scorer_embed(sw, x, unit=True):
    _f32(x)
    n = x.shape[0]
    z = torch.empty_like(x)
    check(lib().st_scorer_embed(C.byref(sw.c), x.data_ptr(), z.data_ptr(), n, 1 if unit else 0, _stream()),
          "st_scorer_embed")
    return z


def scorer_edit_step(sw, latent, adam_m, adam_v, like_z, unlike_z, lr, step, active, tau=0.1, not_nll=False,
                     simplest=False, just_use_one=False, want_loss=False, want_grad=False, want_scores=False, optimizer="Adam"):
    """-> (loss, grad), or (loss, grad, scores [B, n_pos + n_neg]) with want_scores.
    optimizer: a torch.optim name (siamese_contrastive.py:780) out of _lib.ST_OPTIMIZERS; adam_m / adam_v are its state."""
    if optimizer not in _lib.ST_OPTIMIZERS:
        raise NotImplementedError("--victim_optimizer %s: the kernel has %s" % (optimizer, ", ".join(_lib.ST_OPTIMIZERS)))
    B = latent.shape[0]
    _f32(latent, (B, 256)); _f32(adam_m, (B, 256)); _f32(adam_v, (B, 256)); _f32(lr, (B,))
    n_pos = 0 if like_z is None else like_z.shape[1]
    n_neg = 0 if unlike_z is None else unlike_z.shape[1]
    if like_z is not None:
        _f32(like_z, (B, n_pos, 256))
    if unlike_z is not None:
        _f32(unlike_z, (B, n_neg, 256))
    assert step.dtype == torch.int32 and step.is_cuda
    assert active is None or (active.dtype == torch.uint8 and active.is_cuda)
    flags = (_lib.ST_F_NOT_NLL if not_nll else 0) | (_lib.ST_F_SIMPLEST if simplest else 0) | \
            (_lib.ST_F_JUST_USE_ONE if just_use_one else 0) | (_lib.ST_OPTIMIZERS[optimizer] << _lib.ST_F_OPT_SHIFT)
    loss = torch.empty(B, dtype=torch.float32, device=latent.device) if want_loss else None
    grad = torch.empty(B, 256, dtype=torch.float32, device=latent.device) if want_grad else None
    scores = torch.empty(B, n_pos + n_neg, dtype=torch.float32, device=latent.device) if want_scores else None
    check(lib().st_scorer_edit_step(C.byref(sw.c), latent.data_ptr(), adam_m.data_ptr(), adam_v.data_ptr(),
                                    _p(like_z), _p(unlike_z), n_pos, n_neg, lr.data_ptr(), step.data_ptr(),
                                    _p(active), B, float(tau), flags, _p(loss), _p(grad), _p(scores), _stream()),
          "st_scorer_edit_step")
    return (loss, grad, scores) if want_scores else (loss, grad)


def _fmt(gemm_format):
    if isinstance(gemm_format, str):
        if gemm_format not in _lib.GEMM_FORMATS:
            raise ValueError("gemm_format %r: expected one of %s" % (gemm_format, sorted(_lib.GEMM_FORMATS)))
        return _lib.GEMM_FORMATS[gemm_format]
    return int(gemm_format)


def pack_weight(W, gemm_format="f32"):
    """nn.Linear weight [N,K] (device f32) -> MFMA-fragment-ordered image (st_pack_weight) for the
    exact f32 kernels ("f32") or the 3-product f16 split kernels ("f16x3")."""
    _f32(W)
    N, K = W.shape
    out = torch.empty(lib().st_packed_weight_floats(N, K), dtype=torch.float32, device=W.device)
    check(lib().st_pack_weight(W.data_ptr(), N, K, _fmt(gemm_format), out.data_ptr(), _stream()), "st_pack_weight")
    return out


def gemm_argmax(h, W, b, ln_a=None, ln_b=None, packed=None, gemm_format="f32"):
    """W [V,256] plain (packed here) or pass packed=pack_weight(W, gemm_format)."""
    _f32(h); _f32(b)
    n, V = h.shape[0], b.shape[0]
    assert h.shape[1] == 256
    W = packed if packed is not None else pack_weight(W, gemm_format)
    ids = torch.empty(n, dtype=torch.int32, device=h.device)
    best = torch.empty(n, dtype=torch.float32, device=h.device)
    gap = torch.empty(n, dtype=torch.float32, device=h.device)
    scratch = torch.empty(max(n, 1) * _lib.ST_ARGMAX_SPLITS * 16, dtype=torch.uint8, device=h.device)
    check(lib().st_gemm_argmax(h.data_ptr(), _p(ln_a), _p(ln_b), W.data_ptr(), b.data_ptr(), n, V, _fmt(gemm_format),
                               ids.data_ptr(), best.data_ptr(), gap.data_ptr(), scratch.data_ptr(), _stream()), "st_gemm_argmax")
    return ids, best, gap


def gemm_argmax_noalloc(h, packed, b, ids, best, gap, ln_a=None, ln_b=None, gemm_format="f32"):
    """st_gemm_argmax_noalloc: every buffer (inputs AND outputs) comes from the caller, the scratch is the library's
    static device array; this wrapper allocates nothing either.  n <= _lib.ST_NOALLOC_MAX_ROWS."""
    _f32(h); _f32(b); _f32(best); _f32(gap)
    n, V = h.shape[0], b.shape[0]
    assert h.shape[1] == 256 and ids.dtype == torch.int32 and ids.numel() >= n and best.numel() >= n and gap.numel() >= n
    check(lib().st_gemm_argmax_noalloc(h.data_ptr(), _p(ln_a), _p(ln_b), packed.data_ptr(), b.data_ptr(), n, V,
                                       _fmt(gemm_format), ids.data_ptr(), best.data_ptr(), gap.data_ptr(), _stream()),
          "st_gemm_argmax_noalloc")


def accept_changes(new_ids, acc_ids, vocab, n_changes, changed, active, max_change=7, unk=1, eos=3, pad=0,
                   canon=None, rec_ids=None, confirm_ids=None):
    B, T = new_ids.shape
    assert new_ids.dtype == torch.int32 and acc_ids.dtype == torch.int32 and new_ids.is_contiguous() and acc_ids.is_contiguous()
    check(lib().st_accept_changes(new_ids.data_ptr(), acc_ids.data_ptr(), T, B, vocab, unk, eos, pad, _p(canon),
                                  max_change, n_changes.data_ptr(), changed.data_ptr(), active.data_ptr(),
                                  _p(rec_ids), _p(confirm_ids), _stream()), "st_accept_changes")


def simloss_format(gemm_format):
    """operand format of the similarity products ("f16x3" default, "f32" exact); returns the previous one"""
    prev = lib().st_simloss_format(_fmt(gemm_format))
    return {v: k for k, v in _lib.GEMM_FORMATS.items()}[prev]


def simloss_fwd(ori, like, tau=0.1, row_offset=0):
    """-> (loss [Bq], lse [Bq]) of the in-batch InfoNCE block (siamese_contrastive.py:1060-1069)."""
    _f32(ori); _f32(like)
    Bq, Bk = ori.shape[0], like.shape[0]
    loss = torch.empty(Bq, dtype=torch.float32, device=ori.device)
    lse = torch.empty(Bq, dtype=torch.float32, device=ori.device)
    nb = lib().st_simloss_scratch_bytes(Bq, Bk)
    scratch = torch.empty(nb, dtype=torch.uint8, device=ori.device)
    check(lib().st_simloss_fwd(ori.data_ptr(), like.data_ptr(), Bq, Bk, float(tau), row_offset, loss.data_ptr(),
                               lse.data_ptr(), scratch.data_ptr(), nb, _stream()), "st_simloss_fwd")
    return loss, lse


def simloss_bwd(ori, like, lse, g, tau=0.1, row_offset=0, d_like=None):
    """-> (d_ori [Bq,256], d_like [Bk,256]); d_like is accumulated into when given."""
    Bq, Bk = ori.shape[0], like.shape[0]
    d_ori = torch.empty_like(ori)
    if d_like is None:
        d_like = torch.zeros_like(like)
    nb = lib().st_simloss_scratch_bytes(Bq, Bk)
    scratch = torch.empty(nb, dtype=torch.uint8, device=ori.device)
    check(lib().st_simloss_bwd(ori.data_ptr(), like.data_ptr(), Bq, Bk, float(tau), row_offset, lse.data_ptr(),
                               g.data_ptr(), d_ori.data_ptr(), d_like.data_ptr(), scratch.data_ptr(), nb, _stream()),
          "st_simloss_bwd")
    return d_ori, d_like


def fgim_perturb(cw, latent, target, eps, want_grad=False):
    """Classifier forward + BCE backward to the latent + x - eps_k * grad (fgim_attack, model.py:454-516).
    latent [B,256], target [B] (0/1), eps: sequence of step sizes -> (pred [B], grad [B,256] or None, out [n_eps,B,256])."""
    _f32(latent)
    B = latent.shape[0]
    target = target.to(device=latent.device, dtype=torch.float32).contiguous().view(-1)
    assert target.numel() == B and latent.shape[1] == 256
    e = torch.as_tensor(list(eps), dtype=torch.float32, device=latent.device)
    pred = torch.empty(B, dtype=torch.float32, device=latent.device)
    grad = torch.empty(B, 256, dtype=torch.float32, device=latent.device) if want_grad else None
    out = torch.empty(e.numel(), B, 256, dtype=torch.float32, device=latent.device)
    check(lib().st_fgim_perturb(C.byref(cw.c), latent.data_ptr(), target.data_ptr(), e.data_ptr(), e.numel(), B,
                                pred.data_ptr(), _p(grad), out.data_ptr(), _stream()), "st_fgim_perturb")
    return pred, grad, out
