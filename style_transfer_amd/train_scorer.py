"""Training of the style scorer on the GPU — host-side mirror of the reference's `train_contrastive`
(siamese_contrastive.py:858-947; SURVEY.md §8f rank 3, first half).

What stays as in the reference: the AE is frozen and only supplies latents (:871, :904-916); SiameseNet and Predictor
train with torch.optim.Adam(lr=--train_siamese_lr) on the NLL form of get_contrastive_loss; one epoch walks
`int(sent_num / batch_size)` batches of the list shuffled with `random.seed(234)` (:276-290, data.py:403-408); the
epoch ends with `Epoch Loss=<sum of batch losses>` and the two checkpoints `siamese_model_params.pkl` /
`predictor_net_params.pkl` (state_dicts, :1556-1562) that `--attack` loads.

What is different: the step (forward, backward, BatchNorm statistics, Adam) is `st_scorer_train_step` +
`st_adam_step` (HIP, include/st_kernels.h (8)); the latents of the 1 + Kp + Kn sentences of every record are encoded
ONCE before the first epoch (the AE is in eval mode, so they never change) instead of once per batch per epoch;
Dropout draws from a counter hash instead of torch's generator (parity tests replay the reference's masks through
`keep_masks`).  Data parallel: every rank takes B / world rows of each batch, BatchNorm statistics are per rank
(torch's DistributedDataParallel default, no SyncBatchNorm), gradients are averaged with one all-reduce of the flat
vector (RCCL over xGMI), running statistics are averaged once per epoch before the checkpoint is written.

There is no CPU fallback: without the HIP library every call raises.
"""
import ctypes as C
import random
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

# (net, tensor) in the order of list(siamese_net.parameters()) + list(predictor_net.parameters())  (:881-882)
PARAM_ORDER = [("s", "fc1.weight"), ("s", "fc1.bias"), ("s", "norm1.weight"), ("s", "norm1.bias"),
               ("s", "fc2.weight"), ("s", "fc2.bias"), ("s", "norm2.weight"), ("s", "norm2.bias"),
               ("s", "fc3.weight"), ("s", "fc3.bias"),
               ("p", "fc1.weight"), ("p", "fc1.bias"), ("p", "norm1.weight"), ("p", "norm1.bias"),
               ("p", "fc2.weight"), ("p", "fc2.bias")]
PARAM_SHAPES = [(512, 256), (512,), (512,), (512,), (512, 512), (512,), (512,), (512,), (256, 512), (256,),
                (100, 256), (100,), (100,), (100,), (256, 100), (256,)]
BN_ORDER = [("s", "norm1.running_mean", 512), ("s", "norm1.running_var", 512), ("s", "norm2.running_mean", 512),
            ("s", "norm2.running_var", 512), ("p", "norm1.running_mean", 100), ("p", "norm1.running_var", 100)]
TP_COUNT = sum(int(np.prod(s)) for s in PARAM_SHAPES)       # == ST_TP_COUNT
TP_SIAMESE_COUNT = sum(int(np.prod(s)) for s in PARAM_SHAPES[:10])
TB_COUNT = sum(n for _, _, n in BN_ORDER)                   # == ST_TB_COUNT
assert (TP_COUNT, TP_SIAMESE_COUNT, TB_COUNT) == (579372, 527616, 2248)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def fresh_state_dicts(seed=None):
    """Initial weights as `SiameseNet(256)` / `Predictor(256)` construct them (nn.Linear / nn.BatchNorm1d defaults:
    kaiming-uniform(a=sqrt(5)) weights, uniform(+-1/sqrt(fan_in)) biases, gamma 1, beta 0, running stats 0 / 1)."""
    g = torch.Generator().manual_seed(seed if seed is not None else random.getrandbits(31))
    sds = {"s": {}, "p": {}}
    for (net, name), shape in zip(PARAM_ORDER, PARAM_SHAPES):
        layer, kind = name.split(".")
        if layer.startswith("norm"):
            sds[net][name] = torch.ones(shape) if kind == "weight" else torch.zeros(shape)
        else:
            w_shape = PARAM_SHAPES[PARAM_ORDER.index((net, layer + ".weight"))]
            bound = 1.0 / np.sqrt(w_shape[1])
            sds[net][name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    for net, name, n in BN_ORDER:
        sds[net][name] = torch.zeros(n) if name.endswith("mean") else torch.ones(n)
        sds[net][name.rsplit(".", 1)[0] + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return sds["s"], sds["p"]


@dataclass
class TrainConfig:
    lr: float = 3e-3                 # --train_siamese_lr (main.py:82)
    tau: float = 0.1                 # --tau
    simplest: bool = False           # --simplest_siamese
    just_use_one: bool = False       # --just_use_one
    drop_p: float = 0.2              # nn.Dropout(0.2) in SiameseNet (:37,:42)
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    seed: int = 234


class ScorerTrainer:
    """Parameters, BatchNorm statistics and Adam moments of SiameseNet + Predictor as flat device vectors."""

    def __init__(self, siamese_sd, predictor_sd, cfg: TrainConfig, device):
        lib()  # fail now if the HIP library is missing
        self.cfg, self.dev = cfg, torch.device(device)
        sds = {"s": siamese_sd, "p": predictor_sd}
        flat = []
        for (net, name), shape in zip(PARAM_ORDER, PARAM_SHAPES):
            t = torch.as_tensor(np.asarray(sds[net][name]), dtype=torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise ValueError("%s.%s has shape %s, expected %s" % (net, name, tuple(t.shape), shape))
            flat.append(t.reshape(-1))
        self.params = torch.cat(flat).to(self.dev)
        self.bn = torch.cat([torch.as_tensor(np.asarray(sds[net][name]), dtype=torch.float32).reshape(-1)
                             for net, name, _ in BN_ORDER]).to(self.dev)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.step_no = 0
        self.calls_tracked = {"s": int(np.asarray(siamese_sd.get("norm1.num_batches_tracked", 0))),
                              "p": int(np.asarray(predictor_sd.get("norm1.num_batches_tracked", 0)))}
        self.loss = torch.zeros(1, device=self.dev)
        self._ws, self._choice = None, None

    # ---- one step -------------------------------------------------------------------------------
    def loss_and_grads(self, ori, like, unlike, keep_masks=None, seed=None):
        """Forward + backward of one batch (get_contrastive_loss + loss.backward(), :918-930).  ori [B,256],
        like [Kp,B,256], unlike [Kn,B,256] device f32.  Fills self.grads / self.loss, moves the BatchNorm running
        statistics; returns `choice` [Kp,B] int32 (device)."""
        cfg = self.cfg
        B, Kp, Kn = ori.shape[0], like.shape[0], unlike.shape[0]
        x = torch.cat([ori.reshape(1, B, 256), like, unlike]).to(torch.float32).contiguous()
        flags = (_lib.ST_F_SIMPLEST if cfg.simplest else 0) | (_lib.ST_F_JUST_USE_ONE if cfg.just_use_one else 0)
        need = C.c_size_t()
        check(lib().st_scorer_train_workspace_bytes(B, Kp, Kn, flags, C.byref(need)), "st_scorer_train_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.dev)
        if self._choice is None or self._choice.numel() < Kp * B:
            self._choice = torch.empty(Kp * B, dtype=torch.int32, device=self.dev)
        if keep_masks is not None:
            keep_masks = keep_masks.to(torch.uint8).contiguous()
            if tuple(keep_masks.shape) != (1 + Kp + Kn, 2, B, 512):
                raise ValueError("keep_masks must be [1+Kp+Kn, 2, B, 512]")
        if seed is None:
            # the rank is part of the seed: the kernel hashes (seed, LOCAL element), and the reference draws independent
            # masks for every row of the global batch
            rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
            seed = (cfg.seed * 0x9E3779B1 + self.step_no + rank * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        check(lib().st_scorer_train_step(_p(self.params), _p(self.bn), _p(x), B, Kp, Kn, cfg.tau, flags, cfg.drop_p,
                                         _p(keep_masks), C.c_uint64(seed), _p(self.grads), _p(self.loss), _p(self._choice),
                                         _p(self._ws), self._ws.numel(), _stream()), "st_scorer_train_step")
        self.calls_tracked["s"] += 1 + Kp + Kn
        if not cfg.simplest:
            self.calls_tracked["p"] += 1
        return self._choice[:Kp * B].view(Kp, B)

    def adam(self):
        """optimizer.step() (:931): Adam over the parameters the reference's optimizer holds (:877-883)."""
        cfg = self.cfg
        self.step_no += 1
        n = TP_SIAMESE_COUNT if cfg.simplest else TP_COUNT
        check(lib().st_adam_step(_p(self.params), _p(self.grads), _p(self.m), _p(self.v), n, cfg.lr, cfg.betas[0], cfg.betas[1],
                                 cfg.eps, self.step_no, _stream()), "st_adam_step")

    def step(self, ori, like, unlike, keep_masks=None, world=1):
        choice = self.loss_and_grads(ori, like, unlike, keep_masks)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.grads)          # one bucket: the whole flat gradient (2.3 MB)
            self.grads /= world
        self.adam()
        return self.loss, choice

    # ---- checkpoints ------------------------------------------------------------------------------
    def state_dicts(self):
        """(SiameseNet state_dict, Predictor state_dict) with the reference's key names, CPU tensors."""
        out = {"s": {}, "p": {}}
        flat = self.params.detach().cpu()
        off = 0
        for (net, name), shape in zip(PARAM_ORDER, PARAM_SHAPES):
            n = int(np.prod(shape))
            out[net][name] = flat[off:off + n].reshape(shape).clone()
            off += n
        bn = self.bn.detach().cpu()
        off = 0
        for net, name, n in BN_ORDER:
            out[net][name] = bn[off:off + n].clone()
            off += n
        for net in ("s", "p"):
            for layer in (("norm1", "norm2") if net == "s" else ("norm1",)):
                out[net][layer + ".num_batches_tracked"] = torch.tensor(self.calls_tracked[net], dtype=torch.long)
        order_s = ["fc1.weight", "fc1.bias", "norm1.weight", "norm1.bias", "norm1.running_mean", "norm1.running_var",
                   "norm1.num_batches_tracked", "fc2.weight", "fc2.bias", "norm2.weight", "norm2.bias", "norm2.running_mean",
                   "norm2.running_var", "norm2.num_batches_tracked", "fc3.weight", "fc3.bias"]
        order_p = ["fc1.weight", "fc1.bias", "norm1.weight", "norm1.bias", "norm1.running_mean", "norm1.running_var",
                   "norm1.num_batches_tracked", "fc2.weight", "fc2.bias"]
        return {k: out["s"][k] for k in order_s}, {k: out["p"][k] for k in order_p}

    def save(self, save_path):
        """save_siamese (:1556-1562): the two files `--attack` loads (torch.save of plain state_dicts)."""
        import os
        s, p = self.state_dicts()
        torch.save(s, os.path.join(save_path, "siamese_model_params.pkl"))
        torch.save(p, os.path.join(save_path, "predictor_net_params.pkl"))


def epoch_order(n_records, batch_size, shuffle=True, seed=234):
    """Record indices of each batch of one pass: the reference shuffles the three lists (ori / like / unlike) with the
    same `random.seed(234)` so they stay aligned (:276-290), then cuts int(n / batch_size) batches (data.py:379,406-408)."""
    idx = list(range(n_records))
    if shuffle:
        random.seed(seed)
        random.shuffle(idx)
    nb = n_records // batch_size
    return [idx[i * batch_size:(i + 1) * batch_size] for i in range(nb)]


def train_contrastive(trainer: ScorerTrainer, ori_latent, like_latent, unlike_latent, batch_size=128, epochs=1, log=print,
                      rank=0, world=1, save_path=None):
    """The epoch loop of train_contrastive (:890-947) over pre-computed latents: ori_latent [S,256],
    like_latent [S,Kp,256], unlike_latent [S,Kn,256] (device).  Returns per-epoch (loss sum, accuracy)."""
    S = ori_latent.shape[0]
    out = []
    per = batch_size // world
    if per * world != batch_size or per < 2:
        raise ValueError("batch_size %d must be a multiple of the world size %d with at least 2 rows per rank" % (batch_size, world))
    log("+" * 20 + "Training Siamese by lr={}".format(trainer.cfg.lr) + "+" * 20)
    for epoch in range(epochs):
        log("*" * 20 + "Start Running Epoch {}".format(epoch) + "*" * 20)
        batches = epoch_order(S, batch_size)
        loss_sum = torch.zeros(1, device=trainer.dev)
        hit = torch.zeros(1, device=trainer.dev)
        cnt = 0
        for rec in batches:
            sel = torch.as_tensor(rec[rank * per:(rank + 1) * per], device=trainer.dev)
            loss, choice = trainer.step(ori_latent[sel], like_latent[sel].transpose(0, 1), unlike_latent[sel].transpose(0, 1), world=world)
            loss_sum += loss
            hit += (choice == 0).sum()
            cnt += choice.numel()
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(loss_sum)
            loss_sum /= world
            dist.all_reduce(hit)
            cnt *= world
            dist.all_reduce(trainer.bn)
            trainer.bn /= world
        ls, acc = float(loss_sum.item()), float(hit.item()) / max(1, cnt)
        log("Epoch Loss={}".format(ls))
        out.append((ls, acc))
        if save_path is not None and rank == 0:
            trainer.save(save_path)
    return out
