"""Shard -> batches -> records: the data-parallel shell around AttackRunner (what attack_contrastive's outer loop,
/root/reference/method/yelp/siamese_contrastive.py:695-743, becomes when the sentences move in batches).

* shard_range: contiguous split of the global sentence range over ranks (one process per GPU, no communication
  while computing; the lr switch and the log order depend on the GLOBAL index, which every batch carries).
* run_shard: the batches of one shard, software-pipelined on the host: a prefetch thread packs batch k+1 and
  copies it to the GPU on a side stream while batch k computes; the consumer callback (record download, log
  formatting, file write) runs on a drain thread, in order, while batch k+1 computes.
* gather_to_rank0: the one collective step of a multi-GPU run.  Only rank 0 writes the log, so only rank 0
  receives, and what travels is the compact LeanRecord (changed rows only) with exact sizes: one all_gather of the
  field sizes, then point-to-point send / recv per field (RCCL over xGMI on GPUs; gloo on CPU in the tests).
"""
from concurrent.futures import ThreadPoolExecutor

import torch

from .attack import LeanRecord


def shard_range(n, rank, world):
    """contiguous split of range(n) over ranks, sizes differing by at most one"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_shard(runner, data, lo, hi, batch, consume, want_latent=False, prefetch=True):
    """Run sentences [lo, hi) of `data` in batches of `batch`.  consume(b0, b1, lean) gets the global range of each
    batch and its LeanRecord ON THE DEVICE, in order, on a drain thread (prefetch=True) or inline.
    Returns the timing totals of the batches."""
    dev = runner.device
    spans = [(b0, min(hi, b0 + batch)) for b0 in range(lo, hi, batch)]
    totals = dict(setup_s=0.0, loop_s=0.0, units=0, deferred_sentences=0)
    if not spans:
        return totals
    side = torch.cuda.Stream(device=dev) if prefetch else None

    def prep(b0, b1):
        if side is None:
            return runner.prepare(data.span(b0, b1))
        with torch.cuda.stream(side):
            p = runner.prepare(data.span(b0, b1))
            side.synchronize()          # resident before the compute stream touches it
        return p

    pre = ThreadPoolExecutor(1) if prefetch else None
    drain = ThreadPoolExecutor(1) if prefetch else None
    pending = []
    try:
        nxt = pre.submit(prep, *spans[0]) if pre else None
        for k, (b0, b1) in enumerate(spans):
            p = nxt.result() if pre else prep(b0, b1)
            if pre and k + 1 < len(spans):
                nxt = pre.submit(prep, *spans[k + 1])
            gidx = torch.arange(b0, b1, device=dev)
            res = runner.run_prepared(p, gidx, fetch=False)
            lean = LeanRecord.from_device(res.device, want_latent=want_latent)
            for key in totals:
                totals[key] += res.timing.get(key, 0)
            del res, p
            if drain:
                ev = torch.cuda.Event()
                ev.record()
                pending.append(drain.submit(_drain_one, consume, b0, b1, lean, ev, dev))
                while len(pending) > 2:          # at most two batches of records waiting for the writer
                    pending.pop(0).result()
            else:
                consume(b0, b1, lean)
        for f in pending:
            f.result()
    finally:
        for ex in (pre, drain):
            if ex:
                ex.shutdown(wait=True)
    return totals


_drain_streams = {}


def _drain_one(consume, b0, b1, lean, ev, dev):
    st = _drain_streams.get(dev)
    if st is None:
        st = _drain_streams[dev] = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        st.wait_event(ev)
        consume(b0, b1, lean)


def gather_to_rank0(rec: LeanRecord, rank, world, device):
    """-> on rank 0 the rank-ordered concatenation of every rank's LeanRecord (tensors on `device`), None elsewhere.
    `device` is the device the backend moves (the GPU for nccl = RCCL, "cpu" for gloo)."""
    import torch.distributed as dist
    if world == 1:
        return rec
    rec = rec.to(device)
    fields = LeanRecord.FIELDS
    mine = torch.tensor([-1 if getattr(rec, f) is None else getattr(rec, f).shape[0] for f in fields], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = torch.stack(sizes).cpu().tolist()          # [world][field] rows (-1 = field absent)
    present = [all(sz[i] >= 0 for sz in sizes) for i in range(len(fields))]
    if rank != 0:
        for i, f in enumerate(fields):
            t = getattr(rec, f)
            if present[i] and t.shape[0] > 0:
                dist.send(t.contiguous(), 0)
        return None
    out = {}
    for i, f in enumerate(fields):
        if not present[i]:
            out[f] = None
            continue
        t = getattr(rec, f)
        parts = [t]
        for r in range(1, world):
            buf = torch.empty((sizes[r][i],) + tuple(t.shape[1:]), dtype=t.dtype, device=device)
            if sizes[r][i] > 0:
                dist.recv(buf, r)
            parts.append(buf)
        out[f] = torch.cat(parts, 0)
    return LeanRecord(**out)
