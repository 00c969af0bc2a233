"""In-batch InfoNCE constraint loss with GLOBAL negatives, sharded over ranks (BASELINE config 5).

Replaces the loss block of train_ae_model_constraint
(/root/reference/method/yelp/siamese_contrastive.py:1055-1072):

    like_latent = cat([like views..., second pass of ori])            # [n_views, B, 256]
    for v: sim = cosine_similarity(ori[:,None,:], like[v]) / tau      # [B, B]
           loss_v = NLL(log_softmax(sim), arange(B))
    constraint_loss = mean_v loss_v

With B split over R ranks (rank r owns rows [r*b, (r+1)*b)): every rank all-gathers the `like` rows of
each view (one fused all_gather of all views: small, latency-bound messages over xGMI), computes its
[b, B] row block with complete rows — the softmax needs no further exchange — and in the backward pass
reduce-scatters d_like (every rank holds a partial sum for ALL B rows).  The scalar loss is
all-reduced.  `fwd`/`bwd` default to the HIP kernels (ops.simloss_fwd / ops.simloss_bwd); the CPU test
injects the oracle (and a gloo stand-in for reduce-scatter) to check the sharding arithmetic.
"""
import torch
import torch.distributed as dist


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def constraint_loss_and_grads(ori, like_views, tau=0.1, fwd=None, bwd=None, reduce_scatter=None):
    """ori [b,256] (this rank's rows), like_views [n_views,b,256] (this rank's rows of every view).
    -> (loss: python float = mean over views and ALL rows, d_ori [b,256], d_like_views [n_views,b,256])
    Gradients are those of the global mean loss, as `constraint_loss.backward()` would give.
    `reduce_scatter(out_flat, in_flat)` defaults to RCCL's reduce_scatter_tensor; a backend without one (the
    gloo CPU test) injects its own, as it injects fwd/bwd."""
    if fwd is None or bwd is None:
        from . import ops
        fwd = fwd or ops.simloss_fwd
        bwd = bwd or ops.simloss_bwd
    rank, world = _world()
    n_views, b, d = like_views.shape
    B = b * world
    if world > 1:
        gathered = torch.empty(world, n_views, b, d, dtype=like_views.dtype, device=like_views.device)
        dist.all_gather_into_tensor(gathered.view(-1), like_views.contiguous().view(-1))
        like_all = gathered.permute(1, 0, 2, 3).reshape(n_views, B, d).contiguous()
    else:
        like_all = like_views.contiguous()
    g = torch.full((b,), 1.0 / (n_views * B), dtype=ori.dtype, device=ori.device)   # d(mean)/d(loss row)
    loss_sum = torch.zeros((), dtype=torch.float64, device=ori.device)
    d_ori = torch.zeros_like(ori)
    d_like_all = torch.zeros_like(like_all)
    for v in range(n_views):
        loss_rows, lse = fwd(ori, like_all[v], tau, rank * b)
        loss_sum += loss_rows.double().sum()
        do, _ = bwd(ori, like_all[v], lse, g, tau, rank * b, d_like_all[v])
        d_ori += do
    if world > 1:
        dist.all_reduce(loss_sum)
        d_like = torch.empty(n_views, b, d, dtype=like_views.dtype, device=like_views.device)
        # reduce-scatter d_like over the row blocks: rank r keeps rows [r*b, (r+1)*b) of every view
        send = d_like_all.view(n_views, world, b, d).permute(1, 0, 2, 3).contiguous()
        (reduce_scatter or dist.reduce_scatter_tensor)(d_like.view(-1), send.view(-1))
    else:
        d_like = d_like_all
    return float(loss_sum.item()) / (n_views * B), d_ori, d_like
