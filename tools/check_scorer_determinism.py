#!/usr/bin/env python3
"""Debug aid: the scorer edit step and embedding, run repeatedly from the same state, must return the same bits."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
from style_transfer_amd import ops, weights as W

dev = torch.device("cuda", 0)
for fmt in ("f16x3", "f32"):
    sc = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                         W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev, gemm_format=fmt)
    torch.manual_seed(0)
    B = 32768
    x0 = torch.rand(B, 256, device=dev) * 8
    like = torch.rand(B * 10, 256, device=dev) * 8
    z1 = ops.scorer_embed(sc, like)
    z2 = ops.scorer_embed(sc, like)
    print(fmt, "embed deterministic:", torch.equal(z1, z2), "differing rows:", int((z1 != z2).any(1).sum()))
    lz = z1.view(B, 10, 256)
    uz = ops.scorer_embed(sc, torch.rand(B * 10, 256, device=dev) * 8).view(B, 10, 256)
    outs = []
    for rep in range(3):
        x = x0.clone()
        m, v = torch.zeros_like(x), torch.zeros_like(x)
        step = torch.zeros(B, dtype=torch.int32, device=dev)
        lr = torch.full((B,), 0.2, device=dev)
        act = torch.ones(B, dtype=torch.uint8, device=dev)
        gs = []
        for it in range(4):
            out = ops.scorer_edit_step(sc, x, m, v, lz, uz, lr, step, act, not_nll=True, want_grad=True, want_loss=True, want_scores=True)
            gs.append(out[1])
            if it == 0:
                first = (out[0].clone(), out[2].clone())
        outs_first = globals().setdefault("firsts", {})
        outs_first[(fmt, rep)] = first
        outs.append((x, torch.stack(gs)))
    for rep in (1, 2):
        dx = (outs[0][0] != outs[rep][0]).any(1)
        dg = (outs[0][1] != outs[rep][1]).any(2)
        print(fmt, "edit step rep", rep, "latents equal:", not bool(dx.any()), "rows differing:", int(dx.sum()),
              "first-step gradient rows differing:", int(dg[0].sum()), "rows:", dx.nonzero().flatten()[:8].tolist())
    if fmt == "f16x3":
        l0, s0 = firsts[(fmt, 0)]
        l1, s1 = firsts[(fmt, 1)]
        print("first step: loss rows differing", int((l0 != l1).sum()), "score rows differing", int((s0 != s1).any(1).sum()),
              "per column:", [(int((s0[:, c] != s1[:, c]).sum())) for c in range(6)])
        g0, g1 = outs[0][1][0], outs[1][1][0]
        rows = (g0 != g1).any(1).nonzero().flatten()
        if rows.numel() == 0:
            print("first-step gradients of the two runs are bit-identical in every row")
            continue
        import collections
        tiles = collections.Counter((rows // 32).tolist())
        print("tiles with differing rows:", len(tiles), "min tile", min(tiles), "max tile", max(tiles), "rows-in-tile histogram:",
              sorted(collections.Counter((rows % 32).tolist()).items()))
        r = int(rows[0])
        d = (g0[r] - g1[r]).abs()
        print("row", r, "differing elements", int((d > 0).sum()), "max |diff| %.3g" % float(d.max()), "max |g| %.3g" % float(g0[r].abs().max()),
              "cols", (d > 0).nonzero().flatten()[:16].tolist())
        for r in rows[:3].tolist():
            d = (g0[r] - g1[r])
            nz = (d != 0).nonzero().flatten()
            print(" row", r, "cols differing", nz.tolist()[:40], "n", len(nz))
            print("   rep0", g0[r][nz[:8]].tolist())
            print("   rep1", g1[r][nz[:8]].tolist())
        rel = ((g0[rows] - g1[rows]).abs().max(1).values / g0[rows].abs().max(1).values)
        print("relative difference over differing rows: median %.3g max %.3g" % (float(rel.median()), float(rel.max())))
