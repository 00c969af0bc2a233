#!/usr/bin/env python3
"""bench.py — style-transferred sentences/sec on yelp-shaped synthetic batches, N GPUs of one node.

A "step" is one pass of the hot path (encode 1+10+10 sentences, like/unlike scorer embeddings, initial
decode, then up to 18 edit iterations of scorer fwd/bwd/Adam + greedy decode + change detection) over
one batch of synthetic yelp-shaped sentences per GPU, with the packed token arrays already resident
in HBM.  Sentences are independent, so ranks share nothing (weak scaling: fixed batch per GPU).  Every
warm-up and timed step runs a DIFFERENT synthetic batch (BASELINE config 4: --steps 31 = 1 M distinct
origins per GPU).

    python bench.py --gpus N --steps K --warmup W        # N > 1 without a launcher: spawns one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

# MI355X_MICROARCH.md dense MFMA peaks (256 CU x 4 SIMD x 2.4 GHz): v_mfma_f32_32x32x2_f32 = 4096 flop / 64 clk,
# v_mfma_f32_32x32x16_f16 = 32768 flop / 32 clk
F32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2516.6
SUSTAINED_F16_16x16x32_TFLOPS = 1820.0   # measured: profiles/r02_mfma_sustained_rate.txt (32x32x16: 1610)


def cpu_baseline(seconds_budget=20.0, shape="yelp"):
    """The CPU oracle (a numpy port of the reference algorithm, KV-cached decode) on a bounded sample of
    the same workload, on this box's host cores.  Reported beside the GPU number; not the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import st_oracle as O
    from style_transfer_amd import weights as W
    from style_transfer_amd.data import AttackSet
    ae = {k: v.numpy() for k, v in W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_%s.safetensors" % shape))).items()}
    ssd = {k: v.numpy() for k, v in W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")).items()}
    psd = {k: v.numpy() for k, v in W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")).items()}
    V = ae["generator.proj.weight"].shape[0]
    data = AttackSet.synthetic(2048, V, shape, seed=1234)
    id_to_word = [str(i) for i in range(V)]
    t0 = time.perf_counter()
    done = 0
    for s in range(len(data)):
        enc = lambda sents: O.encode_latent(ae, O.pad_sequences(sents, V))
        ori = enc([data.origin[s]])
        like, unlike = enc(data.like[s]), enc(data.unlike[s])
        O.attack_one(ae, ssd, psd, ori, like, unlike, 0.2, id_to_word, lambda m: None, not_nll=True)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    return {"value": done / dt, "unit": "sentences/s", "cores": cores, "kind": "port",
            "sample": ("%d synthetic %s-shaped sentences through oracle/st_oracle.py (numpy f32, batch 1, KV-cached decode with "
                       "early exit, no log decodes), %.1f s; the imported reference itself ran 12.4 s/sentence on 8 cores "
                       "(BASELINE.md, yelp)") % (done, shape, dt)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=98304,
                    help="sentences per GPU per step (K/V cache: 121 KB per decode job with the first-layer table, 13 jobs per sentence in "
                         "the first round: 155 GB of the 288 GB at the default; same box: 49152 -> 165.5 k, 65536 -> 170.3 k, 81920 -> "
                         "172.0 k, 98304 -> 174.9 k, 114688 -> 165.1 k sentences/s)")
    ap.add_argument("--shape", default="yelp", choices=["yelp", "amazon"])
    ap.add_argument("--tokens", default="zipf", choices=["zipf", "uniform", "real"],
                    help="synthetic token ids: unigram (Zipfian) from word_to_id.txt counts, uniform in [4,V), or the "
                         "1000 real yelp test sentences tiled to the batch size")
    ap.add_argument("--fixed-steps", action="store_true", help="no early exit at EOS: always 59 decode steps")
    ap.add_argument("--log-decodes", action="store_true", help="include the 1+10+10 Rec/Sim/Ref log decodes in the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--iters-per-decode", default="12,6", help="edit iterations decoded per round: an int or a schedule like 9,5,4")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-defer-stragglers", action="store_true", help="A/B: sentences whose decodes do not end stay in their rounds (AttackConfig.defer_stragglers=False)")
    ap.add_argument("--no-split-by-length", action="store_true", help="A/B: mixed-length encoder chunks stay whole (AttackConfig.split_by_length=False)")
    ap.add_argument("--sync-every", type=int, default=None, help="decode steps between two host reads of the live count (default: the package's)")
    ap.add_argument("--same-batch", action="store_true",
                    help="reuse ONE synthetic batch for every step (round-1 behaviour); default: a different batch per step, all "
                         "resident in HBM before the timed region")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="skip the short extra runs reported in config (exact-f32 format, fixed 59 steps, log decodes on, end to end)")
    ap.add_argument("--gemm-format", default="f16x3", choices=["f16x3", "f32"],
                    help="operand format of the AE weight GEMMs: 3-product f16 split (f32-grade results) or exact f32 MFMA")
    ap.add_argument("--scorer-format", default=None, choices=["f16x3", "f32"], help="operand format of the style scorer (default: --gemm-format)")
    ap.add_argument("--no-l0-table", action="store_true", help="A/B: project the first decoder layer's q | k | v per step instead of "
                    "reading the (token, position) table")
    ap.add_argument("--config", default="attack", choices=["attack", "simloss"],
                    help="attack: the headline workload (BASELINE configs 1-4); simloss: BASELINE config 5, one contrastive constraint-loss "
                         "step (11 views, forward + backward) with B = --simloss-batch GLOBAL in-batch negatives, rows sharded over the GPUs")
    ap.add_argument("--simloss-batch", type=int, default=8192)
    ap.add_argument("--dry-run", action="store_true",
                    help="control-flow rehearsal without a GPU: launch, rendezvous (gloo), barrier, max-over-ranks timing and the "
                         "one JSON line; no kernel runs and `value` is null")
    return ap.parse_args(argv)


def self_launch(a, argv, poll_s=0.2, grace_s=10.0):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start one fresh worker process per GPU (before
    this process has made any GPU call); rank 0 prints the JSON line on the inherited stdout.  The workers rendezvous
    through a FILE store in a private temporary directory (no port to lose to another process).  The children are
    polled: when any of them exits non-zero the others are terminated (then killed) and that return code is passed on,
    so a dead worker cannot leave its siblings, and this launcher, waiting in a collective."""
    import shutil
    import tempfile
    td = tempfile.mkdtemp(prefix="st_bench_rdzv_")
    procs = {}
    try:
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                       ST_BENCH_INIT_FILE=os.path.join(td, "store"))
            for k in ("MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            procs[r] = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env)
        rc = 0
        live = dict(procs)
        while live and rc == 0:
            for r, p in list(live.items()):
                code = p.poll()
                if code is None:
                    continue
                del live[r]
                if code != 0:
                    rc = code if code > 0 else 128 - code       # killed by signal s: 128 + s, as a shell reports it
                    print("[bench] rank %d exited with %d: stopping the other %d workers" % (r, code, len(live)),
                          file=sys.stderr, flush=True)
                    break
            if live and rc == 0:
                time.sleep(poll_s)
        if live:                                                # a sibling failed: nobody is left to finish the collectives
            for p in live.values():
                p.terminate()
            end = time.monotonic() + grace_s
            for p in live.values():
                try:
                    p.wait(timeout=max(0.0, end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        return rc
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        shutil.rmtree(td, ignore_errors=True)


def init_group(backend, world, rank, dev=None):
    """One rendezvous for every N > 1 path of this file: torch.distributed.run's environment (MASTER_ADDR / MASTER_PORT),
    or the file store self_launch names; a FINITE timeout, so that a rank whose peer died fails instead of waiting."""
    import datetime
    import torch.distributed as dist
    kw = dict(timeout=datetime.timedelta(seconds=float(os.environ.get("ST_BENCH_TIMEOUT_S", "600"))))
    f = os.environ.get("ST_BENCH_INIT_FILE")
    if f:
        kw.update(init_method="file://" + f, rank=rank, world_size=world)
    if backend == "nccl":
        kw["device_id"] = dev
    dist.init_process_group(backend, **kw)


def algorithmic_flops(ae, B_total, units, enc_rows, enc_sq, kp, kn, work):
    """minimal formulation (KV cache, cross-attention folded, no log-softmax).
    work = (token_steps, keys_attended, decode_jobs) counted on the device."""
    d, dff, N, V = 256, ae.d_ff, ae.n_layers, ae.vocab
    layer = 2.0 * N * (4 * d * d + 2 * d * dff)
    tok, keys, jobs = work
    decode = tok * (layer + 2.0 * d * V) + keys * (2.0 * N * 2 * d) + jobs * (2.0 * N * 2 * d * d)
    encode = enc_rows * layer + enc_sq * (4.0 * N * d)
    scorer_mac = 256 * 512 + 512 * 512 + 512 * 256
    scorer = units * 2.0 * 2.0 * 575488 + B_total * (kp + kn) * 2.0 * scorer_mac
    return dict(decode=decode, encode=encode, scorer=scorer, total=decode + encode + scorer)


def simloss_cpu_baseline(views):
    """oracle/st_oracle.py simloss + simloss_grads (numpy, f64 accumulate) on a B = 2048 block, one view; reported per
    8192-sentence 11-view step by the flop ratio (the work is exactly quadratic in B and linear in views)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import st_oracle as O
    rng = np.random.default_rng(5)
    Bs = 2048
    ori, like = rng.random((Bs, 256), np.float32) * 10, rng.random((Bs, 256), np.float32) * 10
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 10.0:
        _, lse = O.simloss(ori, like, 0.1, 0)
        O.simloss_grads(ori, like, np.full(Bs, 1.0 / Bs), 0.1, 0)
        n += 1
    dt = (time.perf_counter() - t0) / n
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    return {"value": 2048 / (dt * views) / 4.0, "unit": "sentences/s", "cores": cores, "kind": "port",
            "sample": "oracle simloss + simloss_grads, B = 2048, one view, %d repeats of %.3f s; value scaled to B = 8192 and 11 views "
                      "(work per sentence 4x that of the sample)" % (n, dt)}


def simloss_main(a):
    """BASELINE config 5: the in-batch contrastive loss block of train_ae_model_constraint (siamese_contrastive.py:1055-1072)
    with B global negatives; every rank owns B / N rows (strong scaling: the global batch is fixed) and runs
    contrastive.constraint_loss_and_grads: one all_gather of the views' latents, its [B/N, B] row block of every view through
    st_simloss_fwd / st_simloss_bwd, one reduce-scatter of d_like."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    from style_transfer_amd import ops
    from style_transfer_amd.contrastive import constraint_loss_and_grads
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        init_group("nccl", world, rank, dev)
    B, views, tau = a.simloss_batch, 11, 0.1
    assert B % (32 * world) == 0
    b = B // world
    ops.simloss_format(a.gemm_format)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_run = a.warmup + a.steps
    # latents of a trained encoder are O(1..10) per coordinate with a common offset (tests/golden/simloss_b128.npz): rand * 10
    oris = [torch.rand(b, 256, device=dev, generator=gen) * 10 for _ in range(n_run)]
    likes = [torch.rand(views, b, 256, device=dev, generator=gen) * 10 for _ in range(n_run)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(a.warmup):
        constraint_loss_and_grads(oris[k], likes[k], tau)
    barrier()
    t0 = time.perf_counter()
    for k in range(a.warmup, n_run):
        loss, d_ori, d_like = constraint_loss_and_grads(oris[k], likes[k], tau)
    barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    sec = float(tt.item()) / a.steps
    # kernels only (no collectives, no host .item()): HIP events on the current stream around the 11 views' launches
    like_all = torch.rand(views, B, 256, device=dev, generator=gen) * 10
    g = torch.full((b,), 1.0 / (views * B), device=dev)
    d_like = torch.zeros_like(like_all)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for v in range(views):
        _, lse = ops.simloss_fwd(oris[0], like_all[v], tau, rank * b)
        ops.simloss_bwd(oris[0], like_all[v], lse, g, tau, rank * b, d_like[v])
    e1.record()
    torch.cuda.synchronize()
    ksec = e0.elapsed_time(e1) * 1e-3
    flops = views * 4 * 2.0 * b * B * 256            # per rank: S = ou.ku^T forward, S again + dS.ku + dS^T.ou backward
    if rank == 0:
        out = {"metric": "contrastive constraint-loss sentences/sec (in-batch InfoNCE, global negatives, forward + backward)",
               "value": B / sec, "unit": "sentences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32 (split-f16 MFMA operands, f32 accumulate)" if a.gemm_format == "f16x3" else "f32",
               "data": "synthetic",
               "config": {"workload": "BASELINE config 5: %d views x [%d, %d] similarity block per GPU, width 256, tau %.1f; all_gather of the "
                                      "views + reduce-scatter of d_like when N > 1" % (views, b, B, tau),
                          "global_batch": B, "rows_per_gpu": b, "views": views, "gemm_format": a.gemm_format,
                          "kernels_only_ms": ksec * 1e3},
               "roofline": {"bound": "mfma", "achieved": flops / ksec / 1e12,
                            "peak": F16_MFMA_PEAK_TFLOPS if a.gemm_format == "f16x3" else F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": flops / ksec / 1e12 / (F16_MFMA_PEAK_TFLOPS if a.gemm_format == "f16x3" else F32_MFMA_PEAK_TFLOPS),
                            "traffic": None,
                            "note": "algorithmic flops = views x 4 GEMM-sized products x 2 b B 256; the f16x3 form issues 3 MFMA products "
                                    "per algorithmic one and recomputes S once more for the transposed block (15 issued per 4 algorithmic)"}}
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = simloss_cpu_baseline(views)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a, argv)
    if a.config == "simloss" and not a.dry_run:
        return simloss_main(a)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)
    if a.dry_run:
        die = os.environ.get("ST_BENCH_TEST_DIE_RANK")          # tests: this rank dies before the rendezvous
        if die is not None and int(die) == rank:
            sys.exit(3)
        if world > 1:
            init_group("gloo", world, rank)
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "style-transferred sentences/sec, yelp-shape", "value": None, "unit": "sentences/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": float(tt.item()) * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": "dry run: no kernels", "dry_run": True}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    from style_transfer_amd import _lib, weights as W
    from style_transfer_amd.attack import AttackConfig, AttackRunner
    from style_transfer_amd.data import AttackSet, load_word_counts, load_word_dict_info
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # one rank per GPU; ST_BENCH_BACKEND=gloo (rehearsal of the N > 1 control flow on a box with fewer GPUs than
    # ranks) maps the ranks onto the visible devices round-robin and reduces the timing on the CPU
    backend = os.environ.get("ST_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local if backend == "nccl" else local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        init_group(backend, world, rank, dev)

    # yelp shape: V = 9339; amazon shape: V = 19724 with its own surrogate AE (no amazon weights ship at all)
    ae_file = os.path.join(GOLD, "surrogate_ae_%s.safetensors" % a.shape)
    ae = W.AEWeights(W.load_state_dict_file(ae_file), dev, gemm_format=a.gemm_format, l0_table=not a.no_l0_table)
    scorer = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                             W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev,
                             gemm_format=a.scorer_format or a.gemm_format)
    ipd = int(a.iters_per_decode) if "," not in a.iters_per_decode else tuple(int(v) for v in a.iters_per_decode.split(","))
    cfg = AttackConfig(not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2, log_decodes=a.log_decodes, early_exit=not a.fixed_steps,
                       iters_per_decode=ipd)
    if a.sync_every is not None:
        cfg.sync_every = a.sync_every
    if a.no_defer_stragglers:
        cfg.defer_stragglers = False
    if a.no_split_by_length:
        cfg.split_by_length = False
    runner = AttackRunner(ae, scorer, cfg, dev)
    B = a.batch
    counts = None
    if a.tokens == "real":
        files = [os.path.join(GOLD, a.shape, "neg10_pos10_test%d.json" % i) for i in (0, 1)]
        real = AttackSet.from_json(files, ae.vocab)
        make = lambda k: real.subset([(i + 137 * rank + 7919 * k) % len(real) for i in range(B)])
    else:
        counts = load_word_counts(os.path.join(GOLD, a.shape, "word_to_id.txt"), ae.vocab) if a.tokens == "zipf" else None
        make = lambda k: AttackSet.synthetic(B, ae.vocab, a.shape, seed=1234 + rank + 7919 * k, word_counts=counts)
    n_run = a.warmup + a.steps
    sets = [make(0)] * n_run if a.same_batch else [make(k) for k in range(n_run)]
    preps = [runner.prepare(sets[0])] * n_run if a.same_batch else [runner.prepare(s) for s in sets]   # resident in HBM
    gidx = torch.arange(B, device=dev) % 1000
    # encoder work of the timed steps (host-known): token rows and sum of squared sentence lengths
    enc_rows = enc_sq = 0
    for s in sets[a.warmup:]:
        for r in (s.origin_r, s.like_r, s.unlike_r):
            ln = r.lens.astype(np.float64)
            enc_rows += float(ln.sum())
            enc_sq += float((ln * ln).sum())
    kp, kn = sets[0].n_pos, sets[0].n_neg

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def reduce_max(x):
        t = torch.tensor([x], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(xs):
        t = torch.tensor(xs, device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    L = _lib.lib()
    work = (C.c_ulonglong * 4)()
    for k in range(a.warmup):
        runner.run_prepared(preps[k], gidx, fetch=False)
    sync()
    L.st_work_counters_read(work, 1)
    L.st_profile_enable(_lib.ST_PROF_HALF_LAYER if a.gemm_format == "f16x3" else _lib.ST_PROF_GENERATOR)
    units = 0
    setup_s = loop_s = 0.0
    n_iter_sum = 0.0
    t0 = time.perf_counter()
    for k in range(a.steps):
        r = runner.run_prepared(preps[a.warmup + k], gidx, fetch=False)
        units += r.timing["units"]
        setup_s += r.timing["setup_s"]
        loop_s += r.timing["loop_s"]
    sync()
    dt = time.perf_counter() - t0
    L.st_profile_enable(0)
    gen_ms, gen_fl, gen_n = C.c_double(), C.c_double(), C.c_longlong()
    L.st_profile_read(C.byref(gen_ms), C.byref(gen_fl), C.byref(gen_n), 1)
    L.st_work_counters_read(work, 1)
    n_iter_mean = float(r.device["was_active"].sum().item()) / B
    del r
    dt = reduce_max(dt)
    tot = reduce_sum([float(work[0]), float(work[1]), float(work[2]), float(units), enc_rows, enc_sq])

    # ---- side runs (one GPU only): the same workload under the other configurations SURVEY 8d asks to report ----
    side = {}
    if world == 1 and not a.no_side_runs and a.gemm_format == "f16x3" and not a.fixed_steps and not a.log_decodes:
        def timed(rn, pp, n=2):
            rn.run_prepared(pp[0], gidx[:pp[0]["B"]], fetch=False)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for i in range(n):
                rn.run_prepared(pp[i % len(pp)], gidx[:pp[0]["B"]], fetch=False)
            torch.cuda.synchronize(dev)
            return pp[0]["B"] * n / (time.perf_counter() - t)

        def fresh():
            runner.ws.buf = None
            torch.cuda.empty_cache()
        # (1) exact-f32 operand format, same batches
        fresh()
        ae32 = W.AEWeights(W.load_state_dict_file(ae_file), dev, gemm_format="f32")
        sc32 = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                               W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev, gemm_format="f32")
        r32 = AttackRunner(ae32, sc32, cfg, dev)
        # (no first-layer table in this format: 241 KB of K/V cache per job, so at most 32768 sentences per batch)
        B32 = min(B, 32768)
        side["exact_f32_format_sentences_per_s"] = timed(r32, [runner.prepare(s.span(0, B32)) for s in sets[:2]])
        side["exact_f32_format_batch"] = B32
        r32.ws.buf = None
        del r32, ae32
        fresh()
        # (2) no early exit: the reference's fixed 59 decode steps (batch 16384: the f32 KV cache of 13 jobs x 59 keys)
        Bs = min(B, 16384)
        small = [runner.prepare(s.span(0, Bs)) for s in sets[:2]]
        cfgf = AttackConfig(**{**cfg.__dict__, "early_exit": False})
        rf = AttackRunner(ae, scorer, cfgf, dev)
        side["fixed_59_decode_steps_sentences_per_s"] = timed(rf, small, 1)
        side["fixed_59_decode_steps_batch"] = Bs
        rf.ws.buf = None
        del rf
        fresh()
        # (3) with the 1+10+10 Rec/Sim/Ref log decodes (batch 8192: 20 more decode jobs per sentence)
        Bl = min(B, 8192)
        smalll = [runner.prepare(s.span(0, Bl)) for s in sets[:2]]
        cfgl = AttackConfig(**{**cfg.__dict__, "log_decodes": True})
        rl = AttackRunner(ae, scorer, cfgl, dev)
        side["with_log_decodes_sentences_per_s"] = timed(rl, smalll, 2)
        rl2 = AttackRunner(ae, scorer, cfg, dev)
        side["without_log_decodes_same_batch_sentences_per_s"] = timed(rl2, smalll, 2)
        side["log_decodes_batch"] = Bl
        rl.ws.buf = rl2.ws.buf = None
        del rl, rl2
        fresh()
        # (3b) BASELINE config 2's regime: batch 256 (latency-bound: every decode step is ~25 launches on a few hundred rows)
        tiny = [runner.prepare(s.span(0, 256)) for s in sets[:2]]
        r256 = AttackRunner(ae, scorer, cfg, dev)
        side["batch_256_sentences_per_s"] = timed(r256, tiny, 4)
        r256.ws.buf = None
        del r256
        fresh()
        # (4) end to end: token arrays on the HOST -> pack -> H2D -> attack -> compact records D2H -> log text -> file,
        # software-pipelined over the batches (style_transfer_amd/pipeline.py), the product CLI's path
        import tempfile
        from style_transfer_amd.logfmt import LogWriter, VocabBytes, format_attack_block
        from style_transfer_amd.pipeline import run_shard
        from style_transfer_amd.data import Ragged
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # (the loader prints the reference's banner line)
            id_to_word, _ = load_word_dict_info(os.path.join(GOLD, a.shape, "word_to_id.txt"), 5)
        vocab = VocabBytes(id_to_word)
        n_e2e = min(len(sets), 4) if not a.same_batch else 1
        big = AttackSet(Ragged.concat([s.origin_r for s in sets[:n_e2e]]), Ragged.concat([s.like_r for s in sets[:n_e2e]]),
                        Ragged.concat([s.unlike_r for s in sets[:n_e2e]]), None, kp, kn)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "log")
            for rep in range(2):      # first pass warms the pinned buffers and threads
                log = LogWriter(path, echo=False)
                consume = lambda b0, b1, lean: log.write_block(format_attack_block(vocab, big.span(b0, b1), lean.to("cpu", pinned=True), b0))
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                run_shard(runner, big, 0, len(big), B, consume)
                log.close()
                e2e = len(big) / (time.perf_counter() - t)
                nbytes = os.path.getsize(path)
        side["end_to_end_sentences_per_s"] = e2e
        side["end_to_end_note"] = ("%d sentences in batches of %d: host token arrays -> pack -> H2D -> attack -> changed rows D2H -> "
                                   "log text (st_format_attack_log) -> file (%.0f MB), prefetch and drain threads overlapped "
                                   "with compute" % (len(big), B, nbytes / 1e6))

    def emit():
        """rank 0: the ONE JSON line, from the timed region's figures and whatever `side` holds by now"""
        if rank != 0:
            return
        if True:
            achieved = gen_fl.value / (gen_ms.value * 1e-3) / 1e12 if gen_ms.value > 0 else 0.0
            split = a.gemm_format == "f16x3"
            peak = F16_MFMA_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS
            # HBM-side bytes of the roofline kernel come from rocprofv3 counter passes of THIS command (counters cannot be
            # read from inside the process): profiles/README.md gives the two commands; tools/traffic_from_pmc.py applies
            # the guide's gfx950 correction.  Only quoted for the configuration the passes were made on.
            traffic = traffic_src = alg_bytes = None
            tfile = os.path.join(ROOT, "profiles", "r03_pmc_bench_half_layer_traffic.json")
            bfile = os.path.join(ROOT, "profiles", "r03_pmc_phase_mfma_busy.json")
            busy = None
            if split and gen_n.value > 0:
                # HBM bytes the kernel has to move, per row and layer (f32 x 256 = 1 KB each): attention output read, layer
                # output written, residual read (not in the first layer when it reads the tables: the kernel computes the
                # embedding row from the L2-resident table), cross-attention constant read (decoder only).  Rows: executed
                # token-steps x decoder layers + encoder token rows x encoder layers (the few steps below 64 live rows run
                # other kernels).
                tabs = ae.dec.l0_table_len > 0
                kb_dec = (3 if tabs else 4) + 4 * (ae.dec.n_layers - 1)
                kb_enc = (2 if tabs else 3) + 3 * (ae.enc.n_layers - 1)
                alg_bytes = (tot[0] * kb_dec + tot[4] * kb_enc) * 1024.0 / world / gen_n.value
                default_cfg = (a.shape == "yelp" and a.tokens == "zipf" and B == 98304 and not a.fixed_steps
                               and a.iters_per_decode == "12,6" and not a.log_decodes)
                if default_cfg and os.path.exists(tfile):
                    traffic = json.load(open(tfile))["bytes_per_launch"]
                    traffic_src = "profiles/r03_pmc_bench_half_layer_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes of this command)"
                if default_cfg and os.path.exists(bfile):
                    busy = json.load(open(bfile))
                    busy["source"] = "profiles/r03_pmc_phase_mfma_busy.json (SQ counter pass of this command, tools/summarize_pmc.py)"
            fl = algorithmic_flops(ae, world * B * a.steps, tot[3], tot[4], tot[5], kp, kn, tot[:3])
            whole = fl["total"] / dt / 1e12
            # share of the algorithmic flops that run as AE matrix products (3 MFMA flops issued per flop in f16x3)
            ae_share = (fl["decode"] + fl["encode"]) / fl["total"]
            out = {
                "metric": "style-transferred sentences/sec, yelp-shape" if a.shape == "yelp" else "style-transferred sentences/sec, amazon-shape",
                "value": world * B * a.steps / dt, "unit": "sentences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32 (AE matrix products: f16 hi+lo split operands, 3 products per term, f32 accumulate)" if split else "f32",
                "data": "synthetic",
                "config": {"workload": "%s-shaped synthetic sentences (SURVEY.md §8d; token ids: %s), %d per GPU per step, flags --attack "
                                       "--not_nll --pos_victim_lr=0.2 --neg_victim_lr=0.2, max_iter 18, max_change 7, 10 like + 10 unlike, "
                                       "surrogate AE checkpoint + shipped (yelp) Siamese/Predictor weights, %s, Rec/Sim/Ref log decodes %s"
                                       % (a.shape, {"zipf": "unigram/Zipfian from word_to_id.txt counts", "uniform": "uniform in [4,V)",
                                                    "real": "the real yelp test set tiled"}[a.tokens], B,
                                          "fixed 59 decode steps" if a.fixed_steps else "early exit at EOS", "on" if a.log_decodes else "off"),
                           "distinct_batches": not a.same_batch, "batch_per_gpu": B, "sentences_per_step": world * B,
                           "parallelism": "sentence-sharded x%d" % world,
                           "mean_edit_iterations_per_sentence": n_iter_mean,
                           "setup_ms_per_step": setup_s / a.steps * 1e3, "edit_loop_ms_per_step": loop_s / a.steps * 1e3,
                           "iters_per_decode": cfg.iters_per_decode, "sync_every": cfg.sync_every, "gemm_format": a.gemm_format,
                           "edit_iteration_units_per_s": tot[3] / dt,
                           "side_runs_same_workload_one_gpu": (side or None) if world == 1 else None,
                           "end_to_end_with_final_gather": (side or None) if world > 1 else None},
                # the whole timed region against the MFMA roofline: executed algorithmic flops / wall time
                "whole_step": {"executed_token_steps": tot[0], "keys_attended": tot[1], "decode_jobs_executed": tot[2],
                               "decode_jobs_needed": world * B * a.steps + tot[3],
                               "encoder_token_rows": tot[4],
                               "algorithmic_flops": fl, "algorithmic_tflops": whole, "peak": peak, "frac": whole / peak,
                               "mfma_issue_frac": whole * ((3 * ae_share + (1 - ae_share)) if split else 1) / peak,
                               # counters cannot be read from inside the process: quoted from the committed counter pass of this command
                               "mfma_busy": busy,
                               "note": "F_tok = 2[N(4d^2 + 2 d d_ff) + d V] + 2048 t per executed token-step (KV cache, cross-attention "
                                       "folded, no log-softmax) + encoder + scorer; token-steps counted on the device from the live "
                                       "counts; wall time includes every launch gap, host sync and the non-GEMM kernels"},
                # achieved = ALGORITHMIC flops of the timed kernel / its HIP-event time.  f16x3 issues 3 MFMA flops per
                # algorithmic flop, so frac <= 1/3 there by construction; mfma_issue_frac counts the issued ones.
                "roofline": {"bound": "mfma",
                             "kernel": ("rp_layer_tail16_kernel (O-projection + residual + LayerNorm + FFN1 + FFN2 on 64-row panels, persistent "
                                        "over panels, v_mfma_f32_16x16x32_f16; 2*rows*256*(256+2*d_ff) flop per launch; largest share of GPU "
                                        "time in this format)") if split
                                       else "rp_gemm_kernel<EPI_ARGMAX,4,false> (generator GEMM + fused argmax; 2*rows*256*V flop per launch)",
                             "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                             "frac": achieved / peak, "traffic": traffic, "traffic_unit": "HBM-side bytes per launch",
                             "traffic_source": traffic_src,
                             "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_note": ("algorithmic = attention output + residual + cross-attention constant read, output written, per "
                                          "row and layer; the first-layer launches compute their residual from the 9.5 MB embedding "
                                          "table, which is counted as cache-resident here but mostly misses the 4 MB of L2 per XCD "
                                          "(the measured traffic did not change when those launches stopped reading x): that and "
                                          "weight fragments that miss L2 are the excess; at 8 TB/s the traffic is %.2f ms of a %.2f ms "
                                          "launch" % (traffic / 8e9, gen_ms.value / max(1, gen_n.value))) if (split and traffic) else None,
                             "mfma_flops_per_algorithmic_flop": 3 if split else 1,
                             "mfma_issue_frac": achieved * (3 if split else 1) / peak,
                             "achieved_over_f32_mfma_peak": achieved / F32_MFMA_PEAK_TFLOPS,
                             # a bare loop of this MFMA shape on random operands sustains 1.82 PFLOP/s on this chip (power-managed
                             # clock; tools/kbench.cpp `p`, profiles/r02_mfma_sustained_rate.txt) - the guide's 2516.6 assumes 2.4 GHz
                             "mfma_issue_frac_of_sustained_rate": (achieved * 3 / SUSTAINED_F16_16x16x32_TFLOPS) if split else None,
                             "launches": int(gen_n.value), "avg_launch_us": gen_ms.value * 1e3 / max(1, gen_n.value),
                             "kernel_time_share": gen_ms.value * 1e-3 / dt},
            }
            if world == 1 and not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, a.shape)
            print(json.dumps(out), flush=True)

    # The N > 1 side run below is the only part of this file that moves data through RCCL (send / recv of the lean records).
    # Its figures are a bonus: if it does not finish in time (a collective that waits on a peer forever), rank 0 still prints
    # the line with the timed region's numbers and every rank leaves with 0 instead of taking the whole run down.
    import threading
    wd = None
    if world > 1 and not a.no_side_runs and not a.fixed_steps:
        def give_up():
            side.clear()
            side["error"] = "end-to-end side run did not finish within %s s: skipped" % os.environ.get("ST_BENCH_E2E_TIMEOUT_S", "300")
            try:
                emit()
            finally:
                sys.stdout.flush()
                os._exit(0)
        wd = threading.Timer(float(os.environ.get("ST_BENCH_E2E_TIMEOUT_S", "300")), give_up)
        wd.daemon = True
        wd.start()
    # ---- N > 1: what BASELINE config 4 names ("... with final all-gather"): every rank streams its shard through the
    # product pipeline (host token arrays -> pack -> H2D -> attack -> lean records), the records of all ranks go to rank 0
    # (pipeline.gather_to_rank0: RCCL send / recv over xGMI), rank 0 formats the log in global order and writes the file.
    # Rank 0 knows every rank's origins, as every rank of the CLI holds the whole input file.
    if world > 1 and not a.no_side_runs and not a.fixed_steps:
        try:
            import contextlib
            import tempfile
            from style_transfer_amd.attack import LeanRecord
            from style_transfer_amd.data import Ragged
            from style_transfer_amd.logfmt import VocabBytes, format_attack_blocks
            from style_transfer_amd.pipeline import gather_to_rank0, run_shard
            n_e2e = 1          # one batch per rank: rank 0 formats and writes world * B sentences (~2.7 KB of log text each)
            big = AttackSet(Ragged.concat([s.origin_r for s in sets[:n_e2e]]), Ragged.concat([s.like_r for s in sets[:n_e2e]]),
                            Ragged.concat([s.unlike_r for s in sets[:n_e2e]]), None, kp, kn)
            everyone = vocab = None
            if rank == 0:
                with contextlib.redirect_stdout(sys.stderr):
                    id_to_word, _ = load_word_dict_info(os.path.join(GOLD, a.shape, "word_to_id.txt"), 5)
                vocab = VocabBytes(id_to_word)
                if a.tokens == "real":
                    origins = [real.subset([(i + 137 * r + 7919 * k) % len(real) for i in range(B)]).origin_r
                               for r in range(world) for k in range(n_e2e)]
                else:
                    origins = [AttackSet.synthetic(B, ae.vocab, a.shape, seed=1234 + r + 7919 * k, word_counts=counts, origins_only=True).origin_r
                               for r in range(world) for k in range(n_e2e)]
                assert np.array_equal(origins[0].tokens, sets[0].origin_r.tokens)
                none = Ragged(np.zeros(0, np.int32), np.zeros(1, np.int64))
                everyone = AttackSet(Ragged.concat(origins), none, none, None, 0, 0)
                everyone_warm = AttackSet(Ragged.concat([o.span(0, min(len(o), 4096)) for o in origins]), none, none, None, 0, 0)
            gdev = dev if backend == "nccl" else "cpu"
            with tempfile.TemporaryDirectory() as td:
                for rep in range(2):          # the first pass (a small span) warms pinned buffers, threads and the RCCL channels
                    kept = []
                    hi = min(len(big), 4096) if rep == 0 else len(big)
                    sync()
                    t = time.perf_counter()
                    run_shard(runner, big, 0, hi, B, lambda b0, b1, lean: kept.append(lean))
                    torch.cuda.synchronize(dev)
                    t_c = time.perf_counter()
                    full = gather_to_rank0(LeanRecord.cat(kept), rank, world, gdev)
                    if backend == "nccl":
                        torch.cuda.synchronize(dev)
                    t_g = time.perf_counter()
                    nbytes = 0
                    if rank == 0:
                        full = full.cpu()
                        with open(os.path.join(td, "log"), "wb") as f:
                            for block in format_attack_blocks(vocab, everyone if rep else everyone_warm, full, 0):
                                f.write(memoryview(block))
                                nbytes += block.size
                    t_w = time.perf_counter()
                    sync()
                    t_e = time.perf_counter()
                    del full, kept
            tot_s = reduce_max(t_e - t)
            side["end_to_end_sentences_per_s"] = world * len(big) / tot_s
            side["compute_ms"] = reduce_max(t_c - t) * 1e3
            side["gather_ms"] = reduce_max(t_g - t_c) * 1e3
            side["log_write_ms"] = (t_w - t_g) * 1e3
            side["end_to_end_note"] = ("%d sentences per rank in batches of %d through pipeline.run_shard on every rank (host token arrays -> "
                                       "pack -> H2D -> attack), lean records of all ranks to rank 0 (gather_ms: all_gather of sizes + "
                                       "exact-size send / recv), rank 0 formats %d sentences in global order and writes the file "
                                       "(%.0f MB, log_write_ms); times are the max over ranks" % (len(big), B, world * len(big), nbytes / 1e6))

        except Exception as e:      # (a full temporary directory, a failed collective): the timed region's line still goes out
            import traceback
            traceback.print_exc()
            side.clear()
            side["error"] = "end-to-end side run failed on rank %d: %s: %s" % (rank, type(e).__name__, e)
            if wd is not None:
                wd.cancel()
            try:
                emit()
            finally:
                sys.stdout.flush()
                os._exit(0)      # peers may be waiting in a collective this rank will never join: their watchdogs end them

    if wd is not None:
        wd.cancel()
    emit()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
