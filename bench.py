#!/usr/bin/env python3
"""bench.py — style-transferred sentences/sec on yelp-shaped synthetic batches, N GPUs of one node.

A "step" is one pass of the hot path (encode 1+10+10 sentences, like/unlike scorer embeddings, initial
decode, then up to 18 edit iterations of scorer fwd/bwd/Adam + greedy decode + change detection) over
one batch of synthetic yelp-shaped sentences per GPU, with the packed token arrays already resident
in HBM.  Sentences are independent, so ranks share nothing (weak scaling: fixed batch per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32768, help="sentences per GPU per step (KV cache: 241 KB per decode job, 13 jobs per sentence in the first round)")
    ap.add_argument("--shape", default="yelp", choices=["yelp", "amazon"])
    ap.add_argument("--tokens", default="zipf", choices=["zipf", "uniform", "real"],
                    help="synthetic token ids: unigram (Zipfian) from word_to_id.txt counts, uniform in [4,V), or the "
                         "1000 real yelp test sentences tiled to the batch size")
    ap.add_argument("--fixed-steps", action="store_true", help="no early exit at EOS: always 59 decode steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--iters-per-decode", default="12,6", help="edit iterations decoded per round: an int or a schedule like 9,5,4")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--sync-every", type=int, default=None, help="decode steps between two host reads of the live count (default: the package's)")
    ap.add_argument("--distinct-batches", action="store_true",
                    help="a different synthetic batch for every warm-up and timed step (BASELINE config 4: --steps 31 = 1 M distinct "
                         "origins per GPU), all resident in HBM before the timed region; default: one batch reused")
    ap.add_argument("--no-f32-reference-run", action="store_true",
                    help="skip the short exact-f32-format run that is reported beside the f16x3 value")
    ap.add_argument("--gemm-format", default="f16x3", choices=["f16x3", "f32"],
                    help="operand format of the AE weight GEMMs: 3-product f16 split (f32-grade results) or exact f32 MFMA")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from style_transfer_amd import _lib, weights as W
    from style_transfer_amd.attack import AttackConfig, AttackRunner
    from style_transfer_amd.data import AttackSet

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # one rank per GPU; ST_BENCH_BACKEND=gloo (rehearsal of the N > 1 control flow on a box with fewer GPUs than
    # ranks) maps the ranks onto the visible devices round-robin and reduces the timing on the CPU
    backend = os.environ.get("ST_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local if backend == "nccl" else local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # yelp shape: V = 9339; amazon shape: V = 19724 with its own surrogate AE (no amazon weights ship at all)
    ae = W.AEWeights(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_%s.safetensors" % a.shape)), dev,
                     gemm_format=a.gemm_format)
    scorer = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                             W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev)
    cfg = AttackConfig(not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2, log_decodes=False, early_exit=not a.fixed_steps,
                       iters_per_decode=(int(a.iters_per_decode) if "," not in a.iters_per_decode
                                         else tuple(int(v) for v in a.iters_per_decode.split(","))))
    if a.sync_every is not None:
        cfg.sync_every = a.sync_every
    runner = AttackRunner(ae, scorer, cfg, dev)
    B = a.batch
    from style_transfer_amd.data import load_word_counts
    if a.tokens == "real":
        files = [os.path.join(GOLD, a.shape, "neg10_pos10_test%d.json" % i) for i in (0, 1)]
        real = AttackSet.from_json(files, ae.vocab)
        data = real.subset([(i + 137 * rank) % len(real) for i in range(B)])
    else:
        counts = load_word_counts(os.path.join(GOLD, a.shape, "word_to_id.txt"), ae.vocab) if a.tokens == "zipf" else None
        data = AttackSet.synthetic(B, ae.vocab, a.shape, seed=1234 + rank, word_counts=counts)
    prep = runner.prepare(data)                       # tokens resident in HBM before the timed region
    preps = [prep] * (a.warmup + a.steps)
    if a.distinct_batches:
        if a.tokens == "real":
            raise SystemExit("--distinct-batches needs synthetic tokens")
        for k in range(1, a.warmup + a.steps):
            preps[k] = runner.prepare(AttackSet.synthetic(B, ae.vocab, a.shape, seed=1234 + rank + 7919 * k, word_counts=counts))
            if rank == 0 and k % 4 == 0:
                print("[bench] generated batch %d of %d" % (k + 1, a.warmup + a.steps), file=sys.stderr, flush=True)
    gidx = torch.arange(B, device=dev) % 1000

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    L = _lib.lib()
    for k in range(a.warmup):
        runner.run_prepared(preps[k], gidx, fetch=False)
    sync()
    L.st_profile_enable(_lib.ST_PROF_HALF_LAYER if a.gemm_format == "f16x3" else _lib.ST_PROF_GENERATOR)
    units = 0
    setup_s = loop_s = 0.0
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
    n_iter_mean = float(r.device["was_active"].sum().item()) / B

    # The same workload once more in the exact-f32 operand format (2 steps after 1 warm-up), reported beside
    # `value` so that the cost / benefit of the split format can be read from one line.
    f32_value = None
    if a.gemm_format == "f16x3" and not a.no_f32_reference_run:
        del r
        runner.ws.buf = None
        torch.cuda.empty_cache()
        ae32 = W.AEWeights(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_%s.safetensors" % a.shape)), dev,
                           gemm_format="f32")
        runner32 = AttackRunner(ae32, scorer, cfg, dev)
        runner32.run_prepared(prep, gidx, fetch=False)
        sync()
        t32 = time.perf_counter()
        for _ in range(2):
            runner32.run_prepared(prep, gidx, fetch=False)
        sync()
        d32 = torch.tensor([time.perf_counter() - t32], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(d32, op=dist.ReduceOp.MAX)
        f32_value = world * B * 2 / float(d32.item())
    tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    if rank == 0:
        achieved = gen_fl.value / (gen_ms.value * 1e-3) / 1e12 if gen_ms.value > 0 else 0.0
        split = a.gemm_format == "f16x3"
        peak = F16_MFMA_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS
        # HBM-side bytes of the roofline kernel come from rocprofv3 counter passes of THIS command (counters cannot be
        # read from inside the process): profiles/README.md gives the two commands; tools/traffic_from_pmc.py applies
        # the guide's gfx950 correction.  Only quoted for the configuration the passes were made on.
        traffic = traffic_src = alg_bytes = None
        tfile = os.path.join(ROOT, "profiles", "r01_pmc_bench_half_layer_traffic.json")
        if split and gen_n.value > 0:
            # per row: attention output + residual + cross-attention constant read, layer output written (f32 x 256)
            alg_bytes = gen_fl.value / gen_n.value / (2.0 * 256 * (256 + 2 * 1024)) * 4 * 1024
            default_cfg = (a.shape == "yelp" and a.tokens == "zipf" and B == 32768 and not a.fixed_steps
                           and a.iters_per_decode == "12,6")
            if default_cfg and os.path.exists(tfile):
                traffic = json.load(open(tfile))["bytes_per_launch"]
                traffic_src = "profiles/r01_pmc_bench_half_layer_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes of this command)"
        out = {
            "metric": "style-transferred sentences/sec, yelp-shape" if a.shape == "yelp" else "style-transferred sentences/sec, amazon-shape",
            "value": world * B * a.steps / dt, "unit": "sentences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (AE matrix products: f16 hi+lo split operands, 3 products per term, f32 accumulate)" if split else "f32",
            "data": "synthetic",
            "config": {"workload": "%s-shaped synthetic sentences (SURVEY.md §8d; token ids: %s), %d per GPU per step, flags --attack "
                                   "--not_nll --pos_victim_lr=0.2 --neg_victim_lr=0.2, max_iter 18, max_change 7, 10 like + 10 unlike, "
                                   "surrogate AE checkpoint + shipped (yelp) Siamese/Predictor weights, %s, Rec/Sim/Ref log decodes off"
                                   % (a.shape, {"zipf": "unigram/Zipfian from word_to_id.txt counts", "uniform": "uniform in [4,V)",
                                                "real": "the real yelp test set tiled"}[a.tokens], B,
                                      "fixed 59 decode steps" if a.fixed_steps else "early exit at EOS"),
                       "distinct_batches": bool(a.distinct_batches), "batch_per_gpu": B, "sentences_per_step": world * B, "parallelism": "sentence-sharded x%d" % world,
                       "mean_edit_iterations_per_sentence": n_iter_mean,
                       "setup_ms_per_step": setup_s / a.steps * 1e3, "edit_loop_ms_per_step": loop_s / a.steps * 1e3,
                       "iters_per_decode": cfg.iters_per_decode, "sync_every": cfg.sync_every, "gemm_format": a.gemm_format,
                       "same_workload_exact_f32_format_sentences_per_s": f32_value,
                       "edit_iteration_units_per_s": world * units / dt},
            # achieved = ALGORITHMIC flops of the timed kernel / its HIP-event time.  f16x3 issues 3 MFMA flops per
            # algorithmic flop, so frac <= 1/3 there by construction; mfma_issue_frac counts the issued ones.
            "roofline": {"bound": "mfma",
                         "kernel": ("rp_layer_tail_kernel (O-projection + residual + LayerNorm + FFN1 + FFN2 on one 64-row panel; "
                                    "2*rows*256*(256+2*d_ff) flop per launch; largest share of GPU time in this format)") if split
                                   else "rp_gemm_kernel<EPI_ARGMAX,4,false> (generator GEMM + fused argmax; 2*rows*256*V flop per launch)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_unit": "HBM-side bytes per launch",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "mfma_flops_per_algorithmic_flop": 3 if split else 1,
                         "mfma_issue_frac": achieved * (3 if split else 1) / peak,
                         "achieved_over_f32_mfma_peak": achieved / F32_MFMA_PEAK_TFLOPS,
                         "launches": int(gen_n.value), "avg_launch_us": gen_ms.value * 1e3 / max(1, gen_n.value),
                         "kernel_time_share": gen_ms.value * 1e-3 / dt},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, a.shape)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
