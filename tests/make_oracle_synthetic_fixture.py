#!/usr/bin/env python3
"""Fixture generator (CPU only, runs the ORACLE, hence under tests/): the lines oracle/st_oracle.py logs for 512 picks out
of the 2^20 synthetic yelp-shaped sentences of the config-4 test (seed 2024), sentence by sentence, batch 1 -- what
tests/test_gpu_attack.py::test_config4_... compares the GPU pipeline with.  Writes tests/golden/oracle_synthetic_picks.json.

    OMP_NUM_THREADS=1 python tests/make_oracle_synthetic_fixture.py --workers 8
"""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")
S = 1 << 20
_state = {}


def picks_of_the_test():
    """the first 64 are the picks the test has always used; 448 more from a second stream"""
    a = np.sort(np.random.default_rng(99).choice(S, 64, replace=False))
    b = np.random.default_rng(100).choice(S, 480, replace=False)
    b = np.sort(np.setdiff1d(b, a)[:448])
    return [int(v) for v in a], [int(v) for v in b]


def oracle_lines(job):
    import st_oracle as O
    from style_transfer_amd import weights as W
    if not _state:
        npsd = lambda sd: {k: v.numpy() for k, v in sd.items()}
        _state["ae"] = npsd(W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_yelp.safetensors"))))
        _state["s"] = npsd(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")))
        _state["p"] = npsd(W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")))
    ae, s, p = _state["ae"], _state["s"], _state["p"]
    g, origin, like, unlike, V = job
    enc = lambda sents: O.encode_latent(ae, O.pad_sequences(sents, V))
    lines = []
    O.attack_one(ae, s, p, enc([origin]), enc(like), enc(unlike), 0.2, [str(i) for i in range(V)], lines.append, not_nll=True)
    return g, lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=8)
    a = ap.parse_args()
    from style_transfer_amd.data import AttackSet, load_word_counts
    V = 9339
    counts = load_word_counts(os.path.join(GOLD, "yelp", "word_to_id.txt"), V)
    data = AttackSet.synthetic(S, V, "yelp", seed=2024, word_counts=counts)
    first, more = picks_of_the_test()
    jobs = [(g, data.origin[g], data.like[g], data.unlike[g], V) for g in first + more]
    del data
    out = {}
    with ProcessPoolExecutor(a.workers) as ex:
        for k, (g, lines) in enumerate(ex.map(oracle_lines, jobs, chunksize=2)):
            out[str(g)] = lines
            if k % 32 == 31:
                print("%d of %d" % (k + 1, len(jobs)), flush=True)
    with open(os.path.join(GOLD, "oracle_synthetic_picks.json"), "w") as f:
        json.dump({"seed": 2024, "sentences": S, "first": first, "more": more, "lines": out}, f)
    print("written")


if __name__ == "__main__":
    main()
