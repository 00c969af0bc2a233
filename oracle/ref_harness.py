"""TEST INFRASTRUCTURE ONLY (oracle side) — imports the *reference* implementation in this container.

Used only by the fixture generators (oracle/make_*.py).  /root/reference does not exist on the GPU
box, so nothing under tests/ -m gpu, smoke() or bench.py imports this file.

The reference's data.py imports nltk at module top (data.py:7-8, only for calc_bleu, unused on the
attack path); nltk is not installed here, so an inert stub is injected before import (an ordinary
ModuleNotFoundError work-around).  main.py is never imported (argparse at import).
"""
import argparse
import os
import sys
import types

REF_ROOT = os.environ.get("ST_REFERENCE_ROOT", "/root/reference")
REF_YELP = os.path.join(REF_ROOT, "method", "yelp")


def available() -> bool:
    return os.path.isdir(REF_YELP)


def import_reference():
    """Return (model, data, siamese_contrastive) reference modules."""
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REF_ROOT)
    if "nltk" not in sys.modules:
        nltk = types.ModuleType("nltk")
        tr = types.ModuleType("nltk.translate")
        bs = types.ModuleType("nltk.translate.bleu_score")

        class SmoothingFunction:  # never called on the attack path
            def method1(self, *a, **k):
                raise NotImplementedError

        bs.SmoothingFunction = SmoothingFunction
        bs.sentence_bleu = lambda *a, **k: 0.0
        tr.bleu_score = bs
        nltk.translate = tr
        sys.modules["nltk"] = nltk
        sys.modules["nltk.translate"] = tr
        sys.modules["nltk.translate.bleu_score"] = bs
    if REF_YELP not in sys.path:
        sys.path.insert(0, REF_YELP)
    import model as ref_model  # noqa
    import data as ref_data  # noqa
    import siamese_contrastive as ref_sc  # noqa
    return ref_model, ref_data, ref_sc


def make_args(task="yelp", **over):
    """argparse.Namespace with main.py's defaults (main.py:35-126) for the attack path."""
    _, ref_data, _ = import_reference()
    data_path = os.path.join(REF_ROOT, "data", task, "processed_files") + "/"
    id_to_word, vocab_size = ref_data.load_word_dict_info(data_path + "word_to_id.txt", 5)
    ns = argparse.Namespace(
        id_pad=0, id_unk=1, id_bos=2, id_eos=3, task=task, word_dict_max_num=5, batch_size=128,
        max_sequence_length=60, num_layers_AE=2, transformer_model_size=256,
        transformer_ff_size=1024, latent_size=256, simplest_siamese=False, tau=0.1, bata=0.1,
        victim_optimizer="Adam", victim_lr=0.25, attack_max_iter=18, attack_max_change=7,
        reverse_order=False, just_use_one=False, attack=True, retext=False, not_nll=False,
        early_stop=False, dance=False, pos_num=None, neg_num=None, pos_victim_lr=0.25,
        neg_victim_lr=0.25, data_path=data_path, id_to_word=id_to_word, vocab_size=vocab_size,
        word_to_id={w: i for i, w in enumerate(id_to_word)},
        current_save_path=os.path.join(REF_YELP, "save", "origin") + "/",
    )
    for k, v in over.items():
        setattr(ns, k, v)
    return ns
