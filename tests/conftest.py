import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _np_sd(sd):
    return {k: v.numpy() for k, v in sd.items()}


@pytest.fixture(scope="session")
def ae_sd_torch():
    from style_transfer_amd import weights as W
    return W.expand_ae_state_dict(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_yelp.safetensors")))


@pytest.fixture(scope="session")
def ae_sd(ae_sd_torch):
    return _np_sd(ae_sd_torch)


@pytest.fixture(scope="session")
def scorer_sd_torch():
    from style_transfer_amd import weights as W
    return (W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
            W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")))


@pytest.fixture(scope="session")
def scorer_sd(scorer_sd_torch):
    return _np_sd(scorer_sd_torch[0]), _np_sd(scorer_sd_torch[1])


@pytest.fixture(scope="session")
def gold_notnll():
    return dict(np.load(os.path.join(GOLD, "attack_notnll.npz")))


@pytest.fixture(scope="session")
def id_to_word():
    from style_transfer_amd.data import load_word_dict_info
    return load_word_dict_info(os.path.join(GOLD, "yelp", "word_to_id.txt"), 5)[0]


def gold(name):
    return dict(np.load(os.path.join(GOLD, name)))
