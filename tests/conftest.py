import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_f1():
    return load_json("f1_logits.json")


@pytest.fixture(scope="session")
def golden_f3():
    return load_json("f3_kats.json")["cases"], load_npz("f3_kats.npz")


@pytest.fixture(scope="session")
def weights0():
    """W(seed=0), verified against the SHA-256 recorded when the golden logits were made."""
    from af_mi355x import synth
    sd = synth.synthetic_state_dict(seed=0)
    want = load_json("f1_logits.json")["weights_sha256"]
    assert synth.state_dict_sha256(sd) == want, "synthetic weight recipe is not reproducible on this box"
    return sd
