import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "golden.npz"))


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_enc():
    return np.load(os.path.join(GOLDEN, "golden_enc.npz"))


@pytest.fixture(scope="session")
def golden_enc_meta():
    with open(os.path.join(GOLDEN, "golden_enc_meta.json")) as f:
        return json.load(f)


def has_gpu():
    import torch
    return torch.cuda.is_available()
