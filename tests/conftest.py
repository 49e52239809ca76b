import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU-side checkers (torch float64 autograd, numpy) are small problems: on the GPU box's 256 hardware threads torch's
    # default pool (128) and the C oracle's OpenMP pool fight over the cores once both exist, and a 0.03 s reference took 48 s
    # (the training tests were 12 of the GPU suite's 14 minutes).  Eight threads are the fastest setting there and here.
    try:
        import torch
        torch.set_num_threads(min(8, os.cpu_count() or 8))
    except Exception:
        pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
