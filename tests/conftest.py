import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracles work on small CPU tensors: with the full core count of a GPU box (hundreds of threads) every ATen call is thread hand-offs
    # -- 150 oracle steps at 3m dims took 465 s there and 4 s with 8 threads. Tests that time something set their own count.
    import torch
    if torch.get_num_threads() > 8:
        torch.set_num_threads(8)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(d, prefix):
    """{name: array} of the entries of `d` under `prefix/`, in stored (= named_parameters) order."""
    from collections import OrderedDict
    return OrderedDict((k[len(prefix):], v) for k, v in d.items() if k.startswith(prefix))


@pytest.fixture(scope="session")
def golden():
    return load_golden
