import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden():
    g = os.path.join(ROOT, "tests", "golden")
    arrs = np.load(os.path.join(g, "rbf_golden.npz"))
    with open(os.path.join(g, "rbf_golden.json")) as f:
        manifest = json.load(f)
    cases = []
    for c in manifest:
        pre = "c%03d_" % c["idx"]
        d = dict(c)
        for key in ("C", "Y", "X", "W", "Lam", "V", "J", "Phi", "Pi"):
            d[key] = arrs[pre + key]
        cases.append(d)
    return cases


@pytest.fixture(scope="session")
def golden():
    return load_golden()


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
