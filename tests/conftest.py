import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library honours its MRBF_* environment switches only while MRBF_EXPERIMENTS=1 is set (include/mrbf.h "Environment switches").
# Several tests select schedule variants / earlier kernel forms / fault injection through those switches, so the gate is open for the
# test session; with no switch set the library runs its defaults exactly as without the gate (tests/test_abi.py pins the gate itself).
os.environ.setdefault("MRBF_EXPERIMENTS", "1")


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
    # extended-precision truth (tests/golden/make_truth.py, mpmath at 60 digits) stored as hi + lo doubles
    tpath = os.path.join(g, "rbf_truth.npz")
    if os.path.exists(tpath):
        tr = np.load(tpath)
        for d in cases:
            pre = "c%03d_" % d["idx"]
            if pre + "W_hi" in tr.files:
                d["truth"] = {key: (tr[pre + key + "_hi"], tr[pre + key + "_lo"]) for key in ("W", "Lam", "V", "J")}
    return cases


def dist_from_truth(x, truth_pair):
    """max |x - truth| / max |truth| with the truth given as (hi, lo) doubles: (x - hi) is exact or nearly so
    (Sterbenz) wherever x is close to hi, and lo restores what the rounding of hi took away."""
    hi, lo = truth_pair
    x = np.asarray(x, dtype=np.float64).reshape(hi.shape)
    return float(np.abs((x - hi) - lo).max() / max(np.abs(hi).max(), 1e-300))


@pytest.fixture(scope="session")
def golden():
    return load_golden()


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
