"""Fit a few seeded problems through the launch chain (n > 512) and save weights, tail coefficients and residuals: the two front ends of
the fit (MRBF_TAILQ=1: tail basis in three launches, small.hip TailQ; =0: the twelve-launch chain) are compared by a test that runs
this script once per setting.   usage: python tools/fit_dump.py out.npz"""
import importlib
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("morbit.jl_amd")

out = {}
cases = [(1500, 5, 3, "cubic"), (2048, 32, 1, "gaussian"), (1100, 64, 2, "multiquadric"), (3000, 33, 2, "cubic"), (640, 17, 16, "cubic"), (513, 3, 1, "cubic"), (777, 4, 3, "cubic"), (530, 64, 1, "gaussian")]
for ci, (n, d, k, kernel) in enumerate(cases):
    rng = np.random.default_rng(100 + ci)
    C = rng.random((n, d))
    Y = np.stack([np.sin(C @ rng.standard_normal(d)) + l * C[:, 0] for l in range(k)], 1)
    m = pkg.update_model(pkg.RbfConfig(kernel=kernel), C, Y)
    W, lam = m.weights, m.poly
    X = C[: min(n, 200)]
    V, _ = m.eval_sites(X)
    out["w%d" % ci] = W
    out["lam%d" % ci] = lam
    out["res%d" % ci] = np.abs(V - Y[: len(X)]).max() / np.abs(Y).max()
    out["path%d" % ci] = m.info["path"]
    m.free()
np.savez(sys.argv[1], **out)
print("ok", {k: float(v) for k, v in out.items() if k.startswith("res")})
print("paths", [int(out["path%d" % i]) for i in range(len(cases))])
