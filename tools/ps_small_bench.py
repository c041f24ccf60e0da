#!/usr/bin/env python3
"""One Pascoletti-Serafini step with Morbit's default budgets at the small dimensions Morbit's own examples run (d = 2 .. 30)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import pascoletti_serafini as ps
for d, n in ((2, 20), (4, 200), (12, 512), (24, 700), (30, 1500)):
    rng = np.random.default_rng(4 + d)
    C = rng.random((n, d))
    Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1) + 0.1 * np.sin(5 * C[:, 0])], axis=1)
    mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
    x = np.full(d, 0.5); lb, ub = x - 0.12, x + 0.12
    fx = pkg.eval_models_at_sites(mod, None, x[None, :])[0]
    best = 1e9
    for rep in range(4):
        st = {}
        t0 = time.perf_counter()
        o = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=1, stats=st)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print("d=%d n=%d: %.2f ms, %d evaluations in %d generations, omega %.5f" % (d, n, best, st["evals_ideal"] + st["evals_ps"], st["generations"], o[0]), flush=True)
    mod.free()
