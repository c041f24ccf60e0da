#!/usr/bin/env python3
"""Latency of small evaluation calls (single points and small batches: backtracking, criticality loops) through mrbf_eval."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
for d, n in ((12, 512), (24, 700), (64, 2145), (64, 8192), (128, 257)):
    rng = np.random.default_rng(d)
    C = rng.random((n, d)); Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d
    mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
    for m in (1, 16, 256):
        X = rng.random((m, d))
        for _ in range(3): pkg.eval_models_at_sites(mod, None, X)
        t0 = time.perf_counter()
        for _ in range(20): v = pkg.eval_models_at_sites(mod, None, X)
        tv = (time.perf_counter() - t0) / 20 * 1e6
        for _ in range(3): pkg.eval_models_and_jacobians_at_sites(mod, None, X) if hasattr(pkg, "eval_models_and_jacobians_at_sites") else None
        print("d=%3d n=%5d m=%3d: values %.1f us per call" % (d, n, m, tv), flush=True)
    mod.free()
