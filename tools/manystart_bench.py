#!/usr/bin/env python3
"""Many-start throughput (BASELINE.json configs[3], C4: d=128, n=257 cubic, 64 starts) through mrbf_batch_run."""
import ctypes
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg  # noqa: E402
from morbit.jl_amd import _lib  # noqa: E402

lib = pkg.load()
P, n, d, k, m = 64, 257, 128, 2, 6450
keep = []
arr = (_lib.Problem * P)()
res = (_lib.Result * P)()
for p in range(P):
    rng = np.random.default_rng(40 + p)
    C = rng.random((n, d))
    f1 = C[:, 0]
    g = 1.0 + 9.0 * C[:, 1:].sum(axis=1) / (d - 1)
    Y = np.ascontiguousarray(np.stack([f1, g * (1.0 - np.sqrt(f1 / g))], axis=1))
    X = rng.random((m, d))
    V = np.empty((m, k))
    keep.append((C, Y, X, V))
    arr[p] = _lib.Problem(n, m, d, k, 0, 1, 3.0, 0.0, C.ctypes.data_as(_lib.c_dp), Y.ctypes.data_as(_lib.c_dp),
                          X.ctypes.data_as(_lib.c_dp), None, None, V.ctypes.data_as(_lib.c_dp), None)
for workers in (1, 2, 4, 8):
    os.environ["MRBF_BATCH_WORKERS"] = str(workers)
    lib.mrbf_batch_run(1, None, P, arr, res)  # warm
    t0 = time.perf_counter()
    rc = lib.mrbf_batch_run(1, None, P, arr, res)
    dt = time.perf_counter() - t0
    ok = all(res[p].status == 0 for p in range(P))
    print("workers/GPU %d: %d problems in %.1f ms -> %.0f problems/s (rc=%d ok=%s, worst residual %.1e)"
          % (workers, P, dt * 1e3, P / dt, rc, ok, max(res[p].fit.rel_residual for p in range(P))), flush=True)
