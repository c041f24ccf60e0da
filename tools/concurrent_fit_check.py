import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, ".")
import morbit, importlib
pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib
lib = pkg.load()
P, n, d, k, m = 24, 1100, 10, 2, 500
keep = []; arr = (_lib.Problem * P)(); res = (_lib.Result * P)()
for p in range(P):
    rng = np.random.default_rng(40 + p)
    C = rng.random((n, d)); Y = np.ascontiguousarray(np.stack([np.sin(C.sum(axis=1)), (C ** 2).sum(axis=1)], axis=1))
    X = rng.random((m, d)); V = np.empty((m, k))
    keep.append((C, Y, X, V))
    arr[p] = _lib.Problem(n, m, d, k, 2, 1, 1.0, 0.5, C.ctypes.data_as(_lib.c_dp), Y.ctypes.data_as(_lib.c_dp), X.ctypes.data_as(_lib.c_dp), None, None, V.ctypes.data_as(_lib.c_dp), None)
for workers in (1, 2, 4):
    os.environ["MRBF_BATCH_WORKERS"] = str(workers)
    lib.mrbf_batch_run(1, None, P, arr, res)
    t0 = time.perf_counter(); rc = lib.mrbf_batch_run(1, None, P, arr, res); dt = time.perf_counter() - t0
    ok = all(res[p].status == 0 for p in range(P))
    print(f"workers {workers}: {P/dt:.0f} problems/s rc={rc} ok={ok} worst residual {max(res[p].fit.rel_residual for p in range(P)):.1e}", flush=True)
