"""Quick correctness + timing check of the persistent Cholesky (impl 3) against LAPACK and the host-driven path (impl 2)."""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import morbit  # noqa
import importlib

pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib

ctx = pkg.default_context()
sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128, 256, 640, 1024, 2048]
impls = [int(s) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for n in sizes:
    rng = np.random.Generator(np.random.PCG64(n))
    G = rng.standard_normal((n, n + 20))
    A = G @ G.T / n + np.eye(n)
    t0 = time.time()
    Lref = np.linalg.cholesky(A) if n <= 8192 else None
    for impl in impls:
        best = 1e9
        for r in range(reps):
            F = np.asfortranarray(A.copy())
            info, ms = ctypes.c_int32(-7), ctypes.c_float()
            ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(F), impl, ctypes.byref(info), ctypes.byref(ms)))
            best = min(best, ms.value)
        err = np.abs(np.tril(F) - Lref).max() / np.abs(Lref).max() if Lref is not None else float("nan")
        up = np.array_equal(np.triu(F, 1), np.triu(A, 1))
        print(f"n={n} impl={impl} info={info.value} ms={best:.3f} TF={n**3/3/best/1e9:.2f} relerr={err:.2e} upper_untouched={up}", flush=True)
