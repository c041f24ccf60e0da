"""Stress the persistent factorisation: many launches over mixed sizes, every result checked against LAPACK (small) or by
reconstruction (large); reports the slowest launch per size (a bounded-spin give-up would show as a ~1 s launch and info < 0)."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, ".")
import morbit, importlib
pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib
ctx = pkg.default_context()
rng = np.random.default_rng(123)
sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [640, 1100, 2048, 3000, 4096, 8192]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for n in sizes:
    G = rng.standard_normal((n, n + 8))
    A = G @ G.T / n + np.eye(n)
    Lref = np.linalg.cholesky(A)
    worst, tot = 0.0, 0.0
    for r in range(reps):
        F = np.asfortranarray(A.copy())
        info, ms = ctypes.c_int32(-7), ctypes.c_float()
        ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(F), 3, ctypes.byref(info), ctypes.byref(ms)))
        err = np.abs(np.tril(F) - Lref).max() / np.abs(Lref).max()
        if info.value != 0 or not (err < 1e-12):
            bad += 1
            print("BAD", n, r, info.value, err, flush=True)
        worst = max(worst, ms.value); tot += ms.value
    print(f"n={n} reps={reps} mean {tot/reps:.3f} ms worst {worst:.3f} ms", flush=True)
print("bad launches:", bad)
