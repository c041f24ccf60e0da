"""Time of the persistent Cholesky alone (mrbf_debug_potrf, impl 3) on a device-resident s.p.d. matrix: no host copies, no LAPACK.
usage: python tools/potrf_time.py 2048,8192 [reps]      (MRBF_MEGA_* environment values select the schedule)
prints per size: min / median ms over the repetitions and the TFLOP/s of n^3 / 3 at the minimum."""
import ctypes
import importlib
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib  # noqa: E402

ctx = pkg.default_context()
sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 8192]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
check = os.environ.get("POTRF_CHECK", "0") != "0"
out = []
for n in sizes:
    g = torch.Generator(device="cuda").manual_seed(n)
    G = torch.randn((n, n + 32), dtype=torch.float64, device="cuda", generator=g)
    A = G @ G.T / n + torch.eye(n, dtype=torch.float64, device="cuda")
    del G
    F = torch.empty_like(A)
    ts = []
    info = ctypes.c_int32(-7)
    for r in range(reps + 1):
        F.copy_(A)
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(F), 3, ctypes.byref(info), ctypes.byref(ms)))
        if r:
            ts.append(ms.value)
    err = float("nan")
    if check:
        L = torch.tril(F.T)  # column-major L of the library = transposed row-major tensor
        R = L @ L.T - A
        err = (R.abs().max() / A.abs().max()).item()
    out.append("n=%d min %.4f med %.4f ms (%.1f TF) info %d%s" % (n, min(ts), float(np.median(ts)), n ** 3 / 3 / min(ts) / 1e9, info.value,
                                                                  (" resid %.1e" % err) if check else ""))
    del A, F
print(" | ".join(out), flush=True)
