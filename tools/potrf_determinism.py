"""Race screen for the persistent factorisation: the same device-resident matrix factored again and again must give the same bits
(its summation orders are fixed by the job tables), and the factor must reproduce the matrix.  A hand-off or LDS-DMA ordering bug
shows up here as a rare differing tile long before it fails a tolerance test.
usage: python tools/potrf_determinism.py 1024,2048,3200,8192 300"""
import ctypes
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib  # noqa: E402

ctx = pkg.default_context()
sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 2048, 8192]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad_total = 0
for n in sizes:
    g = torch.Generator(device="cuda").manual_seed(n)
    G = torch.randn((n, n + 32), dtype=torch.float64, device="cuda", generator=g)
    A = G @ G.T / n + torch.eye(n, dtype=torch.float64, device="cuda")
    del G
    F = torch.empty_like(A)
    ref = None
    info = ctypes.c_int32(-7)
    bad = 0
    t0 = time.time()
    for r in range(reps):
        F.copy_(A)
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(F), 3, ctypes.byref(info), ctypes.byref(ms)))
        if info.value != 0:
            bad += 1
            print("  n=%d rep %d: info %d" % (n, r, info.value), flush=True)
            continue
        if ref is None:
            ref = F.clone()
            L = torch.tril(ref.T)
            resid = ((L @ L.T - A).abs().max() / A.abs().max()).item()
        elif not torch.equal(F, ref):
            bad += 1
            d = (F - ref).abs()
            idx = torch.nonzero(d > 0)
            print("  n=%d rep %d: %d entries differ (max %.2e), first at %s" % (n, r, idx.shape[0], d.max().item(), idx[0].tolist()), flush=True)
    print("n=%d: %d launches, %d differing, residual of the first %.1e (%.1f s)" % (n, reps, bad, resid, time.time() - t0), flush=True)
    bad_total += bad
    del A, F, ref
sys.exit(1 if bad_total else 0)
