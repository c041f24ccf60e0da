"""wall time of one STOCHASTIC ranking (a generation with infeasible individuals: lam transposition phases) through mrbf_debug_ps_rank:
one workgroup (impl 0), a wave per 64 individuals (impl 1), the same without waiting for the neighbours (7: what the phases alone
cost), sixteen workgroups (impl 6).  Populations below 1024: impl 0 = a wave per 96 individuals inside the one workgroup, impl 9 the
one-pair-per-thread loop through LDS it replaces.  The transfers (3 uploads, 2 downloads, one
synchronisation) are the same for every impl; the first line per size (impl 5 on a feasible generation: a plain sort) shows their share."""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")
import sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib
ctx = pkg.Context()
for lam in (280, 520, 1000, 1320, 2600, 5160):
    rng = np.random.default_rng(lam)
    f = rng.random(lam); phi = np.where(rng.random(lam) < 0.5, 0.0, rng.random(lam)); order = np.empty(lam, dtype=np.int32)
    ref = None
    for impl in ((5, 0, 9) if lam < 1024 else (5, 0, 1, 7, 6)):
        ph = np.zeros(lam) if impl == 5 else phi
        call = lambda: ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(ph), 5, 1, impl, order.ctypes.data_as(_lib.c_ip), None))
        for _ in range(3): call()
        n = 20 if impl == 0 else 100
        t0 = time.perf_counter()
        for _ in range(n): call()
        dt = (time.perf_counter() - t0) / n * 1e6
        if impl not in (5, 7):
            if ref is None: ref = order.copy()
            assert np.array_equal(order, ref), (lam, impl)
        print("lam %d impl %d: %.1f us per call%s" % (lam, impl, dt, " (plain sort: the transfers' share)" if impl == 5 else ""), flush=True)
