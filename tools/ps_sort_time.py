"""wall time of one ranking through mrbf_debug_ps_rank: plain sort in registers (impl 0) against the one-pair-per-thread form (impl 3)"""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")
import sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib
ctx = pkg.Context()
for lam in (1320, 2600, 5160):
    f = np.random.default_rng(lam).random(lam); phi = np.zeros(lam); order = np.empty(lam, dtype=np.int32)
    for impl in (0, 3, 4, 5):
        call = lambda: ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(phi), 5, 1, impl, order.ctypes.data_as(_lib.c_ip), None))
        for _ in range(5): call()
        t0 = time.perf_counter()
        for _ in range(200): call()
        print("lam %d impl %d: %.1f us per call (incl. ~constant transfers)" % (lam, impl, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
