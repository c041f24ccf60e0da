"""mrbf_affine_select alone: d = 128, 300 / 1000 candidates, wall time per call (and per pick).  usage: python tools/affine_bench.py [d] [mc]"""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")
import sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import sampling as sp
d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(1)
x = rng.random(d)
seeds = list(x + 0.2 * (2 * rng.random((mc, d)) - 1))
ctx = pkg.default_context()
for rep in range(5):
    f = sp.AffinelyIndependentPointFilter(x, seeds, pivot_val=0.02, ctx=ctx)
    t0 = time.perf_counter()
    got = f.collect()
    dt = (time.perf_counter() - t0) * 1e3
    print("d=%d mc=%d: %d picks in %.2f ms (%.1f us per pick)" % (d, mc, len(got), dt, dt * 1e3 / max(len(got), 1)), flush=True)
