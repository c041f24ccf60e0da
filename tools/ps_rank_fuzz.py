"""Random populations through the ranking kernels against the NumPy oracle (one-off stress run; the fixed cases live in tests/).
usage: python tools/ps_rank_fuzz.py [cases]"""
import os, sys
os.environ.setdefault("MRBF_EXPERIMENTS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib
from oracle import ps_rank_oracle as pro
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2026)
ctx = pkg.Context()
bad = 0
for case in range(ncase):
    lam = int(rng.integers(64, 7169))
    kind = case % 4
    f = rng.random(lam)
    if kind == 0:
        phi = np.where(rng.random(lam) < rng.random(), rng.random(lam), 0.0)
    elif kind == 1:
        f = np.round(f, 1); phi = np.round(np.where(rng.random(lam) < 0.5, rng.random(lam), 0.0), 1)
    elif kind == 2:
        f = np.sort(f); phi = np.zeros(lam); phi[rng.integers(0, lam, 5)] = rng.random(5)
    else:
        phi = rng.random(lam); nout = int(rng.integers(0, 20)); f[lam - nout:] = np.inf; phi[lam - nout:] = np.inf
    seed, gen = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 200))
    want, phases = pro.stochastic_rank(f, phi, seed=seed, gen=gen)
    impls = (0, 1, 6) if lam >= 1024 else (0, 9)
    for impl in impls:
        order = np.empty(lam, dtype=np.int32)
        ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(phi), seed, gen, impl, order.ctypes.data_as(_lib.c_ip), None))
        if not np.array_equal(order, want):
            bad += 1
            print("MISMATCH lam %d kind %d impl %d seed %d gen %d first %d" % (lam, kind, impl, seed, gen, int(np.argmax(order != want))), flush=True)
    print("case %2d lam %4d kind %d phases %4d ok" % (case, lam, kind, phases), flush=True)
print("mismatches:", bad)
