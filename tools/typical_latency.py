#!/usr/bin/env python3
"""Wall time of the calls one Morbit iteration makes at the sizes Morbit's own examples run (d = 2 .. 30, n up to (d + 1)(d + 2) / 2):
fit, value / Jacobian at one point, a batch of 30 points, the backtracking call."""
import os, sys, time
os.environ.setdefault("MRBF_EXPERIMENTS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import rbf_model as rm
from oracle import c_oracle            # the CPU side of the table: the reference's call pattern, as bench.py's
from oracle import rbf_oracle as orc   # cpu_baseline_faithful does it -- single-threaded per-pair assembly + LAPACK LU, one-point closures

def tm(f, reps=30):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6

for d, n in ((2, 6), (5, 21), (10, 66), (20, 231), (30, 496), (30, 1500)):
    rng = np.random.default_rng(d)
    C = rng.random((n, d)); Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d
    cfg = pkg.RbfConfig(kernel="cubic")
    mods = []
    t_fit = tm(lambda: mods.append(pkg.update_model(cfg, C, Y)), 10)
    mod = mods[-1]
    for m_ in mods[:-1]: m_.free()
    x = rng.random(d); X30 = rng.random((30, d))
    t_v1 = tm(lambda: rm.eval_models(mod, None, x))
    t_j1 = tm(lambda: rm.get_jacobian(mod, None, x))
    t_v30 = tm(lambda: rm.eval_models_at_sites(mod, None, X30))
    t_j30 = tm(lambda: rm.get_jacobians_at_sites(mod, None, X30))
    # CPU reference pattern at the same sizes (us): fit = faithful assembly + dense LU of the saddle system (RbfModel.jl:759-763); a value /
    # a gradient sweep at one point = one closure call per output (AbstractSurrogateInterface.jl:98-106), k = 2 outputs here
    kid, a_, b_ = rm._get_kernel_params(1.0, cfg)
    def cpu_fit():
        c_oracle.gram_cols(C, kid, a_, b_, n)
        return orc.fit(C, Y, kid, a_, b_, 1)
    c_fit = tm(cpu_fit, 5)
    ref = cpu_fit()
    x1 = x[None, :]
    c_v1 = tm(lambda: c_oracle.eval_loop(C, ref.w, ref.lam, kid, a_, b_, 1, x1, want_jac=False), 20) * 2      # one sweep per output
    c_j1 = tm(lambda: c_oracle.eval_loop(C, ref.w, ref.lam, kid, a_, b_, 1, x1, want_jac=True), 20) * 2
    c_v30 = tm(lambda: c_oracle.eval_loop(C, ref.w, ref.lam, kid, a_, b_, 1, X30, want_jac=False), 5) * 2
    c_j30 = tm(lambda: c_oracle.eval_loop(C, ref.w, ref.lam, kid, a_, b_, 1, X30, want_jac=True), 5) * 2
    print("d=%2d n=%4d: fit %7.1f us (cpu %8.1f) | value at a point %5.1f (%6.1f) | Jacobian at a point %5.1f (%6.1f) | 30 values %5.1f (%7.1f) | "
          "30 Jacobians %5.1f (%7.1f)" % (d, n, t_fit, c_fit, t_v1, c_v1, t_j1, c_j1, t_v30, c_v30, t_j30, c_j30), flush=True)
    mod.free()
