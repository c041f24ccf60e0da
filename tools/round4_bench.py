#!/usr/bin/env python3
"""Round 4 of the site selection + the fit from its kept factor at a realistic size: d = 64, 10^4 database sites in the box, a unisolvent
start set of d + 1 sites, max_points = (d + 1)(d + 2) / 2 = 2145 (RbfModel.jl:356).  Wall times of mrbf_round4 / mrbf_fit_from_round4 and
of the ordinary fit on the same training set."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import sampling

d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(1)
x = np.full(d, 0.5)
start = np.vstack([x[None, :], x[None, :] + 0.3 * np.eye(d)])            # centre + d affinely independent sites
cand = x[None, :] + 0.4 * (2.0 * rng.random((mc, d)) - 1.0)
cfg = pkg.RbfConfig(kernel="cubic")
f = lambda S: np.stack([((S - 0.3) ** 2).sum(1), ((S - 0.7) ** 2).sum(1)], axis=1) / d
for rep in range(3):
    t0 = time.perf_counter()
    acc, st = sampling.rbf_round4_device(cfg, start, cand, 1.0, keep_state=True)
    t1 = time.perf_counter()
    S = st.training_sites
    try:
        mod = sampling.fit_from_round4(st, f(S))
    except pkg._lib.MrbfError as e:   # the kept factor refused (residual tripwire): the bindings then take the ordinary fit (mrbf_dispatch_after)
        print("   (fit from the kept factor refused: %s)" % str(e)[:110], flush=True)
        mod = pkg.update_model(cfg, S, f(S))
    t2 = time.perf_counter()
    mod2 = pkg.update_model(cfg, S, f(S))
    t3 = time.perf_counter()
    dw = np.abs(mod.weights - mod2.weights).max() / np.abs(mod2.weights).max()
    print("d=%d candidates=%d: round 4 %.1f ms -> %d accepted (n = %d); fit from the kept factor %.2f ms (residual %.1e), ordinary fit %.2f ms; "
          "weights differ by %.1e" % (d, mc, (t1 - t0) * 1e3, len(acc), S.shape[0], (t2 - t1) * 1e3, mod.info["rel_residual"], (t3 - t2) * 1e3, dw), flush=True)
    mod.free(); mod2.free(); st.free()
