#!/usr/bin/env python3
"""Race screen of the two-stream round-4 walk and of the several-workgroup PS ranking: repeated calls on the same inputs must give the same
accepted list, bit-identical weights of the fit from the kept factor, and bit-identical PS steps."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import sampling
from morbit.jl_amd import pascoletti_serafini as ps

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for d, mc in ((64, 10000), (128, 6000), (24, 3000)):
    rng = np.random.default_rng(1)
    x = np.full(d, 0.5)
    start = np.vstack([x[None, :], x[None, :] + 0.3 * np.eye(d)])
    cand = x[None, :] + 0.4 * (2.0 * rng.random((mc, d)) - 1.0)
    cfg = pkg.RbfConfig(kernel="cubic")
    f = lambda S: np.stack([((S - 0.3) ** 2).sum(1), ((S - 0.7) ** 2).sum(1)], axis=1) / d
    seen = {}
    for r in range(reps):
        acc, st = sampling.rbf_round4_device(cfg, start, cand, 1.0, keep_state=True)
        S = st.training_sites
        mod = sampling.fit_from_round4(st, f(S))
        h = hashlib.sha1(np.asarray(acc, dtype=np.int64).tobytes() + mod.weights.tobytes()).hexdigest()[:12]
        seen[h] = seen.get(h, 0) + 1
        mod.free(); st.free()
    print("round 4 d=%d candidates=%d: %d calls, %d accepted, distinct (accepted list, weights) results: %s" % (d, mc, reps, len(acc), seen), flush=True)
    assert len(seen) == 1

d, n = 256, 2048
rng = np.random.default_rng(5)
C = rng.random((n, d)); Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d
mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
x = np.full(d, 0.5); lb, ub = x - 0.1, x + 0.1
fx = pkg.eval_models_at_sites(mod, None, x[None, :])[0]
seen = {}
for r in range(max(4, reps // 3)):
    omega, (xt, mt, _) = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=7)
    h = hashlib.sha1(np.float64(omega).tobytes() + xt.tobytes()).hexdigest()[:12]
    seen[h] = seen.get(h, 0) + 1
print("PS step d=%d (ranking on sixteen workgroups per run): distinct (omega, x_trial) results: %s" % (d, seen), flush=True)
assert len(seen) == 1
mod.free()
