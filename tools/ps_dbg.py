"""timing experiments for the device PS step (MRBF_PS_DBG bits: 1 no ranking, 2 no breeding)"""
import os, sys, time
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import pascoletti_serafini as ps
rng = np.random.default_rng(0)
d, n = 12, 512
C = rng.random((n, d))
Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1) + 0.1 * np.sin(5 * C[:, 0])], axis=1)
mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
x = np.full(d, 0.5); x[1] = 0.85; x[3] = 0.2
lb, ub = x - 0.1, x + 0.1
fx = pkg.eval_models_at_sites(mod, None, x[None, :])[0]
for rep in range(3):
    st = {}
    ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=rep, stats=st)
print("dbg", os.environ.get("MRBF_PS_DBG", "0"), "ms", st["ms_total"], "generations", st["generations"])
