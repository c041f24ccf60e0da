"""One Pascoletti-Serafini step with the solver on the device at the BASELINE dimensions (Morbit's default budgets):
d = 64 on a C3-shaped model (n = 8192), d = 128 on one C4 start (n = 257), d = 256 on a C5-shaped model (n = 2048 here)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import workloads as wl
from morbit.jl_amd import pascoletti_serafini as ps
which = sys.argv[1] if len(sys.argv) > 1 else "64,128,256"
for d in [int(v) for v in which.split(",")]:
    if d == 64:
        C = wl.problem("C3")[0]; Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d; cfg = pkg.RbfConfig(kernel="multiquadric")
    elif d == 128:
        C, Y, _ = wl.problem("C4", 0); cfg = pkg.RbfConfig(kernel="cubic")
    else:
        rng = np.random.default_rng(5); C = rng.random((2048, d)); Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d; cfg = pkg.RbfConfig(kernel="cubic")
    mod = pkg.update_model(cfg, C, Y)
    x = C[0].copy() if d == 128 else np.full(d, 0.5)
    lb, ub = np.maximum(x - 0.1, 0), np.minimum(x + 0.1, 1)
    fx = pkg.eval_models_at_sites(mod, None, x[None, :])[0]
    for rep in range(3):
        st = {}
        t0 = time.perf_counter()
        out = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=rep, stats=st)
        wall = (time.perf_counter() - t0) * 1e3
        ne = st["evals_ideal"] + st["evals_ps"]
        print("d=%d n=%d: %d evaluations in %d generations: %.1f ms (events %.1f ms) -> %.2f M evaluations/s, omega %.4g" % (
            d, C.shape[0], ne, st["generations"], wall, st["ms_total"], ne / wall / 1e3, out[0]), flush=True)
    mod.free()
