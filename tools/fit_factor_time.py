"""ms_factor / ms_factor_device of full fits (the factorisation with the right-hand sides riding along): python tools/fit_factor_time.py 2048,8192"""
import importlib, numpy as np, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("morbit.jl_amd")
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 8192]
out = []
for n in sizes:
    rng = np.random.default_rng(3)
    d = 64
    C = rng.random((n, d)); Y = np.stack([((C-1)**2).sum(1), ((C+1)**2).sum(1)], 1) / d
    cfg = pkg.RbfConfig(kernel="multiquadric")
    ts = []
    for i in range(9):
        m = pkg.update_model(cfg, C, Y); ts.append((m.info["ms_factor"], m.info.get("ms_factor_device", 0.0))); res = m.info["rel_residual"]; m.free()
    ts = ts[2:]
    out.append("n=%d factor min %.4f med %.4f (device min %.4f) res %.1e" % (n, min(t[0] for t in ts), float(np.median([t[0] for t in ts])), min(t[1] for t in ts), res))
print(" | ".join(out))
