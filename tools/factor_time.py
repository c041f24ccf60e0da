"""Factor-phase time of mrbf_fit over problem sizes (multiquadric + linear tail, halton-like random sites): median of a few fits.
usage: python tools/factor_time.py 1024 2048 4096 8192 16384"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
pkg = importlib.import_module("morbit.jl_amd")
sizes = [int(x) for x in sys.argv[1:]] or [2048, 4096, 8192]
for n in sizes:
    d = 32
    rng = np.random.default_rng(n)
    C = rng.random((n, d))
    Y = np.sin(C.sum(axis=1, keepdims=True))
    cfg = pkg.RbfConfig(kernel="multiquadric")
    ts, res = [], None
    for rep in range(3 if n > 8192 else 7):
        mod = pkg.update_model(cfg, C, Y)
        ts.append(mod.fit_info["ms_factor"] if hasattr(mod, "fit_info") else mod.info["ms_factor"])
        res = (mod.fit_info if hasattr(mod, "fit_info") else mod.info)["rel_residual"]
        mod.free()
    print(f"n {n:6d}  factor median {np.median(ts[1:]):8.4f} ms  min {min(ts[1:]):8.4f}  rel_residual {res:.1e}")
