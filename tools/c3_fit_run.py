import importlib, numpy as np, os, sys
pkg = importlib.import_module("morbit.jl_amd")
rng = np.random.default_rng(3)
n, d = 8192, 64
C = rng.random((n, d)); Y = np.stack([((C-1)**2).sum(1), ((C+1)**2).sum(1)], 1) / d
cfg = pkg.RbfConfig(kernel="multiquadric")
for i in range(3):
    m = pkg.update_model(cfg, C, Y); print("ms_factor", m.info["ms_factor"], m.info.get("ms_factor_device"), m.info["ms_project"]); m.free()
