import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import morbit.jl_amd as pkg
from morbit.jl_amd import workloads as wl
C, Y, X = wl.problem("C4", 0)
cfg = pkg.RbfConfig(kernel="cubic")
for _ in range(3):
    mod = pkg.update_model(cfg, C, Y)
    mod.free()
rng = np.random.default_rng(0)
for n, d in ((100, 10), (512, 64), (60, 5)):
    C = rng.random((n, d)); Y = (C**2).sum(1, keepdims=True)
    for _ in range(2):
        mod = pkg.update_model(cfg, C, Y); mod.free()
