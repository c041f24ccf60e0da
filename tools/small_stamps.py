"""Phase times of the one-launch small fit (MRBF_SMALL_STAMPS=1 prints them per call): C4 start, then shapes that separate the
cost of the operands (d) from the cost per output element (n)."""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MRBF_SMALL_STAMPS", "1")
import numpy as np
import morbit.jl_amd as pkg
from morbit.jl_amd import workloads as wl
cfg = pkg.RbfConfig(kernel="cubic")
C, Y, X = wl.problem("C4", 0)
for _ in range(3):
    pkg.update_model(cfg, C, Y).free()
rng = np.random.default_rng(0)
for n, d in ((257, 16), (257, 64), (129, 128), (512, 64), (100, 10)):
    C = rng.random((n, d)); Y = (C**2).sum(1, keepdims=True)
    for _ in range(2):
        pkg.update_model(cfg, C, Y).free()
