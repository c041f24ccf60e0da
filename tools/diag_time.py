import sys, ctypes, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import morbit, importlib
pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib
ctx = pkg.default_context()
rng = np.random.default_rng(0)
G = rng.standard_normal((128, 160)); A = np.asfortranarray(G @ G.T / 128 + np.eye(128))
for impl in (0, 1):
    ctx.set_option(_lib.OPT_DIAG_IMPL, impl)
    ms = ctypes.c_float(); cyc = ctypes.c_double(); us = ctypes.c_double()
    ctx.check(ctx.lib.mrbf_debug_diag(ctx.h, _lib.as_ptr(A), 200, ctypes.byref(ms), ctypes.byref(cyc), ctypes.byref(us)))
    print("diag impl", impl, "us per call", ms.value * 1e3)
