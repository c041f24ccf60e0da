"""Experiment: C3 cycles from W concurrent contexts (host threads), optionally with smaller persistent grids: does overlapping one problem's
chain-bound tail with another's bulk phases raise cycles/s?"""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import ctypes, os, sys, threading, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib, workloads as wl

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = wl.CONFIGS["C3"]
n, d, k, m = cfg["n"], cfg["d"], cfg["k"], cfg["m"]
C, Y, X = wl.problem("C3")
dC, dY, dX = torch.from_numpy(C).cuda(), torch.from_numpy(Y).cuda(), torch.from_numpy(X).cuda()
rcfg = pkg.RbfConfig(kernel="multiquadric")
kid, a, b = pkg.rbf_model._get_kernel_params(1.0, rcfg)

class Worker:
    def __init__(self):
        self.ctx = pkg.Context(0)
        self.ctx.set_option(_lib.OPT_RESIDUAL, 0)
        self.dV = torch.empty((m, k), dtype=torch.float64, device="cuda")
        self.dJ = torch.empty((m, d, k), dtype=torch.float64, device="cuda")
        self.fi, self.ei = _lib.FitInfo(), _lib.EvalInfo()
    def cycle(self):
        ctx, lib = self.ctx, self.ctx.lib
        h = _lib.c_vp()
        ctx.check(lib.mrbf_fit(ctx.h, n, d, k, _lib.as_ptr(dC), _lib.as_ptr(dY), kid, a, b, 1, ctypes.byref(h), None, None, ctypes.byref(self.fi)))
        ctx.check(lib.mrbf_eval(ctx.h, h, m, _lib.as_ptr(dX), _lib.as_ptr(self.dV), _lib.as_ptr(self.dJ), ctypes.byref(self.ei)))
        ctx.check(lib.mrbf_free_model(ctx.h, h))

ws = [Worker() for _ in range(W)]
for w in ws:
    w.cycle(); w.cycle()
torch.cuda.synchronize()
def run(w):
    for _ in range(cycles):
        w.cycle()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(w,)) for w in ws]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("contexts %d MRBF_MEGA_GRID=%s: %d cycles in %.3f s -> %.1f cycles/s (factor %.2f ms, device clock %.2f ms, fallbacks %d)" % (
    W, os.environ.get("MRBF_MEGA_GRID", "512"), W * cycles, dt, W * cycles / dt, ws[0].fi.ms_factor, ws[0].fi.ms_factor_device, ws[0].fi.fallbacks))
