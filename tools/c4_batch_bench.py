#!/usr/bin/env python3
"""C4 (BASELINE.json configs[3]: ZDT1 d=128, 64 Halton starts, n=257 cubic deg 1, m=6450 values + Jacobians per start) through ONE
mrbf_batch_run call per step, inputs and outputs resident in HBM (torch tensors handed over as device pointers)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg  # noqa: E402
from morbit.jl_amd import _lib  # noqa: E402
from morbit.jl_amd import workloads as wl  # noqa: E402

lib = pkg.load()
cfg = wl.CONFIGS["C4"]
P, n, d, k, m = cfg["problems"], cfg["n"], cfg["d"], cfg["k"], cfg["m"]
P = int(os.environ.get("P", P))
want_jac = int(os.environ.get("JAC", "1"))
arr = (_lib.Problem * P)()
res = (_lib.Result * P)()
keep = []
ptr = lambda t: _lib.ctypes.cast(t.data_ptr(), _lib.c_dp)
for p in range(P):
    C, Y, X = wl.problem("C4", p)
    dC, dY, dX = torch.from_numpy(C).cuda(), torch.from_numpy(Y).cuda(), torch.from_numpy(X).cuda()
    dV = torch.empty((m, k), dtype=torch.float64, device="cuda")
    dJ = torch.empty((m, d, k), dtype=torch.float64, device="cuda") if want_jac else None
    dW = torch.empty((n, k), dtype=torch.float64, device="cuda")
    keep.append((dC, dY, dX, dV, dJ, dW))
    arr[p] = _lib.Problem(n, m, d, k, 0, 1, 3.0, 0.0, ptr(dC), ptr(dY), ptr(dX), ptr(dW), None, ptr(dV), ptr(dJ) if want_jac else None)
torch.cuda.synchronize()
for it in range(6):
    t0 = time.perf_counter()
    rc = lib.mrbf_batch_run(1, None, P, arr, res)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(res[p].status == 0 for p in range(P))
    print("step %d: %d problems in %.2f ms -> %.0f problems/s (rc=%d ok=%s, worst residual %.1e, fit launch %.3f ms, eval launches %.3f ms)"
          % (it, P, dt * 1e3, P / dt, rc, ok, max(res[p].fit.rel_residual for p in range(P)), res[0].fit.ms_factor, res[0].ms_eval), flush=True)
