"""Kernel-only time of mrbf_eval (events inside the library) for the C3 model: python tools/eval_time.py  (MRBF_LIB selects the build)"""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib, workloads as wl
ctx = pkg.default_context(); lib = ctx.lib
for cfgname in ("C3", "C2"):
    cfg = wl.CONFIGS[cfgname]
    C, Y, X = wl.problem(cfgname, 0)
    if X is None or len(X) == 0:
        X = np.random.default_rng(1).random((10000, C.shape[1]))
    m = pkg.update_model(pkg.RbfConfig(kernel=cfg["kernel"]), C, Y)
    dX = torch.tensor(X, device="cuda"); mq = dX.shape[0]
    dV = torch.empty((mq, m.k), dtype=torch.float64, device="cuda"); dJ = torch.empty((mq, m.d, m.k), dtype=torch.float64, device="cuda")
    ts = []
    for _ in range(30):
        ei = _lib.EvalInfo()
        ctx.check(lib.mrbf_eval(ctx.h, m.model, mq, _lib.as_ptr(dX), _lib.as_ptr(dV), _lib.as_ptr(dJ), ctypes.byref(ei)))
        ts.append(ei.asdict()["ms_total"])
    print(cfgname, "n=%d d=%d m=%d: eval min %.4f med %.4f ms" % (C.shape[0], C.shape[1], mq, min(ts), float(np.median(ts))), {k: v for k, v in ei.asdict().items() if "ms" in k})
    m.free()
