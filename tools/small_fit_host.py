"""Where a small fit's wall time goes: host arrays vs device tensors handed to mrbf_fit, and the kernel's own time (MRBF_SMALL_STAMPS)."""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib
ctx = pkg.default_context(); lib = ctx.lib
rng = np.random.default_rng(0)
for n, d in ((100, 10), (300, 24), (500, 32)):
    C = rng.random((n, d)); Y = (C**2).sum(1, keepdims=True); k = 1
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, pkg.RbfConfig(kernel="cubic"))
    W = np.empty((n, k)); L = np.empty((d + 1, k))
    dC = torch.tensor(C, device="cuda"); dY = torch.tensor(Y, device="cuda"); dW = torch.empty((n, k), dtype=torch.float64, device="cuda")
    def run(Cp, Yp, Wp, Lp):
        h = _lib.c_vp(); info = _lib.FitInfo()
        ctx.check(lib.mrbf_fit(ctx.h, n, d, k, Cp, Yp, kid, a, b, 1, ctypes.byref(h), Wp, Lp, ctypes.byref(info)))
        lib.mrbf_free_model(ctx.h, h)
    for name, args in (("host in/out", (_lib.as_ptr(C), _lib.as_ptr(Y), _lib.as_ptr(W), _lib.as_ptr(L))),
                       ("device in/out", (_lib.as_ptr(dC), _lib.as_ptr(dY), _lib.as_ptr(dW), None)),
                       ("device in, no out", (_lib.as_ptr(dC), _lib.as_ptr(dY), None, None))):
        for _ in range(5): run(*args)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter(); run(*args); ts.append(time.perf_counter() - t0)
        print("n=%d d=%d %-18s %.0f us (min %.0f)" % (n, d, name, 1e6 * np.median(ts), 1e6 * min(ts)), flush=True)
