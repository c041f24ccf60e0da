import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib
ctx = pkg.default_context(); lib = ctx.lib
rng = np.random.default_rng(0)
n, d, k = 300, 24, 1
C = rng.random((n, d)); Y = (C**2).sum(1, keepdims=True)
kid, a, b = pkg.rbf_model._get_kernel_params(1.0, pkg.RbfConfig(kernel="cubic"))
dC = torch.tensor(C, device="cuda"); dY = torch.tensor(Y, device="cuda"); dW = torch.empty((n, k), dtype=torch.float64, device="cuda")
for _ in range(12):
    h = _lib.c_vp(); info = _lib.FitInfo()
    ctx.check(lib.mrbf_fit(ctx.h, n, d, k, _lib.as_ptr(dC), _lib.as_ptr(dY), kid, a, b, 1, ctypes.byref(h), _lib.as_ptr(dW), None, ctypes.byref(info)))
    lib.mrbf_free_model(ctx.h, h)
