"""Wall time of mrbf_eval at one site (what eval_models / get_jacobian of the plug-in do per call) and at a small batch, host arrays."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import morbit.jl_amd as pkg
from morbit.jl_amd import _lib
ctx = pkg.default_context(); lib = ctx.lib
rng = np.random.default_rng(0)
for n, d, k in ((100, 10, 2), (300, 24, 2), (2048, 32, 1)):
    C = rng.random((n, d)); Y = np.stack([np.sin(C.sum(1) * (l + 1)) for l in range(k)], 1)
    m = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
    for mq in (1, 64):
        X = rng.random((mq, d)); V = np.empty((mq, k)); J = np.empty((mq, d, k))
        for want in ("values", "values+jac"):
            jp = _lib.as_ptr(J) if want != "values" else None
            for _ in range(20): ctx.check(lib.mrbf_eval(ctx.h, m.model, mq, _lib.as_ptr(X), _lib.as_ptr(V), jp, None))
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); ctx.check(lib.mrbf_eval(ctx.h, m.model, mq, _lib.as_ptr(X), _lib.as_ptr(V), jp, None)); ts.append(time.perf_counter() - t0)
            print("n=%d d=%d k=%d m=%d %-11s %.1f us (min %.1f)" % (n, d, k, mq, want, 1e6 * np.median(ts), 1e6 * min(ts)), flush=True)
    m.free()
