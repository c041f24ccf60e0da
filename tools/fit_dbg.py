import sys, numpy as np
sys.path.insert(0, ".")
import morbit, importlib
pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib
ctx = pkg.default_context()
for n in (2300, 1300, 2300):
    rng = np.random.default_rng(77)
    C = rng.random((n, 20)); Y = rng.standard_normal((n, 2))
    for impl in (3,):
        ctx.set_option(_lib.OPT_CHOL_IMPL, impl)
        try:
            m = pkg.update_model(pkg.RbfConfig(kernel="multiquadric"), C, Y, ctx=ctx)
            print(n, impl, m.info["path"], m.info["rel_residual"], flush=True)
        except Exception as e:
            print(n, impl, "ERR", e, flush=True)
