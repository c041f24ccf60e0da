"""Race screen at the level of the fit (the persistent factorisation with the right-hand sides riding along, the front end on its side
stream, backward substitution, tail): the same problem fitted again and again must give the same bits.
usage: python tools/fit_determinism.py 2048,8192 200"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("morbit.jl_amd")
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 8192]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for n in sizes:
    rng = np.random.default_rng(n)
    d = 64 if n >= 4096 else 32
    C = rng.random((n, d)); Y = np.stack([((C - 1) ** 2).sum(1), np.sin(C.sum(1))], 1) / d
    cfg = pkg.RbfConfig(kernel="multiquadric")
    ref = None; bad = 0; t0 = time.time()
    for r in range(reps):
        m = pkg.update_model(cfg, C, Y)
        w = np.concatenate([m.weights.ravel(), m.poly.ravel()]).copy(); res = m.info["rel_residual"]; m.free()
        if ref is None: ref = w
        elif not np.array_equal(ref, w): bad += 1
    print("n=%d d=%d: %d fits, %d differing, residual %.1e (%.1f s)" % (n, d, reps, bad, res, time.time() - t0), flush=True)
