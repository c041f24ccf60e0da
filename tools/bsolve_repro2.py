"""test_batch_run_concurrent_persistent_kernels as a script: batch (four contexts at a time) and single calls against the host-launched
block solve computed in a child process."""
import os, sys, subprocess, pickle
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
n_list = (512, 640, 900, 1300, 1800, 2048, 1100, 777)
def problems():
    out = []
    for p, n in enumerate(n_list * 2):
        rng = np.random.Generator(np.random.PCG64(300 + p))
        C = rng.random((n, 16))
        Y = np.stack([np.sin(C.sum(axis=1)), (C ** 2).sum(axis=1) / 16], axis=1)
        out.append((C, Y, rng.random((50, 16))))
    return out
import morbit.jl_amd as pkg
cfg = pkg.RbfConfig(kernel="multiquadric", polynomial_degree=1)
if len(sys.argv) > 1:
    pickle.dump([pkg.update_model(cfg, C, Y).weights.copy() for C, Y, X in problems()], open(sys.argv[1], "wb"))
    sys.exit(0)
subprocess.check_call([sys.executable, __file__, "/tmp/bs_ref.pkl"], env=dict(os.environ, MRBF_BACKSOLVE_LAUNCHES="1"))
ref = pickle.load(open("/tmp/bs_ref.pkl", "rb"))
import test_gpu_configs as T
kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
for rep in range(3):
    res, Ws, Ls, Vs, _ = T._batch(problems(), kid, a, b, 1)
    for p, (C, Y, X) in enumerate(problems()):
        single = pkg.update_model(cfg, C, Y).weights
        db = np.abs(Ws[p] - ref[p]).max(axis=0) / np.abs(ref[p]).max(axis=0)
        ds = np.abs(single - ref[p]).max(axis=0) / np.abs(ref[p]).max(axis=0)
        flag = "BAD" if max(db.max(), ds.max()) > 1e-9 else ""
        print("rep %d p=%2d n=%4d batch %s single %s residual(batch) %.1e fallbacks %d %s" % (rep, p, C.shape[0], db, ds, res[p].fit.rel_residual, res[p].fit.fallbacks, flag), flush=True)
