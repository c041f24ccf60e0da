"""Sequential single fits of changing size on ONE context against the host-launched block solve (MRBF_BACKSOLVE_LAUNCHES=1 in a child
process): finds results that depend on what the context solved before."""
import os, sys, subprocess, pickle
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sizes = [640, 900, 640, 1300, 900, 2048, 777, 900, 1800, 900]
def problems():
    out = []
    for p, n in enumerate(sizes):
        rng = np.random.Generator(np.random.PCG64(300 + p))
        C = rng.random((n, 16))
        Y = np.stack([np.sin(C.sum(axis=1)), (C ** 2).sum(axis=1) / 16], axis=1)
        out.append((C, Y))
    return out
def run():
    import morbit.jl_amd as pkg
    cfg = pkg.RbfConfig(kernel="multiquadric", polynomial_degree=1)
    return [pkg.update_model(cfg, C, Y).weights.copy() for C, Y in problems()]
if len(sys.argv) > 1:
    pickle.dump(run(), open(sys.argv[1], "wb"))
    sys.exit(0)
env = dict(os.environ, MRBF_BACKSOLVE_LAUNCHES="1")
subprocess.check_call([sys.executable, __file__, "/tmp/bs_ref.pkl"], env=env)
ref = pickle.load(open("/tmp/bs_ref.pkl", "rb"))
for rep in range(3):
    got = run()
    for n, a, b in zip(sizes, got, ref):
        d = np.abs(a - b).max(axis=0) / np.abs(b).max(axis=0)
        print("rep %d n=%4d rel diff per column %s %s" % (rep, n, d, "BAD" if d.max() > 1e-9 else ""), flush=True)
