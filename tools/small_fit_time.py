"""Wall time of single small fits (n <= 512: the one-launch path) for a few shapes; MRBF_LIB selects the build."""
import os, sys, time
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import morbit.jl_amd as pkg
cfg = pkg.RbfConfig(kernel="cubic")
rng = np.random.default_rng(0)
out = []
for n, d in ((100, 10), (153, 16), (300, 24), (400, 32), (500, 32), (300, 48), (512, 64), (257, 128), (300, 128)):
    C = rng.random((n, d)); Y = (C**2).sum(1, keepdims=True)
    for _ in range(3):
        pkg.update_model(cfg, C, Y).free()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); m = pkg.update_model(cfg, C, Y); ts.append(time.perf_counter() - t0); res = m.info["rel_residual"]; m.free()
    out.append("n=%d d=%d: %.0f us (res %.0e)" % (n, d, 1e6 * float(np.median(ts)), res))
print(" | ".join(out))
