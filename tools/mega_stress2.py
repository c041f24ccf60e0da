"""Per-launch times of the persistent factorisation over mixed sizes (the order of gpurun_out/stress/log2.txt), several rounds:
prints every launch's hipEvent time and host wall time so that an outlier can be placed (first launch of a shape = job-table rebuild +
workspace growth inside the timed bracket, or a later one)."""
import ctypes, sys, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, ".")
import morbit, importlib
pkg = importlib.import_module("morbit.jl_amd")
from morbit.jl_amd import _lib
ctx = pkg.default_context()
rng = np.random.default_rng(123)
sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [257, 384, 385, 700, 1280, 1920, 2560, 3200, 1920, 257, 1920, 1100, 1920]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
mats = {}
for n in set(sizes):
    G = rng.standard_normal((n, n + 8))
    mats[n] = G @ G.T / n + np.eye(n)
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 8):
    for n in sizes:
        A = mats[n]
        dA = torch.from_numpy(np.asfortranarray(A)).cuda()
        ms_all, wall_all, dev_all = [], [], []
        for r in range(reps):
            dF = dA.clone()
            torch.cuda.synchronize()
            info, ms = ctypes.c_int32(-7), ctypes.c_float()
            t0 = time.perf_counter()
            ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, ctypes.c_void_p(dF.data_ptr()), 3, ctypes.byref(info), ctypes.byref(ms)))
            wall_all.append((time.perf_counter() - t0) * 1e3)
            ms_all.append(ms.value)
            dev_all.append(ctx.get_option(_lib.OPT_LAST_DEVICE_MS))
            assert info.value == 0
        med = sorted(ms_all)[len(ms_all) // 2]
        out = [i for i, v in enumerate(ms_all) if v > 1.5 * med]
        line = "round %d n=%5d median %.3f ms (device clock %.3f) slow_launches %d" % (rnd, n, med, sorted(dev_all)[len(dev_all) // 2], int(ctx.get_option(_lib.OPT_SLOW_LAUNCHES)))
        for i in out:
            line += " | OUTLIER rep %d: events %.3f ms, device clock %.3f ms, host wall %.2f ms" % (i, ms_all[i], dev_all[i], wall_all[i])
        print(line, flush=True)
