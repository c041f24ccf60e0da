"""values-only evaluation of a PS-sized population on the C3-shaped model (d = 64, n = 8192, multiquadric, k = 2): device time of
mrbf_eval (events around centring + sweep + combine) against the split of the centre range.  usage: MRBF_EXPERIMENTS=1 python tools/eval_vals_sweep.py"""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import morbit.jl_amd as pkg
    from morbit.jl_amd import workloads as wl
    C = wl.problem("C3")[0]; d = C.shape[1]
    Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d
    mod = pkg.update_model(pkg.RbfConfig(kernel="multiquadric"), C, Y)
    ctx = mod.ctx
    ctx.set_option(pkg._lib.OPT_TIMING, 1) if hasattr(ctx, "set_option") else None
    out = {}
    for m in (1320, 2640, 10000):
        X = torch.from_numpy(np.random.default_rng(m).random((m, d))).cuda()
        V = torch.empty((m, 2), dtype=torch.float64, device="cuda")
        for _ in range(3): mod.eval_sites(X, out_vals=V)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        for _ in range(50): mod.eval_sites(X, out_vals=V)
        torch.cuda.synchronize()
        out[m] = (time.perf_counter() - t0) / 50 * 1e6
    print(json.dumps(out))
    sys.exit(0)
for ns in ("0", "6", "8", "11", "12", "16", "22", "24", "32"):
    env = dict(os.environ, MRBF_EXPERIMENTS="1")
    if ns != "0": env["MRBF_EVAL_NSPLIT"] = ns
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print("nsplit %s: wall us per values-only call (device-resident in/out, incl. the synchronisation) %s" % ("auto" if ns == "0" else ns, line[-1] if line else r.stderr[-300:]), flush=True)
