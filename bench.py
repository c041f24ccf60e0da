#!/usr/bin/env python3
"""bench.py -- RBF build+solve+eval cycles/sec (BASELINE.json metric) on N MI355X GPUs of one node.

One STEP = one cycle of the hot path on one batch of synthetic input already resident in HBM:
    mrbf_fit  (Gram assembly -> projection -> Cholesky -> solve for k right-hand sides)
  + mrbf_eval (values + Jacobians at m query points)
Default workload = BASELINE.json configs[2] ("C3": d=64, n=8192 centres, multiquadric, degree-1 tail, k=2,
m=10000 evals, 1 GPU) -- the configuration the north-star targets are quoted on.  With N > 1 every rank
runs its own independent problem of that size (the path shards by problem, no data-path collective);
`value` = cycles of all ranks / wall time of the slowest rank.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {  # BASELINE.json configs -> sizes
    "C2": dict(kernel="gaussian", n=2048, d=32, k=1, m=0, deg=1, seed=2,
               desc="C2: d=32 n=2048 gaussian deg1 k=1, single build+solve"),
    "C3": dict(kernel="multiquadric", n=8192, d=64, k=2, m=10000, deg=1, seed=3,
               desc="C3: d=64 n=8192 multiquadric deg1 k=2, build+solve + 10000 evals (values+Jacobians)"),
    "C4": dict(kernel="cubic", n=257, d=128, k=2, m=6450, deg=1, seed=40,
               desc="C4: ZDT1-shaped d=128 n=257 cubic deg1 k=2, build+solve + 6450 evals per start"),
    "C5": dict(kernel="cubic", n=16384, d=256, k=2, m=1024, deg=1, seed=1000,
               desc="C5: d=256 n=16384 cubic deg1 k=2, build+solve + 1024 evals"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TF = 78.6   # MI355X FP64 matrix peak (spec; v_mfma_f64_16x16x4_f64 = 16 FMA/clk/SIMD x 1024 SIMD x 2.4 GHz)


def synth(cfg, rank):
    rng = np.random.Generator(np.random.PCG64(cfg["seed"] + 7919 * rank))
    n, d, k, m = cfg["n"], cfg["d"], cfg["k"], cfg["m"]
    C = rng.random((n, d))
    if cfg["kernel"] == "cubic" and d == 128:  # ZDT1 (formulas from MultiObjectiveProblems.jl)
        f1 = C[:, 0]
        g = 1.0 + 9.0 * C[:, 1:].sum(axis=1) / (d - 1)
        Y = np.stack([f1, g * (1.0 - np.sqrt(f1 / g))], axis=1)
    else:
        cols = [((C - 1.0) ** 2).sum(axis=1) / d, ((C + 1.0) ** 2).sum(axis=1) / d]
        Y = np.stack(cols[:k], axis=1)
    X = np.random.Generator(np.random.PCG64(cfg["seed"] + 1 + 7919 * rank)).random((max(m, 1), d))
    return C, Y, X


def algorithmic(cfg):
    """SURVEY.md section 8d per-unit figures for one cycle"""
    n, d, k, m = cfg["n"], cfg["d"], cfg["k"], cfg["m"]
    q = 0 if cfg["deg"] < 0 else (1 if cfg["deg"] == 0 else d + 1)
    return dict(
        gram_bytes=8.0 * n * d + 8.0 * n * n,                        # read centres once + write full Phi
        gram_flops=float(n) * n * d,                                 # GEMM form on the lower triangle (2 * n^2/2 * d)
        factor_flops=n ** 3 / 3.0,                                   # potrf
        project_flops=4.0 * n * n * q,                               # symm + syr2k (+ syrk 1 n^2 q not counted)
        solve_flops=2.0 * n * n * k,
        eval_flops=float(m) * n * (3 * d + 2 * k + 2 * k * d),
        eval_bytes=8.0 * (n * d + n * k + m * d + m * k + m * k * d),
    )


def cpu_baseline(cfg, C, Y, X):
    """The oracle ("port"), vectorised NumPy + LAPACK on the host cores, on a bounded sample of the same workload:
    the full fit (assembly + dense LU of the saddle system, as the reference's `\\`) and min(m, 1024) evaluations,
    eval time scaled to m."""
    from oracle import rbf_oracle as orc

    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    kid = orc.KERNEL_IDS[cfg["kernel"]]
    a, b = orc.kernel_params(cfg["kernel"])
    t0 = time.perf_counter()
    mod = orc.fit(C, Y, kid, a, b, cfg["deg"])
    t_fit = time.perf_counter() - t0
    ms = min(cfg["m"], 1024)
    t_eval = 0.0
    if ms > 0:
        t0 = time.perf_counter()
        mod.values(X[:ms])
        mod.jacs(X[:ms])
        t_eval = (time.perf_counter() - t0) * (cfg["m"] / ms)
    cyc = t_fit + t_eval
    return dict(value=1.0 / cyc, unit="cycles/s", cores=int(cores), kind="port",
                sample="full fit n=%d (NumPy assembly + LAPACK dgesv) %.2fs + %d of %d evals (values+Jacobians) scaled %.2fs"
                       % (cfg["n"], t_fit, ms, cfg["m"], t_eval)), mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gram-mode", type=int, default=0)
    ap.add_argument("--chol-impl", type=int, default=0)
    ap.add_argument("--eval-impl", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    import morbit.jl_amd as pkg
    from morbit.jl_amd import _lib

    cfg = CONFIGS[args.config]
    n, d, k, m = cfg["n"], cfg["d"], cfg["k"], cfg["m"]
    C, Y, X = synth(cfg, rank)
    rcfg = pkg.RbfConfig(kernel=cfg["kernel"], polynomial_degree=cfg["deg"])
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, rcfg)

    # inputs + outputs resident in HBM before the timed region
    dC = torch.from_numpy(C).cuda()
    dY = torch.from_numpy(Y).cuda()
    dX = torch.from_numpy(X).cuda()
    dV = torch.empty((max(m, 1), k), dtype=torch.float64, device="cuda")
    dJ = torch.empty((max(m, 1), d, k), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()

    ctx = pkg.Context(local_rank)
    ctx.set_option(_lib.OPT_GRAM_MODE, args.gram_mode)
    ctx.set_option(_lib.OPT_CHOL_IMPL, args.chol_impl)
    ctx.set_option(_lib.OPT_EVAL_IMPL, args.eval_impl)
    lib = ctx.lib
    finfo, einfo = _lib.FitInfo(), _lib.EvalInfo()

    def cycle():
        h = _lib.c_vp()
        ctx.check(lib.mrbf_fit(ctx.h, n, d, k, _lib.as_ptr(dC), _lib.as_ptr(dY), kid, a, b, cfg["deg"], ctypes.byref(h),
                               None, None, ctypes.byref(finfo)))
        if m > 0:
            ctx.check(lib.mrbf_eval(ctx.h, h, m, _lib.as_ptr(dX), _lib.as_ptr(dV), _lib.as_ptr(dJ), ctypes.byref(einfo)))
        ctx.check(lib.mrbf_free_model(ctx.h, h))

    # one checked cycle (residual through the eval kernels), then the timed loop without the extra check
    ctx.set_option(_lib.OPT_RESIDUAL, 1)
    cycle()
    check = dict(path=finfo.path, rel_residual=finfo.rel_residual, max_pitw=finfo.max_pitw)
    assert finfo.rel_residual < 1e-9, "fit residual too large: %r" % (check,)
    ctx.set_option(_lib.OPT_RESIDUAL, 0)
    for _ in range(args.warmup):
        cycle()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    phases = {p: 0.0 for p in ("gram", "project", "factor", "solve", "eval")}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle()
        phases["gram"] += finfo.ms_gram
        phases["project"] += finfo.ms_project
        phases["factor"] += finfo.ms_factor
        phases["solve"] += finfo.ms_solve
        phases["eval"] += einfo.ms_total if m > 0 else 0.0
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for p in phases:
        phases[p] /= args.steps  # ms per launch, hipEvents on the library's stream inside the timed region

    if rank == 0:
        alg = algorithmic(cfg)
        kernels = {
            "gram": dict(bound="hbm", achieved=alg["gram_bytes"] / (phases["gram"] * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                         unit="GB/s", ms=phases["gram"]),
            "factor": dict(bound="mfma", achieved=alg["factor_flops"] / (phases["factor"] * 1e-3) / 1e12,
                           peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s", ms=phases["factor"]),
            "project": dict(bound="mfma", achieved=alg["project_flops"] / max(phases["project"], 1e-9) / 1e9,
                            peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s", ms=phases["project"]),
        }
        if m > 0:
            kernels["eval"] = dict(bound="mfma", achieved=alg["eval_flops"] / (phases["eval"] * 1e-3) / 1e12,
                                   peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s", ms=phases["eval"])
        for kd in kernels.values():
            kd["frac"] = kd["achieved"] / kd["peak"]
        dom = max(("gram", "factor", "eval") if m > 0 else ("gram", "factor"), key=lambda p: phases[p])
        roof = dict(kernels[dom])
        roof["kernel"] = dom
        roof["traffic"] = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                roof["traffic"] = json.load(open(pmc)).get(args.config, {}).get(dom)
            except Exception:
                pass
        out = {
            "metric": "RBF build+solve+eval cycles/sec", "value": world * args.steps / elapsed, "unit": "cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["desc"], "n": n, "d": d, "k": k, "m": m, "kernel": cfg["kernel"],
                       "polynomial_degree": cfg["deg"], "problems_per_step": world, "parallelism": "problem-sharded x%d" % world},
            "roofline": roof, "kernels": kernels, "phases_ms": phases, "check": check,
        }
        if not args.no_cpu_baseline and world == 1:
            cb, _ = cpu_baseline(cfg, C, Y, X)
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
