#!/usr/bin/env python3
"""bench.py -- RBF build+solve+eval cycles/sec (BASELINE.json metric) on N MI355X GPUs of one node.

One CYCLE = the hot path on one problem whose inputs are already resident in HBM:
    mrbf_fit  (Gram assembly -> projection -> Cholesky -> solve for k right-hand sides)
  + mrbf_eval (values + Jacobians at m query points)
Workloads (morbit.jl_amd/workloads.py):
  --config C3 (default)  BASELINE.json configs[2]: d=64, n=8192, multiquadric, degree-1 tail, k=2, m=10000 -- the configuration the
                         north-star targets are quoted on.  One STEP = one cycle per rank: with N > 1 every rank runs its own
                         independent problem of that size (the path shards by problem, no data-path collective): weak scaling,
                         `value` = cycles of all ranks / wall time of the slowest rank.
  --config C4 | C5       the many-start workloads: one STEP = the whole batch (64 ZDT1 starts / --problems P of the 256 C5 problems),
                         problem p on rank p % N, one all_gather of the fixed-size result records per step (the only collective):
                         strong scaling, `value` = problems of the batch / wall time of the slowest rank.
  --config C2            single build+solve (no evaluations)

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5] [--problems P] [--no-cpu-baseline] [--no-callers]
With --gpus N > 1 and no torch.distributed environment the script launches the N ranks itself (python -m torch.distributed.run,
rendezvous on 127.0.0.1) BEFORE anything touches a GPU, and exits with the launcher's code; under an external
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it runs as one rank.
  --dry-run              launcher / sharding / gather rehearsal on CPU (gloo, no GPU, no engine): prints the JSON line with
                         "dry_run": true and value null.  Used by tests/test_bench_launch.py.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (~6.3 TB/s achievable)
GRAM_W_ISSUED_PER_N2D = 4   # gram_w_kernel computes every off-diagonal tile of X X^T in both orientations: n^2 d on top of the useful 3 n^2 d
FP64_MFMA_PEAK_TF = 78.6   # MI355X FP64 matrix peak (spec; v_mfma_f64_16x16x4_f64 = 16 FMA/clk/SIMD x 1024 SIMD x 2.4 GHz)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--no-batch", action="store_true", help="C4: per-problem mrbf_fit / mrbf_eval calls from worker threads instead of mrbf_batch_run")
    ap.add_argument("--problems", type=int, default=None, help="problems per step of the many-start configs (default: C4 64, C5 8 per rank)")
    ap.add_argument("--workers", type=int, default=0, help="host threads (contexts) per rank for the many-start configs (default: 4 for n <= 2048, else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--callers-only", action="store_true", help="(internal) print the callers' side measurement as one JSON line and exit")
    ap.add_argument("--no-callers", action="store_true", help="skip the side measurement of the path's callers (site selection round 4, Pascoletti-Serafini step)")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--gram-mode", type=int, default=0)
    ap.add_argument("--chol-impl", type=int, default=0)
    ap.add_argument("--eval-impl", type=int, default=0)
    return ap.parse_args()


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """N > 1 without a launcher: start the ranks as children of this (GPU-free) process and return their exit code"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.call(cmd, env=env)


# ---- CPU baselines (the oracle is the checker and the timed baseline, never the product) ---------------------------------
def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        return os.cpu_count() or 1


def cpu_baselines(cfg, C, Y, X):
    """Both modes of BASELINE.md section 2 on a bounded sample of the same workload, LAPACK LU shared between them:
       port      vectorised NumPy assembly (thread pool) + LAPACK dgesv + batched evaluation             -> `cpu_baseline`
       faithful  the reference's call pattern, single-threaded loops of the C restatement: per-pair norm(x - c) assembly, the same
                 dgesv, evaluation ONE point and ONE output per closure call, value and gradient sweeps separate
                 (AbstractSurrogateInterface.jl:98-106)                                                  -> `cpu_baseline_faithful`"""
    import scipy.linalg
    from oracle import c_oracle
    from oracle import rbf_oracle as orc

    kid = orc.KERNEL_IDS[cfg["kernel"]]
    a, b = orc.kernel_params(cfg["kernel"])
    n, m, k, deg = cfg["n"], cfg["m"], cfg["k"], cfg["deg"]
    cores = int(blas_threads())
    scipy.linalg.solve(np.eye(256) + 0.01 * np.ones((256, 256)), np.ones((256, 2)), check_finite=False)  # LAPACK / thread pool start-up is not timed
    t0 = time.perf_counter()
    Phi, Pi = orc.gram(C, kid, a, b, deg)
    t_asm_port = time.perf_counter() - t0
    q = Pi.shape[1]
    S = orc.saddle_matrix(Phi, Pi)
    rhs = np.vstack([Y, np.zeros((q, k))])
    t0 = time.perf_counter()
    sol = scipy.linalg.solve(S, rhs, assume_a="gen", check_finite=False)
    t_lu = time.perf_counter() - t0
    del S
    mod = orc.OracleModel(C, sol[:n].copy(), sol[n:].copy(), kid, a, b, deg)
    ms = min(m, 1024)
    t_eval_port = 0.0
    if ms > 0:
        t0 = time.perf_counter()
        mod.values(X[:ms])
        mod.jacs(X[:ms])
        t_eval_port = (time.perf_counter() - t0) * (m / ms)
    port = dict(value=1.0 / (t_asm_port + t_lu + t_eval_port), unit="cycles/s", cores=cores, kind="port", blas_threads=cores,
                sample="full fit n=%d: NumPy assembly %.2fs + LAPACK dgesv (N=%d) %.2fs; %d of %d evals (values+Jacobians, batched) "
                       "scaled to %.2fs" % (n, t_asm_port, n + q, t_lu, ms, m, t_eval_port))
    # faithful mode: bounded samples, scaled (assembly is O(rows), evaluation O(points))
    rows = min(n, max(64, int(2.0e9 / max(1, n * cfg["d"]))))           # ~2e9 scalar pair-dimension steps
    t0 = time.perf_counter()
    c_oracle.gram_cols(C, kid, a, b, rows)
    t_asm_f = (time.perf_counter() - t0) * (n / rows)
    mf = min(m, 64)
    t_eval_f = 0.0
    if mf > 0:
        t0 = time.perf_counter()
        c_oracle.eval_loop(C, mod.w, mod.lam, kid, a, b, deg, X[:mf], want_jac=True)
        t_eval_f = (time.perf_counter() - t0) * (m / mf)
    faithful = dict(value=1.0 / (t_asm_f + t_lu + t_eval_f), unit="cycles/s", cores=1, kind="port", blas_threads=cores,
                    sample="single-threaded C loops: per-pair norm assembly of %d of %d rows scaled to %.2fs + the same LAPACK dgesv %.2fs "
                           "(%d BLAS threads, as Julia's `\\`); %d of %d points, one output per closure call, value and gradient sweeps "
                           "separate, scaled to %.2fs" % (rows, n, t_asm_f, t_lu, cores, mf, m, t_eval_f))
    return port, faithful


# ---- one rank ----------------------------------------------------------------------------------------------------------
def gram_full_ms(worker, n, d, reps=7):
    """Median hipEvent time (ms) of the full-matrix assembly behind mrbf_gram (RBF.get_matrices, RbfModel.jl:374-375) on rank 0's
    first problem, device pointers in and out, outside the timed region; None when it cannot be measured."""
    try:
        import torch
        from morbit.jl_amd import _lib
        ctx = worker.ctx
        dC = worker.gram_inputs[0]
        kid, a, b, deg = worker.gram_inputs[1]
        Phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
        ts = []
        for _ in range(reps + 1):
            ms = ctypes.c_float()
            ctx.check(ctx.lib.mrbf_gram(ctx.h, n, d, _lib.as_ptr(dC), kid, a, b, deg, _lib.as_ptr(Phi), None, ctypes.byref(ms)))
            ts.append(ms.value)
        torch.cuda.synchronize()
        del Phi
        ts = sorted(ts[1:])
        return float(ts[len(ts) // 2])
    except Exception as e:  # the figure is an extra; never let it take the bench line down
        sys.stderr.write("gram_full_ms: %r\n" % (e,))
        return None


def measure_callers():
    """Side measurement, not part of `value`: the callers either side of the hot path at the BASELINE dimensions,
    through the same library -- wall time of mrbf_round4 (incl. the upload of the candidates) at d = 64 with 10^4 candidates and at
    d = 128 with 6000 (all accepted), and of one Pascoletti-Serafini step with Morbit's default budgets at d = 64 / 128 / 256 (workloads
    of tools/round4_bench.py and tools/ps_bench2.py; best of three / two calls after one warm-up call), and of the affine filter's pick
    loop (mrbf_affine_select) at d = 128 with 300 candidates."""
    import numpy as np
    import morbit.jl_amd as pkg
    from morbit.jl_amd import sampling, workloads as wl
    from morbit.jl_amd import pascoletti_serafini as ps
    res = {}
    try:
        for d, mc in ((64, 10000), (128, 6000)):
            rng = np.random.default_rng(1)
            x = np.full(d, 0.5)
            start = np.vstack([x[None, :], x[None, :] + 0.3 * np.eye(d)])
            cand = x[None, :] + 0.4 * (2.0 * rng.random((mc, d)) - 1.0)
            cfg = pkg.RbfConfig(kernel="cubic")
            best, nacc = 1e30, 0
            for rep in range(3):
                t0 = time.perf_counter()
                acc, st = sampling.rbf_round4_device(cfg, start, cand, 1.0, keep_state=True)
                dt = (time.perf_counter() - t0) * 1e3
                st.free()
                nacc = len(acc)
                if rep > 0:
                    best = min(best, dt)
            res["round4_d%d_%dcand" % (d, mc)] = {"ms": round(best, 3), "accepted": nacc}
        for d in (64, 128, 256):
            if d == 64:
                C = wl.problem("C3")[0]
                Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d
                cfg = pkg.RbfConfig(kernel="multiquadric")
            elif d == 128:
                C, Y, _ = wl.problem("C4", 0)
                cfg = pkg.RbfConfig(kernel="cubic")
            else:
                rng = np.random.default_rng(5)
                C = rng.random((2048, d))
                Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1)], 1) / d
                cfg = pkg.RbfConfig(kernel="cubic")
            mod = pkg.update_model(cfg, C, Y)
            x = C[0].copy() if d == 128 else np.full(d, 0.5)
            lb, ub = np.maximum(x - 0.1, 0), np.minimum(x + 0.1, 1)
            fx = pkg.eval_models_at_sites(mod, None, x[None, :])[0]
            best, omega, ne = 1e30, 0.0, 0
            for rep in range(3):
                stt = {}
                t0 = time.perf_counter()
                o = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=0, stats=stt)
                dt = (time.perf_counter() - t0) * 1e3
                if rep > 0:
                    best = min(best, dt)
                omega, ne = float(o[0]), int(stt["evals_ideal"] + stt["evals_ps"])
            mod.free()
            res["ps_step_d%d" % d] = {"ms": round(best, 3), "evaluations": ne, "omega": omega, "n": int(C.shape[0])}
        # the affine filter's pick loop: d = 128, 300 candidates in the box, as tools/affine_bench.py
        d, mc = 128, 300
        rng = np.random.default_rng(1)
        x = rng.random(d)
        seeds = list(x + 0.2 * (2 * rng.random((mc, d)) - 1))
        best, npick = 1e30, 0
        for rep in range(3):
            flt = sampling.AffinelyIndependentPointFilter(x, seeds, pivot_val=0.02)
            t0 = time.perf_counter()
            got = flt.collect()
            dt = (time.perf_counter() - t0) * 1e3
            npick = len(got)
            if rep > 0:
                best = min(best, dt)
        res["affine_select_d128_300cand"] = {"ms": round(best, 3), "picks": npick}
    except Exception as e:  # the side measurement must never cost the bench line
        res["error"] = repr(e)
    return res


def main():
    args = parse_args()
    if args.callers_only:  # child process of the default run (see `callers` below)
        print(json.dumps(measure_callers()), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # nothing below has run: no GPU call in the parent
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node == --gpus (or let bench.py spawn the ranks)"
                         % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    from morbit.jl_amd import manystart
    from morbit.jl_amd import workloads as wl

    cfg = wl.CONFIGS[args.config]
    many = args.config in ("C4", "C5")
    steps = args.steps if args.steps is not None else (3 if args.config == "C5" else (5 if many else 200))
    warmup = args.warmup if args.warmup is not None else (1 if many else 5)
    n, d, k, m = cfg["n"], cfg["d"], cfg["k"], cfg["m"]
    if many:
        P = args.problems if args.problems is not None else (cfg["problems"] if args.config == "C4" else 8 * world)
    else:
        P = world  # one problem per rank per step
    mine = manystart.shard_indices(P, rank, world)  # problem p -> rank p % world

    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    device = "cpu" if dry else "cuda"
    if not dry:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    PH = ("gram", "project", "factor", "solve", "eval")
    check = {}
    workers = []
    nworkers = 1
    if dry:
        class DryWorker:
            phases = {p: 0.0 for p in PH}
            fallbacks = 0

            def cycle(self, p):
                time.sleep(0.001)
                return [float(p), 0.0, 2.0, 0.0, 0.0, 0.0, 1.0, 0.0]
        workers = [DryWorker()]
    else:
        import morbit.jl_amd as pkg
        from morbit.jl_amd import _lib

        rcfg = pkg.RbfConfig(kernel=cfg["kernel"], polynomial_degree=cfg["deg"])
        kid, a, b = pkg.rbf_model._get_kernel_params(1.0, rcfg)
        # inputs of this rank's problems resident in HBM before the timed region
        host, dev = {}, {}
        for p in mine:
            C, Y, X = wl.problem(args.config, p)
            if p == mine[0]:
                host[p] = (C, Y, X)
            dev[p] = (torch.from_numpy(C).cuda(), torch.from_numpy(Y).cuda(), torch.from_numpy(X).cuda())
        torch.cuda.synchronize()
        # small problems are launch-latency bound: several host threads, each with its own context (stream, workspace), keep
        # more kernels in flight on the GPU -- the mrbf_batch_run arrangement (api.hip), here with device-resident inputs
        nworkers = args.workers if args.workers > 0 else (4 if (many and n <= 2048) else 1)
        nworkers = max(1, min(nworkers, max(1, len(mine))))

        class Worker:
            def __init__(self):
                self.ctx = pkg.Context(local_rank)
                self.ctx.set_option(_lib.OPT_GRAM_MODE, args.gram_mode)
                self.ctx.set_option(_lib.OPT_CHOL_IMPL, args.chol_impl)
                self.ctx.set_option(_lib.OPT_EVAL_IMPL, args.eval_impl)
                self.finfo, self.einfo = _lib.FitInfo(), _lib.EvalInfo()
                self.dV = torch.empty((max(m, 1), k), dtype=torch.float64, device="cuda")
                self.dJ = torch.empty((max(m, 1), d, k), dtype=torch.float64, device="cuda")
                self.dW = torch.empty((n, k), dtype=torch.float64, device="cuda")
                self.sums = torch.empty(2, dtype=torch.float64, device="cuda")
                self.phases = {p: 0.0 for p in PH}
                self.fallbacks = 0
                self.factor_ms = []   # per cycle: spread of the dominant kernel (a stalled persistent launch must show up here)
                self.factor_dev_ms = []   # the same launches by the kernel's own clock (excludes host-side gaps between the events)
                self.slow_launches = 0
                self.gram_inputs = (dev[mine[0]][0], (kid, a, b, cfg["deg"])) if mine else None

            def cycle(self, p):
                ctx, lib, finfo, einfo = self.ctx, self.ctx.lib, self.finfo, self.einfo
                dC, dY, dX = dev[p]
                h = _lib.c_vp()
                ctx.check(lib.mrbf_fit(ctx.h, n, d, k, _lib.as_ptr(dC), _lib.as_ptr(dY), kid, a, b, cfg["deg"], ctypes.byref(h),
                                       _lib.as_ptr(self.dW) if many else None, None, ctypes.byref(finfo)))
                if m > 0:
                    ctx.check(lib.mrbf_eval(ctx.h, h, m, _lib.as_ptr(dX), _lib.as_ptr(self.dV), _lib.as_ptr(self.dJ), ctypes.byref(einfo)))
                ctx.check(lib.mrbf_free_model(ctx.h, h))
                ph = self.phases
                ph["gram"] += finfo.ms_gram
                ph["project"] += finfo.ms_project
                ph["factor"] += finfo.ms_factor
                self.factor_ms.append(float(finfo.ms_factor))
                self.factor_dev_ms.append(float(finfo.ms_factor_device))
                self.slow_launches = int(finfo.slow_launches)
                ph["solve"] += finfo.ms_solve
                ph["eval"] += einfo.ms_total if m > 0 else 0.0
                if finfo.fallbacks & ~_lib.FB_LU:
                    self.fallbacks += 1
                return [float(p), 0.0, float(finfo.path), float(finfo.rel_residual), 0.0, 0.0, finfo.ms_total, einfo.ms_total if m > 0 else 0.0]

        class BatchWorker:
            """C4: all of this rank's problems through ONE mrbf_batch_run per step (one-launch small-problem fit + batched fused
            evaluation, csrc/batch.hip); inputs and outputs are torch tensors resident in HBM, handed over as device pointers"""

            def __init__(self, subset=None):
                mine = self.mine = list(subset) if subset is not None else mine_all
                self.ctx = pkg.Context(local_rank)          # keeps the device selected; the library pools its own batch contexts
                self.lib = self.ctx.lib
                self.finfo = _lib.FitInfo()
                self.phases = {p: 0.0 for p in PH}
                self.fallbacks = 0
                self.factor_ms = []
                P_ = len(mine)
                self.arr = (_lib.Problem * max(P_, 1))()
                self.res = (_lib.Result * max(P_, 1))()
                self.keep = []
                dp = lambda t: ctypes.cast(t.data_ptr(), _lib.c_dp)
                for j, p in enumerate(mine):
                    dC, dY, dX = dev[p]
                    dV = torch.empty((max(m, 1), k), dtype=torch.float64, device="cuda")
                    dJ = torch.empty((max(m, 1), d, k), dtype=torch.float64, device="cuda")
                    dW = torch.empty((n, k), dtype=torch.float64, device="cuda")
                    self.keep.append((dV, dJ, dW))
                    self.arr[j] = _lib.Problem(n, m, d, k, kid, cfg["deg"], a, b, dp(dC), dp(dY), dp(dX), dp(dW), None, dp(dV), dp(dJ))

            def step(self):
                mine = self.mine
                rc = self.lib.mrbf_batch_run(1, (ctypes.c_int32 * 1)(local_rank), len(mine), self.arr, self.res)
                assert rc == 0, "mrbf_batch_run failed: %d" % rc
                # the result records as columns of one byte view over the ctypes array (a Python loop over 64 structs costs ~0.4 ms a step)
                raw = np.frombuffer(self.res, dtype=np.uint8).reshape(-1, ctypes.sizeof(_lib.Result))[:len(mine)]
                col = lambda off, dt: raw[:, off:off + np.dtype(dt).itemsize].copy().view(dt)[:, 0]
                R, F = _lib.Result, _lib.FitInfo
                status = col(R.status.offset, np.int32)
                assert not status.any(), [(p, int(st)) for p, st in zip(mine, status) if st]
                fo = R.fit.offset
                recs = np.stack([np.asarray(mine, dtype=np.float64), np.zeros(len(mine)), col(fo + F.path.offset, np.int32).astype(np.float64),
                                 col(fo + F.rel_residual.offset, np.float64), col(R.checksum_w.offset, np.float64),
                                 col(R.checksum_vals.offset, np.float64), col(fo + F.ms_total.offset, np.float32).astype(np.float64),
                                 col(R.ms_eval.offset, np.float32).astype(np.float64)], axis=1) if mine else np.zeros((0, 8))
                self.fallbacks += int(np.count_nonzero(col(fo + F.fallbacks.offset, np.int32) & ~_lib.FB_LU)) if mine else 0
                if mine:   # the batch's times are shared by its members: count them once per step, spread over the problems below
                    self.phases["factor"] += self.res[0].fit.ms_factor
                    self.phases["eval"] += self.res[0].ms_eval
                    self.factor_ms.append(float(self.res[0].fit.ms_factor))
                    self.finfo = self.res[0].fit
                return recs

        mine_all = mine
        batched = many and n <= 512 and d <= 128 and not args.no_batch
        workers = [BatchWorker()] if batched else [Worker() for _ in range(nworkers)]
        # one checked cycle per rank (residual through the eval kernels), then the timed loop without the extra check
        w0 = workers[0]
        if batched:
            if mine:   # (the batched path always carries its residual check: it runs beside the evaluation of the queries)
                w0.step()
                check.update(path=w0.finfo.path, rel_residual=w0.finfo.rel_residual, max_pitw=w0.finfo.max_pitw, batched=True)
                assert w0.finfo.rel_residual < 1e-9, "fit residual too large: %r" % (check,)
        else:
            w0.ctx.set_option(_lib.OPT_RESIDUAL, 1)
            if mine:
                w0.cycle(mine[0])
                check.update(path=w0.finfo.path, rel_residual=w0.finfo.rel_residual, max_pitw=w0.finfo.max_pitw)
                assert w0.finfo.rel_residual < 1e-9, "fit residual too large: %r" % (check,)
            for w in workers:
                w.ctx.set_option(_lib.OPT_RESIDUAL, 0)

    pool = None
    if len(workers) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(len(workers))

    def step():
        if not dry and batched:
            recs = workers[0].step()
            local = np.asarray(recs, dtype=np.float64).reshape(-1, manystart.RECORD_LEN)
            return manystart.gather_records(local, P, device=device)
        if pool is None:
            recs = [workers[0].cycle(p) for p in mine]
        else:  # worker w takes this rank's problems w, w + nworkers, ... (ctypes releases the GIL inside the library)
            parts = list(pool.map(lambda wi: [workers[wi].cycle(p) for p in mine[wi::len(workers)]], range(len(workers))))
            recs = [r for part in parts for r in part]
        if many:  # the batch's only collective: fixed-size records, gathered to every rank
            local = np.asarray(recs, dtype=np.float64).reshape(-1, manystart.RECORD_LEN)
            return manystart.gather_records(local, P, device=device)
        return None

    for _ in range(warmup):
        step()
    for w in workers:
        for p in PH:
            w.phases[p] = 0.0
        w.fallbacks = 0
        if hasattr(w, "factor_ms"):
            w.factor_ms.clear()
        if hasattr(w, "factor_dev_ms"):
            w.factor_dev_ms.clear()
    barrier()
    t0 = time.perf_counter()
    table = None
    for _ in range(steps):
        table = step()
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    rank_times = [elapsed_local]
    if world > 1:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_times = [float(x.item()) for x in allt]
        elapsed = max(rank_times)
    ncyc = max(1, steps * len(mine))
    # ms per launch, hipEvents on the library's stream inside the timed region (rank 0's problems)
    phases = {p: sum(w.phases[p] for w in workers) / ncyc for p in PH}
    nfb = sum(w.fallbacks for w in workers)
    if nfb:
        check["fits_with_fallback"] = nfb
    fm = sorted(v for w in workers for v in getattr(w, "factor_ms", []))
    if fm:
        med = fm[len(fm) // 2]
        check["factor_ms"] = dict(min=fm[0], median=med, max=fm[-1], slow_launches=sum(1 for v in fm if v > 1.5 * med))
    fd = sorted(v for w in workers for v in getattr(w, "factor_dev_ms", []) if v > 0)
    if fd:
        # the persistent factorisation's own clock (first workgroup in -> last workgroup out): a launch that is slow here stalled on the
        # device; one that is slow in factor_ms only had a host-side gap inside the hipEvent bracket
        check["factor_device_ms"] = dict(min=fd[0], median=fd[len(fd) // 2], max=fd[-1],
                                         slow_launches_ctx=sum(getattr(w, "slow_launches", 0) for w in workers))

    if rank == 0:
        alg = wl.algorithmic(args.config)
        out = {
            "metric": "RBF build+solve+eval cycles/sec", "value": None if dry else P * steps / elapsed, "unit": "cycles/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "strong" if many else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["desc"], "n": n, "d": d, "k": k, "m": m, "kernel": cfg["kernel"],
                       "polynomial_degree": cfg["deg"], "problems_per_step": P, "contexts_per_gpu": len(workers),
                       "batched": bool(not dry and batched),
                       "parallelism": "problem-sharded x%d (problem p on rank p %% %d%s)" % (world, world, ", one all_gather of records per step" if many else "")},
            "rank_seconds": rank_times, "rank_imbalance": max(rank_times) / max(min(rank_times), 1e-12),
        }
        if dry:
            out["dry_run"] = True
        if many and table is not None:
            out["records"] = {"gathered": int(table.shape[0]), "failed": int(np.sum(table[:, 1] != 0)),
                              "worst_rel_residual": (float(np.nanmax(table[:, 3])) if np.isfinite(table[:, 3]).any() else None)}
            assert table.shape[0] == P and np.array_equal(table[:, 0], np.arange(P)), "record gather lost problems"
        if not dry:
            # The fit's own assembly at d <= 64 with a polynomial tail is gram_w_kernel (gram_fused.hip): lower triangle only, fused with
            # W = Phi [1 Xc].  USEFUL work: n^2 d (products with the centres on the lower triangle) + 2 n^2 d (W) = 3 n^2 d flops against
            # 4 n^2 + 8 n d bytes, i.e. MFMA-bound; `frac` is priced on those useful flops.  What the kernel issues on top of that
            # (mirrored tiles computed a second time, if the build still does) is reported beside it as `issued_flops`, never in `frac`.
            # The full-matrix kernel behind mrbf_gram (RBF.get_matrices) is measured below (kernels.gram_full: 8 n^2 + 8 n d bytes, HBM roof).
            q_tail = 0 if cfg["deg"] < 0 else (1 if cfg["deg"] == 0 else d + 1)
            fused_gram = (not batched) and d <= 64 and q_tail >= 1 and n >= 1024 and args.gram_mode == 0
            gram_mfma = d >= 96 or fused_gram
            gram_flops = 3.0 * n * n * d if fused_gram else alg["gram_flops"]
            # projection: with the fused assembly the product Phi Q1 (and its read of Phi) is inside gram_w_kernel and counted there; what
            # the phase still does is the rank-2q update of the lower triangle: 2 n^2 q flops, lower triangle read + written = 2 * 4 n^2 B
            proj_flops = 2.0 * n * n * q_tail if fused_gram else alg["project_flops"]
            proj_bytes = 2.0 * 4.0 * n * n if fused_gram else alg["project_bytes"]
            proj_mfma = proj_flops / (FP64_MFMA_PEAK_TF * 1e12) > (1.0 if fused_gram else 2.0) * proj_bytes / (HBM_PEAK_GBS * 1e9)
            kernels = {
                "gram": dict(bound="mfma" if gram_mfma else "hbm",
                             achieved=(gram_flops / max(phases["gram"], 1e-9) / 1e9) if gram_mfma else alg["gram_bytes"] / max(phases["gram"], 1e-9) / 1e6,
                             peak=FP64_MFMA_PEAK_TF if gram_mfma else HBM_PEAK_GBS, unit="TFLOP/s" if gram_mfma else "GB/s", ms=phases["gram"]),
                "factor": dict(bound="mfma", achieved=alg["factor_flops"] / max(phases["factor"] * 1e-3, 1e-12) / 1e12, peak=FP64_MFMA_PEAK_TF,
                               unit="TFLOP/s", ms=phases["factor"]),
                # general path: the projection's 4 n^2 q flops clearly outweigh its 3 x 8 n^2 bytes from q ~ 128 on (C5, q = 257: 3.5 ms at the
                # fp64 peak against 0.8 ms at the HBM peak)
                "project": (dict(bound="mfma", achieved=proj_flops / max(phases["project"], 1e-9) / 1e9, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                                 ms=phases["project"], bytes_gbs=proj_bytes / max(phases["project"], 1e-9) / 1e6) if proj_mfma else
                            dict(bound="hbm", achieved=proj_bytes / max(phases["project"], 1e-9) / 1e6, peak=HBM_PEAK_GBS, unit="GB/s",
                                 ms=phases["project"], flops_tf=proj_flops / max(phases["project"], 1e-9) / 1e9)),
            }
            if fused_gram:
                pair_ms = phases["gram"] + phases["project"]
                kernels["gram_project"] = dict(bound="mfma", achieved=(gram_flops + proj_flops) / max(pair_ms, 1e-9) / 1e9, peak=FP64_MFMA_PEAK_TF,
                                               unit="TFLOP/s", ms=pair_ms, note="assembly + projection together: 3 n^2 d + 2 n^2 q useful flops")
            if fused_gram:
                kernels["gram"]["note"] = "gram_w_kernel: lower triangle + W = Phi [1 Xc] in one pass (3 n^2 d useful flops, 4 n^2 + 8 n d bytes)"
                kernels["gram"]["issued_flops"] = float(GRAM_W_ISSUED_PER_N2D) * n * n * d
                kernels["project"]["note"] = "rank-2q update of the lower triangle (the product Phi Q1 is inside gram_w_kernel): 2 n^2 q flops, 8 n^2 bytes"
                gf = gram_full_ms(workers[0], n, d) if workers and getattr(workers[0], "gram_inputs", None) else None
                if gf:
                    kernels["gram_full"] = dict(bound="hbm", achieved=alg["gram_bytes"] / gf / 1e6, peak=HBM_PEAK_GBS, unit="GB/s", ms=gf,
                                                note="mrbf_gram (full matrix, RBF.get_matrices), median of 7 launches outside the timed region")
            if m > 0:
                kernels["eval"] = dict(bound="mfma", achieved=alg["eval_flops"] / max(phases["eval"] * 1e-3, 1e-12) / 1e12, peak=FP64_MFMA_PEAK_TF,
                                       unit="TFLOP/s", ms=phases["eval"])
            if batched:
                # one launch does the whole fit of every problem of the batch (small_fit_kernel): one entry, priced with all of its flops
                fit_flops = alg["gram_flops"] + alg["project_flops"] + alg["factor_flops"] + alg["solve_flops"]
                kernels = {"fit": dict(bound="mfma", achieved=fit_flops / max(phases["factor"] * 1e-3, 1e-12) / 1e12, peak=FP64_MFMA_PEAK_TF,
                                       unit="TFLOP/s", ms=phases["factor"], note="small_fit_kernel, whole batch in one launch; ms = batch time / problems"),
                           "eval": kernels["eval"]}
            for kd in kernels.values():
                kd["frac"] = kd["achieved"] / kd["peak"]
            dom = max([p for p in ("gram", "factor", "eval") if (p in kernels or p == "factor") and (m > 0 or p != "eval")], key=lambda p: phases[p])
            if batched and dom == "factor":
                dom = "fit"
            roof = dict(kernels[dom])
            roof["kernel"] = dom
            roof["traffic"] = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    roof["traffic"] = json.load(open(pmc)).get(args.config, {}).get(dom)
                except Exception:
                    pass
            out.update({"roofline": roof, "kernels": kernels, "phases_ms": phases, "check": check})
            if batched and world == 1 and P >= 16:
                # what ONE of eight GPUs would see of this batch (problems p = 0, 8, 16, ...): the same library call on an eighth of the
                # problems, timed on this GPU -- total / share bounds the strong scaling of the many-start path from one-GPU numbers
                share = BatchWorker(mine[::8])
                for _ in range(max(2, warmup)):
                    share.step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps):
                    share.step()
                torch.cuda.synchronize()
                ms8 = (time.perf_counter() - t1) / steps * 1e3
                out["one_of_eight_gpus"] = dict(problems=len(share.mine), ms_per_step=ms8, fit_ms=float(share.res[0].fit.ms_factor),
                                                eval_ms=float(share.res[0].ms_eval), speedup_bound_8gpu=out["ms_per_step"] / ms8)
            if many and not batched and world == 1 and P >= 8:
                # C5-sized problems: the share ONE of eight GPUs would see (problems p = 0, 8, 16, ...), the same per-problem cycles on
                # this GPU.  No phase is shared between problems, so total / share is the bound on the 1 -> 8 GPU speed-up that follows
                # from one-GPU numbers (the counterpart of the C4 entry above)
                share = mine[::8]
                for p_ in share[:1]:
                    workers[0].cycle(p_)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps):
                    for p_ in share:
                        workers[0].cycle(p_)
                torch.cuda.synchronize()
                ms8 = (time.perf_counter() - t1) / steps * 1e3
                out["one_of_eight_gpus"] = dict(problems=len(share), ms_per_step=ms8, speedup_bound_8gpu=out["ms_per_step"] / ms8)
            if not args.no_cpu_baseline and world == 1 and args.config in ("C2", "C3", "C4"):
                C, Y, X = host[mine[0]]
                port, faithful = cpu_baselines(cfg, C, Y, X)
                out["cpu_baseline"] = port
                out["cpu_baseline_faithful"] = faithful
            if not args.no_callers and world == 1 and args.config == "C3":
                # in a child process of its own: a second context in THIS process would share hardware queues with the bench's streams
                # (the walk's two streams then serialise: 6.5 instead of 5.1 ms) -- a caller of the library has one context
                import subprocess
                try:
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--callers-only"], capture_output=True, text=True, timeout=300)
                    out["callers"] = json.loads(cp.stdout.strip().splitlines()[-1])
                except Exception as e:  # the side measurement must never cost the bench line
                    out["callers"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if pool is not None:
        pool.shutdown()
    for w in workers:
        if hasattr(w, "ctx"):
            w.ctx.close()


if __name__ == "__main__":
    main()
