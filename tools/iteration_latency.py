"""Per-iteration call latencies of the iteration rehearsal (tests/iteration_rehearsal.py) at Morbit's own sizes, with the CPU
reference pattern timed beside every call at the same sizes (the stated baseline, not a target):

  update_model   oracle: single-threaded per-pair norm assembly (C restatement) + LAPACK LU of the saddle system   (RbfModel.jl:759-763)
  round 4        the host mirror of Wild's incremental loop (Givens / bordered Cholesky bookkeeping per candidate, kernel values
                 from the NumPy oracle)                                                                              (RbfModel.jl:352-499)
  PS step        evaluations the device step used x the cost of ONE one-point closure call per objective (value sweep), measured on
                 a sample with the C restatement's one-point loop                                                    (descent.jl:434-510)
  backtrack      first accepted index + 2 sequential one-point evaluations                                            (descent.jl:150-185)

usage: python tools/iteration_latency.py [C1|C4] [iterations]   -> table on stdout (copy to profiles/)
"""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import morbit  # noqa: E402,F401
import morbit.jl_amd as pkg  # noqa: E402
from morbit.jl_amd import sampling as sp  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import rbf_oracle as orc  # noqa: E402
from tests import test_gpu_iteration as tgi  # noqa: E402


def cpu_reference(run, rec):
    """the reference's call pattern on the host at this iteration's sizes (ms)"""
    cfg = run.cfg
    kid, a, b = pkg.rbf_model._get_kernel_params(rec["delta"], cfg)
    S, V = np.array(run.sites), np.array(run.values)
    n, d = rec["n"], run.d
    idx = rec["training"]
    C, Y = S[idx], V[idx]
    out = {}
    t0 = time.perf_counter()
    c_oracle.gram_cols(C, kid, a, b, n)                      # faithful assembly, one thread
    Phi, Pi = orc.gram(C, kid, a, b, cfg.polynomial_degree)
    t_asm = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = orc.fit(C, Y, kid, a, b, cfg.polynomial_degree)    # (assembles again, vectorised) + LAPACK LU
    out["update_model"] = ((time.perf_counter() - t0) + t_asm) * 1e3
    # round 4: the host mirror's incremental loop
    n0 = d + 1
    found = idx[:n0]
    kb = lambda X, Cc: orc.phi(kid, a, b, orc.pairwise_dist(np.atleast_2d(X), np.atleast_2d(Cc)))
    x = S[idx[0]]
    d2 = cfg.θ_enlarge_2 * run.delta_max
    lb2, ub2 = run._box(x, d2)
    t0 = time.perf_counter()
    sp._rbf_round4(S[: rec["n_db_after_round3"]], lb2, ub2, x, rec["delta"], list(found), cfg, kernel_block=kb)
    out["round4"] = (time.perf_counter() - t0) * 1e3
    # one-point closures
    m = 64
    X = x[None, :] + 0.01 * np.random.default_rng(0).standard_normal((m, d))
    t0 = time.perf_counter()
    c_oracle.eval_loop(C, ref.w, ref.lam, kid, a, b, cfg.polynomial_degree, X, want_jac=False)
    per_point = (time.perf_counter() - t0) / m               # all k outputs of one point; the reference sweeps once PER output
    k = Y.shape[1]
    out["ps_step"] = rec["ps_evals"] * k * per_point * 1e3
    out["backtrack"] = (rec.get("backtrack_loops", 0) + 2) * k * per_point * 1e3
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "C4"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    make = tgi._c1 if which == "C1" else tgi._c4
    make(3).run(3)                                           # start-up (code objects, arenas) is not what the table is about
    run = make(7)
    print("# %s rehearsal, %d iterations on one context; ms per call (wall, incl. host staging), CPU reference pattern beside it" % (which, iters))
    print("# (the CPU side runs in this process every fifth iteration: a device call right after it may show a one-off 50-90 ms while the")
    print("#  BLAS thread pool winds down -- an artefact of this table, not of the call: tests/test_gpu_iteration.py has no such outliers)")
    print("# it   n  n_db   delta    omega      rho  | affine  round4(cpu)      fit(cpu)        ps_step(cpu)      backtrack(cpu) | fit path")
    for it in range(iters):
        rec = run.iterate(it)
        # the database as round 4 saw it: without the trial point this iteration appended afterwards
        rec["n_db_after_round3"] = len(run.sites) - (1 if "rho" in rec and not np.isnan(rec.get("rho", np.nan)) else 0)
        rec["training"] = rec.get("training_indices")
        cpu = {}
        if rec["training"] is not None and (it % 5 == 0 or it == iters - 1):
            cpu = cpu_reference(run, rec)
        ms = rec["ms"]
        f = lambda key: ("%7.2f" % ms[key]) if key in ms else "      -"
        g = lambda key: ("(%9.1f)" % cpu[key]) if key in cpu else "(        -)"
        print("%4d %4d %5d %7.4f %8.2e %8.3f | %s %s%s %s%s %s%s %s%s | %s" % (
            it, rec["n"], rec["n_db"], rec["delta"], rec["omega"], rec.get("rho", float("nan")), f("affine_filter"), f("round4"), g("round4"),
            f("update_model"), g("update_model"), f("ps_step"), g("ps_step"), f("backtrack"), g("backtrack"), rec.get("fit")), flush=True)
    run.keeper.drop("db")


if __name__ == "__main__":
    main()
