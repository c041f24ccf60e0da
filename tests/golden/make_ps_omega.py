"""Independent reference values for the Pascoletti-Serafini step (tests/golden/ps_omega.json).

What omega SHOULD be for four fixed problems, computed WITHOUT any of the product's code: the model is the oracle's CPU fit
(oracle/rbf_oracle.py: LAPACK solve of the saddle system), the subproblem of /root/reference/src/descent.jl:434-510

    min t   s.t.  m_l(x) - m_l(x_n) - t r_l <= 0 (l = 1..k),  t in [-1, 0],  lb <= x <= ub

is solved by SciPy's SLSQP with analytic Jacobians from several starts (x_n itself, x_n pushed along minus the mean gradient,
random points of the box); omega* = -min t over the starts.  Two directions per problem:

  bench     r = m(x_n) - reference point (-1, -1): the paper benchmark's configuration (examples/large_scale_benchmarks.jl:215-219)
  default   r = f(x_n) - local ideal point (descent.jl:369-412), the ideal point by L-BFGS-B on every objective over the box

The GPU tests rebuild the same models on the device (weights agree to ~1e-10, tests/test_gpu_parity.py) and compare the device
step's omega with these numbers.  Runs in the CPU container:  python tests/golden/make_ps_omega.py
"""
import json
import os
import sys
import time

import numpy as np
import scipy.optimize as so

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

from oracle import rbf_oracle as orc  # noqa: E402

wl = importlib.import_module("morbit.jl_amd.workloads")


def problems():
    """name -> (sites, values, kernel name, x_n, box half width); the SAME construction as tests/test_pascoletti_serafini.py"""
    out = {}
    for d, n in ((24, 700), (256, 2048)):
        rng = np.random.default_rng(d)
        C = rng.random((n, d))
        Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1)], axis=1) / d
        out["d%d" % d] = (C, Y, "cubic", np.full(d, 0.9), 0.1)
    C = wl.problem("C3")[0]
    Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1)], axis=1) / 64
    out["d64"] = (C, Y, "multiquadric", np.full(64, 0.9), 0.1)
    C, Y, _ = wl.problem("C4", 0)
    out["d128"] = (C, Y, "cubic", C[0].copy(), 0.1)
    # conflicting objectives (round 5): the Pareto set of the two quadratics is the segment 0.3 * ones .. 0.7 * ones; a start at its
    # middle, 0.5 * ones, pushed off it by +- 0.15 per coordinate (fixed seed) has gradients that nearly oppose each other -- a narrow
    # descent cone, omega* small but not zero
    for name, base in (("d64c", "d64"), ("d256c", "d256")):
        C, Y, kernel, _, half = out[base]
        d = C.shape[1]
        x = 0.5 + 0.15 * (2.0 * np.random.default_rng(1000 + d).random(d) - 1.0)
        out[name] = (C, Y, kernel, x, half)
    return out


def solve_ps(model, x, lb, ub, mx, r, rng, nstart=6):
    d = x.size
    k = mx.size

    def cons(z):
        return -(model.values(z[None, 1:])[0] - mx - z[0] * r)          # SLSQP: c(z) >= 0

    def cons_jac(z):
        J = model.jacs(z[None, 1:])[0]                                    # k x d
        return -np.hstack([-r[:, None], J])

    g = model.jacs(x[None, :])[0] / r[:, None]
    starts = [np.concatenate([[0.0], x])]
    step = -g.mean(axis=0)
    if np.abs(step).max() > 0:
        for s in (0.25, 1.0):
            starts.append(np.concatenate([[-0.01], np.clip(x + s * (ub - lb).max() * step / np.abs(step).max(), lb, ub)]))
    while len(starts) < nstart:
        starts.append(np.concatenate([[0.0], lb + rng.random(d) * (ub - lb)]))
    best_t, best_x = 0.0, x.copy()
    bounds = [(-1.0, 0.0)] + list(zip(lb, ub))
    for z0 in starts:
        res = so.minimize(lambda z: z[0], z0, jac=lambda z: np.eye(1, d + 1, 0)[0], method="SLSQP", bounds=bounds,
                          constraints=[dict(type="ineq", fun=cons, jac=cons_jac)], options=dict(maxiter=400, ftol=1e-12))
        z = np.clip(res.x, [b[0] for b in bounds], [b[1] for b in bounds])
        # the t the returned x really achieves (feasible by construction)
        t = float(np.max((model.values(z[None, 1:])[0] - mx) / r))
        t = min(max(t, -1.0), 0.0) if t <= 0 else 0.0
        if t < best_t:
            best_t, best_x = t, z[1:].copy()
    return -best_t, best_x


def local_ideal_point(model, x, lb, ub, rng, nstart=4):
    k = model.values(x[None, :]).shape[1]
    ideal = np.empty(k)
    for l in range(k):
        best = np.inf
        for s in range(nstart):
            x0 = x if s == 0 else lb + rng.random(x.size) * (ub - lb)
            res = so.minimize(lambda z: model.values(z[None, :])[0, l], x0, jac=lambda z: model.jacs(z[None, :])[0, l], method="L-BFGS-B",
                              bounds=list(zip(lb, ub)), options=dict(maxiter=500, ftol=1e-14, gtol=1e-10))
            best = min(best, float(res.fun))
        ideal[l] = best
    return ideal


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ps_omega.json")
    out = {"_doc": "omega* of the Pascoletti-Serafini subproblem by SciPy SLSQP on the oracle's model; see make_ps_omega.py"}
    if os.path.exists(path) and "--all" not in sys.argv:   # keep what is on file, compute what is missing (--all: everything again)
        out.update(json.load(open(path)))
    fits = {}
    for name, (C, Y, kernel, x, half) in problems().items():
        if name in out:
            continue
        t0 = time.time()
        kid = orc.KERNEL_IDS[kernel]
        a, b = orc.kernel_params(kernel)
        key = (id(C), kernel)
        if key not in fits:
            fits[key] = orc.fit(C, Y, kid, a, b, 1)
        model = fits[key]
        lb, ub = np.maximum(x - half, 0.0), np.minimum(x + half, 1.0)
        mx = model.values(x[None, :])[0]
        rng = np.random.default_rng(99)
        r_bench = mx + 1.0
        om_b, xb = solve_ps(model, x, lb, ub, mx, r_bench, rng)
        ideal = local_ideal_point(model, x, lb, ub, rng)
        r_def = mx - ideal
        om_d, xd = solve_ps(model, x, lb, ub, mx, r_def, rng) if np.all(r_def > 0) else (0.0, x)
        out[name] = dict(n=int(C.shape[0]), d=int(C.shape[1]), kernel=kernel, mx=mx.tolist(), omega_bench=om_b, ideal=ideal.tolist(),
                         r_default=r_def.tolist(), omega_default=om_d, step_bench=float(np.abs(xb - x).max()))
        print("%-5s n=%d: omega* bench %.5f, default %.5f (r_default %s)  [%.1f s]" % (name, C.shape[0], om_b, om_d, np.round(r_def, 5), time.time() - t0))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
