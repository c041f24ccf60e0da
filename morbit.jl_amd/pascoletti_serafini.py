"""Population-batched Pascoletti-Serafini descent step.

Mirrors the reference's `PascolettiSerafiniConfig`, `_get_global_dir`, `_ps_max_evals`, `compute_local_ideal_point`,
`_ps_optimization` and `get_criticality(::PascolettiSerafiniConfig, ...)` (src/descent.jl:320-581).  The reference hands
one-point closures to NLopt's GN_ISRES (descent.jl:385, :505): every candidate costs k sweeps over all n centres.  NLopt is
not part of this path; the subproblem solver is owned here -- an ISRES-style (mu, lambda) evolution strategy with
stochastic ranking (Runarsson & Yao 2005, the algorithm behind GN_ISRES; population 20 (dim + 1) and survivor fraction
1/7 as NLopt's defaults) whose whole generation is evaluated by ONE batched surrogate call
(`eval_container_objectives_at_scaled_sites` -> `mrbf_eval`).  The k single-objective minimisations of the local ideal
point run as k populations side by side in the same batched calls.

Parity with the reference is NOT the NLopt trajectory (random, and NLopt's own stream) but the contract of
`get_criticality`: the problem solved (variables chi = [t; x], t in [-1, 0], x in [lb_eff, ub_eff], objective t,
constraints m_l(x) - m_l(x_n) - t r_l <= 0, descent.jl:434-447, :478-510), the evaluation budgets (`_ps_max_evals`,
descent.jl:414-432), the start value t0 = -0.5 (:555), the criticality short-cut `any(r .<= 0)` (:546-549), the failure
fallback (:571-572) and the returned tuple `omega = |tau|, (x_trial, mx_trial, step length in the inf-norm)` (:573-579).
A returned point is always feasible for the subproblem (tau <= 0 means no modelled objective gets worse along r).
"""
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np


@dataclass
class PascolettiSerafiniConfig:
    """descent.jl:324-350 (same fields and defaults; the algorithm symbols are kept for interface compatibility)."""
    reference_point: Sequence[float] = field(default_factory=list)
    reference_direction: Sequence[float] = field(default_factory=list)
    trust_region_factor: float = 1.0
    max_ps_problem_evals: int = -1
    max_ps_polish_evals: int = -1
    max_ideal_point_problem_evals: int = -1
    main_algo: str = "GN_ISRES"
    reference_algo: str = "GN_ISRES"
    reference_trust_region_factor: float = 1.1
    ps_polish_algo: Optional[str] = None

    def __post_init__(self):
        assert all(v > 0 for v in self.reference_direction), "The components of the `reference_direction` cannot be negative."


def _get_global_dir(cfg, fx):
    """descent.jl:360-368"""
    if len(cfg.reference_direction):
        return np.asarray(cfg.reference_direction, dtype=np.float64)
    if len(cfg.reference_point):
        return np.asarray(fx, dtype=np.float64) - np.asarray(cfg.reference_point, dtype=np.float64)
    return None


def _ps_max_evals(desc_cfg, n_vars):
    """descent.jl:414-432"""
    max_evals = 500 * (n_vars + 1) if desc_cfg.max_ps_problem_evals < 0 else desc_cfg.max_ps_problem_evals
    if desc_cfg.ps_polish_algo is None:
        return max_evals, 0
    if desc_cfg.max_ps_polish_evals < 0:
        g = int(np.floor(max_evals * 3 / 4))
        return g, max_evals - g
    return max_evals, desc_cfg.max_ps_polish_evals


# ---- ISRES, a whole generation per evaluation call ------------------------------------------------------------------
def _stochastic_ranking(f, phi, rng, pf=0.45):
    """Runarsson & Yao's stochastic ranking (lam bubble sweeps, adjacent individuals compared by objective when both are
    feasible or with probability pf, else by constraint violation): `mrbf_stochastic_rank`, host code of libmrbf -- NLopt does
    this in C as well; the sequential sweeps cost ~6 ms per generation in NumPy and ~0.1 ms there."""
    from . import _lib
    lib = _lib.load()
    lam = int(f.shape[0])
    f = np.ascontiguousarray(f, dtype=np.float64)
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    u = rng.random(lam * max(lam - 1, 1))
    idx = np.empty(lam, dtype=np.int32)
    rc = lib.mrbf_stochastic_rank(lam, _lib.as_ptr(f), _lib.as_ptr(phi), _lib.as_ptr(u), float(pf), _lib.as_ptr(idx))
    if rc != 0:
        raise _lib.MrbfError(rc, "mrbf_stochastic_rank")
    return idx.astype(np.int64)


class _Isres:
    """State of one (mu, lambda) run; `ask()` gives the generation to evaluate, `tell(f, g)` ranks it and breeds the next one."""

    def __init__(self, lb, ub, x0, max_evals, rng, xtol_rel=1e-3, extra_starts=()):
        self.lb, self.ub = np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64)
        n = self.lb.size
        self.n, self.rng, self.max_evals, self.xtol_rel = n, rng, max_evals, xtol_rel
        self.lam = 20 * (n + 1)
        self.mu = int(np.ceil(self.lam / 7.0))
        span = self.ub - self.lb
        self.X = self.lb + rng.random((self.lam, n)) * span
        self.X[0] = np.clip(np.asarray(x0, dtype=np.float64), self.lb, self.ub)
        for j, xs in enumerate(extra_starts):
            self.X[1 + j] = np.clip(np.asarray(xs, dtype=np.float64), self.lb, self.ub)
        self.S = np.tile(span / np.sqrt(n), (self.lam, 1))
        self.tau = 1.0 / np.sqrt(2.0 * np.sqrt(n))     # phi = 1 (expected rate of convergence)
        self.taup = 1.0 / np.sqrt(2.0 * n)
        self.alpha, self.gamma = 0.2, 0.85
        self.evals = 0
        self.best_x, self.best_f, self.best_phi = self.X[0].copy(), np.inf, np.inf
        self.done = False

    def ask(self):
        budget = self.max_evals - self.evals
        return self.X[: max(0, min(self.lam, budget))]

    def tell(self, f, G):
        """f: (m,) objective, G: (m, n_constraints) constraint values (<= 0 feasible) for the first m individuals of the generation."""
        m = f.shape[0]
        self.evals += m
        if m == 0:
            self.done = True
            return
        phi = np.sum(np.maximum(G, 0.0) ** 2, axis=1) if G.size else np.zeros(m)
        bad = ~np.isfinite(f) | ~np.isfinite(phi)
        f = np.where(bad, np.inf, f)
        phi = np.where(bad, np.inf, phi)
        # best-so-far: feasible beats infeasible, then the objective (NLopt keeps the best feasible point it has seen)
        feas = phi == 0.0
        j = int(np.argmin(np.where(feas, f, np.inf))) if feas.any() else int(np.argmin(phi))   # this generation's best
        if (feas[j] and (self.best_phi > 0.0 or f[j] < self.best_f)) or (not feas[j] and phi[j] < self.best_phi):
            self.best_x, self.best_f, self.best_phi = self.X[j].copy(), float(f[j]), float(phi[j])
        if self.evals >= self.max_evals or m < self.lam:
            self.done = True
            return
        order = _stochastic_ranking(f, phi, self.rng)
        P, PS = self.X[order[: self.mu]], self.S[order[: self.mu]]
        # stop like NLopt's xtol_rel on the survivors
        if np.all(np.ptp(P, axis=0) <= self.xtol_rel * np.maximum(np.abs(P[0]), 1e-300)):
            self.done = True
            return
        lam, mu, n = self.lam, self.mu, self.n
        par = np.arange(lam) % mu
        Xn, Sn = P[par].copy(), PS[par].copy()
        # differential variation along the direction towards the best individual (first mu - 1 offspring)
        nd = mu - 1
        if nd > 0:
            xd = P[:nd] + self.gamma * (P[0][None, :] - P[1:nd + 1])
            inside = (xd >= self.lb) & (xd <= self.ub)
            Xn[:nd] = np.where(inside, xd, P[:nd])
        # log-normal self-adaptive mutation for the rest, re-drawing components that leave the box (10 tries, then the parent's)
        m = lam - nd
        pr = par[nd:]
        s = PS[pr] * np.exp(self.taup * self.rng.standard_normal((m, 1)) + self.tau * self.rng.standard_normal((m, n)))
        s = np.minimum(s, (self.ub - self.lb) / np.sqrt(n))
        xm = P[pr] + s * self.rng.standard_normal((m, n))
        for _ in range(10):
            out = (xm < self.lb) | (xm > self.ub)
            if not out.any():
                break
            xm = np.where(out, P[pr] + s * self.rng.standard_normal((m, n)), xm)
        xm = np.where((xm < self.lb) | (xm > self.ub), P[pr], xm)
        Xn[nd:] = xm
        Sn[nd:] = PS[pr] + self.alpha * (s - PS[pr])  # exponential smoothing
        self.X, self.S = Xn, Sn


def _run_populations(runs, evaluate):
    """Advance several ISRES runs in lock step; `evaluate(list of (m_r, n) arrays) -> list of (f_r, G_r)` is called once per
    generation for all runs together (one batched surrogate sweep).  Returns the number of batched calls."""
    calls = 0
    while not all(r.done for r in runs):
        asks = [r.ask() if not r.done else r.X[:0] for r in runs]
        results = evaluate(asks)
        calls += 1
        for r, (f, G) in zip(runs, results):
            if not r.done:
                r.tell(np.asarray(f, dtype=np.float64), np.asarray(G, dtype=np.float64))
    return calls


def compute_local_ideal_point(x_n, lb_eff, ub_eff, eval_objectives: Callable, n_out, max_evals, rng,
                              eval_constraints: Optional[Callable] = None, stats=None):
    """descent.jl:404-412: the k minima of the single objectives over the local box -- k populations, one batched call per generation.
    eval_objectives(X) -> (m, k); eval_constraints(X) -> (m, n_c) with <= 0 feasible (optional)."""
    runs = [_Isres(lb_eff, ub_eff, x_n, max_evals, np.random.default_rng(rng.integers(2 ** 63))) for _ in range(n_out)]

    def evaluate(asks):
        sizes = [a.shape[0] for a in asks]
        X = np.vstack(asks) if sum(sizes) else np.empty((0, len(x_n)))
        F = eval_objectives(X) if X.shape[0] else np.empty((0, n_out))
        G = eval_constraints(X) if (eval_constraints is not None and X.shape[0]) else np.empty((X.shape[0], 0))
        out, o = [], 0
        for l, m in enumerate(sizes):
            out.append((F[o:o + m, l], G[o:o + m]))
            o += m
        return out

    calls = _run_populations(runs, evaluate)
    if stats is not None:
        stats["ideal_point_calls"] = calls
        stats["ideal_point_evals"] = sum(r.evals for r in runs)
    return np.array([r.best_f for r in runs])


def _ps_optimization(t0, x, lb, ub, eval_objectives: Callable, mx, r, max_evals, rng, eval_constraints=None, stats=None):
    """descent.jl:478-510 with the constraint functions of :434-447 folded in: minimise t over chi = [t; x]."""
    lbc = np.concatenate([[-1.0], np.asarray(lb, dtype=np.float64)])
    ubc = np.concatenate([[0.0], np.asarray(ub, dtype=np.float64)])
    # besides the reference's start [t0; x] the population holds [0; x], which is always feasible (m(x) - m(x) - 0 r = 0): the
    # step can then never fail for lack of a feasible individual, and tau <= 0 always
    run = _Isres(lbc, ubc, np.concatenate([[t0], x]), max_evals, rng, extra_starts=[np.concatenate([[0.0], x])])
    mx, r = np.asarray(mx, dtype=np.float64), np.asarray(r, dtype=np.float64)

    def evaluate(asks):
        chi = asks[0]
        if chi.shape[0] == 0:
            return [(np.empty(0), np.empty((0, mx.size)))]
        X = np.ascontiguousarray(chi[:, 1:])
        F = eval_objectives(X)
        G = F - mx[None, :] - chi[:, :1] * r[None, :]          # m_l(x) - m_l(x_n) - t r_l <= 0
        if eval_constraints is not None:
            G = np.hstack([G, eval_constraints(X)])
        return [(chi[:, 0].copy(), G)]

    calls = _run_populations([run], evaluate)
    if stats is not None:
        stats["ps_calls"] = calls
        stats["ps_evals"] = run.evals
    if run.best_phi > 0.0 or not np.isfinite(run.best_f):
        return np.nan, np.full(len(x), np.nan), "FAILURE"
    return run.best_f, run.best_x[1:].copy(), ("MAXEVAL_REACHED" if run.evals >= max_evals else "XTOL_REACHED")


def _min_norm_weights(M):
    """weights of the minimum-norm point of the convex hull of vectors with Gram matrix M (Gilbert / Frank-Wolfe steps)"""
    k = M.shape[0]
    lam = np.full(k, 1.0 / k)
    for _ in range(500):
        Ml = M @ lam
        gg = float(lam @ Ml)
        b = int(np.argmin(Ml))
        if gg - Ml[b] <= 1e-14 * max(gg, 1e-300):
            break
        den = gg - 2.0 * Ml[b] + M[b, b]
        gamma = min(1.0, max(0.0, (gg - Ml[b]) / den)) if den > 0.0 else 1.0
        lam *= (1.0 - gamma)
        lam[b] += gamma
    return lam


def _polish(tau, x_min, lb, ub, eval_objectives, eval_jacobians, mx, r, max_evals, eval_constraints=None, n_step=12):
    """Gradient polish of the PS solution (the reference hands a local NLopt algorithm, descent.jl:560-569): multi-objective steepest
    descent on max_l (m_l(x) - m_l(x_n)) / r_l -- minus the minimum-norm convex combination of the nearly active scaled gradients,
    projected at active bounds -- with all step sizes of a line search in ONE batched call; every accepted iterate is feasible.
    The same algorithm as `ps_polish` in csrc/ps_solver.hip."""
    t, x = float(tau), np.array(x_min, dtype=np.float64)
    lb, ub = np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64)
    evals = 0
    while evals + 1 + n_step <= max_evals:
        F = (eval_objectives(x[None, :])[0] - mx) / r
        J = eval_jacobians(x[None, :])[0]                      # k x d
        evals += 1
        act = np.flatnonzero(F >= F.max() - 0.05)
        G = J[act] / r[act, None]
        blocked = ((x <= lb) & np.all(G > 0.0, axis=0)) | ((x >= ub) & np.all(G < 0.0, axis=0))
        G = np.where(blocked[None, :], 0.0, G)
        lam = _min_norm_weights(G @ G.T)
        d = -(lam @ G)
        dmax, width = np.abs(d).max(), (ub - lb).max()
        if not dmax > 0.0 or not width > 0.0:
            break
        steps = (width / dmax) * 0.5 ** np.arange(n_step)
        XT = np.clip(x[None, :] + steps[:, None] * d[None, :], lb, ub)
        FT = eval_objectives(XT)
        evals += n_step
        tt = np.clip(np.max((FT - mx[None, :]) / r[None, :], axis=1), -1.0, 0.0)
        feas = np.all(FT - mx[None, :] - tt[:, None] * r[None, :] <= 1e-14, axis=1) & np.isfinite(tt)
        if eval_constraints is not None:
            feas &= np.all(eval_constraints(XT) <= 0.0, axis=1)
        ok = feas & (tt < t - 1e-12)
        if not ok.any():
            break
        j = int(np.argmin(np.where(ok, tt, np.inf)))
        x, t = XT[j], float(tt[j])
        if t <= -1.0:
            break
    return t, x, "SUCCESS"


def _ps_step_problem(desc_cfg, models, roles, k, x, x_n, fx_n, lb_eff, ub_eff, lin=None, seed=0, stats=None, eq_tol=-1.0):
    """one mrbf_ps_step_problem call; returns (rc, result) -- rc != 0: no result"""
    import ctypes

    from . import _lib

    ctx = models[0].ctx
    x, x_n = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(x_n, dtype=np.float64)
    d = x_n.size
    lb, ub = np.ascontiguousarray(lb_eff, dtype=np.float64), np.ascontiguousarray(ub_eff, dtype=np.float64)
    fx = np.ascontiguousarray(fx_n, dtype=np.float64)
    r = _get_global_dir(desc_cfg, fx)
    r = None if r is None else np.ascontiguousarray(r, dtype=np.float64)
    g_evals, l_evals = _ps_max_evals(desc_cfg, d)
    opts = _lib.PsOptions(max_ideal_evals=int(desc_cfg.max_ideal_point_problem_evals), max_ps_evals=int(g_evals),
                          max_polish_evals=int(l_evals), reserved=0, seed=int(seed) & (2 ** 64 - 1), t0=-0.5, xtol_rel=1e-3)
    A_eq, b_eq, A_in, b_in = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (lin or (None,) * 4)]
    handles = (ctypes.c_void_p * len(models))(*[m.model.value if hasattr(m.model, "value") else m.model for m in models])
    roles_c = (ctypes.c_int32 * max(len(roles), 1))(*roles)
    prob = _lib.PsProblem(n_models=len(models), n_objectives=k, models=handles, roles=roles_c,
                          n_lin_eq=0 if b_eq is None else b_eq.size, n_lin_ineq=0 if b_in is None else b_in.size,
                          A_eq=None if A_eq is None else A_eq.ctypes.data, b_eq=None if b_eq is None else b_eq.ctypes.data,
                          A_ineq=None if A_in is None else A_in.ctypes.data, b_ineq=None if b_in is None else b_in.ctypes.data, eq_tol=float(eq_tol))
    info = _lib.PsInfo()
    xt, mt, r_out = np.empty(d), np.empty(k), np.empty(k)
    rc = ctx.lib.mrbf_ps_step_problem(ctx.h, ctypes.byref(prob), _lib.as_ptr(x_n), _lib.as_ptr(lb), _lib.as_ptr(ub), _lib.as_ptr(fx),
                                      _lib.as_ptr(r), ctypes.byref(opts), _lib.as_ptr(xt), _lib.as_ptr(mt), _lib.as_ptr(r_out), ctypes.byref(info))
    if rc != 0:
        return rc, None
    if stats is not None:
        stats.update(info.asdict())
        stats["r"] = r_out.copy()
        stats["path"] = "device"
    if info.status == _lib.PS_CRITICAL:
        return 0, (0, x_n.copy(), mt, 0)          # descent.jl:546-549
    if info.status == _lib.PS_FAILURE:
        return 0, (0, x.copy(), mt, 0)            # descent.jl:571-572
    return 0, (abs(float(info.tau)), (xt, mt, float(np.linalg.norm(x - xt, ord=np.inf))))


def get_criticality_device(desc_cfg, model, x, x_n, fx_n, lb_eff, ub_eff, seed=0, stats=None):
    """descent.jl:512-581 with the whole subproblem solver on the device for objectives that are the outputs of ONE grouped
    RbfModel `model` and no constraints.  Same returns as `get_criticality`; raises when the device call refuses (use
    `get_criticality_container` for the routing Morbit's `get_criticality` method gets in HipRbf.jl)."""
    rc, out = _ps_step_problem(desc_cfg, [model], list(range(model.num_outputs)), model.num_outputs, x, x_n, fx_n, lb_eff, ub_eff,
                               seed=seed, stats=stats)
    model.ctx.check(rc)
    return out


def get_criticality_container(desc_cfg, sc, scal, x, x_n, fx_n, lb_eff, ub_eff, lin=None, seed=0, rng=None, stats=None, eq_tol=1e-8):
    """`get_criticality(::PascolettiSerafiniConfig, mop, scal, x_it, x_it_n, db, sc, ac)` (descent.jl:512-581) as HipRbf.jl routes it:
    the decision table of the library (mrbf_dispatch_ps / mrbf_dispatch_after) picks the device solver (objectives and modelled
    constraints over several grouped models, linear constraints `lin = (A_eq, b_eq, A_ineq, b_ineq)` in scaled variables) or the
    reference method -- here the host-loop mirror on batched container sweeps.  Never raises because of a size limit.
    `eq_tol`: an equality constraint counts as satisfied when |h| <= eq_tol (mrbf_ps_problem.eq_tol)."""
    from . import _lib
    from . import surrogates as sg

    lib = _lib.load()
    plan = sg.container_plan(sc)
    d = int(np.asarray(x_n).size)
    lin = lin or (None, None, None, None)
    n_lin = sum(0 if b is None else int(np.asarray(b).size) for b in (lin[1], lin[3]))
    if lib.mrbf_dispatch_ps(d, plan["k"], len(plan["models"]), plan["n_con"], n_lin, plan["n_foreign"]) == _lib.DISPATCH_DEVICE:
        rc, out = _ps_step_problem(desc_cfg, plan["models"], plan["roles"], plan["k"], x, x_n, fx_n, lb_eff, ub_eff, lin=lin, seed=seed, stats=stats, eq_tol=eq_tol)
        if rc == 0:
            return out
        if not lib.mrbf_dispatch_after(_lib.ENTRY_PS_STEP, rc):
            plan["models"][0].ctx.check(rc)
    # the reference method: one-point handles in Morbit, the population-batched host loop here
    if stats is not None:
        stats["path"] = "reference"

    def ev(X):
        return sg.eval_container_objectives_at_scaled_sites(sc, scal, X)

    def jac(X):
        return sg.eval_container_objectives_jacobian_at_scaled_sites(sc, scal, X)

    has_con = plan["n_con"] > 0 or n_lin > 0

    def con(X):
        X = np.atleast_2d(X)
        cols = []
        if sc.lists["nl_eq_constraint"]:
            h = sg.eval_container_nl_eq_constraints_at_scaled_sites(sc, scal, X)
            cols.append(np.where(np.abs(h) > eq_tol, np.abs(h), 0.0))
        if sc.lists["nl_ineq_constraint"]:
            cols.append(sg.eval_container_nl_ineq_constraints_at_scaled_sites(sc, scal, X))
        if lin[1] is not None and np.asarray(lin[1]).size:
            h = X @ np.asarray(lin[0], dtype=np.float64).T - np.asarray(lin[1], dtype=np.float64)[None, :]
            cols.append(np.where(np.abs(h) > eq_tol, np.abs(h), 0.0))
        if lin[3] is not None and np.asarray(lin[3]).size:
            cols.append(X @ np.asarray(lin[2], dtype=np.float64).T - np.asarray(lin[3], dtype=np.float64)[None, :])
        return np.hstack(cols)

    return get_criticality(desc_cfg, x, x_n, fx_n, lb_eff, ub_eff, ev, eval_jacobians=jac, eval_constraints=con if has_con else None,
                           rng=rng if rng is not None else np.random.default_rng(seed), stats=stats)


def get_criticality(desc_cfg, x, x_n, fx_n, lb_eff, ub_eff, eval_objectives: Callable, eval_jacobians: Optional[Callable] = None,
                    eval_constraints: Optional[Callable] = None, rng=None, stats=None):
    """descent.jl:512-581.  `x` / `x_n`: scaled iterate and the point the step starts from (the same unless a normal step was taken);
    `fx_n`: true objective values at x_n; `eval_objectives(X) -> (m, k)` surrogate values (e.g.
    `lambda X: eval_container_objectives_at_scaled_sites(sc, scal, X)`).  Returns `(0, x_n, mx, 0)` when critical, else
    `(omega, (x_trial, mx_trial, step_length))` exactly like the reference."""
    rng = np.random.default_rng(0) if rng is None else rng
    x, x_n = np.asarray(x, dtype=np.float64), np.asarray(x_n, dtype=np.float64)
    n_vars = x_n.size
    r = _get_global_dir(desc_cfg, fx_n)
    max_evals_ip = 500 * (n_vars + 1) if desc_cfg.max_ideal_point_problem_evals < 0 else desc_cfg.max_ideal_point_problem_evals
    mx = np.asarray(eval_objectives(x_n[None, :])[0], dtype=np.float64)
    if r is None:
        ideal = compute_local_ideal_point(x_n, lb_eff, ub_eff, eval_objectives, mx.size, max_evals_ip, rng, eval_constraints, stats)
        r = np.asarray(fx_n, dtype=np.float64) - ideal
    if np.any(r <= 0):
        return 0, x_n.copy(), mx, 0
    g_evals, l_evals = _ps_max_evals(desc_cfg, n_vars)
    tau, x_min, ret = _ps_optimization(-0.5, x_n, lb_eff, ub_eff, eval_objectives, mx, r, g_evals, rng, eval_constraints, stats)
    fail = ret == "FAILURE" or not np.isfinite(tau) or np.any(np.isnan(x_min))
    if l_evals > 0 and not fail and eval_jacobians is not None:
        t2, x2, ret2 = _polish(tau, x_min, np.asarray(lb_eff), np.asarray(ub_eff), eval_objectives, eval_jacobians, mx, r, l_evals, eval_constraints)
        if ret2 != "FAILURE" and np.isfinite(t2) and not np.any(np.isnan(x2)):
            tau, x_min = t2, x2
    if fail:
        return 0, x.copy(), mx, 0
    omega = abs(float(tau))
    mx_trial = np.asarray(eval_objectives(x_min[None, :])[0], dtype=np.float64)
    return omega, (x_min, mx_trial, float(np.linalg.norm(x - x_min, ord=np.inf)))
