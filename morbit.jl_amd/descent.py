"""Host-side mirror of the descent-step consumers of the surrogate path.

Mirrors /root/reference/src/descent.jl: SteepestDescentConfig backtracking fields (:55-66),
_armijo_condition (:137-143) and _backtrack (:150-185).  The reference evaluates the <= max_loops
step sizes one after another; here all of them go to the device in one batch (mrbf_backtrack when the
objectives are one grouped RbfModel, else one batched container sweep) and the same loop logic picks
the step, so the returned (x_plus, mx_plus, step) are those of the sequential loop.
"""
import ctypes
import math
from dataclasses import dataclass

import numpy as np

from . import _lib
from . import surrogates as sg


@dataclass
class SteepestDescentConfig:
    strict_backtracking: bool = True
    armijo_const_rhs: float = 1e-6
    armijo_const_shrink: float = 0.75
    min_stepsize: float = 10 * np.finfo(np.float64).eps
    max_loops: int = None
    normalize: bool = True

    def __post_init__(self):
        if self.max_loops is None:  # descent.jl:61-66
            base = self.min_stepsize if self.min_stepsize > 0 else np.finfo(np.float64).eps
            self.max_loops = int(math.floor(math.log(base) / math.log(self.armijo_const_shrink)))
        assert self.armijo_const_rhs > 0


def _armijo_condition(strict, Mx, Mx_plus, step_size, omega, const_rhs):
    if strict:  # Val{true}, descent.jl:137-139
        return bool(np.all((Mx - Mx_plus) >= step_size * const_rhs * omega))
    return bool(np.max(Mx) - np.max(Mx_plus) >= step_size * const_rhs * omega)  # descent.jl:141-143


def _single_model(sc):
    """the inner RbfModel when the decision table (mrbf_dispatch_backtrack, the same call HipRbf.jl makes) sends `_backtrack` to
    mrbf_backtrack: every objective a RefSurrogate row of ONE grouped device model, outputs in order; else None"""
    plan = sg.container_plan(sc, objectives_only=True)
    lib = _lib.load()
    if lib.mrbf_dispatch_backtrack(len(plan["models"]), plan["n_foreign"], int(plan["in_order"])) == _lib.DISPATCH_DEVICE:
        return plan["models"][0]
    return None


def _backtrack(x, direction, step_size, omega, sc, cfg, scal=None):
    """Returns (x_plus, mx_plus, step) like descent.jl:150-185, plus the loop count as 4th value."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    direction = np.ascontiguousarray(direction, dtype=np.float64)
    min_step = cfg.min_stepsize if cfg.min_stepsize >= 0 else np.finfo(np.float64).eps
    inner = _single_model(sc)
    if inner is not None:
        ctx = inner.ctx
        xp = np.empty_like(x)
        mxp = np.empty(inner.num_outputs)
        step = np.empty_like(x)
        nl = ctypes.c_int32()
        ctx.check(ctx.lib.mrbf_backtrack(ctx.h, inner.model, _lib.as_ptr(x), _lib.as_ptr(direction), float(step_size),
                                         float(omega), int(cfg.strict_backtracking), cfg.armijo_const_rhs,
                                         cfg.armijo_const_shrink, min_step, cfg.max_loops, _lib.as_ptr(xp),
                                         _lib.as_ptr(mxp), _lib.as_ptr(step), ctypes.byref(nl)))
        return xp, mxp, step, nl.value
    # general container: one batched sweep over all trial points, then the reference's loop logic
    steps = [float(step_size)]
    for _ in range(cfg.max_loops):
        steps.append(steps[-1] * cfg.armijo_const_shrink)
    X = np.vstack([x[None, :]] + [x[None, :] + s * direction[None, :] for s in steps])
    M = sg.eval_container_objectives_at_scaled_sites(sc, scal, X)
    mx = M[0]
    i = 0
    while i < cfg.max_loops:
        if _armijo_condition(cfg.strict_backtracking, mx, M[i + 1], steps[i], omega, cfg.armijo_const_rhs):
            break
        if steps[i] <= min_step:
            break
        i += 1
    return X[i + 1], M[i + 1], steps[i] * direction, i
