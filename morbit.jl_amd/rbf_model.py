"""Host-side mirror of Morbit's RBF surrogate interface for the hot path.

Mirrors names, argument meaning and error behaviour of /root/reference/src/models/RbfModel.jl
(RbfConfig :66-112, RbfModel :33-46, _get_kernel_params :665-690, update_model :743-767,
eval_models / get_gradient / get_jacobian :783-800) on top of the C ABI in include/mrbf.h.
All arithmetic runs in libmrbf's HIP kernels; nothing here computes a surrogate on the CPU.
`scal` arguments are accepted and ignored exactly like the reference (models live in scaled space).
"""
import ast
import ctypes
import math
import operator
from dataclasses import dataclass, fields

import numpy as np

from . import _lib

# Morbit.RbfKernels (RbfModel.jl:48-54); index = C-ABI kernel id
RbfKernels = ["cubic", "inv_multiquadric", "multiquadric", "thin_plate_spline", "gaussian"]


@dataclass(frozen=True)
class RbfConfig:
    """Configuration type for local RBF surrogate models (RbfModel.jl:58-112). Same fields and defaults."""

    kernel: str = "cubic"
    shape_parameter: object = float("nan")  # Union{String, Float64}
    polynomial_degree: int = 1
    θ_enlarge_1: float = 2.0
    θ_enlarge_2: float = 2.0
    θ_pivot: float = None  # 1 / (2 θ_enlarge_1)
    θ_pivot_cholesky: float = 1e-7
    require_linear: bool = True
    max_model_points: int = -1
    use_max_points: bool = False
    optimized_sampling: bool = True
    max_evals: int = 2 ** 63 - 1  # typemax(Int64)

    def __post_init__(self):
        if self.θ_pivot is None:
            object.__setattr__(self, "θ_pivot", 1.0 / (2.0 * self.θ_enlarge_1))
        sp = self.shape_parameter
        is_str = isinstance(sp, str)
        if not is_str:
            sp = float(sp)
            object.__setattr__(self, "shape_parameter", sp)
        nan = (not is_str) and math.isnan(sp)
        # the @assert block, RbfModel.jl:102-111
        assert self.θ_enlarge_1 * self.θ_pivot <= 1, "θ_pivot must be <= θ_enlarge_1^(-1)."
        assert self.kernel in RbfKernels, "`kernel` not supported. See `RbfKernels` for available symbols."
        if not is_str:
            assert self.kernel != "thin_plate_spline" or nan or (sp % 1 == 0 and sp >= 1), \
                "Invalid shape_parameter for :thin_plate_spline."
            assert self.kernel != "cubic" or nan or (sp % 1 == 0 and sp % 2 == 1), "Invalid shape_parameter for :cubic."
            assert nan or sp > 0, "Shape parameter must be strictly positive."
        assert self.θ_enlarge_1 >= 1 and self.θ_enlarge_2 >= 1, "θ's must be >= 1."
        assert self.polynomial_degree in (-1, 0, 1), "polynomial_degree must be -1, 0 or 1 (RbfModel.jl:21)"

    # isequal / hash over all fields so configs are combinable (RbfModel.jl:125-130); NaN == NaN like isequal
    def _key(self):
        out = []
        for f in fields(self):
            v = getattr(self, f.name)
            out.append("NaN" if isinstance(v, float) and math.isnan(v) else v)
        return tuple(out)

    def __eq__(self, other):
        return isinstance(other, RbfConfig) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())


def max_evals(cfg):
    return cfg.max_evals  # RbfModel.jl:120


def combinable(cfg):
    return True  # RbfModel.jl:121


_BINOPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
           ast.Pow: operator.pow}
_FUNCS = {"sqrt": math.sqrt, "exp": math.exp, "log": math.log, "abs": abs, "min": min, "max": max}


def parse_shape_param_string(delta, expr_str):
    """Evaluate a shape-parameter string such as "1/Δ" with Δ bound (RbfModel.jl:135-143).
    The reference Meta.parse + @eval's the string; here a small arithmetic evaluator is used instead."""
    src = expr_str.replace("^", "**")
    tree = ast.parse(src, mode="eval")

    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return float(node.value)
        if isinstance(node, ast.Name) and node.id in ("Δ", "delta", "Delta"):
            return float(delta)
        if isinstance(node, ast.BinOp) and type(node.op) in _BINOPS:
            return _BINOPS[type(node.op)](ev(node.left), ev(node.right))
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = ev(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in _FUNCS:
            return float(_FUNCS[node.func.id](*[ev(a) for a in node.args]))
        raise ValueError("unsupported expression in shape parameter string: %r" % expr_str)

    return float(ev(tree))


def _get_kernel_params(delta, cfg):
    """(kernel_id, a, b) for the C ABI; follows _get_kernel_params (RbfModel.jl:665-690) with the package
    defaults substituted where the reference returns `nothing` (NaN shape parameter, :673)."""
    sp = cfg.shape_parameter
    if isinstance(sp, str):
        sp = parse_shape_param_string(delta, sp)
    kid = RbfKernels.index(cfg.kernel)
    nan = math.isnan(sp)
    if cfg.kernel == "gaussian":
        return kid, (1.0 if nan else sp), 0.0
    if cfg.kernel in ("inv_multiquadric", "multiquadric"):
        return kid, (1.0 if nan else sp), 0.5
    if cfg.kernel == "cubic":
        return kid, (3.0 if nan else float(int(sp))), 0.0
    return kid, (2.0 if nan else float(int(sp))), 0.0  # thin_plate_spline


class RbfModel:
    """Wraps a device-resident interpolation model (RbfModel.jl:33-38): `model` is the mrbf_model handle."""

    def __init__(self, ctx, handle, n, d, k, q, fully_linear=False, weights=None, poly=None, info=None):
        self.ctx, self.model = ctx, handle
        self.n, self.d, self.k, self.q = n, d, k, q
        self.fully_linear = bool(fully_linear)
        self.weights, self.poly, self.info = weights, poly, info

    @property
    def num_outputs(self):
        return self.k

    def free(self):
        if self.model is not None and self.ctx.h:
            self.ctx.lib.mrbf_free_model(self.ctx.h, self.model)
        self.model = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # ---- batched entry points (what SURVEY.md section 8f rank 2 adds to the container API)
    def eval_sites(self, X, want_values=True, want_jac=False, out_vals=None, out_jac=None, info=None):
        """values (m x k) and/or Jacobians (m x k x d) at m scaled sites in one mrbf_eval call.
        X / outputs may be NumPy arrays (host) or torch CUDA tensors (used in place)."""
        lib, ctx = self.ctx.lib, self.ctx
        is_np = not hasattr(X, "data_ptr")
        if is_np:
            X = _lib.host_f64(X).reshape(-1, self.d)
        m = int(X.shape[0])
        vals = jac = None
        if want_values:
            vals = out_vals if out_vals is not None else (np.empty((m, self.k)) if is_np else None)
            assert vals is not None, "pass out_vals for device tensors"
        if want_jac:
            jac = out_jac if out_jac is not None else (np.empty((m, self.d, self.k)) if is_np else None)
            assert jac is not None, "pass out_jac for device tensors"
        ei = _lib.EvalInfo()
        ctx.check(lib.mrbf_eval(ctx.h, self.model, m, _lib.as_ptr(X), _lib.as_ptr(vals), _lib.as_ptr(jac), ctypes.byref(ei)))
        if info is not None:
            info.update(ei.asdict())
        if want_jac and isinstance(jac, np.ndarray) and out_jac is None:
            jac = np.transpose(jac, (0, 2, 1))  # per-point k x d column-major block -> (m, k, d) view
        return vals, jac


def fully_linear(rbf):
    return rbf.fully_linear  # RbfModel.jl:40


def num_outputs(rbf):
    return rbf.num_outputs  # RbfModel.jl:41


def set_fully_linear(rbf, val):
    rbf.fully_linear = bool(val)  # RbfModel.jl:43-46
    return None


def update_model(cfg, training_sites, training_values, delta=1.0, fully_linear=False, ctx=None):
    """The arithmetic half of update_model (RbfModel.jl:743-767): given the training sites and values that
    `_collect_indices(meta)` selected from the database, assemble + factorise + solve on the GPU.
    Returns an RbfModel (the reference returns `(RbfModel(inner_model, meta.fully_linear), meta)`)."""
    ctx = ctx or _lib.default_context()
    C = _lib.host_f64(training_sites)
    assert C.ndim == 2, "training_sites: n sites of dimension d"
    n, d = C.shape
    Y = _lib.host_f64(training_values).reshape(n, -1)
    k = Y.shape[1]
    kid, a, b = _get_kernel_params(delta, cfg)
    q = 0 if cfg.polynomial_degree < 0 else (1 if cfg.polynomial_degree == 0 else d + 1)
    W = np.empty((n, k))
    L = np.empty((max(q, 1), k))
    h = _lib.c_vp()
    info = _lib.FitInfo()
    ctx.check(ctx.lib.mrbf_fit(ctx.h, n, d, k, _lib.as_ptr(C), _lib.as_ptr(Y), kid, a, b, cfg.polynomial_degree,
                               ctypes.byref(h), _lib.as_ptr(W), _lib.as_ptr(L), ctypes.byref(info)))
    return RbfModel(ctx, h, n, d, k, q, fully_linear, W, L[:q], info.asdict())


init_model = update_model      # RbfModel.jl:738-741 delegates
improve_model = update_model   # RbfModel.jl:770-776 delegates


def model_from_coeffs(cfg, sites, weights, poly, delta=1.0, fully_linear=False, ctx=None):
    ctx = ctx or _lib.default_context()
    C = _lib.host_f64(sites)
    n, d = C.shape
    W = _lib.host_f64(weights).reshape(n, -1)
    k = W.shape[1]
    kid, a, b = _get_kernel_params(delta, cfg)
    q = 0 if cfg.polynomial_degree < 0 else (1 if cfg.polynomial_degree == 0 else d + 1)
    L = _lib.host_f64(poly).reshape(q, k) if q else None
    h = _lib.c_vp()
    ctx.check(ctx.lib.mrbf_model_from_coeffs(ctx.h, n, d, k, _lib.as_ptr(C), _lib.as_ptr(W), _lib.as_ptr(L), kid, a, b,
                                             cfg.polynomial_degree, ctypes.byref(h)))
    return RbfModel(ctx, h, n, d, k, q, fully_linear, W, L, None)


def eval_models(mod, scal, x_hat, ell=None):
    """Evaluate `mod` at scaled site x̂; with `ell` (int or list of ints) only those outputs
    (RbfModel.jl:783-790; RefSurrogate passes index vectors, AbstractSurrogateInterface.jl:159-164)."""
    vals, _ = mod.eval_sites(np.asarray(x_hat, dtype=np.float64)[None, :])
    v = vals[0]
    return v if ell is None else v[ell]


def get_gradient(mod, scal, x_hat, ell):
    """Gradient vector of output `ell` at x̂ (RbfModel.jl:792-795). Bit-equal to the Jacobian row (same kernel)."""
    return get_jacobian(mod, scal, x_hat)[ell]


def get_jacobian(mod, scal, x_hat, rows=None):
    """k x d Jacobian (or selected rows) at x̂ (RbfModel.jl:797-800)."""
    _, jac = mod.eval_sites(np.asarray(x_hat, dtype=np.float64)[None, :], want_values=False, want_jac=True)
    J = np.ascontiguousarray(jac[0])
    return J if rows is None else J[rows]


# batched twins
def eval_models_at_sites(mod, scal, X):
    return mod.eval_sites(X)[0]


def get_jacobians_at_sites(mod, scal, X):
    return np.ascontiguousarray(mod.eval_sites(X, want_values=False, want_jac=True)[1])


def get_matrices(cfg, centers, delta=1.0, ctx=None, want_pi=True):
    """Phi (n x n) and Pi (n x q): RBF.get_matrices(phi, centers; poly_deg) as used by round 4 (RbfModel.jl:374-375)."""
    ctx = ctx or _lib.default_context()
    C = _lib.host_f64(centers)
    n, d = C.shape
    kid, a, b = _get_kernel_params(delta, cfg)
    q = 0 if cfg.polynomial_degree < 0 else (1 if cfg.polynomial_degree == 0 else d + 1)
    Phi = np.empty((n, n), order="F")
    Pi = np.empty((n, max(q, 1)), order="F")
    ms = ctypes.c_float()
    ctx.check(ctx.lib.mrbf_gram(ctx.h, n, d, _lib.as_ptr(C), kid, a, b, cfg.polynomial_degree, _lib.as_ptr(Phi),
                                _lib.as_ptr(Pi) if (want_pi and q) else None, ctypes.byref(ms)))
    return Phi, Pi[:, :q], ms.value
