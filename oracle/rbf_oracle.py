"""CPU oracle for Morbit.jl's RbfConfig hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain fp64 NumPy restatement of the arithmetic behind
    update_model -> RBF.RBFInterpolationModel(...)      /root/reference/src/models/RbfModel.jl:743-767
    eval_models / get_gradient / get_jacobian           /root/reference/src/models/RbfModel.jl:783-800
    RBF.get_matrices (Phi, Pi)                          /root/reference/src/models/RbfModel.jl:374-375
    _get_kernel_params                                  /root/reference/src/models/RbfModel.jl:665-690

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it, and only as the checker.  The product path (morbit.jl_amd) never imports it.

PARITY UNPINNED BY THE REFERENCE: the arithmetic itself lives in the third-party
Julia package RadialBasisFunctionModels.jl (compat 0.3.4, /root/reference/Project.toml:25,50)
which is not vendored under /root/reference, and Julia is not installed, so the
reference cannot be run here and its test-suite holds no golden numbers for this
path (test/rbf_models.jl asserts properties only: interpolation :104,
grad == Jacobian row :105-109, grad ~ AD :110-115, container Jacobian ~ AD :164-168).
The oracle is therefore pinned by
  (i)  those properties (tests/test_oracle.py),
  (ii) the sign convention Morbit itself relies on: Z'PhiZ must be positive
       definite for cholesky() at RbfModel.jl:394-395, i.e. kernels are
       conditionally POSITIVE definite (multiquadric enters as -sqrt(1+(a r)^2)),
  (iii) an independent implementation, scipy.interpolate.RBFInterpolator, on
       interpolant values (tests/golden/make_golden.py).

Published radial functions restated (RadialBasisFunctionModels.jl 0.3.x docs):
    Gaussian(a=1)              phi(r) = exp(-(a r)^2)                    cpd order 0
    Multiquadric(a=1,b=1/2)    phi(r) = (-1)^ceil(b) (1+(a r)^2)^b       cpd order ceil(b)
    InverseMultiquadric(a,b)   phi(r) = (1+(a r)^2)^(-b)                 cpd order 0
    Cubic(b=3)                 phi(r) = (-1)^ceil(b/2) r^b               cpd order ceil(b/2)
    ThinPlateSpline(k=2)       phi(r) = (-1)^(k+1) r^(2k) log r, 0 at 0  cpd order k+1
Polynomial tail: canonical monomials of degree <= poly_deg (<= 1 in Morbit,
RbfModel.jl:21,73-74), ordered [1, x_1, ..., x_d].
Weights: dense solve of the saddle system [Phi Pi; Pi' 0][w; lam] = [Y; 0] (LAPACK LU).
"""
import math

import numpy as np

# order = Morbit.RbfKernels, RbfModel.jl:48-54
KERNEL_IDS = {
    "cubic": 0,
    "inv_multiquadric": 1,
    "multiquadric": 2,
    "thin_plate_spline": 3,
    "gaussian": 4,
}
KERNEL_NAMES = {v: k for k, v in KERNEL_IDS.items()}


def kernel_params(kernel, shape_parameter=float("nan")):
    """(a, b) for the C-ABI from Morbit's (kernel, shape_parameter).

    Follows _get_kernel_params (RbfModel.jl:665-690); NaN -> package defaults
    (RbfModel.jl:673).  a = alpha (gaussian / (inv_)multiquadric), beta (cubic)
    or k (thin plate spline); b = exponent beta of (inv_)multiquadric, else 0.
    """
    sp = float(shape_parameter)
    if kernel == "gaussian":
        return (1.0 if math.isnan(sp) else sp, 0.0)
    if kernel in ("multiquadric", "inv_multiquadric"):
        return (1.0 if math.isnan(sp) else sp, 0.5)
    if kernel == "cubic":
        return (3.0 if math.isnan(sp) else float(int(sp)), 0.0)
    if kernel == "thin_plate_spline":
        return (2.0 if math.isnan(sp) else float(int(sp)), 0.0)
    raise ValueError("kernel not supported: %r" % (kernel,))


def cpd_order(kid, a, b):
    if kid == 0:
        return int(math.ceil(a / 2.0))
    if kid == 2:
        return int(math.ceil(b))
    if kid == 3:
        return int(a) + 1
    return 0


def phi(kid, a, b, rho):
    """Radial function value at rho >= 0 (array or scalar)."""
    rho = np.asarray(rho, dtype=np.float64)
    if kid == 4:
        return np.exp(-(a * rho) ** 2)
    if kid == 2:
        return (-1.0) ** math.ceil(b) * (1.0 + (a * rho) ** 2) ** b
    if kid == 1:
        return (1.0 + (a * rho) ** 2) ** (-b)
    if kid == 0:
        return (-1.0) ** math.ceil(a / 2.0) * rho ** a
    if kid == 3:
        k = int(a)
        with np.errstate(divide="ignore", invalid="ignore"):
            v = (-1.0) ** (k + 1) * rho ** (2 * k) * np.log(rho)
        return np.where(rho == 0.0, 0.0, v)
    raise ValueError(kid)


def dphi_over_rho(kid, a, b, rho):
    """psi(rho) = phi'(rho) / rho, with the rho = 0 term defined as its limit, or 0
    where the limit does not exist (the reference's gradient at a centre is the
    AD derivative, test/rbf_models.jl:99-115, for which that term vanishes)."""
    rho = np.asarray(rho, dtype=np.float64)
    if kid == 4:
        return -2.0 * a * a * np.exp(-(a * rho) ** 2)
    if kid == 2:
        return (-1.0) ** math.ceil(b) * 2.0 * a * a * b * (1.0 + (a * rho) ** 2) ** (b - 1.0)
    if kid == 1:
        return -2.0 * a * a * b * (1.0 + (a * rho) ** 2) ** (-b - 1.0)
    if kid == 0:
        sgn = (-1.0) ** math.ceil(a / 2.0)
        if a >= 2.0:
            return sgn * a * rho ** (a - 2.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            v = sgn * a * rho ** (a - 2.0)  # beta = 1: phi' (x-c)/rho has no limit at 0 -> 0
        return np.where(rho == 0.0, 0.0, v)
    if kid == 3:
        k = int(a)
        with np.errstate(divide="ignore", invalid="ignore"):
            v = (-1.0) ** (k + 1) * rho ** (2 * k - 2) * (2 * k * np.log(rho) + 1.0)
        return np.where(rho == 0.0, 0.0, v)
    raise ValueError(kid)


def poly_dim(d, deg):
    """dim of the polynomial space, deg in {-1, 0, 1} (RbfModel.jl:433 binomial(n+deg, n))."""
    if deg < 0:
        return 0
    if deg == 0:
        return 1
    if deg == 1:
        return d + 1
    raise ValueError("polynomial_degree must be <= 1 (RbfModel.jl:21)")


def poly_matrix(X, deg):
    """Pi (m x q): rows [1, x_1..x_d] truncated to the degree."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    m, d = X.shape
    q = poly_dim(d, deg)
    P = np.empty((m, q))
    if q >= 1:
        P[:, 0] = 1.0
    if q > 1:
        P[:, 1:] = X
    return P


def pairwise_dist(X, C, threads=None):
    """||x - c||_2 in DIFFERENCE form (what norm(x - c) does in the reference).
    Row chunks sized for the cache, spread over a thread pool (NumPy releases the GIL):
    this is the "best-effort CPU mode" of BASELINE.md section 2."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    X = np.atleast_2d(X)
    C = np.atleast_2d(C)
    out = np.empty((X.shape[0], C.shape[0]))
    step = max(1, int(2 ** 18 // max(1, C.shape[0] * C.shape[1])))

    def work(i):
        diff = X[i : i + step, None, :] - C[None, :, :]
        out[i : i + step] = np.sqrt(np.einsum("ijk,ijk->ij", diff, diff))

    starts = range(0, X.shape[0], step)
    nthreads = threads or min(len(starts), os.cpu_count() or 1)
    if nthreads <= 1 or X.shape[0] * C.shape[0] < 1 << 16:
        for i in starts:
            work(i)
    else:
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(work, starts))
    return out


def gram(C, kid, a, b, deg):
    """Phi (n x n), Pi (n x q) -- RBF.get_matrices (RbfModel.jl:374-375)."""
    C = np.asarray(C, dtype=np.float64)
    R = pairwise_dist(C, C)
    R = 0.5 * (R + R.T)
    np.fill_diagonal(R, 0.0)
    return phi(kid, a, b, R), poly_matrix(C, deg)


class OracleModel:
    """What RBF.RBFInterpolationModel holds: centres, weights (n x k), poly coeffs (q x k)."""

    def __init__(self, C, w, lam, kid, a, b, deg):
        self.C, self.w, self.lam = C, w, lam
        self.kid, self.a, self.b, self.deg = kid, a, b, deg
        self.num_outputs = w.shape[1]

    # ---- one point at a time: the reference's call pattern (RbfModel.jl:783-800)
    def value(self, x, ell=None):
        x = np.asarray(x, dtype=np.float64)
        rho = np.sqrt(((x[None, :] - self.C) ** 2).sum(axis=1))
        v = phi(self.kid, self.a, self.b, rho) @ self.w
        if self.lam.shape[0]:
            v = v + poly_matrix(x[None, :], self.deg)[0] @ self.lam
        return v if ell is None else v[ell]

    def grad(self, x, ell):
        return self.jac(x)[ell]

    def jac(self, x, rows=None):
        x = np.asarray(x, dtype=np.float64)
        diff = x[None, :] - self.C
        rho = np.sqrt((diff ** 2).sum(axis=1))
        psi = dphi_over_rho(self.kid, self.a, self.b, rho)
        J = (self.w * psi[:, None]).T @ diff  # k x d
        if self.lam.shape[0] > 1:
            J = J + self.lam[1:, :].T
        return J if rows is None else J[rows]

    # ---- batched (vectorised "best-effort CPU mode")
    def values(self, X):
        X = np.atleast_2d(X)
        V = phi(self.kid, self.a, self.b, pairwise_dist(X, self.C)) @ self.w
        if self.lam.shape[0]:
            V = V + poly_matrix(X, self.deg) @ self.lam
        return V

    def jacs(self, X):
        X = np.atleast_2d(X)
        m, d = X.shape
        k = self.num_outputs
        A = dphi_over_rho(self.kid, self.a, self.b, pairwise_dist(X, self.C))  # m x n
        J = np.empty((m, k, d))
        for l in range(k):
            Al = A * self.w[None, :, l]
            J[:, l, :] = Al.sum(axis=1)[:, None] * X - Al @ self.C
            if self.lam.shape[0] > 1:
                J[:, l, :] += self.lam[1:, l][None, :]
        return J


def saddle_matrix(Phi, Pi):
    n, q = Pi.shape
    S = np.zeros((n + q, n + q))
    S[:n, :n] = Phi
    S[:n, n:] = Pi
    S[n:, :n] = Pi.T
    return S


def fit(C, Y, kid, a, b, deg):
    """update_model's inner call (RbfModel.jl:759-763): assemble + dense LU solve."""
    import scipy.linalg

    C = np.asarray(C, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    n = C.shape[0]
    Phi, Pi = gram(C, kid, a, b, deg)
    q = Pi.shape[1]
    rhs = np.vstack([Y, np.zeros((q, Y.shape[1]))])
    if n < q:
        # fewer sites than tail terms (test/rbf_models.jl:35-44 builds such models): the saddle matrix is
        # singular; take the minimum-norm solution (what a pivoted-QR / pinv `\` returns)
        sol = np.linalg.lstsq(saddle_matrix(Phi, Pi), rhs, rcond=None)[0]
    else:
        sol = scipy.linalg.solve(saddle_matrix(Phi, Pi), rhs, assume_a="gen")
    return OracleModel(C, sol[:n].copy(), sol[n:].copy(), kid, a, b, deg)


def rel_residual(model, Y):
    """||Phi w + Pi lam - Y||_F / ||Y||_F at the training sites."""
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    return float(np.linalg.norm(model.values(model.C) - Y) / max(np.linalg.norm(Y), 1e-300))


# ---- descent.jl:137-185 restated for the batched-backtracking parity tests
def armijo_condition(strict, mx, mx_plus, step_size, omega, const_rhs):
    if strict:
        return bool(np.all((mx - mx_plus) >= step_size * const_rhs * omega))
    return bool(np.max(mx) - np.max(mx_plus) >= step_size * const_rhs * omega)


def backtrack(eval_objectives, x, direction, step_size, omega, strict=True,
              const_rhs=1e-6, shrink=0.75, min_stepsize=10 * np.finfo(float).eps,
              max_loops=None):
    """Sequential loop of _backtrack (descent.jl:150-185); returns (x_plus, mx_plus, step, i)."""
    if max_loops is None:
        max_loops = int(math.floor(math.log(min_stepsize) / math.log(shrink)))
    mx = eval_objectives(x)
    x_plus = x + step_size * direction
    mx_plus = eval_objectives(x_plus)
    i = 0
    while i < max_loops:
        if armijo_condition(strict, mx, mx_plus, step_size, omega, const_rhs):
            break
        if step_size <= min_stepsize:
            break
        step_size *= shrink
        x_plus = x + step_size * direction
        mx_plus = eval_objectives(x_plus)
        i += 1
    return x_plus, mx_plus, step_size * direction, i
