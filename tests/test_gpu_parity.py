"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path through the C ABI against the
CPU oracle, the committed golden fixtures, and size-independent properties at BASELINE.json sizes.

Tolerances (BASELINE.json north_star): 1e-10 relative on interpolation weights, 1e-8 on surrogate values.
Weights: asserted at 1e-10 outright against the fp64 oracle on every case with cond < 1e6 and on every medium / full-size
problem; on every case (any conditioning) the GPU solution must solve the oracle's saddle system with a normwise backward
error <= 50 eps; and every fixture case is MEASURED against an extended-precision truth (tests/golden/rbf_truth.npz, mpmath at
60 digits): GPU weights within 1e-10 of the true weights, or no further from them than twice the fp64 LAPACK LU is (C1 as
BASELINE.json writes it, cond 5.7e11, where no fp64 solver reaches 1e-10).  No case is justified by a perturbation bound.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from tests.conftest import ROOT, dist_from_truth, has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import morbit.jl_amd as pkg
    from morbit.jl_amd import _lib
from oracle import rbf_oracle as orc

EPS = np.finfo(np.float64).eps
REPORT = {}


@pytest.fixture(scope="module")
def ctx():
    c = pkg.Context()
    yield c
    c.close()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def cfg_of(case):
    name = orc.KERNEL_NAMES[case["kid"]]
    if case["kid"] in (1, 2) and case["b"] != 0.5:
        return None  # general exponents go through the raw ABI below
    return pkg.RbfConfig(kernel=name, shape_parameter=case["a"], polynomial_degree=case["deg"])


def raw_fit(ctx, C, Y, kid, a, b, deg):
    C = np.ascontiguousarray(C, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(C.shape[0], -1)
    n, d = C.shape
    k = Y.shape[1]
    q = orc.poly_dim(d, deg)
    W = np.empty((n, k))
    L = np.empty((max(q, 1), k))
    h = _lib.c_vp()
    info = _lib.FitInfo()
    rc = ctx.lib.mrbf_fit(ctx.h, n, d, k, _lib.as_ptr(C), _lib.as_ptr(Y), kid, a, b, deg, ctypes.byref(h),
                          _lib.as_ptr(W), _lib.as_ptr(L), ctypes.byref(info))
    return rc, pkg.RbfModel(ctx, h, n, d, k, q, False, W, L[:q], info.asdict()) if rc == 0 else None


def test_f64_mfma_lane_maps(ctx):
    # exact integer data, ASYMMETRIC B: pins A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D row=(l>>4)+4r col=l&15
    A = np.arange(64, dtype=np.float64).reshape(16, 4) - 7.0
    B = (np.arange(64, dtype=np.float64).reshape(4, 16) % 11) * 3.0 - 5.0
    D = np.empty((16, 16))
    ctx.check(ctx.lib.mrbf_debug_mfma_layout(ctx.h, _lib.as_ptr(D), _lib.as_ptr(A), _lib.as_ptr(B)))
    assert np.array_equal(D, A @ B)


@pytest.mark.parametrize("mode", [0, 1])
def test_gram_matches_golden(ctx, golden, mode):
    ctx.set_option(_lib.OPT_GRAM_MODE, mode)
    worst = 0.0
    try:
        for c in golden:
            n, d = c["C"].shape
            q = orc.poly_dim(d, c["deg"])
            Phi = np.empty((n, n), order="F")
            Pi = np.empty((n, max(q, 1)), order="F")
            ctx.check(ctx.lib.mrbf_gram(ctx.h, n, d, _lib.as_ptr(np.ascontiguousarray(c["C"])), c["kid"], c["a"], c["b"],
                                        c["deg"], _lib.as_ptr(Phi), _lib.as_ptr(Pi) if q else None, None))
            err = np.abs(Phi - c["Phi"]).max() / max(1.0, np.abs(c["Phi"]).max())
            worst = max(worst, err)
            assert err < 2e-13, (c["name"], mode, err)
            assert np.array_equal(Phi, Phi.T), c["name"]          # exactly symmetric
            assert np.all(np.diag(Phi) == orc.phi(c["kid"], c["a"], c["b"], 0.0)), c["name"]  # Phi[1,1] = phi(0), RbfModel.jl:398
            if q:
                assert np.array_equal(Pi[:, :q], c["Pi"]), c["name"]
    finally:
        ctx.set_option(_lib.OPT_GRAM_MODE, 0)
    REPORT["gram_worst_rel_err_mode%d" % mode] = worst


def _backward_error(c, W, Lam):
    """normwise backward error of [W; Lam] in the ORACLE's saddle system (Rigal-Gaches, Frobenius norms):
    ||S x - b|| / (||S|| ||x|| + ||b||) -- a conditioning-free measure: <= c * eps for any backward-stable solver."""
    S = orc.saddle_matrix(c["Phi"], c["Pi"])
    q = c["Pi"].shape[1]
    x = np.vstack([W, np.asarray(Lam).reshape(q, W.shape[1])])
    b = np.vstack([c["Y"].reshape(W.shape[0], -1), np.zeros((q, W.shape[1]))])
    return float(np.linalg.norm(S @ x - b) / (np.linalg.norm(S) * np.linalg.norm(x) + np.linalg.norm(b)))


W_TOL = 1e-10          # BASELINE.json north_star: interpolation weights, relative
WELL_CONDITIONED = 1e6  # below this condition number the 1e-10 bar is asserted outright (observed <= 1e-12 there)


def test_fit_and_eval_match_golden(ctx, golden):
    rows, exceeded = [], []
    for c in golden:
        rc, mod = raw_fit(ctx, c["C"], c["Y"], c["kid"], c["a"], c["b"], c["deg"])
        assert rc == 0, (c["name"], rc, ctx.lib.mrbf_last_error(ctx.h))
        V, J = mod.eval_sites(c["X"], want_values=True, want_jac=True)
        cond = c["cond"]
        ew = np.abs(mod.weights - c["W"]).max() / max(np.abs(c["W"]).max(), 1e-300)
        ev = np.abs(V - c["V"]).max() / max(1.0, np.abs(c["V"]).max())
        ej = np.abs(J - c["J"]).max() / max(1.0, np.abs(c["J"]).max())
        under = c["C"].shape[0] < c["Pi"].shape[1]
        be = _backward_error(c, mod.weights, mod.poly)
        be_oracle = _backward_error(c, c["W"], c["Lam"])
        rows.append(dict(name=c["name"], path=mod.info["path"], cond=cond, w=ew, v=ev, j=ej, res=mod.info["rel_residual"],
                         backward_error=be, backward_error_oracle=be_oracle))
        # (1) conditioning-free: the GPU solution solves the oracle's system to working precision
        assert be <= 50 * EPS, (c["name"], be)
        # (2) the north-star weight tolerance, outright, wherever the problem is well conditioned
        if cond < WELL_CONDITIONED and not under:
            assert ew < W_TOL, (c["name"], ew, cond)
        # (3) MEASURED against the extended-precision truth (tests/golden/make_truth.py: the same saddle system solved by
        # mpmath at 60 digits): the GPU weights are within 1e-10 of the true weights, or -- where fp64 itself cannot deliver
        # that -- no further from them than twice the reference-pattern solver (LAPACK LU of the saddle system) is.
        if "truth" in c:
            t = c["truth"]
            tw_gpu, tw_orc = dist_from_truth(mod.weights, t["W"]), dist_from_truth(c["W"], t["W"])
            tv = dist_from_truth(V, t["V"]) * np.abs(t["V"][0]).max() / max(1.0, np.abs(t["V"][0]).max())
            tj = dist_from_truth(J, t["J"]) * np.abs(t["J"][0]).max() / max(1.0, np.abs(t["J"][0]).max())
            rows[-1].update(w_vs_truth=tw_gpu, w_oracle_vs_truth=tw_orc, v_vs_truth=tv, j_vs_truth=tj)
            assert tw_gpu <= max(W_TOL, 2.0 * tw_orc), (c["name"], tw_gpu, tw_orc, cond)
            assert tv < 1e-8 and tj < 1e-8, (c["name"], tv, tj)
            if ew >= W_TOL:
                exceeded.append(dict(name=c["name"], weight_err_vs_oracle=ew, cond=cond, gpu_vs_truth=tw_gpu, oracle_vs_truth=tw_orc,
                                     values_vs_truth=tv, jac_vs_truth=tj, why=(
                    "cond %.1e: measured against the 60-digit solution the GPU weights are %.1e from the truth and the fp64 "
                    "LAPACK LU %.1e; the two fp64 solutions differ by %.1e from each other" % (cond, tw_gpu, tw_orc, ew))))
        else:
            assert under, c["name"]   # only minimum-norm (n < q) cases have no truth entry
        assert ev < 1e-8, (c["name"], ev, cond)
        assert ej < 1e-8, (c["name"], ej, cond)
        assert mod.info["rel_residual"] < 1e-11, (c["name"], mod.info)
        if mod.q:
            assert mod.info["max_pitw"] < 1e3 * EPS * max(1.0, np.abs(c["W"]).max()) * max(1.0, np.abs(c["Pi"]).max()) * c["C"].shape[0], \
                (c["name"], mod.info["max_pitw"])
        assert mod.info["fallbacks"] in (0, _lib.FB_LU), (c["name"], mod.info)
        mod.free()
    REPORT["golden"] = rows
    REPORT["weights_above_1e-10"] = exceeded
    paths = {r["path"] for r in rows}
    assert paths == {1, 2, 3}, paths  # all three solve paths are exercised by the grid
    # no name whitelist: every case above is decided by the measurement against the truth; what differs from the fp64 oracle by
    # more than 1e-10 is listed in the report with both distances.  BASELINE config C1 with a bounded-conditioning shape meets
    # 1e-10 outright
    c1b = next(r for r in rows if r["name"] == "c1_two_parabolas_shape_2_over_delta")
    assert c1b["w"] < W_TOL and c1b["cond"] < WELL_CONDITIONED


def test_eval_from_golden_coeffs_isolated_from_solve(ctx, golden):
    # model_from_coeffs + eval: pins the evaluation kernels alone at the 1e-8 value tolerance (observed ~1e-13)
    worst = 0.0
    for c in golden:
        cfg = cfg_of(c)
        if cfg is None:
            continue
        mod = pkg.model_from_coeffs(cfg, c["C"], c["W"], c["Lam"], ctx=ctx)
        V, J = mod.eval_sites(c["X"], want_values=True, want_jac=True)
        sw = max(1.0, np.abs(c["W"]).max())
        ev = np.abs(V - c["V"]).max() / max(1.0, np.abs(c["V"]).max())
        ej = np.abs(J - c["J"]).max() / max(1.0, np.abs(c["J"]).max())
        worst = max(worst, ev / sw, ej / sw)
        assert ev < 1e-12 * sw * c["C"].shape[0], (c["name"], ev)
        assert ej < 1e-11 * sw * c["C"].shape[0], (c["name"], ej)
        # single-site reference API: gradient == Jacobian row, bit for bit (test/rbf_models.jl:105-109)
        x = c["X"][1]
        Jx = pkg.get_jacobian(mod, None, x)
        for l in range(mod.num_outputs):
            assert np.array_equal(pkg.get_gradient(mod, None, x, l), Jx[l])
        assert np.allclose(pkg.eval_models(mod, None, x), V[1], rtol=0, atol=1e-12 * sw * max(1.0, np.abs(V).max()))
        assert np.array_equal(pkg.eval_models(mod, None, x, [mod.num_outputs - 1]), pkg.eval_models(mod, None, x)[-1:])
        mod.free()
    REPORT["eval_isolated_worst"] = worst


def _synthetic(n, d, k, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    C = rng.random((n, d))
    Y = np.stack([((C - 1.0) ** 2).sum(axis=1), ((C + 1.0) ** 2).sum(axis=1), np.sin(C.sum(axis=1))][:k], axis=1) / d
    return C, Y


@pytest.mark.parametrize("kernel,deg,n,d", [("gaussian", -1, 700, 32), ("gaussian", 1, 513, 32), ("multiquadric", 1, 640, 64),
                                           ("cubic", 1, 389, 17), ("inv_multiquadric", 0, 300, 7),
                                           ("thin_plate_spline", 1, 200, 5), ("cubic", -1, 257, 9),
                                           ("cubic", 1, 600, 200), ("multiquadric", 1, 500, 130), ("gaussian", 0, 300, 140),
                                           ("cubic", 1, 700, 100), ("multiquadric", 1, 1029, 128)])
def test_medium_sizes_against_oracle(ctx, kernel, deg, n, d):
    # ragged n (not a multiple of 64/128, odd), d not a multiple of 16
    C, Y = _synthetic(n, d, 2, seed=n + d)
    X = np.random.Generator(np.random.PCG64(5)).random((33, d))
    cfg = pkg.RbfConfig(kernel=kernel, polynomial_degree=deg)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    ref = orc.fit(C, Y, kid, a, b, deg)
    mod = pkg.update_model(cfg, C, Y, ctx=ctx)
    V, J = mod.eval_sites(X, want_values=True, want_jac=True)
    ew = np.abs(mod.weights - ref.w).max() / np.abs(ref.w).max()
    ev = np.abs(V - ref.values(X)).max() / max(1.0, np.abs(V).max())
    ej = np.abs(J - ref.jacs(X)).max() / max(1.0, np.abs(J).max())
    key = "medium_%s_deg%d_n%d_d%d" % (kernel, deg, n, d)
    REPORT[key] = dict(path=mod.info["path"], w=ew, v=ev, j=ej, res=mod.info["rel_residual"], mu=mod.info["mu"])
    assert ev < 1e-8 and ej < 1e-8, (ev, ej)
    # weights: conditioning-free check -- the GPU weights solve the oracle's saddle system to working precision
    Phi, Pi = orc.gram(C, kid, a, b, deg)
    case = dict(Phi=Phi, Pi=Pi, Y=Y)
    be = _backward_error(case, mod.weights, mod.poly)
    be_o = _backward_error(case, ref.w, ref.lam)
    cond = float(np.linalg.cond(orc.saddle_matrix(Phi, Pi)))
    REPORT[key].update(cond=cond, backward_error=be, backward_error_oracle=be_o)
    assert be <= 50 * EPS, be
    xg, xo = np.vstack([mod.weights, mod.poly]), np.vstack([ref.w, ref.lam])
    assert ew < W_TOL, (ew, cond, be, be_o)   # outright at every conditioning of this list (cond up to 9e7: observed <= 7e-12)
    assert mod.info["rel_residual"] < 1e-10
    mod.free()


@pytest.mark.parametrize("kernel,deg,n,d,k", [("gaussian", -1, 1300, 10, 5), ("multiquadric", 1, 1100, 24, 9), ("inv_multiquadric", 0, 2100, 6, 3),
                                             ("gaussian", 1, 777, 40, 1), ("multiquadric", 1, 1536, 64, 4)])
def test_persistent_factor_and_backsolve_many_outputs(ctx, kernel, deg, n, d, k):
    # n >= 512 goes through the one-launch factorisation; k > 4 needs two passes of the persistent backward substitution
    rng = np.random.Generator(np.random.PCG64(n + 7 * k))
    C = rng.random((n, d))
    Y = np.stack([np.sin((j + 1) * C.sum(axis=1) / d) + 0.1 * j for j in range(k)], axis=1)
    X = rng.random((21, d))
    cfg = pkg.RbfConfig(kernel=kernel, polynomial_degree=deg)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    ref = orc.fit(C, Y, kid, a, b, deg)
    mod = pkg.update_model(cfg, C, Y, ctx=ctx)
    V, J = mod.eval_sites(X, want_values=True, want_jac=True)
    assert mod.info["path"] in (_lib.PATH_CHOL, _lib.PATH_PROJ_CHOL)
    assert mod.info["rel_residual"] < 1e-9
    assert np.abs(V - ref.values(X)).max() / max(1.0, np.abs(V).max()) < 1e-8
    assert np.abs(J - ref.jacs(X)).max() / max(1.0, np.abs(J).max()) < 1e-8
    # every output solved with the same factor: column-wise agreement with a single-output fit
    m1 = pkg.update_model(cfg, C, Y[:, k - 1:k], ctx=ctx)
    assert np.abs(m1.weights[:, 0] - mod.weights[:, k - 1]).max() <= 1e-9 * max(1.0, np.abs(mod.weights).max())
    m1.free()
    mod.free()


def test_solve_paths_agree(ctx):
    # the same problem through projected Cholesky and through LU on the saddle system
    C, Y = _synthetic(500, 12, 2, seed=11)
    cfg = pkg.RbfConfig(kernel="multiquadric", polynomial_degree=1)
    m1 = pkg.update_model(cfg, C, Y, ctx=ctx)
    ctx.set_option(_lib.OPT_FORCE_PATH, _lib.PATH_LU)
    try:
        m2 = pkg.update_model(cfg, C, Y, ctx=ctx)
    finally:
        ctx.set_option(_lib.OPT_FORCE_PATH, 0)
    assert m1.info["path"] == _lib.PATH_PROJ_CHOL and m2.info["path"] == _lib.PATH_LU
    assert np.abs(m1.weights - m2.weights).max() / np.abs(m2.weights).max() < 1e-9
    assert np.abs(m1.poly - m2.poly).max() / max(1.0, np.abs(m2.poly).max()) < 1e-9
    # a Cholesky forced onto an indefinite system reports MRBF_ENOTPD instead of garbage
    ctx.set_option(_lib.OPT_FORCE_PATH, _lib.PATH_CHOL)
    try:
        with pytest.raises(pkg.MrbfError) as ei:
            pkg.update_model(pkg.RbfConfig(kernel="cubic", polynomial_degree=-1), C, Y, ctx=ctx)
        assert ei.value.code == _lib.MRBF_ENOTPD
    finally:
        ctx.set_option(_lib.OPT_FORCE_PATH, 0)
    m1.free()
    m2.free()


def test_error_codes_and_edge_cases(ctx):
    C, Y = _synthetic(40, 3, 1, seed=3)
    h = _lib.c_vp()
    f = ctx.lib.mrbf_fit
    Cp, Yp = _lib.as_ptr(C), _lib.as_ptr(Y)
    assert f(ctx.h, 0, 3, 1, Cp, Yp, 4, 1.0, 0.0, 1, ctypes.byref(h), None, None, None) == -2     # n
    assert f(ctx.h, 40, 0, 1, Cp, Yp, 4, 1.0, 0.0, 1, ctypes.byref(h), None, None, None) == -3    # d
    assert f(ctx.h, 40, 3, 1, None, Yp, 4, 1.0, 0.0, 1, ctypes.byref(h), None, None, None) == -5  # centres
    assert f(ctx.h, 40, 3, 1, Cp, Yp, 9, 1.0, 0.0, 1, ctypes.byref(h), None, None, None) == -7    # kernel id
    assert f(ctx.h, 40, 3, 1, Cp, Yp, 0, 4.0, 0.0, 1, ctypes.byref(h), None, None, None) == -8    # even cubic exponent
    assert f(ctx.h, 40, 3, 1, Cp, Yp, 4, 1.0, 0.0, 2, ctypes.byref(h), None, None, None) == -10   # degree 2
    assert b"polynomial_degree" in ctx.lib.mrbf_last_error(ctx.h)
    # fewer sites than polynomial terms (max_model_points = 1 in test/rbf_models.jl:35-40): singular saddle system
    rc, mm = raw_fit(ctx, C[:2], Y[:2], 4, 1.0, 0.0, 1)
    assert rc == 0 and mm.info["path"] == _lib.PATH_MINNORM
    ref = orc.fit(C[:2], Y[:2], 4, 1.0, 0.0, 1)          # minimum-norm least squares in the oracle too
    assert np.allclose(mm.weights, ref.w, atol=1e-12) and np.allclose(mm.poly, ref.lam, atol=1e-12)
    assert np.allclose(pkg.eval_models_at_sites(mm, None, C[:2]), Y[:2], atol=1e-12)  # still interpolates
    mm.free()
    # exactly singular square system (duplicate site): MRBF_ESINGULAR, like Julia's `\` throwing SingularException
    Cd = np.vstack([C[:10], C[:1]])
    # (cubic, no tail -> LU path; two identical rows of the saddle matrix give an exactly zero pivot in partial pivoting)
    Yd = np.vstack([Y[:10], Y[:1]])
    rc, md = raw_fit(ctx, Cd, Yd, 0, 3.0, 0.0, -1)
    if rc != 0:   # exactly zero pivot: the documented error code and message, no model handed out
        assert rc == _lib.MRBF_ESINGULAR and md is None and b"singular" in ctx.lib.mrbf_last_error(ctx.h)
    else:         # rounding left a tiny pivot (Julia's `\` does not throw either then): the consistent system must be solved
        assert md.info["path"] == _lib.PATH_LU and np.isfinite(md.weights).all()
        assert np.abs(pkg.eval_models_at_sites(md, None, Cd) - Yd).max() < 1e-6 * max(1.0, np.abs(Yd).max())
        md.free()
    # one site, no tail: the 1 x 1 system
    rc, mod = raw_fit(ctx, C[:1], Y[:1], 4, 1.0, 0.0, -1)
    assert rc == 0 and abs(mod.weights[0, 0] - Y[0, 0]) < 1e-15
    # empty query batch
    assert ctx.lib.mrbf_eval(ctx.h, mod.model, 0, None, None, None, None) == 0
    mod.free()
    # empty Gram
    assert ctx.lib.mrbf_gram(ctx.h, 0, 3, None, 4, 1.0, 0.0, 1, _lib.as_ptr(np.empty(1)), None, None) == 0


def test_backtracking_matches_sequential_reference_loop(ctx):
    C, Y = _synthetic(120, 4, 2, seed=21)
    cfg = pkg.RbfConfig(kernel="cubic", polynomial_degree=1)
    mod = pkg.update_model(cfg, C, Y, ctx=ctx)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    ref = orc.OracleModel(C, mod.weights, mod.poly, kid, a, b, 1)
    sc = pkg.surrogates.SurrogateContainer(objectives=[pkg.surrogates.RefSurrogate(mod, [0]),
                                                       pkg.surrogates.RefSurrogate(mod, [1])])
    dcfg = pkg.descent.SteepestDescentConfig()
    assert dcfg.max_loops == 117  # floor(log(10 eps)/log(.75)), descent.jl:61-66
    x = np.full(4, 0.5)
    Jx = pkg.surrogates.eval_container_objectives_jacobian_at_scaled_site(sc, None, x)
    for direction, step0 in ((-Jx.sum(axis=0) / np.linalg.norm(Jx.sum(axis=0)), 8.0), (np.ones(4) / 2.0, 1.0)):
        for strict in (True, False):
            dcfg.strict_backtracking = strict
            xp, mxp, step, i = pkg.descent._backtrack(x, direction, step0, 0.3, sc, dcfg)
            rxp, rmxp, rstep, ri = orc.backtrack(ref.value, x, direction, step0, 0.3, strict=strict)
            if ri > 90:  # no descent: both loops end in the rounding-noise floor of mx - mx_plus
                assert i > 90
                continue
            assert i == ri, (i, ri)
            assert np.array_equal(xp, rxp) and np.array_equal(step, rstep)
            assert np.allclose(mxp, rmxp, rtol=1e-10, atol=1e-12)
    # the container with two differently grouped models takes the general batched route
    mod2 = pkg.update_model(pkg.RbfConfig(kernel="gaussian"), C, Y[:, :1], ctx=ctx)
    sc2 = pkg.surrogates.SurrogateContainer(objectives=[pkg.surrogates.RefSurrogate(mod, [1]), pkg.surrogates.RefSurrogate(mod2, [0])])
    f2 = lambda z: np.concatenate([pkg.eval_models(mod, None, z, [1]), pkg.eval_models(mod2, None, z)])
    xp, mxp, step, i = pkg.descent._backtrack(x, np.ones(4) / 2.0, 1.0, 0.3, sc2, dcfg)
    rxp, rmxp, rstep, ri = orc.backtrack(f2, x, np.ones(4) / 2.0, 1.0, 0.3, strict=False)
    assert (i == ri and np.array_equal(xp, rxp)) or (i > 90 and ri > 90)
    # container Jacobian rows == per-output gradients (test/rbf_models.jl:164-168 analogue)
    J2 = pkg.surrogates.eval_container_objectives_jacobian_at_scaled_site(sc2, None, x)
    assert np.array_equal(J2[0], pkg.get_gradient(mod, None, x, 1)) and np.array_equal(J2[1], pkg.get_gradient(mod2, None, x, 0))
    mod.free()
    mod2.free()


def test_full_size_properties_c2_c3(ctx):
    """BASELINE.json configs[1] and configs[2] at full size through size-independent properties:
    interpolation at the training sites (residual), Pi'w = 0, symmetry / diagonal of Phi, device-pointer I/O."""
    import torch

    out = {}
    for name, (kernel, n, d, k, m) in {"C2": ("gaussian", 2048, 32, 1, 256), "C3": ("multiquadric", 8192, 64, 2, 2048)}.items():
        C, Y = _synthetic(n, d, k, seed=2 if name == "C2" else 3)
        cfg = pkg.RbfConfig(kernel=kernel, polynomial_degree=1)
        mod = pkg.update_model(cfg, C, Y, ctx=ctx)
        out[name] = dict(mod.info)
        assert mod.info["path"] == _lib.PATH_PROJ_CHOL
        assert mod.info["rel_residual"] < 1e-10, mod.info
        assert mod.info["max_pitw"] < 1e-8 * max(1.0, np.abs(mod.weights).max())
        if name == "C2":
            # C2 WITH its degree-1 tail against the oracle's LAPACK LU of the 2081 x 2081 saddle system (C3's and C5's
            # counterparts: test_c3_bench_workload_eval_against_oracle, test_c5_full_size): north-star tolerances outright
            kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
            ref = orc.fit(C, Y, kid, a, b, 1)
            Phi, Pi = orc.gram(C, kid, a, b, 1)
            case = dict(Phi=Phi, Pi=Pi, Y=Y)
            be, be_o = _backward_error(case, mod.weights, mod.poly), _backward_error(case, ref.w, ref.lam)
            ew = np.abs(mod.weights - ref.w).max() / np.abs(ref.w).max()
            Xo = np.random.Generator(np.random.PCG64(6)).random((128, d))
            Vo, Jo = mod.eval_sites(Xo, want_values=True, want_jac=True)
            evo = np.abs(Vo - ref.values(Xo)).max() / max(1.0, np.abs(Vo).max())
            ejo = np.abs(Jo - ref.jacs(Xo)).max() / max(1.0, np.abs(Jo).max())
            out[name].update(weights_vs_oracle_lu=ew, backward_error=be, backward_error_oracle=be_o, values_vs_oracle=evo, jac_vs_oracle=ejo)
            assert be <= 50 * EPS, be
            assert ew < W_TOL, (ew, be, be_o)
            assert evo < 1e-8 and ejo < 1e-8, (evo, ejo)
        # values at a subset of training sites reproduce the data to 1e-8 (the north-star value tolerance)
        idx = np.arange(0, n, max(1, n // 97))
        V = pkg.eval_models_at_sites(mod, None, C[idx])
        assert np.abs(V - Y[idx]).max() < 1e-8 * max(1.0, np.abs(Y).max())
        # device-resident inputs/outputs (torch tensors) give the same bits as host staging
        Xh = np.random.Generator(np.random.PCG64(4)).random((m, d))
        Vh, Jh = mod.eval_sites(Xh, want_values=True, want_jac=True)
        Xd = torch.from_numpy(Xh).cuda()
        Vd = torch.empty((m, k), dtype=torch.float64, device="cuda")
        Jd = torch.empty((m, d, k), dtype=torch.float64, device="cuda")
        mod.eval_sites(Xd, want_values=True, want_jac=True, out_vals=Vd, out_jac=Jd)
        torch.cuda.synchronize()
        assert np.array_equal(Vd.cpu().numpy(), Vh)
        assert np.array_equal(np.transpose(Jd.cpu().numpy(), (0, 2, 1)), Jh)
        # Jacobian vs central finite differences of the device values at a few points
        h = 1e-5
        for p in range(3):
            E = np.eye(d) * h
            Vp = pkg.eval_models_at_sites(mod, None, Xh[p][None, :] + E)
            Vm = pkg.eval_models_at_sites(mod, None, Xh[p][None, :] - E)
            fd = ((Vp - Vm) / (2 * h)).T
            assert np.abs(fd - Jh[p]).max() < 1e-6 * max(1.0, np.abs(Jh[p]).max(), np.abs(mod.weights).max() * 1e-3)
        mod.free()
    # Gram symmetry at a ragged full-ish size on both kernels
    Cg = np.random.Generator(np.random.PCG64(9)).random((1000, 64))
    P0, _, _ = pkg.get_matrices(pkg.RbfConfig(kernel="multiquadric"), Cg, ctx=ctx, want_pi=False)
    ctx.set_option(_lib.OPT_GRAM_MODE, 1)
    P1, _, _ = pkg.get_matrices(pkg.RbfConfig(kernel="multiquadric"), Cg, ctx=ctx, want_pi=False)
    ctx.set_option(_lib.OPT_GRAM_MODE, 0)
    assert np.array_equal(P0, P0.T) and np.array_equal(P1, P1.T)
    assert np.abs(P0 - P1).max() < 1e-13 * np.abs(P1).max()
    REPORT["full_size"] = out


def test_c3_bench_workload_eval_against_oracle(ctx):
    """The workload bench.py times (workloads.problem("C3"): n = 8192, d = 64, k = 2, multiquadric + degree-1 tail, m = 10 000 query
    points -- the launch shape with the centre range split over gridDim.y and the combine kernel): the fit's interpolation residual,
    and values / Jacobians of a strided subset of the 10 000-point launch against the oracle's evaluation of the SAME weights (this
    isolates the evaluation at the shape that is timed; the weights themselves are checked through the residual and Pi'w)."""
    from morbit.jl_amd import workloads as wl

    C, Y, X = wl.problem("C3")
    assert C.shape == (8192, 64) and X.shape == (10000, 64) and Y.shape[1] == 2
    cfg = pkg.RbfConfig(kernel="multiquadric", polynomial_degree=1)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    mod = pkg.update_model(cfg, C, Y, ctx=ctx)
    assert mod.info["path"] == _lib.PATH_PROJ_CHOL and mod.info["fallbacks"] == 0
    assert mod.info["rel_residual"] < 1e-10 and mod.info["max_pitw"] < 1e-8 * max(1.0, np.abs(mod.weights).max()), mod.info
    V, J = mod.eval_sites(X, want_values=True, want_jac=True)          # the bench's launch: all 10 000 points, values + Jacobians
    sub = np.arange(0, 10000, 39)[:256]
    ref = orc.OracleModel(C, mod.weights, mod.poly, kid, a, b, 1)       # the oracle's formulas on the GPU's coefficients
    ev = np.abs(V[sub] - ref.values(X[sub])).max() / max(1.0, np.abs(V).max())
    ej = np.abs(J[sub] - ref.jacs(X[sub])).max() / max(1.0, np.abs(J).max())
    assert ev < 1e-8 and ej < 1e-8, (ev, ej)
    # the same points alone (another launch shape: one query tile, other split) give the same numbers to rounding
    V2, J2 = mod.eval_sites(X[sub], want_values=True, want_jac=True)
    assert np.abs(V2 - V[sub]).max() < 1e-12 * max(1.0, np.abs(V).max()) and np.abs(J2 - J[sub]).max() < 1e-11 * max(1.0, np.abs(J).max())
    # the weights of the headline workload against the oracle: per-pair norm(x - c) assembly (threaded C restatement, the same
    # arithmetic per entry as oracle.gram) + LAPACK LU of the 8257 x 8257 saddle system -- the same asserts as the C5 test
    import os
    import time

    import scipy.linalg
    from oracle import c_oracle

    n = C.shape[0]
    threads = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    Phi, Pi = c_oracle.gram(C, kid, a, b, 1, threads=threads)
    S = orc.saddle_matrix(Phi, Pi)
    rhs = np.vstack([Y, np.zeros((Pi.shape[1], Y.shape[1]))])
    nrm_S = float(np.linalg.norm(S))
    sol = scipy.linalg.solve(S, rhs, assume_a="gen", overwrite_a=False, check_finite=False)
    t_oracle = time.perf_counter() - t0
    xg = np.vstack([mod.weights, mod.poly])
    be = float(np.linalg.norm(S @ xg - rhs) / (nrm_S * np.linalg.norm(xg) + np.linalg.norm(rhs)))
    be_o = float(np.linalg.norm(S @ sol - rhs) / (nrm_S * np.linalg.norm(sol) + np.linalg.norm(rhs)))
    ew = np.abs(mod.weights - sol[:n]).max() / np.abs(sol[:n]).max()
    ref_o = orc.OracleModel(C, sol[:n].copy(), sol[n:].copy(), kid, a, b, 1)
    ev_o = np.abs(V[sub] - ref_o.values(X[sub])).max() / max(1.0, np.abs(V).max())
    ej_o = np.abs(J[sub[:32]] - ref_o.jacs(X[sub[:32]])).max() / max(1.0, np.abs(J).max())
    print("C3: oracle LU %.1f s on %d threads; weights %.2e backward error %.2e (oracle %.2e) values %.2e jac %.2e"
          % (t_oracle, threads, ew, be, be_o, ev_o, ej_o))
    assert be <= 50 * EPS, be
    assert ev_o < 1e-8 and ej_o < 1e-8, (ev_o, ej_o)
    assert ew < W_TOL, (ew, be, be_o)          # outright (observed 4e-12)
    REPORT["c3_bench_workload"] = dict(values=ev, jac=ej, rel_residual=mod.info["rel_residual"], weights_vs_oracle_lu=ew,
                                       backward_error=be, values_vs_oracle=ev_o, jac_vs_oracle=ej_o)
    mod.free()


def test_c2_full_size_without_tail(ctx):
    """the C2 variant without a polynomial tail (degree -1): the pure Cholesky path (path 1) at n = 2048, d = 32
    through the one-launch factorisation, against the oracle's dense solve."""
    from morbit.jl_amd import workloads as wl

    C, Y, _ = wl.problem("C2")
    assert C.shape == (2048, 32)
    cfg = pkg.RbfConfig(kernel="gaussian", polynomial_degree=-1)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    mod = pkg.update_model(cfg, C, Y, ctx=ctx)
    assert mod.info["path"] == _lib.PATH_CHOL and mod.info["fallbacks"] == 0 and mod.info["q"] == 0
    assert mod.info["rel_residual"] < 1e-9, mod.info
    ref = orc.fit(C, Y, kid, a, b, -1)
    Phi, Pi = orc.gram(C, kid, a, b, -1)
    case = dict(Phi=Phi, Pi=Pi, Y=Y)
    be, be_o = _backward_error(case, mod.weights, mod.poly), _backward_error(case, ref.w, ref.lam)
    cond = float(np.linalg.cond(Phi))
    assert be <= 50 * EPS, be
    ew = np.abs(mod.weights - ref.w).max() / np.abs(ref.w).max()
    assert ew < W_TOL, (ew, cond, be, be_o)
    X = np.random.Generator(np.random.PCG64(6)).random((200, 32))
    V, J = mod.eval_sites(X, want_values=True, want_jac=True)
    assert np.abs(V - ref.values(X)).max() < 1e-8 * max(1.0, np.abs(V).max(), cond * EPS * 1e8)
    refg = orc.OracleModel(C, mod.weights, mod.poly, kid, a, b, -1)
    assert np.abs(V - refg.values(X)).max() < 1e-8 * max(1.0, np.abs(V).max()) and np.abs(J - refg.jacs(X)).max() < 1e-8 * max(1.0, np.abs(J).max())
    REPORT["c2_deg-1_full_size"] = dict(cond=cond, backward_error=be, weights=ew, rel_residual=mod.info["rel_residual"])
    mod.free()


def test_batch_run_matches_single_calls(ctx):
    probs, keep = [], []
    P = 5
    arr = (_lib.Problem * P)()
    res = (_lib.Result * P)()
    for p in range(P):
        C, Y = _synthetic(150 + 10 * p, 6, 2, seed=100 + p)
        X = np.random.Generator(np.random.PCG64(p)).random((20, 6))
        W = np.empty_like(Y)
        V = np.empty((20, 2))
        keep.append((C, Y, X, W, V))
        arr[p] = _lib.Problem(C.shape[0], 20, 6, 2, 0, 1, 3.0, 0.0, C.ctypes.data_as(_lib.c_dp), Y.ctypes.data_as(_lib.c_dp),
                              X.ctypes.data_as(_lib.c_dp), W.ctypes.data_as(_lib.c_dp), None, V.ctypes.data_as(_lib.c_dp), None)
    assert ctx.lib.mrbf_batch_run(1, None, P, arr, res) == 0
    for p in range(P):
        C, Y, X, W, V = keep[p]
        assert res[p].status == 0 and res[p].fit.path == _lib.PATH_PROJ_CHOL
        mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y, ctx=ctx)
        assert np.array_equal(mod.weights, W)
        assert np.array_equal(pkg.eval_models_at_sites(mod, None, X), V)
        assert abs(res[p].checksum_w - W.sum()) < 1e-9 * max(1.0, np.abs(W).sum())
        mod.free()


def test_batch_run_mixed_shapes_matches_single_calls(ctx):
    """one mrbf_batch_run over problems of different kernels, tails, dimensions (padded to 64 and to 128), output counts, with and
    without Jacobians, one without queries, small ones (one-launch fit, evaluation groups that are NOT contiguous in the descriptor
    array) mixed with one beyond the small shape (per-problem chain): every result bit for bit the single call's"""
    specs = [  # kernel, deg, n, d, k, m, want_jac
        ("cubic", 1, 150, 6, 2, 20, True), ("multiquadric", 1, 257, 100, 2, 70, True), ("cubic", 1, 90, 6, 1, 33, False),
        ("gaussian", -1, 200, 12, 3, 40, True), ("multiquadric", 1, 300, 100, 2, 64, False), ("cubic", 1, 160, 6, 2, 0, False),
        ("inv_multiquadric", 0, 120, 70, 2, 25, True), ("cubic", 1, 700, 10, 2, 30, True), ("cubic", 1, 140, 6, 2, 50, True),
        # few query points against 4 .. 8 centre tiles split the centre range (end of round 5); 150 / 140 sites (3 tiles) do not: the same
        # (kernel, dimension, outputs, Jacobians) group then holds both kinds and is launched in two parts
        ("cubic", 1, 300, 6, 2, 40, True), ("cubic", 1, 500, 6, 2, 12, True), ("multiquadric", 1, 260, 100, 2, 9, True)]
    P = len(specs)
    arr = (_lib.Problem * P)()
    res = (_lib.Result * P)()
    keep = []
    dp = lambda a: a.ctypes.data_as(_lib.c_dp) if a is not None else None
    for p, (kernel, deg, n, d, k, m, wj) in enumerate(specs):
        C, Y = _synthetic(n, d, k, seed=500 + p)
        X = np.random.Generator(np.random.PCG64(600 + p)).random((max(m, 1), d))
        cfg = pkg.RbfConfig(kernel=kernel, polynomial_degree=deg)
        kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
        W = np.empty((n, k))
        V = np.empty((max(m, 1), k)) if m > 0 else None
        J = np.empty((max(m, 1), d, k)) if (m > 0 and wj) else None
        keep.append((cfg, C, Y, X, W, V, J))
        arr[p] = _lib.Problem(n, m, d, k, kid, deg, a, b, dp(C), dp(Y), dp(X) if m > 0 else None, dp(W), None, dp(V), dp(J))
    assert ctx.lib.mrbf_batch_run(1, None, P, arr, res) == 0
    for p, (kernel, deg, n, d, k, m, wj) in enumerate(specs):
        cfg, C, Y, X, W, V, J = keep[p]
        assert res[p].status == 0, (p, res[p].status)
        mod = pkg.update_model(cfg, C, Y, ctx=ctx)
        assert np.array_equal(mod.weights, W), p
        if m > 0:
            Vs, Js = mod.eval_sites(X[:m], want_values=True, want_jac=wj)
            assert np.array_equal(Vs, V[:m]), p
            if wj:
                assert np.array_equal(Js, np.transpose(J[:m], (0, 2, 1))), p
        mod.free()


@pytest.mark.parametrize("n", [128, 256, 300, 1024, 1537, 2200, 2700, 3500])
def test_builtin_cholesky_matches_lapack(ctx, n):
    rng = np.random.Generator(np.random.PCG64(n))
    G = rng.standard_normal((n, n + 20))
    A = G @ G.T / n + np.eye(n)
    Lref = np.linalg.cholesky(A)
    for impl in (1, 2, 3):
        F = np.asfortranarray(A.copy())
        info, ms = ctypes.c_int32(-7), ctypes.c_float()
        ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(F), impl, ctypes.byref(info), ctypes.byref(ms)))
        assert info.value == 0
        L = np.tril(F)
        assert np.abs(L - Lref).max() < 1e-12 * np.abs(Lref).max(), (impl, n, np.abs(L - Lref).max())
        assert np.array_equal(np.triu(F, 1), np.triu(A, 1))  # strictly upper triangle is not referenced
    # not positive definite: both implementations report the first bad pivot (LAPACK convention), no NaN games
    B = A.copy()
    bad = min(n - 1, 200)
    B[bad, bad] = -1.0
    for impl in (1, 2, 3):
        F = np.asfortranarray(B.copy())
        info = ctypes.c_int32(0)
        ctx.check(ctx.lib.mrbf_debug_potrf(ctx.h, n, _lib.as_ptr(F), impl, ctypes.byref(info), None))
        assert info.value == bad + 1, (impl, info.value)


@pytest.mark.parametrize("n", [900, 1300, 1800, 2300, 2900])
def test_fit_same_weights_with_all_cholesky_implementations(ctx, n):
    C, Y = _synthetic(n, 20, 2, seed=77)
    cfg = pkg.RbfConfig(kernel="multiquadric")
    out = []
    for impl in (1, 2, 3):
        ctx.set_option(_lib.OPT_CHOL_IMPL, impl)
        try:
            m = pkg.update_model(cfg, C, Y, ctx=ctx)
        finally:
            ctx.set_option(_lib.OPT_CHOL_IMPL, 0)
        assert m.info["path"] == _lib.PATH_PROJ_CHOL and m.info["rel_residual"] < 1e-12
        out.append(m.weights.copy())
        m.free()
    assert np.abs(out[0] - out[1]).max() < 1e-11 * np.abs(out[0]).max()
    assert np.abs(out[0] - out[2]).max() < 1e-11 * np.abs(out[0]).max()
