"""GPU tests of the BASELINE.json configurations that take special code paths (run with -m gpu):

C4  64 ZDT1 starts, d = 128, n = 257, cubic + degree-1 tail, m = 6450 through mrbf_batch_run: host-driven blocked Cholesky
    (n < 512), the dpad = 128 instance of the fused evaluation kernel, 4 worker contexts on one GPU.
C5  d = 256, n = 16384, cubic + degree-1 tail (q = 257): wide-tail projection (symm_panel column groups), 128 block columns in
    windows of 8 in the persistent factorisation, the GEMM evaluation pipeline (d > 128), the rocBLAS trsm branch of the tail.
plus a batch of problems with n in [512, 2048]: several persistent factorisations / backward substitutions from different
contexts resident on one GPU at the same time (what mrbf_batch_run does for n <= 2048).

Full-size checks are size-independent properties (interpolation residual, Pi'w = 0, Jacobian vs finite differences, batch ==
single calls, device == host buffers); one C4 start and the C5 problem are also compared with the CPU oracle.
"""
import ctypes
import os
import time

import numpy as np
import pytest

from tests.conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import morbit.jl_amd as pkg
    from morbit.jl_amd import _lib
from morbit.jl_amd import workloads as wl
from oracle import c_oracle
from oracle import rbf_oracle as orc

EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def ctx():
    c = pkg.Context()
    yield c
    c.close()


def _dp(a):
    return a.ctypes.data_as(_lib.c_dp) if a is not None else None


def _batch(problems, kid, a, b, deg, want_jac=()):
    """run (C, Y, X) problems through mrbf_batch_run on GPU 0; returns (results, W list, V list, J dict)"""
    P = len(problems)
    arr = (_lib.Problem * P)()
    res = (_lib.Result * P)()
    Ws, Ls, Vs, Js = [], [], [], {}
    for p, (C, Y, X) in enumerate(problems):
        n, d = C.shape
        k = Y.shape[1]
        q = orc.poly_dim(d, deg)
        W, L, V = np.empty((n, k)), np.empty((max(q, 1), k)), np.empty((X.shape[0], k))
        J = np.empty((X.shape[0], d, k)) if p in want_jac else None
        Ws.append(W), Ls.append(L), Vs.append(V)
        if J is not None:
            Js[p] = J
        arr[p] = _lib.Problem(n, X.shape[0], d, k, kid, deg, a, b, _dp(C), _dp(Y), _dp(X), _dp(W), _dp(L), _dp(V), _dp(J))
    rc = _lib.load().mrbf_batch_run(1, None, P, arr, res)
    assert rc == 0, rc
    return res, Ws, Ls, Vs, {p: np.transpose(J, (0, 2, 1)) for p, J in Js.items()}


def _fd_jacobian(mod, x, h=1e-6):
    d = x.size
    E = np.eye(d) * h
    Vp = pkg.eval_models_at_sites(mod, None, x[None, :] + E)
    Vm = pkg.eval_models_at_sites(mod, None, x[None, :] - E)
    return ((Vp - Vm) / (2 * h)).T


def test_c4_many_start_batch(ctx):
    """BASELINE.json configs[3]: all 64 ZDT1 starts in one mrbf_batch_run call"""
    cfgw = wl.CONFIGS["C4"]
    P = cfgw["problems"]
    problems = [wl.problem("C4", p) for p in range(P)]
    cfg = pkg.RbfConfig(kernel="cubic", polynomial_degree=1)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    t0 = time.perf_counter()
    res, Ws, Ls, Vs, Js = _batch(problems, kid, a, b, 1, want_jac=(0, 17))
    wall = time.perf_counter() - t0
    for p in range(P):
        C, Y, X = problems[p]
        r = res[p]
        assert r.status == 0, (p, r.status)
        assert r.fit.path == _lib.PATH_PROJ_CHOL and r.fit.fallbacks == 0, (p, r.fit.path, r.fit.fallbacks)
        assert r.fit.n == 257 and r.fit.q == 129
        assert r.fit.rel_residual < 1e-11, (p, r.fit.rel_residual)
        assert r.fit.max_pitw < 1e-11 * max(1.0, np.abs(Ws[p]).max()), (p, r.fit.max_pitw)
        assert np.isfinite(Vs[p]).all() and abs(r.checksum_vals - Vs[p].sum()) <= 1e-9 * np.abs(Vs[p]).sum()
        assert abs(r.checksum_w - Ws[p].sum()) <= 1e-9 * max(1.0, np.abs(Ws[p]).sum())
    # identical to single calls on one context (bit for bit), and interpolation at the sites to the value tolerance
    for p in (0, 17, 40, 63):
        C, Y, X = problems[p]
        mod = pkg.update_model(cfg, C, Y, ctx=ctx)
        assert np.array_equal(mod.weights, Ws[p]) and np.array_equal(mod.poly, Ls[p])
        V, J = mod.eval_sites(X, want_values=True, want_jac=True)
        assert np.array_equal(V, Vs[p])
        if p in Js:
            assert np.array_equal(J, Js[p])
        assert np.abs(pkg.eval_models_at_sites(mod, None, C) - Y).max() < 1e-8 * max(1.0, np.abs(Y).max())
        # Jacobian against central differences of the device values
        for t in (1, 5):
            fd = _fd_jacobian(mod, X[t])
            assert np.abs(fd - J[t]).max() < 1e-6 * max(1.0, np.abs(J[t]).max()), (p, t, np.abs(fd - J[t]).max())
        mod.free()
    # a batch whose arena would exceed the budget is worked off in halves (here 16 starts under a 64 MB budget: groups of four, i.e.
    # clusters of eight workgroups instead of four): the same numbers, bit for bit
    os.environ["MRBF_BATCH_ARENA_MB"] = "64"
    try:
        res2, Ws2, Ls2, Vs2, _ = _batch(problems[:16], kid, a, b, 1)
    finally:
        del os.environ["MRBF_BATCH_ARENA_MB"]
    for p in range(16):
        assert res2[p].status == 0 and res2[p].fit.fallbacks == 0
        assert np.array_equal(Ws2[p], Ws[p]) and np.array_equal(Ls2[p], Ls[p]) and np.array_equal(Vs2[p], Vs[p])
    # one start against the CPU oracle (LU of the saddle system): north-star tolerances outright (cond ~ 2e3)
    C, Y, X = problems[0]
    ref = orc.fit(C, Y, kid, a, b, 1)
    ew = np.abs(Ws[0] - ref.w).max() / np.abs(ref.w).max()
    ev = np.abs(Vs[0][:512] - ref.values(X[:512])).max() / max(1.0, np.abs(Vs[0]).max())
    ej = np.abs(Js[0][:64] - ref.jacs(X[:64])).max() / max(1.0, np.abs(Js[0]).max())
    assert ew < 1e-10 and ev < 1e-8 and ej < 1e-8, (ew, ev, ej)
    print("C4: 64 starts in %.3f s wall (%.1f problems/s incl. host staging), weights vs oracle %.1e" % (wall, P / wall, ew))


def test_c5_full_size_single_problem(ctx):
    """BASELINE.json configs[4]: one of the 256 problems at full size (d = 256, n = 16384, q = 257, m = 1024)"""
    import torch

    C, Y, X = wl.problem("C5", 0)
    n, d = C.shape
    cfg = pkg.RbfConfig(kernel="cubic", polynomial_degree=1)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    mod = pkg.update_model(cfg, C, Y, ctx=ctx)
    info = dict(mod.info)
    assert info["path"] == _lib.PATH_PROJ_CHOL and info["fallbacks"] == 0, info
    assert info["n"] == 16384 and info["q"] == 257
    assert info["rel_residual"] < 1e-10, info
    assert info["max_pitw"] < 1e-9 * max(1.0, np.abs(mod.weights).max()), info
    m = X.shape[0]
    V, J = mod.eval_sites(X, want_values=True, want_jac=True)
    assert np.isfinite(V).all() and np.isfinite(J).all()
    # interpolation at a subset of the sites
    idx = np.arange(0, n, 131)
    assert np.abs(pkg.eval_models_at_sites(mod, None, C[idx]) - Y[idx]).max() < 1e-8 * max(1.0, np.abs(Y).max())
    # device-resident buffers give the same bits as host staging
    Xd = torch.from_numpy(X).cuda()
    Vd = torch.empty((m, 2), dtype=torch.float64, device="cuda")
    Jd = torch.empty((m, d, 2), dtype=torch.float64, device="cuda")
    mod.eval_sites(Xd, want_values=True, want_jac=True, out_vals=Vd, out_jac=Jd)
    torch.cuda.synchronize()
    assert np.array_equal(Vd.cpu().numpy(), V)
    assert np.array_equal(np.transpose(Jd.cpu().numpy(), (0, 2, 1)), J)
    # Jacobian against central differences at two points
    for t in (0, 7):
        fd = _fd_jacobian(mod, X[t], h=1e-5)
        assert np.abs(fd - J[t]).max() < 1e-5 * max(1.0, np.abs(J[t]).max(), np.abs(mod.weights).max() * 1e-3), np.abs(fd - J[t]).max()
    # once against the CPU oracle: per-pair norm(x - c) assembly (threaded C restatement, same arithmetic per entry) + LAPACK LU of
    # the 16641 x 16641 saddle system
    threads = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    Phi, Pi = c_oracle.gram(C, kid, a, b, 1, threads=threads)
    import scipy.linalg

    S = orc.saddle_matrix(Phi, Pi)
    rhs = np.vstack([Y, np.zeros((Pi.shape[1], Y.shape[1]))])
    nrm_S = float(np.linalg.norm(S))
    sol = scipy.linalg.solve(S, rhs, assume_a="gen", overwrite_a=False, check_finite=False)
    t_oracle = time.perf_counter() - t0
    ref = orc.OracleModel(C, sol[:n].copy(), sol[n:].copy(), kid, a, b, 1)
    xg = np.vstack([mod.weights, mod.poly])
    be = float(np.linalg.norm(S @ xg - rhs) / (nrm_S * np.linalg.norm(xg) + np.linalg.norm(rhs)))
    be_o = float(np.linalg.norm(S @ sol - rhs) / (nrm_S * np.linalg.norm(sol) + np.linalg.norm(rhs)))
    ew = np.abs(mod.weights - ref.w).max() / np.abs(ref.w).max()
    ev = np.abs(V[:64] - ref.values(X[:64])).max() / max(1.0, np.abs(V).max())
    ej = np.abs(J[:16] - ref.jacs(X[:16])).max() / max(1.0, np.abs(J).max())
    print("C5: fit %.1f ms (gram %.2f project %.2f factor %.2f solve %.2f), oracle %.1f s on %d threads; weights %.2e values %.2e "
          "jac %.2e backward error %.2e (oracle %.2e)" % (info["ms_total"], info["ms_gram"], info["ms_project"], info["ms_factor"],
                                                          info["ms_solve"], t_oracle, threads, ew, ev, ej, be, be_o))
    assert be <= 50 * EPS, be
    assert ev < 1e-8 and ej < 1e-8, (ev, ej)
    assert ew < 1e-10, (ew, be, be_o)          # outright (observed 1.8e-11)
    mod.free()


@pytest.mark.parametrize("n_list", [(512, 640, 900, 1300, 1800, 2048, 1100, 777)])
def test_batch_run_concurrent_persistent_kernels(ctx, n_list):
    """n in [512, 2048]: every problem goes through potrf_mega_kernel + backsolve_persistent_kernel, four worker contexts at a time on
    one GPU (api.hip: per_dev = 4 for n <= 2048); results must equal single calls bit for bit and no fit may have needed a fallback"""
    problems = []
    for p, n in enumerate(n_list * 2):
        rng = np.random.Generator(np.random.PCG64(300 + p))
        C = rng.random((n, 16))
        Y = np.stack([np.sin(C.sum(axis=1)), (C ** 2).sum(axis=1) / 16], axis=1)
        problems.append((C, Y, rng.random((50, 16))))
    cfg = pkg.RbfConfig(kernel="multiquadric", polynomial_degree=1)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    res, Ws, Ls, Vs, _ = _batch(problems, kid, a, b, 1)
    for p, (C, Y, X) in enumerate(problems):
        assert res[p].status == 0 and res[p].fit.path == _lib.PATH_PROJ_CHOL
        assert res[p].fit.fallbacks == 0, (p, res[p].fit.fallbacks, hex(res[p].fit.giveup_code))
        assert res[p].fit.rel_residual < 1e-11
        mod = pkg.update_model(cfg, C, Y, ctx=ctx)
        assert np.array_equal(mod.weights, Ws[p])
        assert np.array_equal(pkg.eval_models_at_sites(mod, None, X), Vs[p])
        mod.free()


def test_giveup_paths_fall_back_on_the_gpu(ctx):
    """a persistent kernel that abandons a dependency must not fail the fit (nor return garbage): one workgroup is made to skip its
    publish (MRBF_OPT_DEBUG_FAULT), the waiting workgroups give up after MRBF_OPT_SPIN_MS, the same mrbf_fit call re-runs the
    host-driven GPU path, reports it in info.fallbacks and returns the same weights"""
    rng = np.random.Generator(np.random.PCG64(77))
    C = rng.random((1300, 12))
    Y = np.stack([np.sin(C.sum(axis=1)), (C ** 2).sum(axis=1)], axis=1)
    cfg = pkg.RbfConfig(kernel="multiquadric", polynomial_degree=1)
    clean = pkg.update_model(cfg, C, Y, ctx=ctx)
    assert clean.info["fallbacks"] == 0 and clean.info["giveup_code"] == 0
    ctx.set_option(_lib.OPT_SPIN_MS, 20)
    try:
        for bit, fb in ((1, _lib.FB_CHOL_HOST_DRIVEN), (2, _lib.FB_BACKSOLVE_BLOCKED)):
            ctx.set_option(_lib.OPT_DEBUG_FAULT, bit)
            t0 = time.perf_counter()
            m = pkg.update_model(cfg, C, Y, ctx=ctx)
            dt = time.perf_counter() - t0
            assert m.info["fallbacks"] == fb, (bit, m.info)
            assert m.info["giveup_code"] != 0
            assert m.info["path"] == _lib.PATH_PROJ_CHOL and m.info["rel_residual"] < 1e-12, m.info
            assert np.abs(m.weights - clean.weights).max() <= 1e-11 * np.abs(clean.weights).max()
            assert dt < 2.0, dt  # gave up after ~20 ms, did not hang
            m.free()
    finally:
        ctx.set_option(_lib.OPT_DEBUG_FAULT, 0)
        ctx.set_option(_lib.OPT_SPIN_MS, 1000)
    # and the next fit on the same context is clean again
    again = pkg.update_model(cfg, C, Y, ctx=ctx)
    assert again.info["fallbacks"] == 0 and np.array_equal(again.weights, clean.weights)
    again.free()
    clean.free()


def test_giveup_in_the_tail_factorisation_is_not_read_as_rank_deficiency(ctx):
    """128 < d <= 256: the Cholesky-QR of the tail basis (a 256-column tall factorisation) goes through the persistent kernel too, on
    the side stream.  When IT gives up (same fault hook: a half tile of the last row is never published) the fit must report a give-up
    and repeat on the host-driven Cholesky path -- not mistake the negative status for a rank-deficient Pi and fall to the LU of the
    saddle system"""
    rng = np.random.Generator(np.random.PCG64(79))
    C = rng.random((1300, 140))
    Y = np.stack([np.sin(C.sum(axis=1) / 10), (C ** 2).sum(axis=1) / 140], axis=1)
    cfg = pkg.RbfConfig(kernel="cubic", polynomial_degree=1)
    clean = pkg.update_model(cfg, C, Y, ctx=ctx)
    assert clean.info["fallbacks"] == 0 and clean.info["path"] == _lib.PATH_PROJ_CHOL
    ctx.set_option(_lib.OPT_SPIN_MS, 20)
    try:
        ctx.set_option(_lib.OPT_DEBUG_FAULT, 1)
        t0 = time.perf_counter()
        m = pkg.update_model(cfg, C, Y, ctx=ctx)
        dt = time.perf_counter() - t0
        assert m.info["path"] == _lib.PATH_PROJ_CHOL, m.info            # still the projected Cholesky, not the LU fallback
        assert m.info["fallbacks"] & _lib.FB_CHOL_HOST_DRIVEN and not (m.info["fallbacks"] & _lib.FB_LU), m.info
        assert m.info["giveup_code"] != 0 and m.info["rel_residual"] < 1e-11, m.info
        assert np.abs(m.weights - clean.weights).max() <= 1e-10 * np.abs(clean.weights).max()
        assert dt < 3.0, dt
        m.free()
    finally:
        ctx.set_option(_lib.OPT_DEBUG_FAULT, 0)
        ctx.set_option(_lib.OPT_SPIN_MS, 1000)
    clean.free()


def test_affinely_dependent_sites_through_the_three_launch_front_end(ctx):
    """n > 512, d <= 64: the tail basis comes from small.hip's TailQ launches.  Sites in a hyperplane (one coordinate a linear function of
    two others) make Xc'Xc singular up to rounding: either the one-workgroup Cholesky there meets a non-positive pivot (flags[1]) and
    the fit takes the LU of the saddle system, or the pivot comes out as a positive rounding residue and the projected path goes on
    with a basis that spans the same space -- in both cases the model must interpolate, with one coordinate exactly duplicated as well"""
    rng = np.random.Generator(np.random.PCG64(80))
    C = rng.random((700, 9))
    C[:, 4] = 0.5 * C[:, 0] - 0.25 * C[:, 2] + 0.3
    Y = np.sin(C.sum(axis=1))[:, None]
    for dup in (False, True):
        if dup:
            C[:, 4] = C[:, 0]
        m = pkg.update_model(pkg.RbfConfig(kernel="gaussian", polynomial_degree=1), C, Y, ctx=ctx)
        assert m.info["path"] in (_lib.PATH_PROJ_CHOL, _lib.PATH_LU), m.info
        V, _ = m.eval_sites(C[:100])
        assert np.abs(V - Y[:100]).max() < 1e-6, (dup, m.info, np.abs(V - Y[:100]).max())
        m.free()
    # the same sites without the dependency: the projected Cholesky path
    C2 = rng.random((700, 9))
    m2 = pkg.update_model(pkg.RbfConfig(kernel="gaussian", polynomial_degree=1), C2, Y, ctx=ctx)
    assert m2.info["path"] == _lib.PATH_PROJ_CHOL and m2.info["rel_residual"] < 1e-10, m2.info
    m2.free()


def test_small_fit_cluster_failure_repeats_with_one_workgroup():
    """n <= 512: the fit runs on a cluster of four workgroups per problem (small.hip); when a member does not arrive at a barrier
    (MRBF_OPT_DEBUG_FAULT bit 2) the siblings give up after MRBF_OPT_SPIN_MS, the same call repeats the launch with one workgroup per
    problem, and the result is bit for bit the clustered one (every tile is computed by the same code whoever takes it)"""
    rng = np.random.Generator(np.random.PCG64(78))
    C = rng.random((300, 20))
    Y = np.stack([np.sin(C.sum(axis=1)), (C ** 2).sum(axis=1)], axis=1)
    cfg = pkg.RbfConfig(kernel="cubic", polynomial_degree=1)
    good = pkg.Context(0)
    clean = pkg.update_model(cfg, C, Y, ctx=good)
    assert clean.info["path"] == _lib.PATH_PROJ_CHOL and clean.info["rel_residual"] < 1e-12
    ctx2 = pkg.Context(0)   # its own context (a barrier that times out three times in a row switches a context to one workgroup per problem)
    ctx2.set_option(_lib.OPT_SPIN_MS, 20)
    ctx2.set_option(_lib.OPT_DEBUG_FAULT, 4)
    t0 = time.perf_counter()
    m = pkg.update_model(cfg, C, Y, ctx=ctx2)
    dt = time.perf_counter() - t0
    assert dt < 2.0, dt
    assert m.info["path"] == _lib.PATH_PROJ_CHOL and m.info["rel_residual"] < 1e-12, m.info
    assert np.array_equal(m.weights, clean.weights)
    ctx2.set_option(_lib.OPT_DEBUG_FAULT, 0)
    m2 = pkg.update_model(cfg, C, Y, ctx=ctx2)   # clusters again (a time-out says nothing about visibility), same numbers
    assert np.array_equal(m2.weights, clean.weights)
    for x in (m, m2, clean):
        x.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,k", [(300, 700, 2), (2048, 5160, 2), (1500, 130, 3), (2048, 64, 1)])
def test_values_only_evaluation_at_d256_takes_two_outputs_per_pass(ctx, n, m, k):
    """129 <= d <= 256: a Jacobian pass carries one output (its accumulator tiles fill the register budget); a pass for values only
    carries two, so the distances and the radial function are computed once for a pair of outputs (the PS solver's populations:
    C5-shaped models, k = 2).  Same arithmetic per output: the values of a values-only call are bit for bit those of a call that
    also asks for the Jacobians (one output per pass), and they match the oracle's model on the same coefficients."""
    from oracle import rbf_oracle as orc

    d = 200
    rng = np.random.default_rng(n + m + k)
    C = rng.random((n, d))
    Y = np.stack([((C - 0.2 - 0.1 * l) ** 2).sum(1) / d + 0.05 * np.sin(3 * C[:, l]) for l in range(k)], axis=1)
    cfg = pkg.RbfConfig(kernel="cubic", polynomial_degree=1)
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
    mod = pkg.update_model(cfg, C, Y, ctx=ctx)
    X = rng.random((m, d))
    V_only = pkg.eval_models_at_sites(mod, None, X)
    V_both, J = mod.eval_sites(X, want_values=True, want_jac=True)
    assert V_only.shape == (m, k) and np.array_equal(V_only, V_both)
    Vo = orc.OracleModel(C, np.asarray(mod.weights), np.asarray(mod.poly), kid, a, b, 1).values(X)
    assert np.abs(V_only - Vo).max() < 1e-9 * max(1.0, np.abs(Vo).max())
    # interpolation
    assert np.abs(pkg.eval_models_at_sites(mod, None, C[:: max(1, n // 50)]) - Y[:: max(1, n // 50)]).max() < 1e-8
    mod.free()
