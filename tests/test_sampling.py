"""Training-site selection mirrors against oracle/sampling_oracle.py."""
import numpy as np
import pytest

import morbit  # noqa: F401  (registers morbit.jl_amd)
import importlib

pkg = importlib.import_module("morbit.jl_amd")
from oracle import rbf_oracle as orc
from oracle import sampling_oracle as so
from conftest import has_gpu

sampling = pkg.sampling


def _db(seed, d, n):
    rng = np.random.default_rng(seed)
    x = np.full(d, 0.5)
    return x, np.vstack([x, rng.random((n, d))])


@pytest.mark.parametrize("d,n,piv", [(2, 20, 0.1), (3, 60, 0.05), (5, 40, 0.3), (4, 3, 1e-3)])
def test_affinely_independent_filter_matches_oracle(d, n, piv):
    x, sites = _db(d, d, n)
    want, Yw, Zw = so.affinely_independent_indices(x, sites[1:], d, piv)
    flt = sampling.AffinelyIndependentPointFilter(x, sites[1:], n=d, pivot_val=piv)
    got = flt.collect()
    assert got == want
    np.testing.assert_allclose(flt.Y, Yw, atol=0)
    assert flt.Z.shape == Zw.shape
    # accepted directions are affinely independent
    assert np.linalg.matrix_rank(flt.Y) == len(got)


def test_filter_edge_cases():
    x = np.zeros(3)
    assert sampling.AffinelyIndependentPointFilter(x, []).collect() == []
    # collinear seeds: only one is accepted
    seeds = [np.array([t, 0.0, 0.0]) for t in (0.1, 0.5, 0.3)]
    assert sampling.AffinelyIndependentPointFilter(x, seeds, pivot_val=1e-3).collect() == [1]
    with pytest.raises(AssertionError):
        sampling.AffinelyIndependentPointFilter(x, seeds, n=0)


def test_find_suitable_points_box_and_exclusions():
    x, sites = _db(7, 3, 50)
    lb, ub = x - 0.3, x + 0.3
    picked, dirs, cand, Y, Z = sampling._find_suitable_points(sites, lb, ub, x, 0, 0.05, already_inspected_indices=[1, 2])
    assert 0 not in cand and 1 not in cand and 2 not in cand
    assert all(np.all(sites[i] >= lb) and np.all(sites[i] <= ub) for i in cand)
    assert set(picked) <= set(cand) and len(picked) <= 3
    assert len(dirs) == 3 - len(picked)
    assert Y.shape == (3, len(picked))


def test_nullify_last_row():
    rng = np.random.default_rng(3)
    R = np.triu(rng.standard_normal((5, 3)))
    R[3:] = 0
    M = np.vstack([R, rng.standard_normal((1, 3))])
    Rn, G = sampling._nullify_last_row(M)
    Ro, Go = so.nullify_last_row(M)
    np.testing.assert_allclose(Rn, Ro, atol=1e-14)
    np.testing.assert_allclose(G, Go, atol=1e-14)
    np.testing.assert_allclose(G @ M, Rn, atol=1e-14)
    np.testing.assert_allclose(G @ G.T, np.eye(6), atol=1e-14)
    assert np.abs(Rn[-1]).max() < 1e-14


CASES = [("cubic", 0, 3.0, 0.0, 1, 3, 60), ("gaussian", 4, 1.0, 0.0, 1, 2, 40), ("multiquadric", 2, 1.0, 0.5, 1, 4, 80),
         ("inv_multiquadric", 1, 1.0, 0.5, 0, 3, 50), ("thin_plate_spline", 3, 2.0, 0.0, 1, 2, 30)]


def _start_set(x, sites, d):
    idx, _, _ = so.affinely_independent_indices(x, sites[1:], d, 0.05)
    return [0] + [i + 1 for i in idx]


def _cfg(name, deg, mp=-1):
    return pkg.RbfConfig(kernel=name, polynomial_degree=deg, max_model_points=mp)


@pytest.mark.parametrize("name,kid,a,b,deg,d,n", CASES)
def test_round4_host_logic_matches_oracle(name, kid, a, b, deg, d, n):
    x, sites = _db(11, d, n)
    start = _start_set(x, sites, d)
    cfg = _cfg(name, deg)
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    kb = lambda X, C: orc.phi(kidp, ap, bp, orc.pairwise_dist(np.atleast_2d(X), np.atleast_2d(C)))
    cands = [i for i in range(len(sites)) if i not in start]
    want = [cands[p] for p in so.rbf_round4(sites[start], sites[cands], kidp, ap, bp, deg)]
    got = sampling._rbf_round4(sites, np.zeros(d), np.ones(d), x, 1.0, start, cfg, kernel_block=kb)
    assert got == want
    assert len(got) <= (d + 1) * (d + 2) // 2 - len(start)


def test_round4_respects_max_points_and_empty():
    x, sites = _db(5, 3, 40)
    start = _start_set(x, sites, 3)
    cfg = _cfg("cubic", 1, mp=len(start) + 2)
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    kb = lambda X, C: orc.phi(kidp, ap, bp, orc.pairwise_dist(np.atleast_2d(X), np.atleast_2d(C)))
    got = sampling._rbf_round4(sites, np.zeros(3), np.ones(3), x, 1.0, start, cfg, kernel_block=kb)
    assert len(got) == 2
    assert sampling._rbf_round4(sites, np.zeros(3), np.ones(3), x, 1.0, start, _cfg("cubic", 1, mp=len(start)), kernel_block=kb) == []
    # no candidate in the box
    assert sampling._rbf_round4(sites, x - 1e-9, x + 1e-9, x, 1.0, [0], cfg, kernel_block=kb) == []


@pytest.mark.parametrize("name,deg,d,ndb", [("cubic", 1, 3, 6), ("gaussian", 1, 2, 0), ("multiquadric", 0, 2, 4)])
def test_round4_use_max_points_matches_independent_oracle(name, deg, d, ndb):
    # RbfModel.jl:405-416: once the database candidates are used up, random box points are tried until max_points is reached
    x, sites = _db(3, d, max(ndb + d + 2, d + 2))
    start = _start_set(x, sites, d)
    sites = sites[: len(start) + ndb] if ndb else sites[start]
    start = [i for i in start if i < len(sites)] if ndb else list(range(len(sites)))
    cfg = pkg.RbfConfig(kernel=name, polynomial_degree=deg, use_max_points=True)
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    kb = lambda X, C: orc.phi(kidp, ap, bp, orc.pairwise_dist(np.atleast_2d(X), np.atleast_2d(C)))
    lb, ub = np.zeros(d), np.ones(d)
    new_sites = []
    got = sampling._rbf_round4(sites, lb, ub, x, 1.0, start, cfg, kernel_block=kb, rng=np.random.default_rng(9), new_sites=new_sites)
    max_points = (d + 1) * (d + 2) // 2
    assert len(start) + len(got) == max_points                      # "use as many as possible"
    cands = [i for i in range(len(sites)) if i not in start]
    rng = np.random.default_rng(9)
    fresh = [sampling._rand_box_point(lb, ub, rng) for _ in range(10 * max_points + 1)]
    want = so.rbf_round4(sites[start], sites[cands], kidp, ap, bp, deg, extra_sites=fresh)
    want_ids, k = [], 0
    for p in want:
        if p < len(cands):
            want_ids.append(cands[p])
        else:
            want_ids.append(len(sites) + k)
            assert np.array_equal(new_sites[k], fresh[p - len(cands)])
            k += 1
    assert got == want_ids and k == len(new_sites)
    # without the flag the round stops when the database candidates run out
    cfg0 = pkg.RbfConfig(kernel=name, polynomial_degree=deg)
    got0 = sampling._rbf_round4(sites, lb, ub, x, 1.0, start, cfg0, kernel_block=kb)
    assert got0 == [i for i in got if i < len(sites)]


def test_round4_needs_device_without_injected_kernels():
    if has_gpu():
        pytest.skip("GPU present")
    x, sites = _db(5, 3, 20)
    with pytest.raises(Exception):
        sampling._rbf_round4(sites, np.zeros(3), np.ones(3), x, 1.0, [0, 1, 2, 3], _cfg("cubic", 1))


@pytest.mark.gpu
@pytest.mark.parametrize("name,kid,a,b,deg,d,n", CASES)
def test_cross_gram_and_round4_on_device(name, kid, a, b, deg, d, n):
    x, sites = _db(11, d, n)
    cfg = _cfg(name, deg)
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    rng = np.random.default_rng(1)
    X = rng.random((37, d))
    K = sampling.cross_gram(cfg, X, sites, 1.0)
    Ko = orc.phi(kidp, ap, bp, orc.pairwise_dist(X, sites))
    np.testing.assert_allclose(K, Ko, rtol=1e-13, atol=1e-14)
    start = _start_set(x, sites, d)
    cands = [i for i in range(len(sites)) if i not in start]
    want = [cands[p] for p in so.rbf_round4(sites[start], sites[cands], kidp, ap, bp, deg)]
    got = sampling._rbf_round4(sites, np.zeros(d), np.ones(d), x, 1.0, start, cfg)
    assert got == want


@pytest.mark.gpu
def test_cross_gram_shapes_and_errors():
    cfg = _cfg("cubic", 1)
    rng = np.random.default_rng(2)
    for m, n, d in [(1, 1, 1), (3, 1000, 7), (513, 257, 64), (100, 100, 130)]:
        X, C = rng.random((m, d)), rng.random((n, d))
        K = sampling.cross_gram(cfg, X, C)
        np.testing.assert_allclose(K, orc.phi(0, 3.0, 0.0, orc.pairwise_dist(X, C)), rtol=1e-13, atol=1e-14)
    # identical points give exactly phi(0)
    K = sampling.cross_gram(_cfg("gaussian", 1), C[:5], C[:5])
    assert np.all(np.diag(K) == 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,deg,d,ndb", [("cubic", 1, 3, 6), ("gaussian", 1, 2, 0), ("multiquadric", 0, 2, 4)])
def test_round4_use_max_points_on_device(name, deg, d, ndb):
    x, sites = _db(3, d, max(ndb + d + 2, d + 2))
    start = _start_set(x, sites, d)
    sites = sites[: len(start) + ndb] if ndb else sites[start]
    start = [i for i in start if i < len(sites)] if ndb else list(range(len(sites)))
    cfg = pkg.RbfConfig(kernel=name, polynomial_degree=deg, use_max_points=True)
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    lb, ub = np.zeros(d), np.ones(d)
    new_sites = []
    got = sampling._rbf_round4(sites, lb, ub, x, 1.0, start, cfg, rng=np.random.default_rng(9), new_sites=new_sites)
    max_points = (d + 1) * (d + 2) // 2
    assert len(start) + len(got) == max_points
    cands = [i for i in range(len(sites)) if i not in start]
    rng = np.random.default_rng(9)
    fresh = [sampling._rand_box_point(lb, ub, rng) for _ in range(10 * max_points + 1)]
    want = so.rbf_round4(sites[start], sites[cands], kidp, ap, bp, deg, extra_sites=fresh)
    want_ids = [cands[p] if p < len(cands) else len(sites) + k for k, p in zip(np.cumsum([p >= len(cands) for p in want]) - 1, want)]
    assert got == want_ids


@pytest.mark.gpu
@pytest.mark.parametrize("name,deg,d,n,k", [("cubic", 1, 3, 60, 2), ("multiquadric", 1, 6, 400, 3), ("gaussian", 1, 10, 1500, 1),
                                            ("thin_plate_spline", 1, 2, 30, 2), ("cubic", 1, 24, 3000, 2)])
def test_round4_device_selection_and_factor_reuse(name, deg, d, n, k):
    """mrbf_round4 against the independent from-scratch oracle (selection), and mrbf_fit_from_round4 against the oracle's dense
    LU of the saddle system on (start + accepted) sites (weights, values, Jacobians at the north-star tolerances): the n^3/3
    factorisation of the fit is replaced by two triangular solves with the factor the selection left behind."""
    rng = np.random.default_rng(100 + d)
    x = np.full(d, 0.5)
    sites = np.vstack([x, rng.random((n, d))])
    start = _start_set(x, sites, d)
    assert len(start) == d + 1
    cands = [i for i in range(len(sites)) if i not in start]
    cfg = _cfg(name, deg, mp=min((d + 1) * (d + 2) // 2, 2 * d + 1 if d > 10 else 10 ** 6))
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    accepted, st = sampling.rbf_round4_device(cfg, sites[start], sites[cands], 1.0, keep_state=True)
    mp = (d + 1) * (d + 2) // 2 if cfg.max_model_points <= 0 else cfg.max_model_points
    if n <= 400:   # the oracle re-factorises per candidate: O(candidates * N^3)
        want = so.rbf_round4(sites[start], sites[cands], kidp, ap, bp, deg, max_points=mp)
        assert accepted == want
    assert len(accepted) == min(mp - len(start), len(cands)) or n <= 400   # generic sites: every candidate passes until max_points
    assert accepted == sorted(accepted)
    S = st.training_sites
    Y = np.stack([(S ** 2).sum(axis=1), np.sin(S.sum(axis=1)), S[:, 0] - S[:, -1]][:k], axis=1)
    mod = sampling.fit_from_round4(st, Y)
    assert mod.info["path"] == _lib_path_round4() and mod.n == S.shape[0]
    ref = orc.fit(S, Y, kidp, ap, bp, deg)
    Phi, Pi = orc.gram(S, kidp, ap, bp, deg)
    cond = float(np.linalg.cond(orc.saddle_matrix(Phi, Pi)))
    X = rng.random((17, d))
    V, J = mod.eval_sites(X, want_values=True, want_jac=True)
    ew = np.abs(mod.weights - ref.w).max() / np.abs(ref.w).max()
    ev = np.abs(V - ref.values(X)).max() / max(1.0, np.abs(V).max())
    ej = np.abs(J - ref.jacs(X)).max() / max(1.0, np.abs(J).max())
    assert mod.info["rel_residual"] < 1e-10, mod.info
    assert ev < 1e-8 and ej < 1e-8, (ev, ej)
    assert ew < 1e-10 or ew < 100 * np.finfo(float).eps * cond, (ew, cond)
    # the same model as the ordinary fit on the same sites
    full = pkg.update_model(cfg, S, Y)
    assert np.abs(full.weights - mod.weights).max() <= max(1e-10, 100 * np.finfo(float).eps * cond) * np.abs(full.weights).max()
    print("round 4 on the device d=%d: %d of %d candidates accepted, fit from the kept factor %.3f ms (ordinary fit %.3f ms), weights vs oracle %.1e (cond %.1e)"
          % (d, len(accepted), len(cands), mod.info["ms_total"], full.info["ms_total"], ew, cond))
    full.free()
    mod.free()
    st.free()


@pytest.mark.gpu
@pytest.mark.parametrize("name,deg,d,n,mp", [("cubic", 1, 130, 150, 231), ("multiquadric", 1, 12, 700, 91), ("gaussian", 1, 40, 900, 500),
                                             ("inv_multiquadric", 0, 8, 333, 200), ("gaussian", -1, 6, 300, 180)])
def test_round4_walk_variants_agree(name, deg, d, n, mp):
    """Round 5 gave the walk three forms (and the right-looking one a set of switches that select its earlier stages): kappa on demand + per-block triangular solve (left-looking), kappa on demand + R kept for every
    candidate ahead (right-looking, no triangular solve against the accepted factor), and the mc x mc construction of rounds 3 / 4
    (MRBF_R4_LAZY=0).  They compute the same quantities in different orders: the accepted lists must be identical, the fits from the
    kept factors equal to rounding, and the list equal to the independent oracle's.  d = 130: q = 131, the decision kernel's
    three-row register variant; every case crosses at least one block boundary."""
    import os

    rng = np.random.default_rng(300 + d)
    x = np.full(d, 0.5)
    sites = np.vstack([x, x + 0.3 * (rng.random((d + n, d)) - 0.5)])
    start = _start_set(x, sites, d)
    assert len(start) == d + 1
    cands = [i for i in range(len(sites)) if i not in start]
    cfg = _cfg(name, deg, mp=mp)
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    S0, Cc = sites[start], sites[cands]
    results = {}
    # (right-looking walk: also with the register / memory decision kernels of rounds 4 / 3, every wave doing both halves of a step, one
    # stream, rocBLAS for the block's Schur complement, the tail of kappa inside the kappa kernel, no kappa ahead of the decisions)
    variants = (("left", {"MRBF_R4_EAGER": "0"}), ("right", {"MRBF_R4_EAGER": "1"}), ("full", {"MRBF_R4_LAZY": "0"}),
                ("right/select1", {"MRBF_R4_SELECT": "1"}), ("right/select0", {"MRBF_R4_SELECT": "0"}), ("right/noduo", {"MRBF_R4_DUO": "0"}),
                ("right/onestream", {"MRBF_R4_SPLIT": "0"}), ("right/blas-schur", {"MRBF_R4_SCHUR": "0"}),
                ("right/tail-in-kappa", {"MRBF_R4_TAILGEMM": "0"}), ("right/no-prek", {"MRBF_R4_PREK": "0"}),
                ("right/blas-update", {"MRBF_R4_CUSTOM": "0"}))
    for tag, env in variants:
        os.environ.update(env)
        try:
            accepted, st = sampling.rbf_round4_device(cfg, S0, Cc, 1.0, keep_state=True)
            if deg == 1:      # (the kept factor serves the fit only for a unisolvent start set, n0 = q: degree 1 here)
                S = st.training_sites
                Y = np.stack([(S ** 2).sum(axis=1), np.sin(S.sum(axis=1))], axis=1)
                mod = sampling.fit_from_round4(st, Y)
                results[tag] = (accepted, mod.weights.copy(), mod.info["rel_residual"])
                mod.free()
            else:
                results[tag] = (accepted, np.ones(1), 0.0)
            st.free()
        finally:
            for kk in env:
                os.environ.pop(kk, None)
    acc_l, w_l, r_l = results["left"]
    for tag, _ in variants[1:]:
        acc, w, r = results[tag]
        assert acc == acc_l, tag
        assert np.abs(w - w_l).max() <= 1e-8 * np.abs(w_l).max(), (tag, np.abs(w - w_l).max() / np.abs(w_l).max())
        assert r < 1e-9 and r_l < 1e-9
    want = so.rbf_round4(S0, Cc, kidp, ap, bp, deg, max_points=mp)
    assert acc_l == want
    print("round 4 walk variants d=%d: %d of %d accepted, identical lists" % (d, len(acc_l), len(cands)))


def _lib_path_round4():
    from morbit.jl_amd import _lib
    return _lib.PATH_ROUND4


@pytest.mark.gpu
def test_round4_device_refusals():
    from morbit.jl_amd import _lib
    x, sites = _db(5, 3, 30)
    cfg = _cfg("cubic", 1)
    # a start set that does not carry the tail (n0 < q): refused with -2, the mirror then takes the host bookkeeping
    with pytest.raises(pkg.MrbfError) as ei:
        sampling.rbf_round4_device(cfg, sites[:2], sites[2:])
    assert ei.value.code == -2
    # affinely dependent start set (4 collinear sites in 3-D): rank-deficient polynomial matrix
    line = np.array([[0.1 * t, 0.2 * t, 0.3 * t] for t in range(1, 5)])
    with pytest.raises(pkg.MrbfError) as ei:
        sampling.rbf_round4_device(cfg, line, sites[2:])
    assert ei.value.code == _lib.MRBF_ESINGULAR
    # max_points already reached / no candidates: nothing selected, no state
    assert sampling.rbf_round4_device(_cfg("cubic", 1, mp=4), sites[:4], sites[4:]) == []
    assert sampling.rbf_round4_device(cfg, sites[:4], np.empty((0, 3))) == []
    # factor reuse needs n0 == q: five start sites in 3-D
    acc, st = sampling.rbf_round4_device(cfg, sites[:5], sites[5:], keep_state=True)
    S = st.training_sites
    with pytest.raises(pkg.MrbfError) as ei:
        sampling.fit_from_round4(st, np.zeros((S.shape[0], 1)))
    assert ei.value.code == -2
    st.free()


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,piv", [(2, 20, 0.1), (5, 400, 0.3), (10, 20000, 0.05), (3, 3, 1e-3)])
def test_affinely_independent_filter_on_device(d, n, piv):
    # the candidate scan of AffinelyIndependentPoints.jl:71-106 through mrbf_affine_scores: same picks, in the same order
    x, sites = _db(40 + d, d, n)
    want, Yw, _ = so.affinely_independent_indices(x, sites[1:], d, piv)
    flt = sampling.AffinelyIndependentPointFilter(x, sites[1:], n=d, pivot_val=piv, ctx=pkg.default_context())
    got = flt.collect()
    assert got == want
    np.testing.assert_allclose(flt.Y, Yw, atol=0)
    # scores of one scan against NumPy, first maximiser on ties (duplicated rows), 2-norm variant
    from morbit.jl_amd import _lib
    import ctypes
    ctx = pkg.default_context()
    rng = np.random.default_rng(d)
    S = rng.standard_normal((max(n, 8), d))
    S[5] = S[2]
    Z = so.orthogonal_complement_matrix(rng.standard_normal((d, max(d // 2, 1))))
    for p_inf in (1, 0):
        vals = np.empty(S.shape[0])
        best, val = ctypes.c_int64(), ctypes.c_double()
        Zf = np.asfortranarray(Z)
        ctx.check(ctx.lib.mrbf_affine_scores(ctx.h, S.shape[0], d, Z.shape[1], _lib.as_ptr(S), _lib.as_ptr(Zf), p_inf, _lib.as_ptr(vals),
                                             ctypes.byref(best), ctypes.byref(val)))
        ref = np.linalg.norm((S @ Z) @ Z.T, ord=np.inf if p_inf else 2, axis=1)
        np.testing.assert_allclose(vals, ref, rtol=1e-12, atol=1e-14)
        assert best.value == int(np.argmax(vals)) and val.value == vals[best.value]
    # identical rows 2 and 5: the first one wins when they are the maximum
    Sm = np.zeros((8, d))
    Sm[2] = Sm[5] = 3.0
    ctx.check(ctx.lib.mrbf_affine_scores(ctx.h, 8, d, d, _lib.as_ptr(Sm), _lib.as_ptr(np.asfortranarray(np.eye(d))), 1, None, ctypes.byref(best), ctypes.byref(val)))
    assert best.value == 2 and val.value == 3.0


@pytest.mark.parametrize("d", [3, 17, 64, 128])
def test_growing_qr_equals_refactoring_from_scratch(d):
    """The filter keeps the Householder factorisation of its Y up to date by one reflector per pick (`sampling._GrowingQR`, twin of
    `HipRbfGrowingQR` in HipRbf.jl) where the reference re-factors Y after every pick (AffinelyIndependentPoints.jl:4-11, :59-60,
    :93-94).  Same reflectors: the normalised complement Z agrees with `_orthogonal_complement_matrix(Y)` to rounding after every
    appended column -- entry by entry, signs included -- also when the factorisation starts from a given Y (round 2)."""
    rng = np.random.default_rng(d)
    Ys = rng.standard_normal((d, d))
    qr = sampling._GrowingQR(d)
    for j in range(d):
        qr.append(Ys[:, j])
        Zi, Zf = qr.complement(), sampling._orthogonal_complement_matrix(Ys[:, : j + 1])
        assert Zi.shape == Zf.shape == (d, d - j - 1)
        if Zi.size:
            assert np.abs(Zi - Zf).max() < 1e-12, (d, j)
    j0 = max(1, d // 3)
    qr = sampling._GrowingQR(d, Ys[:, :j0])
    qr.append(Ys[:, j0])
    Zf = sampling._orthogonal_complement_matrix(Ys[:, : j0 + 1])
    if Zf.size:
        assert np.abs(qr.complement() - Zf).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,piv,p", [(128, 300, 0.05, np.inf), (64, 2000, 0.02, np.inf), (33, 500, 0.2, 2), (128, 90, 0.05, np.inf), (7, 5000, 0.45, np.inf)])
def test_affine_select_on_device_is_the_filter(d, n, piv, p):
    """mrbf_affine_select (round 6): the filter's whole pick loop in one device call, the Householder factorisation of the chosen
    directions grown by a reflector per pick on the device.  Against the host filter (same rule, host scan + `_GrowingQR`) and the
    independent oracle: same picks in the same order, the same final Y, the final Z to rounding -- from scratch (round 1), continuing
    from a given Y / Z with fewer picks wanted (round 2, RbfModel.jl:252-266), with the 2-norm, with fewer candidates than directions,
    and when the pivot test stops the loop early."""
    x, sites = _db(90 + d, d, n)
    seeds = sites[1:]
    host = sampling.AffinelyIndependentPointFilter(x, seeds, n=d, pivot_val=piv, p=p)
    monkey = host.collect.__func__            # force the host path whatever the decision table says
    import morbit.jl_amd._lib as L
    real = L.load().mrbf_dispatch_affine
    try:
        L.load().mrbf_dispatch_affine = lambda *a: L.DISPATCH_REFERENCE
        want = monkey(host)
    finally:
        L.load().mrbf_dispatch_affine = real
    dev = sampling.AffinelyIndependentPointFilter(x, seeds, n=d, pivot_val=piv, p=p, ctx=pkg.default_context())
    got = dev.collect()
    assert got == want, (len(got), len(want))
    np.testing.assert_allclose(dev.Y, host.Y, atol=0)
    assert dev.Z.shape == host.Z.shape
    if dev.Z.size:
        # the complement basis is unique up to rounding: same reflectors on host and device
        assert np.abs(dev.Z - host.Z).max() < 1e-10, np.abs(dev.Z - host.Z).max()
        assert np.abs(dev.Z.T @ dev.Y).max() < 1e-12 * max(1.0, np.abs(dev.Y).max())          # orthogonal to everything chosen
    if np.isinf(p):
        idx, Yw, _ = so.affinely_independent_indices(x, seeds, d, piv)
        assert got == idx
    # round 2: continue from the first half of the picks with a new, larger candidate set
    if len(got) >= 4:
        half = len(got) // 2
        Y0 = host.Y[:, :half]
        Z0 = sampling._orthogonal_complement_matrix(Y0, p)
        more = list(seeds) + [x + 0.9 * (s - x) for s in seeds[: n // 2]]
        h2 = sampling.AffinelyIndependentPointFilter(x, more, n=3, Y=Y0, Z=Z0, pivot_val=piv, p=p)
        try:
            L.load().mrbf_dispatch_affine = lambda *a: L.DISPATCH_REFERENCE
            w2 = monkey(h2)
        finally:
            L.load().mrbf_dispatch_affine = real
        d2 = sampling.AffinelyIndependentPointFilter(x, more, n=3, Y=Y0, Z=Z0, pivot_val=piv, p=p, ctx=pkg.default_context())
        assert d2.collect() == w2 and len(w2) <= 3
        np.testing.assert_allclose(d2.Y, h2.Y, atol=0)
