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
