"""Population-batched Pascoletti-Serafini step (morbit.jl_amd/pascoletti_serafini.py) against the contract of
descent.jl:512-581.  CPU tests drive it with analytic models; the GPU test with a fitted RBF container."""
import importlib

import numpy as np
import pytest

import morbit  # noqa: F401

pkg = importlib.import_module("morbit.jl_amd")
ps = importlib.import_module("morbit.jl_amd.pascoletti_serafini")


def _quadratics(centres):
    C = np.asarray(centres, dtype=np.float64)

    def f(X):
        X = np.atleast_2d(X)
        return np.stack([np.sum((X - c) ** 2, axis=1) for c in C], axis=1)

    def jac(X):
        X = np.atleast_2d(X)
        return np.stack([2.0 * (X - c) for c in C], axis=1)

    return f, jac


def test_config_mirror_and_budgets():
    cfg = ps.PascolettiSerafiniConfig()
    assert cfg.main_algo == "GN_ISRES" and cfg.reference_trust_region_factor == 1.1 and cfg.ps_polish_algo is None
    with pytest.raises(AssertionError):
        ps.PascolettiSerafiniConfig(reference_direction=[1.0, -1.0])
    assert ps._ps_max_evals(cfg, 3) == (2000, 0)                                         # 500 (n+1), no polish
    assert ps._ps_max_evals(ps.PascolettiSerafiniConfig(ps_polish_algo="LD_MMA"), 3) == (1500, 500)
    assert ps._ps_max_evals(ps.PascolettiSerafiniConfig(ps_polish_algo="LD_MMA", max_ps_polish_evals=7, max_ps_problem_evals=100), 3) == (100, 7)
    assert ps._get_global_dir(cfg, [1.0, 2.0]) is None
    assert np.array_equal(ps._get_global_dir(ps.PascolettiSerafiniConfig(reference_direction=[1.0, 2.0]), [5.0, 5.0]), [1.0, 2.0])
    assert np.array_equal(ps._get_global_dir(ps.PascolettiSerafiniConfig(reference_point=[1.0, 2.0]), [5.0, 5.0]), [4.0, 3.0])


def test_stochastic_rank_helper_properties():
    rng = np.random.default_rng(0)
    f = rng.standard_normal(40)
    # all feasible: a plain sort by objective, whatever the random numbers
    idx = ps._stochastic_ranking(f, np.zeros(40), rng)
    assert np.array_equal(np.sort(idx), np.arange(40)) and np.all(np.diff(f[idx]) >= 0)
    # pf = 0: infeasible pairs are compared by violation only -> feasible individuals (sorted by f) first, then by violation
    phi = np.where(np.arange(40) % 3 == 0, rng.random(40) + 0.1, 0.0)
    idx = ps._stochastic_ranking(f, phi, rng, pf=0.0)
    nfeas = int((phi == 0).sum())
    assert np.all(phi[idx[:nfeas]] == 0) and np.all(np.diff(f[idx[:nfeas]]) >= 0) and np.all(np.diff(phi[idx[nfeas:]]) >= 0)


def test_isres_generation_batches_and_constrained_optimum():
    # min x0 + x1  s.t.  1 - x0 x1 <= 0 in [0, 4]^2  -> (1, 1), value 2
    rng = np.random.default_rng(1)
    run = ps._Isres([0.0, 0.0], [4.0, 4.0], [3.0, 3.0], 6000, rng)
    sizes = []

    def evaluate(asks):
        X = asks[0]
        sizes.append(X.shape[0])
        return [(X[:, 0] + X[:, 1], (1.0 - X[:, 0] * X[:, 1])[:, None])]

    calls = ps._run_populations([run], evaluate)
    assert calls == len(sizes) and max(sizes) == 60 and run.evals <= 6000      # population 20 (n + 1), one call per generation
    assert run.best_phi == 0.0 and abs(run.best_f - 2.0) < 5e-2
    assert np.all(run.best_x >= 0) and np.all(run.best_x <= 4) and run.best_x[0] * run.best_x[1] >= 1.0


def test_local_ideal_point_runs_all_objectives_side_by_side():
    f, _ = _quadratics([[0.2, 0.2], [0.8, 0.6]])
    calls = []

    def ev(X):
        calls.append(X.shape[0])
        return f(X)

    stats = {}
    ideal = ps.compute_local_ideal_point(np.array([0.5, 0.5]), [0.3, 0.3], [0.7, 0.7], ev, 2, 1500, np.random.default_rng(3), stats=stats)
    # minima over the box: objective 0 at (0.3, 0.3) -> 0.02; objective 1 at (0.7, 0.6) -> 0.01
    assert np.allclose(ideal, [0.02, 0.01], atol=2e-3)
    assert stats["ideal_point_calls"] == len(calls) and max(calls) == 2 * 60    # both populations in one batched call


@pytest.mark.parametrize("polish", [False, True])
def test_get_criticality_contract(polish):
    f, jac = _quadratics([[0.2, 0.2, 0.5], [0.8, 0.6, 0.5]])
    x = np.array([0.5, 0.9, 0.5])
    lb, ub = x - 0.2, x + 0.2
    cfg = ps.PascolettiSerafiniConfig(ps_polish_algo="LD_MMA" if polish else None)
    stats = {}
    omega, rest = ps.get_criticality(cfg, x, x, f(x)[0], lb, ub, f, eval_jacobians=jac, rng=np.random.default_rng(5), stats=stats)[:2]
    x_trial, mx_trial, sl = rest
    mx = f(x)[0]
    assert 0 < omega <= 1.0
    assert np.all(x_trial >= lb - 1e-15) and np.all(x_trial <= ub + 1e-15)
    assert np.allclose(mx_trial, f(x_trial)[0]) and sl == pytest.approx(np.abs(x - x_trial).max())
    # feasibility of the subproblem at tau = -omega: every modelled objective improves by at least omega * r_l
    ideal = ps.compute_local_ideal_point(x, lb, ub, f, 2, 2000, np.random.default_rng(5))
    r = mx - ideal
    assert np.all(mx_trial - mx + omega * r <= 5e-3 * np.abs(r) + 1e-12)
    assert np.all(mx_trial < mx)
    assert stats["ps_evals"] <= 500 * 4 and stats["ps_calls"] <= stats["ps_evals"] // 80 + 2   # ~ one call per 80-point generation


def test_critical_point_and_given_direction():
    f, _ = _quadratics([[0.5, 0.5]])
    x = np.array([0.5, 0.5])            # the minimiser of the single objective: ideal point == f(x) -> r = 0 -> critical
    out = ps.get_criticality(ps.PascolettiSerafiniConfig(), x, x, f(x)[0], x - 0.1, x + 0.1, f, rng=np.random.default_rng(0))
    assert out[0] == 0 and np.array_equal(out[1], x) and out[3] == 0
    # a user-supplied direction skips the ideal-point runs
    f2, _ = _quadratics([[0.0, 0.0], [1.0, 0.0]])
    x = np.array([0.5, 0.4])
    stats = {}
    omega, (xt, mt, sl) = ps.get_criticality(ps.PascolettiSerafiniConfig(reference_direction=[1.0, 1.0]), x, x, f2(x)[0], x - 0.3, x + 0.3,
                                             f2, rng=np.random.default_rng(2), stats=stats)
    assert "ideal_point_calls" not in stats and omega > 0 and np.all(mt <= f2(x)[0] - omega + 1e-6)


@pytest.mark.gpu
def test_ps_step_on_device_container():
    from conftest import has_gpu
    assert has_gpu()
    rng = np.random.default_rng(4)
    d, n = 4, 200
    C = rng.random((n, d))
    Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1)], axis=1)
    mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
    sur = pkg.surrogates.RefSurrogate(mod, [0, 1])
    sc = pkg.surrogates.SurrogateContainer(objectives=[sur])
    launches = []

    def ev(X):
        launches.append(X.shape[0])
        return pkg.surrogates.eval_container_objectives_at_scaled_sites(sc, None, X)

    x = np.array([0.5, 0.9, 0.5, 0.15])   # off the segment between the two minimisers: not Pareto-critical
    lb, ub = x - 0.15, x + 0.15
    fx = ev(x[None, :])[0]
    stats = {}
    omega, (xt, mt, sl) = ps.get_criticality(ps.PascolettiSerafiniConfig(max_ps_problem_evals=1500, max_ideal_point_problem_evals=1500),
                                             x, x, fx, lb, ub, ev, rng=np.random.default_rng(9), stats=stats)
    assert omega > 0 and np.all(mt < fx) and np.all(xt >= lb) and np.all(xt <= ub)
    assert np.allclose(mt, ev(xt[None, :])[0], rtol=1e-12, atol=1e-12)
    # one surrogate sweep per generation: the number of launches is ~ evaluations / population, not evaluations
    assert len(launches) < 60 and sum(launches) > 2500
    mod.free()


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,given_dir", [(4, 200, False), (12, 512, False), (6, 300, True), (20, 600, False)])
def test_ps_step_with_the_solver_on_the_device(d, n, given_dir):
    """mrbf_ps_step: population state, ranking and breeding on the device.  Asserted: the contract of get_criticality
    (descent.jl:512-581) -- budgets, feasibility of the returned point for the subproblem, omega = |tau|, trial point inside
    the box, model values of the trial point -- and that the device solver is at least as good as the host-loop mirror."""
    rng = np.random.default_rng(4 + d)
    C = rng.random((n, d))
    Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1) + 0.1 * np.sin(5 * C[:, 0])], axis=1)
    mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
    ev = lambda X: pkg.eval_models_at_sites(mod, None, X)
    x = np.full(d, 0.5)
    x[1], x[3] = 0.85, 0.2
    lb, ub = x - 0.12, x + 0.12
    fx = ev(x[None, :])[0]
    cfg = ps.PascolettiSerafiniConfig(reference_direction=[1.0, 0.5]) if given_dir else ps.PascolettiSerafiniConfig()
    stats = {}
    omega, (xt, mt, sl) = ps.get_criticality_device(cfg, mod, x, x, fx, lb, ub, seed=11, stats=stats)
    r = stats["r"]
    assert stats["status"] == 0 and omega == abs(stats["tau"]) and omega > 0
    assert np.all(xt >= lb) and np.all(xt <= ub) and abs(sl - np.abs(x - xt).max()) < 1e-15
    assert np.allclose(mt, ev(xt[None, :])[0], rtol=0, atol=1e-12)
    # feasible for the subproblem: m_l(x_trial) - m_l(x_n) - tau r_l <= 0 (descent.jl:443), so every objective improves by >= omega r_l
    assert np.all(mt - fx + omega * r <= 1e-9 * max(1.0, np.abs(fx).max()))
    # info.tau is what the returned point achieves (recomputed on the host from a fresh sweep): an individual recorded as the
    # feasible best must BE feasible
    assert np.max((ev(xt[None, :])[0] - fx) / r) <= stats["tau"] + 1e-9
    # budgets (descent.jl:416, :527): never more evaluations than allowed; one batched sweep per generation
    assert stats["evals_ps"] <= 500 * (d + 1)
    if given_dir:
        assert stats["evals_ideal"] == 0 and np.array_equal(r, [1.0, 0.5])
    else:
        assert 0 < stats["evals_ideal"] <= 2 * 500 * (d + 1) and np.all(r > 0)
    assert stats["generations"] <= (stats["evals_ideal"] + stats["evals_ps"]) // (20 * (d + 1)) + 4
    # same seed, same step (counter-based generator); another seed, another trajectory but the same contract
    stats2 = {}
    omega2, (xt2, _, _) = ps.get_criticality_device(cfg, mod, x, x, fx, lb, ub, seed=11, stats=stats2)
    assert omega2 == omega and np.array_equal(xt2, xt)
    # generations without an infeasible individual are ranked by a bitonic network instead of the transposition phases: the
    # order (hence the whole trajectory) must be the same as with the phases everywhere (MRBF_PS_DBG=4 switches the network off)
    import os
    os.environ["MRBF_PS_DBG"] = "4"
    try:
        omega3, (xt3, _, _) = ps.get_criticality_device(cfg, mod, x, x, fx, lb, ub, seed=11)
    finally:
        del os.environ["MRBF_PS_DBG"]
    assert omega3 == omega and np.array_equal(xt3, xt)
    # at least as good as the host-loop mirror on the same problem with the same direction (both are stochastic: 20 % slack)
    cfg_r = ps.PascolettiSerafiniConfig(reference_direction=list(r))
    omega_h, _ = ps.get_criticality(cfg_r, x, x, fx, lb, ub, ev, rng=np.random.default_rng(3))[:2]
    omega_d, _ = ps.get_criticality_device(cfg_r, mod, x, x, fx, lb, ub, seed=5)[:2]
    assert omega_d >= 0.8 * omega_h, (omega_d, omega_h)
    print("PS step d=%d: %.2f ms on the device, %d evaluations in %d generations, omega %.4f (host mirror %.4f)"
          % (d, stats["ms_total"], stats["evals_ideal"] + stats["evals_ps"], stats["generations"], omega_d, omega_h))
    mod.free()


def _check_ps_contract(ev_obj, ev_con, x, lb, ub, fx, omega, xt, mt, stats, lin=None, strict=True):
    """the contract of get_criticality for a returned (omega, x_trial, mx_trial), recomputed on the host from batched sweeps"""
    r = stats["r"]
    assert stats["status"] == 0 and omega == abs(stats["tau"]) and 0 <= omega <= 1.0 and (omega > 0 or not strict)
    assert np.all(xt >= lb) and np.all(xt <= ub)
    assert np.allclose(mt, ev_obj(xt[None, :])[0], rtol=0, atol=1e-10 * max(1.0, np.abs(mt).max()))
    # tau is what the trial point achieves: max_l (m_l(x_trial) - m_l(x_n)) / r_l <= tau  (feasible for descent.jl:443), recomputed
    mx = ev_obj(x[None, :])[0]
    tau_host = np.max((ev_obj(xt[None, :])[0] - mx) / r)
    assert tau_host <= stats["tau"] + 1e-9, (tau_host, stats["tau"])
    if ev_con is not None:
        g = ev_con(xt[None, :])[0]
        assert np.all(g <= 1e-8), g
    if lin is not None and lin[3] is not None:
        assert np.all(lin[2] @ xt - lin[3] <= 1e-8)


@pytest.mark.gpu
def test_ps_ranking_on_several_compute_units_is_the_same_ranking():
    """Populations of 1024 individuals and more (d >= 51) are ranked on several compute units per run (ps_rank_wave_kernel: one wave per
    64 individuals, blocks of 32 transposition phases on windows with halos, exchanged through global memory; MRBF_PS_MULTI=2: round 5's
    sixteen workgroups, ps_rank_sort_kernel).  Same comparisons, same draws: the step must be THE
    step of the one-workgroup ranking (MRBF_PS_MULTI=0) bit for bit -- also when the several-workgroup kernel gives up (MRBF_PS_DBG=64:
    the failure word is set at once and the finishing launch runs the phases itself), and when the phases are forced on generations
    that a sort would rank (MRBF_PS_DBG=4)."""
    import os
    d, n = 110, 400
    rng = np.random.default_rng(77)
    C = rng.random((n, d))
    Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1) + 0.1 * np.sin(5 * C[:, 0])], axis=1) / d
    mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
    x = np.full(d, 0.5)
    x[1], x[3] = 0.85, 0.2
    lb, ub = x - 0.1, x + 0.1
    fx = pkg.eval_models_at_sites(mod, None, x[None, :])[0]
    cfg = ps.PascolettiSerafiniConfig()
    out = {}
    for tag, env in (("multi", {}), ("single", {"MRBF_PS_MULTI": "0"}), ("workgroups", {"MRBF_PS_MULTI": "2"}), ("gave-up", {"MRBF_PS_DBG": "64"}), ("multi/phases", {"MRBF_PS_DBG": "4"}),
                     ("own-centring", {"MRBF_PS_FUSEPAD": "0"}),   # the evaluation's centring launch instead of the breeding kernel's centred copy
                     ("own-scoring", {"MRBF_PS_FUSESCORE": "0"}),  # ps_score_kernel instead of the ranking kernel scoring its own run
                     ("single/phases", {"MRBF_PS_MULTI": "0", "MRBF_PS_DBG": "4"})):
        os.environ.update(env)
        try:
            st = {}
            omega, (xt, mt, _) = ps.get_criticality_device(cfg, mod, x, x, fx, lb, ub, seed=3, stats=st)
            out[tag] = (omega, xt.copy(), st["evals_ideal"] + st["evals_ps"], st["ms_total"])
        finally:
            for kk in env:
                os.environ.pop(kk, None)
    for tag in ("single", "workgroups", "gave-up", "multi/phases", "single/phases", "own-centring", "own-scoring"):
        assert out[tag][0] == out["multi"][0] and np.array_equal(out[tag][1], out["multi"][1]) and out[tag][2] == out["multi"][2], tag
    assert out["multi"][0] > 0
    print("PS ranking d=%d (lambda %d): %.1f ms with a wave per 64 individuals, %.1f ms on sixteen workgroups per run, %.1f ms on one, %.1f ms after a give-up; identical steps"
          % (d, 20 * (d + 2), out["multi"][3], out["workgroups"][3], out["single"][3], out["gave-up"][3]))
    mod.free()


@pytest.mark.gpu
@pytest.mark.parametrize("d", [24, 64, 128, 256])
def test_ps_step_on_device_at_baseline_dimensions(d):
    """The device solver at the dimensions of BASELINE.json: d = 64 on a C3-shaped model (the C3 centres, n = 8192, multiquadric),
    d = 128 on one C4 start (ZDT1, n = 2d + 1 = 257 -- the reference's own `:ps` example, examples/example_zdt.jl:39) and a d between
    the small tests and those.  (a) the paper benchmark's configuration (examples/large_scale_benchmarks.jl:215-219: direction from a
    reference point, 50 (d + 1) global + 100 (d + 1) polish evaluations): contract asserts, tau recomputed on the host, same seed ->
    same step; (b) Morbit's defaults (local ideal point + 500 (d + 1) evaluations, no polish): contract + wall time."""
    from morbit.jl_amd import workloads

    if d == 64:
        C = workloads.problem("C3")[0]
        assert C.shape == (8192, 64)
        Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1)], axis=1) / d
        cfg = pkg.RbfConfig(kernel="multiquadric")
    elif d == 128:
        C, Y, _ = workloads.problem("C4", 0)
        assert C.shape == (257, 128)
        cfg = pkg.RbfConfig(kernel="cubic")
    else:
        rng = np.random.default_rng(d)
        C = rng.random((700 if d < 256 else 2048, d))     # d = 256: C5's dimension
        Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1)], axis=1) / d
        cfg = pkg.RbfConfig(kernel="cubic")
    mod = pkg.update_model(cfg, C, Y)
    ev = lambda X: pkg.eval_models_at_sites(mod, None, X)
    if d == 128:
        x = C[0].copy()                    # the start point itself (first training site)
    else:
        x = np.full(d, 0.5)
        x[1], x[3] = 0.8, 0.25
    lb, ub = np.maximum(x - 0.1, 0.0), np.minimum(x + 0.1, 1.0)
    fx = ev(x[None, :])[0]
    bench = ps.PascolettiSerafiniConfig(reference_point=[-1.0, -1.0], max_ps_problem_evals=50 * (d + 1), max_ps_polish_evals=100 * (d + 1),
                                        ps_polish_algo="LD_MMA")
    stats = {}
    omega, (xt, mt, sl) = ps.get_criticality_device(bench, mod, x, x, fx, lb, ub, seed=21, stats=stats)
    _check_ps_contract(ev, None, x, lb, ub, fx, omega, xt, mt, stats)
    assert stats["evals_ideal"] == 0 and stats["evals_ps"] <= 50 * (d + 1) and 0 < stats["evals_polish"] <= 100 * (d + 1)
    assert np.allclose(stats["r"], fx + 1.0) and abs(sl - np.abs(x - xt).max()) < 1e-15 and np.all(mt < fx)
    stats2 = {}
    omega2, (xt2, _, _) = ps.get_criticality_device(bench, mod, x, x, fx, lb, ub, seed=21, stats=stats2)
    assert omega2 == omega and np.array_equal(xt2, xt)
    stats3 = {}
    omega3, rest = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=22, stats=stats3)[:2]
    assert stats3["status"] == 0
    _check_ps_contract(ev, None, x, lb, ub, fx, omega3, rest[0], rest[1], stats3)      # omega > 0: a step is found with the defaults too
    assert stats3["evals_ps"] <= 500 * (d + 1) and stats3["evals_ideal"] <= 2 * 500 * (d + 1)
    print("PS step d=%d n=%d: benchmark budgets %.1f ms (%d + %d polish evaluations, %d generations, omega %.5f); Morbit defaults %.1f ms "
          "(%d evaluations, %d generations, omega %.5f)" % (d, C.shape[0], stats["ms_total"], stats["evals_ps"], stats["evals_polish"],
                                                            stats["generations"], omega, stats3["ms_total"], stats3["evals_ideal"] + stats3["evals_ps"],
                                                            stats3["generations"], omega3))
    mod.free()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["d24", "d64", "d128", "d256"])
def test_ps_step_omega_against_independent_reference(name):
    """What omega should be: tests/golden/ps_omega.json holds omega* of the subproblem (descent.jl:434-510) for four fixed problems
    -- d = 24, d = 64 on the C3 sites, d = 128 on C4 start 0 (ZDT1), d = 256 (C5's dimension) -- computed by SciPy's SLSQP on the
    ORACLE's model from several starts (tests/golden/make_ps_omega.py; none of the product's code).  The start points are clearly
    non-critical (0.9 * ones for the two quadratics, the Halton start for ZDT1).  Asserted: with the paper benchmark's budgets
    (50 (d + 1) global + 100 (d + 1) polish evaluations, examples/large_scale_benchmarks.jl:215-219) the device step reaches at least
    half of omega*; with Morbit's defaults (GN_ISRES, 500 (d + 1) evaluations, no polish) it finds a step at all (omega > 0; under
    optimize() omega = 0 reads "critical" and ends the run) and a sizeable part of what is possible."""
    import importlib.util
    import json
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    gold = json.load(open(os.path.join(here, "golden", "ps_omega.json")))[name]
    spec = importlib.util.spec_from_file_location("make_ps_omega", os.path.join(here, "golden", "make_ps_omega.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    C, Y, kernel, x, half = mk.problems()[name]
    d = C.shape[1]
    mod = pkg.update_model(pkg.RbfConfig(kernel=kernel), C, Y)
    ev = lambda X: pkg.eval_models_at_sites(mod, None, X)
    lb, ub = np.maximum(x - half, 0.0), np.minimum(x + half, 1.0)
    fx = ev(x[None, :])[0]
    assert np.allclose(fx, gold["mx"], rtol=0, atol=1e-8 * max(1.0, np.abs(fx).max()))     # the same model as the oracle's
    bench = ps.PascolettiSerafiniConfig(reference_point=[-1.0, -1.0], max_ps_problem_evals=50 * (d + 1), max_ps_polish_evals=100 * (d + 1),
                                        ps_polish_algo="LD_MMA")
    st = {}
    omega, (xt, mt, sl) = ps.get_criticality_device(bench, mod, x, x, fx, lb, ub, seed=31, stats=st)
    _check_ps_contract(ev, None, x, lb, ub, fx, omega, xt, mt, st)
    assert st["evals_ps"] <= 50 * (d + 1) and st["evals_polish"] <= 100 * (d + 1)
    assert omega >= 0.5 * gold["omega_bench"], (omega, gold["omega_bench"])
    assert omega <= gold["omega_bench"] * (1 + 1e-6) + 1e-9, "the device found more than the reference solver: regenerate the golden file"
    st2 = {}
    omega2, rest = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=32, stats=st2)[:2]
    assert st2["status"] == 0
    _check_ps_contract(ev, None, x, lb, ub, fx, omega2, rest[0], rest[1], st2)
    assert st2["evals_ps"] <= 500 * (d + 1) and st2["evals_ideal"] <= 2 * 500 * (d + 1)
    assert omega2 > 0
    # the direction is the device's own (its local ideal point): compare what the step achieves, objective by objective, with
    # what the reference solver achieves for ITS direction -- at least a quarter of that relative improvement
    r_gold = np.array(gold["r_default"])
    assert np.all(st2["r"] > 0) and np.all(st2["r"] <= r_gold * 1.05 + 1e-12), (st2["r"], r_gold)    # no ideal point below the true one
    assert omega2 * np.min(st2["r"] / r_gold) >= 0.25 * gold["omega_default"] or omega2 >= 0.25 * gold["omega_default"], (omega2, st2["r"], gold)
    line = ("%-6s n=%d d=%d: benchmark budgets omega %.5f / omega* %.5f = %.3f in %.1f ms (%d + %d evaluations); defaults omega %.5f (own r), %.5f in "
            "the reference's direction / omega* %.5f = %.3f in %.1f ms (%d evaluations)"
            % (name, C.shape[0], d, omega, gold["omega_bench"], omega / gold["omega_bench"], st["ms_total"], st["evals_ps"], st["evals_polish"], omega2,
               float(-np.max((rest[1] - fx) / r_gold)), gold["omega_default"], float(-np.max((rest[1] - fx) / r_gold)) / gold["omega_default"],
               st2["ms_total"], st2["evals_ideal"] + st2["evals_ps"]))
    print(line)
    _append_ratio_line(line)
    mod.free()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["d64c", "d256c"])
def test_ps_step_omega_conflicting_objectives(name):
    """The hard case of the same comparison (round 5): the start lies next to the Pareto set of the two quadratics -- 0.5 * ones pushed
    off the segment 0.3 * ones .. 0.7 * ones by +- 0.15 per coordinate --, the two gradients nearly oppose each other and the descent
    cone is narrow: omega* (SciPy SLSQP on the oracle's model, tests/golden/make_ps_omega.py) is small but not zero.  Asserted, each as
    ONE condition: benchmark budgets: omega >= 0.5 omega*; Morbit's defaults: what the device's step achieves, measured in the
    REFERENCE's direction r* = f(x) - ideal point*, is at least a quarter of omega*_default.  The achieved ratios are printed and, with
    MRBF_TEST_WRITE_RATIOS=1, appended to gpurun_out/ps_omega_ratios.txt (promoted to profiles/r05_ps_omega.txt)."""
    import importlib.util
    import json
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    gold = json.load(open(os.path.join(here, "golden", "ps_omega.json")))[name]
    spec = importlib.util.spec_from_file_location("make_ps_omega", os.path.join(here, "golden", "make_ps_omega.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    C, Y, kernel, x, half = mk.problems()[name]
    d = C.shape[1]
    mod = pkg.update_model(pkg.RbfConfig(kernel=kernel), C, Y)
    ev = lambda X: pkg.eval_models_at_sites(mod, None, X)
    lb, ub = np.maximum(x - half, 0.0), np.minimum(x + half, 1.0)
    fx = ev(x[None, :])[0]
    assert np.allclose(fx, gold["mx"], rtol=0, atol=1e-8 * max(1.0, np.abs(fx).max()))     # the same model as the oracle's
    assert 0 < gold["omega_bench"] < 0.05 and gold["omega_default"] > 0
    bench = ps.PascolettiSerafiniConfig(reference_point=[-1.0, -1.0], max_ps_problem_evals=50 * (d + 1), max_ps_polish_evals=100 * (d + 1),
                                        ps_polish_algo="LD_MMA")
    st = {}
    omega, (xt, mt, sl) = ps.get_criticality_device(bench, mod, x, x, fx, lb, ub, seed=41, stats=st)
    _check_ps_contract(ev, None, x, lb, ub, fx, omega, xt, mt, st)
    assert omega >= 0.5 * gold["omega_bench"], (omega, gold["omega_bench"])
    assert omega <= gold["omega_bench"] * (1 + 1e-6) + 1e-9, "the device found more than the reference solver: regenerate the golden file"
    st2 = {}
    omega2, rest = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=42, stats=st2)[:2]
    assert st2["status"] == 0
    _check_ps_contract(ev, None, x, lb, ub, fx, omega2, rest[0], rest[1], st2)
    r_gold = np.array(gold["r_default"])
    omega2_ref_units = float(-np.max((rest[1] - fx) / r_gold))      # the step's tau in the reference's direction
    assert omega2 > 0 and omega2_ref_units >= 0.25 * gold["omega_default"], (omega2, omega2_ref_units, gold["omega_default"], st2["r"], r_gold)
    line = ("%-6s n=%d d=%d: benchmark budgets omega %.5f / omega* %.5f = %.3f in %.1f ms (%d + %d evaluations); defaults omega %.5f (own r), %.5f in "
            "the reference's direction / omega* %.5f = %.3f in %.1f ms (%d evaluations)"
            % (name, C.shape[0], d, omega, gold["omega_bench"], omega / gold["omega_bench"], st["ms_total"], st["evals_ps"], st["evals_polish"],
               omega2, omega2_ref_units, gold["omega_default"], omega2_ref_units / gold["omega_default"], st2["ms_total"],
               st2["evals_ideal"] + st2["evals_ps"]))
    print(line)
    _append_ratio_line(line)
    mod.free()


def _append_ratio_line(line):
    import os

    if os.environ.get("MRBF_TEST_WRITE_RATIOS"):      # opt-in side effect: the ratio lines for profiles/
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "ps_omega_ratios.txt"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass


@pytest.mark.gpu
def test_ps_step_container_with_two_models_and_constraints():
    """mrbf_ps_step_problem through the routed `get_criticality_container`: objectives from two grouped models, a modelled inequality
    constraint (a row of the first model), a modelled equality constraint and a linear inequality of the MOP; compared with the
    reference method (host loop) on the same container."""
    rng = np.random.default_rng(12)
    d, n = 6, 400
    C = rng.random((n, d))
    Ya = np.stack([np.sum((C - 0.3) ** 2, axis=1), C[:, 0] - 0.52, C[:, 2] - C[:, 4]], axis=1)     # objective 0 | g(x) <= 0 | h(x) = 0
    Yb = np.sum((C - 0.7) ** 2, axis=1, keepdims=True)                                               # objective 1
    ma = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Ya)
    mb = pkg.update_model(pkg.RbfConfig(kernel="multiquadric"), C, Yb)
    R = pkg.surrogates.RefSurrogate
    sc = pkg.surrogates.SurrogateContainer(objectives=[R(ma, [0]), R(mb, [0])], nl_ineq_constraints=[R(ma, [1])], nl_eq_constraints=[R(ma, [2])])
    x = np.array([0.5, 0.85, 0.4, 0.2, 0.4, 0.5])
    lb, ub = x - 0.12, x + 0.12
    A, b = np.array([[0.0, 1.0, 0.0, 1.0, 0.0, 0.0]]), np.array([1.0])        # x1 + x3 <= 1
    lin = (None, None, A, b)
    ev = lambda X: pkg.surrogates.eval_container_objectives_at_scaled_sites(sc, None, X)
    fx = ev(x[None, :])[0]
    cfgps = ps.PascolettiSerafiniConfig()
    stats = {}
    omega, (xt, mt, sl) = ps.get_criticality_container(cfgps, sc, None, x, x, fx, lb, ub, lin=lin, seed=5, stats=stats, eq_tol=0.02)
    assert stats["path"] == "device"
    g = lambda X: pkg.surrogates.eval_container_nl_ineq_constraints_at_scaled_sites(sc, None, X)
    h = pkg.surrogates.eval_container_nl_eq_constraints_at_scaled_sites(sc, None, xt[None, :])[0]
    _check_ps_contract(ev, g, x, lb, ub, fx, omega, xt, mt, stats, lin=lin)
    assert np.all(np.abs(h) <= 0.02 + 1e-12), h                                 # equality within the tolerance the problem states
    assert np.all(mt < fx)
    stats2 = {}
    omega2, (xt2, _, _) = ps.get_criticality_container(cfgps, sc, None, x, x, fx, lb, ub, lin=lin, seed=5, stats=stats2, eq_tol=0.02)
    assert omega2 == omega and np.array_equal(xt2, xt)
    # the reference method on the same container and direction
    cfg_r = ps.PascolettiSerafiniConfig(reference_direction=list(stats["r"]))
    plan_models = pkg.surrogates.container_plan(sc)["models"]
    assert plan_models == [ma, mb]
    import morbit.jl_amd.pascoletti_serafini as psm
    real = psm._ps_step_problem
    psm._ps_step_problem = lambda *a, **k: (-2, None)                         # the device refuses -> reference method
    try:
        st_h = {}
        omega_h, _ = ps.get_criticality_container(cfg_r, sc, None, x, x, fx, lb, ub, lin=lin, seed=3, stats=st_h, eq_tol=0.02)[:2]
    finally:
        psm._ps_step_problem = real
    assert st_h["path"] == "reference"
    omega_d, _ = ps.get_criticality_container(cfg_r, sc, None, x, x, fx, lb, ub, lin=lin, seed=7, eq_tol=0.02)[:2]
    assert omega_d >= 0.8 * omega_h, (omega_d, omega_h)
    print("PS step, 2 models + 3 constraints, d=%d: %.2f ms on the device, omega %.4f (reference method %.4f)" % (d, stats["ms_total"], omega_d, omega_h))
    ma.free()
    mb.free()


@pytest.mark.gpu
def test_ps_step_device_critical_and_argument_errors():
    rng = np.random.default_rng(1)
    C = rng.random((150, 3))
    Y = np.sum((C - 0.5) ** 2, axis=1, keepdims=True)       # one objective, minimiser inside the box
    mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
    ev = lambda X: pkg.eval_models_at_sites(mod, None, X)
    # the model's own minimiser over the box is (numerically) x itself when f(x_n) is set to the model minimum: r <= 0 -> critical
    x = np.full(3, 0.5)
    fx = ev(x[None, :])[0] - 1.0                             # pretend the true value is below every model value in the box
    out = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, x - 0.05, x + 0.05, seed=1)
    assert out[0] == 0 and np.array_equal(out[1], x) and out[3] == 0 and np.allclose(out[2], ev(x[None, :])[0], atol=1e-13)
    from morbit.jl_amd import _lib
    import ctypes
    ctx = mod.ctx
    opts, info = _lib.PsOptions(-1, -1, 0, 0, 1, -0.5, 1e-3), _lib.PsInfo()
    buf = np.zeros(3)
    args = lambda **kw: [kw.get("x", _lib.as_ptr(x)), kw.get("lb", _lib.as_ptr(x - 0.1)), kw.get("ub", _lib.as_ptr(x + 0.1)), _lib.as_ptr(fx), None]
    f = ctx.lib.mrbf_ps_step
    assert f(ctx.h, None, *args(), ctypes.byref(opts), _lib.as_ptr(buf), _lib.as_ptr(buf), None, ctypes.byref(info)) == -2
    assert f(ctx.h, mod.model, *args(x=None), ctypes.byref(opts), _lib.as_ptr(buf), _lib.as_ptr(buf), None, ctypes.byref(info)) == -3
    assert f(ctx.h, mod.model, *args(lb=_lib.as_ptr(x + 1.0)), ctypes.byref(opts), _lib.as_ptr(buf), _lib.as_ptr(buf), None, ctypes.byref(info)) == -4
    bad = _lib.PsOptions(-1, -1, 0, 0, 1, 0.5, 1e-3)
    assert f(ctx.h, mod.model, *args(), ctypes.byref(bad), _lib.as_ptr(buf), _lib.as_ptr(buf), None, ctypes.byref(info)) == -8
    mod.free()


def _rank_fixtures(lam, seed):
    """populations for one ranking: (name, f, phi)"""
    rng = np.random.default_rng(seed)
    out = []
    f = np.sort(rng.random(lam))
    # (a) a CONSISTENT population (feasible individuals ranked by f on top, a tenth infeasible at the bottom with violation and
    #     objective rising together), shuffled inside windows of eight: ranked after a few phases -- a true fixed point, the no-swap
    #     exit is taken at the first test after it
    nin = lam // 10
    fa, pa = f.copy(), np.zeros(lam)
    pa[lam - nin:] = np.sort(rng.random(nin)) + 0.1
    perm = np.arange(lam)
    for s in range(0, lam - 8, 8):
        perm[s:s + 8] = rng.permutation(perm[s:s + 8])
    out.append(("consistent-window-shuffled", fa[perm], pa[perm]))
    # (b) ranked, but the last individual is infeasible with the BEST objective value: by phi it belongs where it is, by f (drawn
    #     with probability 0.45 per comparison) it moves up and drifts back -- sixteen quiet phases happen now and then while it sits
    #     at the bottom and are NOT a fixed point; 256 quiet phases practically never are taken
    fb, pb = f.copy(), np.zeros(lam)
    fb[-1], pb[-1] = -1.0, 0.5
    out.append(("one-dissenter-at-the-bottom", fb, pb))
    # (c) a random generation with a third infeasible and a few individuals outside the budget (phi = inf): no early exit
    fc, pc = rng.random(lam), np.where(rng.random(lam) < 0.33, rng.random(lam), 0.0)
    fc[-7:], pc[-7:] = np.inf, np.inf
    out.append(("random-third-infeasible", fc, pc))
    # (d) heavy ties in both keys (objective values on a grid of 100, violations on a grid of 10): the wave kernel compares RANKS of
    #     f and phi -- equal values must stay equal there, or a tie would turn into a swap
    fd, pd = np.round(rng.random(lam), 2), np.where(rng.random(lam) < 0.4, np.round(rng.random(lam), 1), 0.0)
    out.append(("ties-in-both-keys", fd, pd))
    # (e) a NaN objective on an infeasible individual: NaN has no rank, the generation must stay in the one-workgroup kernel -- and
    #     every path must still give that kernel's order
    fe, pe = fc.copy(), pc.copy()
    bad = int(np.argmax(pe > 0))
    fe[bad] = np.nan
    out.append(("nan-objective", fe, pe))
    return out


# (lam, seed) for which, on fixture (b), a no-swap test every sixteen phases leaves early with another order than the test every 256
RULE_SENSITIVE = {1024: 35, 1320: 24}


def test_rank_oracle_quiet_stretch_is_not_a_fixed_point():
    """CPU: the plain NumPy ranking (oracle/ps_rank_oracle.py).  Deterministic comparisons end in the sorted order; a consistent
    population is a fixed point whatever the test distance; with one dissenting individual the place of the no-swap test changes the
    order -- the ADVICE r5 finding, and the reason the two device kernels must share one rule per population size."""
    from oracle import ps_rank_oracle as pro

    f = np.random.default_rng(0).random(300)
    order, phases = pro.stochastic_rank(f, np.zeros(300), seed=5, gen=2)
    assert np.array_equal(order, np.argsort(f, kind="stable")) and phases <= 300
    for lam, seed in RULE_SENSITIVE.items():
        fx = _rank_fixtures(lam, seed=lam)
        o256, p256 = pro.stochastic_rank(fx[0][1], fx[0][2], seed=seed, gen=3)
        o16, p16 = pro.stochastic_rank(fx[0][1], fx[0][2], seed=seed, gen=3, quiet=16)
        assert p16 < p256 < lam and np.array_equal(o256, o16)            # fixed point: same order, both leave early
        o256, p256 = pro.stochastic_rank(fx[1][1], fx[1][2], seed=seed, gen=3)
        o16, p16 = pro.stochastic_rank(fx[1][1], fx[1][2], seed=seed, gen=3, quiet=16)
        assert p16 < lam and p256 == lam and not np.array_equal(o256, o16), (lam, p16, p256)
        assert sorted(o256) == list(range(lam))


@pytest.mark.gpu
@pytest.mark.parametrize("lam", [1024, 1027, 1320, 2600, 5160, 7168])   # (1027: odd, a short last group of phases; 7168: the population limit)
def test_ps_ranking_kernels_share_one_exit_rule(lam):
    """the one-workgroup ranking used to leave after sixteen quiet phases, the sixteen-workgroup ranking only at its
    256-phase chunk boundaries -- different orders whenever the exit is taken early, e.g. after a counter time-out on a shared device.
    Now one rule per population size.  Crafted generations (nearly ranked, a few infeasible individuals at the parent boundary: the
    exit IS taken early) through mrbf_debug_ps_rank: one workgroup == one wave per 64 individuals on as many compute units (round 6,
    impl 1) == that kernel giving up at once == sixteen workgroups (round 5's form, impl 6) ==
    the plain NumPy loop of
    oracle/ps_rank_oracle.py, entry by entry."""
    import ctypes

    from morbit.jl_amd import _lib
    from oracle import ps_rank_oracle as pro

    ctx = pkg.Context()
    try:
        took_early = 0
        for seed, gen in ((RULE_SENSITIVE.get(lam, 11), 3), (2 ** 40 + 17, 0)):
            for name, f, phi in _rank_fixtures(lam, seed=lam):
                want, phases = pro.stochastic_rank(f, phi, seed=seed, gen=gen)
                took_early += phases < lam
                got = {}
                for impl in (0, 1, 2, 6):
                    order = np.empty(lam, dtype=np.int32)
                    gave_up = ctypes.c_int32(-1)
                    ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(phi), seed, gen, impl,
                                                         order.ctypes.data_as(_lib.c_ip), ctypes.byref(gave_up)))
                    # (the NaN generation is never handed over, so there is nothing to give up on)
                    assert gave_up.value == (1 if impl == 2 and name != "nan-objective" else 0), (name, impl, gave_up.value)
                    got[impl] = order
                for impl in (0, 1, 2, 6):
                    assert np.array_equal(got[impl], want), (lam, name, seed, gen, impl, int(np.argmax(got[impl] != want)))
        assert took_early >= 2, took_early     # the fixtures do exercise the early exit
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lam", [66, 130, 280, 523, 700, 1000, 1023])
def test_ps_ranking_of_small_populations_in_registers(lam):
    """Populations below 1024 (d < 50: Morbit's own problem sizes) are ranked inside ps_rank_kernel's one workgroup.  Round 6: a wave
    per 96 individuals with the records (rank keys + index) in registers, sixteen phases between workgroup barriers, the draws and
    keys made by all threads (rank_small_waves) -- instead of one pair per thread through LDS with a barrier per phase (impl 9).
    Same comparisons, draws and exit rule (a no-swap test every sixteen phases): both orders == the NumPy oracle's, entry by entry,
    on the fixtures that take the early exit, on ties in both keys, and with a NaN (which stays in the old loop)."""
    from morbit.jl_amd import _lib
    from oracle import ps_rank_oracle as pro

    ctx = pkg.Context()
    try:
        took_early = 0
        for seed, gen in ((11, 3), (2 ** 40 + 17, 0)):
            for name, f, phi in _rank_fixtures(lam, seed=lam):
                want, phases = pro.stochastic_rank(f, phi, seed=seed, gen=gen)
                took_early += phases < lam
                for impl in (0, 9):
                    order = np.empty(lam, dtype=np.int32)
                    ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(phi), seed, gen, impl, order.ctypes.data_as(_lib.c_ip), None))
                    assert np.array_equal(order, want), (lam, name, seed, gen, impl, int(np.argmax(order != want)))
        assert took_early >= 2, took_early
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lam", [130, 400, 700, 1000, 1320, 2600, 5160, 7000])   # (400, 1000: launches of N threads -- one element per thread)
def test_ps_plain_sort_in_registers_is_the_sort(lam):
    """A generation without infeasible individuals is ranked by a bitonic network (ps_rank_kernel).  Round 6: E = 2 / 4 / 8 elements per
    thread in registers -- compare-exchanges inside the thread, through wave shuffles, and only the widest strides through LDS.  Against
    NumPy's stable order (objective, ties by index; individuals outside the budget -- f = phi = inf -- last) and against the
    one-pair-per-thread form it replaces (impl 3): identical orders, also with duplicated objective values.  Populations of 1024 and
    more are ranked in the step by counting instead (impl 1: rank(i) = #{j : f_j < f_i, or equal and j < i} over the whole chip):
    the same order again."""
    import ctypes

    from morbit.jl_amd import _lib

    ctx = pkg.Context()
    try:
        rng = np.random.default_rng(lam)
        for case in range(3):
            f = rng.random(lam)
            phi = np.zeros(lam)
            if case == 1:
                f = np.round(f, 2)                       # many ties: the index decides
            if case == 2:
                f[-9:], phi[-9:] = np.inf, np.inf        # outside the budget
                f[: lam // 3] = f[lam // 3: 2 * (lam // 3)]
            want = np.lexsort((np.arange(lam), f))
            got = {}
            for impl in (0, 3) + ((1,) if lam >= 1024 else ()):
                order = np.empty(lam, dtype=np.int32)
                ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(phi), 5, 1, impl, order.ctypes.data_as(_lib.c_ip), None))
                got[impl] = order
            assert np.array_equal(got[0], want), (lam, case, int(np.argmax(got[0] != want)))
            assert np.array_equal(got[3], want), (lam, case)
            if lam >= 1024:     # the step's own path for such populations: positions by counting on the whole chip (ps_rank_prep_kernel)
                assert np.array_equal(got[1], want), (lam, case, int(np.argmax(got[1] != want)))
            # the step itself only reads the mu = ceil(lam / 7) parents: they are found by a sampled threshold + compaction and ranked
            # alone (select_parents) -- the same parents in the same order as the full sort's prefix (impl 5), whatever the ties
            mu = (lam + 6) // 7
            for impl in (4, 5):
                order = np.empty(lam, dtype=np.int32)
                ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(phi), 5, 1, impl, order.ctypes.data_as(_lib.c_ip), None))
                assert np.array_equal(order[:mu], want[:mu]), (lam, case, impl, int(np.argmax(order[:mu] != want[:mu])))
                assert np.all(order[mu:] == -1)
        # massive ties (every objective value equal: the PS run's t = 0 individuals): the threshold takes everything, the selection
        # steps aside for the full sort -- same parents
        f = np.zeros(lam)
        phi = np.zeros(lam)
        for impl in (4, 5):
            order = np.empty(lam, dtype=np.int32)
            ctx.check(ctx.lib.mrbf_debug_ps_rank(ctx.h, lam, _lib.as_ptr(f), _lib.as_ptr(phi), 5, 1, impl, order.ctypes.data_as(_lib.c_ip), None))
            assert np.array_equal(order[: (lam + 6) // 7], np.arange((lam + 6) // 7)), (lam, impl)
    finally:
        ctx.close()
