"""CPU tests of the bindings' decision table (include/mrbf.h: mrbf_dispatch_*; pure host code of libmrbf) and of the routing the
Python mirror builds on it -- the same calls morbit.jl_amd/julia/HipRbf.jl makes, so what is pinned here is what a Morbit run with
`HipRbfConfig` does: which of `_backtrack`, `get_criticality(::PascolettiSerafiniConfig)`, the affine filter scan, `_rbf_round4`
and `update_model` reach a device entry point and which run Morbit's own method (here: the host mirrors).  No GPU is needed: device
models are test doubles that evaluate through the oracle, device entry points are replaced by recorders."""
import os
import re

import numpy as np
import pytest

import morbit.jl_amd as pkg
from morbit.jl_amd import _lib, descent, rbf_model, sampling, surrogates
from morbit.jl_amd import pascoletti_serafini as ps
from oracle import rbf_oracle as orc
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def test_decision_table(lib):
    D, R = _lib.DISPATCH_DEVICE, _lib.DISPATCH_REFERENCE
    # Pascoletti-Serafini: every BASELINE dimension is on the device (C3 d = 64, C4 d = 128 -- examples/example_zdt.jl:39 --, C5 d = 256)
    for d in (2, 12, 64, 128, 256, 356):
        assert lib.mrbf_dispatch_ps(d, 2, 1, 0, 0, 0) == D
    assert lib.mrbf_dispatch_ps(357, 2, 1, 0, 0, 0) == R                 # population 20 (d + 2) beyond one workgroup's LDS
    assert lib.mrbf_dispatch_ps(64, 8, 1, 0, 0, 0) == D and lib.mrbf_dispatch_ps(64, 9, 1, 0, 0, 0) == R
    assert lib.mrbf_dispatch_ps(64, 2, 3, 5, 7, 0) == D                  # several grouped models, modelled + linear constraints
    assert lib.mrbf_dispatch_ps(64, 2, 9, 0, 0, 0) == R and lib.mrbf_dispatch_ps(64, 2, 1, 33, 0, 0) == R
    assert lib.mrbf_dispatch_ps(64, 2, 1, 0, 0, 1) == R                  # a CompositeSurrogate / ExactModel in the container
    assert lib.mrbf_dispatch_ps(64, 0, 0, 0, 0, 0) == R
    # backtracking
    assert lib.mrbf_dispatch_backtrack(1, 0, 1) == D
    assert lib.mrbf_dispatch_backtrack(2, 0, 1) == R and lib.mrbf_dispatch_backtrack(1, 1, 1) == R and lib.mrbf_dispatch_backtrack(1, 0, 0) == R
    assert lib.mrbf_dispatch_backtrack(0, 1, 0) == R                     # only an ExactModel: Morbit's own loop
    # affine scan
    assert lib.mrbf_dispatch_affine(10, 5) == R and lib.mrbf_dispatch_affine(20000, 24) == D and lib.mrbf_dispatch_affine(0, 5) == R
    assert lib.mrbf_dispatch_affine(255, 128) == D and lib.mrbf_dispatch_affine(100, 128) == R and lib.mrbf_dispatch_affine(500, 48) == R
    # round 4: the start set must be able to carry the tail
    assert lib.mrbf_dispatch_round4(4, 3, 1, 50) == D and lib.mrbf_dispatch_round4(3, 3, 1, 50) == R
    assert lib.mrbf_dispatch_round4(1, 3, -1, 50) == D and lib.mrbf_dispatch_round4(1, 3, 0, 50) == D and lib.mrbf_dispatch_round4(4, 3, 1, 0) == R
    # fit: factor reuse iff the kept state is exactly the training set and the start set is unisolvent
    F, K = _lib.FIT_FULL, _lib.FIT_FROM_ROUND4
    assert lib.mrbf_dispatch_fit(10, 4, 4, 6, 1) == K and lib.mrbf_dispatch_fit(1024, 65, 65, 959, 1) == K
    assert lib.mrbf_dispatch_fit(2145, 65, 65, 2080, 1) == F      # large training sets: the ordinary fit is as fast and more accurate
    assert lib.mrbf_dispatch_fit(10, 4, 4, 6, 0) == F and lib.mrbf_dispatch_fit(11, 4, 4, 6, 1) == F
    assert lib.mrbf_dispatch_fit(10, 5, 4, 5, 1) == F and lib.mrbf_dispatch_fit(4, 4, 4, 0, 1) == F and lib.mrbf_dispatch_fit(10, 0, 4, 0, 1) == F
    # return codes that mean "take the reference method", not an error
    assert lib.mrbf_dispatch_after(_lib.ENTRY_ROUND4, -2) == 1 and lib.mrbf_dispatch_after(_lib.ENTRY_ROUND4, _lib.MRBF_ESINGULAR) == 1
    assert lib.mrbf_dispatch_after(_lib.ENTRY_ROUND4, _lib.MRBF_EHIP) == 0 and lib.mrbf_dispatch_after(_lib.ENTRY_ROUND4, 0) == 0
    # the size limits of mrbf_round4 itself never raise in a binding: asked beforehand they say "reference", met afterwards too
    assert lib.mrbf_dispatch_round4(70, 64, 1, 30000) == D and lib.mrbf_dispatch_round4(70, 64, 1, 30001) == R
    assert lib.mrbf_dispatch_round4(2000, 1024, 1, 100) == D and lib.mrbf_dispatch_round4(2000, 1025, 1, 100) == R
    for rc in (-3, -5, _lib.MRBF_ENOMEM):
        assert lib.mrbf_dispatch_after(_lib.ENTRY_ROUND4, rc) == 1
    assert lib.mrbf_dispatch_after(_lib.ENTRY_FIT_FROM_ROUND4, _lib.MRBF_ENOTPD) == 1
    assert lib.mrbf_dispatch_after(_lib.ENTRY_PS_STEP, -2) == 1 and lib.mrbf_dispatch_after(_lib.ENTRY_PS_STEP, -4) == 0
    assert lib.mrbf_dispatch_after(_lib.ENTRY_BACKTRACK, -2) == 0


def test_julia_binding_uses_the_same_table():
    """HipRbf.jl cannot run here; what can be pinned is that it takes every routing decision from the library and never raises on a
    size limit: each dispatch function is bound, every device call that can refuse is followed by the fallback test."""
    src = open(os.path.join(ROOT, "morbit.jl_amd", "julia", "HipRbf.jl"), encoding="utf-8").read()
    for fn in ("mrbf_dispatch_ps", "mrbf_dispatch_backtrack", "mrbf_dispatch_affine", "mrbf_dispatch_round4", "mrbf_dispatch_fit", "mrbf_dispatch_after"):
        assert "(:%s, libmrbf)" % fn in src, fn
    for routed in ("_dispatch_ps(", "_dispatch_backtrack(", "_dispatch_affine(", "_dispatch_round4(", "_dispatch_fit("):
        assert len(re.findall(re.escape(routed), src)) >= 2, routed      # definition + use
    assert len(re.findall(r"_fallback_rc\(", src)) >= 4                     # definition + round 4, factor reuse, PS step
    assert "invoke(get_criticality" in src and "invoke(_backtrack" in src and "invoke(Base.iterate" in src
    assert "mrbf_ps_step_problem" in src and "error(" not in src.split("# ---- descent consumers")[1]
    # struct mirrors: field order of include/mrbf.h
    m = re.search(r"struct MrbfPsProblem.*?\nend", src, flags=re.S).group(0)
    assert re.sub(r"\s+", " ", m).startswith("struct MrbfPsProblem # mirrors mrbf_ps_problem, 72 bytes n_models::Int32; n_objectives::Int32 "
                                              "models::Ptr{Ptr{Cvoid}}; roles::Ptr{Int32} n_lin_eq::Int32; n_lin_ineq::Int32 A_eq::Ptr{Float64}")


class DeviceModelDouble(rbf_model.RbfModel):
    """an `RbfModel` (so the plan counts it as a device model) whose sweeps go through the oracle"""

    def __init__(self, ref):
        super().__init__(None, None, ref.C.shape[0], ref.C.shape[1], ref.num_outputs, 0, True)
        self.ref, self.sweeps = ref, 0

    def eval_sites(self, X, want_values=True, want_jac=False, **kw):
        self.sweeps += 1
        X = np.atleast_2d(X)
        return (self.ref.values(X) if want_values else None), (self.ref.jacs(X) if want_jac else None)


class Outer:
    num_outputs = 1

    def eval(self, xi):
        return np.array([xi[-1] ** 2])

    def jacobian(self, xi):
        J = np.zeros((1, xi.size))
        J[0, -1] = 2 * xi[-1]
        return J


@pytest.fixture()
def models():
    rng = np.random.default_rng(0)
    C = rng.random((40, 3))
    Y = np.stack([((C - 0.3) ** 2).sum(1), ((C - 0.7) ** 2).sum(1), C[:, 0] - 0.55], axis=1)
    m3 = DeviceModelDouble(orc.fit(C, Y, 0, 3.0, 0.0, 1))
    m1 = DeviceModelDouble(orc.fit(C, Y[:, 1:2], 0, 3.0, 0.0, 1))
    return m3, m1


def test_container_plan(models):
    m3, m1 = models
    R = surrogates.RefSurrogate
    sc = surrogates.SurrogateContainer(objectives=[R(m3, [0]), R(m1, [0])], nl_ineq_constraints=[R(m3, [2])])
    plan = surrogates.container_plan(sc)
    assert plan["models"] == [m3, m1] and plan["roles"] == [0, _lib.ROLE_NONE, _lib.ROLE_INEQ, 1]
    assert (plan["k"], plan["n_con"], plan["n_foreign"], plan["in_order"]) == (2, 1, 0, False)
    # objectives = the outputs of one model in order -> the single-model entry points apply (constraints elsewhere do not matter
    # to backtracking)
    sc1 = surrogates.SurrogateContainer(objectives=[R(m1, [0])], nl_eq_constraints=[R(m3, [2])])
    assert surrogates.container_plan(sc1, objectives_only=True)["in_order"] and not surrogates.container_plan(sc1)["in_order"]
    assert descent._single_model(sc1) is m1
    assert descent._single_model(surrogates.SurrogateContainer(objectives=[R(m3, [1, 0, 2])])) is None      # out of order
    assert descent._single_model(sc) is None                                                                    # two models
    # foreign: composite surrogate, a non-device model, a row used twice
    comp = surrogates.CompositeSurrogate(m3, Outer(), [0])
    assert surrogates.container_plan(surrogates.SurrogateContainer(objectives=[comp]))["n_foreign"] == 1
    assert descent._single_model(surrogates.SurrogateContainer(objectives=[comp])) is None
    assert surrogates.container_plan(surrogates.SurrogateContainer(objectives=[R(object(), [0])]))["n_foreign"] == 1
    assert surrogates.container_plan(surrogates.SurrogateContainer(objectives=[R(m3, [0])], nl_ineq_constraints=[R(m3, [0])]))["n_foreign"] == 1


def test_ps_routing_device_vs_reference(models, monkeypatch):
    m3, m1 = models
    R = surrogates.RefSurrogate
    x = np.array([0.5, 0.85, 0.5])
    lb, ub = x - 0.15, x + 0.15
    calls = []

    def fake_device(desc_cfg, mods, roles, k, *a, lin=None, seed=0, stats=None, eq_tol=-1.0):
        calls.append((list(mods), list(roles), k, lin))
        return fake_device.rc, ((0.25, (x.copy(), np.zeros(k), 0.0)) if fake_device.rc == 0 else None)

    class Ctx:
        def check(self, rc):
            raise pkg.MrbfError(rc, "device error")

    m3.ctx = Ctx()
    monkeypatch.setattr(ps, "_ps_step_problem", fake_device)
    cfg = ps.PascolettiSerafiniConfig(max_ps_problem_evals=600, max_ideal_point_problem_evals=400)
    sc = surrogates.SurrogateContainer(objectives=[R(m3, [0]), R(m1, [0])], nl_ineq_constraints=[R(m3, [2])])
    fx = surrogates.eval_container_objectives_at_scaled_site(sc, None, x)
    A = np.array([[1.0, 1.0, 0.0]])
    # 1. the device solver takes a container with two grouped models, a modelled and a linear constraint
    fake_device.rc = 0
    stats = {}
    out = ps.get_criticality_container(cfg, sc, None, x, x, fx, lb, ub, lin=(None, None, A, np.array([1.4])), stats=stats)
    assert out[0] == 0.25 and len(calls) == 1 and calls[0][0] == [m3, m1] and calls[0][1] == [0, -1, -3, 1] and calls[0][2] == 2
    # 2. a refusal of the device call (-2) is not an error: the reference method runs on the same arguments
    fake_device.rc = -2
    sweeps0 = m3.sweeps
    stats = {}
    omega, (xt, mt, sl) = ps.get_criticality_container(cfg, sc, None, x, x, fx, lb, ub, lin=(None, None, A, np.array([1.4])), seed=3, stats=stats)
    assert stats["path"] == "reference" and m3.sweeps > sweeps0 and len(calls) == 2
    assert omega > 0 and np.all(mt < fx) and np.all(xt >= lb) and np.all(xt <= ub)
    assert xt[0] - 0.55 <= 1e-8 + 1e-3 and xt[0] + xt[1] <= 1.4 + 1e-8     # modelled (m3 row 2 ~ x0 - 0.55) and linear constraint hold
    # 3. any other code is an error
    fake_device.rc = _lib.MRBF_EHIP
    with pytest.raises(pkg.MrbfError):
        ps.get_criticality_container(cfg, sc, None, x, x, fx, lb, ub)
    # 4. a CompositeSurrogate objective never reaches the device
    n = len(calls)
    comp = surrogates.CompositeSurrogate(m3, Outer(), [0])
    sc2 = surrogates.SurrogateContainer(objectives=[comp, R(m1, [0])])
    fx2 = surrogates.eval_container_objectives_at_scaled_site(sc2, None, x)
    stats = {}
    out = ps.get_criticality_container(cfg, sc2, None, x, x, fx2, lb, ub, seed=1, stats=stats)
    assert len(calls) == n and stats["path"] == "reference" and out[0] >= 0


def test_round4_and_fit_routing(monkeypatch):
    rng = np.random.default_rng(5)
    d = 3
    sites = rng.random((30, d))
    values = (sites ** 2).sum(1, keepdims=True)
    cfg = pkg.RbfConfig(kernel="cubic")
    x = sites[0]
    lb, ub = np.zeros(d), np.ones(d)
    log = []

    class State:
        def __init__(self, n0, accepted, handle=1):
            self.start_sites, self.accepted, self.handle, self.ctx, self.freed = np.zeros((n0, d)), list(accepted), handle, None, False

        def free(self):
            self.freed = True

    def fake_round4(cfg_, C0, Xc, delta=1.0, ctx=None, keep_state=False, rc_only=False):
        log.append(("round4", C0.shape[0], Xc.shape[0], keep_state))
        st = State(C0.shape[0], [2, 0, 5]) if keep_state else None
        return fake_round4.rc, ([2, 0, 5] if fake_round4.rc == 0 else []), (st if fake_round4.rc == 0 else None)

    def fake_fit_from(state, Y, fully_linear=False, rc_only=False):
        log.append(("fit_from_round4", Y.shape[0]))
        return fake_fit_from.rc, ("kept-factor model" if fake_fit_from.rc == 0 else None)

    def fake_update(cfg_, S, Y, delta=1.0, fully_linear=False, ctx=None):
        log.append(("fit", S.shape[0]))
        return "full model"

    monkeypatch.setattr(sampling, "rbf_round4_device", fake_round4)
    monkeypatch.setattr(sampling, "fit_from_round4", fake_fit_from)
    monkeypatch.setattr(rbf_model, "update_model", fake_update)
    keeper = sampling.Round4Keeper()
    start = [0, 1, 2, 3]                                   # n0 = q = d + 1: unisolvent
    cand = sampling.results_in_box_indices(sites, lb, ub, start)
    # device selection, factor kept, fit from the kept factor
    fake_round4.rc, fake_fit_from.rc = 0, 0
    st = {}
    r4 = sampling._rbf_round4(sites, lb, ub, x, 1.0, start, cfg, keeper=keeper, db_key="db", stats=st)
    assert st["round4"] == "device" and r4 == [cand[2], cand[0], cand[5]] and log[-1] == ("round4", 4, len(cand), True)
    st = {}
    assert sampling.update_model_from_selection(cfg, sites, values, start + r4, keeper=keeper, db_key="db", stats=st) == "kept-factor model"
    assert st["fit"] == "from_round4" and "db" not in keeper.kept
    # the training set changed after round 4 (e.g. another site added): the kept factor is dropped, full fit
    r4 = sampling._rbf_round4(sites, lb, ub, x, 1.0, start, cfg, keeper=keeper, db_key="db")
    st = {}
    assert sampling.update_model_from_selection(cfg, sites, values, start + r4 + [29], keeper=keeper, db_key="db", stats=st) == "full model"
    assert st["fit"] == "full" and log[-1] == ("fit", 8)
    # the device refuses the factor reuse (-2 / not p.d.): full fit, no error; a HIP error is an error
    r4 = sampling._rbf_round4(sites, lb, ub, x, 1.0, start, cfg, keeper=keeper, db_key="db")
    fake_fit_from.rc = _lib.MRBF_ENOTPD
    assert sampling.update_model_from_selection(cfg, sites, values, start + r4, keeper=keeper, db_key="db") == "full model"
    # a start set larger than q: selection on the device, but the kept factor does not cover the start set's own null space -> full fit
    start5 = [0, 1, 2, 3, 4]
    fake_fit_from.rc = 0
    r4 = sampling._rbf_round4(sites, lb, ub, x, 1.0, start5, cfg, keeper=keeper, db_key="db")
    n_before = len(log)
    assert sampling.update_model_from_selection(cfg, sites, values, start5 + r4, keeper=keeper, db_key="db") == "full model"
    assert [e[0] for e in log[n_before:]] == ["fit"]
    # a start set that cannot carry the tail (n0 < q) never reaches the device: Morbit's own bookkeeping (host mirror, oracle kernel)
    kid, a, b = rbf_model._get_kernel_params(1.0, cfg)
    n_before = len(log)
    st = {}
    got = sampling._rbf_round4(sites, lb, ub, x, 1.0, [0, 1], cfg, kernel_block=lambda X, C: orc.phi(kid, a, b, orc.pairwise_dist(np.atleast_2d(X), np.atleast_2d(C))), stats=st)
    assert st["round4"] == "reference" and len(log) == n_before and isinstance(got, list)
    # a rank-deficient start set is found out by the device call: ESINGULAR -> reference method, not an error
    fake_round4.rc = _lib.MRBF_ESINGULAR

    def no_device_blocks(*a, **k):
        raise RuntimeError("reference bookkeeping reached")     # (its kernel blocks would need the GPU here)

    monkeypatch.setattr(rbf_model, "get_matrices", no_device_blocks)
    with pytest.raises(RuntimeError, match="reference bookkeeping reached"):
        sampling._rbf_round4(sites, lb, ub, x, 1.0, start, cfg)


def test_affine_filter_routing(monkeypatch):
    rng = np.random.default_rng(2)
    d = 4
    x = rng.random(d)
    calls = []

    def fake_select(self, Sd, qr, want):       # what mrbf_affine_select does, on the host: scan + one reflector per pick
        calls.append(Sd.shape)
        got = []
        Z = qr.complement(self.p)
        while len(got) < want:
            v = np.abs((Sd @ Z) @ Z.T).max(axis=1)
            b = int(np.argmax(v))
            if not v[b] > self.pivot_val:
                break
            qr.append(Sd[b].copy())
            Sd[b] = 0.0
            Z = qr.complement(self.p)
            got.append(b)
        return got, Z

    monkeypatch.setattr(sampling.AffinelyIndependentPointFilter, "_select_device", fake_select)
    small = [x + 0.1 * rng.standard_normal(d) for _ in range(50)]
    assert len(sampling.AffinelyIndependentPointFilter(x, small, pivot_val=1e-3).collect()) == d and not calls     # host BLAS
    big = [x + 0.1 * rng.standard_normal(d) for _ in range(9000)]
    flt = sampling.AffinelyIndependentPointFilter(x, big, pivot_val=1e-3)
    got = flt.collect()
    assert len(got) == d and calls == [(9000, d)] and flt.Y.shape == (d, d) and flt.Z.shape == (d, 0)              # ONE device call
    ref = sampling.AffinelyIndependentPointFilter(x, big[:50], pivot_val=1e-3)                                     # host path, same rule
    assert ref.collect() == sampling.AffinelyIndependentPointFilter(x, big[:50], pivot_val=1e-3, ctx=object()).collect()


def test_mega_job_tables_are_consistent():
    """The job tables of the persistent factorisation (chol_mega.hip: build_job_tables) are pure host code: for the shapes and schedule
    parameters the library picks by size -- and for degenerate ones -- every tile is finished exactly once, receives exactly the windows
    nbulk_updates promises, queues are ordered.  (Runs under ASan / UBSan with `make -C morbit.jl_amd/csrc asan`.)"""
    import ctypes

    from morbit.jl_amd import _lib

    lib = _lib.load()
    out = (ctypes.c_int64 * 6)()
    seen = {}
    # (nt, mt, slack, slack_chain, first, win, srows, half_cols): potrf_mega_tall's choices for n = 256 .. 16384 (+1 row tile of
    # right-hand sides), and corner cases: one block column, more streamed rows than rows, no slack beyond one
    cases = [(2, 3, 3, 6, 1, 4, 5, 0), (16, 17, 3, 6, 1, 4, 5, 0), (48, 49, 3, 6, 1, 4, 5, 0), (64, 65, 3, 7, 1, 6, 3, 0), (96, 97, 3, 7, 1, 8, 3, 0),
             (128, 129, 3, 6, 1, 8, 2, 0), (1, 1, 1, 1, 1, 1, 0, 0), (1, 2, 3, 6, 1, 4, 5, 0), (5, 9, 1, 1, 1, 1, 7, 2), (33, 33, 2, 2, 4, 4, 0, 3),
             (20, 24, 3, 9, 2, 5, 2, 1)]
    for c in cases:
        assert lib.mrbf_debug_mega_tables(*c, out) == 0, c
        npanel, nbulk, nchain, nwin, checksum, bad = list(out)
        assert bad == 0, (c, list(out))
        nt, mt, srows = c[0], c[1], c[6]
        tiles = sum(mt - cc for cc in range(nt))
        chain_tiles = sum(1 + min(srows, mt - cc - 1) for cc in range(nt))
        assert nchain == chain_tiles and npanel == 2 * (tiles - chain_tiles), (c, list(out))
        seen[c] = checksum
    # deterministic
    for c in cases[:3]:
        assert lib.mrbf_debug_mega_tables(*c, out) == 0 and out[4] == seen[c]
    # refused parameter sets
    assert lib.mrbf_debug_mega_tables(0, 1, 3, 6, 1, 4, 5, 0, out) == -1
    assert lib.mrbf_debug_mega_tables(4, 3, 3, 6, 1, 4, 5, 0, out) == -1
    assert lib.mrbf_debug_mega_tables(4, 5, 3, 2, 1, 4, 5, 0, out) == -3
    assert lib.mrbf_debug_mega_tables(4, 5, 3, 6, 5, 4, 5, 0, out) == -3
    assert lib.mrbf_debug_mega_tables(4, 5, 3, 6, 1, 4, 5, 0, None) == -9
    # round-4 options (per-column streamed rows at the edges, 64-row bulk halves in the last block columns, chain tiles' queues):
    # the defaults potrf_mega_tall picks for n = 4096 .. 16384, every option alone, all together, and tails longer than the matrix
    opt_cases = [((32, 33, 3, 6, 1, 4, 5, 0), (0, 16, 20, 0, 0)), ((64, 65, 3, 7, 1, 6, 3, 0), (0, 16, 20, 0, 0)), ((96, 97, 3, 7, 1, 8, 3, 0), (0, 16, 20, 0, 0)),
                 ((128, 129, 3, 6, 1, 8, 2, 0), (0, 16, 20, 0, 0)), ((64, 65, 3, 7, 1, 6, 3, 0), (8, 0, 0, 0, 0)), ((64, 65, 3, 7, 1, 6, 3, 0), (0, 0, 20, 3, 0)),
                 ((64, 65, 3, 7, 1, 6, 3, 0), (0, 0, 0, 0, 1)), ((64, 65, 3, 7, 1, 6, 3, 2), (6, 24, 28, 2, 1)), ((10, 12, 2, 4, 1, 3, 2, 0), (40, 40, 40, 1, 1)),
                 ((5, 5, 1, 1, 1, 1, 0, 0), (2, 2, 5, 0, 1))]
    for c, o in opt_cases:
        opt = (ctypes.c_int32 * 5)(*o)
        assert lib.mrbf_debug_mega_tables2(*c, opt, out) == 0, (c, o)
        assert out[5] == 0, (c, o, list(out))
        nt, mt = c[0], c[1]
        tiles = sum(mt - cc for cc in range(nt))
        assert out[2] + out[0] // 2 == tiles, (c, o, list(out))  # every tile finished by one chain job or two panel halves
    # streamed tiles of five-row block columns as two 64-row jobs (bit 1 of the last option): one extra chain job per such tile
    half_cases = [((32, 33, 3, 6, 1, 4, 5, 0), (0, 16, 20, 0, 2), range(32)), ((64, 65, 3, 7, 1, 6, 3, 0), (8, 16, 20, 0, 3), list(range(8)) + list(range(48, 64))),
                  ((64, 65, 3, 7, 1, 6, 3, 0), (0, 0, 0, 0, 2), []), ((5, 5, 1, 1, 1, 1, 5, 0), (0, 0, 0, 0, 2), range(5)), ((16, 20, 3, 6, 1, 4, 5, 1), (0, 0, 8, 0, 3), range(16))]
    for c, o, cols in half_cases:
        opt = (ctypes.c_int32 * 5)(*o)
        assert lib.mrbf_debug_mega_tables2(*c, opt, out) == 0, (c, o)
        assert out[5] == 0, (c, o, list(out))
        nt, mt = c[0], c[1]
        tiles = sum(mt - cc for cc in range(nt))
        extra = sum(min(5, mt - 1 - cc) for cc in cols)
        assert out[2] + out[0] // 2 == tiles + extra, (c, o, list(out))
    # panel tiles more than t block rows below the streamed ones as ONE 128-row job (bits 18.. of the last option = t + 1): one panel job
    # less per such tile, every tile still finished exactly once
    for c, o, t in [((32, 33, 3, 6, 1, 4, 5, 0), (0, 16, 20, 0, 1 << 18), 0), ((64, 65, 3, 7, 1, 6, 3, 0), (0, 16, 20, 0, (4 << 18) | 2), 3),
                    ((20, 24, 3, 9, 2, 5, 2, 1), (0, 0, 0, 0, 9 << 18), 8)]:
        opt = (ctypes.c_int32 * 5)(*o)
        assert lib.mrbf_debug_mega_tables2(*c, opt, out) == 0, (c, o)
        assert out[5] == 0, (c, o, list(out))
        nt, mt, srows = c[0], c[1], c[6]
        def sr(cc):
            edge = cc >= nt - o[1] and o[1] > 0 and srows < 5
            return 5 if edge else srows
        nfull = sum(1 for cc in range(nt) for i in range(cc + 1, mt) if i > cc + sr(cc) and i - cc - sr(cc) > t)
        nhalf_tiles = sum(1 for cc in range(nt) for i in range(cc + 1, mt) if i > cc + sr(cc)) - nfull
        assert out[0] == nfull + 2 * nhalf_tiles, (c, o, list(out), nfull, nhalf_tiles)
    # the block row below the square holds at most 64 non-zero rows (bit 26: the fit's right-hand sides): its panel tiles are ONE job on the
    # upper half that publishes for both, its bulk updates one half job per window -- every tile still finished / updated exactly once
    for c, o in [((32, 33, 3, 6, 1, 4, 5, 0), (0, 16, 20, 0, 1 << 26)), ((64, 65, 3, 7, 1, 6, 3, 0), (0, 16, 20, 0, (1 << 26) | 2)),
                 ((96, 97, 3, 7, 1, 8, 3, 0), (0, 16, 20, 0, (1 << 26) | (9 << 18))), ((5, 6, 1, 1, 1, 1, 0, 0), (0, 0, 0, 0, 1 << 26))]:
        opt = (ctypes.c_int32 * 5)(*o)
        assert lib.mrbf_debug_mega_tables2(*c, opt, out) == 0, (c, o)
        assert out[5] == 0, (c, o, list(out))
        opt0 = (ctypes.c_int32 * 5)(o[0], o[1], o[2], o[3], o[4] & ~(1 << 26))
        out0 = (ctypes.c_int64 * 6)()
        assert lib.mrbf_debug_mega_tables2(*c, opt0, out0) == 0 and out0[5] == 0
        assert out[0] < out0[0] and out[1] <= out0[1], (c, o, list(out), list(out0))  # fewer panel jobs, no more bulk jobs
    opt = (ctypes.c_int32 * 5)(0, 0, 0, 0, 0)
    for c in cases[:6]:  # without options: the same tables as the plain entry point
        assert lib.mrbf_debug_mega_tables2(*c, opt, out) == 0 and out[4] == seen[c], c
    assert lib.mrbf_debug_mega_tables2(4, 5, 3, 6, 1, 4, 5, 0, None, out) == -9
    assert lib.mrbf_debug_mega_tables2(4, 5, 3, 6, 1, 4, 5, 0, (ctypes.c_int32 * 5)(0, -1, 0, 0, 0), out) == -9
