"""CPU tests of the host-side mirror (no GPU): RbfConfig validation / hashing, kernel-parameter mapping,
shape-parameter strings, container layout + batched twins and the backtracking loop logic, using a test
double in place of the device model (the double evaluates through the oracle)."""
import math

import numpy as np
import pytest

import morbit.jl_amd as pkg
from morbit.jl_amd import descent, rbf_model, surrogates
from oracle import rbf_oracle as orc


def test_rbfconfig_defaults_and_asserts():
    c = pkg.RbfConfig()
    assert (c.kernel, c.polynomial_degree, c.θ_enlarge_1, c.θ_pivot, c.θ_pivot_cholesky, c.max_model_points) == \
        ("cubic", 1, 2.0, 0.25, 1e-7, -1)                      # RbfModel.jl:66-100
    assert math.isnan(c.shape_parameter) and c.max_evals == 2 ** 63 - 1
    for bad in (dict(kernel="exp"), dict(kernel="cubic", shape_parameter=2.0), dict(kernel="thin_plate_spline", shape_parameter=1.5),
                dict(kernel="gaussian", shape_parameter=-1.0), dict(θ_enlarge_1=0.5), dict(θ_pivot=0.9), dict(polynomial_degree=2)):
        with pytest.raises(AssertionError):
            pkg.RbfConfig(**bad)
    pkg.RbfConfig(kernel="cubic", shape_parameter=5.0)
    pkg.RbfConfig(kernel="gaussian", shape_parameter="1/Δ")


def test_config_hash_and_equality_make_configs_combinable():
    a, b = pkg.RbfConfig(kernel="gaussian"), pkg.RbfConfig(kernel="gaussian")
    assert a == b and hash(a) == hash(b) and len({a, b}) == 1     # NaN shape parameters compare isequal (RbfModel.jl:125-130)
    assert a != pkg.RbfConfig(kernel="gaussian", shape_parameter=2.0)
    assert pkg.combinable(a) and pkg.max_evals(pkg.RbfConfig(max_evals=7)) == 7


def test_kernel_param_mapping_matches_oracle_and_reference():
    for name in pkg.RbfKernels:
        cfg = pkg.RbfConfig(kernel=name)
        kid, a, b = rbf_model._get_kernel_params(0.3, cfg)
        assert kid == orc.KERNEL_IDS[name] and (a, b) == orc.kernel_params(name)
    assert rbf_model._get_kernel_params(0.5, pkg.RbfConfig(kernel="multiquadric", shape_parameter="1/Δ")) == (2, 2.0, 0.5)
    assert rbf_model._get_kernel_params(1.0, pkg.RbfConfig(kernel="cubic", shape_parameter=5.0)) == (0, 5.0, 0.0)
    assert pkg.parse_shape_param_string(0.25, "2*Δ^2 + 1") == 1.125
    assert pkg.parse_shape_param_string(4.0, "sqrt(Δ)/2") == 1.0
    with pytest.raises(ValueError):
        pkg.parse_shape_param_string(1.0, "__import__('os').system('true')")


class OracleBackedModel:
    """test double with the RbfModel surface the container / descent mirrors use"""

    def __init__(self, ref):
        self.ref, self.k, self.d, self.fully_linear, self.ctx = ref, ref.num_outputs, ref.C.shape[1], True, None
        self.sweeps = 0

    @property
    def num_outputs(self):
        return self.k

    def eval_sites(self, X, want_values=True, want_jac=False, **kw):
        self.sweeps += 1
        X = np.atleast_2d(X)
        return (self.ref.values(X) if want_values else None), (self.ref.jacs(X) if want_jac else None)


@pytest.fixture()
def two_models():
    rng = np.random.default_rng(0)
    C = rng.random((30, 3))
    Y = np.stack([(C ** 2).sum(1), np.sin(C.sum(1)), C[:, 0]], axis=1)
    m3 = OracleBackedModel(orc.fit(C, Y, 0, 3.0, 0.0, 1))
    m1 = OracleBackedModel(orc.fit(C, Y[:, :1], 4, 1.0, 0.0, 1))
    return m3, m1


def test_container_dispatch_layout_and_single_sweep(two_models):
    m3, m1 = two_models
    sc = surrogates.SurrogateContainer(objectives=[surrogates.RefSurrogate(m3, [2]), surrogates.RefSurrogate(m1, [0]),
                                                   surrogates.RefSurrogate(m3, [0, 1])])
    x = np.array([0.2, 0.4, 0.6])
    v = surrogates.eval_container_objectives_at_scaled_site(sc, None, x)
    r3, r1 = m3.ref.value(x), m1.ref.value(x)
    assert np.allclose(v, [r3[2], r1[0], r3[0], r3[1]])            # vcat over objective indices, SurrogateContainer.jl:263-267
    assert m3.sweeps == 1                                          # the reference does one sweep per objective index
    J = surrogates.eval_container_objectives_jacobian_at_scaled_site(sc, None, x)
    assert J.shape == (4, 3) and np.allclose(J[2:], m3.ref.jac(x)[:2])
    X = np.random.default_rng(1).random((5, 3))
    V = surrogates.eval_container_objectives_at_scaled_sites(sc, None, X)
    assert all(np.allclose(V[p], surrogates.eval_container_objectives_at_scaled_site(sc, None, X[p])) for p in range(5))
    # empty lists: MIN_PRECISION[] and a 0 x d matrix (SurrogateContainer.jl:259, :265)
    assert surrogates.eval_container_nl_eq_constraints_at_scaled_site(sc, None, x).shape == (0,)
    assert surrogates.eval_container_nl_ineq_constraints_jacobian_at_scaled_site(sc, None, x).shape == (0, 3)
    assert sc.fully_linear()
    # optim handles: one closure per scalar output, gradient filled in place (AbstractSurrogateInterface.jl:98-106)
    hs = surrogates.get_objectives_optim_handles(sc, None)
    assert len(hs) == 4
    g = np.zeros(3)
    assert np.isclose(hs[3](x, g), r3[1]) and np.allclose(g, m3.ref.jac(x)[1])
    assert np.isclose(hs[0](x, np.zeros(0)), r3[2])


def test_backtrack_general_route_matches_reference_loop(two_models):
    m3, m1 = two_models
    sc = surrogates.SurrogateContainer(objectives=[surrogates.RefSurrogate(m3, [0]), surrogates.RefSurrogate(m1, [0])])
    f = lambda z: np.array([m3.ref.value(z)[0], m1.ref.value(z)[0]])
    cfg = descent.SteepestDescentConfig()
    assert cfg.max_loops == 117
    x = np.array([0.5, 0.5, 0.5])
    for strict in (True, False):
        cfg.strict_backtracking = strict
        for dirn, s0 in ((np.array([-1.0, -1.0, -1.0]) / math.sqrt(3), 2.0), (np.array([1.0, 0.0, 0.0]), 1.0)):
            xp, mxp, step, i = descent._backtrack(x, dirn, s0, 0.4, sc, cfg)
            rxp, rmxp, rstep, ri = orc.backtrack(f, x, dirn, s0, 0.4, strict=strict)
            if ri > 90:  # no descent along this direction: both loops run into the rounding-noise floor of mx - mx_plus
                assert i > 90
                continue
            assert i == ri and np.array_equal(xp, rxp) and np.array_equal(step, rstep) and np.allclose(mxp, rmxp)
    assert descent._armijo_condition(True, np.array([1.0, 1.0]), np.array([0.5, 1.0]), 1.0, 1.0, 1e-6) is False
    assert descent._armijo_condition(False, np.array([1.0, 2.0]), np.array([1.5, 1.0]), 1.0, 1.0, 1e-6) is True


class _Outer:
    """phi(xi) = [xi[0] * g0 + g1^2, sin(g0) + sum(t)] on xi = [t (3); g (2)] -- an exactly evaluated outer function"""
    num_outputs = 2

    def eval(self, xi):
        t, g = xi[:3], xi[3:]
        return np.array([t[0] * g[0] + g[1] ** 2, np.sin(g[0]) + t.sum()])

    def jacobian(self, xi):
        t, g = xi[:3], xi[3:]
        return np.array([[g[0], 0.0, 0.0, t[0], 2.0 * g[1]], [1.0, 1.0, 1.0, np.cos(g[0]), 0.0]])


class _Scaler:
    """affine unscaling t = lb + w * x (AbstractVarScaler surface used by CompositeSurrogate)"""
    lb, w = np.array([1.0, -2.0, 0.5]), np.array([2.0, 3.0, 0.25])

    def untransform(self, x):
        return self.lb + self.w * np.asarray(x)

    def jacobian_of_unscaling(self):
        return np.diag(self.w)


def test_composite_surrogate_chain_rule_and_batched_twins(two_models):
    # AbstractSurrogateInterface.jl:136-154, :175-229: f(x) = phi([T(x); g(x)]),  Df = D_t phi J + D_g phi Dg
    m3, m1 = two_models
    comp = surrogates.CompositeSurrogate(m3, _Outer(), [2, 0])
    scal = _Scaler()
    x = np.array([0.3, 0.5, 0.7])
    g = m3.ref.value(x)[[2, 0]]
    xi = np.concatenate([scal.untransform(x), g])
    assert np.allclose(surrogates.eval_models(comp, scal, x), _Outer().eval(xi))
    J = surrogates.get_jacobian(comp, scal, x)
    h = 1e-6
    fd = np.stack([(surrogates.eval_models(comp, scal, x + h * e) - surrogates.eval_models(comp, scal, x - h * e)) / (2 * h) for e in np.eye(3)], axis=1)
    assert np.allclose(J, fd, atol=1e-6)
    assert np.array_equal(surrogates.get_gradient(comp, scal, x, 1), J[1]) and np.array_equal(surrogates.get_jacobian(comp, scal, x, [1]), J[1:])
    # scal = None: identity unscaling
    assert np.allclose(surrogates.eval_models(comp, None, x), _Outer().eval(np.concatenate([x, g])))
    # in a container next to a RefSurrogate of the same grouped model: ONE sweep for values, one for Jacobians
    sc = surrogates.SurrogateContainer(objectives=[surrogates.RefSurrogate(m3, [1]), comp], nl_ineq_constraints=[comp])
    X = np.random.default_rng(2).random((4, 3))
    before = m3.sweeps
    V = surrogates.eval_container_objectives_at_scaled_sites(sc, scal, X)
    assert m3.sweeps == before + 1 and V.shape == (4, 3)
    for p in range(4):
        assert np.allclose(V[p, 1:], surrogates.eval_models(comp, scal, X[p])) and np.isclose(V[p, 0], m3.ref.value(X[p])[1])
    JJ = surrogates.eval_container_objectives_jacobian_at_scaled_sites(sc, scal, X)
    for p in range(4):
        assert np.allclose(JJ[p, 1:], surrogates.get_jacobian(comp, scal, X[p])) and np.allclose(JJ[p, 0], m3.ref.jac(X[p])[1])
    assert np.allclose(surrogates.eval_container_nl_ineq_constraints_at_scaled_site(sc, scal, x), surrogates.eval_models(comp, scal, x))
    assert sc.fully_linear()
    # the backtracking twin takes the general batched route for composites
    assert descent._single_model(sc) is None
