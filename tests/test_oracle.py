"""CPU tests of the oracle itself: golden vectors, the properties the reference's
test/rbf_models.jl asserts (:104 interpolation, :105-109 grad == Jacobian row,
:110-115 grad ~ AD), and the C restatement against the NumPy one."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import rbf_oracle as orc


def test_golden_reproduced(golden):
    for c in golden:
        mod = orc.fit(c["C"], c["Y"], c["kid"], c["a"], c["b"], c["deg"])
        scale = max(1.0, np.abs(c["V"]).max())
        assert np.abs(mod.values(c["X"]) - c["V"]).max() <= 1e-9 * scale * max(1.0, c["cond"] * 1e-8), c["name"]
        assert np.allclose(mod.jacs(c["X"]), c["J"], rtol=1e-6, atol=1e-6 * scale), c["name"]


def test_oracle_against_extended_precision_truth(golden):
    """tests/golden/rbf_truth.npz: the same saddle systems solved by mpmath at 60 digits from an independent restatement
    of the radial functions.  MEASURES the fp64 oracle's forward error (LAPACK LU, what Julia's `\\` does): within
    cond x 8 eps on the weights everywhere, 1e-10 outright below cond 1e6, values / Jacobians 1e-8 on every case --
    incl. C1 as BASELINE.json writes it (cond 5.7e11: weights 8e-7 from the truth, values 1e-11)."""
    from tests.conftest import dist_from_truth

    eps = np.finfo(np.float64).eps
    seen = 0
    for c in golden:
        if "truth" not in c:
            continue
        seen += 1
        t = c["truth"]
        ew = dist_from_truth(c["W"], t["W"])
        assert ew <= 8 * eps * c["cond"], (c["name"], ew, c["cond"])
        if c["cond"] < 1e6:
            assert ew < 1e-10, (c["name"], ew)
        scale_v = max(1.0, np.abs(t["V"][0]).max()) / max(np.abs(t["V"][0]).max(), 1e-300)
        scale_j = max(1.0, np.abs(t["J"][0]).max()) / max(np.abs(t["J"][0]).max(), 1e-300)
        assert dist_from_truth(c["V"], t["V"]) / scale_v < 1e-8, c["name"]
        assert dist_from_truth(c["J"], t["J"]) / scale_j < 1e-8, c["name"]
    assert seen == len(golden)


def test_interpolation_at_all_sites(golden):
    # test/rbf_models.jl:104 checks the centre only; we check every training site
    for c in golden:
        mod = orc.OracleModel(c["C"], c["W"], c["Lam"], c["kid"], c["a"], c["b"], c["deg"])
        res = np.abs(mod.values(c["C"]) - c["Y"]).max() / max(1.0, np.abs(c["Y"]).max())
        assert res < 1e-13 * max(1e3, c["cond"]), (c["name"], res)
        if c["Lam"].shape[0]:
            # Pi' w = 0 (second block row of the saddle system)
            assert np.abs(c["Pi"].T @ c["W"]).max() < 1e-12 * max(1e3, c["cond"]) * max(1.0, np.abs(c["W"]).max())


def test_grad_is_jacobian_row_and_matches_complex_step_free_fd(golden):
    for c in golden:
        mod = orc.OracleModel(c["C"], c["W"], c["Lam"], c["kid"], c["a"], c["b"], c["deg"])
        for p in (0, 1, 2):  # p = 0 is a centre: rho = 0 term
            x = c["X"][p]
            J = mod.jac(x)
            for l in range(mod.num_outputs):
                assert np.array_equal(mod.grad(x, l), J[l])  # rbf_models.jl:105-109 uses ==
            if p == 0 and c["kid"] == 0 and c["a"] < 2:
                continue  # -r is not differentiable at a centre
            h = 1e-6
            fd = np.empty_like(J)
            for t in range(x.size):
                e = np.zeros_like(x)
                e[t] = h
                fd[:, t] = (mod.value(x + e) - mod.value(x - e)) / (2 * h)
            tol = 2e-5 * max(1.0, np.abs(J).max(), np.abs(c["W"]).max() * 1e-3)
            assert np.abs(fd - J).max() < tol, (c["name"], p, np.abs(fd - J).max())


def test_sign_convention_makes_projected_gram_positive_definite():
    # RbfModel.jl:391-395 takes cholesky(Z' Phi Z) -> kernels must be conditionally POSITIVE definite
    rng = np.random.default_rng(0)
    C = rng.random((40, 3))
    for kname, kid in orc.KERNEL_IDS.items():
        a, b = orc.kernel_params(kname)
        deg = 1
        if orc.cpd_order(kid, a, b) - 1 > deg:
            continue  # thin plate spline k=2 needs degree 2, which Morbit does not offer
        Phi, Pi = orc.gram(C, kid, a, b, deg)
        Q, _ = np.linalg.qr(Pi, mode="complete")
        Z = Q[:, Pi.shape[1]:]
        ev = np.linalg.eigvalsh(Z.T @ Phi @ Z)
        assert ev.min() > 0, (kname, ev.min())


def test_kernel_params_follow_morbit_mapping():
    assert orc.kernel_params("gaussian", 2.5) == (2.5, 0.0)
    assert orc.kernel_params("multiquadric", 2.0) == (2.0, 0.5)      # (sp, 1//2) RbfModel.jl:682
    assert orc.kernel_params("inv_multiquadric") == (1.0, 0.5)
    assert orc.kernel_params("cubic", 5.0) == (5.0, 0.0)             # Int(sp) RbfModel.jl:684
    assert orc.kernel_params("thin_plate_spline") == (2.0, 0.0)
    with pytest.raises(ValueError):
        orc.kernel_params("exp")  # the docstring's :exp is not an accepted symbol (RbfModel.jl:53,67)


def test_c_restatement_matches_numpy(golden):
    for c in golden[::3]:
        Phi, Pi = c_oracle.gram(c["C"], c["kid"], c["a"], c["b"], c["deg"])
        assert np.allclose(Phi, c["Phi"], rtol=1e-13, atol=1e-14), c["name"]
        assert np.array_equal(Pi, c["Pi"])
        W, Lam, info = c_oracle.fit(c["C"], c["Y"], c["kid"], c["a"], c["b"], c["deg"])
        assert info == 0
        V, J = c_oracle.eval_loop(c["C"], W, Lam, c["kid"], c["a"], c["b"], c["deg"], c["X"])
        scale = max(1.0, np.abs(c["V"]).max())
        assert np.abs(V - c["V"]).max() < 1e-13 * scale * max(1e3, c["cond"]), c["name"]
        assert np.allclose(J, c["J"], rtol=1e-6, atol=1e-7 * scale * max(1.0, c["cond"] * 1e-6)), c["name"]


def test_backtrack_restatement_first_armijo_index():
    # descent.jl:150-185: stops at the first step size satisfying the Armijo condition
    f = lambda x: np.array([np.sum((x - 1.0) ** 2), np.sum((x + 1.0) ** 2)])
    x = np.array([0.3, -0.2])
    d = np.array([-1.0, 1.0]) / np.sqrt(2)
    xp, mxp, step, i = orc.backtrack(f, x, d, 4.0, omega=0.5)
    assert i > 0 and np.all(f(x) - mxp >= np.linalg.norm(step) * 1e-6 * 0.5 - 1e-15)
