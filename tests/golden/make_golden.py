"""Generates tests/golden/rbf_golden.npz from the CPU oracle (oracle/rbf_oracle.py).

The reference holds no golden vectors for this path, and cannot be
run here (Julia absent), so these fixtures are produced by the repo's own fp64
restatement and, before being written, cross-checked on interpolant VALUES against
an independent implementation: scipy.interpolate.RBFInterpolator (same radial
functions: gaussian, multiquadric = -sqrt(1+(eps r)^2), inverse_multiquadric,
cubic = r^3, quintic = -r^5 = Cubic(beta=5), linear = -r = Cubic(beta=1),
thin_plate_spline = r^2 log r = ThinPlateSpline(k=1)).

Case grid mirrors test/rbf_models.jl:27-30 of the reference: num_vars in {2,5,10} x
Morbit.RbfKernels x polynomial_degree in -1:1, objective f1 = sum(x.^2) (:5);
plus the two-parabolas case of examples/example_two_parabolas.jl:38-48 (config C1).

Run:  python tests/golden/make_golden.py      (writes next to this file)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import rbf_oracle as orc  # noqa: E402

SCIPY_NAME = {  # (kid, a-or-None) -> scipy kernel name
    (4, None): "gaussian",
    (2, None): "multiquadric",
    (1, None): "inverse_multiquadric",
    (0, 3.0): "cubic",
    (0, 5.0): "quintic",
    (0, 1.0): "linear",
    (3, 1.0): "thin_plate_spline",
}


def scipy_values(C, Y, X, kid, a, b, deg):
    from scipy.interpolate import RBFInterpolator
    import warnings

    name = SCIPY_NAME.get((kid, None)) or SCIPY_NAME.get((kid, a))
    if name is None or (kid in (1, 2) and b != 0.5):
        return None
    eps = a if kid in (1, 2, 4) else 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        itp = RBFInterpolator(C, Y, kernel=name, epsilon=eps, degree=deg)
    return itp(X)


def make_cases():
    rng = np.random.Generator(np.random.PCG64(1234))  # test/runtests.jl:4 seeds 1234
    cases = []
    for d in (2, 5, 10):
        for kname, kid in orc.KERNEL_IDS.items():
            for deg in (-1, 0, 1):
                for shape in (float("nan"), "alt"):
                    if shape == "alt":
                        # one non-default shape per kernel (only for deg 1 to keep the file small)
                        if deg != 1:
                            continue
                        a, b = {0: (5.0, 0.0), 1: (0.7, 0.5), 2: (1.3, 0.5), 3: (1.0, 0.0), 4: (2.0, 0.0)}[kid]
                    else:
                        a, b = orc.kernel_params(kname)
                    n = (d + 1) * (d + 2) // 2 if d <= 5 else 3 * d + 4  # RbfModel.jl:356 default cap for small d
                    C = rng.random((n, d))
                    Y = (C ** 2).sum(axis=1, keepdims=True)  # f1, test/rbf_models.jl:5
                    X = np.vstack([C[:1], rng.random((6, d))])  # first query = a centre (rho = 0)
                    cases.append(dict(name="grid_d%d_%s_deg%d_%s" % (d, kname, deg, "alt" if shape == "alt" else "def"),
                                      C=C, Y=Y, X=X, kid=kid, a=a, b=b, deg=deg))
    # C1: two parabolas, d = 2, k = 2, n = 20, multiquadric, deg 1
    x0 = np.array([-np.pi, 2.71828])
    rng1 = np.random.Generator(np.random.PCG64(1234))
    C = x0[None, :] + 0.2 * (2.0 * rng1.random((20, 2)) - 1.0)
    C[0] = x0
    Y = np.stack([((C - 1.0) ** 2).sum(axis=1), ((C + 1.0) ** 2).sum(axis=1)], axis=1)
    X = np.vstack([C[:2], x0[None, :] + 0.2 * (2.0 * rng1.random((8, 2)) - 1.0)])
    a, b = orc.kernel_params("multiquadric")
    cases.append(dict(name="c1_two_parabolas", C=C, Y=Y, X=X, kid=2, a=a, b=b, deg=1))
    # general-exponent multiquadrics (pow path)
    C = rng.random((30, 3))
    Y = np.stack([np.sin(C.sum(axis=1)), (C ** 2).sum(axis=1)], axis=1)
    X = rng.random((5, 3))
    cases.append(dict(name="mq_beta1p5", C=C, Y=Y, X=X, kid=2, a=0.9, b=1.5, deg=1))
    cases.append(dict(name="imq_beta1", C=C, Y=Y, X=X, kid=1, a=1.1, b=1.0, deg=0))
    # C1 again with a shape parameter that bounds the conditioning: shape_parameter = "2/Δ" with the
    # trust-region radius Δ = 0.2 of the sample box, i.e. alpha = 10 -> cond ~ 5e4, so the 1e-10 weight tolerance of
    # BASELINE.json applies outright (with the package default alpha = 1 the 20 sites in a box of radius 0.2 give cond 5.7e11)
    c1 = next(c for c in cases if c["name"] == "c1_two_parabolas")
    cases.append(dict(name="c1_two_parabolas_shape_2_over_delta", C=c1["C"], Y=c1["Y"], X=c1["X"], kid=2, a=2.0 / 0.2, b=0.5, deg=1))
    return cases


def main():
    arrays, manifest, worst = {}, [], 0.0
    for i, c in enumerate(make_cases()):
        mod = orc.fit(c["C"], c["Y"], c["kid"], c["a"], c["b"], c["deg"])
        V = mod.values(c["X"])
        J = mod.jacs(c["X"])
        # one-point path must agree with the batched path
        for p in range(c["X"].shape[0]):
            assert np.allclose(mod.value(c["X"][p]), V[p], rtol=1e-12, atol=1e-12)
            assert np.allclose(mod.jac(c["X"][p]), J[p], rtol=1e-10, atol=1e-10)
        sv = scipy_values(c["C"], c["Y"], c["X"], c["kid"], c["a"], c["b"], c["deg"])
        checked = sv is not None
        if checked:
            err = float(np.max(np.abs(sv - V)) / max(1.0, np.max(np.abs(V))))
            worst = max(worst, err)
            assert err < 1e-7, (c["name"], err)
        Phi, Pi = orc.gram(c["C"], c["kid"], c["a"], c["b"], c["deg"])
        pre = "c%03d_" % i
        for key, val in (("C", c["C"]), ("Y", c["Y"]), ("X", c["X"]), ("W", mod.w), ("Lam", mod.lam),
                         ("V", V), ("J", J), ("Phi", Phi), ("Pi", Pi)):
            arrays[pre + key] = val
        manifest.append(dict(idx=i, name=c["name"], kid=c["kid"], a=c["a"], b=c["b"], deg=c["deg"],
                             scipy_checked=checked, rel_residual=orc.rel_residual(mod, c["Y"]),
                             cond=float(np.linalg.cond(orc.saddle_matrix(Phi, Pi)))))
    np.savez_compressed(os.path.join(HERE, "rbf_golden.npz"), **arrays)
    with open(os.path.join(HERE, "rbf_golden.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote %d cases; worst scipy value mismatch %.2e" % (len(manifest), worst))


if __name__ == "__main__":
    main()
