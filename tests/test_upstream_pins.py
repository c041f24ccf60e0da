"""Pins of the CPU oracle to the upstream package (RadialBasisFunctionModels 0.3.4), SURVEY.md section 8c.

`tools/pin_upstream.jl` -- run once by someone with Julia + Morbit -- writes tests/golden/upstream_pins.json: Phi / Pi of
`RBF.get_matrices` for the five kernels, values / Jacobians / numeric fields of a few `RBF.RBFInterpolationModel`s and one
`_rbf_round4` index list, all on inputs given by a closed formula that this file regenerates bit for bit.  When the file exists the
oracle is compared with it and the four guesses of DESIGN.md section 4 are settled; while it does not, the parity of this repository
is "unpinned" and the test says so.  The comparison code itself is exercised on a file the oracle writes (round trip + a corrupted
copy), so it is known to bite."""
import json
import os

import numpy as np
import pytest

from oracle import rbf_oracle as orc
from oracle import sampling_oracle as sorc
from tests.conftest import ROOT

PINS = os.path.join(ROOT, "tests", "golden", "upstream_pins.json")
PRIMES = [2.0, 3.0, 5.0, 7.0, 11.0]
KERNELS = ["cubic", "inv_multiquadric", "multiquadric", "thin_plate_spline", "gaussian"]     # Morbit.RbfKernels


def weyl(n, d):
    return np.array([[np.fmod((i + 1) * np.sqrt(PRIMES[t]), 1.0) for t in range(d)] for i in range(n)])


def inputs():
    C = weyl(12, 3)
    Y = np.stack([((C - 0.3) ** 2).sum(1), ((C + 0.5) ** 2).sum(1)], axis=1)
    probe = weyl(5, 3) * 0.9 + 0.05
    return C, Y, probe


def oracle_pins():
    """what the oracle says for every key tools/pin_upstream.jl writes (the keys it can know)"""
    C, Y, probe = inputs()
    out = {}
    for name in KERNELS:
        kid = orc.KERNEL_IDS[name]
        a, b = orc.kernel_params(name)
        Phi, Pi = orc.gram(C, kid, a, b, 1)
        out["get_matrices_%s_Phi_first_rows" % name] = Phi[:3].tolist()
        out["get_matrices_%s_Pi_rows_are_sites" % name] = Pi.tolist()
        out["kernels_of_probe_%s" % name] = orc.phi(kid, a, b, orc.pairwise_dist(probe[:1], C))[0].tolist()
        out["polys_of_probe_%s" % name] = orc.poly_matrix(probe[:1], 1)[0].tolist()
    for tag, name, deg in (("cubic_deg1", "cubic", 1), ("tps_deg1", "thin_plate_spline", 1), ("multiquadric_deg1", "multiquadric", 1),
                           ("gaussian_deg0", "gaussian", 0), ("cubic_degm1", "cubic", -1)):
        kid = orc.KERNEL_IDS[name]
        a, b = orc.kernel_params(name)
        ref = orc.fit(C, Y, kid, a, b, deg)
        out[tag + "_values"] = ref.values(probe).tolist()
        out[tag + "_jac_first_probe"] = ref.jacs(probe[:1])[0].tolist()
        out[tag + "_weights"] = ref.w.tolist()
        out[tag + "_tail"] = ref.lam.tolist()
    return out


def compare(pins, tol=1e-9):
    """-> list of (key, verdict) for every pin the oracle can be held against; raises AssertionError on a mismatch"""
    mine = oracle_pins()
    verdicts = []

    def close(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return a.shape == b.shape and np.allclose(a, b, rtol=tol, atol=tol * max(1.0, np.abs(b).max()))

    for name in KERNELS:
        key = "get_matrices_%s_Phi_first_rows" % name
        if key in pins:
            assert close(pins[key], mine[key]), "radial function %s: sign or scale differs from upstream (guess 2)" % name
            verdicts.append((key, "radial function matches"))
        key = "get_matrices_%s_Pi" % name
        if key in pins:
            Pi_up = np.asarray(pins[key], dtype=np.float64)
            Pi_me = np.asarray(mine["get_matrices_%s_Pi_rows_are_sites" % name])
            if Pi_up.shape == Pi_me.shape:
                orientation = "N x dim(Pi): rows are sites"
            else:
                assert Pi_up.shape == Pi_me.T.shape, ("unexpected shape of Pi", Pi_up.shape)
                orientation, Pi_up = "dim(Pi) x N: rows are basis functions (guess 4 of DESIGN.md is WRONG)", Pi_up.T
            assert close(Pi_up, Pi_me), "polynomial basis order differs from [1, x_1 .. x_d] (guess 1): columns %r" % (Pi_up[0].tolist(),)
            verdicts.append((key, orientation))
        for key in ("kernels_of_probe_%s" % name, "polys_of_probe_%s" % name):
            if key in pins:
                assert close(pins[key], mine[key]), key
                verdicts.append((key, "matches"))
    for tag in ("cubic_deg1", "tps_deg1", "multiquadric_deg1", "gaussian_deg0", "cubic_degm1"):
        if tag + "_error" in pins:
            verdicts.append((tag, "upstream raised: %s" % pins[tag + "_error"]))
            assert tag == "cubic_degm1", "upstream failed on a well-posed model: %s" % pins[tag + "_error"]
            continue
        for what in ("_values", "_jac_first_probe"):
            if tag + what in pins:
                assert close(pins[tag + what], mine[tag + what]), tag + what
                verdicts.append((tag + what, "matches"))
        # numeric fields of the upstream model, whatever they are called: any that has the shape of the weights / the tail must agree
        # (up to the transpose) -- this is what pins the SIGN of the conditionally positive definite kernels and the tail order
        for key, val in pins.items():
            if not key.startswith(tag + "_field_"):
                continue
            v = np.asarray(val, dtype=np.float64)
            for mine_key in (tag + "_weights", tag + "_tail"):
                m = np.asarray(mine[mine_key])
                for cand in (v, v.T if v.ndim == 2 else v):
                    if cand.shape == m.shape and m.size:
                        assert close(cand, m), "%s differs from the oracle's %s" % (key, mine_key)
                        verdicts.append((key, "equals the oracle's %s" % mine_key.split("_")[-1]))
    if "round4_accepted_positions" in pins:
        x, lb, ub = (np.asarray(pins[k]) for k in ("round4_x", "round4_lb", "round4_ub"))
        S = lb + (ub - lb) * weyl(40, 3)
        sites = np.vstack([x[None, :], S])
        start = [int(v) for v in pins["round4_start_positions"]]
        kid = orc.KERNEL_IDS["cubic"]
        a, b = orc.kernel_params("cubic")
        cand = [i for i in range(sites.shape[0]) if i not in start]
        got = sorc.rbf_round4(list(sites[start]), list(sites[cand]), kid, a, b, 1, theta_pivot_cholesky=1e-7)
        assert [cand[i] for i in got] == [int(v) for v in pins["round4_accepted_positions"]], "round 4 picks differ (guess 4)"
        verdicts.append(("round4_accepted_positions", "same sites in the same order"))
    return verdicts


def test_comparison_code_round_trips_and_bites(tmp_path):
    mine = oracle_pins()
    pins = {k: v for k, v in mine.items() if "rows_are_sites" not in k and not k.endswith(("_weights", "_tail"))}
    for name in KERNELS:
        pins["get_matrices_%s_Pi" % name] = mine["get_matrices_%s_Pi_rows_are_sites" % name]
    pins["cubic_deg1_field_rbf_weights"] = np.asarray(mine["cubic_deg1_weights"]).T.tolist()      # stored the other way round
    pins["cubic_deg1_field_poly_coeffs"] = mine["cubic_deg1_tail"]
    # one round-4 call, as the Julia script sets it up (centre + first d Weyl sites as the start set)
    x, lb, ub = np.full(3, 0.5), np.zeros(3), np.ones(3)
    sites = np.vstack([x[None, :], lb + (ub - lb) * weyl(40, 3)])
    cand = list(range(4, 41))
    kid = orc.KERNEL_IDS["cubic"]
    got = sorc.rbf_round4(list(sites[:4]), list(sites[cand]), kid, *orc.kernel_params("cubic"), 1)
    pins.update(round4_x=x.tolist(), round4_lb=lb.tolist(), round4_ub=ub.tolist(), round4_start_positions=[0, 1, 2, 3],
                round4_accepted_positions=[cand[i] for i in got])
    p = tmp_path / "pins.json"
    p.write_text(json.dumps(pins))
    verdicts = compare(json.loads(p.read_text()))
    assert len(verdicts) >= 5 * 4 + 8 and any("weights" in v for _, v in verdicts) and any("tail" in v for _, v in verdicts)
    # a flipped sign of the cubic (the convention of guess 2), a reversed tail basis (guess 1), a transposed Pi (guess 4) are all seen
    bad = dict(pins)
    bad["get_matrices_cubic_Phi_first_rows"] = (-np.asarray(pins["get_matrices_cubic_Phi_first_rows"])).tolist()
    with pytest.raises(AssertionError, match="guess 2"):
        compare(bad)
    bad = dict(pins)
    bad["get_matrices_cubic_Pi"] = np.asarray(pins["get_matrices_cubic_Pi"])[:, ::-1].tolist()
    with pytest.raises(AssertionError, match="guess 1"):
        compare(bad)
    bad = dict(pins)
    bad["round4_accepted_positions"] = pins["round4_accepted_positions"][::-1]
    with pytest.raises(AssertionError, match="guess 4"):
        compare(bad)
    tr = dict(pins)
    tr["get_matrices_cubic_Pi"] = np.asarray(pins["get_matrices_cubic_Pi"]).T.tolist()
    assert any("WRONG" in v for _, v in compare(tr))
    err = dict(pins)
    for k in [k for k in err if k.startswith("cubic_degm1")]:
        del err[k]
    err["cubic_degm1_error"] = "ArgumentError: polynomial degree too low"
    assert any("upstream raised" in v for _, v in compare(err))


def test_oracle_against_upstream_pins():
    if not os.path.exists(PINS):
        pytest.skip("parity unpinned: tests/golden/upstream_pins.json is absent -- run tools/pin_upstream.jl with Julia + Morbit + "
                    "RadialBasisFunctionModels 0.3.4 and commit its output (DESIGN.md section 4)")
    verdicts = compare(json.load(open(PINS)))
    assert len(verdicts) >= 10, verdicts
    for k, v in verdicts:
        print("%-45s %s" % (k, v))
