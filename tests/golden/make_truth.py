"""Extended-precision truth for every case of tests/golden/rbf_golden.npz  ->  tests/golden/rbf_truth.npz.

Why: two fixture cases are too ill conditioned for "GPU weights == fp64 oracle weights to 1e-10" to be decidable in
fp64 (C1 as BASELINE.json writes it -- 20 sites in a box of radius 0.2 with the package-default shape, cond 5.7e11 --
and the general-exponent multiquadric, cond 1.6e7): two backward-stable fp64 solvers legitimately differ there by
cond x eps.  Instead of arguing with a perturbation bound, this script MEASURES how far either is from the solution:
the saddle system  [Phi Pi; Pi' 0][w; lam] = [Y; 0]  (what RBF.RBFInterpolationModel solves,
/root/reference/src/models/RbfModel.jl:759-763) is assembled from the fixture's fp64 sites / values taken as exact
rationals and solved with mpmath at 60 significant digits (pivoted LU); values and Jacobians at the fixture's query
sites are evaluated from that solution in the same arithmetic.  Radial functions: the ones oracle/rbf_oracle.py
documents (this file restates them independently, in mpmath, difference-form distances).

The truth is stored as hi + lo pairs of float64 (hi = nearest double, lo = the rest), i.e. to ~1e-32 relative.
No code of the product or of the fp64 oracle is imported.  Needs mpmath (1.3.0 in the build container); the tests
only read the .npz.

Run:  python tests/golden/make_truth.py
"""
import json
import math
import os

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DIGITS = 60


def phi_mp(kid, a, b, rho):
    if kid == 4:
        return mp.exp(-(a * rho) ** 2)
    if kid == 2:
        return (-1) ** math.ceil(float(b)) * (1 + (a * rho) ** 2) ** b
    if kid == 1:
        return (1 + (a * rho) ** 2) ** (-b)
    if kid == 0:
        return (-1) ** math.ceil(float(a) / 2.0) * (rho ** a if rho != 0 else mp.mpf(0))
    if kid == 3:
        k = int(a)
        return mp.mpf(0) if rho == 0 else (-1) ** (k + 1) * rho ** (2 * k) * mp.log(rho)
    raise ValueError(kid)


def psi_mp(kid, a, b, rho):
    """phi'(rho) / rho; at rho = 0 the limit, or 0 where there is none (the AD derivative of the reference's
    test, /root/reference/test/rbf_models.jl:99-115, sees no contribution from that term)."""
    if kid == 4:
        return -2 * a * a * mp.exp(-(a * rho) ** 2)
    if kid == 2:
        return (-1) ** math.ceil(float(b)) * 2 * a * a * b * (1 + (a * rho) ** 2) ** (b - 1)
    if kid == 1:
        return -2 * a * a * b * (1 + (a * rho) ** 2) ** (-b - 1)
    if kid == 0:
        sgn = (-1) ** math.ceil(float(a) / 2.0)
        if rho == 0:
            return sgn * a * (mp.mpf(1) if a == 2 else mp.mpf(0))
        return sgn * a * rho ** (a - 2)
    if kid == 3:
        k = int(a)
        return mp.mpf(0) if rho == 0 else (-1) ** (k + 1) * rho ** (2 * k - 2) * (2 * k * mp.log(rho) + 1)
    raise ValueError(kid)


def dist(x, c):
    return mp.sqrt(mp.fsum((xi - ci) ** 2 for xi, ci in zip(x, c)))


def solve_case(C, Y, X, kid, a, b, deg):
    n, d = C.shape
    k = Y.shape[1]
    q = 0 if deg < 0 else (1 if deg == 0 else d + 1)
    a, b = mp.mpf(a), mp.mpf(b)
    Cm = [[mp.mpf(float(v)) for v in row] for row in C]
    N = n + q
    S = mp.zeros(N, N)
    for i in range(n):
        for j in range(i + 1):
            v = phi_mp(kid, a, b, dist(Cm[i], Cm[j]) if i != j else mp.mpf(0))
            S[i, j] = v
            S[j, i] = v
        for t in range(q):
            p = mp.mpf(1) if t == 0 else Cm[i][t - 1]
            S[i, n + t] = p
            S[n + t, i] = p
    rhs = mp.zeros(N, k)
    for i in range(n):
        for l in range(k):
            rhs[i, l] = mp.mpf(float(Y[i, l]))
    sol = mp.zeros(N, k)
    for l in range(k):  # mpmath's lu_solve takes one right-hand side at a time
        col = mp.lu_solve(S, rhs[:, l])
        for i in range(N):
            sol[i, l] = col[i]
    # residual of the 60-digit solution, for the manifest
    res = S * sol - rhs
    resn = max(abs(res[i, l]) for i in range(N) for l in range(k))
    W = [[sol[i, l] for l in range(k)] for i in range(n)]
    Lam = [[sol[n + t, l] for l in range(k)] for t in range(q)]
    m = X.shape[0]
    V = [[mp.mpf(0)] * k for _ in range(m)]
    J = [[[mp.mpf(0)] * d for _ in range(k)] for _ in range(m)]
    for p in range(m):
        x = [mp.mpf(float(v)) for v in X[p]]
        for l in range(k):
            acc = mp.mpf(0)
            g = [mp.mpf(0)] * d
            for i in range(n):
                rho = dist(x, Cm[i])
                acc += W[i][l] * phi_mp(kid, a, b, rho)
                s = W[i][l] * psi_mp(kid, a, b, rho)
                if s != 0:
                    for t in range(d):
                        g[t] += s * (x[t] - Cm[i][t])
            if q >= 1:
                acc += Lam[0][l]
            if q > 1:
                for t in range(d):
                    acc += Lam[1 + t][l] * x[t]
                    g[t] += Lam[1 + t][l]
            V[p][l] = acc
            J[p][l] = g
    return W, Lam, V, J, resn


def hi_lo(arr, shape):
    hi = np.empty(shape)
    lo = np.empty(shape)
    flat_hi, flat_lo = hi.reshape(-1), lo.reshape(-1)
    for idx, v in enumerate(arr):
        h = float(v)
        flat_hi[idx] = h
        flat_lo[idx] = float(v - mp.mpf(h))
    return hi, lo


def flatten(x):
    if isinstance(x, list):
        out = []
        for e in x:
            out.extend(flatten(e))
        return out
    return [x]


def main():
    mp.mp.dps = DIGITS
    man = json.load(open(os.path.join(HERE, "rbf_golden.json")))
    z = np.load(os.path.join(HERE, "rbf_golden.npz"))
    arrays, notes = {}, []
    for c in man:
        pre = "c%03d_" % c["idx"]
        C, Y, X = z[pre + "C"], z[pre + "Y"], z[pre + "X"]
        Y = Y.reshape(C.shape[0], -1)
        n, d = C.shape
        k = Y.shape[1]
        q = 0 if c["deg"] < 0 else (1 if c["deg"] == 0 else d + 1)
        if n < q:
            continue  # under-determined (minimum-norm) cases have no unique saddle solution
        W, Lam, V, J, resn = solve_case(C, Y, X, c["kid"], c["a"], c["b"], c["deg"])
        for key, val, shape in (("W", W, (n, k)), ("Lam", Lam, (q, k)), ("V", V, (X.shape[0], k)), ("J", J, (X.shape[0], k, d))):
            hi, lo = hi_lo(flatten(val), shape)
            arrays[pre + key + "_hi"] = hi
            arrays[pre + key + "_lo"] = lo
        # distance of the committed fp64 oracle solution from the truth (for the record; asserted in tests/test_oracle.py)
        Wo = z[pre + "W"]
        eo = max(abs(mp.mpf(float(Wo[i, l])) - W[i][l]) for i in range(n) for l in range(k)) / max(abs(W[i][l]) for i in range(n) for l in range(k))
        notes.append(dict(idx=c["idx"], name=c["name"], cond=c["cond"], digits=DIGITS, mp_residual=float(resn), oracle_weight_err=float(eo)))
        print("%-48s cond %.1e  oracle weights off by %.2e  (mp residual %.1e)" % (c["name"], c["cond"], float(eo), float(resn)), flush=True)
    np.savez_compressed(os.path.join(HERE, "rbf_truth.npz"), **arrays)
    with open(os.path.join(HERE, "rbf_truth.json"), "w") as f:
        json.dump(notes, f, indent=1)
    print("wrote %d cases" % len(notes))


if __name__ == "__main__":
    main()
