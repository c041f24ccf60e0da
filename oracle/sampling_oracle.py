"""CPU restatement (TEST INFRASTRUCTURE ONLY) of Morbit's training-site selection helpers.

  AffinelyIndependentPointFilter / _orthogonal_complement_matrix   src/models/AffinelyIndependentPoints.jl:4-106
  nullify_last_row                                                  src/utilities.jl:437-448
  _rbf_round4                                                       src/models/RbfModel.jl:352-499

Restated as written, including two quirks of the reference that decide which sites are accepted:
  * `chol_pivot = theta_pivot_cholesky^2` is compared as `tau^2 > chol_pivot^2` (RbfModel.jl:370, :452);
  * `Z = Q[:, N+1:end]` with Q of order N (RbfModel.jl:391) is EMPTY for every start set, so the test only sees the
    null-space directions added during this round.
Kernel values come from oracle/rbf_oracle.py.  Not pinned by any fixture of the reference (test/rbf_models.jl:73-86 only
checks that round 4 runs); pinned here by the invariant that every accepted site passes the test when it is re-derived
from scratch.
"""
import math

import numpy as np

from oracle import rbf_oracle as orc


def orthogonal_complement_matrix(Y, p=np.inf):
    Q, _ = np.linalg.qr(Y, mode="complete")
    Z = Q[:, Y.shape[1]:]
    if Z.shape[1] > 0:
        Z = Z / np.linalg.norm(Z, ord=p, axis=0)[None, :]
    return Z


def affinely_independent_indices(x0, seeds, n, pivot_val, Y=None, Z=None, p=np.inf):
    """collect(AffinelyIndependentPointFilter(...; return_indices = true)) -> (indices, Y, Z)."""
    x0 = np.asarray(x0, dtype=np.float64)
    d = x0.size
    shifted = [np.asarray(s, dtype=np.float64) - x0 for s in seeds]
    Y = np.empty((d, 0)) if Y is None else np.array(Y, dtype=np.float64)
    Z = np.eye(d) if Z is None else np.array(Z, dtype=np.float64)
    out = []
    if not shifted:
        return out, Y, Z
    # first iterate(): the seed of largest norm, unconditionally (AffinelyIndependentPoints.jl:50-67)
    i = int(np.argmax([np.linalg.norm(s, ord=p) for s in shifted]))
    cand = [c for c in range(len(shifted)) if c != i]
    Y = np.hstack([Y, shifted[i][:, None]])
    Z = orthogonal_complement_matrix(Y, p)
    out.append(i)
    while len(out) < n and cand:
        best_val, best_index = -np.inf, -1
        for c in cand:
            val = np.linalg.norm(Z @ (Z.T @ shifted[c]), ord=p) if Z.shape[1] else 0.0
            if val > best_val:
                best_val, best_index = val, c
        if not best_val > pivot_val:
            break
        Y = np.hstack([Y, shifted[best_index][:, None]])
        Z = orthogonal_complement_matrix(Y, p)
        cand.remove(best_index)
        out.append(best_index)
    return out, Y, Z


def nullify_last_row(R):
    R = np.array(R, dtype=np.float64)
    m, n = R.shape
    G = np.eye(m)
    for j in range(min(m - 1, n)):
        a, b = R[j, j], R[m - 1, j]
        r = math.hypot(a, b)
        if r == 0.0:
            continue
        c, s = a / r, b / r
        g = np.eye(m)
        g[j, j], g[j, m - 1], g[m - 1, j], g[m - 1, m - 1] = c, s, -s, c
        R = g @ R
        G = g @ G
    return R, G


def rbf_round4(centers, candidates, kid, a, b, deg, theta_pivot_cholesky=1e-7, max_points=None, kernel_block=None):
    """Returns the positions (into `candidates`) of the accepted sites, in acceptance order.
    kernel_block(X, C) -> phi(||x - c||) lets the product mirror inject device-computed kernel values."""
    centers = [np.asarray(c, dtype=np.float64) for c in centers]
    d = centers[0].size
    N = len(centers)
    if max_points is None or max_points <= 0:
        max_points = (d + 1) * (d + 2) // 2
    if kernel_block is None:
        kernel_block = lambda X, C: orc.phi(kid, a, b, orc.pairwise_dist(np.atleast_2d(X), np.atleast_2d(C)))
    accepted = []
    if not (N < max_points and len(candidates) > 0):
        return accepted
    chol_pivot = theta_pivot_cholesky ** 2
    C0 = np.array(centers)
    Phi = kernel_block(C0, C0)
    Phi = 0.5 * (Phi + Phi.T)
    Pi = orc.poly_matrix(C0, deg)
    q = Pi.shape[1]
    Q, Rr = np.linalg.qr(Pi, mode="complete") if q > 0 else (np.eye(N), np.zeros((N, 0)))
    R = np.vstack([Rr[: min(N, q)], np.zeros((N - min(N, q), q))]) if q > 0 else np.zeros((N, 0))
    Z = Q[:, N:]  # empty: RbfModel.jl:391 as written
    L = np.zeros((0, 0))
    Linv = np.zeros((0, 0))
    phi0 = float(Phi[0, 0])
    cur = list(centers)
    for pos, xi in enumerate(candidates):
        if N >= max_points:
            break
        xi = np.asarray(xi, dtype=np.float64)
        phixi = kernel_block(xi[None, :], np.array(cur))[0]
        pixi = orc.poly_matrix(xi[None, :], deg)[0]
        Rxi, G = nullify_last_row(np.vstack([R, pixi[None, :]]))
        if N < math.comb(d + max(deg, 0), d) and deg >= 0:
            if np.linalg.norm(Rxi[-1, :]) <= np.finfo(float).eps * 10:
                continue
        Gt = G.T
        gt = Gt[:-1, -1]
        gh = G[-1, -1]
        Qg = Q @ gt
        v = Z.T @ (Phi @ Qg + phixi * gh)
        sigma = Qg @ Phi @ Qg + 2.0 * gh * (phixi @ Qg) + gh * gh * phi0
        tau2 = sigma - (np.linalg.norm(Linv @ v) ** 2 if v.size else 0.0)
        if tau2 > chol_pivot ** 2:
            accepted.append(pos)
            tau = math.sqrt(tau2)
            Qn = np.zeros((N + 1, N + 1))
            Qn[:N, :N] = Q
            Qn[N, N] = 1.0
            Q = Qn @ Gt
            Z = np.block([[Z, Qg[:, None]], [np.zeros((1, Z.shape[1])), np.array([[gh]])]])
            row = (v @ Linv.T) if v.size else np.zeros(0)
            L = np.block([[L, np.zeros((L.shape[0], 1))], [row[None, :], np.array([[tau]])]])
            Linv = np.block([[Linv, np.zeros((Linv.shape[0], 1))], [-(row @ Linv)[None, :] / tau, np.array([[1.0 / tau]])]])
            R = Rxi
            Phi = np.block([[Phi, phixi[:, None]], [phixi[None, :], np.array([[phi0]])]])
            cur.append(xi)
            N += 1
    return accepted
