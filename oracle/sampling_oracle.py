"""CPU restatement (TEST INFRASTRUCTURE ONLY) of Morbit's training-site selection helpers.

  AffinelyIndependentPointFilter / _orthogonal_complement_matrix   src/models/AffinelyIndependentPoints.jl:4-106
  nullify_last_row                                                  src/utilities.jl:437-448
  _rbf_round4                                                       src/models/RbfModel.jl:352-499

Restated as written, including two quirks of the reference that decide which sites are accepted:
  * `chol_pivot = theta_pivot_cholesky^2` is compared as `tau^2 > chol_pivot^2` (RbfModel.jl:370, :452);
  * `Z = Q[:, N+1:end]` with Q of order N (RbfModel.jl:391) is EMPTY for every start set, so the test only sees the
    null-space directions added during this round.
Kernel values come from oracle/rbf_oracle.py.  Not pinned by any fixture of the reference (test/rbf_models.jl:73-86 only
checks that round 4 runs).  `rbf_round4` below shares NO code or recurrence with the product's incremental algorithm: it
re-derives the acceptance quantity of every candidate from scratch with LAPACK (null-space basis by SVD, two Cholesky
determinants).
Reading of the reference this restatement rests on (unverifiable here): `RBF.get_matrices` returns the
polynomial matrix as N x dim(Pi) (rows = sites), so that `qr(Matrix(Pi'))` at :381 is the QR of an N-row matrix and
`[R; pi_xi']` at :426 is well formed for every N; if it were dim(Pi) x N, as the comment at :377 says, the code would only
run for N = dim(Pi).
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


def _logdet_spd(M):
    """log det of a symmetric matrix by LAPACK Cholesky; None when it is not numerically positive definite"""
    if M.shape[0] == 0:
        return 0.0
    try:
        L = np.linalg.cholesky(0.5 * (M + M.T))
    except np.linalg.LinAlgError:
        return None
    return 2.0 * float(np.sum(np.log(np.diag(L))))


def _round4_basis(Pi0, P_acc):
    """Orthonormal basis of the directions Wild's test of the reference sees after the sites with polynomial rows P_acc have
    been accepted, derived FROM SCRATCH (scipy.linalg.null_space = LAPACK SVD): the vectors orthogonal to range(Pi) -- they
    annihilate the tail, columns of Z in RbfModel.jl:391 -- AND to the null-space directions of the start set, which the
    reference's `Z = Q[:, N+1:end]` leaves out (empty initial Z).  Only the spanned subspace matters for the determinants."""
    import scipy.linalg

    N0, q = Pi0.shape
    j = P_acc.shape[0]
    Q2 = scipy.linalg.null_space(Pi0.T) if q > 0 else np.eye(N0)       # N0 x (N0 - rank Pi0)
    A = np.block([[Pi0, Q2], [P_acc, np.zeros((j, Q2.shape[1]))]])      # (N0 + j) x (q + N0 - rank)
    return scipy.linalg.null_space(A.T)                                 # (N0 + j) x j when Pi0 has full column rank


def rbf_round4(centers, candidates, kid, a, b, deg, theta_pivot_cholesky=1e-7, max_points=None, kernel_block=None,
               extra_sites=(), max_tries=None):
    """INDEPENDENT restatement of _rbf_round4 (RbfModel.jl:352-499): positions (into `candidates`, then len(candidates) + position
    into `extra_sites`) of the accepted sites, in acceptance order.

    Nothing is updated incrementally here -- no Givens rotations, no bordered L / L^-1, no running Phi.  For every candidate
    the quantities of the reference's acceptance test are recomputed from scratch with LAPACK:
        Phi_aug   kernel matrix of (start set + accepted + candidate), assembled anew
        Z_aug     null-space basis by SVD (see _round4_basis)
        tau^2     = det(Z_aug' Phi_aug Z_aug) / det(Z' Phi Z)   -- the Schur complement  sigma - ||L^-1 v||^2  of :449, which is
                  what the bordered Cholesky of :467-475 divides by; both determinants from Cholesky factorisations
    and the candidate is accepted iff tau^2 > (theta_pivot_cholesky^2)^2, the reference's doubly squared threshold (:370, :452).
    The rank guard of :433-438 (N < dim of the polynomial space: the new row must raise the rank of the polynomial matrix) is
    the norm of the least-squares residual of the new row against the present rows.
    `extra_sites`: with cfg.use_max_points the reference draws random box points once the database candidates are used up
    (:405-416), at most max_tries = 10 max_points (+1, `num_tries <= max_tries`) of them; the caller passes the drawn points.
    kernel_block(X, C) -> phi(||x - c||) lets a test inject device-computed kernel values."""
    centers = [np.asarray(c, dtype=np.float64) for c in centers]
    d = centers[0].size
    N0 = len(centers)
    if max_points is None or max_points <= 0:
        max_points = (d + 1) * (d + 2) // 2
    if max_tries is None:
        max_tries = 10 * max_points
    if kernel_block is None:
        kernel_block = lambda X, C: orc.phi(kid, a, b, orc.pairwise_dist(np.atleast_2d(X), np.atleast_2d(C)))
    accepted = []
    if not (N0 < max_points and (len(candidates) > 0 or len(extra_sites) > 0)):
        return accepted
    thr = (theta_pivot_cholesky ** 2) ** 2
    C0 = np.array(centers)
    Pi0 = orc.poly_matrix(C0, deg)
    q = Pi0.shape[1]
    dim_poly = math.comb(d + deg, d) if deg >= 0 else 0
    acc_sites = []           # accepted sites of this round
    logdet_cur = 0.0         # log det(Z' Phi Z) of the current set (empty Z at the start: det = 1)
    stream = [(pos, np.asarray(x, dtype=np.float64)) for pos, x in enumerate(candidates)]
    tries = 0
    extra = iter(enumerate(extra_sites))
    while N0 + len(acc_sites) < max_points:
        if stream:
            pos, xi = stream.pop(0)
        else:
            if tries > max_tries:
                break
            nxt = next(extra, None)
            if nxt is None:
                break
            pos, xi = len(candidates) + nxt[0], np.asarray(nxt[1], dtype=np.float64)
            tries += 1
        N = N0 + len(acc_sites)
        P_acc = orc.poly_matrix(np.array(acc_sites), deg) if acc_sites else np.zeros((0, q))
        p_xi = orc.poly_matrix(xi[None, :], deg)
        if deg >= 0 and N < dim_poly:
            P_now = np.vstack([Pi0, P_acc])
            coef = np.linalg.lstsq(P_now.T, p_xi[0], rcond=None)[0]
            if np.linalg.norm(p_xi[0] - P_now.T @ coef) <= np.finfo(float).eps * 10:
                continue
        S = np.vstack([C0] + [s[None, :] for s in acc_sites] + [xi[None, :]])
        Phi_aug = kernel_block(S, S)
        Phi_aug = 0.5 * (Phi_aug + Phi_aug.T)
        B = _round4_basis(Pi0, np.vstack([P_acc, p_xi]))
        ld = _logdet_spd(B.T @ Phi_aug @ B)
        if ld is None:
            continue         # not positive definite: tau^2 <= 0
        tau2 = math.exp(ld - logdet_cur)
        if tau2 > thr:
            accepted.append(pos)
            acc_sites.append(xi)
            logdet_cur = ld
    return accepted
