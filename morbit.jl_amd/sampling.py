"""Host-side mirror of Morbit's RBF training-site selection around the device kernels.

Mirrors AffinelyIndependentPointFilter / _find_suitable_points (src/models/AffinelyIndependentPoints.jl:14-106,
src/models/RbfModel.jl:205-238) -- small d x d QRs that stay on the host, SURVEY.md section 8 row a12 -- and _rbf_round4
(src/models/RbfModel.jl:352-499, row a11).  Round 4's per-candidate kernel vectors `kernels(xi)` (:421) and the second
Gram assembly `RBF.get_matrices` (:374) are taken from the device in two batched calls (mrbf_gram for the start set,
mrbf_cross_gram for every candidate against every start site and every other candidate); the Givens / Cholesky
bookkeeping of Wild's test is sequential O(N^2) host work exactly as in the reference, including its acceptance rule
tau^2 > (theta_pivot_cholesky^2)^2 and its empty initial Z (RbfModel.jl:370, :391, :452).
"""
import ctypes
import math

import numpy as np

from . import _lib
from . import rbf_model as rm


def _orthogonal_complement_matrix(Y, p=np.inf):
    Q, _ = np.linalg.qr(Y, mode="complete")
    Z = Q[:, Y.shape[1]:]
    if Z.shape[1] > 0:
        Z = Z / np.linalg.norm(Z, ord=p, axis=0)[None, :]
    return Z


class _GrowingQR:
    """Householder QR of Y = [y_1 .. y_j] kept up to date as columns are appended, with the FULL orthogonal factor Q (d x d)
    explicit: appending y_{j+1} is one reflector H on the trailing rows, Q <- Q diag(I_j, H), i.e. a rank-1 update of Q's trailing
    columns -- O(d^2) per pick.  The reference re-factors Y from scratch after every pick (`_orthogonal_complement_matrix`,
    AffinelyIndependentPoints.jl:4-11: qr(Y), O(d^3) per pick, O(d^4) per filter -- 140-280 ms per model update at d = 128 in the
    iteration rehearsal, more than round 4, the fit and the Pascoletti-Serafini step together).  Same reflectors as LAPACK's dgeqrf
    (dlarfg: beta = -sign(alpha) ||x||, v_1 = 1), so Q agrees with qr(Y)'s to rounding and the picks are the reference's."""

    def __init__(self, d, Y=None):
        self.d = d
        if Y is None or Y.shape[1] == 0:
            self.Q, self.j = np.eye(d), 0
        else:
            self.Q, _ = np.linalg.qr(Y, mode="complete")
            self.j = Y.shape[1]

    def append(self, y):
        j, Q = self.j, self.Q
        if j >= self.d:
            return
        x = Q[:, j:].T @ y                      # the new column in the current basis, trailing part
        alpha, xn = x[0], np.linalg.norm(x[1:])
        if xn != 0.0:                           # dlarfg
            beta = -math.copysign(math.hypot(alpha, xn), alpha)
            tau = (beta - alpha) / beta
            v = x / (alpha - beta)
            v[0] = 1.0
            Qt = Q[:, j:]
            Qt -= np.outer(Qt @ v, tau * v)     # Q[:, j:] H
        self.j = j + 1

    def complement(self, p=np.inf):
        Z = self.Q[:, self.j:].copy()
        if Z.shape[1] > 0:
            Z /= np.linalg.norm(Z, ord=p, axis=0)[None, :]
        return Z


class AffinelyIndependentPointFilter:
    """Greedy filter: repeatedly the candidate maximising ||Z Z'(xi - x0)||_p, accepted while it exceeds pivot_val."""

    def __init__(self, x_0, seeds, n=None, Y=None, Z=None, p=np.inf, pivot_val=1e-3, ctx=None):
        self.x_0 = np.asarray(x_0, dtype=np.float64)
        self.shifted = [np.asarray(s, dtype=np.float64) - self.x_0 for s in seeds]
        d = self.x_0.size
        self.n = d if n is None else n
        assert self.n > 0, "`x_0` must not be empty and `n` must be positive."
        self.Y = np.empty((d, 0)) if Y is None else np.array(Y, dtype=np.float64)
        self.Z = np.eye(d) if Z is None else np.array(Z, dtype=np.float64)
        self.p, self.pivot_val = p, pivot_val
        # databases with many sites in the box: the candidate scan (two tall products + a reduction per pick) runs on the device
        # (mrbf_affine_scores) when the decision table says so (mrbf_dispatch_affine, as in HipRbf.jl's iterate method); a given
        # `ctx` forces the device scan (tests)
        self.ctx = ctx

    def _scores_device(self, S):
        ctx = self.ctx or _lib.default_context()
        Z = np.asfortranarray(self.Z)
        best, val = ctypes.c_int64(-1), ctypes.c_double()
        ctx.check(ctx.lib.mrbf_affine_scores(ctx.h, S.shape[0], S.shape[1], Z.shape[1], _lib.as_ptr(S), _lib.as_ptr(Z) if Z.shape[1] else None,
                                             1 if np.isinf(self.p) else 0, None, ctypes.byref(best), ctypes.byref(val)))
        return int(best.value), float(val.value)

    def _select_device(self, Sd, qr, want):
        """up to `want` further picks from the rows of Sd (picked rows zeroed) after the directions in `qr`: (positions, final Z)"""
        ctx = self.ctx or _lib.default_context()
        d = self.x_0.size
        picks = np.empty(max(want, 1), dtype=np.int64)
        npick = ctypes.c_int32(0)
        Zbuf = np.empty((max(d - qr.j, 1), d))                          # (column-major d x dz)
        Q0 = np.asfortranarray(qr.Q)
        ctx.check(ctx.lib.mrbf_affine_select(ctx.h, Sd.shape[0], d, _lib.as_ptr(Sd), qr.j, _lib.as_ptr(Q0), want, float(self.pivot_val),
                                             1 if np.isinf(self.p) else 0, picks.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                             ctypes.byref(npick), _lib.as_ptr(Zbuf)))
        got = [int(v) for v in picks[: npick.value]]
        return got, Zbuf[: d - qr.j - len(got)].T.copy()

    def _take(self, qr, i):
        self.Y = np.hstack([self.Y, self.shifted[i][:, None]])
        qr.append(self.shifted[i])
        self.Z = qr.complement(self.p)

    def collect(self):
        out = []
        if not self.shifted:
            return out
        i = int(np.argmax([np.linalg.norm(s, ord=self.p) for s in self.shifted]))
        cand = [c for c in range(len(self.shifted)) if c != i]
        qr = _GrowingQR(self.x_0.size, self.Y)
        self._take(qr, i)
        out.append(i)
        S = np.array(self.shifted) if self.shifted else np.empty((0, self.x_0.size))
        on_device = self.ctx is not None or \
            _lib.load().mrbf_dispatch_affine(len(cand), self.x_0.size) == _lib.DISPATCH_DEVICE
        if on_device:
            # the whole pick loop in ONE device call (mrbf_affine_select): the factorisation grows by a reflector per pick on the
            # device, no host round trip between picks (until round 6: one mrbf_affine_scores call + a host QR update per pick)
            Sd = np.ascontiguousarray(S, dtype=np.float64)
            Sd[i] = 0.0   # chosen sites score 0 (the reference removes them from the candidate list)
            want = min(self.n - len(out), self.x_0.size - qr.j)
            got, Z = self._select_device(Sd, qr, want)
            if got:
                self.Y = np.hstack([self.Y, S[got].T])
                self.Z = Z
                out.extend(got)
            return out
        while len(out) < self.n and cand:
            if self.Z.shape[1]:
                P = (S[cand] @ self.Z) @ self.Z.T                     # all candidates at once: rows Z Z'(xi - x0)
                vals = np.linalg.norm(P, ord=self.p, axis=1)
            else:
                vals = np.zeros(len(cand))
            b = int(np.argmax(vals))                                   # first maximiser, like the `>` scan of the reference
            if not vals[b] > self.pivot_val:
                break
            best = cand[b]
            self._take(qr, best)
            cand.remove(best)
            out.append(best)
        return out


def results_in_box_indices(sites, lb, ub, exclude_indices=()):
    """indices of database sites inside the box (Databases.jl:324-327); `sites` is an (N, d) array or list"""
    ex = set(exclude_indices)
    lb, ub = np.asarray(lb), np.asarray(ub)
    return [i for i, s in enumerate(sites) if i not in ex and np.all(lb <= s) and np.all(s <= ub)]


def _find_suitable_points(sites, lb, ub, x, x_index, piv_val, already_inspected_indices=(), Y=None, Z=None, n_missing=None,
                          collect_improving_directions=True):
    """RbfModel.jl:205-238 -> (filtered_indices, improving_directions, candidate_indices, Y, Z)"""
    x = np.asarray(x, dtype=np.float64)
    cand = results_in_box_indices(sites, lb, ub, [x_index, *already_inspected_indices])
    flt = AffinelyIndependentPointFilter(x, [sites[i] for i in cand], n=x.size if n_missing is None else n_missing, Y=Y, Z=Z,
                                         p=np.inf, pivot_val=piv_val)
    picked = [cand[i] for i in flt.collect()]
    dirs = [flt.Z[:, j].copy() for j in range(flt.Z.shape[1])][::-1] if collect_improving_directions else None
    return picked, dirs, cand, flt.Y, flt.Z


def _nullify_last_row(R):
    """Givens rotations that zero the appended last row of an upper-triangular R (utilities.jl:437-448)."""
    R = np.array(R, dtype=np.float64)
    m, n = R.shape
    G = np.eye(m)
    for j in range(min(m - 1, n)):
        a, b = R[j, j], R[m - 1, j]
        r = math.hypot(a, b)
        if r == 0.0:
            continue
        c, s = a / r, b / r
        rj, rm_ = R[j].copy(), R[m - 1].copy()
        R[j], R[m - 1] = c * rj + s * rm_, -s * rj + c * rm_
        gj, gm = G[j].copy(), G[m - 1].copy()
        G[j], G[m - 1] = c * gj + s * gm, -s * gj + c * gm
    return R, G


def cross_gram(cfg, X, C, delta=1.0, ctx=None):
    """phi(||x_i - c_j||) for all pairs on the device (mrbf_cross_gram)."""
    ctx = ctx or _lib.default_context()
    X, C = _lib.host_f64(X), _lib.host_f64(C)
    m, d = X.shape
    n = C.shape[0]
    kid, a, b = rm._get_kernel_params(delta, cfg)
    K = np.empty((m, n))
    ctx.check(ctx.lib.mrbf_cross_gram(ctx.h, m, n, d, _lib.as_ptr(X), _lib.as_ptr(C), kid, a, b, _lib.as_ptr(K)))
    return K


def _rand_box_point(lb, ub, rng):
    """utilities.jl:303: uniform point of the box"""
    lb, ub = np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64)
    return lb + (ub - lb) * rng.random(lb.size)


class Round4State:
    """the factors mrbf_round4 left on the device (start sites, candidates, accepted positions, kappa-factor): hand it to
    `fit_from_round4` to get the model on (start sites + accepted sites) without a new factorisation"""

    def __init__(self, ctx, handle, cfg, delta, start_sites, cand_sites, accepted):
        self.ctx, self.handle, self.cfg, self.delta = ctx, handle, cfg, delta
        self.start_sites, self.cand_sites, self.accepted = start_sites, cand_sites, list(accepted)

    @property
    def training_sites(self):
        return np.vstack([self.start_sites, self.cand_sites[self.accepted]]) if self.accepted else self.start_sites.copy()

    def free(self):
        if self.handle is not None and self.ctx.h:
            self.ctx.lib.mrbf_free_round4(self.ctx.h, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def rbf_round4_device(cfg, start_sites, cand_sites, delta=1.0, ctx=None, keep_state=False, rc_only=False):
    """_rbf_round4's selection (RbfModel.jl:352-499) as ONE device call: positions (into cand_sites) of the accepted sites in
    acceptance order [, Round4State].  Raises MrbfError(-2 / ESINGULAR) when the start set does not carry the polynomial tail;
    with `rc_only` returns (rc, accepted, state) and raises nothing (what the routed `_rbf_round4` uses)."""
    ctx = ctx or _lib.default_context()
    C0, Xc = _lib.host_f64(start_sites), _lib.host_f64(cand_sites)
    n0, d = C0.shape
    mc = Xc.shape[0]
    kid, a, b = rm._get_kernel_params(delta, cfg)
    acc = np.zeros(max(mc, 1), dtype=np.int32)
    nacc = ctypes.c_int32()
    h = _lib.c_vp()
    rc = ctx.lib.mrbf_round4(ctx.h, n0, d, _lib.as_ptr(C0), mc, _lib.as_ptr(Xc) if mc else None, kid, a, b, cfg.polynomial_degree,
                             int(cfg.max_model_points), float(cfg.θ_pivot_cholesky), _lib.as_ptr(acc), ctypes.byref(nacc),
                             ctypes.byref(h) if keep_state else None)
    if rc != 0:
        if rc_only:
            return rc, [], None
        ctx.check(rc)
    accepted = [int(v) for v in acc[: nacc.value]]
    state = Round4State(ctx, h if h.value else None, cfg, delta, C0, Xc.reshape(mc, d), accepted) if keep_state else None
    if rc_only:
        return rc, accepted, state
    return (accepted, state) if keep_state else accepted


def fit_from_round4(state, training_values, fully_linear=False, rc_only=False):
    """update_model (RbfModel.jl:743-767) for the training set (start sites + sites accepted by round 4) from the factor round 4
    left on the device -- the reference's TODO at RbfModel.jl:657-660.  `training_values`: (n0 + n_accepted) x k in that order.
    With `rc_only` returns (rc, model or None) and raises nothing."""
    ctx = state.ctx
    S = state.training_sites
    n, d = S.shape
    Y = _lib.host_f64(training_values).reshape(n, -1)
    k = Y.shape[1]
    q = 0 if state.cfg.polynomial_degree < 0 else (1 if state.cfg.polynomial_degree == 0 else d + 1)
    W, L = np.empty((n, k)), np.empty((max(q, 1), k))
    h, info = _lib.c_vp(), _lib.FitInfo()
    if state.handle is None:  # nothing was selected / no state kept: the ordinary fit
        mod = rm.update_model(state.cfg, S, Y, state.delta, fully_linear, ctx=ctx)
        return (0, mod) if rc_only else mod
    rc = ctx.lib.mrbf_fit_from_round4(ctx.h, state.handle, k, _lib.as_ptr(Y), ctypes.byref(h), _lib.as_ptr(W), _lib.as_ptr(L), ctypes.byref(info))
    if rc != 0:
        if rc_only:
            return rc, None
        ctx.check(rc)
    mod = rm.RbfModel(ctx, h, n, d, k, q, fully_linear, W, L[:q], info.asdict())
    return (0, mod) if rc_only else mod


class Round4Keeper:
    """HipRbf.jl's `_ROUND4_KEPT`: the factor round 4 left behind, per (sub-)database, together with the training indices it
    describes (start indices, then accepted indices, in factor order); `update_model_from_selection` consumes it."""

    def __init__(self):
        self.kept = {}

    def put(self, db_key, state, training_indices):
        self.drop(db_key)
        self.kept[db_key] = (state, list(training_indices))

    def pop(self, db_key):
        return self.kept.pop(db_key, None)

    def drop(self, db_key):
        old = self.kept.pop(db_key, None)
        if old is not None and old[0] is not None:
            old[0].free()


def update_model_from_selection(cfg, sites, values, training_indices, delta=1.0, fully_linear=False, keeper=None, db_key=None, ctx=None,
                                stats=None):
    """`update_model(mod, meta, cfg::HipRbfConfig, ...)` of HipRbf.jl (RbfModel.jl:743-767): the model on the training set
    `_collect_indices(meta)` -- from the kept round-4 factor when the decision table (mrbf_dispatch_fit / mrbf_dispatch_after) says
    it describes exactly this training set, else Gram + factorisation + solve (`mrbf_fit`).  `sites` / `values`: the database."""
    lib = _lib.load()
    idx = list(training_indices)
    S, Y = np.asarray(sites, dtype=np.float64)[idx], np.asarray(values, dtype=np.float64)[idx]
    kept = keeper.pop(db_key) if keeper is not None else None
    if kept is not None:
        state, ids = kept
        d = S.shape[1]
        q = 0 if cfg.polynomial_degree < 0 else (1 if cfg.polynomial_degree == 0 else d + 1)
        n0, nacc = (state.start_sites.shape[0], len(state.accepted)) if state is not None and state.handle is not None else (0, 0)
        if lib.mrbf_dispatch_fit(len(idx), n0, q, nacc, int(ids == idx)) == _lib.FIT_FROM_ROUND4:
            rc, mod = fit_from_round4(state, Y, fully_linear, rc_only=True)
            if rc == 0:
                if stats is not None:
                    stats["fit"] = "from_round4"
                state.free()
                return mod
            if not lib.mrbf_dispatch_after(_lib.ENTRY_FIT_FROM_ROUND4, rc):
                state.ctx.check(rc)
        if state is not None:
            state.free()
    if stats is not None:
        stats["fit"] = "full"
    return rm.update_model(cfg, S, Y, delta, fully_linear, ctx=ctx)


def _rbf_round4(sites, lb_2, ub_2, x, delta, indices_found_so_far, cfg, ctx=None, kernel_block=None, rng=None, new_sites=None,
                keeper=None, db_key=None, stats=None):
    """Wild's second selection round (RbfModel.jl:352-499): database indices of additional training sites that keep the
    Cholesky factors of Z'Phi Z bounded.  `sites` is the database as an (N_db, d) array; candidates are the box members
    not yet chosen, in database order.  With `cfg.use_max_points` random box points are tried once the database candidates
    are used up (RbfModel.jl:405-416, at most 10 max_points + 1 of them); an accepted one is appended to the list `new_sites`
    (the caller's `new_result!`, :455-457) and gets the index len(sites) + its position there."""
    sites = np.asarray(sites, dtype=np.float64)
    d = sites.shape[1]
    max_points = (d + 1) * (d + 2) // 2 if cfg.max_model_points <= 0 else cfg.max_model_points
    N = len(indices_found_so_far)
    cand = results_in_box_indices(sites, lb_2, ub_2, indices_found_so_far)
    round4 = []
    if not (N < max_points and (cand or cfg.use_max_points)):
        return round4
    if cfg.use_max_points:
        # the random points the loop may need, drawn up front so that their kernel values ride in the same batched device call
        # (the reference draws them one by one from the global RNG; the sequence of box points is the same)
        rng = np.random.default_rng() if rng is None else rng
        max_tries = 10 * max_points
        fresh = np.array([_rand_box_point(lb_2, ub_2, rng) for _ in range(max_tries + 1)])
        new_sites = [] if new_sites is None else new_sites
    else:
        fresh = np.empty((0, d))
    chol_pivot = cfg.θ_pivot_cholesky ** 2
    deg = cfg.polynomial_degree
    C0 = sites[list(indices_found_so_far)]
    Xc = np.vstack([sites[cand], fresh]) if len(cand) else fresh
    ids = list(cand) + [-1] * fresh.shape[0]
    if keeper is not None:
        keeper.drop(db_key)          # whatever was kept belongs to an older training set
    if kernel_block is None and _lib.load().mrbf_dispatch_round4(N, d, deg, Xc.shape[0]) == _lib.DISPATCH_DEVICE:
        # the whole selection as one device call (round4.hip), routed like `_rbf_round4(..., cfg::HipRbfConfig)` in HipRbf.jl; the
        # incremental host bookkeeping below is Morbit's own method: start sets that cannot carry the polynomial tail (n0 < q, decided
        # up front) or turn out rank deficient (the device call says so: mrbf_dispatch_after)
        rc, accepted, state = rbf_round4_device(cfg, C0, Xc, delta, ctx=ctx, keep_state=keeper is not None, rc_only=True)
        if rc == 0:
            for pos in accepted:
                id_ = ids[pos]
                if id_ < 0:
                    new_sites.append(Xc[pos].copy())
                    id_ = sites.shape[0] + len(new_sites) - 1
                round4.append(id_)
            if keeper is not None and state is not None:
                keeper.put(db_key, state, list(indices_found_so_far) + round4)
            if stats is not None:
                stats["round4"] = "device"
            return round4
        if not _lib.load().mrbf_dispatch_after(_lib.ENTRY_ROUND4, rc):
            (ctx or _lib.default_context()).check(rc)
    if stats is not None:
        stats["round4"] = "reference"
    if kernel_block is None:
        # two device calls replace one kernels(xi) call per candidate plus RBF.get_matrices
        Phi, Pi, _ = rm.get_matrices(cfg, C0, delta, ctx=ctx)
        Phi = np.array(Phi)
        K_all = cross_gram(cfg, Xc, np.vstack([C0, Xc]), delta, ctx=ctx)      # candidates x (start set + candidates)
    else:
        Phi = kernel_block(C0, C0)
        Pi = np.hstack([np.ones((N, 1)), C0])[:, : (0 if deg < 0 else (1 if deg == 0 else d + 1))]
        K_all = kernel_block(Xc, np.vstack([C0, Xc]))
    q = Pi.shape[1]
    Q, Rr = (np.linalg.qr(Pi, mode="complete") if q > 0 else (np.eye(N), np.zeros((N, 0))))
    R = np.vstack([Rr[: min(N, q)], np.zeros((N - min(N, q), q))]) if q > 0 else np.zeros((N, 0))
    Z = Q[:, N:]                     # empty, as in the reference (RbfModel.jl:391)
    L = np.zeros((0, 0))
    Linv = np.zeros((0, 0))
    phi0 = float(Phi[0, 0])
    cols = list(range(N))            # columns of K_all that are current centres
    n0 = N
    for pos, id_ in enumerate(ids):
        if N >= max_points:
            break
        xi = Xc[pos]
        phixi = K_all[pos, cols]
        pixi = np.concatenate([[1.0], xi])[:q]
        Rxi, G = _nullify_last_row(np.vstack([R, pixi[None, :]]))
        if deg >= 0 and N < math.comb(d + deg, d):
            if np.linalg.norm(Rxi[-1, :]) <= np.finfo(float).eps * 10:
                continue             # the rank of R is not augmented by adding xi
        gt, gh = G[-1, :-1], G[-1, -1]      # last column of G' without / with its last entry
        Qg = Q @ gt
        v = Z.T @ (Phi @ Qg + phixi * gh)
        sigma = Qg @ Phi @ Qg + 2.0 * gh * (phixi @ Qg) + gh * gh * phi0
        tau2 = sigma - (np.linalg.norm(Linv @ v) ** 2 if v.size else 0.0)
        if tau2 > chol_pivot ** 2:
            if id_ < 0:  # a fresh random site: stored by the caller (new_result!, RbfModel.jl:455-457)
                new_sites.append(xi.copy())
                id_ = sites.shape[0] + len(new_sites) - 1
            round4.append(id_)
            tau = math.sqrt(tau2)
            Qn = np.zeros((N + 1, N + 1))
            Qn[:N, :N] = Q
            Qn[N, N] = 1.0
            Q = Qn @ G.T
            Z = np.block([[Z, Qg[:, None]], [np.zeros((1, Z.shape[1])), np.array([[gh]])]])
            row = (v @ Linv.T) if v.size else np.zeros(0)
            L = np.block([[L, np.zeros((L.shape[0], 1))], [row[None, :], np.array([[tau]])]])
            Linv = np.block([[Linv, np.zeros((Linv.shape[0], 1))], [-(row @ Linv)[None, :] / tau, np.array([[1.0 / tau]])]])
            R = Rxi
            Phi = np.block([[Phi, phixi[:, None]], [phixi[None, :], np.array([[phi0]])]])
            cols.append(n0 + pos)
            N += 1
    return round4
