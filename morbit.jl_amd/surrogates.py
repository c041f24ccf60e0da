"""Host-side mirror of the surrogate dispatch layer the descent code calls.

Mirrors /root/reference/src/AbstractSurrogateInterface.jl (RefSurrogate :122-134, CompositeSurrogate :136-154, :159-229;
_get_optim_handle :98-106) and /root/reference/src/SurrogateContainer.jl (eval / Jacobian dispatch
:220-269) for RBF models, and adds the batched `..._at_scaled_sites` twins.
Unlike the reference, the k objectives that share one grouped RbfModel cost ONE device sweep, not k.
"""
import numpy as np

from . import rbf_model as rm


class RefSurrogate:
    """RefSurrogate(model_ref, output_indices): selects outputs of a grouped inner model."""

    def __init__(self, model, output_indices, nl_index=None):
        self.model = model
        self.output_indices = list(output_indices)
        self.nl_index = nl_index

    @property
    def num_outputs(self):
        return len(self.output_indices)


class IdentityScaler:
    """stand-in for an AbstractVarScaler that leaves the variables alone (untransform = identity, unit Jacobian)"""

    def untransform(self, x_scaled):
        return np.asarray(x_scaled, dtype=np.float64)

    def jacobian_of_unscaling(self, n_vars=None):
        return None  # identity


class CompositeSurrogate:
    """CompositeSurrogate(model_ref, outer_ref, inner_output_indices) -- AbstractSurrogateInterface.jl:136-154: models
    f(x) = phi([T(x); g(x)]) with g the selected outputs of the (grouped) inner surrogate, T the affine unscaling of x and phi
    an outer vector function that is evaluated exactly.  `outer` mirrors the two calls the reference makes on its
    AbstractVecFun: `outer.eval(xi)` (eval_vfun, :189-191) and `outer.jacobian(xi)` (_get_jacobian, :226) with xi = [t; g]."""

    def __init__(self, model, outer, inner_output_indices, num_outputs=None, nl_index=None):
        self.model = model
        self.outer = outer
        self.inner_output_indices = list(inner_output_indices)
        self.nl_index = nl_index
        self._num_outputs = num_outputs

    @property
    def num_outputs(self):
        if self._num_outputs is None:
            self._num_outputs = int(getattr(self.outer, "num_outputs"))
        return self._num_outputs

    # the two host-side pieces of the chain rule, shared by the single-site API and the batched twins
    def _xi(self, scal, x_scaled, g):
        t = (scal or IdentityScaler()).untransform(x_scaled)
        return np.concatenate([np.asarray(t, dtype=np.float64), np.asarray(g, dtype=np.float64)])  # _eval_inner, :175-188

    def _chain(self, scal, x_scaled, xi, Dg, rows=None):
        """_composite_jac (:193-207): Df = D_t phi J_unscale + D_g phi Dg"""
        Dphi = np.atleast_2d(np.asarray(self.outer.jacobian(xi), dtype=np.float64))
        if rows is not None:
            Dphi = Dphi[rows]
        n = len(x_scaled)
        J = (scal or IdentityScaler()).jacobian_of_unscaling()
        Dt = Dphi[:, :n] if J is None else Dphi[:, :n] @ np.asarray(J, dtype=np.float64)
        return Dt + Dphi[:, n:] @ Dg


def eval_models(sur, scal, x_hat, ell=None):
    if isinstance(sur, CompositeSurrogate):  # AbstractSurrogateInterface.jl:189-191
        g = rm.eval_models(sur.model, scal, x_hat, sur.inner_output_indices)
        v = np.asarray(sur.outer.eval(sur._xi(scal, x_hat, g)), dtype=np.float64)
        return v if ell is None else v[ell]
    if isinstance(sur, RefSurrogate):  # AbstractSurrogateInterface.jl:159-164
        idx = sur.output_indices if ell is None else np.asarray(sur.output_indices)[ell]
        return rm.eval_models(sur.model, scal, x_hat, idx)
    return rm.eval_models(sur, scal, x_hat, ell)


def get_gradient(sur, scal, x_hat, ell):
    if isinstance(sur, CompositeSurrogate):  # AbstractSurrogateInterface.jl:209-215
        return get_jacobian(sur, scal, x_hat, [ell])[0]
    if isinstance(sur, RefSurrogate):
        return rm.get_gradient(sur.model, scal, x_hat, sur.output_indices[ell])
    return rm.get_gradient(sur, scal, x_hat, ell)


def get_jacobian(sur, scal, x_hat, rows=None):
    if isinstance(sur, CompositeSurrogate):  # AbstractSurrogateInterface.jl:224-229: one inner sweep gives g and Dg
        V, Jm = sur.model.eval_sites(np.asarray(x_hat, dtype=np.float64)[None, :], want_values=True, want_jac=True)
        g, Dg = V[0][sur.inner_output_indices], Jm[0][sur.inner_output_indices]
        return sur._chain(scal, x_hat, sur._xi(scal, x_hat, g), Dg, rows)
    if isinstance(sur, RefSurrogate):  # AbstractSurrogateInterface.jl:217-219
        idx = sur.output_indices if rows is None else list(np.asarray(sur.output_indices)[rows])
        return rm.get_jacobian(sur.model, scal, x_hat, idx)
    return rm.get_jacobian(sur, scal, x_hat, rows)


def _get_optim_handle(sur, scal, ell):
    """NLopt-style closure (x, g) -> value, filling g in place when non-empty (AbstractSurrogateInterface.jl:98-106)."""

    def handle(x, g):
        if len(g) > 0:
            g[:] = get_gradient(sur, scal, x, ell)
        return eval_models(sur, scal, x, ell)

    return handle


_MIN_PRECISION = np.float32  # globals.jl:11


class SurrogateContainer:
    """Holds RefSurrogates for objectives / nonlinear eq / ineq constraints (SurrogateContainer.jl:101-114)."""

    def __init__(self, objectives=(), nl_eq_constraints=(), nl_ineq_constraints=()):
        self.lists = {"objective": list(objectives), "nl_eq_constraint": list(nl_eq_constraints),
                      "nl_ineq_constraint": list(nl_ineq_constraints)}

    def fully_linear(self, kind=None):
        kinds = [kind] if kind else list(self.lists)
        return all(rm.fully_linear(s.model if isinstance(s, (RefSurrogate, CompositeSurrogate)) else s) for kd in kinds for s in self.lists[kd])

    # ---- single site (reference API)
    def _eval_at_site(self, kind, scal, x_scaled):
        surs = self.lists[kind]
        if not surs:
            return np.empty(0, dtype=_MIN_PRECISION)  # SurrogateContainer.jl:265
        return self._eval_at_sites(kind, scal, np.asarray(x_scaled, dtype=np.float64)[None, :])[0]

    def _jac_at_site(self, kind, scal, x_scaled):
        surs = self.lists[kind]
        if not surs:
            return np.empty((0, len(x_scaled)), dtype=_MIN_PRECISION)  # SurrogateContainer.jl:259
        return self._jac_at_sites(kind, scal, np.asarray(x_scaled, dtype=np.float64)[None, :])[0]

    # ---- batched twins: one device sweep per distinct inner model (RefSurrogates select columns, CompositeSurrogates apply their
    #      outer function / chain rule row by row on the host: the outer function is exact and cheap, the sweep is the cost)
    def _grouped(self, kind):
        groups = {}
        for pos, s in enumerate(self.lists[kind]):
            inner = s.model if isinstance(s, (RefSurrogate, CompositeSurrogate)) else s
            groups.setdefault(id(inner), (inner, []))[1].append((pos, s))
        return groups.values()

    def _layout(self, kind):
        sizes = [s.num_outputs for s in self.lists[kind]]
        return np.concatenate([[0], np.cumsum(sizes)]).astype(int)

    def _eval_at_sites(self, kind, scal, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        offs = self._layout(kind)
        out = np.empty((X.shape[0], offs[-1]))
        for inner, members in self._grouped(kind):
            V = rm.eval_models_at_sites(inner, scal, X)
            for pos, s in members:
                if isinstance(s, CompositeSurrogate):
                    for p in range(X.shape[0]):
                        out[p, offs[pos]:offs[pos + 1]] = s.outer.eval(s._xi(scal, X[p], V[p, s.inner_output_indices]))
                else:
                    idx = s.output_indices if isinstance(s, RefSurrogate) else list(range(inner.num_outputs))
                    out[:, offs[pos]:offs[pos + 1]] = V[:, idx]
        return out

    def _jac_at_sites(self, kind, scal, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        offs = self._layout(kind)
        out = np.empty((X.shape[0], offs[-1], X.shape[1]))
        for inner, members in self._grouped(kind):
            need_vals = any(isinstance(s, CompositeSurrogate) for _, s in members)
            V, J = inner.eval_sites(X, want_values=need_vals, want_jac=True)
            J = np.ascontiguousarray(J)
            for pos, s in members:
                if isinstance(s, CompositeSurrogate):
                    ii = s.inner_output_indices
                    for p in range(X.shape[0]):
                        out[p, offs[pos]:offs[pos + 1], :] = s._chain(scal, X[p], s._xi(scal, X[p], V[p, ii]), J[p][ii])
                else:
                    idx = s.output_indices if isinstance(s, RefSurrogate) else list(range(inner.num_outputs))
                    out[:, offs[pos]:offs[pos + 1], :] = J[:, idx, :]
        return out

    def _optim_handles(self, kind, scal):
        return [_get_optim_handle(s, scal, l) for s in self.lists[kind] for l in range(s.num_outputs)]


def container_plan(sc, objectives_only=False):
    """What the device entry points need to know about a container (HipRbf.jl `_container_plan`): its distinct grouped device
    models, the role of every output row of each (objective position l >= 0, ROLE_EQ, ROLE_INEQ, ROLE_NONE), the number of objective
    rows k, of modelled constraint rows and of "foreign" surrogates (CompositeSurrogates, anything that is not a device RbfModel,
    a model row used twice) -- with a foreign one the decision table sends the call to the reference method."""
    from . import _lib
    models, roles = [], []
    k = n_con = n_foreign = 0

    def slot(m):
        for i, mm in enumerate(models):
            if mm is m:
                return i
        models.append(m)
        roles.append([_lib.ROLE_NONE] * m.num_outputs)
        return len(models) - 1

    for kind, role in (("objective", 0), ("nl_eq_constraint", _lib.ROLE_EQ), ("nl_ineq_constraint", _lib.ROLE_INEQ)):
        if objectives_only and role != 0:     # _backtrack only evaluates the objectives (descent.jl:161-179)
            continue
        for s in sc.lists[kind]:
            inner = s.model if isinstance(s, (RefSurrogate, CompositeSurrogate)) else s
            if isinstance(s, CompositeSurrogate) or not isinstance(inner, rm.RbfModel):
                n_foreign += 1
                if role == 0:
                    k += s.num_outputs
                else:
                    n_con += s.num_outputs
                continue
            i = slot(inner)
            for oi in (s.output_indices if isinstance(s, RefSurrogate) else range(inner.num_outputs)):
                if roles[i][oi] != _lib.ROLE_NONE:
                    n_foreign += 1                      # one row in two roles: not expressible
                roles[i][oi] = k if role == 0 else role
                if role == 0:
                    k += 1
                else:
                    n_con += 1
    in_order = len(models) == 1 and n_con == 0 and roles[0] == list(range(models[0].num_outputs))
    return {"models": models, "roles": [r for rr in roles for r in rr], "k": k, "n_con": n_con, "n_foreign": n_foreign,
            "in_order": in_order}


def _make(kind, plural):
    def at_site(sc, scal, x_scaled):
        return sc._eval_at_site(kind, scal, x_scaled)

    def jac_at_site(sc, scal, x_scaled):
        return sc._jac_at_site(kind, scal, x_scaled)

    def at_sites(sc, scal, X):
        return sc._eval_at_sites(kind, scal, X)

    def jac_at_sites(sc, scal, X):
        return sc._jac_at_sites(kind, scal, X)

    def handles(sc, scal):
        return sc._optim_handles(kind, scal)

    g = globals()
    g["eval_container_%s_at_scaled_site" % plural] = at_site                 # SurrogateContainer.jl:263-267
    g["eval_container_%s_jacobian_at_scaled_site" % plural] = jac_at_site    # SurrogateContainer.jl:257-261
    g["eval_container_%s_at_scaled_sites" % plural] = at_sites               # batched twin
    g["eval_container_%s_jacobian_at_scaled_sites" % plural] = jac_at_sites  # batched twin
    g["get_%s_optim_handles" % plural] = handles                             # SurrogateContainer.jl:248-255


for _kind in ("objective", "nl_eq_constraint", "nl_ineq_constraint"):  # SurrogateContainer.jl:234
    _make(_kind, _kind + "s")
