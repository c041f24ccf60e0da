"""Host-side mirror of the surrogate dispatch layer the descent code calls.

Mirrors /root/reference/src/AbstractSurrogateInterface.jl (RefSurrogate :122-134, :159-229;
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


def eval_models(sur, scal, x_hat, ell=None):
    if isinstance(sur, RefSurrogate):  # AbstractSurrogateInterface.jl:159-164
        idx = sur.output_indices if ell is None else np.asarray(sur.output_indices)[ell]
        return rm.eval_models(sur.model, scal, x_hat, idx)
    return rm.eval_models(sur, scal, x_hat, ell)


def get_gradient(sur, scal, x_hat, ell):
    if isinstance(sur, RefSurrogate):
        return rm.get_gradient(sur.model, scal, x_hat, sur.output_indices[ell])
    return rm.get_gradient(sur, scal, x_hat, ell)


def get_jacobian(sur, scal, x_hat, rows=None):
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
        return all(rm.fully_linear(s.model if isinstance(s, RefSurrogate) else s) for kd in kinds for s in self.lists[kd])

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

    # ---- batched twins: one device sweep per distinct inner model
    def _grouped(self, kind):
        groups = {}
        for pos, s in enumerate(self.lists[kind]):
            inner = s.model if isinstance(s, RefSurrogate) else s
            idx = s.output_indices if isinstance(s, RefSurrogate) else list(range(inner.num_outputs))
            groups.setdefault(id(inner), (inner, []))[1].append((pos, idx))
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
            for pos, idx in members:
                out[:, offs[pos]:offs[pos + 1]] = V[:, idx]
        return out

    def _jac_at_sites(self, kind, scal, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        offs = self._layout(kind)
        out = np.empty((X.shape[0], offs[-1], X.shape[1]))
        for inner, members in self._grouped(kind):
            J = rm.get_jacobians_at_sites(inner, scal, X)
            for pos, idx in members:
                out[:, offs[pos]:offs[pos + 1], :] = J[:, idx, :]
        return out

    def _optim_handles(self, kind, scal):
        return [_get_optim_handle(s, scal, l) for s in self.lists[kind] for l in range(s.num_outputs)]


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
