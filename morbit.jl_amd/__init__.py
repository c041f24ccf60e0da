"""MI355X-native RBF surrogate engine for Morbit.jl's RbfConfig hot path.

csrc/            hand-written HIP kernels (gfx950) + the C ABI of include/mrbf.h -> libmrbf.so
julia/HipRbf.jl  the ccall wrapper a Morbit maintainer drops in (cannot run here: no Julia)
rbf_model.py     host mirror of RbfConfig / RbfModel / update_model / eval_models / get_gradient / get_jacobian
surrogates.py    host mirror of RefSurrogate + SurrogateContainer eval/Jacobian dispatch (+ batched twins)
descent.py       host mirror of _backtrack / _armijo_condition with the batched step-size sweep
pascoletti_serafini.py  host mirror of the Pascoletti-Serafini descent step with an own population-batched ISRES loop
sampling.py      host mirror of AffinelyIndependentPointFilter / _find_suitable_points / _rbf_round4 (device kernel blocks)
manystart.py     many-problem mode: shard independent problems over ranks (torch.distributed, RCCL)

Importing the package does not need a GPU; any compute call does, and fails loudly without one.
"""
from . import _lib  # noqa: F401
from ._lib import Context, MrbfError, default_context, load  # noqa: F401
from .rbf_model import (RbfConfig, RbfKernels, RbfModel, combinable, eval_models, eval_models_at_sites,  # noqa: F401
                        fully_linear, get_gradient, get_jacobian, get_jacobians_at_sites, get_matrices, improve_model,
                        init_model, max_evals, model_from_coeffs, num_outputs, parse_shape_param_string,
                        set_fully_linear, update_model)
from . import descent, pascoletti_serafini, sampling, surrogates  # noqa: F401
