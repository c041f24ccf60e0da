# HipRbf.jl -- drop-in surrogate family for Morbit.jl backed by libmrbf.so (MI355X).
#
# NOT EXECUTED IN THIS REPOSITORY'S BUILD CONTAINER (no Julia there): this is the binding a Morbit maintainer adds next to
# src/models/RbfModel.jl (`include("models/HipRbf.jl")` in src/Morbit.jl after :87).  It implements the surrogate interface of
# src/AbstractSurrogateInterface.jl for a new config type `HipRbfConfig` by replacing every call into
# RadialBasisFunctionModels.jl on the hot path, plus the batched twins the descent code needs:
#     RBF.RBFInterpolationModel(...)               RbfModel.jl:759-763              -> mrbf_fit / mrbf_fit_from_round4
#     mod.model(x̂[, ℓ]), RBF.grad, RBF.jac         RbfModel.jl:784-799              -> mrbf_eval
#     RBF.get_matrices, kernels(ξ), the whole loop  RbfModel.jl:352-499 (_rbf_round4) -> mrbf_round4
#     candidate scan of the affine filter           AffinelyIndependentPoints.jl:71-106 -> mrbf_affine_scores
#     eval_container_*_at_scaled_site[s]            SurrogateContainer.jl:234-269     -> one mrbf_eval per grouped model
#     _backtrack                                    descent.jl:150-185                -> mrbf_backtrack
#     get_criticality(::PascolettiSerafiniConfig)   descent.jl:512-581                -> mrbf_ps_step
# Every ccall below has a 1:1 ctypes twin in morbit.jl_amd/_lib.py, which is what tests/ execute; struct layouts are pinned by
# tests/test_abi.py.  Conventions: a context is owned by one Julia task at a time (one per thread, created under a lock);
# models keep their context alive and are released through it; buffers passed to ccall are GC.@preserve'd.

const libmrbf = get(ENV, "MRBF_LIB", "libmrbf.so")

struct MrbfFitInfo          # mirrors mrbf_fit_info (include/mrbf.h), 72 bytes
    path::Int32; factor_info::Int32; n::Int32; q::Int32
    rel_residual::Float64; max_pitw::Float64; mu::Float64
    ms_gram::Float32; ms_project::Float32; ms_factor::Float32; ms_solve::Float32; ms_check::Float32; ms_total::Float32
    fallbacks::Int32; giveup_code::Int32
end
struct MrbfPsOptions        # mirrors mrbf_ps_options, 40 bytes
    max_ideal_evals::Int32; max_ps_evals::Int32; max_polish_evals::Int32; reserved::Int32
    seed::UInt64; t0::Float64; xtol_rel::Float64
end
struct MrbfPsInfo           # mirrors mrbf_ps_info, 32 bytes
    status::Int32; generations::Int32; evals_ideal::Int32; evals_ps::Int32; evals_polish::Int32; ms_total::Float32
    tau::Float64
end

const MRBF_KERNEL_ID = Dict(k => Int32(i - 1) for (i, k) in enumerate(RbfKernels))  # order of RbfModel.jl:48-54

# ---- context: one per thread, created under a lock; live models are registered so that shutdown never outruns them -------
mutable struct MrbfContext
    handle::Ptr{Cvoid}
    models::Set{Ptr{Cvoid}}        # handles of live models / round-4 states created through this context
    lock::ReentrantLock
    function MrbfContext(device::Integer = -1)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mrbf_init, libmrbf), Int32, (Int32, Ref{Ptr{Cvoid}}), device, h)
        rc == 0 || error("mrbf_init failed ($rc): ", unsafe_string(ccall((:mrbf_last_error, libmrbf), Cstring, (Ptr{Cvoid},), C_NULL)))
        ctx = new(h[], Set{Ptr{Cvoid}}(), ReentrantLock())
        finalizer(_shutdown!, ctx)
        return ctx
    end
end
function _shutdown!(ctx::MrbfContext)
    # finalizers may run on any thread: never block in one -- hand the work to a task if the lock is busy
    if !trylock(ctx.lock)
        @async _shutdown!(ctx)
        return nothing
    end
    try
        if ctx.handle != C_NULL
            for m in ctx.models   # models first: mrbf_shutdown deletes the context they would be released through
                ccall((:mrbf_free_model, libmrbf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.handle, m)
            end
            empty!(ctx.models)
            ccall((:mrbf_shutdown, libmrbf), Int32, (Ptr{Cvoid},), ctx.handle)
            ctx.handle = C_NULL   # later model finalizers see a closed context and do nothing
        end
    finally
        unlock(ctx.lock)
    end
    return nothing
end

const _CTX = Dict{Int,MrbfContext}()
const _CTX_LOCK = ReentrantLock()
"The calling thread's context (a ctx is not thread-safe; ctxs are independent)."
mrbf_context() = lock(_CTX_LOCK) do
    get!(() -> MrbfContext(), _CTX, Threads.threadid())
end

function _check(ctx::MrbfContext, rc::Int32)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:mrbf_last_error, libmrbf), Cstring, (Ptr{Cvoid},), ctx.handle))
    # MRBF_ENOTPD = 1, MRBF_ESINGULAR = 2: numerical failures the algorithm can react to (rebuild / not fully linear)
    rc in (1, 2) ? throw(LinearAlgebra.SingularException(Int(rc))) : error("libmrbf error $rc: $msg")
end

# ---- config: every RbfConfig field, same defaults and the same six assertions (RbfModel.jl:66-112) ------------------------
@with_kw struct HipRbfConfig <: AbstractSurrogateConfig
    kernel::Symbol = :cubic
    shape_parameter::Union{String,Float64} = NaN
    polynomial_degree::Int64 = 1
    θ_enlarge_1::Float64 = 2
    θ_enlarge_2::Float64 = 2
    θ_pivot::Float64 = 1 / (2 * θ_enlarge_1)
    θ_pivot_cholesky::Float64 = 1e-7
    require_linear::Bool = true
    max_model_points::Int64 = -1
    use_max_points::Bool = false
    optimized_sampling = true
    max_evals::Int64 = typemax(Int64)
    @assert θ_enlarge_1 * θ_pivot ≤ 1 "θ_pivot must be <= θ_enlarge_1^(-1)."
    @assert kernel ∈ RbfKernels "`kernel` not supported. See `Morbit.RbfKernels` for available symbols."
    @assert kernel != :thin_plate_spline || shape_parameter isa String || isnan(shape_parameter) ||
            (shape_parameter % 1 == 0 && shape_parameter >= 1) "Invalid shape_parameter for :thin_plate_spline."
    @assert kernel != :cubic || shape_parameter isa String || isnan(shape_parameter) ||
            (shape_parameter % 1 == 0 && shape_parameter % 2 == 1) "Invalid shape_parameter for :cubic."
    @assert shape_parameter isa String || isnan(shape_parameter) || shape_parameter > 0 "Shape parameter must be strictly positive."
    @assert θ_enlarge_1 >= 1 && θ_enlarge_2 >= 1 "θ's must be >= 1."
    @assert -1 <= polynomial_degree <= 1
end
# the sampling code only reads fields, so it works on either config type
_as_rbf_config(cfg::HipRbfConfig) = RbfConfig(; (fn => getfield(cfg, fn) for fn in fieldnames(RbfConfig))...)

max_evals(cfg::HipRbfConfig)::Int = cfg.max_evals
combinable(cfg::HipRbfConfig)::Bool = true
Base.hash(cfg::HipRbfConfig, h::UInt) = hash(Tuple(getfield(cfg, fn) for fn in fieldnames(HipRbfConfig)), h)
Base.isequal(a::HipRbfConfig, b::HipRbfConfig) = all(isequal(getfield(a, fn), getfield(b, fn)) for fn in fieldnames(HipRbfConfig))
Base.:(==)(a::HipRbfConfig, b::HipRbfConfig) = isequal(a, b)
get_saveable_type(::HipRbfConfig, x::AbstractVector{F}, y) where {F<:AbstractFloat} = RbfMeta{F,Nothing}

# (kernel id, a, b) from _get_kernel_params (RbfModel.jl:665-690); NaN -> the package defaults
function _mrbf_kernel_params(Δ, cfg)
    p = _get_kernel_params(Δ, _as_rbf_config(cfg))
    kid = MRBF_KERNEL_ID[cfg.kernel]
    cfg.kernel == :gaussian && return kid, Float64(something(p, 1.0)), 0.0
    cfg.kernel in (:multiquadric, :inv_multiquadric) && return kid, Float64(p === nothing ? 1.0 : p[1]), 0.5
    cfg.kernel == :cubic && return kid, Float64(something(p, 3)), 0.0
    return kid, Float64(something(p, 2)), 0.0   # :thin_plate_spline
end

# ---- model ------------------------------------------------------------------------------------------------------------------
mutable struct HipRbfModel <: AbstractSurrogate
    ctx::MrbfContext            # keeps the context alive as long as the model is reachable
    handle::Ptr{Cvoid}          # mrbf_model*, device resident centres / weights
    n_vars::Int
    num_outputs::Int
    fully_linear::Bool
    info::MrbfFitInfo
    function HipRbfModel(ctx, handle, n_vars, k, fl, info)
        m = new(ctx, handle, n_vars, k, fl, info)
        lock(ctx.lock) do
            push!(ctx.models, handle)
        end
        finalizer(_free_model!, m)
        return m
    end
end
function _free_model!(m::HipRbfModel)
    ctx = m.ctx
    if !trylock(ctx.lock)
        @async _free_model!(m)
        return nothing
    end
    try
        if ctx.handle != C_NULL && m.handle in ctx.models   # not yet released by the context's own shutdown
            delete!(ctx.models, m.handle)
            ccall((:mrbf_free_model, libmrbf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.handle, m.handle)
        end
        m.handle = C_NULL
    finally
        unlock(ctx.lock)
    end
    return nothing
end
fully_linear(m::HipRbfModel)::Bool = m.fully_linear
num_outputs(m::HipRbfModel) = m.num_outputs
set_fully_linear!(m::HipRbfModel, val) = (m.fully_linear = val; nothing)

# ---- two-phase construction: phase I (which sites) stays Morbit's control flow, its arithmetic goes to the device -------------
prepare_init_model(cfg::HipRbfConfig, args...; kwargs...) = prepare_init_model(_as_rbf_config(cfg), args...; kwargs...)
prepare_update_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) =
    prepare_update_model(nothing, meta, _as_rbf_config(cfg), args...; kwargs...)
prepare_improve_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) =
    prepare_improve_model(nothing, meta, _as_rbf_config(cfg), args...; kwargs...)

init_model(meta::RbfMeta, cfg::HipRbfConfig, func_indices, mop, scal, iter_data, sdb, ac; kwargs...) =
    update_model(nothing, meta, cfg, func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
improve_model(mod, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) = update_model(mod, meta, cfg, args...; kwargs...)

# sites / values in the ABI layouts: centres n x d row-major == the d x n column-major Matrix, values n x k row-major == k x n.
# (`reinterpret(reshape, Float64, ::Vector{SVector{d,Float64}})` would be a copy-free view of the same bytes; Morbit's sites may be
#  Float32 or plain Vectors, so the binding materialises one Float64 matrix per call: n d doubles, negligible next to the fit.)
_as_matrix(v) = Matrix{Float64}(reduce(hcat, v))

function update_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig,
                      func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
    db = get_sub_db(sdb, func_indices)
    Δ = get_delta(iter_data)
    training_results = get_result.(db, _collect_indices(meta))                 # RbfModel.jl:754-757
    C = _as_matrix(get_site.(training_results))
    Y = _as_matrix(get_value.(training_results))
    d, n = size(C)
    k = size(Y, 1)
    kid, a, b = _mrbf_kernel_params(Δ, cfg)
    ctx = mrbf_context()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    info = Ref{MrbfFitInfo}()
    GC.@preserve C Y begin
        rc = ccall((:mrbf_fit, libmrbf), Int32,
                   (Ptr{Cvoid}, Int64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64, Int32,
                    Ref{Ptr{Cvoid}}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfFitInfo}),
                   ctx.handle, n, d, k, C, Y, kid, a, b, cfg.polynomial_degree, h, C_NULL, C_NULL, info)
    end
    _check(ctx, rc)
    @logmsg loglevel3 "The model is $(meta.fully_linear ? "" : "not ")fully linear (solve path $(info[].path), residual $(info[].rel_residual))."
    return HipRbfModel(ctx, h[], d, k, meta.fully_linear, info[]), meta
end

# ---- evaluation: single site (reference API) and batched twins -----------------------------------------------------------------
function _mrbf_eval(mod::HipRbfModel, X::Matrix{Float64}; values::Bool = true, jac::Bool = false)
    d, m, k = mod.n_vars, size(X, 2), mod.num_outputs            # X is d x m column-major == m x d row-major
    V = values ? Matrix{Float64}(undef, k, m) : nothing          # k x m column-major == m x k row-major
    J = jac ? Array{Float64,3}(undef, k, d, m) : nothing         # per point a k x d column-major block
    GC.@preserve X V J begin
        rc = ccall((:mrbf_eval, libmrbf), Int32,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                   mod.ctx.handle, mod.handle, m, X, values ? pointer(V) : C_NULL, jac ? pointer(J) : C_NULL, C_NULL)
    end
    _check(mod.ctx, rc)
    return V, J
end

"Evaluate `mod` at scaled site `x̂` (RbfModel.jl:783-785)."
eval_models(mod::HipRbfModel, scal::AbstractVarScaler, x̂::Vec) = vec(_mrbf_eval(mod, reshape(Vector{Float64}(x̂), :, 1))[1])
"Evaluate output(s) `ℓ` (RbfModel.jl:788-790; RefSurrogate passes index vectors)."
eval_models(mod::HipRbfModel, scal::AbstractVarScaler, x̂::Vec, ℓ) = eval_models(mod, scal, x̂)[ℓ]
"k x d Jacobian (or rows) at `x̂` (RbfModel.jl:797-800)."
function get_jacobian(mod::HipRbfModel, scal::AbstractVarScaler, x̂::Vec, rows = nothing)
    J = _mrbf_eval(mod, reshape(Vector{Float64}(x̂), :, 1); values = false, jac = true)[2][:, :, 1]
    return isnothing(rows) ? J : J[rows, :]
end
"Gradient of output `ℓ` (RbfModel.jl:792-795): the Jacobian row, bit for bit (test/rbf_models.jl:105-109)."
get_gradient(mod::HipRbfModel, scal::AbstractVarScaler, x̂::Vec, ℓ) = vec(get_jacobian(mod, scal, x̂, ℓ))

# batched twins (m sites as the columns of X): one device sweep for all k outputs of all sites
eval_models_at_sites(mod::HipRbfModel, scal, X::AbstractMatrix) = _mrbf_eval(mod, Matrix{Float64}(X))[1]
get_jacobians_at_sites(mod::HipRbfModel, scal, X::AbstractMatrix) = _mrbf_eval(mod, Matrix{Float64}(X); values = false, jac = true)[2]

# ---- container twins: eval_container_{objectives,nl_eq_constraints,nl_ineq_constraints}[_jacobian]_at_scaled_sites ------------
# (SurrogateContainer.jl:234-269 does one inner-model call per objective index; here every distinct grouped HipRbfModel is swept
#  ONCE for all sites, RefSurrogates select rows, CompositeSurrogates apply their outer function / chain rule per site,
#  AbstractSurrogateInterface.jl:136-154, :175-229.)  Sites are the columns of X_scaled (d x m).
_inner(s::RefSurrogate) = s.model_ref[]
_inner(s::CompositeSurrogate) = s.model_ref[]
function _sweep_groups(surrogates, scal, X::Matrix{Float64}; jac::Bool)
    cache = IdDict{Any,Any}()
    for s in surrogates
        m = _inner(s)
        haskey(cache, m) && continue
        cache[m] = m isa HipRbfModel ? _mrbf_eval(m, X; values = true, jac = jac) : nothing
    end
    return cache
end
function _container_values_at_sites(surrogates, scal, X_scaled::AbstractMatrix)
    X = Matrix{Float64}(X_scaled)
    m = size(X, 2)
    isempty(surrogates) && return Matrix{MIN_PRECISION}(undef, 0, m)          # SurrogateContainer.jl:265
    cache = _sweep_groups(surrogates, scal, X; jac = false)
    rows = map(surrogates) do s
        sweep = cache[_inner(s)]
        if sweep === nothing                                                  # some other surrogate family: site by site
            reduce(hcat, [eval_models(s, scal, X[:, p]) for p = 1:m])
        elseif s isa RefSurrogate
            sweep[1][s.output_indices, :]
        else                                                                  # CompositeSurrogate: φ([T(x); g(x)]) per site
            reduce(hcat, [eval_vfun(s.outer_ref[], [untransform(X[:, p], scal); sweep[1][s.inner_output_indices, p]]) for p = 1:m])
        end
    end
    return reduce(vcat, rows)                                                 # Σk x m
end
function _container_jacobians_at_sites(surrogates, scal, X_scaled::AbstractMatrix)
    X = Matrix{Float64}(X_scaled)
    d, m = size(X)
    isempty(surrogates) && return Array{MIN_PRECISION,3}(undef, 0, d, m)      # SurrogateContainer.jl:259
    cache = _sweep_groups(surrogates, scal, X; jac = true)
    blocks = map(surrogates) do s
        sweep = cache[_inner(s)]
        if sweep === nothing
            cat([get_jacobian(s, scal, X[:, p]) for p = 1:m]...; dims = 3)
        elseif s isa RefSurrogate
            sweep[2][s.output_indices, :, :]
        else
            cat([begin
                     gx = [untransform(X[:, p], scal); sweep[1][s.inner_output_indices, p]]
                     _composite_jac(_get_jacobian(s.outer_ref[], gx), sweep[2][s.inner_output_indices, :, p], scal, X[:, p])
                 end for p = 1:m]...; dims = 3)
        end
    end
    return reduce(vcat, blocks)                                               # Σk x d x m
end
for (fld, plural) in ((:objectives, :objectives), (:nl_eq_constraints, :nl_eq_constraints), (:nl_ineq_constraints, :nl_ineq_constraints))
    vals = Symbol("eval_container_", plural, "_at_scaled_sites")
    jacs = Symbol("eval_container_", plural, "_jacobian_at_scaled_sites")
    @eval begin
        $vals(sc::SurrogateContainer, scal, X::AbstractMatrix) = _container_values_at_sites(getfield(sc, $(QuoteNode(fld))), scal, X)
        $jacs(sc::SurrogateContainer, scal, X::AbstractMatrix) = _container_jacobians_at_sites(getfield(sc, $(QuoteNode(fld))), scal, X)
    end
end

# the objectives' single grouped model when every objective is a RefSurrogate of it, outputs in order (then the C entry points that
# take one model apply); nothing otherwise
function _single_objective_model(sc::SurrogateContainer)
    objs = sc.objectives
    (isempty(objs) || !all(o -> o isa RefSurrogate, objs)) && return nothing
    m = _inner(first(objs))
    (m isa HipRbfModel && all(o -> _inner(o) === m, objs)) || return nothing
    return reduce(vcat, [o.output_indices for o in objs]) == collect(1:num_outputs(m)) ? m : nothing
end

# ---- descent consumers ------------------------------------------------------------------------------------------------------------
"All Armijo step sizes of `_backtrack` (descent.jl:150-185) in one batch; returns (x₊, mx₊, step) like the reference."
function _backtrack(x::AbstractVector, dir::AbstractVector, step_size, ω, sc::SurrogateContainer, cfg, scal; model::HipRbfModel = _single_objective_model(sc))
    x = Vector{Float64}(x); dir = Vector{Float64}(dir)
    d, k = model.n_vars, model.num_outputs
    x₊, mx₊, step, loops = similar(x), Vector{Float64}(undef, k), similar(x), Ref{Int32}(0)
    GC.@preserve x dir x₊ mx₊ step begin
        rc = ccall((:mrbf_backtrack, libmrbf), Int32,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int32, Float64, Float64, Float64, Int32,
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                   model.ctx.handle, model.handle, x, dir, step_size, ω, cfg.strict_backtracking, cfg.armijo_const_rhs,
                   cfg.armijo_const_shrink, cfg.min_stepsize >= 0 ? cfg.min_stepsize : eps(Float64), cfg.max_loops,
                   x₊, mx₊, step, loops)
    end
    _check(model.ctx, rc)
    return x₊, mx₊, step
end

"""
Pascoletti-Serafini descent step (descent.jl:512-581) with the subproblem solver on the device (`mrbf_ps_step`): applies when the
objectives are the outputs of one grouped `HipRbfModel` and the problem carries no modelled or MOP constraints; otherwise call
Morbit's own method (NLopt with the one-point handles of `get_objectives_optim_handles`).  Same returns as the reference.
"""
function get_criticality(desc_cfg::PascolettiSerafiniConfig, mop, scal, x_it, x_it_n, data_base, sc::SurrogateContainer, algo_config;
                         model::Union{Nothing,HipRbfModel} = _single_objective_model(sc), seed::UInt64 = rand(UInt64))
    constrained = !isempty(sc.nl_eq_constraints) || !isempty(sc.nl_ineq_constraints) ||
                  !isempty(get_eq_constraints_optim_handles(mop, scal)) || !isempty(get_ineq_constraints_optim_handles(mop, scal))
    if model === nothing || constrained
        return invoke(get_criticality, Tuple{PascolettiSerafiniConfig,Any,Any,Any,Any,Any,Any,Any}, desc_cfg, mop, scal, x_it, x_it_n, data_base, sc, algo_config)
    end
    x = Vector{Float64}(get_x_scaled(x_it)); x_n = Vector{Float64}(get_x_scaled(x_it_n)); fx_n = Vector{Float64}(get_fx(x_it_n))
    d, k = length(x_n), model.num_outputs
    lb_eff, ub_eff = local_bounds(scal, x, get_delta(x_it))
    lb = Vector{Float64}(lb_eff); ub = Vector{Float64}(ub_eff)
    r = _get_global_dir(desc_cfg, fx_n)                                        # descent.jl:360-368
    rbuf = r === nothing ? nothing : Vector{Float64}(r)
    g_evals, l_evals = _ps_max_evals(desc_cfg, d)                              # descent.jl:414-432
    opts = Ref(MrbfPsOptions(desc_cfg.max_ideal_point_problem_evals, g_evals, l_evals, 0, seed, -0.5, 1e-3))
    info = Ref{MrbfPsInfo}()
    x_trial, mx_trial = Vector{Float64}(undef, d), Vector{Float64}(undef, k)
    GC.@preserve x_n lb ub fx_n rbuf x_trial mx_trial begin
        rc = ccall((:mrbf_ps_step, libmrbf), Int32,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfPsOptions},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfPsInfo}),
                   model.ctx.handle, model.handle, x_n, lb, ub, fx_n, rbuf === nothing ? C_NULL : pointer(rbuf), opts,
                   x_trial, mx_trial, C_NULL, info)
    end
    _check(model.ctx, rc)
    Xet = eltype(get_x_scaled(x_it_n))
    info[].status == 1 && return 0, copy(get_x_scaled(x_it_n)), mx_trial, 0    # critical: some r_l <= 0 (descent.jl:546-549)
    info[].status == 2 && return 0, copy(get_x_scaled(x_it)), mx_trial, 0      # failure (descent.jl:571-572)
    return Xet(abs(info[].tau)), (Xet.(x_trial), mx_trial, norm(x .- x_trial, Inf))
end

# ---- site selection on the device -----------------------------------------------------------------------------------------------------
mutable struct HipRound4State
    ctx::MrbfContext
    handle::Ptr{Cvoid}
end
"""
`_rbf_round4` (RbfModel.jl:352-499) as one device call.  `start_sites` / `cand_sites`: vectors of sites (the sites found so far; the
box candidates in database order, followed -- with `use_max_points` -- by the random box points the caller drew).  Returns the
positions of the accepted candidates in acceptance order and, with `keep_state`, the factors for `fit_from_round4`.
"""
function rbf_round4_device(cfg::HipRbfConfig, Δ, start_sites, cand_sites; keep_state::Bool = false)
    C0 = _as_matrix(start_sites); Xc = isempty(cand_sites) ? Matrix{Float64}(undef, size(C0, 1), 0) : _as_matrix(cand_sites)
    d, n0 = size(C0); mc = size(Xc, 2)
    kid, a, b = _mrbf_kernel_params(Δ, cfg)
    ctx = mrbf_context()
    acc = Vector{Int32}(undef, max(mc, 1)); nacc = Ref{Int32}(0); st = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve C0 Xc acc begin
        rc = ccall((:mrbf_round4, libmrbf), Int32,
                   (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int32, Float64, Float64, Int32, Int32, Float64,
                    Ptr{Int32}, Ref{Int32}, Ptr{Ptr{Cvoid}}),
                   ctx.handle, n0, d, C0, mc, Xc, kid, a, b, cfg.polynomial_degree, cfg.max_model_points, cfg.θ_pivot_cholesky,
                   acc, nacc, keep_state ? Base.unsafe_convert(Ptr{Ptr{Cvoid}}, st) : C_NULL)
    end
    _check(ctx, rc)
    accepted = Int.(acc[1:nacc[]]) .+ 1                                         # 1-based positions into cand_sites
    keep_state || return accepted
    state = HipRound4State(ctx, st[])
    finalizer(s -> (s.handle != C_NULL && s.ctx.handle != C_NULL &&
                    ccall((:mrbf_free_round4, libmrbf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), s.ctx.handle, s.handle); s.handle = C_NULL), state)
    return accepted, state
end
"The model on (start sites, accepted sites) from the factor round 4 kept (RbfModel.jl:657-660): `values` in that order, k x n."
function fit_from_round4(state::HipRound4State, values, n_vars::Int, fully_linear::Bool)
    Y = _as_matrix(values); k = size(Y, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL); info = Ref{MrbfFitInfo}()
    GC.@preserve Y begin
        rc = ccall((:mrbf_fit_from_round4, libmrbf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Float64}, Ref{Ptr{Cvoid}}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfFitInfo}),
                   state.ctx.handle, state.handle, k, Y, h, C_NULL, C_NULL, info)
    end
    _check(state.ctx, rc)
    return HipRbfModel(state.ctx, h[], n_vars, k, fully_linear, info[])
end
"Scores `‖Z (Zᵀ(ξ - x₀))‖_p` of all filter candidates and the first maximiser (AffinelyIndependentPoints.jl:71-106)."
function affine_scores(shifted_seeds, Z::AbstractMatrix, p = Inf)
    S = _as_matrix(shifted_seeds); d, mc = size(S); Zm = Matrix{Float64}(Z)
    ctx = mrbf_context(); best = Ref{Int64}(-1); val = Ref{Float64}(-Inf)
    GC.@preserve S Zm begin
        rc = ccall((:mrbf_affine_scores, libmrbf), Int32, (Ptr{Cvoid}, Int64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ref{Int64}, Ref{Float64}),
                   ctx.handle, mc, d, size(Zm, 2), S, Zm, isinf(p) ? 1 : 0, C_NULL, best, val)
    end
    _check(ctx, rc)
    return Int(best[]) + 1, val[]
end
