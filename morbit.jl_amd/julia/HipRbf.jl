# HipRbf.jl -- drop-in surrogate family for Morbit.jl backed by libmrbf.so (MI355X).
#
# NOT EXECUTED IN THIS REPOSITORY'S BUILD CONTAINER (no Julia there): this is the binding a Morbit maintainer
# adds next to src/models/RbfModel.jl (`include("models/HipRbf.jl")` in src/Morbit.jl after :87).  It implements
# the surrogate interface of src/AbstractSurrogateInterface.jl for a new config type `HipRbfConfig` by
#   * reusing Morbit's own site selection unchanged (rounds 1-4, prepare_update_model, RbfModel.jl:518-655), and
#   * replacing the three calls into RadialBasisFunctionModels.jl on the hot path
#         RBF.RBFInterpolationModel(...)        RbfModel.jl:759-763   -> mrbf_fit
#         mod.model(x̂[, ℓ])                     RbfModel.jl:784, :789 -> mrbf_eval (values)
#         RBF.grad / RBF.jac                    RbfModel.jl:794, :799 -> mrbf_eval (Jacobian)
#     plus batched twins the descent code can call (`eval_models_at_sites`, `get_jacobians_at_sites`,
#     `backtrack_batched`) -- SURVEY.md section 8f rank 2.
# The same ABI is exercised end to end by tests/ through ctypes (morbit.jl_amd/_lib.py mirrors these ccalls 1:1).

const libmrbf = get(ENV, "MRBF_LIB", "libmrbf.so")

struct MrbfFitInfo          # mirrors mrbf_fit_info (include/mrbf.h), 64 bytes
    path::Int32; factor_info::Int32; n::Int32; q::Int32
    rel_residual::Float64; max_pitw::Float64; mu::Float64
    ms_gram::Float32; ms_project::Float32; ms_factor::Float32; ms_solve::Float32; ms_check::Float32; ms_total::Float32
end

const MRBF_KERNEL_ID = Dict(k => Int32(i - 1) for (i, k) in enumerate(RbfKernels))  # order of RbfModel.jl:48-54

mutable struct MrbfContext
    handle::Ptr{Cvoid}
    function MrbfContext(device::Integer = -1)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mrbf_init, libmrbf), Int32, (Int32, Ref{Ptr{Cvoid}}), device, h)
        rc == 0 || error("mrbf_init failed ($rc): ", unsafe_string(ccall((:mrbf_last_error, libmrbf), Cstring, (Ptr{Cvoid},), C_NULL)))
        ctx = new(h[])
        finalizer(c -> ccall((:mrbf_shutdown, libmrbf), Int32, (Ptr{Cvoid},), c.handle), ctx)
        return ctx
    end
end

# one context per Julia thread (a ctx is not thread-safe; ctxs are independent)
const _CTX = Dict{Int,MrbfContext}()
mrbf_context() = get!(() -> MrbfContext(), _CTX, Threads.threadid())

function _check(ctx::MrbfContext, rc::Int32)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:mrbf_last_error, libmrbf), Cstring, (Ptr{Cvoid},), ctx.handle))
    # MRBF_ENOTPD = 1, MRBF_ESINGULAR = 2: numerical failures the algorithm can react to (rebuild / not fully linear)
    rc in (1, 2) ? throw(LinearAlgebra.SingularException(Int(rc))) : error("libmrbf error $rc: $msg")
end

# ---- config: every RbfConfig field, same defaults and assertions (RbfModel.jl:66-112) -------------------------------
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
    @assert -1 <= polynomial_degree <= 1
end
# the sampling code only reads fields, so it works on either config type
_as_rbf_config(cfg::HipRbfConfig) = RbfConfig(; (fn => getfield(cfg, fn) for fn in fieldnames(RbfConfig))...)

max_evals(cfg::HipRbfConfig)::Int = cfg.max_evals
combinable(cfg::HipRbfConfig)::Bool = true
Base.hash(cfg::HipRbfConfig, h::UInt) = hash(getfield.(cfg, Tuple(fn for fn ∈ fieldnames(HipRbfConfig))), h)
Base.isequal(a::HipRbfConfig, b::HipRbfConfig) = all(isequal(getfield(a, fn), getfield(b, fn)) for fn in fieldnames(HipRbfConfig))
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

# ---- model ------------------------------------------------------------------------------------------------------------
mutable struct HipRbfModel <: AbstractSurrogate
    ctx::MrbfContext
    handle::Ptr{Cvoid}          # mrbf_model*, device resident centres / weights
    n_vars::Int
    num_outputs::Int
    fully_linear::Bool
    info::MrbfFitInfo
    function HipRbfModel(ctx, handle, n_vars, k, fl, info)
        m = new(ctx, handle, n_vars, k, fl, info)
        finalizer(x -> ccall((:mrbf_free_model, libmrbf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.handle), m)
        return m
    end
end
fully_linear(m::HipRbfModel)::Bool = m.fully_linear
num_outputs(m::HipRbfModel) = m.num_outputs
set_fully_linear!(m::HipRbfModel, val) = (m.fully_linear = val; nothing)

# ---- two-phase construction: phase I is Morbit's own (sites only), phase II calls the GPU ------------------------------
prepare_init_model(cfg::HipRbfConfig, args...; kwargs...) = prepare_init_model(_as_rbf_config(cfg), args...; kwargs...)
prepare_update_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) =
    prepare_update_model(nothing, meta, _as_rbf_config(cfg), args...; kwargs...)
prepare_improve_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) =
    prepare_improve_model(nothing, meta, _as_rbf_config(cfg), args...; kwargs...)

init_model(meta::RbfMeta, cfg::HipRbfConfig, func_indices, mop, scal, iter_data, sdb, ac; kwargs...) =
    update_model(nothing, meta, cfg, func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
improve_model(mod, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) = update_model(mod, meta, cfg, args...; kwargs...)

function update_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig,
                      func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
    db = get_sub_db(sdb, func_indices)
    Δ = get_delta(iter_data)
    training_results = get_result.(db, _collect_indices(meta))                 # RbfModel.jl:754-757
    sites = get_site.(training_results)                                        # Vector{SVector{d,F}} (or Vector{Vector})
    vals = get_value.(training_results)
    n, d, k = length(sites), length(first(sites)), length(first(vals))
    # zero-copy views in the ABI layouts: centres n x d row-major == d x n column-major, values n x k row-major == k x n
    C = d <= 64 && eltype(sites) <: StaticArrays.SVector ? reinterpret(reshape, Float64, sites) : reduce(hcat, sites)
    Y = reduce(hcat, vals)
    C = Matrix{Float64}(C); Y = Matrix{Float64}(Y)
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

# ---- evaluation: single site (reference API) and batched twins -----------------------------------------------------------
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

"All Armijo step sizes of `_backtrack` (descent.jl:150-185) in one batch; returns (x₊, mx₊, step, loops)."
function backtrack_batched(mod::HipRbfModel, x::Vector{Float64}, dir::Vector{Float64}, step_size, ω, cfg)
    d, k = mod.n_vars, mod.num_outputs
    x₊, mx₊, step, loops = similar(x), Vector{Float64}(undef, k), similar(x), Ref{Int32}(0)
    GC.@preserve x dir x₊ mx₊ step begin
        rc = ccall((:mrbf_backtrack, libmrbf), Int32,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int32, Float64, Float64, Float64, Int32,
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                   mod.ctx.handle, mod.handle, x, dir, step_size, ω, cfg.strict_backtracking, cfg.armijo_const_rhs,
                   cfg.armijo_const_shrink, cfg.min_stepsize >= 0 ? cfg.min_stepsize : eps(Float64), cfg.max_loops,
                   x₊, mx₊, step, loops)
    end
    _check(mod.ctx, rc)
    return x₊, mx₊, step, Int(loops[])
end
