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
#     get_criticality(::PascolettiSerafiniConfig)   descent.jl:512-581                -> mrbf_ps_step_problem
# Which of Morbit's calls go to the device and which to Morbit's own methods is decided by the mrbf_dispatch_* functions of the
# library (include/mrbf.h, "decision table"), the same ones the Python mirror calls; no method here raises because of a size limit.
# Every ccall below has a 1:1 ctypes twin in morbit.jl_amd/_lib.py, which is what tests/ execute; struct layouts are pinned by
# tests/test_abi.py; tests/test_julia_binding.py parses every ccall of this file against include/mrbf.h.
# Conventions: a context is owned by one Julia task at a time (one per thread, created under a lock) and is NOT thread-safe, while
# finalizers run on whichever thread triggers the GC -- so EVERY ccall that passes a context handle runs inside `_locked(ctx)`, i.e.
# holding `ctx.lock` (finalizers `trylock` and re-schedule themselves); models keep their context alive and are released through
# it; buffers passed to ccall are GC.@preserve'd.
# INERT FOR NON-USERS: nothing in this file changes a run that does not select `HipRbfConfig`.  The three methods that extend Morbit
# functions on Morbit's own types (`Base.iterate` of the affine filter, `_backtrack`, `get_criticality` of the PS step) first look
# for a HipRbf object -- the task-local scan state set by `prepare_update_model(..., ::HipRbfConfig, ...)`, a `HipRbfModel` in the
# container -- and hand over to Morbit's method by `invoke` before anything of libmrbf is touched when there is none.

const libmrbf = get(ENV, "MRBF_LIB", "libmrbf.so")

struct MrbfFitInfo          # mirrors mrbf_fit_info (include/mrbf.h), 80 bytes
    path::Int32; factor_info::Int32; n::Int32; q::Int32
    rel_residual::Float64; max_pitw::Float64; mu::Float64
    ms_gram::Float32; ms_project::Float32; ms_factor::Float32; ms_solve::Float32; ms_check::Float32; ms_total::Float32
    fallbacks::Int32; giveup_code::Int32
    ms_factor_device::Float32; slow_launches::Int32
end
struct MrbfPsOptions        # mirrors mrbf_ps_options, 40 bytes
    max_ideal_evals::Int32; max_ps_evals::Int32; max_polish_evals::Int32; reserved::Int32
    seed::UInt64; t0::Float64; xtol_rel::Float64
end
struct MrbfPsInfo           # mirrors mrbf_ps_info, 32 bytes
    status::Int32; generations::Int32; evals_ideal::Int32; evals_ps::Int32; evals_polish::Int32; ms_total::Float32
    tau::Float64
end

struct MrbfPsProblem        # mirrors mrbf_ps_problem, 72 bytes
    n_models::Int32; n_objectives::Int32
    models::Ptr{Ptr{Cvoid}}; roles::Ptr{Int32}
    n_lin_eq::Int32; n_lin_ineq::Int32
    A_eq::Ptr{Float64}; b_eq::Ptr{Float64}; A_ineq::Ptr{Float64}; b_ineq::Ptr{Float64}
    eq_tol::Float64
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

"""
Run `f(handle)` holding the context's lock.  A context is documented as not thread-safe (include/mrbf.h) and model / round-4
finalizers may run on any thread (whichever triggers the GC, e.g. under the `Threads.@threads` loop of
examples/large_scale_benchmarks.jl:253): they `trylock` this same lock and re-schedule themselves when it is busy, so a handle is
never released from thread B while thread A is inside a compute call of the same context.  Reentrant.
"""
function _locked(f, ctx::MrbfContext)
    lock(ctx.lock)
    try
        ctx.handle == C_NULL && error("libmrbf context was shut down")
        return f(ctx.handle)
    finally
        unlock(ctx.lock)
    end
end

function _check(ctx::MrbfContext, rc::Int32)
    rc == 0 && return nothing
    msg = _locked(ctx) do h
        unsafe_string(ccall((:mrbf_last_error, libmrbf), Cstring, (Ptr{Cvoid},), h))
    end
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

# ---- the decision table (include/mrbf.h, "decision table"): pure host functions of libmrbf that the Python mirror calls as well,
#      so both bindings route every call the same way; 1 = device entry point, 0 = Morbit's own method
_dispatch_ps(d, k, n_models, n_nl, n_lin, n_foreign) =
    ccall((:mrbf_dispatch_ps, libmrbf), Int32, (Int32, Int32, Int32, Int32, Int32, Int32), d, k, n_models, n_nl, n_lin, n_foreign) == 1
_dispatch_backtrack(n_models, n_foreign, in_order::Bool) =
    ccall((:mrbf_dispatch_backtrack, libmrbf), Int32, (Int32, Int32, Int32), n_models, n_foreign, in_order) == 1
_dispatch_affine(n_candidates, d) = ccall((:mrbf_dispatch_affine, libmrbf), Int32, (Int64, Int32), n_candidates, d) == 1
_dispatch_round4(n0, d, deg, n_candidates) =
    ccall((:mrbf_dispatch_round4, libmrbf), Int32, (Int64, Int32, Int32, Int64), n0, d, deg, n_candidates) == 1
_dispatch_fit(n_training, n0, q, n_accepted, same_sites::Bool) =
    ccall((:mrbf_dispatch_fit, libmrbf), Int32, (Int64, Int64, Int32, Int32, Int32), n_training, n0, q, n_accepted, same_sites) == 1
"Does return code `rc` of device entry point `entry` (1 round4, 2 fit_from_round4, 3 ps_step) mean: take the reference method?"
_fallback_rc(entry, rc) = ccall((:mrbf_dispatch_after, libmrbf), Int32, (Int32, Int32), entry, rc) == 1

# ---- two-phase construction: phase I (which sites) stays Morbit's control flow (RbfModel.jl:506-655 is generic in `cfg`); the
#      pieces of it that are arithmetic are re-routed by dispatch on the config / element type:
#        _rbf_round4(..., cfg::HipRbfConfig)                       -> mrbf_round4, factors kept for the fit
#        iterate(::AffinelyIndependentPointFilter{Float64}, n)    -> mrbf_affine_scores for large candidate sets, ONLY while the
#                                                                    task-local scan state of a HipRbfConfig's update is set
_get_signature(cfg::HipRbfConfig) = (cfg.θ_pivot, cfg.θ_enlarge_1, cfg.θ_enlarge_2, cfg.optimized_sampling)   # RbfModel.jl:114

function prepare_init_model(cfg::HipRbfConfig, func_indices, mop, scal, id, sdb, ac; ensure_fully_linear = true, kwargs...)
    F = eltype(get_x_scaled(id))                                                 # RbfModel.jl:506-513
    meta = RbfMeta{F,typeof(func_indices)}(; signature = _get_signature(cfg), func_indices)
    return prepare_update_model(nothing, meta, cfg, func_indices, mop, scal, id, sdb, ac; ensure_fully_linear, kwargs...)
end
# Morbit's generic method (first argument Union{Nothing,RbfModel}, `cfg` untyped); run with `nothing` as the model it never looks at.
# Two methods so that neither is ambiguous with the generic one for a `nothing` model.
const _GENERIC_PREPARE_SIG = Tuple{Union{Nothing,RbfModel},RbfMeta,Any,Any,Any,Any,Any,Any,Any}
# While Morbit's generic method runs FOR A HipRbfConfig, the task carries a scan state: that -- not the element type of the filter --
# is what routes the affine filter's candidate scan (rounds 1-2) to the device.  The filter is built without a config
# (RbfModel.jl:219), so the config cannot be dispatched on there; a plain `RbfConfig` run never sets the key, on any task.
const _HIP_AFFINE_KEY = :HipRbf_affine_scan
# Householder QR of Y = [y_1 .. y_j] kept up to date as the filter picks sites, with the FULL orthogonal factor explicit: appending a
# column is one reflector on the trailing rows, Q <- Q diag(I_j, H) -- a rank-1 update of Q's trailing columns, O(d^2) per pick.  Morbit
# re-factors Y after every pick (`_orthogonal_complement_matrix`, AffinelyIndependentPoints.jl:4-11: O(d^3) per pick, O(d^4) per filter;
# 140-280 ms per model update at d = 128, more than round 4, the fit and the descent step together -- profiles/r06_iteration_c4.txt).
# Same reflectors as LAPACK's geqrf (larfg: beta = -sign(alpha) |x|, v_1 = 1): Q equals qr(Y)'s to rounding, the picks are Morbit's.
# (1:1 twin of `_GrowingQR` in morbit.jl_amd/sampling.py, which tests/test_sampling.py holds against the from-scratch factorisation.)
mutable struct HipRbfGrowingQR
    Q::Matrix{Float64}
    j::Int
end
function HipRbfGrowingQR(Y::AbstractMatrix)
    d = size(Y, 1)
    Q = size(Y, 2) == 0 ? Matrix{Float64}(I, d, d) : qr(Matrix{Float64}(Y)).Q * Matrix{Float64}(I, d, d)
    return HipRbfGrowingQR(Q, size(Y, 2))
end
function _append!(g::HipRbfGrowingQR, y::AbstractVector)
    d = size(g.Q, 1); j = g.j
    j >= d && return g
    Qt = view(g.Q, :, j+1:d)
    x = Qt' * Vector{Float64}(y)                       # the new column in the current basis, trailing part
    α = x[1]; xn = norm(view(x, 2:length(x)))
    if xn != 0
        β = -copysign(hypot(α, xn), α)
        τ = (β - α) / β
        v = x ./ (α - β); v[1] = 1.0
        Qt .-= (Qt * v) * (τ .* v)'                    # Q[:, j+1:d] H
    end
    g.j = j + 1
    return g
end
function _complement(g::HipRbfGrowingQR, p)
    Z = g.Q[:, g.j+1:end]
    size(Z, 2) > 0 && (Z ./= norm.(eachcol(Z), p)')   # AffinelyIndependentPoints.jl:7-9
    return Z
end

mutable struct HipRbfPickQueue               # what ONE mrbf_affine_select call picked for a filter, handed out pick by pick
    picks::Vector{Int}
    pos::Int
    Z::Matrix{Float64}                      # the filter's final complement basis
end
mutable struct HipRbfAffineScan
    seeds::IdDict{Any,Matrix{Float64}}      # filter => d x mc matrix of shifted seeds, picked columns zeroed (task-local, no global)
    qrs::IdDict{Any,HipRbfGrowingQR}        # filter => the growing factorisation of its Y (host scan)
    queues::IdDict{Any,HipRbfPickQueue}     # filter => the device's picks (device scan)
end
HipRbfAffineScan() = HipRbfAffineScan(IdDict{Any,Matrix{Float64}}(), IdDict{Any,HipRbfGrowingQR}(), IdDict{Any,HipRbfPickQueue}())
_hip_affine_scan() = get(task_local_storage(), _HIP_AFFINE_KEY, nothing)
_prepare_with_device_scan(meta, cfg::HipRbfConfig, args...; kwargs...) =
    task_local_storage(_HIP_AFFINE_KEY, HipRbfAffineScan()) do
        invoke(prepare_update_model, _GENERIC_PREPARE_SIG, nothing, meta, cfg, args...; kwargs...)
    end
prepare_update_model(mod::Nothing, meta::RbfMeta, cfg::HipRbfConfig, func_indices, mop, scal, iter_data, sdb, ac; kwargs...) =
    _prepare_with_device_scan(meta, cfg, func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
prepare_update_model(mod::HipRbfModel, meta::RbfMeta, cfg::HipRbfConfig, func_indices, mop, scal, iter_data, sdb, ac; kwargs...) =
    _prepare_with_device_scan(meta, cfg, func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
# the improvement step only reads config fields (RbfModel.jl:699-732)
prepare_improve_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) =
    prepare_improve_model(nothing, meta, _as_rbf_config(cfg), args...; kwargs...)

init_model(meta::RbfMeta, cfg::HipRbfConfig, func_indices, mop, scal, iter_data, sdb, ac; kwargs...) =
    update_model(nothing, meta, cfg, func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
improve_model(mod, meta::RbfMeta, cfg::HipRbfConfig, args...; kwargs...) = update_model(mod, meta, cfg, args...; kwargs...)

# sites / values in the ABI layouts: centres n x d row-major == the d x n column-major Matrix, values n x k row-major == k x n.
# A Vector{SVector{d,Float64}} / Vector{MVector{k,Float64}} already IS those bytes: reinterpret it (no copy); anything else
# (Float32 sites, plain Vectors) is materialised once.
_as_matrix(v::Vector{StaticArrays.SVector{N,Float64}}) where {N} = reshape(reinterpret(Float64, v), N, length(v))   # zero-copy view
_as_matrix(v) = Matrix{Float64}(reduce(hcat, v))      # MVectors (heap objects), Float32 sites, plain Vectors: one d x n copy
_dense(A::Matrix{Float64}) = A
_dense(A::Base.ReshapedArray{Float64,2,<:Base.ReinterpretArray{Float64}}) = A   # dense, stride 1: ccall takes its pointer as it is
_dense(A) = Matrix{Float64}(A)

# ---- round 4 (RbfModel.jl:352-499) inside Morbit's own prepare_update_model: same arguments, same return value (database indices
#      of the accepted sites in acceptance order); the whole selection is ONE device call and its factor is kept for update_model
const _ROUND4_KEPT = IdDict{Any,Any}()          # sub-database => (state, training indices in factor order)
const _ROUND4_LOCK = ReentrantLock()
function _rbf_round4(db, lb_2, ub_2, x::AbstractVector{F}, Δ, indices_found_so_far, cfg::HipRbfConfig) where {F}
    lock(_ROUND4_LOCK) do
        delete!(_ROUND4_KEPT, db)               # whatever was kept belongs to an older training set
    end
    n_vars = length(x)
    max_points = cfg.max_model_points <= 0 ? Int(((n_vars + 1) * (n_vars + 2)) / 2) : cfg.max_model_points   # :356
    N = length(indices_found_so_far)
    candidate_indices_4 = results_in_box_indices(db, lb_2, ub_2, indices_found_so_far)
    (N < max_points && (!isempty(candidate_indices_4) || cfg.use_max_points)) || return Int[]                  # :367
    # with use_max_points the reference draws random box points one by one once the database candidates are used up (:405-416, at
    # most 10 max_points + 1); here they are drawn up front so that they ride in the same device call
    fresh = cfg.use_max_points ? [_rand_box_point(lb_2, ub_2, F) for _ = 1:(10 * max_points + 1)] : Vector{Vector{F}}()
    cand_sites = [get_site.(db, candidate_indices_4); fresh]
    if _dispatch_round4(N, n_vars, cfg.polynomial_degree, length(cand_sites))
        rc, accepted, state = rbf_round4_device(cfg, Δ, get_site.(db, indices_found_so_far), cand_sites; keep_state = true, rc_only = true)
        if rc == 0
            round4_indices = Int[]
            for pos in accepted
                id = pos <= length(candidate_indices_4) ? candidate_indices_4[pos] : new_result!(db, cand_sites[pos], F[])   # :455-457
                push!(round4_indices, id)
            end
            if state !== nothing
                lock(_ROUND4_LOCK) do
                    _ROUND4_KEPT[db] = (state, [collect(Int, indices_found_so_far); round4_indices])
                end
            end
            return round4_indices
        end
        _fallback_rc(1, rc) || _check(mrbf_context(), rc)       # a start set without the tail / rank deficient: Morbit's own loop
    end
    return _rbf_round4(db, lb_2, ub_2, x, Δ, indices_found_so_far, _as_rbf_config(cfg))
end

# ---- rounds 1-2: the candidate scan of the affine filter (AffinelyIndependentPoints.jl:71-106) with many candidates; picks, order
#      and the Y / Z bookkeeping are the reference's.  Routed by the task-local scan state of a HipRbfConfig's model update: without
#      it (every plain RbfConfig run) the first statement hands over to Morbit's own method -- no ccall, no library needed.
function Base.iterate(filter::AffinelyIndependentPointFilter{Float64,VF,SV}, num_found::Int) where {VF,SV}
    scan = _hip_affine_scan()
    scan === nothing && return invoke(Base.iterate, Tuple{AffinelyIndependentPointFilter,Int}, filter, num_found)
    done() = (delete!(scan.seeds, filter); delete!(scan.qrs, filter); delete!(scan.queues, filter); nothing)
    num_found == filter.n && return done()
    isempty(filter.candidate_indices) && return done()
    S = get!(scan.seeds, filter) do
        M = Matrix{Float64}(_as_matrix(filter.shifted_seeds))    # a copy: picked columns are zeroed below
        for j in setdiff(eachindex(filter.shifted_seeds), filter.candidate_indices)
            M[:, j] .= 0                                   # chosen sites score 0 (the reference removes them from the list)
        end
        M
    end
    if haskey(scan.queues, filter) || _dispatch_affine(length(filter.candidate_indices), length(filter.x_0))
        # many candidates (decision table): the whole remaining selection is ONE device call (mrbf_affine_select: the factorisation of Y
        # grows by a reflector per pick on the device, no host round trip between picks); its picks are handed out one per call
        queue = get!(scan.queues, filter) do
            g = HipRbfGrowingQR(filter.Y)
            picks, Zf = affine_select(S, g.Q, g.j, filter.n - num_found, Float64(filter.pivot_val), filter.p)
            HipRbfPickQueue(picks, 0, Zf)
        end
        queue.pos >= length(queue.picks) && return done()
        queue.pos += 1
        i = queue.picks[queue.pos]
        filter.Y = hcat(filter.Y, filter.shifted_seeds[i])
        setdiff!(filter.candidate_indices, i)
        queue.pos == length(queue.picks) && (filter.Z = queue.Z)
        return (filter.return_indices ? i : filter.seeds[i]), num_found + 1
    end
    # few candidates: two host BLAS products per pick -- the same scores -- and the factorisation grown by a reflector per pick
    best_index, best_val = _affine_scores_host(S, filter.Z, filter.p)
    if best_index >= 1 && best_val > filter.pivot_val
        i = best_index
        g = get!(() -> HipRbfGrowingQR(filter.Y), scan.qrs, filter)     # (factor of the sites found so far, before this one joins)
        filter.Y = hcat(filter.Y, filter.shifted_seeds[i])
        _append!(g, filter.shifted_seeds[i])
        filter.Z = _complement(g, filter.p)
        setdiff!(filter.candidate_indices, i)
        S[:, i] .= 0
        return (filter.return_indices ? i : filter.seeds[i]), num_found + 1
    end
    return done()
end
"`‖Z (Zᵀ s)‖_p` of every column s of S on the host; first maximiser like the `>` scan of AffinelyIndependentPoints.jl:80-89"
function _affine_scores_host(S::Matrix{Float64}, Z::AbstractMatrix, p)
    size(Z, 2) == 0 && return 0, -Inf
    P = Z * (Z' * S)
    best_val, best = findmax([norm(view(P, :, c), p) for c in axes(P, 2)])
    return best, best_val
end

# ---- phase II: the fit.  With a kept round-4 factor that describes exactly this training set: two triangular solves
#      (mrbf_fit_from_round4, the reference's TODO at RbfModel.jl:657-660); otherwise Gram + factorisation + solve (mrbf_fit)
function update_model(mod::Union{Nothing,HipRbfModel}, meta::RbfMeta, cfg::HipRbfConfig,
                      func_indices, mop, scal, iter_data, sdb, ac; kwargs...)
    db = get_sub_db(sdb, func_indices)
    Δ = get_delta(iter_data)
    training_indices = _collect_indices(meta)                                  # RbfModel.jl:754-757
    training_results = get_result.(db, training_indices)
    sites = get_site.(training_results); values = get_value.(training_results)
    kept = lock(_ROUND4_LOCK) do
        pop!(_ROUND4_KEPT, db, nothing)
    end
    if kept !== nothing
        state, ids = kept
        n0, nacc, q = _round4_dims(state, cfg, length(first(sites)))
        if _dispatch_fit(length(training_indices), n0, q, nacc, ids == training_indices)
            rc, model = fit_from_round4(state, values, length(first(sites)), meta.fully_linear; rc_only = true)
            _free_round4!(state)                 # released here, not left to the finalizer (as sampling.py does)
            rc == 0 && return model, meta
            _fallback_rc(2, rc) || _check(state.ctx, rc)
        else
            _free_round4!(state)
        end
    end
    C = _dense(_as_matrix(sites))
    Y = _dense(_as_matrix(values))
    d, n = size(C)
    k = size(Y, 1)
    kid, a, b = _mrbf_kernel_params(Δ, cfg)
    ctx = mrbf_context()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    info = Ref{MrbfFitInfo}()
    rc = GC.@preserve C Y begin
        _locked(ctx) do hctx
            ccall((:mrbf_fit, libmrbf), Int32,
                  (Ptr{Cvoid}, Int64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64, Int32,
                   Ref{Ptr{Cvoid}}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfFitInfo}),
                  hctx, n, d, k, C, Y, kid, a, b, cfg.polynomial_degree, h, C_NULL, C_NULL, info)
        end
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
    rc = GC.@preserve X V J begin
        _locked(mod.ctx) do hctx
            ccall((:mrbf_eval, libmrbf), Int32,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                  hctx, mod.handle, m, X, values ? pointer(V) : C_NULL, jac ? pointer(J) : C_NULL, C_NULL)
        end
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
# (SurrogateContainer.jl:234-269 does one inner-model call per function index; here every distinct grouped HipRbfModel is swept
#  ONCE for all sites, RefSurrogates select rows, CompositeSurrogates apply their outer function / chain rule per site,
#  AbstractSurrogateInterface.jl:136-154, :175-229.)  Sites are the columns of X_scaled (d x m).
_inner(s::RefSurrogate) = s.model_ref[]
_inner(s::CompositeSurrogate) = s.model_ref[]
_inner(s) = s
# the container keeps dictionaries function index => surrogate (SurrogateContainer.jl:111-113); order = key order, as in :248-267
_container_surrogates(sc::SurrogateContainer, ::Val{:objectives}) = [get_surrogates(sc, ind) for ind in get_objective_indices(sc)]
_container_surrogates(sc::SurrogateContainer, ::Val{:nl_eq_constraints}) = [get_surrogates(sc, ind) for ind in get_nl_eq_constraint_indices(sc)]
_container_surrogates(sc::SurrogateContainer, ::Val{:nl_ineq_constraints}) = [get_surrogates(sc, ind) for ind in get_nl_ineq_constraint_indices(sc)]
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
            reduce(hcat, [_eval_models_vec(s, scal, X[:, p]) for p = 1:m])
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
for plural in (:objectives, :nl_eq_constraints, :nl_ineq_constraints)
    vals = Symbol("eval_container_", plural, "_at_scaled_sites")
    jacs = Symbol("eval_container_", plural, "_jacobian_at_scaled_sites")
    @eval begin
        $vals(sc::SurrogateContainer, scal, X::AbstractMatrix) = _container_values_at_sites(_container_surrogates(sc, Val($(QuoteNode(plural)))), scal, X)
        $jacs(sc::SurrogateContainer, scal, X::AbstractMatrix) = _container_jacobians_at_sites(_container_surrogates(sc, Val($(QuoteNode(plural)))), scal, X)
    end
end

"Does the container hold a `HipRbfModel` at all?  Pure Julia: asked before anything of libmrbf is touched (inertness for non-users)."
function _touches_device(sc::SurrogateContainer; objectives_only::Bool = false)
    kinds = objectives_only ? (Val(:objectives),) : (Val(:objectives), Val(:nl_eq_constraints), Val(:nl_ineq_constraints))
    for kind in kinds
        for s in _container_surrogates(sc, kind)
            _inner(s) isa HipRbfModel && return true
        end
    end
    return false
end

# What the device entry points need to know about a container: its distinct grouped HipRbfModels, the role of every output row
# (objective position l >= 0, MRBF_ROLE_EQ = -2, MRBF_ROLE_INEQ = -3, MRBF_ROLE_NONE = -1) and how many surrogates are "foreign"
# (CompositeSurrogates, other model families, or a model row used twice): with a foreign one the reference methods run.
function _container_plan(sc::SurrogateContainer; objectives_only::Bool = false)
    models = HipRbfModel[]; roles = Vector{Vector{Int32}}()
    k = 0; n_con = 0; n_foreign = 0
    function slot(m)
        i = findfirst(x -> x === m, models)
        i === nothing || return i
        push!(models, m); push!(roles, fill(Int32(-1), num_outputs(m)))
        return length(models)
    end
    for (kind, role) in ((Val(:objectives), 0), (Val(:nl_eq_constraints), -2), (Val(:nl_ineq_constraints), -3))
        objectives_only && role != 0 && continue                              # _backtrack only evaluates the objectives (descent.jl:161-179)
        for s in _container_surrogates(sc, kind)
            # a bare HipRbfModel in a container list contributes all its rows in order (as surrogates.py container_plan does)
            inner = s isa RefSurrogate ? _inner(s) : s
            if !(s isa CompositeSurrogate) && inner isa HipRbfModel
                i = slot(inner)
                for oi in (s isa RefSurrogate ? s.output_indices : 1:num_outputs(inner))
                    roles[i][oi] == -1 || (n_foreign += 1)                    # one row in two roles: not expressible
                    roles[i][oi] = role == 0 ? Int32(k) : Int32(role)
                    role == 0 ? (k += 1) : (n_con += 1)
                end
            else
                n_foreign += 1
                role == 0 ? (k += num_outputs(s)) : (n_con += num_outputs(s))
            end
        end
    end
    in_order = length(models) == 1 && n_con == 0 && roles[1] == Int32.(0:num_outputs(models[1])-1)
    return (; models, roles = isempty(roles) ? Int32[] : reduce(vcat, roles), k, n_con, n_foreign, in_order)
end

# ---- descent consumers ------------------------------------------------------------------------------------------------------------
"All Armijo step sizes of `_backtrack` (descent.jl:150-185) in one batch; returns (x₊, mx₊, step) like the reference."
function _backtrack(x::AbstractVector{F}, dir, step_size, ω, sc::SurrogateContainer, cfg, scal) where {F<:AbstractFloat}
    # no HipRbfModel among the objectives (every run that does not use HipRbfConfig): Morbit's own loop, libmrbf is not touched
    _touches_device(sc; objectives_only = true) ||
        return invoke(_backtrack, Tuple{AbstractVector{F},Any,Any,Any,Any,Any,Any}, x, dir, step_size, ω, sc, cfg, scal)
    plan = _container_plan(sc; objectives_only = true)
    if !_dispatch_backtrack(length(plan.models), plan.n_foreign, plan.in_order)
        return invoke(_backtrack, Tuple{AbstractVector{F},Any,Any,Any,Any,Any,Any}, x, dir, step_size, ω, sc, cfg, scal)
    end
    model = plan.models[1]
    x64 = Vector{Float64}(x); dir64 = Vector{Float64}(dir)
    k = model.num_outputs
    x₊, mx₊, step, loops = similar(x64), Vector{Float64}(undef, k), similar(x64), Ref{Int32}(0)
    rc = GC.@preserve x64 dir64 x₊ mx₊ step begin
        _locked(model.ctx) do hctx
            ccall((:mrbf_backtrack, libmrbf), Int32,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int32, Float64, Float64, Float64, Int32,
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                  hctx, model.handle, x64, dir64, step_size, ω, cfg.strict_backtracking, cfg.armijo_const_rhs,
                  cfg.armijo_const_shrink, cfg.min_stepsize >= 0 ? cfg.min_stepsize : eps(F), cfg.max_loops,
                  x₊, mx₊, step, loops)
        end
    end
    _check(model.ctx, rc)
    return F.(x₊), F.(mx₊), F.(step)
end

"""
Pascoletti-Serafini descent step (descent.jl:512-581) with the subproblem solver on the device (`mrbf_ps_step_problem`): objectives
and modelled constraints may be spread over several grouped `HipRbfModel`s, the MOP may carry linear constraints.  Whenever the
decision table says so (a `CompositeSurrogate` or another model family in the container, sizes outside the device path) or the
device call asks for it, Morbit's own method runs on the same arguments -- this method never raises because of a size limit.
Same returns as the reference.
"""
function get_criticality(desc_cfg::PascolettiSerafiniConfig, mop, scal, x_it, x_it_n, data_base, sc::SurrogateContainer, algo_config;
                         seed::UInt64 = rand(UInt64))
    reference() = invoke(get_criticality, Tuple{PascolettiSerafiniConfig,Any,Any,Any,Any,Any,Any,Any}, desc_cfg, mop, scal, x_it, x_it_n, data_base, sc, algo_config)
    _touches_device(sc) || return reference()      # no HipRbfModel in the container: Morbit's own method, libmrbf is not touched
    plan = _container_plan(sc)
    x = Vector{Float64}(get_x_scaled(x_it)); x_n = Vector{Float64}(get_x_scaled(x_it_n)); fx_n = Vector{Float64}(get_fx(x_it_n))
    d, k = length(x_n), plan.k
    A_eq, b_eq = transformed_linear_eq_constraints(scal, mop)                  # AbstractMOPInterface.jl:463-481: A x_scaled (=, <=) b
    A_in, b_in = transformed_linear_ineq_constraints(scal, mop)
    _dispatch_ps(d, k, length(plan.models), plan.n_con, length(b_eq) + length(b_in), plan.n_foreign) || return reference()
    lb_eff, ub_eff = local_bounds(scal, x, get_delta(x_it))
    lb = Vector{Float64}(lb_eff); ub = Vector{Float64}(ub_eff)
    r = _get_global_dir(desc_cfg, fx_n)                                        # descent.jl:360-368
    rbuf = r === nothing ? nothing : Vector{Float64}(r)
    g_evals, l_evals = _ps_max_evals(desc_cfg, d)                              # descent.jl:414-432
    opts = Ref(MrbfPsOptions(desc_cfg.max_ideal_point_problem_evals, g_evals, l_evals, 0, seed, -0.5, 1e-3))
    info = Ref{MrbfPsInfo}()
    x_trial, mx_trial = Vector{Float64}(undef, d), Vector{Float64}(undef, k)
    handles = Ptr{Cvoid}[m.handle for m in plan.models]
    roles = plan.roles
    Aeq = Matrix{Float64}(transpose(Matrix(A_eq))); beq = Vector{Float64}(b_eq)   # row-major rows x d == the d x rows column-major matrix
    Ain = Matrix{Float64}(transpose(Matrix(A_in))); bin = Vector{Float64}(b_in)
    ctx = plan.models[1].ctx
    rc = GC.@preserve handles roles Aeq beq Ain bin x_n lb ub fx_n rbuf x_trial mx_trial begin
        prob = Ref(MrbfPsProblem(length(handles), k, pointer(handles), pointer(roles), length(beq), length(bin),
                                 isempty(beq) ? C_NULL : pointer(Aeq), isempty(beq) ? C_NULL : pointer(beq),
                                 isempty(bin) ? C_NULL : pointer(Ain), isempty(bin) ? C_NULL : pointer(bin), -1.0))
        _locked(ctx) do hctx
            ccall((:mrbf_ps_step_problem, libmrbf), Int32,
                  (Ptr{Cvoid}, Ref{MrbfPsProblem}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfPsOptions},
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfPsInfo}),
                  hctx, prob, x_n, lb, ub, fx_n, rbuf === nothing ? C_NULL : pointer(rbuf), opts, x_trial, mx_trial, C_NULL, info)
        end
    end
    rc != 0 && _fallback_rc(3, rc) && return reference()
    _check(ctx, rc)
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
function _free_round4!(s::HipRound4State)
    ctx = s.ctx
    if !trylock(ctx.lock)                 # also a finalizer: never block, never release under a running call of the same context
        @async _free_round4!(s)
        return nothing
    end
    try
        if s.handle != C_NULL && ctx.handle != C_NULL
            ccall((:mrbf_free_round4, libmrbf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.handle, s.handle)
        end
        s.handle = C_NULL
    finally
        unlock(ctx.lock)
    end
    return nothing
end
"(n0, accepted sites, q) of a kept round-4 factor"
function _round4_dims(state::HipRound4State, cfg, n_vars)
    n0 = Ref{Int64}(0); nc = Ref{Int64}(0); nacc = Ref{Int32}(0)
    ccall((:mrbf_round4_sites, libmrbf), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int32}), state.handle, n0, nc, nacc)
    q = cfg.polynomial_degree < 0 ? 0 : (cfg.polynomial_degree == 0 ? 1 : n_vars + 1)
    return n0[], Int(nacc[]), q
end
"""
`_rbf_round4` (RbfModel.jl:352-499) as one device call.  `start_sites` / `cand_sites`: vectors of sites (the sites found so far; the
box candidates in database order, followed -- with `use_max_points` -- by the random box points the caller drew).  Returns the
positions of the accepted candidates in acceptance order and, with `keep_state`, the factors for `fit_from_round4` (`nothing` when
there was nothing to select: the library then keeps no state).  With `rc_only` the return code comes first and nothing is raised.
"""
function rbf_round4_device(cfg::HipRbfConfig, Δ, start_sites, cand_sites; keep_state::Bool = false, rc_only::Bool = false)
    C0 = _dense(_as_matrix(start_sites)); Xc = isempty(cand_sites) ? Matrix{Float64}(undef, size(C0, 1), 0) : _dense(_as_matrix(cand_sites))
    d, n0 = size(C0); mc = size(Xc, 2)
    kid, a, b = _mrbf_kernel_params(Δ, cfg)
    ctx = mrbf_context()
    acc = Vector{Int32}(undef, max(mc, 1)); nacc = Ref{Int32}(0); st = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve C0 Xc acc begin
        _locked(ctx) do hctx
            ccall((:mrbf_round4, libmrbf), Int32,
                  (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int32, Float64, Float64, Int32, Int32, Float64,
                   Ptr{Int32}, Ref{Int32}, Ptr{Ptr{Cvoid}}),
                  hctx, n0, d, C0, mc, Xc, kid, a, b, cfg.polynomial_degree, cfg.max_model_points, cfg.θ_pivot_cholesky,
                  acc, nacc, keep_state ? Base.unsafe_convert(Ptr{Ptr{Cvoid}}, st) : C_NULL)
        end
    end
    if rc != 0
        rc_only && return rc, Int[], nothing
        _check(ctx, rc)
    end
    accepted = Int.(acc[1:nacc[]]) .+ 1                                         # 1-based positions into cand_sites
    state = nothing
    if keep_state && st[] != C_NULL                                             # NULL: nothing to select, no state (mrbf.h)
        state = HipRound4State(ctx, st[])
        finalizer(_free_round4!, state)
    end
    rc_only && return rc, accepted, state
    return keep_state ? (accepted, state) : accepted
end
"The model on (start sites, accepted sites) from the factor round 4 kept (RbfModel.jl:657-660): `values` in that order, k x n."
function fit_from_round4(state::HipRound4State, values, n_vars::Int, fully_linear::Bool; rc_only::Bool = false)
    Y = _dense(_as_matrix(values)); k = size(Y, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL); info = Ref{MrbfFitInfo}()
    rc = GC.@preserve Y begin
        _locked(state.ctx) do hctx
            ccall((:mrbf_fit_from_round4, libmrbf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Float64}, Ref{Ptr{Cvoid}}, Ptr{Float64}, Ptr{Float64}, Ref{MrbfFitInfo}),
                  hctx, state.handle, k, Y, h, C_NULL, C_NULL, info)
        end
    end
    if rc != 0
        rc_only && return rc, nothing
        _check(state.ctx, rc)
    end
    model = HipRbfModel(state.ctx, h[], n_vars, k, fully_linear, info[])
    return rc_only ? (rc, model) : model
end
"""
The filter's whole pick loop in one device call (`mrbf_affine_select`): `S` d x mc shifted seeds (picked columns zero), `Q0` the full
orthogonal factor of the directions chosen so far (first `j0` columns), up to `want` further picks above `pivot_val`.  Returns the
1-based positions in pick order and the final p-normalised complement basis.
"""
function affine_select(S::Matrix{Float64}, Q0::Matrix{Float64}, j0::Int, want::Int, pivot_val::Float64, p)
    d, mc = size(S)
    ctx = mrbf_context()
    picks = Vector{Int64}(undef, max(want, 1)); npick = Ref{Int32}(0)
    Zbuf = Matrix{Float64}(undef, d, max(d - j0, 1))
    rc = GC.@preserve S Q0 picks Zbuf begin
        _locked(ctx) do hctx
            ccall((:mrbf_affine_select, libmrbf), Int32,
                  (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Float64, Int32, Ptr{Int64}, Ref{Int32}, Ptr{Float64}),
                  hctx, mc, d, S, j0, Q0, want, pivot_val, isinf(p) ? 1 : 0, picks, npick, Zbuf)
        end
    end
    _check(ctx, rc)
    np = Int(npick[])
    return Int.(picks[1:np]) .+ 1, Zbuf[:, 1:(d - j0 - np)]
end
"Scores `‖Z (Zᵀ(ξ - x₀))‖_p` of all filter candidates and the first maximiser (AffinelyIndependentPoints.jl:71-106)."
affine_scores(shifted_seeds::AbstractVector, Z::AbstractMatrix, p = Inf) = affine_scores(_dense(_as_matrix(shifted_seeds)), Z, p)
function affine_scores(S::AbstractMatrix{Float64}, Z::AbstractMatrix, p = Inf)     # S: d x mc, one shifted seed per column
    d, mc = size(S); Zm = Matrix{Float64}(Z)
    ctx = mrbf_context(); best = Ref{Int64}(-1); val = Ref{Float64}(-Inf)
    rc = GC.@preserve S Zm begin
        _locked(ctx) do hctx
            ccall((:mrbf_affine_scores, libmrbf), Int32, (Ptr{Cvoid}, Int64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ref{Int64}, Ref{Float64}),
                  hctx, mc, d, size(Zm, 2), S, Zm, isinf(p) ? 1 : 0, C_NULL, best, val)
        end
    end
    _check(ctx, rc)
    return Int(best[]) + 1, val[]
end
