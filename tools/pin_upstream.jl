# pin_upstream.jl -- run ONCE by a Morbit maintainer (Julia with Morbit.jl and RadialBasisFunctionModels 0.3.4, the compat pin of
# /root/reference/Project.toml:25,50) to pin this repository's CPU oracle to the upstream arithmetic:
#
#     julia --project=<Morbit checkout> tools/pin_upstream.jl            # writes tests/golden/upstream_pins.json
#
# The build container of this repository has no Julia and the package source is not in the reference tree, so oracle/rbf_oracle.py
# restates the published radial functions and DESIGN.md section 4 lists four guesses nothing here can check.  This script emits the few
# numbers that settle them; tests/test_upstream_pins.py compares the oracle with the file when it exists (and reports "parity
# unpinned" when it does not).  Inputs are closed formulas (Weyl sequences) that the Python test regenerates bit for bit -- no file
# has to travel in the other direction.  Every pin is wrapped in try / catch: whatever could be produced is written.
#
#   guess 1  order of the polynomial tail basis            <- Pi of RBF.get_matrices (the call Morbit makes, RbfModel.jl:374-375)
#   guess 2  sign / scale of every radial function          <- Phi of the same call, all five kernels of Morbit.RbfKernels
#   guess 3  polynomial_degree below the kernel's order     <- RBF.RBFInterpolationModel(sites, values, :cubic, nothing, -1): error or values
#   guess 4  orientation of Pi in _rbf_round4               <- size of the returned matrices + the index list of one _rbf_round4 call
using Morbit, Printf, LinearAlgebra
const RBF = Morbit.RBF

# ---- inputs: C[i, t] = frac((i + 1) sqrt(p_t)), the same in tests/test_upstream_pins.py
const PRIMES = [2.0, 3.0, 5.0, 7.0, 11.0]
weyl(n, d) = [mod((i) * sqrt(PRIMES[t]), 1.0) for i = 1:n, t = 1:d]          # n x d, rows = sites
values_of(C) = hcat(vec(sum((C .- 0.3) .^ 2, dims = 2)), vec(sum((C .+ 0.5) .^ 2, dims = 2)))   # n x 2

jnum(x::Real) = isfinite(x) ? @sprintf("%.17g", x) : "null"
jarr(v::AbstractVector) = "[" * join(jnum.(v), ", ") * "]"
jmat(M::AbstractMatrix) = "[" * join([jarr(M[i, :]) for i = 1:size(M, 1)], ", ") * "]"
jstr(s) = "\"" * replace(string(s), "\\" => "\\\\", "\"" => "'", "\n" => " ") * "\""

out = String[]
pin(key, val) = push!(out, "  " * jstr(key) * ": " * val)

n, d = 12, 3
C = weyl(n, d)
Y = values_of(C)
sites = [C[i, :] for i = 1:n]
vals = [Y[i, :] for i = 1:n]
probe = weyl(5, d) .* 0.9 .+ 0.05
pin("inputs", "{\"n\": $n, \"d\": $d, \"formula\": \"C[i,t] = frac(i sqrt(p_t)), p = 2,3,5; Y = [sum((C-0.3)^2), sum((C+0.5)^2)]; probe = 0.9 weyl(5,3) + 0.05\"}")

# ---- guesses 1, 2, 4: the matrices Morbit itself asks for
for kernel in Morbit.RbfKernels
    try
        cfg = RbfConfig(; kernel, polynomial_degree = 1)
        φ = Morbit._get_radial_function(1.0, cfg)
        Φᵀ, Πᵀ, kernels, polys = RBF.get_matrices(φ, sites; poly_deg = 1)
        pin("get_matrices_$(kernel)_Phi_size", jarr(collect(Float64, size(Φᵀ))))
        pin("get_matrices_$(kernel)_Pi_size", jarr(collect(Float64, size(Πᵀ))))
        pin("get_matrices_$(kernel)_Phi_first_rows", jmat(Matrix{Float64}(Φᵀ)[1:3, :]))
        pin("get_matrices_$(kernel)_Pi", jmat(Matrix{Float64}(Πᵀ)))
        pin("kernels_of_probe_$(kernel)", jarr(Vector{Float64}(kernels(probe[1, :]))))      # kernels(xi), RbfModel.jl:421
        pin("polys_of_probe_$(kernel)", jarr(Vector{Float64}(polys(probe[1, :]))))          # polys(xi), RbfModel.jl:424
    catch e
        pin("get_matrices_$(kernel)_error", jstr(e))
    end
end

# ---- the model itself: values / Jacobians at the probe points and whatever numeric fields it holds (weights, tail coefficients)
function dump_model(tag, kernel, params, deg)
    try
        model = RBF.RBFInterpolationModel(sites, vals, kernel, params, deg; save_matrices = false)
        pin("$(tag)_values", jmat(reduce(hcat, [Vector{Float64}(model(probe[i, :])) for i = 1:size(probe, 1)])'))
        pin("$(tag)_jac_first_probe", jmat(Matrix{Float64}(RBF.jac(model, probe[1, :]))))
        pin("$(tag)_grad_output2_first_probe", jarr(Vector{Float64}(RBF.grad(model, probe[1, :], 2))))
        for fn in fieldnames(typeof(model))
            v = getfield(model, fn)
            if v isa AbstractMatrix{<:Real}
                pin("$(tag)_field_$(fn)", jmat(Matrix{Float64}(v)))
            elseif v isa AbstractVector{<:Real}
                pin("$(tag)_field_$(fn)", jarr(Vector{Float64}(v)))
            elseif v isa AbstractVector && !isempty(v) && first(v) isa AbstractVector{<:Real}
                pin("$(tag)_field_$(fn)", jmat(reduce(hcat, [Vector{Float64}(x) for x in v])'))
            end
        end
        pin("$(tag)_fieldnames", jstr(join(string.(fieldnames(typeof(model))), ",")))
    catch e
        pin("$(tag)_error", jstr(e))
    end
end
dump_model("cubic_deg1", :cubic, nothing, 1)                 # guess 1 (tail order through the coefficient fields), weights sign
dump_model("tps_deg1", :thin_plate_spline, nothing, 1)       # guess 2 for the thin plate spline
dump_model("multiquadric_deg1", :multiquadric, nothing, 1)
dump_model("gaussian_deg0", :gaussian, nothing, 0)
dump_model("cubic_degm1", :cubic, nothing, -1)               # guess 3: does upstream refuse, raise the degree, or solve the indefinite system?

# ---- guess 4: one _rbf_round4 index list (set-up as in test/rbf_models.jl:6-24, :73-86, sites from the formula instead of rand)
try
    cfg = RbfConfig(; kernel = :cubic, polynomial_degree = 1)
    mop = MOP(d)
    f1w = Morbit.make_vec_fun(x -> sum(x .^ 2); n_out = 1, model_cfg = cfg)
    nl_ind = Morbit._add_function!(mop, f1w)
    Morbit._add_objective!(mop, nl_ind)
    x0 = fill(0.5, d)
    smop, id, sdb, sc, ac, filter, scal = Morbit.initialize_data(mop, x0; algo_config = AlgorithmConfig(; max_evals = 1))
    db = Morbit.get_sub_db(sdb, (nl_ind,))
    x = Morbit.get_x_scaled(id)
    Δ = Morbit.get_delta(id)
    lb, ub = Morbit.local_bounds(scal, x, cfg.θ_enlarge_2 * Morbit.delta_max(ac))
    start = [Morbit.get_x_index(id, (nl_ind,))]
    S = weyl(40, d)
    ids = Int[]
    for i = 1:size(S, 1)
        push!(ids, Morbit.new_result!(db, lb .+ (ub .- lb) .* S[i, :]))
    end
    # a unisolvent start set: the centre and the first d sites; the rest are round-4 candidates in database order
    start_ids = [start; ids[1:d]]
    r4 = Morbit._rbf_round4(db, lb, ub, x, Δ, start_ids, cfg)
    pin("round4_x", jarr(Vector{Float64}(x)))
    pin("round4_lb", jarr(Vector{Float64}(lb)))
    pin("round4_ub", jarr(Vector{Float64}(ub)))
    pin("round4_delta", jnum(Float64(Δ)))
    pin("round4_start_positions", jarr(Float64.([0; 1:d])))                  # 0 = the centre, i = i-th Weyl site
    pin("round4_accepted_positions", jarr(Float64.([findfirst(==(i), ids) for i in r4])))
catch e
    pin("round4_error", jstr(e))
end

pin("versions", jstr("Julia $(VERSION)"))
path = joinpath(@__DIR__, "..", "tests", "golden", "upstream_pins.json")
open(path, "w") do io
    write(io, "{\n" * join(out, ",\n") * "\n}\n")
end
println("wrote ", path, " with ", length(out), " pins")
