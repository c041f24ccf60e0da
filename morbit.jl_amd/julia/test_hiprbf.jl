# Maintainer-side tests of the HipRbf plug-in (morbit.jl_amd/julia/HipRbf.jl) -- the property tests Morbit holds for RbfConfig
# (/root/reference/test/rbf_models.jl:99-115 values / gradients, :121-168 site rounds and container Jacobians) with HipRbfConfig in
# its place, plus one optimize() run with the Pascoletti-Serafini descent step.
#
# NOT executed in the build container (no Julia there, SURVEY.md section 8c).  To run it: a Morbit checkout with
# `include("models/HipRbf.jl"); export HipRbfConfig` added to src/Morbit.jl (INTEGRATION.md), a built libmrbf.so, one MI355X:
#
#     MRBF_LIB=/path/to/morbit.jl_amd/libmrbf.so julia --project=. /path/to/morbit.jl_amd/julia/test_hiprbf.jl
#
# Tolerances are those of the north star: 1e-8 relative on surrogate values, gradients against ForwardDiff (Morbit.AD) as in the
# reference's own test.
using Morbit
using Test

const f1 = x -> sum(x .^ 2)

function _initialize(model_cfg, algo_config = nothing; num_vars = 3, constrained = false)
    mop = constrained ? MOP(fill(0.25, num_vars), fill(0.75, num_vars)) : MOP(num_vars)
    f1_wrapped = Morbit.make_vec_fun(f1; n_out = 1, model_cfg)
    nl_ind = Morbit._add_function!(mop, f1_wrapped)
    objf_ind = Morbit._add_objective!(mop, nl_ind)
    x0 = rand(num_vars)
    smop, id, sdb, sc, ac, filter, scal = Morbit.initialize_data(mop, x0; algo_config)
    return nl_ind, objf_ind, smop, id, sdb, sc, ac, filter, scal
end

@testset "HipRbfConfig mirrors RbfConfig" begin
    for kernel in Morbit.RbfKernels, deg in -1:1
        a = HipRbfConfig(; kernel, polynomial_degree = deg)
        b = HipRbfConfig(; kernel, polynomial_degree = deg)
        @test isequal(a, b) && hash(a) == hash(b)
        @test Morbit.combinable(a)
        @test Morbit.max_evals(a) == Morbit.max_evals(RbfConfig(; kernel, polynomial_degree = deg))
    end
    @test_throws AssertionError HipRbfConfig(; kernel = :no_such_kernel)
    @test_throws AssertionError HipRbfConfig(; polynomial_degree = 2)
end

# test/rbf_models.jl:24-115 with the device plug-in: budgets respected, models from too few points, full linearity after an
# update with many database points, interpolation at the iterate, gradient == Jacobian row, gradient ~ ForwardDiff
@testset "values and derivatives (test/rbf_models.jl:99-115)" begin
    for num_vars in [2, 5, 10], kernel in Morbit.RbfKernels, polynomial_degree in -1:1, constrained in [true, false]
        model_cfg = HipRbfConfig(; kernel, polynomial_degree, max_evals = 1, max_model_points = 1)
        nl_ind, objf_ind, smop, id, sdb, sc, ac, filter, scal = _initialize(model_cfg; num_vars, constrained)
        @test Morbit.num_evals(Morbit._get(smop, nl_ind)) == 1
        @test Morbit.num_evals(Morbit._get(smop, objf_ind)) == 1

        if polynomial_degree == 1
            Δ = Morbit.get_delta(id)
            x = Morbit.get_x_scaled(id)
            db = Morbit.get_sub_db(sdb, (nl_ind,))
            lb, ub = Morbit.local_bounds(scal, x, Δ)
            w = ub .- lb
            for i = 1:50*num_vars
                Morbit.new_result!(db, lb .+ w .* rand(num_vars))
            end
            Morbit.update_surrogates!(sc, smop, scal, id, sdb, ac)
            @test Morbit.fully_linear(sc)
        end

        model_cfg = HipRbfConfig(; kernel, polynomial_degree)
        algo_config = AlgorithmConfig(; max_evals = 1)
        nl_ind, objf_ind, smop, id, sdb, sc, ac, filter, scal = _initialize(model_cfg, algo_config; num_vars, constrained)
        @test Morbit.num_evals(Morbit._get(smop, nl_ind)) == 1

        # round 4 runs with fewer than num_vars + 1 points (device call or Morbit's own loop, by the decision table)
        db = Morbit.get_sub_db(sdb, (nl_ind,))
        Δ = Morbit.get_delta(id)
        θ = model_cfg.θ_enlarge_2
        x = Morbit.get_x_scaled(id)
        indices = [Morbit.get_x_index(id, (nl_ind,))]
        lb, ub = Morbit.local_bounds(scal, x, θ * Morbit.delta_max(ac))
        w = ub .- lb
        for i = 1:10*num_vars
            Morbit.new_result!(db, lb .+ w .* rand(num_vars))
        end
        r4 = Morbit._rbf_round4(db, lb, ub, x, Δ, indices, model_cfg)
        # the same call through Morbit's own method picks the same sites in the same order
        @test r4 == Morbit._rbf_round4(db, lb, ub, x, Δ, indices, Morbit._as_rbf_config(model_cfg))

        x = Morbit.get_x_scaled(id)
        x_unscaled = Morbit.get_x(id)
        mod = Morbit.get_surrogates(sc, nl_ind)
        dm = Morbit.get_gradient(mod, scal, x, 1)
        @test Morbit.eval_models(mod, scal, x)[end] ≈ f1(x_unscaled)
        @test dm == vec(Morbit.eval_container_jacobian_at_func_index_at_scaled_site(sc, scal, x, nl_ind))
        @test dm ≈ Morbit.AD.gradient(ξ -> Morbit.eval_models(mod, scal, ξ)[end], x)
        # the batched twin returns what the one-site calls return
        X = hcat(x, x .+ 1e-3)
        V = Morbit.eval_container_objectives_at_scaled_sites(sc, scal, X)
        @test V[:, 1] ≈ Morbit.eval_container_objectives_at_scaled_site(sc, scal, x) rtol = 1e-12
    end
end

# test/rbf_models.jl:121-168: rounds 1-3 do not depend on the kernel, container Jacobian ~ AD Jacobian
@testset "site rounds and container Jacobian (test/rbf_models.jl:121-168)" begin
    mop = MOP(2)
    objf_ind_1 = add_objective!(mop, f1; n_out = 1, model_cfg = HipRbfConfig(; kernel = :gaussian))
    objf_ind_2 = add_objective!(mop, x -> sum(abs.(x)); n_out = 1, model_cfg = HipRbfConfig(; kernel = :multiquadric))
    x0 = rand(2)
    smop, id, sdb, sc, ac, filter, scal = Morbit.initialize_data(mop, x0; algo_config = AlgorithmConfig(; max_evals = 1))
    nl_ind_1 = Morbit.get_surrogates(sc, objf_ind_1).nl_index
    nl_ind_2 = Morbit.get_surrogates(sc, objf_ind_2).nl_index
    db_1 = Morbit.get_sub_db(sdb, (nl_ind_1,))
    db_2 = Morbit.get_sub_db(sdb, (nl_ind_2,))
    for i = 1:20
        ξ = rand(2)
        Morbit.new_result!(db_1, ξ)
        Morbit.new_result!(db_2, ξ)
    end
    Morbit.update_surrogates!(sc, smop, scal, id, sdb, ac; ensure_fully_linear = true)
    meta_1 = Morbit.get_meta(sc.surrogates[1])
    meta_2 = Morbit.get_meta(sc.surrogates[2])
    for fn in [:round1_indices, :round2_indices, :round3_indices]
        ind_1 = getfield(meta_1, fn)
        ind_2 = getfield(meta_2, fn)
        @test all(Morbit.get_site(db_1, i1) == Morbit.get_site(db_2, i2) for (i1, i2) in zip(ind_1, ind_2))
    end
    x = Morbit.get_x_scaled(id)
    @test Morbit.eval_container_objectives_jacobian_at_scaled_site(sc, scal, x) ≈
          Morbit.AD.jacobian(ξ -> Morbit.eval_container_objectives_at_scaled_site(sc, scal, ξ), x)
end

# the same numbers as Morbit's own RbfConfig models on one fixed training set (1e-8 on values, the north star's tolerance)
@testset "HipRbfModel vs RbfModel on the same sites" begin
    for kernel in Morbit.RbfKernels
        mop_a = MOP(3); mop_b = MOP(3)
        add_objective!(mop_a, f1; n_out = 1, model_cfg = RbfConfig(; kernel))
        add_objective!(mop_b, f1; n_out = 1, model_cfg = HipRbfConfig(; kernel))
        x0 = [0.3, 0.6, 0.1]
        Morbit.Random.seed!(1234)       # test/runtests.jl:4
        a = Morbit.initialize_data(mop_a, x0; algo_config = AlgorithmConfig(; max_evals = 30))
        Morbit.Random.seed!(1234)
        b = Morbit.initialize_data(mop_b, x0; algo_config = AlgorithmConfig(; max_evals = 30))
        xa = Morbit.get_x_scaled(a[2])
        va = Morbit.eval_container_objectives_at_scaled_site(a[4], a[7], xa)
        vb = Morbit.eval_container_objectives_at_scaled_site(b[4], b[7], xa)
        @test va ≈ vb rtol = 1e-8
        Ja = Morbit.eval_container_objectives_jacobian_at_scaled_site(a[4], a[7], xa)
        Jb = Morbit.eval_container_objectives_jacobian_at_scaled_site(b[4], b[7], xa)
        @test Ja ≈ Jb rtol = 1e-6
    end
end

# examples/example_two_parabolas.jl with the device models, steepest descent and the Pascoletti-Serafini step (descent.jl:512-581,
# routed to mrbf_ps_step_problem by get_criticality(::PascolettiSerafiniConfig, ...) of the plug-in)
@testset "optimize() on two parabolas" begin
    g1 = x -> sum((x .- 1) .^ 2)
    g2 = x -> sum((x .+ 1) .^ 2)
    for descent_method in [:steepest_descent, :ps]
        mop = MOP(2)
        add_objective!(mop, g1; n_out = 1, model_cfg = HipRbfConfig(; kernel = :multiquadric))
        add_objective!(mop, g2; n_out = 1, model_cfg = HipRbfConfig(; kernel = :multiquadric))
        x, fx, ret_code, _ = optimize(mop, [-π, 2.71828]; algo_config = AlgorithmConfig(; descent_method, max_iter = 20))
        @test x[1] ≈ x[2] atol = 0.1       # the Pareto set is the diagonal between the two minima
        @test all(isfinite, fx)
    end
end

# Round 6: the plug-in is inert for users who do not select HipRbfConfig.  A plain RbfConfig run -- model update with many database
# sites (the affine filter iterates), steepest descent (`_backtrack`) and the Pascoletti-Serafini step (`get_criticality`) -- makes no
# ccall: it gives Morbit's own result and creates no context.  (Run this testset FIRST in a fresh session to see that it also works
# with MRBF_LIB pointing nowhere: the three methods the plug-in adds on Morbit's own types hand over before libmrbf is touched.)
@testset "inert for plain RbfConfig runs" begin
    n_ctx = length(Morbit._CTX)
    g1 = x -> sum((x .- 1) .^ 2)
    g2 = x -> sum((x .+ 1) .^ 2)
    for descent_method in [:steepest_descent, :ps]
        mop = MOP(2)
        add_objective!(mop, g1; n_out = 1, model_cfg = RbfConfig(; kernel = :multiquadric))
        add_objective!(mop, g2; n_out = 1, model_cfg = RbfConfig(; kernel = :multiquadric))
        x, fx, ret_code, _ = optimize(mop, [-π, 2.71828]; algo_config = AlgorithmConfig(; descent_method, max_iter = 10))
        @test all(isfinite, fx)
    end
    @test length(Morbit._CTX) == n_ctx                       # no context was created: nothing of libmrbf ran
    @test Morbit._hip_affine_scan() === nothing              # the task-local scan state only lives inside a HipRbfConfig update
end

# Round 6: the filter's growing Householder factorisation against qr(Y) from scratch (what `_orthogonal_complement_matrix` does after
# every pick, AffinelyIndependentPoints.jl:4-11), and the device selection against the host loop
@testset "affine filter: one reflector per pick, device selection" begin
    for d in (3, 17, 64)
        Y = randn(d, d)
        g = Morbit.HipRbfGrowingQR(Matrix{Float64}(undef, d, 0))
        for j = 1:d
            Morbit._append!(g, Y[:, j])
            Z = Morbit._complement(g, Inf)
            Zf = Morbit._orthogonal_complement_matrix(Y[:, 1:j], Inf)
            @test size(Z) == size(Zf)
            isempty(Z) || @test maximum(abs.(Z .- Zf)) < 1e-12
        end
        g2 = Morbit.HipRbfGrowingQR(Y[:, 1:2])               # round 2: continue from a given Y
        Morbit._append!(g2, Y[:, 3])
        d >= 4 && @test maximum(abs.(Morbit._complement(g2, Inf) .- Morbit._orthogonal_complement_matrix(Y[:, 1:3], Inf))) < 1e-12
    end
    d, mc = 64, 400
    x0 = rand(d); S = 0.2 .* (2 .* rand(d, mc) .- 1)
    first = argmax([norm(S[:, c], Inf) for c = 1:mc])
    g = Morbit.HipRbfGrowingQR(reshape(S[:, first], d, 1))
    Sd = copy(S); Sd[:, first] .= 0
    picks, Zf = Morbit.affine_select(Sd, g.Q, g.j, d - 1, 0.02, Inf)
    # the host loop with the same rule
    want = Int[]; Sh = copy(Sd); gh = Morbit.HipRbfGrowingQR(reshape(S[:, first], d, 1))
    while length(want) < d - 1
        i, v = Morbit._affine_scores_host(Sh, Morbit._complement(gh, Inf), Inf)
        (i >= 1 && v > 0.02) || break
        push!(want, i); Morbit._append!(gh, S[:, i]); Sh[:, i] .= 0
    end
    @test picks == want
    @test size(Zf, 2) == d - 1 - length(picks)
end
