/* mrbf.h -- C ABI of the MI355X-native RBF surrogate engine (libmrbf.so).
 *
 * Drop-in boundary for Morbit.jl's RbfConfig hot path.  Morbit reaches this
 * arithmetic through Julia multiple dispatch into RadialBasisFunctionModels.jl;
 * each entry point below names the reference interface it replaces.  A Julia
 * maintainer binds these with `ccall` (see INTEGRATION.md and
 * morbit.jl_amd/julia/HipRbf.jl); tests and bench bind them with ctypes.
 *
 * Conventions
 *   - every call returns int32: 0 ok; <0 = -(1-based index of the invalid
 *     argument); >0 numerical / runtime error (MRBF_E*).  Nothing throws or
 *     aborts across the ABI.  mrbf_last_error(ctx) gives a static string.
 *   - all numbers are fp64.  Buffers may be HOST or DEVICE pointers (detected
 *     with hipPointerGetAttributes); host buffers are copied over PCIe inside
 *     the call, device buffers are used in place on the context's stream.
 *   - the caller owns in/out buffers for the duration of the call (Julia:
 *     GC.@preserve); the library owns ctx and model handles.
 *   - one ctx per host thread; a ctx is not thread-safe; ctxs are independent.
 *   - layouts (chosen so that Julia's native arrays are passed zero-copy):
 *       centres  n x d row-major  == Julia d x n column-major
 *                                 == reinterpret(reshape, Float64, Vector{SVector{d}})
 *       values   n x k row-major  == Julia k x n column-major (Vector{MVector{k}})
 *       weights  n x k row-major, poly q x k row-major, basis [1, x_1..x_d]
 *       X        m x d row-major;  vals m x k row-major
 *       jac      per point a k x d COLUMN-major block (unsafe_wrap as Matrix k x d)
 *       Phi n x n, Pi n x q column-major.
 */
#ifndef MRBF_H
#define MRBF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mrbf_ctx mrbf_ctx;
typedef struct mrbf_model mrbf_model;

/* kernel ids in the order of Morbit.RbfKernels (src/models/RbfModel.jl:48-54).
 * (a, b) are what _get_kernel_params (RbfModel.jl:665-690) hands to the package:
 *   cubic: a = beta (odd, default 3)              inv_multiquadric / multiquadric: a = alpha, b = beta (default 1, 1/2)
 *   thin_plate_spline: a = k (default 2)          gaussian: a = alpha (default 1) */
enum {
    MRBF_CUBIC = 0,
    MRBF_INV_MULTIQUADRIC = 1,
    MRBF_MULTIQUADRIC = 2,
    MRBF_THIN_PLATE_SPLINE = 3,
    MRBF_GAUSSIAN = 4
};

enum {
    MRBF_OK = 0,
    MRBF_ENOTPD = 1,    /* Cholesky met a non-positive pivot and the LU retry was disabled */
    MRBF_ESINGULAR = 2, /* LU met an exactly zero pivot (info.factor_info = its 1-based index) */
    MRBF_EHIP = 3,      /* a HIP runtime call failed */
    MRBF_EBLAS = 4,     /* a rocBLAS / rocSOLVER call failed */
    MRBF_ENOMEM = 5,
    MRBF_ENODEVICE = 6, /* no usable GPU: the library never computes on the CPU */
    MRBF_ENCCL = 7
};

/* solve paths reported in mrbf_fit_info.path */
enum {
    MRBF_PATH_CHOL = 1,      /* Phi SPD, no tail: potrf(Phi) */
    MRBF_PATH_PROJ_CHOL = 2, /* tail + conditionally p.d. kernel: potrf(P Phi P + mu Q1 Q1') on null(Pi') */
    MRBF_PATH_LU = 3,        /* indefinite saddle system [Phi Pi; Pi' 0]: getrf */
    MRBF_PATH_MINNORM = 4,   /* n < q (fewer sites than tail terms): minimum-norm solution by SVD */
    MRBF_PATH_ROUND4 = 5     /* mrbf_fit_from_round4: two triangular solves with the factor the site selection left behind */
};

/* mrbf_set_option keys */
enum {
    MRBF_OPT_GRAM_MODE = 1,    /* 0 = MFMA GEMM-form (default), 1 = VALU difference-form (reference arithmetic) */
    MRBF_OPT_RESIDUAL = 2,     /* 1 = compute rel_residual / max|Pi'w| after a fit (default 1) */
    MRBF_OPT_FORCE_PATH = 3,   /* 0 = automatic, else one of MRBF_PATH_* */
    MRBF_OPT_CHOL_IMPL = 4,    /* 0 = library default (3 from 256 columns on, else 2), 1 = rocSOLVER potrf, 2 = built-in blocked MFMA Cholesky driven by
                                  host launches, 3 = the same factorisation as one persistent launch */
    MRBF_OPT_EVAL_IMPL = 5,    /* 0 = default, 1 = GEMM pipeline, 2 = fused MFMA kernel */
    MRBF_OPT_TIMING = 6,       /* 1 = record per-phase hipEvents (default 1) */
    MRBF_OPT_DIAG_IMPL = 7,    /* diagonal-block kernel of the built-in Cholesky: 0 = MFMA-tiled (default), 1 = column sweep (host-driven
                                  factorisation only) */
    MRBF_OPT_CHOL_WINDOW = 8,  /* panels aggregated per trailing update: 0 = size-dependent schedule (default), else 1, 2 or 4 */
    MRBF_OPT_SPIN_MS = 9,      /* wall-clock limit (ms) a persistent kernel waits on one dependency without progress before it gives up
                                  and the call falls back to the host-driven GPU path (default 1000) */
    MRBF_OPT_LAST_DEVICE_MS = 11, /* read only: time the last persistent factorisation spent on the device, by its own clock (first
                                  workgroup in -> last workgroup out); unlike the hipEvent phases it excludes whatever the host did
                                  between the events */
    MRBF_OPT_SLOW_LAUNCHES = 12,  /* read only: persistent factorisations of this context that took more than twice the shortest seen
                                  at their shape (cumulative) */
    MRBF_OPT_ARENA_BYTES = 13,   /* read only: bytes of device memory the context's grow-only arena (named work buffers) and its pool of released
                                  * model blocks hold: constant in steady state (a caller can watch it over its iterations) */
    MRBF_OPT_LIVE_HANDLES = 14,  /* read only: models + round-4 states created through this context and not yet released */
    MRBF_OPT_DEBUG_FAULT = 10  /* test hook: bit 0 = one workgroup of the persistent factorisation skips a publish, bit 1 = one workgroup of
                                  the persistent backward substitution does (the next fit must fall back and still return the right weights), bit 2 = one
                                  member of a small fit's workgroup cluster leaves early (the fit is repeated with one workgroup per problem) */
};

/* bits of mrbf_fit_info.fallbacks: a GPU path that was abandoned for another GPU path inside the same call (rc stays 0) */
enum {
    MRBF_FB_CHOL_HOST_DRIVEN = 1, /* persistent Cholesky gave up on a dependency -> re-assembled, host-driven blocked Cholesky */
    MRBF_FB_BACKSOLVE_BLOCKED = 2, /* persistent backward substitution gave up -> one launch per block row */
    MRBF_FB_LU = 4                /* Cholesky met a non-positive pivot / rank-deficient tail -> LU of the saddle system */
};

typedef struct {
    int32_t path;         /* MRBF_PATH_* actually taken */
    int32_t factor_info;  /* potrf / getrf info (0 = clean) */
    int32_t n, q;         /* system sizes */
    double rel_residual;  /* ||Phi w + Pi lam - Y||_F / ||Y||_F recomputed through the eval kernels; NaN if disabled */
    double max_pitw;      /* max |Pi' w|; NaN if disabled */
    double mu;            /* shift used by the projected Cholesky (0 otherwise) */
    float ms_gram, ms_project, ms_factor, ms_solve, ms_check, ms_total; /* hipEvent times on the ctx stream */
    int32_t fallbacks;    /* MRBF_FB_* bits */
    int32_t giveup_code;  /* diagnostic code of the last give-up (0 if none) */
    float ms_factor_device; /* persistent factorisation: its time by the launch's own clock (0 on other paths); ms_factor - this =
                               host-side gaps inside the hipEvent bracket (descheduled launcher thread, job-table rebuild) */
    int32_t slow_launches;  /* MRBF_OPT_SLOW_LAUNCHES of the context after this fit */
} mrbf_fit_info;

typedef struct {
    float ms_total;
    float ms_dist, ms_kernel, ms_contract; /* phases of the evaluation (0 when fused) */
} mrbf_eval_info;

/* per-GPU context: stream, rocBLAS handle, grow-only workspace arena.
 * device_id < 0 selects the current device. */
int32_t mrbf_init(int32_t device_id, mrbf_ctx **ctx);
int32_t mrbf_shutdown(mrbf_ctx *ctx);
const char *mrbf_last_error(const mrbf_ctx *ctx); /* ctx may be NULL: last init error */
const char *mrbf_version(void);
int32_t mrbf_set_option(mrbf_ctx *ctx, int32_t key, double value);
int32_t mrbf_get_option(const mrbf_ctx *ctx, int32_t key, double *value);
/* run on the caller's hipStream_t (e.g. torch's current stream); NULL restores the ctx's own stream */
int32_t mrbf_set_stream(mrbf_ctx *ctx, void *hip_stream);
int32_t mrbf_sync(mrbf_ctx *ctx);

/* Phi (n x n) and Pi (n x q) -- replaces RBF.get_matrices(phi, centers; poly_deg)
 * (src/models/RbfModel.jl:374-375).  Pi_out may be NULL.  *ms (may be NULL)
 * receives the hipEvent time of the assembly kernels alone. */
int32_t mrbf_gram(mrbf_ctx *ctx, int64_t n, int32_t d, const double *centres, int32_t kernel_id, double a, double b,
                  int32_t poly_deg, double *Phi_out, double *Pi_out, float *ms);

/* rectangular kernel block K[i][j] = phi(||x_i - c_j||), m x n row-major -- replaces the per-candidate
 * kernels(xi) / RBF.make_kernel calls of the round-4 site selection (src/models/RbfModel.jl:421, :487): all candidates
 * against all current and prospective centres in one launch (difference-form arithmetic, like norm(x - c)). */
int32_t mrbf_cross_gram(mrbf_ctx *ctx, int64_t m, int64_t n, int32_t d, const double *X, const double *centres, int32_t kernel_id,
                        double a, double b, double *K_out);

/* Gram assembly + factorisation + solve for all k outputs -- replaces
 * RBF.RBFInterpolationModel(sites, values, kernel, params, poly_deg) in update_model
 * (src/models/RbfModel.jl:759-763).  Keeps centres / weights resident on the device in
 * *model.  weights_out (n x k), poly_out (q x k) and info may be NULL. */
int32_t mrbf_fit(mrbf_ctx *ctx, int64_t n, int32_t d, int32_t k, const double *centres, const double *values,
                 int32_t kernel_id, double a, double b, int32_t poly_deg, mrbf_model **model, double *weights_out,
                 double *poly_out, mrbf_fit_info *info);

/* build a model from known coefficients (no solve): lets a caller that kept the
 * round-4 factors (RbfModel.jl:657-660 note) or a checkpoint skip the fit */
int32_t mrbf_model_from_coeffs(mrbf_ctx *ctx, int64_t n, int32_t d, int32_t k, const double *centres,
                               const double *weights, const double *poly, int32_t kernel_id, double a, double b,
                               int32_t poly_deg, mrbf_model **model);

/* values and Jacobians at m points, all k outputs in one sweep -- replaces
 * model(x), model(x, l), RBF.grad(model, x, l), RBF.jac(model, x, rows)
 * (src/models/RbfModel.jl:783-800) and the per-output closures of
 * _get_optim_handle (src/AbstractSurrogateInterface.jl:98-106).
 * vals_out (m x k) or jac_out (m x [k x d col-major]) may be NULL. */
int32_t mrbf_eval(mrbf_ctx *ctx, const mrbf_model *model, int64_t m, const double *X, double *vals_out,
                  double *jac_out, mrbf_eval_info *info);

/* all step sizes of the Armijo backtracking loop in one batch -- replaces the
 * sequential loop of _backtrack (src/descent.jl:150-185) with condition
 * _armijo_condition (src/descent.jl:137-143).  Evaluates x + step0*shrink^i*dir,
 * i = 0..max_loops, and returns the first i that satisfies the condition (or the
 * index at which the reference loop would have stopped).  x_plus (d), mx_plus (k),
 * step (d) and n_loops are outputs (host or device). */
int32_t mrbf_backtrack(mrbf_ctx *ctx, const mrbf_model *model, const double *x, const double *dir, double step0,
                       double omega, int32_t strict, double const_rhs, double shrink, double min_stepsize,
                       int32_t max_loops, double *x_plus, double *mx_plus, double *step, int32_t *n_loops);

/* ---- rounds 1-2 of the training-site selection: the candidate scan of the affinely-independent-point filter ------------
 * val(xi) = || Z (Z' (xi - x0)) ||_p for all candidates at once and the first maximiser -- replaces the loop over
 * filter.candidate_indices in Base.iterate(::AffinelyIndependentPointFilter, n) (src/models/AffinelyIndependentPoints.jl:71-106).
 * shifted: mc x d row-major rows xi - x0 (host or device; already chosen sites: pass a zero row); Z: d x dz column-major, the
 * p-normalised complement basis of AffinelyIndependentPoints.jl:4-12 (the caller's d x d QR); p_is_inf: 1 = inf-norm (the
 * filter's default in _find_suitable_points, RbfModel.jl:226), 0 = 2-norm.  vals_out (mc) may be NULL. */
int32_t mrbf_affine_scores(mrbf_ctx *ctx, int64_t mc, int32_t d, int32_t dz, const double *shifted, const double *Z, int32_t p_is_inf,
                           double *vals_out, int64_t *argmax, double *maxval);
/* The whole pick loop of the filter in one call (round 6): from the directions chosen so far -- Q0, d x d column-major, an orthogonal
 * matrix whose first j0 columns span them (the Q of qr(Y), full; NULL with j0 = 0) -- pick up to max_picks further candidates, each
 * time the first maximiser of || Z (Z' (xi - x0)) ||_p while that exceeds pivot_val (AffinelyIndependentPoints.jl:71-106), with the
 * Householder factorisation of Y = [chosen directions] grown by one reflector per pick on the device instead of qr(Y) from scratch
 * after every pick (:59-60, :93-94): O(d^2 + d mc) per pick, no host round trip between picks.  shifted: mc x d row-major rows
 * xi - x0 (host or device; sites chosen already: zero rows).  picked_out[max_picks]: positions of the picked candidates in pick order;
 * Z_out (d x (d - j0 - *n_picked), column-major, host or device, may be NULL): the p-normalised complement basis afterwards -- the
 * filter's final Z (its improving directions, RbfModel.jl:231-235). */
int32_t mrbf_affine_select(mrbf_ctx *ctx, int64_t mc, int32_t d, const double *shifted, int32_t j0, const double *Q0, int32_t max_picks,
                           double pivot_val, int32_t p_is_inf, int64_t *picked_out, int32_t *n_picked, double *Z_out);

/* ---- round 4 of the training-site selection on the device, with factor reuse -------------------------------------------
 * mrbf_round4 replaces _rbf_round4 (src/models/RbfModel.jl:352-499): start_sites (n0 x d, the sites found so far -- centre and
 * rounds 1-3 -- which must carry the polynomial tail: n0 >= q, full rank) and cand_sites (mc x d, the database candidates in
 * database order; with use_max_points the caller appends the random box points of :405-416) -> positions (into cand_sites) of
 * the accepted sites in acceptance order.  Acceptance test as in the reference: tau^2 > (theta_pivot_cholesky^2)^2 (:370, :452),
 * at most max_points sites in total (<= 0: (d+1)(d+2)/2, :356).  The kernel vectors of ALL candidates, the reference's
 * kernels(xi) / get_matrices calls (:374, :421), are three batched launches; the selection is a left-looking Cholesky
 * factorisation that skips the rejected columns (round4.hip).  *state (may be NULL) keeps the factors.
 * mrbf_fit_from_round4: the model on (start sites, accepted sites) -- values ((n0 + n_accepted) x k row-major, in that order)
 * -- from the kept factor instead of a new factorisation (the reference's TODO at RbfModel.jl:657-660); needs n0 == q
 * (returns -2 otherwise: use mrbf_fit).  info.path = MRBF_PATH_ROUND4. */
typedef struct mrbf_round4_state mrbf_round4_state;
int32_t mrbf_round4(mrbf_ctx *ctx, int64_t n0, int32_t d, const double *start_sites, int64_t mc, const double *cand_sites,
                    int32_t kernel_id, double a, double b, int32_t poly_deg, int32_t max_points, double theta_pivot_cholesky,
                    int32_t *accepted_out, int32_t *n_accepted, mrbf_round4_state **state);
int32_t mrbf_fit_from_round4(mrbf_ctx *ctx, const mrbf_round4_state *state, int32_t k, const double *values, mrbf_model **model,
                             double *weights_out, double *poly_out, mrbf_fit_info *info);
int32_t mrbf_round4_sites(const mrbf_round4_state *state, int64_t *n0, int64_t *n_candidates, int32_t *n_accepted);
int32_t mrbf_free_round4(mrbf_ctx *ctx, mrbf_round4_state *state);

int32_t mrbf_model_dims(const mrbf_model *model, int64_t *n, int32_t *d, int32_t *k, int32_t *q);
int32_t mrbf_free_model(mrbf_ctx *ctx, mrbf_model *model);

/* many-problem mode (the reference's Threads.@threads loop over independent
 * problems, examples/large_scale_benchmarks.jl:253): problem p runs on device
 * device_ids[p % n_dev] (device_ids == NULL: devices 0 .. n_dev-1), one host
 * thread + pooled ctx per entry of device_ids inside this process (entries may
 * repeat: {0, 0} is two workers on GPU 0), fit + eval per problem, one
 * fixed-size result record per problem; a failing problem sets `status` on its
 * own record only, the call itself returns 0 unless an argument is invalid or
 * a device cannot be initialised.
 * Every buffer of a problem may be a HOST pointer or a DEVICE pointer (detected
 * per pointer, as in mrbf_fit / mrbf_eval; outputs given as device pointers are
 * written in place, without a host round trip).  With n_dev > 1 a DEVICE buffer
 * of problem p must live on device_ids[p % n_dev], the device that runs it --
 * the library does no peer access; host buffers have no such constraint. */
typedef struct {
    int64_t n, m;
    int32_t d, k, kernel_id, poly_deg;
    double a, b;
    const double *centres; /* host or device, n x d row-major */
    const double *values;  /* host or device, n x k row-major */
    const double *X;       /* host or device, m x d row-major (may be NULL when m == 0) */
    double *weights_out;   /* host or device n x k, or NULL */
    double *poly_out;      /* host or device q x k, or NULL */
    double *vals_out;      /* host or device m x k, or NULL */
    double *jac_out;       /* host or device m x (k x d column-major blocks), or NULL */
} mrbf_problem;

typedef struct {
    int32_t status;   /* return code of the failing call, 0 if ok */
    int32_t device;   /* GPU that ran it */
    mrbf_fit_info fit;
    float ms_eval;
    double checksum_w;    /* sum of weights: order-fixed reduction on the device for the problems of the batched small-problem path
                             (batch.hip), a host loop over the output buffer for problems on the per-problem chain */
    double checksum_vals; /* sum of values, likewise */
} mrbf_result;

int32_t mrbf_batch_run(int32_t n_dev, const int32_t *device_ids, int64_t n_problems, const mrbf_problem *problems,
                       mrbf_result *results);

/* debug / test hooks (exported so the parity tests can pin kernel-level behaviour) */
/* Environment switches.  The library reads a number of MRBF_* variables (schedule experiments of the persistent factorisation,
 * earlier forms of kernels kept selectable for A/B runs, diagnostics: INTEGRATION.md "Tuning knobs").  NONE of them is honoured
 * unless MRBF_EXPERIMENTS=1 is set in the same environment: a stray variable never changes the algorithm of a production run.
 * mrbf_debug_env (host only): 1 if switch `name` is set and the gate is open, else 0. */
int32_t mrbf_debug_env(const char *name);
int32_t mrbf_debug_mfma_layout(mrbf_ctx *ctx, double *out16x16_a_times_b, const double *A16x4, const double *B4x16);
int32_t mrbf_debug_potrf(mrbf_ctx *ctx, int64_t n, double *A_colmajor_inout, int32_t impl, int32_t *info, float *ms);
int32_t mrbf_debug_diag(mrbf_ctx *ctx, const double *A128, int32_t reps, float *ms_per_call, double *shader_cycles,
                        double *realtime_us);
/* micro-benchmarks (tools/microbench.py): sustained f64 MFMA issue rate and the rocBLAS dgemm reference */
int32_t mrbf_debug_mfma_peak(mrbf_ctx *ctx, int32_t blocks_per_cu, int32_t threads, int32_t iters, float *ms, double *tflops);
int32_t mrbf_debug_mfma_asm(mrbf_ctx *ctx, int32_t variant, int32_t blocks_per_cu, int32_t iters, float *ms, double *tflops,
                            double *cycles_per_mfma);
int32_t mrbf_debug_dgemm(mrbf_ctx *ctx, int32_t m, int32_t n, int32_t k, float *ms, double *tflops);
/* Host only (no GPU needed): the job tables of the persistent factorisation for nt block columns / mt block rows and the given
 * schedule parameters, checked against their invariants.  out[0..3] = panel jobs, bulk jobs, chain jobs, windows; out[4] = checksum;
 * out[5] = violated invariants (0 when the tables are consistent).  Returns 0, or -(argument index) for a parameter out of range. */
int32_t mrbf_debug_mega_tables(int32_t nt, int32_t mt, int32_t slack, int32_t slack_chain, int32_t first, int32_t win, int32_t srows,
                               int32_t half_cols, int64_t *out6);
/* The same with the round-4 schedule options: opt5 = { head columns, tail columns of the chain-bound edge regime, block columns at the
 * end whose bulk jobs are 64-row halves, only a tile's last so many windows as halves (0: all), chain tiles' window jobs in queues of
 * their own }.  The last option is a bit field: bit 0 chain tiles' queues; bit 1 streamed tiles of five-row block columns as two 64-row
 * jobs (bits 2..9 / 10..17: only the first / last so many block columns, 0 / 0 = all); bits 18..25 = t + 1: panel tiles more than t
 * block rows below the streamed ones as one 128-row job; bit 26: the block rows below the square hold at most 64 non-zero rows (one
 * job per tile and window on the upper half that publishes for both).  The invariants cover both queue classes, the per-column
 * number of streamed rows and every one of these job shapes (each tile finished / updated exactly once). */
int32_t mrbf_debug_mega_tables2(int32_t nt, int32_t mt, int32_t slack, int32_t slack_chain, int32_t first, int32_t win, int32_t srows,
                                int32_t half_cols, const int32_t *opt5, int64_t *out6);

/* ---- Pascoletti-Serafini descent step with the subproblem solver on the device ----------------------------------------
 * Replaces, for objectives that share ONE grouped RBF model and carry no modelled constraints, the NLopt runs inside
 * get_criticality(::PascolettiSerafiniConfig, ...) (src/descent.jl:512-581): compute_local_ideal_point (:404-412, k runs of
 * _min_component :369-387) when no direction is given, and _ps_optimization (:478-510) with the constraint functions of
 * :434-447 (chi = [t; x], t in [-1,0], x in [lb_eff, ub_eff], minimise t subject to m_l(x) - m_l(x_n) - t r_l <= 0).
 * Population state, ranking and breeding live on the device; one batched surrogate sweep evaluates a whole generation of
 * all runs.  x_n, lb_eff, ub_eff: d; fx_n (true objective values at x_n, used for r = fx_n - ideal) and r_or_null (the direction
 * of _get_global_dir, :360-368): k, host or device.  Outputs x_trial (d), mx_trial (k), r_out (k, may be NULL).
 * info.status: MRBF_PS_OK -> omega = |info.tau| (:573-579); MRBF_PS_CRITICAL -> some r_l <= 0 (:546-549): x_trial = x_n,
 * mx_trial = m(x_n), tau = 0; MRBF_PS_FAILURE -> no feasible point (:571-572), outputs as for CRITICAL. */
enum { MRBF_PS_OK = 0, MRBF_PS_CRITICAL = 1, MRBF_PS_FAILURE = 2 };
typedef struct {
    int32_t max_ideal_evals;  /* per objective; < 0: 500 (d + 1)   (descent.jl:527) */
    int32_t max_ps_evals;     /* < 0: 500 (d + 1); the caller applies _ps_max_evals' 3/4 split (descent.jl:414-432) */
    int32_t max_polish_evals; /* 0: no polish (ps_polish_algo = nothing) */
    int32_t reserved;
    uint64_t seed;            /* counter-based generator key: same seed, same step */
    double t0;                /* start value of t, -0.5 in the reference (descent.jl:555) */
    double xtol_rel;          /* <= 0: 1e-3 (descent.jl:379, :485) */
} mrbf_ps_options;
typedef struct {
    int32_t status;           /* MRBF_PS_* */
    int32_t generations;      /* batched generations over all runs */
    int32_t evals_ideal, evals_ps, evals_polish; /* surrogate evaluations spent */
    float ms_total;           /* hipEvent time of the whole step on the ctx stream */
    double tau;
} mrbf_ps_info;
int32_t mrbf_ps_step(mrbf_ctx *ctx, const mrbf_model *model, const double *x_n, const double *lb_eff, const double *ub_eff,
                     const double *fx_n, const double *r_or_null, const mrbf_ps_options *opts, double *x_trial, double *mx_trial,
                     double *r_out, mrbf_ps_info *info);

/* The same step for a whole SurrogateContainer: objectives and modelled constraints spread over several grouped RBF models
 * (src/SurrogateContainer.jl:48-64), the nonlinear (in)equality constraint handles of descent.jl:389-402 / :448-466 and the MOP's
 * linear constraints in scaled variables (transformed_linear_eq/ineq_constraints, src/AbstractMOPInterface.jl:463-495).
 * roles: one entry per output row of every model, concatenated in model order: l >= 0 -> this row is objective l of the
 * container's objective vector; MRBF_ROLE_EQ / MRBF_ROLE_INEQ -> modelled constraint h(x) = 0 / g(x) <= 0; MRBF_ROLE_NONE ->
 * not used.  A_eq (n_lin_eq x d row-major), b_eq: A x = b; A_ineq, b_ineq: A x <= b (host or device; may be NULL when the
 * count is 0).  An equality counts as satisfied when |h| <= eq_tol (< 0: 1e-8; NLopt's default tolerance 0 would make every
 * point infeasible).  The ideal-point runs (descent.jl:404-412) carry the same constraints.  fx_n, r, mx_trial, r_out have
 * n_objectives entries.  Limits of the device path: mrbf_dispatch_ps. */
enum { MRBF_ROLE_NONE = -1, MRBF_ROLE_EQ = -2, MRBF_ROLE_INEQ = -3 };
typedef struct {
    int32_t n_models, n_objectives;
    const mrbf_model *const *models;
    const int32_t *roles;
    int32_t n_lin_eq, n_lin_ineq;
    const double *A_eq, *b_eq, *A_ineq, *b_ineq;
    double eq_tol;
} mrbf_ps_problem;
int32_t mrbf_ps_step_problem(mrbf_ctx *ctx, const mrbf_ps_problem *problem, const double *x_n, const double *lb_eff,
                             const double *ub_eff, const double *fx_n, const double *r_or_null, const mrbf_ps_options *opts,
                             double *x_trial, double *mx_trial, double *r_out, mrbf_ps_info *info);
/* test hook: ONE stochastic ranking (the ISRES-style pairwise rule of the PS solver, descent.jl:478-510's optimiser) of a given
 * generation -- f[lam] objective values, phi[lam] constraint violations (0 feasible, inf outside the budget) -- by the kernels of
 * the step: impl 0 one workgroup, 1 several compute units (lam >= 1024: one wave per 64 individuals, the records in registers),
 * 2 the same giving up at once (the time-out path), 6 the sixteen-workgroup form of round 5 (records in LDS),
 * 7 a timing experiment (1 without the wait for the neighbours: not a ranking), 9 as 0 with the one-pair-per-thread loop through LDS
 * for populations below 1024 too (round 6 ranks those with a wave per 96 individuals inside the one workgroup),
 * 3 one workgroup with the plain sort of a feasible generation in its one-pair-per-thread form (the form before round 6),
 * 4 one workgroup with the step's own mu = ceil(lam / 7): only the parents are ranked (order_out[mu ..] = -1), 5 the same without
 * the parent selection (the whole population sorted).
 * order_out[lam]: individuals in rank order; *gave_up: the several-compute-unit kernel's failure word. */
int32_t mrbf_debug_ps_rank(mrbf_ctx *ctx, int32_t lam, const double *f, const double *phi, uint64_t seed, int32_t gen, int32_t impl,
                           int32_t *order_out, int32_t *gave_up);

/* ---- the decision table of the host bindings ---------------------------------------------------------------------------
 * Which implementation a binding (morbit.jl_amd/julia/HipRbf.jl, the Python mirror) takes for one call of Morbit's interface:
 * the device entry point (MRBF_DISPATCH_DEVICE) or Morbit's own method on the same arguments (MRBF_DISPATCH_REFERENCE; Julia:
 * `invoke` of the generic method, Python: the host mirror).  Pure host code (no ctx, no GPU), so both bindings make exactly the
 * same decisions and the CPU tests pin them.  A binding never raises because a size limit was hit.
 *   mrbf_dispatch_ps         get_criticality(::PascolettiSerafiniConfig, ...) (src/descent.jl:512-581).  n_foreign = objectives or
 *                            modelled constraints that are not RefSurrogate rows of a device model (CompositeSurrogate,
 *                            ExactModel, Taylor / Lagrange models); n_nl_constraints = modelled constraint rows, n_lin_constraints =
 *                            linear constraint rows of the MOP.
 *   mrbf_dispatch_backtrack  _backtrack (src/descent.jl:150-185): device iff the objectives are the outputs, in order, of ONE
 *                            device model (then mrbf_backtrack applies); otherwise the reference loop (on batched container
 *                            sweeps where the binding has them).
 *   mrbf_dispatch_affine     candidate scan of the affinely-independent-point filter (AffinelyIndependentPoints.jl:71-106).
 *   mrbf_dispatch_round4     _rbf_round4 (src/models/RbfModel.jl:352-499): device iff the start set can carry the tail (n0 >= q)
 *                            and there are candidates.
 *   mrbf_dispatch_fit        update_model (RbfModel.jl:743-767): MRBF_FIT_FROM_ROUND4 iff a kept round-4 state describes exactly
 *                            the training set (state_n0 + state_n_accepted == n_training, same sites in the same order:
 *                            same_sites != 0) with a unisolvent start set (state_n0 == q), at least one accepted site and at most
 *                            1024 training sites (beyond that the ordinary fit is as fast and more accurate).
 *   mrbf_dispatch_after      the return code rc of a device entry point (MRBF_ENTRY_*) that means "take the reference method
 *                            for this call" (start set without the tail or rank deficient, limits of the device path) rather
 *                            than an error: 1 = fall back, 0 = rc is what it says. */
enum { MRBF_DISPATCH_REFERENCE = 0, MRBF_DISPATCH_DEVICE = 1 };
enum { MRBF_FIT_FULL = 0, MRBF_FIT_FROM_ROUND4 = 1 };
enum { MRBF_ENTRY_ROUND4 = 1, MRBF_ENTRY_FIT_FROM_ROUND4 = 2, MRBF_ENTRY_PS_STEP = 3, MRBF_ENTRY_BACKTRACK = 4, MRBF_ENTRY_AFFINE = 5 };
int32_t mrbf_dispatch_ps(int32_t d, int32_t k, int32_t n_models, int32_t n_nl_constraints, int32_t n_lin_constraints, int32_t n_foreign);
int32_t mrbf_dispatch_backtrack(int32_t n_objective_models, int32_t n_foreign, int32_t outputs_in_order);
int32_t mrbf_dispatch_affine(int64_t n_candidates, int32_t d);
int32_t mrbf_dispatch_round4(int64_t n0, int32_t d, int32_t poly_deg, int64_t n_candidates);
int32_t mrbf_dispatch_fit(int64_t n_training, int64_t state_n0, int32_t state_q, int32_t state_n_accepted, int32_t same_sites);
int32_t mrbf_dispatch_after(int32_t entry, int32_t rc);

/* ---- host-side helper of the Pascoletti-Serafini subproblem solver -------------------------------------------------
 * Stochastic ranking of one ISRES generation (Runarsson & Yao): lam sweeps over adjacent individuals, compared by
 * objective f when both are feasible (phi == 0) or with probability pf, else by constraint violation phi; stops after a
 * sweep without a swap.  Replaces what NLopt's GN_ISRES does internally for `_ps_optimization` /
 * `compute_local_ideal_point` (src/descent.jl:369-387, :478-510); the generation itself is evaluated by ONE mrbf_eval.
 * u: lam * (lam - 1) uniform random numbers in [0,1) (sweep-major), idx_out: the ranked permutation (best first).
 * Pure host code (no ctx, no device). */
int32_t mrbf_stochastic_rank(int32_t lam, const double *f, const double *phi, const double *u, double pf, int32_t *idx_out);

#ifdef __cplusplus
}
#endif
#endif /* MRBF_H */
