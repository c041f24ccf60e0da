// The decision table of the host bindings (include/mrbf.h, "decision table"): which call of Morbit's surrogate / descent interface
// goes to a device entry point and which to Morbit's own method.  Pure host code: HipRbf.jl and the Python mirror call these
// functions instead of each carrying a copy of the rules, and tests/test_dispatch.py pins them without a GPU.
#include "common.hpp"

namespace {
// limits of ps_solver.hip (MAXLAM, MAXOBJ, MAXMODELS, MAXCON); the population of the PS run is 20 (d + 2)
constexpr int PS_MAXLAM = 7168, PS_MAXOBJ = 8, PS_MAXMODELS = 8, PS_MAXCON = 32, PS_MAXLIN = 256;
// below this many candidate coordinates one host BLAS call beats a launch + two copies (tools/affine_bench.py)
constexpr int64_t AFFINE_MIN_WORK = 4096 * 8;
constexpr int64_t FIT_REUSE_MAX_N = 1024;
}  // namespace

extern "C" {

int32_t mrbf_dispatch_ps(int32_t d, int32_t k, int32_t n_models, int32_t n_nl_constraints, int32_t n_lin_constraints, int32_t n_foreign) {
    if (n_foreign != 0) return MRBF_DISPATCH_REFERENCE;                       // outer functions / other model families live on the host
    if (d < 1 || 20 * (d + 2) > PS_MAXLAM) return MRBF_DISPATCH_REFERENCE;    // d <= 356
    if (k < 1 || k > PS_MAXOBJ) return MRBF_DISPATCH_REFERENCE;
    if (n_models < 1 || n_models > PS_MAXMODELS) return MRBF_DISPATCH_REFERENCE;
    if (n_nl_constraints < 0 || n_nl_constraints > PS_MAXCON) return MRBF_DISPATCH_REFERENCE;
    if (n_lin_constraints < 0 || n_lin_constraints > PS_MAXLIN) return MRBF_DISPATCH_REFERENCE;
    return MRBF_DISPATCH_DEVICE;
}

int32_t mrbf_dispatch_backtrack(int32_t n_objective_models, int32_t n_foreign, int32_t outputs_in_order) {
    return (n_objective_models == 1 && n_foreign == 0 && outputs_in_order != 0) ? MRBF_DISPATCH_DEVICE : MRBF_DISPATCH_REFERENCE;
}

int32_t mrbf_dispatch_affine(int64_t n_candidates, int32_t d) {
    if (d < 1 || n_candidates < 1) return MRBF_DISPATCH_REFERENCE;
    // round 6: the device entry point is the whole pick loop (mrbf_affine_select, ~50 us per pick whatever the candidate count); on the host a
    // pick costs two products of d x (d - j) x candidates.  From d = 64 on the device wins as soon as there are more candidates than
    // directions (d = 128, 255 candidates: 6.5 against 22 ms per model update, profiles/r06_iteration_c4.txt); below, only for large boxes
    if (d >= 64 && n_candidates >= d) return MRBF_DISPATCH_DEVICE;
    return n_candidates * (int64_t)d >= AFFINE_MIN_WORK ? MRBF_DISPATCH_DEVICE : MRBF_DISPATCH_REFERENCE;
}

int32_t mrbf_dispatch_round4(int64_t n0, int32_t d, int32_t poly_deg, int64_t n_candidates) {
    const int q = mrbf::poly_dim(d, poly_deg);
    if (n_candidates < 1 || n0 < 1 || n0 > 8192) return MRBF_DISPATCH_REFERENCE;
    // the limits of mrbf_round4 itself (round4.hip: d <= 1024, at most 30 000 candidates -- four candidate x candidate matrices live on
    // the device): a database box beyond them takes the reference's loop, it does not raise
    if (d < 1 || d > 1024 || n_candidates > 30000) return MRBF_DISPATCH_REFERENCE;
    return n0 >= q ? MRBF_DISPATCH_DEVICE : MRBF_DISPATCH_REFERENCE;
}

int32_t mrbf_dispatch_fit(int64_t n_training, int64_t state_n0, int32_t state_q, int32_t state_n_accepted, int32_t same_sites) {
    if (state_n0 < 1 || state_n_accepted < 1 || !same_sites) return MRBF_FIT_FULL;
    if (state_n0 != state_q) return MRBF_FIT_FULL;  // the kept factor only covers the directions added by round 4
    // beyond ~1000 sites the ordinary fit (persistent Cholesky) is as fast as the two triangular solves with the kept factor (d = 64,
    // n = 2145: 1.6 ms both, tools/round4_bench.py) and more accurate (the kappa matrix is formed with cancellation: residual 2e-11
    // against 1e-15): reuse pays below that
    if (n_training > FIT_REUSE_MAX_N) return MRBF_FIT_FULL;
    return state_n0 + state_n_accepted == n_training ? MRBF_FIT_FROM_ROUND4 : MRBF_FIT_FULL;
}

int32_t mrbf_dispatch_after(int32_t entry, int32_t rc) {
    switch (entry) {
        // (-3 / -5: dimension / candidate count beyond the device path; MRBF_ENOMEM: the candidate matrices did not fit)
        case MRBF_ENTRY_ROUND4: return rc == -2 || rc == -3 || rc == -5 || rc == MRBF_ESINGULAR || rc == MRBF_ENOMEM;
        case MRBF_ENTRY_FIT_FROM_ROUND4: return rc == -2 || rc == MRBF_ESINGULAR || rc == MRBF_ENOTPD;
        case MRBF_ENTRY_PS_STEP: return rc == -2;
        default: return 0;
    }
}

}  // extern "C"
