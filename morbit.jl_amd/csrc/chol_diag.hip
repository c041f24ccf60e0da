// D: Cholesky of one 128 x 128 diagonal block + inverse of its factor -- MFMA-tiled version (stand-alone launch).
// The algorithm lives in chol_diag_core.hpp.
#include "chol_diag_core.hpp"

namespace mrbf {

__global__ __launch_bounds__(256, 1) void chol_diag_v4_kernel(double *__restrict__ A, int64_t lda, double *__restrict__ Linv,
                                                              int *__restrict__ info, int col0, int dbg) {
    __shared__ __attribute__((aligned(16))) diagcore::DiagV4Shared sh;
    if (*info != 0) return;
    __builtin_amdgcn_s_setprio(3);  // latency-critical: outrank the bulk update's waves sharing this CU's SIMDs
    diagcore::v4d acc[diagcore::NSLOT];
    const int bad = diagcore::diag_v4_core<false, false, false>(A, lda, Linv, sh, acc, nullptr, nullptr, dbg);
    if (bad && threadIdx.x == 0) *info = col0 + bad;
}

// wave-specialised variant (chol_diag_core.hpp, v5), MRBF_OPT_DIAG_IMPL = 3: an experiment kept selectable and tested; it is
// not faster than v4 (tools/diagbench: 54-59 us against 55-57 us per block)
__global__ __launch_bounds__(256, 1) void chol_diag_v5_kernel(double *__restrict__ A, int64_t lda, double *__restrict__ Linv,
                                                              int *__restrict__ info, int col0) {
    __shared__ __attribute__((aligned(16))) diagcore::DiagV5Shared sh;
    if (*info != 0) return;
    __builtin_amdgcn_s_setprio(3);
    diagcore::v4d acc[diagcore::NSLOT5];
    const int bad = diagcore::diag_v5_core<false, false, false>(A, lda, Linv, sh, acc, nullptr, nullptr);
    if (bad && threadIdx.x == 0) *info = col0 + bad;
}

int launch_diag_v4(mrbf_ctx *ctx, hipStream_t st, double *Ajj, int64_t lda, double *Linv, int *dinfo, int col0) {
    static const int dbg = getenv("MRBF_DIAG_DBG") ? atoi(getenv("MRBF_DIAG_DBG")) : 0;
    if (ctx->diag_impl == 3 && !dbg)
        hipLaunchKernelGGL(chol_diag_v5_kernel, dim3(1), dim3(256), 0, st, Ajj, lda, Linv, dinfo, col0);
    else
        hipLaunchKernelGGL(chol_diag_v4_kernel, dim3(1), dim3(256), 0, st, Ajj, lda, Linv, dinfo, col0, dbg);
    return 0;
}

}  // namespace mrbf
