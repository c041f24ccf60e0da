// Built-in blocked Cholesky (placeholder: forwards to rocSOLVER until the MFMA panel/update kernels land).
#include "common.hpp"

namespace mrbf {

int potrf_blocked(mrbf_ctx *ctx, int64_t n, double *A, int64_t lda, int *dinfo) {
    MRBF_BLAS(ctx, rocsolver_dpotrf(ctx->blas, rocblas_fill_lower, (int)n, A, (int)lda, dinfo));
    return 0;
}

}  // namespace mrbf
