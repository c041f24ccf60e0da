// Cholesky factorisation (lower) of the n x n s.p.d. system matrix.
//   impl 1: rocSOLVER dpotrf
//   impl 2: built-in blocked right-looking Cholesky (chol_blocked.hip)
#include "common.hpp"

namespace mrbf {

int potrf_blocked(mrbf_ctx *ctx, int64_t n, double *A, int64_t lda, int *dinfo);  // chol_blocked.hip

int potrf_lower(mrbf_ctx *ctx, int impl, int64_t n, double *A, int64_t lda, int *info_host) {
    int *dinfo;
    MRBF_TRY(get_buf(ctx, S_INFO, (size_t)4, &dinfo));
    if (impl == 0) impl = 1;
    if (impl == 2) {
        MRBF_TRY(potrf_blocked(ctx, n, A, lda, dinfo));
    } else {
        MRBF_BLAS(ctx, rocsolver_dpotrf(ctx->blas, rocblas_fill_lower, (int)n, A, (int)lda, dinfo));
    }
    MRBF_HIP(ctx, hipMemcpyAsync(info_host, dinfo, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_debug_potrf(mrbf_ctx *ctx, int64_t n, double *A, int32_t impl, int32_t *info, float *ms) {
    if (!ctx) return -1;
    if (n <= 0) return -2;
    if (!A) return -3;
    (void)hipSetDevice(ctx->device);
    double *dA;
    const bool dev = is_device_ptr(A);
    if (dev) {
        dA = A;
    } else {
        MRBF_TRY(get_buf(ctx, S_PHI, (size_t)n * n, &dA));
        MRBF_HIP(ctx, hipMemcpyAsync(dA, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    int hinfo = 0;
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    MRBF_TRY(potrf_lower(ctx, impl, n, dA, n, &hinfo));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    if (ms) MRBF_HIP(ctx, hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
    if (info) *info = hinfo;
    if (!dev) {
        MRBF_HIP(ctx, hipMemcpyAsync(A, dA, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MRBF_OK;
}
