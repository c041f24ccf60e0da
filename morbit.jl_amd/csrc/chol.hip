// Cholesky factorisation (lower) of the n x n s.p.d. system matrix.
//   impl 1: rocSOLVER dpotrf
//   impl 2: built-in blocked right-looking Cholesky, host-driven launches (chol_blocked.hip)
//   impl 3: the same factorisation as one persistent launch (chol_mega.hip)
#include <chrono>

#include "common.hpp"

namespace mrbf {

int potrf_blocked(mrbf_ctx *ctx, int64_t n, double *A, int64_t lda, int *dinfo);  // chol_blocked.hip
int debug_diag(mrbf_ctx *ctx, const double *A128_dev, int reps, float *ms_per_call, unsigned long long *stamps_host);
int launch_pad_identity(mrbf_ctx *ctx, double *A, int64_t n, int64_t npad, int64_t ld);  // chol_blocked.hip

// A holds an n x n s.p.d. matrix in a buffer of leading dimension lda >= round_up(n, 128) whose rows/columns
// [n, lda) belong to the caller too: the built-in factorisation pads them with the identity and works on
// the 128-aligned order.
int potrf_lower(mrbf_ctx *ctx, int impl, int64_t n, double *A, int64_t lda, int *info_host) {
    int *dinfo;
    MRBF_TRY(get_buf(ctx, S_INFO, (size_t)4, &dinfo));
    const int64_t npad = round_up(n, 128);
    if (impl == 0) impl = (lda >= npad) ? ((npad >= ctx->mega_min && npad <= ctx->mega_max) ? 3 : 2) : 1;
    if (impl == 2 || impl == 3) {
        const int saved = ctx->chol_impl, saved_min = ctx->mega_min;
        ctx->chol_impl = impl;
        if (impl == 3) ctx->mega_min = 0;
        struct Restore {
            mrbf_ctx *c;
            int a, b;
            ~Restore() { c->chol_impl = a; c->mega_min = b; }
        } restore{ctx, saved, saved_min};
        if (lda < npad) return fail(ctx, MRBF_EHIP, "built-in Cholesky needs lda >= round_up(n,128)");
        MRBF_TRY(launch_pad_identity(ctx, A, n, npad, lda));
        MRBF_TRY(potrf_blocked(ctx, npad, A, lda, dinfo));
    } else {
        MRBF_BLAS(ctx, rocsolver_dpotrf(ctx->blas, rocblas_fill_lower, (int)n, A, (int)lda, dinfo));
    }
    MRBF_HIP(ctx, hipMemcpyAsync(info_host, dinfo, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MRBF_TRY(mega_collect_stat(ctx));
    return 0;
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_debug_potrf(mrbf_ctx *ctx, int64_t n, double *A, int32_t impl, int32_t *info, float *ms) {
    if (!ctx) return -1;
    if (n <= 0) return -2;
    if (!A) return -3;
    (void)hipSetDevice(ctx->device);
    // work on a 128-padded copy (ld = npad) so that both implementations see the same layout
    const int64_t npad = round_up(n, 128);
    double *dA;
    MRBF_TRY(get_buf(ctx, S_PHI, (size_t)npad * npad, &dA));
    const hipMemcpyKind in = is_device_ptr(A) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out = is_device_ptr(A) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    MRBF_HIP(ctx, hipMemcpy2DAsync(dA, (size_t)npad * sizeof(double), A, (size_t)n * sizeof(double), (size_t)n * sizeof(double),
                                   (size_t)n, in, ctx->stream));
    int hinfo = 0;
    static const double host_trace_ms = mrbf_env("MRBF_MEGA_HOSTTRACE") ? atof(mrbf_env("MRBF_MEGA_HOSTTRACE")) : 0.0;
    auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double h0 = now_ms();
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    const double h1 = now_ms();
    MRBF_TRY(potrf_lower(ctx, impl, n, dA, npad, &hinfo));  // (ends with a stream synchronisation)
    const double h2 = now_ms();
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    const double h3 = now_ms();
    if (host_trace_ms > 0.0 && h3 - h0 > host_trace_ms)
        fprintf(stderr, "mrbf_debug_potrf host trace (n %lld): record ev0 %.3f ms | potrf_lower incl. sync %.3f | record ev1 + sync %.3f\n", (long long)n,
                h1 - h0, h2 - h1, h3 - h2);
    if (ms) MRBF_HIP(ctx, hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
    if (info) *info = hinfo;
    MRBF_HIP(ctx, hipMemcpy2DAsync(A, (size_t)n * sizeof(double), dA, (size_t)npad * sizeof(double), (size_t)n * sizeof(double),
                                   (size_t)n, out, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MRBF_OK;
}

extern "C" int32_t mrbf_debug_diag(mrbf_ctx *ctx, const double *A128, int32_t reps, float *ms_per_call, double *shader_cycles,
                                   double *realtime_us) {
    if (!ctx) return -1;
    if (!A128) return -2;
    (void)hipSetDevice(ctx->device);
    const double *dA;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, A128, 128 * 128, &dA));
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    float ms = 0;
    MRBF_TRY(debug_diag(ctx, dA, reps, &ms, st));
    if (ms_per_call) *ms_per_call = ms;
    if (shader_cycles) *shader_cycles = (double)st[0];
    if (realtime_us) *realtime_us = (double)st[1] / 100.0;
    if (mrbf_env("MRBF_DIAG_VERBOSE"))
        fprintf(stderr, "diag segments (cycles over 128 columns, wave 0): barrier %llu read %llu critical %llu rest %llu\n", st[2], st[3],
                st[4], st[5]);
    return MRBF_OK;
}
