// Scores of the affinely-independent-point filter for ALL candidates at once -- replaces the candidate loop of
// Base.iterate(::AffinelyIndependentPointFilter, num_found) (/root/reference/src/models/AffinelyIndependentPoints.jl:71-106,
// used by _find_suitable_points, src/models/RbfModel.jl:205-238): for every candidate xi of the database box
//     val(xi) = || Z (Z' (xi - x0)) ||_p ,   p = inf (the filter's default, RbfModel.jl:226) or 2,
// and the FIRST maximiser (the reference's `>` scan keeps the earliest).  Z (d x dz, the p-normalised complement basis of the
// directions chosen so far) is recomputed by the caller after every pick -- a d x d QR, SURVEY.md section 8 row a12 -- the two
// tall products and the reduction over up to 10^5..10^6 database sites run here.
#include "common.hpp"

namespace mrbf {

// val[c] = norm_p(U[:, c]), U d x mc column-major
__global__ __launch_bounds__(256) void col_norms_kernel(const double *__restrict__ U, int d, int64_t mc, int use_inf, double *__restrict__ val) {
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= mc) return;
    double s = 0.0;
    for (int i = lane; i < d; i += 64) {
        const double u = fabs(U[i + c * d]);
        s = use_inf ? fmax(s, u) : fma(u, u, s);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(s, off);
        s = use_inf ? fmax(s, o) : s + o;
    }
    if (lane == 0) val[c] = use_inf ? s : sqrt(s);
}

// first maximiser: out[0] = index, out[1] = value (as doubles); single workgroup, ties -> smallest index
__global__ __launch_bounds__(1024) void argmax_first_kernel(const double *__restrict__ val, int64_t mc, double *__restrict__ out) {
    __shared__ double sv[1024];
    __shared__ long long si[1024];
    double bv = -INFINITY;
    long long bi = -1;
    for (int64_t c = threadIdx.x; c < mc; c += 1024) {
        const double v = val[c];
        if (v > bv) {  // ascending c per thread: `>` keeps the first maximiser of the thread's subsequence
            bv = v;
            bi = c;
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const double ov = sv[threadIdx.x + w];
            const long long oi = si[threadIdx.x + w];
            if (oi >= 0 && (si[threadIdx.x] < 0 || ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (double)si[0];
        out[1] = sv[0];
    }
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_affine_scores(mrbf_ctx *ctx, int64_t mc, int32_t d, int32_t dz, const double *shifted, const double *Z, int32_t p_is_inf,
                                      double *vals_out, int64_t *argmax, double *maxval) {
    if (!ctx) return -1;
    if (mc < 0 || mc > ((int64_t)1 << 26)) return fail(ctx, -2, "mc out of range");
    if (d < 1 || d > 4096) return fail(ctx, -3, "d out of range");
    if (dz < 0 || dz > d) return fail(ctx, -4, "dz must lie in [0, d]");
    if (argmax) *argmax = -1;
    if (maxval) *maxval = -INFINITY;
    if (mc == 0) return MRBF_OK;
    if (!shifted) return fail(ctx, -5, "shifted is NULL");
    if (dz > 0 && !Z) return fail(ctx, -6, "Z is NULL");
    (void)hipSetDevice(ctx->device);
    const double *S, *Zd = nullptr;
    double *T1, *U, *val, *res;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, shifted, (size_t)mc * d, &S));
    MRBF_TRY(get_buf(ctx, S_STAGE_C, (size_t)std::max(dz, 1) * mc, &T1));
    MRBF_TRY(get_buf(ctx, S_STAGE_D, (size_t)d * mc, &U));
    MRBF_TRY(get_buf(ctx, S_EVAL_SA, (size_t)mc, &val));
    MRBF_TRY(get_buf(ctx, S_MISC, (size_t)8, &res));
    if (dz > 0) {
        MRBF_TRY(stage_in(ctx, S_STAGE_B, Z, (size_t)d * dz, &Zd));
        const double one = 1.0, zero = 0.0;
        // the candidates, mc x d row-major, are S' (d x mc) column-major:  T1 = Z' S' (dz x mc),  U = Z T1 (d x mc)
        MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_transpose, rocblas_operation_none, dz, (int)mc, d, &one, Zd, d, S, d, &zero, T1, dz));
        MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, d, (int)mc, dz, &one, Zd, d, T1, dz, &zero, U, d));
    } else {
        MRBF_HIP(ctx, hipMemsetAsync(U, 0, (size_t)d * mc * sizeof(double), ctx->stream));  // empty complement: every score is 0
    }
    hipLaunchKernelGGL(col_norms_kernel, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, ctx->stream, U, d, mc, p_is_inf ? 1 : 0, val);
    hipLaunchKernelGGL(argmax_first_kernel, dim3(1), dim3(1024), 0, ctx->stream, val, mc, res);
    MRBF_HIP(ctx, hipGetLastError());
    double h[2] = {-1.0, 0.0};
    MRBF_HIP(ctx, hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    if (vals_out) MRBF_HIP(ctx, hipMemcpyAsync(vals_out, val, (size_t)mc * sizeof(double), hipMemcpyDefault, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (argmax) *argmax = (int64_t)h[0];
    if (maxval) *maxval = h[1];
    return MRBF_OK;
}
