// Batched surrogate value + Jacobian evaluation, all k outputs per sweep -- replaces
// model(x[, l]) / RBF.grad / RBF.jac (/root/reference/src/models/RbfModel.jl:783-800).
//
//   s_l(x)  = sum_i w_li phi(rho_i) + p_l(x)
//   J_l(x)  = sum_i w_li psi(rho_i) (x - c_i) + grad p_l,   psi = phi'(rho)/rho
//           = (sum_i a_li) x - sum_i a_li c_i,               a_li = w_li psi(rho_i)
//
// Implementation 1 ("GEMM pipeline"): distances in GEMM form on centred coordinates
// (rocBLAS dgemm for the plain inner-product block), a fused radial-function / weighting /
// row-reduction kernel, one more plain dgemm for sum_i a_li c_i, and an assembly kernel that
// writes the per-point k x d column-major Jacobian blocks.
#include "radial.hpp"

namespace mrbf {

template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double *red /* 4*NV doubles */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        double x = v[c];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) red[wave * NV + c] = x;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] = (red[c] + red[NV + c]) + (red[2 * NV + c] + red[3 * NV + c]);
    __syncthreads();
}

// one block per query point p; E[i + p*ldE] = <xc_p, cc_i>
template <int KID, int KB>
__global__ __launch_bounds__(256) void eval_rows_kernel(const double *__restrict__ E, int64_t n, int64_t ldE,
                                                        const double *__restrict__ xsq, const double *__restrict__ sq,
                                                        const double *__restrict__ W, int k, int l0, KP kp,
                                                        double *__restrict__ A, int64_t mb,
                                                        const double *__restrict__ Xorig, int d,
                                                        const double *__restrict__ lam, int q,
                                                        double *__restrict__ vals, double *__restrict__ sa_out) {
    __shared__ double red[4 * 2 * KB];
    const int64_t p = blockIdx.x;
    const double xs = xsq[p];
    const double *Ep = E + p * ldE;
    double acc[2 * KB];  // [0..KB) value sums, [KB..2KB) sums of a_l
#pragma unroll
    for (int c = 0; c < 2 * KB; ++c) acc[c] = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        double s = fma(-2.0, Ep[i], xs + sq[i]);
        s = s > 0.0 ? s : 0.0;
        double phi, psi;
        rbf_phi_psi<KID>(s, kp, phi, psi);
#pragma unroll
        for (int l = 0; l < KB; ++l) {
            if (l0 + l < k) {
                const double w = W[i * k + l0 + l];
                acc[l] = fma(w, phi, acc[l]);
                const double a = w * psi;
                acc[KB + l] += a;
                if (A) A[((int64_t)(l0 + l) * mb + p) * n + i] = a;
            }
        }
    }
    // polynomial tail p_l(x) = lam[0,l] + sum_t lam[t+1,l] x_t  (original coordinates)
    if (q > 0) {
        for (int t = threadIdx.x; t < q; t += 256) {
            const double xv = (t == 0) ? 1.0 : Xorig[p * d + (t - 1)];
#pragma unroll
            for (int l = 0; l < KB; ++l)
                if (l0 + l < k) acc[l] = fma(lam[(int64_t)t * k + l0 + l], xv, acc[l]);
        }
    }
    block_reduce<2 * KB>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int l = 0; l < KB; ++l)
            if (l0 + l < k) {
                if (vals) vals[p * k + l0 + l] = acc[l];
                if (sa_out) sa_out[p * k + l0 + l] = acc[KB + l];
            }
    }
}

// jac[p][t*k + l] = sa[p,l] * xc[p,t] - Jt[(l*mb + p)*d + t] + lam[t+1, l]
__global__ void jac_assemble_kernel(const double *__restrict__ sa, const double *__restrict__ Xc, int dpad,
                                    const double *__restrict__ Jt, int64_t mb, int d, int k,
                                    const double *__restrict__ lam, int q, double *__restrict__ jac) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= mb * d) return;
    const int64_t p = idx / d;
    const int t = (int)(idx % d);
    const double x = Xc[p * dpad + t];
    for (int l = 0; l < k; ++l) {
        double v = fma(sa[p * k + l], x, -Jt[((int64_t)l * mb + p) * d + t]);
        if (q > 1) v += lam[(int64_t)(t + 1) * k + l];
        jac[p * (int64_t)k * d + (int64_t)t * k + l] = v;
    }
}

static int eval_gemm_pipeline(mrbf_ctx *ctx, const mrbf_model *M, int64_t m, const double *X, double *vals, double *jac,
                              mrbf_eval_info *info) {
    const int64_t n = M->n;
    const int d = M->d, dpad = M->dpad, k = M->k, q = M->q;
    // query chunk: E (n x mb) + A (k x mb x n) under ~1.5 GB
    int64_t mb = (int64_t)(1.5e9 / (8.0 * (double)n * (jac ? (1 + k) : 1)));
    mb = std::max<int64_t>(64, std::min<int64_t>(mb, m));
    mb = std::min<int64_t>(mb, 65535 * 16);
    double *Xq, *xsq, *E, *A = nullptr, *Jt = nullptr, *sa;
    MRBF_TRY(get_buf(ctx, S_EVAL_XC, (size_t)mb * dpad, &Xq));
    MRBF_TRY(get_buf(ctx, S_EVAL_XSQ, (size_t)mb, &xsq));
    MRBF_TRY(get_buf(ctx, S_EVAL_E, (size_t)mb * n, &E));
    MRBF_TRY(get_buf(ctx, S_EVAL_SA, (size_t)mb * k, &sa));
    if (jac) {
        MRBF_TRY(get_buf(ctx, S_EVAL_A, (size_t)k * mb * n, &A));
        MRBF_TRY(get_buf(ctx, S_EVAL_J, (size_t)k * mb * d, &Jt));
    }
    const bool timing = ctx->timing && info;
    float t_dist = 0.f, t_kern = 0.f, t_con = 0.f;
    const double one = 1.0, zero = 0.0;
    for (int64_t p0 = 0; p0 < m; p0 += mb) {
        const int64_t mc = std::min(mb, m - p0);
        const double *Xp = X + p0 * d;
        if (timing) MRBF_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
        MRBF_TRY(launch_center_pad(ctx, Xp, mc, d, M->mean, nullptr, Xq, mc, dpad, xsq));
        // E (n x mc) = Cc (n x dpad) * Xq^T: column-major views  Xc^T(dpad x npad) and Xq^T(dpad x mc)
        MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_transpose, rocblas_operation_none, (int)n, (int)mc, dpad,
                                     &one, M->Xc, dpad, Xq, dpad, &zero, E, (int)n));
        if (timing) MRBF_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
        for (int l0 = 0; l0 < k; l0 += 4) {
            const int kb = std::min(4, k - l0);
            double *vp = vals ? vals + p0 * k : nullptr;
#define MRBF_LAUNCH_ROWS(KBV)                                                                                        \
    MRBF_DISPATCH_KID(M->kp.kid, hipLaunchKernelGGL((eval_rows_kernel<KID, KBV>), dim3((unsigned)mc), dim3(256), 0,  \
                                                    ctx->stream, E, n, n, xsq, M->sq, M->W, k, l0, M->kp, A, mc, Xp, d, \
                                                    M->lam, q, vp, sa))
            if (kb == 1) { MRBF_LAUNCH_ROWS(1); }
            else if (kb == 2) { MRBF_LAUNCH_ROWS(2); }
            else { MRBF_LAUNCH_ROWS(4); }
#undef MRBF_LAUNCH_ROWS
        }
        MRBF_HIP(ctx, hipGetLastError());
        if (timing) MRBF_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
        if (jac) {
            // Jt (d x k*mc) = Cc^T (d x n) * A (n x k*mc)
            MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, d, (int)(k * mc),
                                         (int)n, &one, M->Xc, dpad, A, (int)n, &zero, Jt, d));
            hipLaunchKernelGGL(jac_assemble_kernel, dim3((unsigned)((mc * d + 255) / 256)), dim3(256), 0, ctx->stream, sa,
                               Xq, dpad, Jt, mc, d, k, M->lam, q, jac + p0 * (int64_t)k * d);
            MRBF_HIP(ctx, hipGetLastError());
        }
        if (timing) {
            MRBF_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
            MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[7]));
            float t;
            MRBF_HIP(ctx, hipEventElapsedTime(&t, ctx->ev[4], ctx->ev[5])); t_dist += t;
            MRBF_HIP(ctx, hipEventElapsedTime(&t, ctx->ev[5], ctx->ev[6])); t_kern += t;
            MRBF_HIP(ctx, hipEventElapsedTime(&t, ctx->ev[6], ctx->ev[7])); t_con += t;
        }
    }
    if (info) {
        info->ms_dist = t_dist;
        info->ms_kernel = t_kern;
        info->ms_contract = t_con;
        info->ms_total = t_dist + t_kern + t_con;
    }
    return 0;
}

int eval_fused(mrbf_ctx *ctx, const mrbf_model *M, int64_t m, const double *X, double *vals, double *jac,
               mrbf_eval_info *info);  // eval_fused.hip

int eval_model(mrbf_ctx *ctx, const mrbf_model *M, int64_t m, const double *Xdev, double *vals_dev, double *jac_dev,
               mrbf_eval_info *info) {
    if (info) std::memset(info, 0, sizeof(*info));
    if (m <= 0) return 0;
    if (M->n == 0) return fail(ctx, -2, "model has no centres");
    if (ctx->eval_impl != 1 && (M->dpad == 64 || M->dpad == 128 || M->dpad == 256)) return eval_fused(ctx, M, m, Xdev, vals_dev, jac_dev, info);
    return eval_gemm_pipeline(ctx, M, m, Xdev, vals_dev, jac_dev, info);
}

}  // namespace mrbf
