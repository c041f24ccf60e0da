// Interpolation-weight solve -- replaces the dense `\` inside RBF.RBFInterpolationModel
// (/root/reference/src/models/RbfModel.jl:759-763):   [Phi Pi; Pi' 0] [w; lam] = [Y; 0].
//
// Paths (mrbf_fit_info.path):
//   CHOL       q = 0, kernel strictly p.d. (gaussian, inverse multiquadric):  potrf(Phi), potrs.
//   PROJ_CHOL  q > 0 and the kernel is conditionally p.d. of order <= deg + 1 (the situation Morbit
//              itself relies on for cholesky(Z' Phi Z), RbfModel.jl:391-395).  With Pi = Q1 R (thin QR)
//              and P = I - Q1 Q1',  K = P Phi P + mu Q1 Q1' is s.p.d.;  K w = P Y gives w in null(Pi'),
//              lam = R^-1 Q1' (Y - Phi w).  K is formed IN PLACE on the lower triangle of Phi by one
//              symm + one syr2k + one syrk (4 n^2 q flops, all level 3), so the factorisation stays an
//              n x n blocked Cholesky on the matrix cores instead of an (n+q) LU.
//   LU         anything else (tail of too low a degree for the kernel's order, failed Cholesky):
//              getrf/getrs on the full saddle matrix.
#include "common.hpp"
#include "small.hpp"

namespace mrbf {

// the fit's read-backs in one launch: flag words, trace / shift and the persistent factorisation's device clock straight into the
// context's pinned words (host memory mapped into the device's address space)
__global__ void fit_publish_kernel(const int *__restrict__ dinfo, const double *__restrict__ scal, const unsigned long long *__restrict__ dstat,
                                   int *__restrict__ hflags, double *__restrict__ hscal, unsigned long long *__restrict__ hstat) {
    const int t = threadIdx.x;
    if (t < 4) hflags[t] = dinfo[t];
    if (t >= 4 && t < 6 && scal) hscal[t - 4] = scal[t - 4];
    if (t == 6 && dstat && hstat) hstat[0] = dstat[0];
}

__global__ void fill_saddle_kernel(const double *__restrict__ Pi, int64_t n, int q, double *__restrict__ S, int64_t N) {
    // S[0:n, n+t] = Pi[:, t];  S[n+t, 0:n] = Pi[:, t]';  S[n:, n:] = 0
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = n * q;
    if (idx < tot) {
        const int64_t i = idx % n;
        const int t = (int)(idx / n);
        const double v = Pi[idx];
        S[i + (n + t) * N] = v;
        S[(n + t) + i * N] = v;
    } else if (idx < tot + (int64_t)q * q) {
        const int64_t r = idx - tot;
        S[(n + r % q) + (n + r / q) * N] = 0.0;
    }
}

// B (N x k col-major, ld N): rows [0,n) from Y (n x k row-major), rows [n,N) zero
__global__ void rhs_from_values_kernel(const double *__restrict__ Y, int64_t n, int k, double *__restrict__ B, int64_t N) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * k) return;
    const int64_t i = idx % N;
    const int l = (int)(idx / N);
    B[idx] = (i < n) ? Y[i * k + l] : 0.0;
}

// W (n x k row-major), Wc (npad x k col-major zero padded), lam (q x k row-major) from solution B (ldB x k col-major)
__global__ void scatter_solution_kernel(const double *__restrict__ B, int64_t ldB, int64_t n, int64_t npad, int k, int q,
                                        int64_t lam_row0, double *__restrict__ W, double *__restrict__ Wc,
                                        double *__restrict__ lam, const double *__restrict__ lamsrc, int64_t ldlam) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < npad * k) {
        const int64_t i = idx % npad;
        const int l = (int)(idx / npad);
        const double v = (i < n) ? B[i + l * ldB] : 0.0;
        Wc[idx] = v;
        if (i < n) W[i * k + l] = v;
    } else if (idx < npad * k + (int64_t)q * k) {
        const int64_t r = idx - npad * k;
        const int t = (int)(r % q), l = (int)(r / q);
        lam[(int64_t)t * k + l] = lamsrc[lam_row0 + t + l * ldlam];
    }
}

__global__ void copy_upper_kernel(const double *__restrict__ A, int64_t lda, int q, double *__restrict__ R) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= q * q) return;
    const int i = idx % q, j = idx / q;
    R[idx] = (i <= j) ? A[i + (int64_t)j * lda] : 0.0;
}

__global__ void trace_kernel(const double *__restrict__ G, int q, double *__restrict__ out) {
    // single block, deterministic
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < q; i += 256) s += G[i + (int64_t)i * q];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void max_abs_kernel(const double *__restrict__ v, int64_t cnt, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < cnt; i += 256) s = fmax(s, fabs(v[i]));
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// out[0] = sum (a-b)^2, out[1] = sum b^2 over cnt entries (single block, deterministic)
__global__ void residual_kernel(const double *__restrict__ a, const double *__restrict__ b, int64_t cnt,
                                double *__restrict__ out) {
    __shared__ double r0[256], r1[256];
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = threadIdx.x; i < cnt; i += 256) {
        const double df = a[i] - b[i];
        s0 = fma(df, df, s0);
        s1 = fma(b[i], b[i], s1);
    }
    r0[threadIdx.x] = s0;
    r1[threadIdx.x] = s1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            r0[threadIdx.x] += r0[threadIdx.x + w];
            r1[threadIdx.x] += r1[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = r0[0];
        out[1] = r1[0];
    }
}

static inline unsigned nblk(int64_t cnt) { return (unsigned)((cnt + 255) / 256); }

void destroy_model(mrbf_ctx *ctx, mrbf_model *M) {
    if (!M) return;
    if (ctx && M->block) --ctx->live_models;
    if (M->block) {
        size_t pooled = 0;
        if (ctx)
            for (auto &b : ctx->model_pool) pooled += b.bytes;
        if (ctx && ctx->model_pool.size() < 8 && pooled + M->block_bytes <= (size_t(8) << 30)) {
            Buf b;
            b.p = M->block;
            b.bytes = M->block_bytes;
            ctx->model_pool.push_back(b);
        } else {
            (void)hipFree(M->block);
        }
    }
    delete M;
}

int build_model_shell(mrbf_ctx *ctx, int64_t n, int d, int k, const double *Cdev, int kid, double a, double b, int deg,
                      mrbf_model **out) {
    mrbf_model *M = new mrbf_model();
    M->n = n;
    M->d = d;
    M->k = k;
    M->deg = deg;
    M->q = poly_dim(d, deg);
    M->npad = round_up(n, 128);
    // the fused evaluation kernels are instantiated for row strides 64, 128 and 256; wider problems keep the minimal padding
    M->dpad = (d <= 64) ? 64 : (d <= 128 ? 128 : (d <= 256 ? 256 : (int)round_up(d, 16)));
    M->kp = make_kp(kid, a, b);
    // one block, carved at 256-byte granularity
    const size_t cnt[7] = {(size_t)n * d, (size_t)M->npad * M->dpad, (size_t)M->npad, (size_t)M->dpad, (size_t)n * k,
                           (size_t)M->npad * k, (size_t)std::max(M->q, 1) * k};
    size_t off[8];
    off[0] = 0;
    for (int i = 0; i < 7; ++i) off[i + 1] = off[i] + ((std::max<size_t>(cnt[i], 2) * sizeof(double) + 255) & ~size_t(255));
    const size_t total = off[7];
    // reuse a released block when one fits without wasting more than half of it
    int best = -1;
    for (int i = 0; i < (int)ctx->model_pool.size(); ++i)
        if (ctx->model_pool[i].bytes >= total && ctx->model_pool[i].bytes <= 2 * total &&
            (best < 0 || ctx->model_pool[i].bytes < ctx->model_pool[best].bytes))
            best = i;
    if (best >= 0) {
        M->block = ctx->model_pool[best].p;
        M->block_bytes = ctx->model_pool[best].bytes;
        ctx->model_pool.erase(ctx->model_pool.begin() + best);
    } else {
        hipError_t e = hipMalloc(&M->block, total);
        if (e != hipSuccess) {
            // drop the pool and retry once before giving up
            for (auto &b : ctx->model_pool) (void)hipFree(b.p);
            ctx->model_pool.clear();
            e = hipMalloc(&M->block, total);
        }
        if (e != hipSuccess) {
            delete M;
            return fail(ctx, MRBF_ENOMEM, "model allocation failed: %s", hipGetErrorString(e));
        }
        M->block_bytes = total;
    }
    ++ctx->live_models;  // (MRBF_OPT_LIVE_HANDLES: models and round-4 states created through this context and not yet released)
    char *base = (char *)M->block;
    M->C = (double *)(base + off[0]);
    M->Xc = (double *)(base + off[1]);
    M->sq = (double *)(base + off[2]);
    M->mean = (double *)(base + off[3]);
    M->W = (double *)(base + off[4]);
    M->Wc = (double *)(base + off[5]);
    M->lam = (double *)(base + off[6]);
    auto init = [&]() -> int {
        if (Cdev != M->C)
            MRBF_HIP(ctx, hipMemcpyAsync(M->C, Cdev, (size_t)n * d * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        MRBF_TRY(launch_center_pad(ctx, M->C, n, d, nullptr, M->mean, M->Xc, M->npad, M->dpad, M->sq));
        return 0;
    };
    const int rc = init();
    if (rc != 0) {
        destroy_model(ctx, M);
        return rc;
    }
    *out = M;
    return 0;
}

static int read_info(mrbf_ctx *ctx, const int *dinfo, int *hinfo) {
    MRBF_HIP(ctx, hipMemcpyAsync(hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- LU on the saddle system ---------------------------------------------------------------------
static int fit_lu(mrbf_ctx *ctx, mrbf_model *M, const double *Y, mrbf_fit_info *info) {
    const int64_t n = M->n, N = n + M->q;
    const int k = M->k, q = M->q;
    double *S, *B, *Pi = nullptr;
    int *ipiv, *dinfo;
    MRBF_TRY(get_buf(ctx, S_PHI, (size_t)N * N, &S));
    MRBF_TRY(get_buf(ctx, S_RHS, (size_t)N * k, &B));
    MRBF_TRY(get_buf(ctx, S_IPIV, (size_t)N, &ipiv));
    MRBF_TRY(get_buf(ctx, S_INFO, (size_t)4, &dinfo));
    hipLaunchKernelGGL(rhs_from_values_kernel, dim3(nblk(N * k)), dim3(256), 0, ctx->stream, Y, n, k, B, N);
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    MRBF_TRY(launch_gram(ctx, ctx->gram_mode, M->C, M->Xc, M->sq, n, M->npad, M->d, M->dpad, M->kp, S, N));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (q > 0) {
        MRBF_TRY(get_buf(ctx, S_PI, (size_t)n * q, &Pi));
        MRBF_TRY(launch_poly_matrix(ctx, M->C, n, M->d, q, Pi, n));
        hipLaunchKernelGGL(fill_saddle_kernel, dim3(nblk(n * q + (int64_t)q * q)), dim3(256), 0, ctx->stream, Pi, n, q, S, N);
    }
    MRBF_HIP(ctx, hipGetLastError());
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    MRBF_BLAS(ctx, rocsolver_dgetrf(ctx->blas, (int)N, (int)N, S, (int)N, ipiv, dinfo));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    int hinfo = 0;
    MRBF_TRY(read_info(ctx, dinfo, &hinfo));
    info->factor_info = hinfo;
    info->path = MRBF_PATH_LU;
    if (hinfo != 0) return fail(ctx, MRBF_ESINGULAR, "saddle matrix is singular (getrf info = %d, N = %lld)", hinfo, (long long)N);
    MRBF_BLAS(ctx, rocsolver_dgetrs(ctx->blas, rocblas_operation_none, (int)N, k, S, (int)N, ipiv, B, (int)N));
    hipLaunchKernelGGL(scatter_solution_kernel, dim3(nblk(M->npad * k + (int64_t)q * k)), dim3(256), 0, ctx->stream, B, N, n,
                       M->npad, k, q, n, M->W, M->Wc, M->lam, B, N);
    MRBF_HIP(ctx, hipGetLastError());
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[3]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_gram, ctx->ev[0], ctx->ev[1]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_project, ctx->ev[1], ctx->ev[5]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_factor, ctx->ev[5], ctx->ev[2]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_solve, ctx->ev[2], ctx->ev[3]));
    return 0;
}

// ---- minimum-norm solve of an under-determined saddle system (n < q: fewer sites than tail terms) ----
// x = V S^+ U' b by rocSOLVER's SVD; singular values below N eps s_max are dropped.  Tiny systems only
// (N = n + q <= 2q).  This is the case test/rbf_models.jl:35-44 builds ("too few points", max_evals = 1).
__global__ void scale_rows_pinv_kernel(double *__restrict__ T, int N, int k, const double *__restrict__ sv) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * k) return;
    const int i = idx % N;
    const double s = sv[i], cut = (double)N * 2.220446049250313e-16 * sv[0];
    T[idx] = (s > cut) ? T[idx] / s : 0.0;
}

static int fit_minnorm(mrbf_ctx *ctx, mrbf_model *M, const double *Y, mrbf_fit_info *info) {
    const int64_t n = M->n, N = n + M->q;
    const int k = M->k, q = M->q;
    const double one = 1.0, zero = 0.0;
    double *S, *B, *Pi, *U, *VT, *sv, *E, *T;
    int *dinfo;
    MRBF_TRY(get_buf(ctx, S_PHI, (size_t)N * N, &S));
    MRBF_TRY(get_buf(ctx, S_RHS, (size_t)N * k, &B));
    MRBF_TRY(get_buf(ctx, S_PI, (size_t)n * q, &Pi));
    MRBF_TRY(get_buf(ctx, S_Q1, (size_t)N * N, &U));
    MRBF_TRY(get_buf(ctx, S_W1, (size_t)N * N, &VT));
    MRBF_TRY(get_buf(ctx, S_TAU, (size_t)2 * N, &sv));
    MRBF_TRY(get_buf(ctx, S_T1, (size_t)N * k, &T));
    MRBF_TRY(get_buf(ctx, S_INFO, (size_t)4, &dinfo));
    E = sv + N;
    hipLaunchKernelGGL(rhs_from_values_kernel, dim3(nblk(N * k)), dim3(256), 0, ctx->stream, Y, n, k, B, N);
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    MRBF_TRY(launch_gram(ctx, ctx->gram_mode, M->C, M->Xc, M->sq, n, M->npad, M->d, M->dpad, M->kp, S, N));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MRBF_TRY(launch_poly_matrix(ctx, M->C, n, M->d, q, Pi, n));
    hipLaunchKernelGGL(fill_saddle_kernel, dim3(nblk(n * q + (int64_t)q * q)), dim3(256), 0, ctx->stream, Pi, n, q, S, N);
    MRBF_HIP(ctx, hipGetLastError());
    MRBF_BLAS(ctx, rocsolver_dgesvd(ctx->blas, rocblas_svect_all, rocblas_svect_all, (int)N, (int)N, S, (int)N, sv, U, (int)N, VT,
                                    (int)N, E, rocblas_outofplace, dinfo));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    int hinfo = 0;
    MRBF_TRY(read_info(ctx, dinfo, &hinfo));
    info->factor_info = hinfo;
    info->path = MRBF_PATH_MINNORM;
    if (hinfo != 0) return fail(ctx, MRBF_ESINGULAR, "SVD of the saddle matrix did not converge (info = %d)", hinfo);
    MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_transpose, rocblas_operation_none, (int)N, k, (int)N, &one, U, (int)N,
                                 B, (int)N, &zero, T, (int)N));
    hipLaunchKernelGGL(scale_rows_pinv_kernel, dim3(nblk(N * k)), dim3(256), 0, ctx->stream, T, (int)N, k, sv);
    MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_transpose, rocblas_operation_none, (int)N, k, (int)N, &one, VT, (int)N,
                                 T, (int)N, &zero, B, (int)N));
    hipLaunchKernelGGL(scatter_solution_kernel, dim3(nblk(M->npad * k + (int64_t)q * k)), dim3(256), 0, ctx->stream, B, N, n,
                       M->npad, k, q, n, M->W, M->Wc, M->lam, B, N);
    MRBF_HIP(ctx, hipGetLastError());
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[3]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_gram, ctx->ev[0], ctx->ev[1]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_factor, ctx->ev[1], ctx->ev[2]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_solve, ctx->ev[2], ctx->ev[3]));
    return 0;
}

// ---- Cholesky paths ------------------------------------------------------------------------------
int potrf_blocked_tall(mrbf_ctx *ctx, int64_t ncols, int64_t mrows, double *A, int64_t lda, int *dinfo, double *linv_all);
int backsolve_blocked(mrbf_ctx *ctx, int64_t npad, const double *L, int64_t lda, const double *linv_all, double *Y, int64_t ldy, int k);
int backsolve_persistent(mrbf_ctx *ctx, int64_t npad, const double *L, int64_t lda, const double *linv_all, double *Y, int64_t ldy, int k,
                         int *status);  // backsolve.hip
int launch_update_lower(mrbf_ctx *ctx, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t nt, int K,
                        const double *cvec, double cs, int64_t cn);
int launch_pad_identity(mrbf_ctx *ctx, double *A, int64_t n, int64_t npad, int64_t ld);
int tsmm_tn(mrbf_ctx *ctx, int64_t n, int p, int r, double alpha, const double *A, int64_t lda, const double *B, int64_t ldb, double beta,
            double *C, int64_t ldc);  // skinny.hip
int symm_panel(mrbf_ctx *ctx, int64_t n, int64_t npad, int q, const double *Phi, int64_t ld, const double *Q, int64_t ldq, double *W,
               int64_t ldwo, bool const_first, double cval);  // skinny.hip

// extra row tile(s) of the extended matrix: row npad + l = right-hand side l (a row), zero beyond k
__global__ void set_rhs_rows_kernel(double *__restrict__ A, int64_t ld, int64_t npad, int xt, const double *__restrict__ B, int k) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npad * xt) return;
    const int l = (int)(idx % xt);
    const int64_t i = idx / xt;
    A[(npad + l) + i * ld] = (l < k) ? B[i + (int64_t)l * npad] : 0.0;
}
__global__ void get_rhs_rows_kernel(const double *__restrict__ A, int64_t ld, int64_t npad, double *__restrict__ B, int k) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npad * k) return;
    const int l = (int)(idx % k);
    const int64_t i = idx / k;
    B[i + (int64_t)l * npad] = A[(npad + l) + i * ld];
}
// Tall[(dq + i) + t*lt] = Xc[i*dpad + t]  (row-major centred coordinates -> column-major block below the Gram matrix)
__global__ void xc_to_tall_kernel(const double *__restrict__ Xc, int64_t n, int d, int dpad, double *__restrict__ Tall, int64_t lt,
                                  int64_t dq) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    const int64_t i = idx / d;
    const int t = (int)(idx % d);
    Tall[(dq + i) + (int64_t)t * lt] = Xc[i * dpad + t];
}
// Q1 = [1/sqrt(n) | Qx]  (npad x q, ld npad, rows >= n zero)
__global__ void build_q1_kernel(const double *__restrict__ Tall, int64_t lt, int64_t dq, int64_t n, int64_t npad, int q,
                                double *__restrict__ Q1) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npad * q) return;
    const int64_t i = idx % npad;
    const int t = (int)(idx / npad);
    double v = 0.0;
    if (i < n) v = (t == 0) ? 1.0 / sqrt((double)n) : Tall[(dq + i) + (int64_t)(t - 1) * lt];
    Q1[idx] = v;
}
// scal[0] = trace(Q1' Phi Q1)  ->  scal[1] = mu = (n phi0 - trace) / (n - q), the mean eigenvalue of Z' Phi Z;
// flags[2] = 1 when mu is not positive (Z' Phi Z cannot be positive definite).  Kept on the device: no host round trip.
__global__ void shift_from_trace_kernel(double *__restrict__ scal, double nphi0, double nmq, int *__restrict__ flags) {
    const double mu = (nphi0 - scal[0]) / nmq;
    const bool ok = mu > 0.0 && mu < 1e300;
    scal[1] = ok ? mu : 1.0;
    flags[2] = ok ? 0 : 1;
}

// PA = [Q1 | V | 0], PB = [V | Q1 | 0] with V = W - (mu/2) Q1   (npad x K2, ld npad)
// Panels of the rank-2q update K = Phi - Q1 V' - V Q1',  V = W - (mu/2) Q1:  PA = [Q | V], PB = [V | Q] (K2 columns, zero padded).
// skip = 1: the first column of Q1 is the constant 1/sqrt(n); its rank-2 term goes through the update kernel's epilogue (v0 = the
// first column of V), the panels hold columns 1 .. q-1 only -- with q = 65 that is K2 = 128 instead of 144.
__global__ void build_panels_kernel(const double *__restrict__ Q1, const double *__restrict__ W, const double *__restrict__ scal,
                                    int64_t n, int64_t npad, int q, int K2, int skip, double *__restrict__ PA, double *__restrict__ PB,
                                    double *__restrict__ v0) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npad * K2) return;
    const double mu = scal[1];
    const int64_t i = idx % npad;
    const int t = (int)(idx / npad);
    const int qq = q - skip;
    double a = 0.0, b = 0.0;
    if (i < n && t < 2 * qq) {
        const int tt = t % qq + skip;
        const double qv = Q1[i + (int64_t)tt * npad];
        const double vv = fma(-0.5 * mu, qv, W[i + (int64_t)tt * npad]);
        a = (t < qq) ? qv : vv;
        b = (t < qq) ? vv : qv;
    }
    PA[idx] = a;
    PB[idx] = b;
    if (skip && t == 0) v0[i] = (i < n) ? fma(-0.5 * mu, Q1[i], W[i]) : 0.0;
}
// lam[0] = z0 / sqrt(n) - mean . lam[1:]   (Pi = Q1 R with R = [[sqrt n, sqrt n mean'], [0, Lx']])
// one workgroup of 64 threads per output: the dot product with the mean is spread over the lanes and summed in a fixed order (a
// serial q-term loop of one thread took 8-27 us)
__global__ __launch_bounds__(64) void finish_lambda_kernel(double *__restrict__ T1, int q, int k, double sqrtn, const double *__restrict__ mean) {
    __shared__ double part[64];
    const int l = blockIdx.x, tid = threadIdx.x;
    if (l >= k) return;
    double acc = 0.0;
    for (int t = 1 + tid; t < q; t += 64) acc = fma(mean[t - 1], T1[t + (int64_t)l * q], acc);
    part[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        double dsum = 0.0;
        for (int u = 0; u < 64; ++u) dsum += part[u];
        T1[(int64_t)l * q] = T1[(int64_t)l * q] / sqrtn - dsum;
    }
}

// B(n x k) -= Q(n x q) T(q x k)  (all column-major): the rank-q correction of the weights.  Workgroup = 64 rows x 4 groups of
// columns t (t = g, g + 4, ...), the four partial sums added in a fixed order (one thread per entry with a serial q-term loop was
// bound by q dependent load latencies: 17-49 us)
__global__ __launch_bounds__(256) void sub_qt_kernel(double *__restrict__ B, int64_t ldb, const double *__restrict__ Q, int64_t ldq,
                                                     const double *__restrict__ T, int64_t n, int q, int k) {
    __shared__ double part[4][64];
    const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 64 + r;
    for (int l = 0; l < k; ++l) {
        double s0 = 0.0, s1 = 0.0;
        if (row < n) {
            int t = g;
            for (; t + 4 < q; t += 8) {
                s0 = fma(Q[row + (int64_t)t * ldq], T[t + (int64_t)l * q], s0);
                s1 = fma(Q[row + (int64_t)(t + 4) * ldq], T[t + 4 + (int64_t)l * q], s1);
            }
            if (t < q) s0 = fma(Q[row + (int64_t)t * ldq], T[t + (int64_t)l * q], s0);
        }
        __syncthreads();
        part[g][r] = s0 + s1;
        __syncthreads();
        if (g == 0 && row < n) B[row + (int64_t)l * ldb] -= (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
    }
}

// x = L^-T b through the stored inverse of the factor (d <= 128: one diagonal block, whose inverse the factorisation produced
// anyway): X[j] = sum_{i >= j} inv[i][j] b[i] -- one pass instead of d dependent steps.  inv: 128 x 128 column-major, zero above.
__global__ __launch_bounds__(256) void apply_linv_t_kernel(const double *__restrict__ inv, int d, double *__restrict__ X, int64_t ldx, int k) {
    __shared__ double bs[128];
    const int j = threadIdx.x & 127, half = threadIdx.x >> 7;  // two threads per unknown: rows i = j + half, j + half + 2, ...
    __shared__ double part[2][128];
    for (int l = 0; l < k; ++l) {
        __syncthreads();
        if (threadIdx.x < 128) bs[threadIdx.x] = threadIdx.x < d ? X[threadIdx.x + (int64_t)l * ldx] : 0.0;
        __syncthreads();
        double v = 0.0;
        if (j < d) {
            const double *col = inv + (int64_t)j * 128;  // column j of the inverse: rows i >= j
            for (int i = j + half; i < d; i += 2) v = fma(col[i], bs[i], v);
        }
        part[half][j] = v;
        __syncthreads();
        if (half == 0 && j < d) X[j + (int64_t)l * ldx] = part[0][j] + part[1][j];
    }
}

// x = L^-T b for the d x d lower factor of the Cholesky-QR (d <= 256) and k right-hand sides, in place in X (ld ldx): one
// workgroup, thread j owns unknown j, back substitution over the columns of L' (64 us of rocBLAS small-trsm launch otherwise)
__global__ __launch_bounds__(256) void trsm_lt_small_kernel(const double *__restrict__ L, int64_t ldl, int d, double *__restrict__ X,
                                                            int64_t ldx, int k) {
    __shared__ double xs[256];
    const int j = threadIdx.x;
    for (int l = 0; l < k; ++l) {
        double v = j < d ? X[j + (int64_t)l * ldx] : 0.0;
        for (int c = d - 1; c >= 0; --c) {
            if (j == c) {
                v /= L[c + (int64_t)c * ldl];
                xs[c] = v;
            }
            __syncthreads();
            if (j < c) v -= L[c + (int64_t)j * ldl] * xs[c];  // (L')(j, c) = L(c, j)
            __syncthreads();
        }
        if (j < d) X[j + (int64_t)l * ldx] = v;
    }
}

// ---- behind the backward substitution: re-projection of the weights, tail coefficients and the scatter into the model in three launches
// (were seven: copy, Q1'w, w -= Q1 (Q1'w), W'w, inv(Lx)' z, finish, scatter -- each 4-15 us behind the other).  The two reductions
// over the sites are taken from the same w:  W'(w - Q1 T2) = W'w - (Q1'W)' T2  with  Q1'W = cg G  (cg = 1 when W = Phi Q1, 1/2 when
// W = Phi Q1 - Q1 G / 2), so that z = T1 - W'w + cg G' T2 needs no second pass (T2 = Q1'w is at rounding level: w solves a system
// whose right-hand side and matrix are projected).
//   tail_dots_kernel    per 64 sites: partial Q1'w and W'w                                   (npad / 64 workgroups)
//   tail_coeff_kernel   fixed-order sums, z, lam = inv(Lx)' z (tail of degree 1), constant term  (one workgroup)
//   tail_apply_kernel   w -= Q1 T2, W / Wc / lam of the model                                  (npad / 64 workgroups)
__global__ __launch_bounds__(256) void tail_dots_kernel(const double *__restrict__ Bw, int64_t ldb, const double *__restrict__ Q1,
                                                        const double *__restrict__ Wm, int64_t ldq, int64_t n, int q, int k,
                                                        double *__restrict__ part) {
    __shared__ double ws[64 * 16];
    __shared__ double red[4][2][64];
    const int tid = threadIdx.x, g = tid >> 6, lane = tid & 63;
    const int64_t R0 = (int64_t)blockIdx.x * 64;
    for (int e = tid; e < 64 * k; e += 256) {
        const int i = e & 63, l = e >> 6;
        ws[e] = R0 + i < n ? Bw[R0 + i + (int64_t)l * ldb] : 0.0;
    }
    __syncthreads();
    // thread (a, g): column a (< q <= 129 -> up to three passes of 64 columns), rows 16 g .. 16 g + 15; the four row groups are added in order
    for (int a0 = 0; a0 < q; a0 += 64) {
        const int a = a0 + lane;
        double qv[16], wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t r = R0 + 16 * g + u;
            const bool ok = a < q && r < n;
            qv[u] = ok ? Q1[r + (int64_t)a * ldq] : 0.0;
            wv[u] = ok ? Wm[r + (int64_t)a * ldq] : 0.0;
        }
        for (int l = 0; l < k; ++l) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double w = ws[l * 64 + 16 * g + u];
                s0 = fma(qv[u], w, s0);
                s1 = fma(wv[u], w, s1);
            }
            __syncthreads();
            red[g][0][lane] = s0;
            red[g][1][lane] = s1;
            __syncthreads();
            if (g == 0 && a < q) {
                double *o = part + ((size_t)blockIdx.x * 2 * k + l) * q + a;
                o[0] = (red[0][0][lane] + red[1][0][lane]) + (red[2][0][lane] + red[3][0][lane]);
                o[(size_t)k * q] = (red[0][1][lane] + red[1][1][lane]) + (red[2][1][lane] + red[3][1][lane]);
            }
        }
    }
}

__global__ __launch_bounds__(1024) void tail_coeff_kernel(const double *__restrict__ part, int nch, int q, int k, int d, const double *__restrict__ T1,
                                                         const double *__restrict__ G, double cg, const double *__restrict__ LxInv,
                                                         double sqrtn, const double *__restrict__ mean, double *__restrict__ T2,
                                                         double *__restrict__ lam_out) {
    __shared__ double t2[129 * 16], zz[129 * 16], xx[129 * 16];
    const int tid = threadIdx.x, nthr = blockDim.x;  // (1024 threads: 2 k q sums of up to 256 partials each, one per thread at k = 2, q = 65)
    // fixed-order sums over the chunks, 32 loads in flight
    for (int e = tid; e < 2 * k * q; e += nthr) {
        double sv = 0.0;
        for (int c0 = 0; c0 < nch; c0 += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = c0 + u < nch ? part[(size_t)(c0 + u) * 2 * k * q + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u) sv += v[u];
        }
        if (e < k * q)
            t2[e] = sv;  // [l][a]
        else
            zz[e - k * q] = sv;
    }
    __syncthreads();
    for (int e = tid; e < k * q; e += nthr) {
        const int l = e / q, a = e % q;
        double corr = 0.0;
        for (int b0 = 0; b0 < q; b0 += 16) {  // sixteen loads of the column in flight
            double gv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) gv[u] = b0 + u < q ? G[b0 + u + (int64_t)a * q] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) corr = fma(gv[u], b0 + u < q ? t2[l * q + b0 + u] : 0.0, corr);
        }
        zz[e] = T1[a + (int64_t)l * q] - zz[e] + cg * corr;
        T2[a + (int64_t)l * q] = t2[e];
    }
    __syncthreads();
    // lam_{1..d} = inv(Lx)' z_{1..d}:  x_j = sum_{i >= j} inv(Lx)(i, j) z_i  -- thread (j, l), sixteen loads of column j in flight
    for (int e = tid; e < k * d; e += nthr) {
        const int l = e / d, j = e % d;
        double acc = 0.0;
        for (int i0 = j; i0 < d; i0 += 16) {
            double iv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) iv[u] = i0 + u < d ? LxInv[i0 + u + (int64_t)j * 128] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = fma(iv[u], i0 + u < d ? zz[l * q + 1 + i0 + u] : 0.0, acc);
        }
        xx[l * q + 1 + j] = acc;
    }
    __syncthreads();
    // constant term: lam_0 = z_0 / sqrt n - sum_t mean_t lam_t   (see finish_lambda_kernel)
    if (tid < k) {
        const int l = tid;
        double dsum = 0.0;
        for (int t0 = 1; t0 < q; t0 += 16) {
            double mv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) mv[u] = t0 + u < q ? mean[t0 + u - 1] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) dsum = fma(mv[u], t0 + u < q ? xx[l * q + t0 + u] : 0.0, dsum);
        }
        xx[l * q] = zz[l * q] / sqrtn - dsum;
    }
    __syncthreads();
    for (int e = tid; e < k * q; e += nthr) lam_out[(e % q) + (int64_t)(e / q) * q] = xx[e];  // q x k column-major like T1w
}

__global__ __launch_bounds__(256) void tail_apply_kernel(const double *__restrict__ Bw, int64_t ldb, const double *__restrict__ Q1, int64_t ldq,
                                                         const double *__restrict__ T2, const double *__restrict__ lamsrc, int64_t n,
                                                         int64_t npad, int q, int k, double *__restrict__ W, double *__restrict__ Wc,
                                                         double *__restrict__ lam) {
    __shared__ double part4[4][64];
    const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
    const int64_t row = (int64_t)blockIdx.x * 64 + r;
    for (int l = 0; l < k; ++l) {
        double s0 = 0.0, s1 = 0.0;
        if (row < n) {
            int t = g;
            for (; t + 4 < q; t += 8) {
                s0 = fma(Q1[row + (int64_t)t * ldq], T2[t + (int64_t)l * q], s0);
                s1 = fma(Q1[row + (int64_t)(t + 4) * ldq], T2[t + 4 + (int64_t)l * q], s1);
            }
            if (t < q) s0 = fma(Q1[row + (int64_t)t * ldq], T2[t + (int64_t)l * q], s0);
        }
        __syncthreads();
        part4[g][r] = s0 + s1;
        __syncthreads();
        if (g == 0 && row < npad) {
            const double v = row < n ? Bw[row + (int64_t)l * ldb] - ((part4[0][r] + part4[1][r]) + (part4[2][r] + part4[3][r])) : 0.0;
            Wc[row + (int64_t)l * npad] = v;
            if (row < n) W[row * k + l] = v;
        }
    }
    if (blockIdx.x == 0)
        for (int e = tid; e < q * k; e += 256) lam[(int64_t)(e % q) * k + e / q] = lamsrc[(e % q) + (int64_t)(e / q) * q];
}

// returns 0 with *not_pd = 1 when a factorisation met a non-positive pivot (caller may retry with LU), and 0 with *gave_up = 1
// when the persistent factorisation abandoned a dependency (caller re-runs this function with the host-driven factorisation)
static int fit_chol(mrbf_ctx *ctx, mrbf_model *M, const double *Y, mrbf_fit_info *info, int *not_pd, int *gave_up) {
    const int64_t n = M->n, npad = M->npad;
    const int k = M->k, q = M->q, d = M->d;
    const double one = 1.0, mhalf = -0.5;
    *not_pd = 0;
    *gave_up = 0;
    const bool builtin = ctx->chol_impl != 1;
    const int xt = builtin ? (int)round_up(k, 128) : 0;  // extra row tiles: right-hand sides ride along the factorisation
    const int64_t ld = npad + xt;
    double *Phi, *B;
    int *dinfo;
    MRBF_TRY(get_buf(ctx, S_PHI, (size_t)ld * npad, &Phi));
    MRBF_TRY(get_buf(ctx, S_RHS, (size_t)npad * k, &B));
    MRBF_TRY(get_buf(ctx, S_INFO, (size_t)4, &dinfo));  // [0] main factorisation, [1] Cholesky-QR of the tail, [2] shift not positive, [3] backward substitution gave up
    MRBF_HIP(ctx, hipMemsetAsync(dinfo, 0, 4 * sizeof(int), ctx->stream));
    // d <= 64: the tail basis, T1 = Q1' Y, B = P Y and the extra rows in three launches (small.hip, TailQ) instead of twelve
    const bool tailq = q > 1 && tail_basis_applies(M) && M->k == k;
    if (!tailq) hipLaunchKernelGGL(rhs_from_values_kernel, dim3(nblk(npad * k)), dim3(256), 0, ctx->stream, Y, n, k, B, npad);
    double *Q1 = nullptr, *Wm = nullptr, *G = nullptr, *T1 = nullptr, *scal = nullptr, *Tall = nullptr, *LxInv = nullptr;
    int64_t lt = 0, dq = 0;
    info->mu = 0.0;
    // The orthonormal basis Q1 of the polynomial tail depends on the centred coordinates only, not on Phi: its chain of small
    // latency-bound kernels (Xc'Xc, 128 x 128 Cholesky, panel solve: ~150 us) runs on a side stream UNDER the Gram kernel.
    hipStream_t main_stream = ctx->stream;
    const bool side = q > 0 && ctx->panel_stream && ctx->panel_stream != main_stream && n >= 2048;
    // d <= 64: the fit's own assembly writes the lower triangle only and accumulates Phi Xc on the way (gram_fused.hip)
    const bool fused = q > 0 && gram_w_applies(ctx, M) && (q == 1 || round_up(d, 128) == 128);
    if (q > 0) {
        MRBF_TRY(get_buf(ctx, S_Q1, (size_t)npad * q, &Q1));
        MRBF_TRY(get_buf(ctx, S_W1, (size_t)npad * q, &Wm));
        MRBF_TRY(get_buf(ctx, S_G, (size_t)q * q, &G));
        MRBF_TRY(get_buf(ctx, S_T1, (size_t)q * k, &T1));
        MRBF_TRY(get_buf(ctx, S_MISC, (size_t)8, &scal));
        if (side) {
            MRBF_HIP(ctx, hipEventRecord(ctx->evx[0], main_stream));
            MRBF_HIP(ctx, hipStreamWaitEvent(ctx->panel_stream, ctx->evx[0], 0));
            ctx->stream = ctx->panel_stream;
        }
        struct RestoreStream {
            mrbf_ctx *c;
            hipStream_t s;
            ~RestoreStream() { c->stream = s; }
        } restore{ctx, main_stream};
        if (tailq) {
            double *scratch;
            MRBF_TRY(get_buf(ctx, S_PI, tail_basis_scratch_doubles(), &scratch));
            MRBF_TRY(get_buf(ctx, S_QR_INV, (size_t)128 * 128, &LxInv));
            MRBF_TRY(launch_tail_basis(ctx, M, Y, scratch, LxInv, T1, Q1, B, builtin ? Phi : nullptr, ld, xt, dinfo));
        } else if (q > 1) {
            // Cholesky-QR of the centred coordinates: [Xc'Xc ; Xc] -> [Lx ; Xc Lx^-T] by the tall blocked Cholesky
            dq = round_up(d, 128);
            lt = dq + npad;
            MRBF_TRY(get_buf(ctx, S_PI, (size_t)lt * dq, &Tall));
            MRBF_HIP(ctx, hipMemsetAsync(Tall, 0, (size_t)lt * dq * sizeof(double), ctx->stream));
            hipLaunchKernelGGL(xc_to_tall_kernel, dim3(nblk(n * d)), dim3(256), 0, ctx->stream, M->Xc, n, d, M->dpad, Tall, lt, dq);
            MRBF_TRY(tsmm_tn(ctx, n, d, d, 1.0, Tall + dq, lt, Tall + dq, lt, 0.0, Tall, lt));  // Gx = Xc' Xc
            MRBF_TRY(launch_pad_identity(ctx, Tall, d, dq, lt));
            // (one diagonal block: its inverse is kept for the tail coefficients of the solve)
            if (dq == 128) MRBF_TRY(get_buf(ctx, S_QR_INV, (size_t)128 * 128, &LxInv));
            MRBF_TRY(potrf_blocked_tall(ctx, dq, lt, Tall, lt, dinfo + 1, LxInv));  // flag read back with the main one
        }
        if (!tailq) hipLaunchKernelGGL(build_q1_kernel, dim3(nblk(npad * q)), dim3(256), 0, ctx->stream, Tall, lt, dq, n, npad, q, Q1);
        if (!fused) MRBF_HIP(ctx, hipMemsetAsync(Wm, 0, (size_t)npad * q * sizeof(double), ctx->stream));
        // B = P Y = Y - Q1 (Q1' Y) (T1 keeps Q1' Y for lam) and, for the built-in factorisation, the right-hand sides as extra rows
        // of the matrix: they depend on Q1 and Y only, so they too run under the Gram kernel
        if (!tailq) {
            MRBF_TRY(tsmm_tn(ctx, n, q, k, 1.0, Q1, npad, B, npad, 0.0, T1, q));
            hipLaunchKernelGGL(sub_qt_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, ctx->stream, B, npad, Q1, npad, T1, n, q, k);
            if (builtin) hipLaunchKernelGGL(set_rhs_rows_kernel, dim3(nblk(npad * xt)), dim3(256), 0, ctx->stream, Phi, ld, npad, xt, B, k);
        }
        if (side) MRBF_HIP(ctx, hipEventRecord(ctx->evx[1], ctx->stream));
    }
    ctx->stream = main_stream;
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));  // ev0..ev1 bracket the Gram kernel alone
    double *Wpart = nullptr, *rspart = nullptr;
    int nseg = 0;
    if (fused)
        MRBF_TRY(launch_gram_w(ctx, M, Phi, ld, &Wpart, &rspart, &nseg));  // lower triangle + partial panels of Phi Xc, row sums
    else
        MRBF_TRY(launch_gram(ctx, ctx->gram_mode, M->C, M->Xc, M->sq, n, M->npad, M->d, M->dpad, M->kp, Phi, ld));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (side) MRBF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evx[1], 0));
    if (fused) {
        // W1 = Phi Q1 from the partial panels, G = Q1' W1, mu, V = W1 - Q1 (G / 2 + mu / 2 I), panels: three small launches
        const int skip = (q > 1 && round_up(2 * (q - 1), 16) < round_up(2 * q, 16)) ? 1 : 0;
        const int K2 = (int)round_up(2 * (q - skip), 16);
        double *PA, *PB, *v0 = nullptr;
        MRBF_TRY(get_buf(ctx, S_STAGE_C, (size_t)npad * K2, &PA));
        MRBF_TRY(get_buf(ctx, S_STAGE_D, (size_t)npad * K2, &PB));
        if (skip) MRBF_TRY(get_buf(ctx, S_V0, (size_t)npad, &v0));
        MRBF_TRY(launch_projection_small(ctx, M, Wpart, rspart, nseg, LxInv, Q1, Wm, G, scal, dinfo, K2, skip, PA, PB, v0));
        MRBF_TRY(launch_update_lower(ctx, PA, npad, PB, npad, Phi, ld, npad / 128, K2, v0, 1.0 / std::sqrt((double)n), n));
    } else if (q > 0) {
        // W1 = Phi Q1 (Phi is stored in full) ; G = Q1' W1 ; W = W1 - 1/2 Q1 G
        MRBF_TRY(symm_panel(ctx, n, npad, q, Phi, ld, Q1, npad, Wm, npad, true, 1.0 / std::sqrt((double)n)));  // column 0 of Q1 is 1/sqrt(n)
        MRBF_TRY(tsmm_tn(ctx, n, q, q, 1.0, Q1, npad, Wm, npad, 0.0, G, q));
        hipLaunchKernelGGL(trace_kernel, dim3(1), dim3(256), 0, ctx->stream, G, q, scal);
        MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, (int)n, q, q, &mhalf, Q1, (int)npad,
                                     G, q, &one, Wm, (int)npad));
        // mu = trace(P Phi P) / (n - q): the mean eigenvalue of Z' Phi Z, so the shift sits inside the spectrum
        hipLaunchKernelGGL(shift_from_trace_kernel, dim3(1), dim3(1), 0, ctx->stream, scal, (double)n * M->kp.phi0,
                           (double)std::max<int64_t>(n - q, 1), dinfo);
        // K = Phi - Q1 V' - V Q1',  V = W - (mu/2) Q1  ( = P Phi P + mu Q1 Q1' ), lower tile pairs, in place, one kernel
        // (the constant first column of Q1 is applied in the update's epilogue when that saves a 16-column chunk)
        const int skip = (q > 1 && round_up(2 * (q - 1), 16) < round_up(2 * q, 16)) ? 1 : 0;
        const int K2 = (int)round_up(2 * (q - skip), 16);
        double *PA, *PB, *v0 = nullptr;
        MRBF_TRY(get_buf(ctx, S_STAGE_C, (size_t)npad * K2, &PA));
        MRBF_TRY(get_buf(ctx, S_STAGE_D, (size_t)npad * K2, &PB));
        if (skip) MRBF_TRY(get_buf(ctx, S_V0, (size_t)npad, &v0));
        hipLaunchKernelGGL(build_panels_kernel, dim3(nblk(npad * K2)), dim3(256), 0, ctx->stream, Q1, Wm, scal, n, npad, q, K2, skip, PA, PB, v0);
        MRBF_TRY(launch_update_lower(ctx, PA, npad, PB, npad, Phi, ld, npad / 128, K2, v0, 1.0 / std::sqrt((double)n), n));
    }
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    int hinfo = 0;
    double *linv_all = nullptr;
    if (builtin) {
        MRBF_TRY(get_buf(ctx, S_CHOL_WS, (size_t)npad * 128, &linv_all));
        MRBF_TRY(launch_pad_identity(ctx, Phi, n, npad, ld));
        if (q == 0) hipLaunchKernelGGL(set_rhs_rows_kernel, dim3(nblk(npad * xt)), dim3(256), 0, ctx->stream, Phi, ld, npad, xt, B, k);
        ctx->mega_xreal = k;  // (rows k .. xt - 1 of the extra block row are zero: the persistent factorisation skips their half tiles)
        ctx->mega_info_clean = 1;  // (dinfo[0 .. 3] were zeroed at the top of this function; nothing has written dinfo[0] since)
        const int rc_potrf = potrf_blocked_tall(ctx, npad, npad + xt, Phi, ld, dinfo, linv_all);
        ctx->mega_xreal = 0;
        ctx->mega_info_clean = 0;
        if (rc_potrf != 0) return rc_potrf;
        MRBF_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    } else {
        int *dpot;
        MRBF_TRY(get_buf(ctx, S_IPIV, (size_t)4, &dpot));
        MRBF_BLAS(ctx, rocsolver_dpotrf(ctx->blas, rocblas_fill_lower, (int)n, Phi, (int)ld, dpot));
        MRBF_HIP(ctx, hipMemcpyAsync(dinfo, dpot, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        MRBF_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    }
    // everything after the factorisation, re-runnable: the right-hand sides are re-read from the extra rows, T1 is copied first
    double *T1w = nullptr;
    if (q > 0) MRBF_TRY(get_buf(ctx, S_T1W, (size_t)q * k, &T1w));
    auto solve_tail = [&](bool persistent) -> int {
        if (builtin) {
            // forward substitution came out of the factorisation (the extra rows); backward substitution with the stored block inverses
            hipLaunchKernelGGL(get_rhs_rows_kernel, dim3(nblk(npad * k)), dim3(256), 0, ctx->stream, Phi, ld, npad, B, k);
            if (persistent)
                MRBF_TRY(backsolve_persistent(ctx, npad, Phi, ld, linv_all, B, npad, k, dinfo + 3));
            else
                MRBF_TRY(backsolve_blocked(ctx, npad, Phi, ld, linv_all, B, npad, k));
        } else {
            MRBF_BLAS(ctx, rocsolver_dpotrs(ctx->blas, rocblas_fill_lower, (int)n, k, Phi, (int)ld, B, (int)npad));
        }
        static const int tail3 = mrbf_env("MRBF_TAIL3") ? atoi(mrbf_env("MRBF_TAIL3")) : 1;
        if (tail3 && q > 0 && q <= 129 && k <= 16 && (q == 1 || LxInv)) {
            // re-projection, tail coefficients and scatter in three launches (tail_dots / tail_coeff / tail_apply above)
            const int nch = (int)(npad / 64);
            double *T2, *part;
            MRBF_TRY(get_buf(ctx, S_T2, (size_t)q * k, &T2));
            MRBF_TRY(get_buf(ctx, S_TAIL_PART, (size_t)nch * 2 * k * q, &part));
            hipLaunchKernelGGL(tail_dots_kernel, dim3((unsigned)nch), dim3(256), 0, ctx->stream, B, npad, Q1, Wm, npad, n, q, k, part);
            hipLaunchKernelGGL(tail_coeff_kernel, dim3(1), dim3(1024), 0, ctx->stream, part, nch, q, k, q > 1 ? d : 0, T1, G, fused ? 1.0 : 0.5, LxInv,
                               std::sqrt((double)n), M->mean, T2, T1w);
            hipLaunchKernelGGL(tail_apply_kernel, dim3((unsigned)nch), dim3(256), 0, ctx->stream, B, npad, Q1, npad, T2, T1w, n, M->npad, q, k, M->W,
                               M->Wc, M->lam);
            MRBF_HIP(ctx, hipGetLastError());
            return 0;
        }
        if (q > 0) {
            double *T2;
            MRBF_TRY(get_buf(ctx, S_T2, (size_t)q * k, &T2));
            MRBF_HIP(ctx, hipMemcpyAsync(T1w, T1, (size_t)q * k * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            // re-project w (rounding hygiene): w -= Q1 (Q1' w)
            MRBF_TRY(tsmm_tn(ctx, n, q, k, 1.0, Q1, npad, B, npad, 0.0, T2, q));
            hipLaunchKernelGGL(sub_qt_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, ctx->stream, B, npad, Q1, npad, T2, n, q, k);
            // z = Q1' Y - (Phi Q1)' w;  (Phi Q1)' w = W' w because Q1' w = 0;  lam = R^-1 z
            MRBF_TRY(tsmm_tn(ctx, n, q, k, -1.0, Wm, npad, B, npad, 1.0, T1w, q));
            if (q > 1) {
                if (LxInv)
                    hipLaunchKernelGGL(apply_linv_t_kernel, dim3(1), dim3(256), 0, ctx->stream, LxInv, d, T1w + 1, (int64_t)q, k);
                else if (d <= 256)
                    hipLaunchKernelGGL(trsm_lt_small_kernel, dim3(1), dim3(256), 0, ctx->stream, Tall, lt, d, T1w + 1, (int64_t)q, k);
                else
                    MRBF_BLAS(ctx, rocblas_dtrsm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                                                 rocblas_diagonal_non_unit, d, k, &one, Tall, (int)lt, T1w + 1, q));
            }
            hipLaunchKernelGGL(finish_lambda_kernel, dim3((unsigned)k), dim3(64), 0, ctx->stream, T1w, q, k, std::sqrt((double)n), M->mean);
        }
        hipLaunchKernelGGL(scatter_solution_kernel, dim3(nblk(M->npad * k + (int64_t)q * k)), dim3(256), 0, ctx->stream, B, npad, n,
                           M->npad, k, q, (int64_t)0, M->W, M->Wc, M->lam, T1w ? T1w : B, (int64_t)q);
        MRBF_HIP(ctx, hipGetLastError());
        return 0;
    };
    MRBF_TRY(solve_tail(true));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    {
        // one host round trip for all the flags of this path, AFTER the solve has been enqueued: the solve kernels run on whatever
        // the factorisation left (they terminate on any input); if a flag is set their output is discarded and the caller falls
        // back to the LU path, which re-assembles Phi.  Reading the flags before the solve cost ~100 us of idle GPU per fit.
        // (flags, shift and the factorisation's device clock land in pinned memory: three asynchronous downloads, ONE wait -- through
        //  pageable memory every download was a round trip of its own, the clock's a synchronous one behind the others)
        int hflags_local[4] = {0, 0, 0, 0};
        double hscal_local[2] = {0.0, 0.0};
        int *hflags = ctx->hpin ? reinterpret_cast<int *>(ctx->hpin + HPIN_FIT_FLAGS) : hflags_local;
        double *hscal = ctx->hpin ? reinterpret_cast<double *>(ctx->hpin + HPIN_FIT_SCAL) : hscal_local;
        hscal[0] = hscal[1] = 0.0;
        static const int publish = mrbf_env("MRBF_FIT_PUBLISH") ? atoi(mrbf_env("MRBF_FIT_PUBLISH")) : 1;
        if (ctx->hpin && publish) {
            // (round 6: ONE tiny launch writes the three into the pinned words -- the block is mapped into the device's address space --
            //  instead of three blit launches with their gaps: 11 us per fit)
            unsigned long long *hstat = nullptr;
            const unsigned long long *dstat = nullptr;
            if (ctx->mega_stat_pending && ctx->mega_stat_dev) {
                ctx->hpin[HPIN_MEGA_STAT] = ~0ull;
                hstat = &ctx->hpin[HPIN_MEGA_STAT];
                dstat = ctx->mega_stat_dev;
                ctx->mega_stat_pending = 2;  // on its way (mega_stat_finish reads the pinned word)
            }
            hipLaunchKernelGGL(fit_publish_kernel, dim3(1), dim3(64), 0, ctx->stream, (const int *)dinfo, q > 0 ? (const double *)scal : nullptr, dstat,
                               hflags, hscal, hstat);
        } else {
        MRBF_HIP(ctx, hipMemcpyAsync(hflags, dinfo, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        if (q > 0) MRBF_HIP(ctx, hipMemcpyAsync(hscal, scal, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_TRY(mega_stat_enqueue(ctx));
        }
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MRBF_TRY(mega_stat_finish(ctx));
        info->ms_factor_device = ctx->last_device_ms;
        hinfo = hflags[0];
        if (hinfo < 0) {
            // the persistent factorisation gave up on a dependency (bounded wait): the matrix is half factored, the caller
            // re-assembles it and factors with the host-driven launches (same kernels, no inter-workgroup waits)
            *gave_up = 1;
            info->giveup_code = -hinfo;
            return 0;
        }
        info->factor_info = hinfo;
        if (q > 0) info->mu = hscal[1];
        if (hflags[1] < 0) {
            // the tail's Cholesky-QR ran through the persistent factorisation (d > 128) and THAT gave up on a dependency: not a rank
            // deficiency -- same remedy as for the main factorisation (host-driven launches), not the LU of the saddle system
            *gave_up = 1;
            info->giveup_code = -hflags[1];
            return 0;
        }
        if (hflags[1] != 0) {  // affinely dependent sites: Pi is rank deficient
            *not_pd = 1;
            info->factor_info = -2;
            return 0;
        }
        if (hflags[2] != 0) {  // trace <= 0: Z' Phi Z cannot be positive definite
            *not_pd = 1;
            info->factor_info = -1;
            return 0;
        }
        if (hinfo != 0) {
            *not_pd = 1;
            return 0;
        }
        if (hflags[3] != 0) {
            // the persistent backward substitution gave up: the factor is intact, redo the solve with one launch per block row
            info->fallbacks |= MRBF_FB_BACKSOLVE_BLOCKED;
            info->giveup_code = hflags[3];
            MRBF_TRY(solve_tail(false));
            MRBF_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
        }
    }
    MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[4]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_gram, ctx->ev[0], ctx->ev[1]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_project, ctx->ev[1], ctx->ev[2]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_factor, ctx->ev[2], ctx->ev[3]));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_solve, ctx->ev[3], ctx->ev[4]));
    info->path = q > 0 ? MRBF_PATH_PROJ_CHOL : MRBF_PATH_CHOL;
    return 0;
}

// ---- small problems: the whole Cholesky-path fit as one launch (small.hip) ---------------------------------------------------
void fill_small_prob(const mrbf_ctx *ctx, const mrbf_model *M, const double *Y, double *ws, int *flags, double *scal, int *cl, smallfit::Prob *P) {
    P->n = (int)M->n;
    P->d = M->d;
    P->k = M->k;
    P->q = M->q;
    P->deg = M->deg;
    P->n16 = (int)round_up(M->n, 16);
    P->npad = (int)M->npad;
    P->dpad = M->dpad;
    P->q16 = (int)round_up(std::max(M->q, 1), 16);
    P->kp = M->kp;
    P->C = M->C;
    P->Y = Y;
    P->Xc = M->Xc;
    P->sq = M->sq;
    P->mean = M->mean;
    P->W = M->W;
    P->Wc = M->Wc;
    P->lam = M->lam;
    P->ws = ws;
    P->flags = flags;
    P->scal = scal;
    P->stamps = nullptr;
    P->cl = cl;
    P->spin_ticks = (unsigned long long)std::max(1, ctx->spin_ms) * 100000ull;  // wall_clock64: 100 MHz
    P->fault = (ctx->debug_fault & 4) ? 1 : 0;
    {
        static const int d6 = mrbf_env("MRBF_SMALL_DIAG6") ? atoi(mrbf_env("MRBF_SMALL_DIAG6")) : 1;  // (round 5: n = 512 0.670 -> 0.628 ms, n = 100 0.171 -> 0.160 ms; 0 selects the v4 core)
        P->diag6 = d6;
    }
    P->mean_given = 0;
}
// what the flags of a small-problem fit mean for mrbf_fit_info (shared with the batched entry point): returns 1 when the problem
// has to be re-done on the LU path
int small_fit_verdict(const mrbf_model *M, const int *hflags, const double *hscal, mrbf_fit_info *info) {
    info->path = M->q > 0 ? MRBF_PATH_PROJ_CHOL : MRBF_PATH_CHOL;
    info->mu = M->q > 0 ? hscal[1] : 0.0;
    info->factor_info = hflags[0];
    if (hflags[1] != 0) {  // affinely dependent sites: Pi is rank deficient
        info->factor_info = -2;
        return 1;
    }
    if (hflags[2] != 0) {  // trace <= 0: Z' Phi Z cannot be positive definite
        info->factor_info = -1;
        return 1;
    }
    return hflags[0] != 0 ? 1 : 0;
}
static int fit_check_enqueue(mrbf_ctx *ctx, mrbf_model *M, const double *Y, double *h, hipEvent_t e0, hipEvent_t e1);
static int fit_check_finish(mrbf_ctx *ctx, const mrbf_model *M, const double *h, mrbf_fit_info *info, hipEvent_t e0, hipEvent_t e1);
// checked (optional): the residual check is enqueued behind the fit launch and read back in the same host round trip as the
// fit's flags (it runs on whatever the launch left -- its kernels terminate on any input -- and is discarded when a flag is set)
static int fit_small(mrbf_ctx *ctx, mrbf_model *M, const double *Y, mrbf_fit_info *info, int *not_pd, int force_nc, int *nc_used,
                     bool *checked = nullptr) {
    *not_pd = 0;
    if (checked) *checked = false;
    smallfit::Prob P;
    const smallfit::Carve cv = smallfit::carve((int)M->npad, (int)round_up(std::max(M->q, 1), 16));
    double *ws, *scal;
    int *flags;
    MRBF_TRY(get_buf(ctx, S_SMALL_WS, cv.total, &ws));
    MRBF_TRY(get_buf(ctx, S_SMALL_FLAGS, (size_t)4, &flags));
    MRBF_TRY(get_buf(ctx, S_MISC, (size_t)8, &scal));
    int *cl;
    MRBF_TRY(get_buf(ctx, S_SMALL_CL, (size_t)smallfit::CL_WORDS, &cl));
    fill_small_prob(ctx, M, Y, ws, flags, scal, cl, &P);
    static const int want_stamps = mrbf_env("MRBF_SMALL_STAMPS") ? atoi(mrbf_env("MRBF_SMALL_STAMPS")) : 0;
    long long *dstamps = nullptr;
    if (want_stamps) {
        MRBF_TRY(get_buf(ctx, S_SMALL_DESC, (size_t)16, &dstamps));
        P.stamps = dstamps;
    }
    // (read-backs through the context's pinned block where there is one: asynchronous downloads, one wait)
    int hflags_local[4] = {0, 0, 0, 0};
    double hscal_local[2] = {0.0, 0.0}, hchk_local[3] = {0, 0, 0};
    int *hflags = ctx->hpin ? reinterpret_cast<int *>(ctx->hpin + HPIN_SMALL_FLAGS) : hflags_local;
    double *hscal = ctx->hpin ? reinterpret_cast<double *>(ctx->hpin + HPIN_SMALL_SCAL) : hscal_local;
    double *hchk = ctx->hpin ? reinterpret_cast<double *>(ctx->hpin + HPIN_SMALL_CHK) : hchk_local;
    int nc = force_nc > 0 ? force_nc : small_fit_cluster(ctx, 1);
    for (int attempt = 0; attempt < 2; ++attempt) {
        MRBF_HIP(ctx, hipMemsetAsync(cl, 0, smallfit::CL_WORDS * sizeof(int), ctx->stream));
        MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        MRBF_TRY(launch_small_fit(ctx, &P, 1, nullptr, nc));
        MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        MRBF_HIP(ctx, hipMemcpyAsync(hflags, flags, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipMemcpyAsync(hscal, scal, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        hchk[0] = hchk[1] = hchk[2] = 0.0;
        const bool with_check = checked && ctx->residual;
        if (with_check) MRBF_TRY(fit_check_enqueue(ctx, M, Y, hchk, ctx->ev[2], ctx->ev[3]));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (with_check && hflags[3] == 0 && hflags[0] == 0 && hflags[1] == 0 && hflags[2] == 0) {
            MRBF_TRY(fit_check_finish(ctx, M, hchk, info, ctx->ev[2], ctx->ev[3]));
            *checked = true;
        }
        if (nc == 1) break;
        if (hflags[3] == 0) {
            ctx->small_timeouts = 0;
            break;
        }
        // the cluster gave up before it had relied on anything: repeat with one workgroup.  Members on different XCDs: clusters are off
        // for this context.  A barrier that timed out (a busy device, a test hook) says nothing about visibility: clusters stay on
        // unless it happens three times in a row.
        if (hflags[3] == 2 || ++ctx->small_timeouts >= 3) ctx->small_nc = 1;
        nc = 1;
    }
    *nc_used = nc;
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_factor, ctx->ev[0], ctx->ev[1]));  // one launch: assembly, projection, factorisation, solve
    if (want_stamps) {
        long long hs[16];
        MRBF_HIP(ctx, hipMemcpy(hs, dstamps, sizeof(hs), hipMemcpyDeviceToHost));
        static const char *names[] = {"centre", "gram", "Q1", "W,G,mu,V", "K update", "rhs", "potrf", "solves", "tail"};
        fprintf(stderr, "small fit n=%lld d=%d q=%d (%d workgroup%s): %.3f ms |", (long long)M->n, M->d, M->q, nc, nc == 1 ? "" : "s", info->ms_factor);
        for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.1f us |", names[i], (hs[i + 1] - hs[i]) * 0.01);
        fprintf(stderr, " [gram: product %.1f us, radial function %.1f us, padding %.1f us]\n", (hs[11] - hs[1]) * 0.01, (hs[12] - hs[11]) * 0.01,
                (hs[2] - hs[12]) * 0.01);
    }
    *not_pd = small_fit_verdict(M, hflags, hscal, info);
    return 0;
}

// residual ||s(C) - Y|| / ||Y|| through the evaluation kernels (an independent code path), and max |Pi' w|: the launches and the
// download of the three sums into h (between the events e0, e1); the caller synchronises the stream and calls fit_check_finish
static int fit_check_enqueue(mrbf_ctx *ctx, mrbf_model *M, const double *Y, double *h, hipEvent_t e0, hipEvent_t e1) {
    const int64_t n = M->n;
    const int k = M->k, q = M->q;
    double *V, *scal;
    MRBF_TRY(get_buf(ctx, S_STAGE_D, (size_t)n * k, &V));
    MRBF_TRY(get_buf(ctx, S_CHECK_SCAL, (size_t)8, &scal));
    MRBF_HIP(ctx, hipEventRecord(e0, ctx->stream));
    ctx->eval_check_call = 1;  // (the residual check keeps the split rule its batch twin uses: eval_nsplit)
    const int rc_eval = eval_model(ctx, M, n, M->C, V, nullptr, nullptr);
    ctx->eval_check_call = 0;
    if (rc_eval != 0) return rc_eval;
    hipLaunchKernelGGL(residual_kernel, dim3(1), dim3(256), 0, ctx->stream, V, Y, n * k, scal);
    if (q > 0) {
        double *Pi, *T;
        MRBF_TRY(get_buf(ctx, S_PI, (size_t)n * q, &Pi));
        MRBF_TRY(get_buf(ctx, S_T2, (size_t)q * k, &T));
        MRBF_TRY(launch_poly_matrix(ctx, M->C, n, M->d, q, Pi, n));
        MRBF_TRY(tsmm_tn(ctx, n, q, k, 1.0, Pi, n, M->Wc, M->npad, 0.0, T, q));
        hipLaunchKernelGGL(max_abs_kernel, dim3(1), dim3(256), 0, ctx->stream, T, (int64_t)q * k, scal + 2);
    }
    MRBF_HIP(ctx, hipGetLastError());
    MRBF_HIP(ctx, hipMemcpyAsync(h, scal, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipEventRecord(e1, ctx->stream));
    return 0;
}
static int fit_check_finish(mrbf_ctx *ctx, const mrbf_model *M, const double *h, mrbf_fit_info *info, hipEvent_t e0, hipEvent_t e1) {
    info->rel_residual = std::sqrt(h[0]) / std::max(std::sqrt(h[1]), 1e-300);
    info->max_pitw = M->q > 0 ? h[2] : 0.0;
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_check, e0, e1));
    return 0;
}
int fit_check(mrbf_ctx *ctx, mrbf_model *M, const double *Y, mrbf_fit_info *info) {
    double h_local[3] = {0, 0, 0};
    double *h = ctx->hpin ? reinterpret_cast<double *>(ctx->hpin + HPIN_RESIDUAL) : h_local;
    h[0] = h[1] = h[2] = 0.0;
    MRBF_TRY(fit_check_enqueue(ctx, M, Y, h, ctx->ev[0], ctx->ev[1]));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return fit_check_finish(ctx, M, h, info, ctx->ev[0], ctx->ev[1]);
}

int fit_model(mrbf_ctx *ctx, mrbf_model *M, const double *Y, mrbf_fit_info *info) {
    mrbf_fit_info local;
    if (!info) info = &local;
    std::memset(info, 0, sizeof(*info));
    info->n = (int32_t)M->n;
    info->q = M->q;
    info->rel_residual = NAN;
    info->max_pitw = NAN;
    const int order = cpd_order(M->kp.kid, M->kp.a, M->kp.b);
    int path = ctx->force_path;
    int small_nc_used = 0;
    bool checked = false;
    if (path == 0) path = (order <= M->deg + 1 && M->n > M->q) ? (M->q > 0 ? MRBF_PATH_PROJ_CHOL : MRBF_PATH_CHOL) : MRBF_PATH_LU;
    if (M->n < M->q) path = MRBF_PATH_MINNORM;  // under-determined tail: minimum-norm coefficients
    if (path == MRBF_PATH_CHOL && M->q > 0) path = MRBF_PATH_PROJ_CHOL;
    if (path == MRBF_PATH_PROJ_CHOL && M->q == 0) path = MRBF_PATH_CHOL;
    if (path == MRBF_PATH_MINNORM) {
        MRBF_TRY(fit_minnorm(ctx, M, Y, info));
    } else if (path != MRBF_PATH_LU && small_fit_applies(ctx, M->n, M->d, M->k, M->q, path)) {
        int not_pd = 0;
        MRBF_TRY(fit_small(ctx, M, Y, info, &not_pd, 0, &small_nc_used, &checked));
        if (not_pd) {
            info->fallbacks |= MRBF_FB_LU;
            if (ctx->force_path != 0)
                return fail(ctx, MRBF_ENOTPD, "Cholesky path forced but the matrix is not positive definite (info = %d)",
                            info->factor_info);
            path = MRBF_PATH_LU;
        }
    } else if (path != MRBF_PATH_LU) {
        int not_pd = 0, gave_up = 0;
        MRBF_TRY(fit_chol(ctx, M, Y, info, &not_pd, &gave_up));
        if (gave_up) {
            // same call, same GPU: re-assemble and factor with the host-driven blocked Cholesky (chol_impl 2)
            const int saved = ctx->chol_impl;
            ctx->chol_impl = 2;
            info->fallbacks |= MRBF_FB_CHOL_HOST_DRIVEN;
            const int rc2 = fit_chol(ctx, M, Y, info, &not_pd, &gave_up);
            ctx->chol_impl = saved;
            if (rc2 != 0) return rc2;
            if (gave_up) return fail(ctx, MRBF_EHIP, "host-driven Cholesky reported a give-up code (0x%x)", info->giveup_code);
        }
        if (not_pd) {
            info->fallbacks |= MRBF_FB_LU;
            if (ctx->force_path != 0)
                return fail(ctx, MRBF_ENOTPD, "Cholesky path forced but the matrix is not positive definite (info = %d)",
                            info->factor_info);
            path = MRBF_PATH_LU;  // retry on the saddle system (re-assembles Phi)
        }
    }
    if (path == MRBF_PATH_LU) MRBF_TRY(fit_lu(ctx, M, Y, info));
    if (ctx->residual && !(checked && path != MRBF_PATH_LU)) MRBF_TRY(fit_check(ctx, M, Y, info));
    if (ctx->residual && small_nc_used > 1 && path != MRBF_PATH_LU && !(info->rel_residual < 1e-6)) {
        // tripwire of the workgroup clusters (small.hip): a clustered fit that does not interpolate is repeated with one workgroup
        // per problem; only if THAT interpolates better was the cluster at fault (an ill-conditioned problem gives the same bits
        // again), and the context stops using clusters
        const double res_cluster = info->rel_residual;
        std::vector<double> wc((size_t)M->n * M->k);
        MRBF_HIP(ctx, hipMemcpy(wc.data(), M->W, wc.size() * sizeof(double), hipMemcpyDeviceToHost));
        int not_pd = 0, nc1 = 1;
        mrbf_fit_info again = *info;
        MRBF_TRY(fit_small(ctx, M, Y, &again, &not_pd, 1, &nc1));
        if (!not_pd) {
            MRBF_TRY(fit_check(ctx, M, Y, &again));
            std::vector<double> w1(wc.size());
            MRBF_HIP(ctx, hipMemcpy(w1.data(), M->W, w1.size() * sizeof(double), hipMemcpyDeviceToHost));
            if (std::memcmp(w1.data(), wc.data(), w1.size() * sizeof(double)) != 0 && (again.rel_residual < res_cluster || !(res_cluster == res_cluster)))
                ctx->small_nc = 1;
            *info = again;
        }
    }
    info->ms_total = info->ms_gram + info->ms_project + info->ms_factor + info->ms_solve;
    info->slow_launches = ctx->slow_launches;
    return 0;
}

}  // namespace mrbf
