// Small layout kernels: centring + zero padding + squared norms, polynomial matrix, transposes.
#include "common.hpp"

namespace mrbf {

// one block per dimension: deterministic tree reduction of the column mean
__global__ void column_mean_kernel(const double *__restrict__ X, int64_t n, int d, double *__restrict__ mean) {
    __shared__ double red[256];
    const int t = blockIdx.x;
    if (t >= d) {  // (blocks d .. dpad - 1 of launch_center_pad: the padding entries -- a memset launch less in front of every fit)
        if (threadIdx.x == 0) mean[t] = 0.0;
        return;
    }
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += X[i * d + t];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[t] = red[0] / (double)n;
}

// Xc[i][t] = X[i][t] - mean[t] (zero in the padding), sq[i] = |Xc[i]|^2.  One wave per row.
__global__ void center_pad_kernel(const double *__restrict__ X, int64_t n, int d, const double *__restrict__ mean,
                                  double *__restrict__ Xc, int64_t npad, int dpad, double *__restrict__ sq) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= npad) return;
    double s = 0.0;
    for (int t = lane; t < dpad; t += 64) {
        double v = 0.0;
        if (row < n && t < d) v = X[row * d + t] - mean[t];
        Xc[row * dpad + t] = v;
        s = fma(v, v, s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) sq[row] = s;
}

int launch_center_pad(mrbf_ctx *ctx, const double *X, int64_t n, int d, const double *mean_or_null, double *mean_out,
                      double *Xc, int64_t npad, int dpad, double *sq) {
    const double *mean = mean_or_null;
    if (!mean) {
        hipLaunchKernelGGL(column_mean_kernel, dim3(dpad > d ? dpad : d), dim3(256), 0, ctx->stream, X, n, d, mean_out);
        mean = mean_out;
    }
    const int rows_per_block = 4;
    hipLaunchKernelGGL(center_pad_kernel, dim3((unsigned)((npad + rows_per_block - 1) / rows_per_block)), dim3(256), 0,
                       ctx->stream, X, n, d, mean, Xc, npad, dpad, sq);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// Pi[i + t*ld] = 1 (t = 0) or C[i][t-1]
__global__ void poly_matrix_kernel(const double *__restrict__ C, int64_t n, int d, int q, double *__restrict__ Pi,
                                   int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int t = 0; t < q; ++t) Pi[i + (int64_t)t * ld] = (t == 0) ? 1.0 : C[i * d + (t - 1)];
}

int launch_poly_matrix(mrbf_ctx *ctx, const double *C, int64_t n, int d, int q, double *Pi, int64_t ldpi) {
    if (q <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(poly_matrix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, C, n, d, q, Pi,
                       ldpi);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// out[c * rows + r] = in[r * cols + c]   (row-major rows x cols  ->  column-major rows x cols)
__global__ void transpose_kernel(const double *__restrict__ in, int64_t rows, int64_t cols, double *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols, c = idx % cols;
    out[c * rows + r] = in[idx];
}

int launch_transpose(mrbf_ctx *ctx, const double *in, int64_t rows, int64_t cols, double *out) {
    if (rows * cols <= 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, ctx->stream, in, rows,
                       cols, out);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf
