// Built-in blocked right-looking Cholesky (lower, column-major, in place) on the fp64 matrix cores.
//
// rocSOLVER's dpotrf spends most of its time in tiny latency-bound kernels at n = 8192 (profiles/r01_a_*:
// potf2_kernel_small 169 us x 64, forward-substitution kernels, small syrk kernels; 5.9 TFLOP/s).  This
// file replaces it for the n x n s.p.d. system of the interpolation-weight solve with three kernels:
//
//   D  chol_diag_kernel    128 x 128 diagonal block: Cholesky in registers (one workgroup, column exchange
//                          through LDS, one barrier per column) + the inverse of the factor (column sweep).
//   T  chol_update_kernel<64, OVERWRITE>   panel  A21 <- A21 * inv(L11)'   -- the triangular solve as a GEMM
//   U  chol_update_kernel<128, LOWER>      trailing update  A22 -= A21 A21'  on lower-triangular tile pairs
//
// all tiles 128-aligned: the caller hands a matrix whose order is a multiple of 128 (identity padding).
// The update kernel is an LDS-tiled v_mfma_f64_16x16x4_f64 GEMM  C(i,j) (op)= sum_k A(i,k) B(j,k):
// 16-column k-chunks staged global -> registers -> LDS ([k][i] layout, 144-double rows: conflict-free
// ds_read_b64 fragments), next chunk's global loads in flight under the current chunk's 64 MFMAs per wave.
#include "common.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int CNB = 128;  // block size of the factorisation
constexpr int CBK = 16;   // k chunk

enum { UPD_LOWER_SUB = 0, UPD_OVERWRITE = 1, UPD_FULL_SUB = 2 };

// C(i,j) (op)= sum_k A(i,k) * B(j,k).  A: (tiles_i*TM) x K, B: (tiles_j*128) x K, all column-major.
// MODE LOWER_SUB: square region, grid.x = lower-triangular tile pairs (TM == 128), C -= ..., strictly-upper
// entries of diagonal tiles left untouched.  OVERWRITE / FULL_SUB: grid = (tiles_i, tiles_j).
template <int TM, int MODE>
__global__ __launch_bounds__(256, 2) void chol_update_kernel(const double *__restrict__ A, int64_t lda,
                                                             const double *__restrict__ B, int64_t ldb,
                                                             double *__restrict__ C, int64_t ldc, int K,
                                                             const int *__restrict__ info) {
    constexpr int LDA_S = TM + 16;   // LDS row strides (doubles); (2*LD) % 64 == 32 -> k and k+1 rows hit disjoint banks
    constexpr int LDB_S = 128 + 16;
    constexpr int NJ = (TM == 128) ? 4 : 2;  // 16-wide j tiles per wave
    constexpr int AL = (TM == 128) ? 4 : 2;  // v2d loads per thread per A chunk
    __shared__ __attribute__((aligned(16))) double smem[CBK * LDA_S + CBK * LDB_S];
    double *As = smem;
    double *Bs = smem + CBK * LDA_S;
    if (info && *info != 0) return;  // an earlier diagonal block was not positive definite
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int ti, tj;
    if (MODE == UPD_LOWER_SUB) {
        int t = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
        while ((t + 1) * (t + 2) / 2 <= (int)blockIdx.x) ++t;
        while (t * (t + 1) / 2 > (int)blockIdx.x) --t;
        ti = t;
        tj = blockIdx.x - t * (t + 1) / 2;
    } else {
        ti = blockIdx.x;
        tj = blockIdx.y;
    }
    const int64_t I0 = (int64_t)ti * TM, J0 = (int64_t)tj * 128;
    const int ioff = (TM == 128) ? (wave >> 1) * 64 : 0;
    const int joff = (TM == 128) ? (wave & 1) * 64 : wave * 32;

    v4d acc[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (v4d){0.0, 0.0, 0.0, 0.0};

    // global -> register staging maps (a wave reads whole columns: 1 KiB / 512 B contiguous)
    const int a_i2 = (TM == 128) ? (tid & 63) * 2 : (tid & 31) * 2;
    const int a_k0 = (TM == 128) ? (tid >> 6) : (tid >> 5);
    constexpr int a_ks = (TM == 128) ? 4 : 8;
    const int b_i2 = (tid & 63) * 2, b_k0 = tid >> 6;
    const double *Ap = A + I0 + a_i2 + (int64_t)a_k0 * lda;
    const double *Bp = B + J0 + b_i2 + (int64_t)b_k0 * ldb;
    v2d ra[AL], rb[4];
#pragma unroll
    for (int u = 0; u < AL; ++u) ra[u] = *(const v2d *)(Ap + (int64_t)(a_ks * u) * lda);
#pragma unroll
    for (int u = 0; u < 4; ++u) rb[u] = *(const v2d *)(Bp + (int64_t)(4 * u) * ldb);

    const int nkc = K / CBK;
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < AL; ++u) *(v2d *)&As[(a_k0 + a_ks * u) * LDA_S + a_i2] = ra[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) *(v2d *)&Bs[(b_k0 + 4 * u) * LDB_S + b_i2] = rb[u];
        __syncthreads();
        if (kc + 1 < nkc) {
            const int64_t ko = (int64_t)(kc + 1) * CBK;
#pragma unroll
            for (int u = 0; u < AL; ++u) ra[u] = *(const v2d *)(Ap + (ko + a_ks * u) * lda);
#pragma unroll
            for (int u = 0; u < 4; ++u) rb[u] = *(const v2d *)(Bp + (ko + 4 * u) * ldb);
        }
#pragma unroll
        for (int kk = 0; kk < CBK / 4; ++kk) {
            double a[4], b[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(kk * 4 + l4) * LDA_S + ioff + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = Bs[(kk * 4 + l4) * LDB_S + joff + j * 16 + l15];
            // D[row = j][col = i] = sum_k Bp(j,k) Ap(i,k): the lane index (l & 15) runs along i, contiguous in C
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[j][i], 0, 0, 0);
        }
    }
    // epilogue (f64 C/D map: col = lane & 15 -> i, row = (lane >> 4) + 4 r -> j)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gj = J0 + joff + j * 16 + l4 + 4 * r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t gi = I0 + ioff + i * 16 + l15;
                double *dst = C + gi + gj * ldc;
                if (MODE == UPD_OVERWRITE) {
                    *dst = acc[j][i][r];
                } else if (MODE == UPD_FULL_SUB) {
                    *dst -= acc[j][i][r];
                } else {
                    if (ti != tj || gi >= gj) *dst -= acc[j][i][r];
                }
            }
        }
}

// ---- D: Cholesky of one 128 x 128 diagonal block + inverse of its factor ---------------------------------
// 256 threads as a 16 x 16 grid; thread (ti, tj) keeps A[ti + 16u][tj + 16v], u, v = 0..7, in registers.
// Right-looking: at column k its 16 owner threads scale the column (1/sqrt of the pivot, broadcast inside
// their 16-lane group) and publish it through a double-buffered LDS vector; one barrier per column; every
// thread then applies the rank-1 update to the part of its 8 x 8 register block that lies in the trailing
// lower triangle.  The 8 phases (k / 16) are separate template instantiations so register indices stay
// static and finished column blocks drop out of the update.
struct DiagShared {
    double Ls[CNB * (CNB + 1)];  // factor, column-major, ld 129 (read by the inverse sweep)
    double colb[2][CNB];
    double rdiag[CNB];
    int sbad;
};

template <int KV>
__device__ __forceinline__ void chol_phase(double (&a)[8][8], DiagShared &sh, int ti, int tj, int &buf, int &bad) {
#pragma unroll 1
    for (int kk = 0; kk < 16; ++kk) {
        const int k = KV * 16 + kk;
        // the pivot lives in lane (ti == kk) of the owner group (tj == kk), register a[KV][KV]
        const double piv = __shfl(a[KV][KV], kk, 16);
        if (tj == kk) {
            double rinv = 0.0;
            if (piv > 0.0) {
                rinv = 1.0 / sqrt(piv);
            } else if (bad == 0) {
                bad = k + 1;
            }
            if (ti == 0) sh.rdiag[k] = rinv;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = ti + 16 * u;
                double val = 0.0;
                if (i >= k) {
                    val = a[u][KV] * rinv;
                    a[u][KV] = val;
                }
                sh.colb[buf][i] = val;
            }
        }
        __syncthreads();
        double li[8], lj[8];
#pragma unroll
        for (int u = KV; u < 8; ++u) li[u] = sh.colb[buf][ti + 16 * u];
#pragma unroll
        for (int v = KV; v < 8; ++v) lj[v] = sh.colb[buf][tj + 16 * v];
#pragma unroll
        for (int v = KV; v < 8; ++v) {
            const int j = tj + 16 * v;
#pragma unroll
            for (int u = v; u < 8; ++u) {  // i >= j needs u >= v
                const int i = ti + 16 * u;
                if (j > k && i >= j) a[u][v] = fma(-li[u], lj[v], a[u][v]);
            }
        }
        buf ^= 1;
    }
}

// X = L^-1 by a right-looking sweep over rows: x[k][:] /= L[k][k]; x[i][:] -= L[i][k] x[k][:] for i > k
template <int KU>
__device__ __forceinline__ void inv_phase(double (&x)[8][8], DiagShared &sh, int ti, int tj, int &buf) {
#pragma unroll 1
    for (int kk = 0; kk < 16; ++kk) {
        const int k = KU * 16 + kk;
        if (ti == kk) {  // owners of row k: one lane per 16-lane group, every tj
            const double rinv = sh.rdiag[k];
#pragma unroll
            for (int v = 0; v <= KU; ++v) {
                const double val = x[KU][v] * rinv;
                x[KU][v] = val;
                sh.colb[buf][tj + 16 * v] = val;
            }
        }
        __syncthreads();
        double xr[8];
#pragma unroll
        for (int v = 0; v <= KU; ++v) xr[v] = sh.colb[buf][tj + 16 * v];
#pragma unroll
        for (int u = KU; u < 8; ++u) {
            const int i = ti + 16 * u;
            const double lik = (i > k) ? sh.Ls[i + k * (CNB + 1)] : 0.0;
#pragma unroll
            for (int v = 0; v <= KU; ++v) x[u][v] = fma(-lik, xr[v], x[u][v]);  // xr is 0 for columns j > k
        }
        buf ^= 1;
    }
}

__global__ __launch_bounds__(256, 1) void chol_diag_kernel(double *__restrict__ A, int64_t lda, double *__restrict__ Linv,
                                                        int *__restrict__ info, int col0) {
    __shared__ DiagShared sh;
    if (*info != 0) return;
    const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
    double a[8][8];
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ti + 16 * u, j = tj + 16 * v;
            a[u][v] = (i >= j) ? A[i + (int64_t)j * lda] : 0.0;
        }
    int buf = 0, bad = 0;
    chol_phase<0>(a, sh, ti, tj, buf, bad);
    chol_phase<1>(a, sh, ti, tj, buf, bad);
    chol_phase<2>(a, sh, ti, tj, buf, bad);
    chol_phase<3>(a, sh, ti, tj, buf, bad);
    chol_phase<4>(a, sh, ti, tj, buf, bad);
    chol_phase<5>(a, sh, ti, tj, buf, bad);
    chol_phase<6>(a, sh, ti, tj, buf, bad);
    chol_phase<7>(a, sh, ti, tj, buf, bad);
    // non-positive pivot: report the 1-based global index of the first one (LAPACK potrf convention)
    if (tid == 0) sh.sbad = 0x7fffffff;
    __syncthreads();
    if (bad) atomicMin(&sh.sbad, bad);
    __syncthreads();
    if (sh.sbad != 0x7fffffff) {
        if (tid == 0) *info = col0 + sh.sbad;
        return;
    }
    // write the factor: global (lower part of the block) + LDS copy for the inverse sweep
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ti + 16 * u, j = tj + 16 * v;
            if (i >= j) A[i + (int64_t)j * lda] = a[u][v];
            sh.Ls[i + j * (CNB + 1)] = (i >= j) ? a[u][v] : 0.0;
        }
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u][v] = (ti + 16 * u == tj + 16 * v) ? 1.0 : 0.0;
    __syncthreads();
    buf = 0;
    inv_phase<0>(a, sh, ti, tj, buf);
    inv_phase<1>(a, sh, ti, tj, buf);
    inv_phase<2>(a, sh, ti, tj, buf);
    inv_phase<3>(a, sh, ti, tj, buf);
    inv_phase<4>(a, sh, ti, tj, buf);
    inv_phase<5>(a, sh, ti, tj, buf);
    inv_phase<6>(a, sh, ti, tj, buf);
    inv_phase<7>(a, sh, ti, tj, buf);
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ti + 16 * u, j = tj + 16 * v;
            Linv[i + j * CNB] = (i >= j) ? a[u][v] : 0.0;
        }
}

// identity padding of rows/columns [n, npad) of an npad x npad column-major matrix (lower part is what matters)
__global__ void pad_identity_kernel(double *__restrict__ A, int64_t n, int64_t npad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pw = npad - n;
    if (idx >= npad * pw) return;
    const int64_t j = n + idx / npad, i = idx % npad;  // columns n..npad-1 (all rows)
    A[i + j * npad] = (i == j) ? 1.0 : 0.0;
    if (i < n) A[j + i * npad] = 0.0;  // mirrored rows n..npad-1 in columns < n
}

int launch_pad_identity(mrbf_ctx *ctx, double *A, int64_t n, int64_t npad) {
    if (npad == n) return 0;
    const int64_t cnt = npad * (npad - n);
    hipLaunchKernelGGL(pad_identity_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, A, n, npad);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// A: n x n with n % 128 == 0, lda % 2 == 0, 16-byte aligned.  dinfo: device int, set to 0 here.
int potrf_blocked(mrbf_ctx *ctx, int64_t n, double *A, int64_t lda, int *dinfo) {
    if (n % CNB != 0 || (lda & 1) || (reinterpret_cast<uintptr_t>(A) & 15))
        return fail(ctx, MRBF_EHIP, "potrf_blocked needs a 128-padded, 16-byte aligned matrix (n=%lld lda=%lld)", (long long)n,
                    (long long)lda);
    double *Linv;
    MRBF_TRY(get_buf(ctx, S_CHOL_WS, (size_t)CNB * CNB, &Linv));
    MRBF_HIP(ctx, hipMemsetAsync(dinfo, 0, sizeof(int), ctx->stream));
    const int nb = (int)(n / CNB);
    for (int j = 0; j < nb; ++j) {
        const int64_t c = (int64_t)j * CNB;
        double *Ajj = A + c + c * lda;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, ctx->stream, Ajj, lda, Linv, dinfo, (int)c);
        const int64_t m = n - c - CNB;
        if (m <= 0) break;
        double *A21 = A + (c + CNB) + c * lda;
        // T: A21 <- A21 * Linv'   (each workgroup owns full rows of the 128-wide panel: in place is safe)
        hipLaunchKernelGGL((chol_update_kernel<64, UPD_OVERWRITE>), dim3((unsigned)(m / 64), 1), dim3(256), 0, ctx->stream,
                           A21, lda, Linv, (int64_t)CNB, A21, lda, CNB, dinfo);
        // U: A22 -= A21 A21'  on lower-triangular tile pairs
        const int64_t mt = m / CNB;
        double *A22 = A + (c + CNB) + (c + CNB) * lda;
        hipLaunchKernelGGL((chol_update_kernel<128, UPD_LOWER_SUB>), dim3((unsigned)(mt * (mt + 1) / 2)), dim3(256), 0,
                           ctx->stream, A21, lda, A21, lda, A22, lda, CNB, dinfo);
    }
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf
