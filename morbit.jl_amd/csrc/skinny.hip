// Tall-skinny products of the projection / solve steps.  rocBLAS handles these shapes poorly on this device
// (profiles/r01_*): a 65 x 65 x 8192 dgemm(T,N) takes 260 us, Phi * Q1 (8192 x 65 x 8192) 900 us.
//
//   tsmm_tn        C (p x r) = beta C + alpha A' B      A: n x p, B: n x r column-major, n large, p, r <= ~300
//                  two deterministic stages: row-chunk partials (VALU, 4 x 4 register blocks from LDS tiles), then a
//                  fixed-order sum over the chunks.
//   symm_panel     W (n x q) = Phi Q                    Phi: n x n (full storage), Q: n x q column-major, q <= 272
//                  f64 MFMA, 128-row blocks x all q columns per workgroup, K split over gridDim.y with per-split
//                  partial panels summed in a second fixed-order pass.
#include "common.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TS_RC = 128;  // rows per chunk (many short chunks: these products are latency-bound, not flop-bound)

// partial[chunk][i + j*p] for the 64 x 64 output block (blockIdx.y, blockIdx.z)
__global__ __launch_bounds__(256) void tsmm_tn_partial_kernel(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                                              int64_t ldb, int64_t n, int p, int r, double *__restrict__ part) {
    __shared__ double As[16][65];
    __shared__ double Bs[16][65];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;
    const int64_t k0 = (int64_t)blockIdx.x * TS_RC;
    const int64_t kend = (k0 + TS_RC < n) ? k0 + TS_RC : n;
    double acc[4][4] = {};
    const int lc = tid >> 2, lk = (tid & 3) * 4;  // staging: column lc (0..63), rows lk..lk+3 of the 16-row slice
    for (int64_t kb = k0; kb < kend; kb += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = kb + lk + u;
            As[lk + u][lc] = (k < kend && i0 + lc < p) ? A[k + (int64_t)(i0 + lc) * lda] : 0.0;
            Bs[lk + u][lc] = (k < kend && j0 + lc < r) ? B[k + (int64_t)(j0 + lc) * ldb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = As[kk][tx + 16 * u];
#pragma unroll
            for (int v = 0; v < 4; ++v) b[v] = Bs[kk][ty + 16 * v];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], b[v], acc[u][v]);
        }
        __syncthreads();
    }
    double *out = part + (size_t)blockIdx.x * p * r;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + tx + 16 * u, j = j0 + ty + 16 * v;
            if (i < p && j < r) out[i + (size_t)j * p] = acc[u][v];
        }
}

__global__ void tsmm_tn_reduce_kernel(const double *__restrict__ part, int nchunks, int p, int r, double alpha, double beta,
                                      double *__restrict__ C, int64_t ldc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p * r) return;
    // eight independent partial sums (fixed assignment c % 8, fixed final order): the loads of a thread are in flight together instead
    // of one dependent ~1 us load after the other (64 chunks: 18 us)
    double sp[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const size_t pr = (size_t)p * r;
    int c = 0;
    for (; c + 8 <= nchunks; c += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) sp[u] += part[(size_t)(c + u) * pr + idx];
    }
    for (int u = 0; c + u < nchunks; ++u) sp[u] += part[(size_t)(c + u) * pr + idx];
    const double s = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
    const int i = idx % p, j = idx / p;
    double *dst = C + i + (int64_t)j * ldc;
    *dst = (beta == 0.0) ? alpha * s : fma(alpha, s, beta * *dst);
}

// r <= 4 output columns (Q1' w, W' w: p x k from n rows): one workgroup per output row i, each thread strides down column i of
// A and the r columns of B, fixed-order tree reduction in LDS -- one ~6 us launch instead of a partial + a reduce kernel (44 us).
template <int R>
__global__ __launch_bounds__(256) void tsmm_tn_skinny_kernel(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                                             int64_t ldb, int64_t n, double alpha, double beta, double *__restrict__ C,
                                                             int64_t ldc) {
    __shared__ double red[R][256];
    const int i = blockIdx.x, tid = threadIdx.x;
    const double *a = A + (int64_t)i * lda;
    double acc[R];
#pragma unroll
    for (int l = 0; l < R; ++l) acc[l] = 0.0;
    // four strides at a time, all their loads issued before the first use (a stride per iteration is one dependent load latency each)
    int64_t k = tid;
    for (; k + 3 * 256 < n; k += 4 * 256) {
        double av[4], bv[4][R];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = a[k + 256 * u];
#pragma unroll
            for (int l = 0; l < R; ++l) bv[u][l] = B[k + 256 * u + (int64_t)l * ldb];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int l = 0; l < R; ++l) acc[l] = fma(av[u], bv[u][l], acc[l]);
    }
    for (; k < n; k += 256) {
        const double av = a[k];
#pragma unroll
        for (int l = 0; l < R; ++l) acc[l] = fma(av, B[k + (int64_t)l * ldb], acc[l]);
    }
#pragma unroll
    for (int l = 0; l < R; ++l) red[l][tid] = acc[l];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int l = 0; l < R; ++l) red[l][tid] += red[l][tid + s];
        }
        __syncthreads();
    }
    if (tid < R) {
        double *dst = C + i + (int64_t)tid * ldc;
        *dst = (beta == 0.0) ? alpha * red[tid][0] : fma(alpha, red[tid][0], beta * *dst);
    }
}

int tsmm_tn(mrbf_ctx *ctx, int64_t n, int p, int r, double alpha, const double *A, int64_t lda, const double *B, int64_t ldb,
            double beta, double *C, int64_t ldc) {
    if (p <= 0 || r <= 0) return 0;
    if (r <= 4 && p <= 4096) {
#define MRBF_TSK(RV) hipLaunchKernelGGL((tsmm_tn_skinny_kernel<RV>), dim3((unsigned)p), dim3(256), 0, ctx->stream, A, lda, B, ldb, n, alpha, beta, C, ldc)
        if (r == 1) MRBF_TSK(1); else if (r == 2) MRBF_TSK(2); else if (r == 3) MRBF_TSK(3); else MRBF_TSK(4);
#undef MRBF_TSK
        MRBF_HIP(ctx, hipGetLastError());
        return 0;
    }
    const int nchunks = (int)((n + TS_RC - 1) / TS_RC);
    double *part;
    MRBF_TRY(get_buf(ctx, S_R, (size_t)nchunks * p * r, &part));
    hipLaunchKernelGGL(tsmm_tn_partial_kernel, dim3(nchunks, (p + 63) / 64, (r + 63) / 64), dim3(256), 0, ctx->stream, A, lda, B, ldb, n,
                       p, r, part);
    hipLaunchKernelGGL(tsmm_tn_reduce_kernel, dim3((p * r + 255) / 256), dim3(256), 0, ctx->stream, part, nchunks, p, r, alpha, beta, C,
                       ldc);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- W = Phi Q on the matrix cores ----------------------------------------------------------------------------------
// Workgroup (bi, s): rows [128 bi, 128 bi + 128) x all NJT*16 columns, k range of split s.  Wave w owns rows 32w..32w+31
// (two 16-row tiles) and every column tile.  Operands staged per 16-wide k chunk: Phi chunk 128 x 16 as in the update
// kernel ([k][i] rows of 144 doubles), Q chunk 16 x (16 NJT) as [k][j].
// CF: additionally the row sums of Phi over this workgroup's k range (the product with a constant column, which then needs no
// MFMA column tile: q = 65 = 1 + 64 runs with 4 column tiles instead of 5) -> column qp_total - 16 of the partial result.
template <int NJT, bool CF>
__global__ __launch_bounds__(256, 2) void symm_panel_kernel(const double *__restrict__ Phi, int64_t ld, const double *__restrict__ Q,
                                                            int64_t ldq, int q, int64_t n, int ksplit_len,
                                                            double *__restrict__ Wpart, int64_t ldw, int qp_total) {
    constexpr int LDA_S = 128 + 16;
    constexpr int LDQ_S = NJT * 16 + ((NJT & 1) ? 0 : 16);  // LD % 32 == 16: the two k rows of a 32-lane half hit disjoint banks
    __shared__ __attribute__((aligned(16))) double As[16 * LDA_S];
    __shared__ __attribute__((aligned(16))) double Qs[16 * LDQ_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t I0 = (int64_t)blockIdx.x * 128;
    const int64_t kbeg = (int64_t)blockIdx.y * ksplit_len;
    const int64_t kend = (kbeg + ksplit_len < n) ? kbeg + ksplit_len : n;
    const int jg0 = blockIdx.z * NJT * 16;  // first column of this workgroup's column group
    v4d acc[NJT][2];
#pragma unroll
    for (int j = 0; j < NJT; ++j) {
        acc[j][0] = (v4d){0.0, 0.0, 0.0, 0.0};
        acc[j][1] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    const int a_i2 = (tid & 63) * 2, a_k0 = tid >> 6;
    // register prefetch, PF chunks of 16 columns ahead (Phi: 4 x 16 B per thread and chunk, Q: NJT doubles): with one chunk in flight
    // the loop was bound by the ~2.5 us a chunk takes to arrive -- 16 KB per workgroup in flight is ~3 TB/s over the chip (measured
    // 2.6), the MFMAs of a chunk take ~1 us
    constexpr int PF = (NJT <= 5) ? 3 : 1;  // nine column tiles leave no registers for a deeper prefetch (C5: 6.2 -> 10.5 ms with three)
    v2d ra[PF][4];
    double rq[PF][NJT];
    auto fetch = [&](int64_t kb, int slot) {
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            if (f == slot) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = a_k0 + 4 * u;
                    ra[f][u] = (v2d){0.0, 0.0};
                    if (kb + k < kend) ra[f][u] = *(const v2d *)(Phi + I0 + a_i2 + (kb + k) * ld);  // a wave reads one whole column (1 KiB)
                }
#pragma unroll
                for (int u = 0; u < NJT; ++u) {
                    const int e = tid + 256 * u;  // e < 16 * 16 * NJT
                    const int k = e & 15, j = jg0 + (e >> 4);
                    rq[f][u] = (j < q && kb + k < kend) ? Q[(kb + k) + (int64_t)j * ldq] : 0.0;  // 16 consecutive k are contiguous
                }
            }
        }
    };
#pragma unroll
    for (int f = 0; f < PF; ++f) fetch(kbeg + 16 * f, f);
    double rs0 = 0.0, rs1 = 0.0;  // CF: this thread's share of the row sums of rows a_i2, a_i2 + 1
#pragma unroll 1
    for (int64_t kb0 = kbeg; kb0 < kend; kb0 += 16 * PF) {
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const int64_t kb = kb0 + 16 * f;
            if (kb >= kend) break;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) *(v2d *)&As[(a_k0 + 4 * u) * LDA_S + a_i2] = ra[f][u];
            if (CF) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rs0 += ra[f][u][0];
                    rs1 += ra[f][u][1];
                }
            }
#pragma unroll
            for (int u = 0; u < NJT; ++u) {
                const int e = tid + 256 * u;
                Qs[(e & 15) * LDQ_S + (e >> 4)] = rq[f][u];
            }
            __syncthreads();
            if (kb + 16 * PF < kend) fetch(kb + 16 * PF, f);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(kk * 4 + l4) * LDA_S + wave * 32 + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < NJT; ++j) {
                const double b = Qs[(kk * 4 + l4) * LDQ_S + j * 16 + l15];
                // D[row = column j of W][col = row i of W]: lane & 15 runs along i, contiguous in the column-major output
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a[i], acc[j][i], 0, 0, 0);
            }
        }
        }
    }
    double *out = Wpart + (size_t)blockIdx.y * ldw * qp_total;
    if (CF && blockIdx.z == 0) {
        __syncthreads();
        As[a_k0 * LDA_S + a_i2] = rs0;
        As[a_k0 * LDA_S + a_i2 + 1] = rs1;
        __syncthreads();
        if (tid < 128) out[(I0 + tid) + (int64_t)(qp_total - 16) * ldw] = (As[tid] + As[LDA_S + tid]) + (As[2 * LDA_S + tid] + As[3 * LDA_S + tid]);
    }
#pragma unroll
    for (int j = 0; j < NJT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gj = jg0 + j * 16 + l4 + 4 * r;
#pragma unroll
            for (int i = 0; i < 2; ++i) out[(I0 + wave * 32 + i * 16 + l15) + (int64_t)gj * ldw] = acc[j][i][r];
        }
}

// cf: column 0 of W is the row-sum column (qp - 16 of the partial results) times cscale, column j >= 1 is partial column j - 1
__global__ void symm_panel_reduce_kernel(const double *__restrict__ Wpart, int nsplit, int64_t ldw, int qp, int64_t n, int q,
                                         double *__restrict__ W, int64_t ldwo, int cf, double cscale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * q) return;
    const int64_t i = idx % n;
    const int j = (int)(idx / n);
    const int jp = cf ? (j == 0 ? qp - 16 : j - 1) : j;
    double s = 0.0;
    for (int c = 0; c < nsplit; ++c) s += Wpart[(size_t)c * ldw * qp + i + (int64_t)jp * ldw];
    W[i + (int64_t)j * ldwo] = (cf && j == 0) ? s * cscale : s;
}

// W (n x q, ld ldwo) = Phi (npad x npad block, ld) * Q (n x q, ldq); rows >= n of W are left untouched.
// const_first: column 0 of Q is the constant cval on rows < n (its product is cval x the row sums of Phi, taken along the way).
int symm_panel(mrbf_ctx *ctx, int64_t n, int64_t npad, int q, const double *Phi, int64_t ld, const double *Q, int64_t ldq, double *W,
               int64_t ldwo, bool const_first, double cval) {
    // the row-sum route only when it saves a column tile
    const bool cf = const_first && q > 1 && (q - 1 + 15) / 16 < (q + 15) / 16;
    const int qm = cf ? q - 1 : q;  // columns that go through the matrix cores
    int njt = (qm + 15) / 16;
    // column tiles per workgroup: at most 9 (144 accumulator VGPRs; 17 would spill into AGPRs, where the f64 MFMA runs at
    // half rate, and to scratch: measured 27 ms instead of ~4 at q = 257); wider panels are cut into column groups
    const int choices[] = {1, 2, 3, 4, 5, 9};
    int pick = 9;
    for (int c : choices)
        if (c >= njt) {
            pick = c;
            break;
        }
    const int ngroups = (njt + pick - 1) / pick;
    static const int nsplit_env = mrbf_env("MRBF_SYMM_SPLIT") ? atoi(mrbf_env("MRBF_SYMM_SPLIT")) : 0;
    const int nsplit = nsplit_env > 0 ? nsplit_env : 8;
    const int klen = (int)(round_up((n + nsplit - 1) / nsplit, 16));
    const int qp = ngroups * pick * 16 + (cf ? 16 : 0);
    double *Wpart;
    MRBF_TRY(get_buf(ctx, S_EVAL_A, (size_t)nsplit * npad * qp, &Wpart));
    dim3 grid((unsigned)(npad / 128), nsplit, ngroups);
    const double *Qm = cf ? Q + ldq : Q;
#define MRBF_SP(NJTV)                                                                                                                  \
    do {                                                                                                                               \
        if (cf)                                                                                                                        \
            hipLaunchKernelGGL((symm_panel_kernel<NJTV, true>), grid, dim3(256), 0, ctx->stream, Phi, ld, Qm, ldq, qm, n, klen, Wpart, npad, qp); \
        else                                                                                                                           \
            hipLaunchKernelGGL((symm_panel_kernel<NJTV, false>), grid, dim3(256), 0, ctx->stream, Phi, ld, Qm, ldq, qm, n, klen, Wpart, npad, qp); \
    } while (0)
    switch (pick) {
        case 1: MRBF_SP(1); break;
        case 2: MRBF_SP(2); break;
        case 3: MRBF_SP(3); break;
        case 4: MRBF_SP(4); break;
        case 5: MRBF_SP(5); break;
        default: MRBF_SP(9); break;
    }
#undef MRBF_SP
    hipLaunchKernelGGL(symm_panel_reduce_kernel, dim3((unsigned)((n * q + 255) / 256)), dim3(256), 0, ctx->stream, Wpart, nsplit, npad,
                       qp, n, q, W, ldwo, cf ? 1 : 0, cval);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf
