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
#include "mega_gemm.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int CNB = 128;  // block size of the factorisation

enum { UPD_LOWER_SUB = 0, UPD_OVERWRITE = 1, UPD_FULL_SUB = 2, UPD_COLUMN_SUB = 3 };
int launch_diag_v4(mrbf_ctx *ctx, hipStream_t st, double *Ajj, int64_t lda, double *Linv, int *dinfo, int col0);  // chol_diag.hip
int potrf_mega_tall(mrbf_ctx *ctx, int64_t ncols, int64_t mrows, double *A, int64_t lda, int *dinfo, double *linv_all);  // chol_mega.hip

// C(i,j) (op)= sum_k A(i,k) * B(j,k).  A: (tiles_i*TM) x K, B: (tiles_j*128) x K, all column-major.
// MODE LOWER_SUB: square region, grid.x = lower-triangular tile pairs (TM == 128), C -= ..., strictly-upper
// entries of diagonal tiles left untouched.  OVERWRITE / FULL_SUB: grid = (tiles_i, tiles_j).
// COLUMN_SUB: one block column whose first row tile is the diagonal tile (grid = (tiles_i, 1)).
template <int TM, int MODE>
__global__ __launch_bounds__(256, 2) void chol_update_kernel(const double *__restrict__ A, int64_t lda,
                                                             const double *__restrict__ B, int64_t ldb,
                                                             double *__restrict__ C, int64_t ldc, int K,
                                                             const int *__restrict__ info, int mt_sq = 0, int ntiles_total = 0,
                                                             const double *__restrict__ cvec = nullptr, double cs = 0.0, int64_t cn = 0) {
    constexpr int NJ = (TM == 128) ? 4 : 2;  // 16-wide j tiles per wave
    // operand pipeline: the LDS-DMA ring of the persistent factorisation's jobs (mega_gemm.hpp; round 4 -- the register-staged loop
    // this kernel had left the matrix pipe idle a third of the time: the projection's rank-2q update ran at 42 TFLOP/s)
    __shared__ __attribute__((aligned(16))) double smem[mega::NSTG * mega::STG];
    if (info && *info != 0) return;  // an earlier diagonal block was not positive definite
    // panel-chain launches (T, U1) share SIMDs with the bulk update's waves: win the issue arbitration
    if (MODE == UPD_OVERWRITE || MODE == UPD_COLUMN_SUB) __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // LOWER_SUB launches may be persistent: gridDim.x workgroups stride over ntiles tiles (ntiles_total > gridDim.x), which
    // leaves CU slots free for the panel chain running on the other stream
    const int tile_end = (MODE == UPD_LOWER_SUB && ntiles_total > 0) ? ntiles_total : (int)blockIdx.x + 1;
    for (int bid = blockIdx.x; bid < tile_end; bid += gridDim.x) {
    int ti, tj;
    if (MODE == UPD_LOWER_SUB) {
        // trapezoid: lower-triangular tile pairs of the mt_sq x mt_sq square, then full rows of tiles below it
        const int ntri = mt_sq * (mt_sq + 1) / 2;
        if (bid < ntri || mt_sq == 0) {
            int t = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
            while ((t + 1) * (t + 2) / 2 <= bid) ++t;
            while (t * (t + 1) / 2 > bid) --t;
            ti = t;
            tj = bid - t * (t + 1) / 2;
        } else {
            ti = mt_sq + (bid - ntri) / mt_sq;
            tj = (bid - ntri) % mt_sq;
        }
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

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (persistent launches: the previous tile's stores are out of the wave's counter)
    // LOWER_SUB (the projection's rank-2q update, the host-driven trailing update): start from the negated tile instead of ending
    // with a read-modify-write -- the 64 loads per lane travel under the operand ring's prologue and the first stages' MFMAs (they
    // are older than every LDS-DMA load, so the ring's counted waits cover them), and the epilogue is stores only (round 5; the
    // persistent factorisation's bulk jobs do the same, chol_mega.hip load_tile_neg)
    constexpr bool NEG_START = (MODE == UPD_LOWER_SUB);
    if (NEG_START) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gj = J0 + joff + j * 16 + l4 + 4 * r;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i][r] = -C[(I0 + ioff + i * 16 + l15) + gj * ldc];
            }
    }
    mega::gemm_acc<TM>(A + I0, lda, B + J0, ldb, K, acc, smem);
    // epilogue (f64 C/D map: col = lane & 15 -> i, row = (lane >> 4) + 4 r -> j).  The read-modify-write is done in
    // batches of 16 values (all loads of a batch issued before its first store): element-wise `*dst -= acc` makes the
    // compiler serialise 64 dependent load -> store round trips, because it cannot prove the addresses distinct.
    // cvec: the rank-2 term of a constant column, C -= cs (e v' + v e') with e = 1 on the first cn rows: applied here instead of
    // spending two of the K columns (and a whole 16-column chunk of MFMAs) on it.  Row / column values fetched once per thread.
    double cvi[4] = {0.0, 0.0, 0.0, 0.0}, cei[4] = {0.0, 0.0, 0.0, 0.0};
    if (MODE == UPD_LOWER_SUB && cvec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gi = I0 + ioff + i * 16 + l15;
            cvi[i] = cs * cvec[gi];
            cei[i] = gi < cn ? cs : 0.0;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        double cv[4][4];
        double cvj[4] = {0.0, 0.0, 0.0, 0.0}, cej[4] = {0.0, 0.0, 0.0, 0.0};
        if (MODE == UPD_LOWER_SUB && cvec) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gj = J0 + joff + j * 16 + l4 + 4 * r;
                cvj[r] = cvec[gj];
                cej[r] = gj < cn ? 1.0 : 0.0;
            }
        }
        if (MODE != UPD_OVERWRITE && !NEG_START) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gj = J0 + joff + j * 16 + l4 + 4 * r;
#pragma unroll
                for (int i = 0; i < 4; ++i) cv[r][i] = C[(I0 + ioff + i * 16 + l15) + gj * ldc];
            }
        }
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
                    *dst = cv[r][i] - acc[j][i][r];
                } else if (MODE == UPD_COLUMN_SUB) {
                    if (gi >= gj) *dst = cv[r][i] - acc[j][i][r];  // only the diagonal 128 x 128 block has gi < gj
                } else {
                    double x = -acc[j][i][r];  // (acc = -C + A B': the new entry is -acc)
                    if (cvec) x -= fma(cei[i], cvj[r], cvi[i] * cej[r]);  // cs (e_i v_j + v_i e_j)
                    if (ti != tj || gi >= gj) *dst = x;
                }
            }
        }
    }
    if (bid + (int)gridDim.x < tile_end) __syncthreads();  // LDS reuse by the next tile
    }  // persistent tile loop
}

// ---- D: Cholesky of one 128 x 128 diagonal block + inverse of its factor, ONE sweep -------------------------
// The block is augmented with an identity below it, [A; I] (256 x 128): the right-looking elimination that
// turns A into L turns the identity rows into I * L^-T, i.e. the inverse comes out of the same 128 column
// steps (no second sweep).  512 threads: threads 0..255 (half 0) hold A as a 16 x 16 thread grid with an
// 8 x 8 register block each (thread (ti, tj): rows ti + 16u, columns tj + 16v); threads 256..511 (half 1)
// hold the identity rows the same way.  Per column k: its owner threads publish the scaled column (half 0)
// / the unscaled column (half 1) and 1/l_kk through a double-buffered LDS vector, ONE barrier, then every
// thread applies the rank-1 update to the live part of its register block.  1/sqrt(pivot) is v_rsq_f64 +
// Newton steps (a correctly rounded sqrt and divide would put ~500 cycles on the critical path per column).
// The 8 phases (k / 16) are separate template instantiations so register indices stay static.
struct __attribute__((aligned(16))) DiagShared {
    double col[2][2 * CNB];  // [buf][0..127] scaled column of L, [128..255] unscaled column of the identity rows
    double rinv[2];
    int sbad;
    unsigned long long seg[4];  // diagnostic launches only: cycles of wave 0 in barrier / read / critical / rest
    int stamp;
};

__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
#pragma unroll
    for (int it = 0; it < 3; ++it) y = y * fma(-h, y * y, 1.5);
    return y;
}

// One phase = 16 columns.  On entry column k = 16 KV of L (scaled), the matching unscaled column of the
// identity rows and 1/l_kk are already published in sh.col[buf] / sh.rinv[buf].  Each iteration: barrier,
// read the published column, FIRST bring the next column (k+1) up to date, scale and publish it (the
// critical path: pivot copy -> rsqrt -> scale -> LDS), THEN apply the rest of the rank-1 update of column k,
// so that work overlaps the other waves' path to the next barrier.  Every thread keeps a private copy `piv`
// of its group's pivot a[j][j], j = tj + 16 KV, updated with the same fma as the real entry, so no
// cross-lane traffic sits on the critical path inside a phase.
template <int KV, bool STAMP>
__device__ __forceinline__ void diag_phase(double (&a)[8][8], DiagShared &sh, int ti, int tj, int half, int &buf, int &bad) {
    constexpr int KN = (KV < 7) ? KV + 1 : 7;  // register column block of the first column of the next phase
    double piv = 0.0;
    if (half == 0) piv = __shfl(a[KV][KV], tj, 16);  // true value lives in lane (ti == tj) of this 16-lane group
#pragma unroll 1
    for (int kk = 0; kk < 16; ++kk) {
        const int k = KV * 16 + kk;
        const bool st = STAMP && threadIdx.x < 64;
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if (st) s0 = __builtin_amdgcn_s_memtime();
        __syncthreads();  // column k, the identity rows' column k and rinv_k are visible
        if (st) s1 = __builtin_amdgcn_s_memtime();
        double lj[8];
#pragma unroll
        for (int v = KV; v < 8; ++v) lj[v] = sh.col[buf][tj + 16 * v];
        if (half == 0) {
            double li[8];
#pragma unroll
            for (int u = KV; u < 8; ++u) li[u] = sh.col[buf][ti + 16 * u];
            if (st) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                s2 = __builtin_amdgcn_s_memtime();
            }
            // ---- critical path: next column kn = k + 1
            if (kk < 15) {
                if (tj > kk) piv = fma(-lj[KV], lj[KV], piv);  // groups whose column of this phase is still ahead
                if (tj == kk + 1) {
                    const int kn = k + 1;
                    double rinv = 0.0;
                    if (piv > 0.0) {
                        rinv = fast_rsqrt(piv);
                    } else if (bad == 0) {
                        bad = kn + 1;
                    }
                    if (ti == 0) sh.rinv[buf ^ 1] = rinv;
                    double w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = ti + 16 * u;
                        double val = 0.0;
                        if (u >= KV && i >= kn) {
                            val = fma(-li[u], lj[KV], a[u][KV]) * rinv;
                            a[u][KV] = val;
                        }
                        w[u] = val;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sh.col[buf ^ 1][ti + 16 * u] = w[u];
                }
            } else if (KV < 7) {
                // last column of the phase: the next column opens phase KV + 1, owner group tj == 0,
                // pivot a[kn][kn] in lane ti == 0 of that group, register a[KN][KN]
                if (tj == 0) {
                    const int kn = k + 1;
                    const double pv = __shfl(fma(-li[KN], lj[KN], a[KN][KN]), 0, 16);
                    double rinv = 0.0;
                    if (pv > 0.0) {
                        rinv = fast_rsqrt(pv);
                    } else if (bad == 0) {
                        bad = kn + 1;
                    }
                    if (ti == 0) sh.rinv[buf ^ 1] = rinv;
                    double w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        double val = 0.0;
                        if (u >= KN) {  // i >= kn = 16 KN always holds here
                            val = fma(-li[u], lj[KN], a[u][KN]) * rinv;
                            a[u][KN] = val;
                        }
                        w[u] = val;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sh.col[buf ^ 1][ti + 16 * u] = w[u];
                }
            }
            if (st) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                s3 = __builtin_amdgcn_s_memtime();
            }
            // ---- rest of the rank-1 update of column k (the next column's entries were done above by its owners).
            // Unconditional FMAs with masked operands: only the diagonal register blocks (u == v) need i >= j,
            // which is the per-thread constant ti >= tj, and only the v == KV block needs j > k (tj > kk).
            const bool own_next = (kk < 15) ? (tj == kk + 1) : (KV < 7 && tj == 0);
            const bool lowtri = ti >= tj;
#pragma unroll
            for (int v = KV; v < 8; ++v) {
                double ljv = lj[v];
                if (v == KV) ljv = (tj > kk && !(own_next && kk < 15)) ? ljv : 0.0;
                if (v == KN && KV < 7) ljv = (own_next && kk == 15) ? 0.0 : ljv;
                a[v][v] = fma(lowtri ? -li[v] : 0.0, ljv, a[v][v]);
#pragma unroll
                for (int u = v + 1; u < 8; ++u) a[u][v] = fma(-li[u], ljv, a[u][v]);
            }
            if (st && threadIdx.x == 0) {
                asm volatile("" ::"v"(a[7][7]));
                const unsigned long long s4 = __builtin_amdgcn_s_memtime();
                sh.seg[0] += s1 - s0;
                sh.seg[1] += s2 - s1;
                sh.seg[2] += s3 - s2;
                sh.seg[3] += s4 - s3;
            }
        } else {
            // identity rows r = ti + 16u: e[r][k] *= rinv_k; e[r][j] -= e[r][k] l_jk (j > k); rows r > k are still zero
            const double rinv = sh.rinv[buf];
            const double ljk = (tj > kk) ? lj[KV] : 0.0;  // j > k matters for the v == KV block only
#pragma unroll
            for (int u = 0; u <= KV; ++u) {
                const double erk = sh.col[buf][CNB + ti + 16 * u] * rinv;
                if (tj == kk) a[u][KV] = erk;
                a[u][KV] = fma(-erk, ljk, a[u][KV]);
#pragma unroll
                for (int v = KV + 1; v < 8; ++v) a[u][v] = fma(-erk, lj[v], a[u][v]);
            }
            // publish the (unscaled) next column of these rows for the next iteration
            if (kk < 15) {
                if (tj == kk + 1) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) sh.col[buf ^ 1][CNB + ti + 16 * u] = (u <= KV) ? a[u][KV] : 0.0;
                }
            } else if (KV < 7) {
                if (tj == 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) sh.col[buf ^ 1][CNB + ti + 16 * u] = (u <= KN) ? a[u][KN] : 0.0;
                }
            }
        }
        buf ^= 1;
    }
}

template <bool STAMP>
__global__ __launch_bounds__(512) void chol_diag_kernel(double *__restrict__ A, int64_t lda, double *__restrict__ Linv,
                                                        int *__restrict__ info, int col0,
                                                        unsigned long long *__restrict__ stamps) {
    __shared__ DiagShared sh;
    if (*info != 0) return;
    unsigned long long t0 = 0, r0 = 0;
    if (STAMP) {
        if (threadIdx.x == 0) sh.seg[0] = sh.seg[1] = sh.seg[2] = sh.seg[3] = 0;
        __syncthreads();  // diagnostic launches only (mrbf_debug_diag): shader-clock and 100 MHz real-time stamps
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int tid = threadIdx.x, ti = tid & 15, tj = (tid >> 4) & 15, half = tid >> 8;
    double a[8][8];
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ti + 16 * u, j = tj + 16 * v;
            if (half == 0)
                a[u][v] = (i >= j) ? A[i + (int64_t)j * lda] : 0.0;
            else
                a[u][v] = (i == j) ? 1.0 : 0.0;
        }
    int buf = 0, bad = 0;
    // prologue: publish column 0 (owner group tj == 0, pivot in its lane ti == 0)
    if (tj == 0) {
        if (half == 0) {
            const double pv = __shfl(a[0][0], 0, 16);
            double rinv = 0.0;
            if (pv > 0.0) {
                rinv = fast_rsqrt(pv);
            } else {
                bad = 1;
            }
            if (ti == 0) sh.rinv[0] = rinv;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u][0] *= rinv;
                sh.col[0][ti + 16 * u] = a[u][0];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) sh.col[0][CNB + ti + 16 * u] = (u == 0) ? a[0][0] : 0.0;
        }
    }
    diag_phase<0, STAMP>(a, sh, ti, tj, half, buf, bad);
    diag_phase<1, STAMP>(a, sh, ti, tj, half, buf, bad);
    diag_phase<2, STAMP>(a, sh, ti, tj, half, buf, bad);
    diag_phase<3, STAMP>(a, sh, ti, tj, half, buf, bad);
    diag_phase<4, STAMP>(a, sh, ti, tj, half, buf, bad);
    diag_phase<5, STAMP>(a, sh, ti, tj, half, buf, bad);
    diag_phase<6, STAMP>(a, sh, ti, tj, half, buf, bad);
    diag_phase<7, STAMP>(a, sh, ti, tj, half, buf, bad);
    // non-positive pivot: report the 1-based global index of the first one (LAPACK potrf convention)
    if (tid == 0) sh.sbad = 0x7fffffff;
    __syncthreads();
    if (bad) atomicMin(&sh.sbad, bad);
    __syncthreads();
    if (sh.sbad != 0x7fffffff) {
        if (tid == 0) *info = col0 + sh.sbad;
        return;
    }
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ti + 16 * u, j = tj + 16 * v;
            if (half == 0) {
                if (i >= j) A[i + (int64_t)j * lda] = a[u][v];  // the factor, lower part of the block in place
            } else {
                // half 1 holds (L^-T)[r = i][j] = Linv[j][r]: lower-triangular inverse, column-major, zeros above
                Linv[j + i * CNB] = (j >= i) ? a[u][v] : 0.0;
            }
        }
    if (STAMP && threadIdx.x == 0) {
        stamps[0] = __builtin_amdgcn_s_memtime() - t0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
        for (int q = 0; q < 4; ++q) stamps[2 + q] = sh.seg[q];
    }
}

// identity padding of rows/columns [n, npad) of the leading npad x npad block of a column-major matrix (leading dim ld)
__global__ void pad_identity_kernel(double *__restrict__ A, int64_t n, int64_t npad, int64_t ld) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pw = npad - n;
    if (idx >= npad * pw) return;
    const int64_t j = n + idx / npad, i = idx % npad;  // columns n..npad-1 (rows 0..npad-1)
    A[i + j * ld] = (i == j) ? 1.0 : 0.0;
    if (i < n) A[j + i * ld] = 0.0;  // mirrored rows n..npad-1 in columns < n
}

int launch_pad_identity(mrbf_ctx *ctx, double *A, int64_t n, int64_t npad, int64_t ld) {
    if (npad == n) return 0;
    const int64_t cnt = npad * (npad - n);
    hipLaunchKernelGGL(pad_identity_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, A, n, npad, ld);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// diagnostic: run the diagonal-block kernel `reps` times back to back on copies of one 128 x 128 s.p.d. block
int debug_diag(mrbf_ctx *ctx, const double *A128_dev, int reps, float *ms_per_call, unsigned long long *stamps_host) {
    const bool with_stamps = reps > 0;
    if (reps < 0) reps = -reps;
    double *W, *Linv;
    int *dinfo;
    unsigned long long *st;
    MRBF_TRY(get_buf(ctx, S_PHI, (size_t)CNB * CNB * reps, &W));
    MRBF_TRY(get_buf(ctx, S_CHOL_WS, (size_t)CNB * CNB, &Linv));
    MRBF_TRY(get_buf(ctx, S_INFO, (size_t)4, &dinfo));
    MRBF_TRY(get_buf(ctx, S_MISC, (size_t)8, (double **)&st));
    for (int r = 0; r < reps; ++r)
        MRBF_HIP(ctx, hipMemcpyAsync(W + (size_t)r * CNB * CNB, A128_dev, sizeof(double) * CNB * CNB, hipMemcpyDeviceToDevice, ctx->stream));
    MRBF_HIP(ctx, hipMemsetAsync(dinfo, 0, sizeof(int), ctx->stream));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (int r = 0; r < reps; ++r)
        if (ctx->diag_impl != 1)
            launch_diag_v4(ctx, ctx->stream, W + (size_t)r * CNB * CNB, (int64_t)CNB, Linv, dinfo, 0);
        else if (with_stamps)
            hipLaunchKernelGGL(chol_diag_kernel<true>, dim3(1), dim3(512), 0, ctx->stream, W + (size_t)r * CNB * CNB, (int64_t)CNB, Linv,
                               dinfo, 0, st);
        else
            hipLaunchKernelGGL(chol_diag_kernel<false>, dim3(1), dim3(512), 0, ctx->stream, W + (size_t)r * CNB * CNB, (int64_t)CNB, Linv,
                               dinfo, 0, (unsigned long long *)nullptr);
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    float t;
    MRBF_HIP(ctx, hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]));
    *ms_per_call = t / reps;
    if (const char *e = mrbf_env("MRBF_DIAG_DBG")) {
        if (atoi(e) & 4) {  // per-segment cycle counters of the last launch (chol_diag_core.hpp)
            double seg[6];
            MRBF_HIP(ctx, hipMemcpy(seg, Linv, sizeof(seg), hipMemcpyDeviceToHost));
            fprintf(stderr, "diag segments (wave 0, cycles over 8 panels): pre-leaf %.0f leaf %.0f post-leaf %.0f wait-X %.0f deferred %.0f wait-Y %.0f\n",
                    seg[0], seg[1], seg[2], seg[3], seg[4], seg[5]);
        }
    }
    MRBF_HIP(ctx, hipMemcpy(stamps_host, st, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

// C(lower tile pairs of an nt x nt tile grid) -= A B'  with K a multiple of 16 (the projection's rank-2q update
// of Phi reuses the trailing-update kernel)
int launch_update_lower(mrbf_ctx *ctx, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t nt,
                        int K, const double *cvec, double cs, int64_t cn) {
    if (nt <= 0 || K <= 0) return 0;
    hipLaunchKernelGGL((chol_update_kernel<128, UPD_LOWER_SUB>), dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, ctx->stream, A,
                       lda, B, ldb, C, ldc, K, (const int *)nullptr, (int)nt, 0, cvec, cs, cn);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// Blocked right-looking Cholesky of the leading ncols x ncols block of a TALL matrix (mrows >= ncols, both
// multiples of 128, lda even, 16-byte aligned): on return the top square holds L and the rows below hold
// A_below * L^-T.  Appending right-hand sides as extra ROWS therefore yields the forward substitution
// L^-1 b for free, and appending a tall matrix X below its Gram matrix X'X yields the Q factor of X
// (Cholesky-QR).  linv_all (optional) receives the 128 x 128 inverses of all diagonal blocks of L.
int potrf_blocked_tall(mrbf_ctx *ctx, int64_t ncols, int64_t mrows, double *A, int64_t lda, int *dinfo, double *linv_all) {
    if (ncols % CNB != 0 || mrows % CNB != 0 || mrows < ncols || (lda & 1) || (reinterpret_cast<uintptr_t>(A) & 15))
        return fail(ctx, MRBF_EHIP, "potrf_blocked needs 128-padded, 16-byte aligned storage (ncols=%lld mrows=%lld lda=%lld)",
                    (long long)ncols, (long long)mrows, (long long)lda);
    // chol_impl 3 (and the default, 0, from mega_min columns on): the whole factorisation as one persistent launch (chol_mega.hip);
    // chol_impl 2 forces the host-driven launches below
    // (measured, end of round 2: the persistent launch wins from 256 columns on -- n = 256: 0.107 vs 0.142 ms, 384: 0.138 vs 0.242,
    //  8192: 4.4 vs 8.2, 16384: 27 vs 35.6; a single 128-column block is faster as one diagonal kernel + panel launch)
    if ((ctx->chol_impl == 3 || (ctx->chol_impl == 0 && ncols <= ctx->mega_max)) && ncols >= ctx->mega_min)
        return potrf_mega_tall(ctx, ncols, mrows, A, lda, dinfo, linv_all);
    double *Lone = nullptr;
    if (!linv_all) MRBF_TRY(get_buf(ctx, S_CHOL_WS, (size_t)CNB * CNB, &Lone));
    // Look-ahead over two streams.  Panel stream P (high priority): D(j), T(j), U1(j) = update of block column j+1
    // only -- everything step j+1's diagonal block and panel depend on.  Main stream M: U2(j) = update of block
    // columns >= j+2, the bulk of the flops, which therefore runs UNDER D(j+1) / T(j+1).
    //   U2(j) needs T(j)                      : M waits evT
    //   U1(j+1) writes column j+2 like U2(j)  : P waits evU2 before U1(j+1)
    // M = stream of the trailing updates (the CU-masked bulk stream), P = panel stream, S = the caller's stream
    hipStream_t S = ctx->stream, M = ctx->bulk_stream, P = ctx->panel_stream;
    hipEvent_t evStart = ctx->evx[0], evT = ctx->evx[1], evU2 = ctx->evx[2], evEnd = ctx->evx[3];
    // small factorisations run on the caller's stream alone: every cross-stream wait costs ~10 us, more than the
    // look-ahead can win back on a handful of panels
    const bool single = ncols <= 4 * CNB;
    if (single) M = P = S;
    MRBF_HIP(ctx, hipMemsetAsync(dinfo, 0, sizeof(int), S));
    if (!single) {
        MRBF_HIP(ctx, hipEventRecord(evStart, S));
        MRBF_HIP(ctx, hipStreamWaitEvent(P, evStart, 0));
        MRBF_HIP(ctx, hipStreamWaitEvent(M, evStart, 0));
    }
    const int nb = (int)(ncols / CNB);
    bool have_u2 = false;
    // Panel aggregation: the bulk update U2 is issued once per WINDOW of W panels with K = 128 W (W = 4 while the
    // trailing matrix is large: a rank-128 update re-streams the whole trailing matrix from HBM for 16 flop/byte,
    // rank-512 for 64).  Inside a window U1 applies all of the window's finished panels to the next block column.
    int w0 = 0, W = 1;
    for (int j = 0; j < nb; ++j) {
        const int64_t c = (int64_t)j * CNB;
        if (j == w0) {
            const int64_t rem = ncols - c;
            W = (rem > 4800) ? 4 : (rem > 2400 ? 2 : 1);
            if (ctx->chol_window > 0) W = ctx->chol_window;
        }
        const bool last_in_window = (j == w0 + W - 1) || (j == nb - 1);
        const int kpan = (int)((j - w0 + 1) * CNB);        // columns of finished panels in this window
        const int64_t cw = (int64_t)w0 * CNB;               // first column of the window
        double *Ajj = A + c + c * lda;
        double *Linv = linv_all ? linv_all + (size_t)j * CNB * CNB : Lone;
        if (ctx->diag_impl == 1)
            hipLaunchKernelGGL(chol_diag_kernel<false>, dim3(1), dim3(512), 0, P, Ajj, lda, Linv, dinfo, (int)c,
                               (unsigned long long *)nullptr);
        else
            launch_diag_v4(ctx, P, Ajj, lda, Linv, dinfo, (int)c);
        const int64_t m = mrows - c - CNB;  // rows below the diagonal block
        if (m <= 0) break;
        double *A21 = A + (c + CNB) + c * lda;
        // T: A21 <- A21 * Linv'   (each workgroup owns full rows of the 128-wide panel: in place is safe)
        hipLaunchKernelGGL((chol_update_kernel<64, UPD_OVERWRITE>), dim3((unsigned)(m / 64), 1), dim3(256), 0, P, A21, lda, Linv,
                           (int64_t)CNB, A21, lda, CNB, dinfo);
        if (c + CNB >= ncols) break;  // no trailing columns
        if (!single) MRBF_HIP(ctx, hipEventRecord(evT, P));
        // U1: block column j+1 gets every finished panel of the window (K = kpan), all row tiles down to the last extra row
        // (a cross-stream wait costs ~10 us even when the event has long fired: wait once per recorded event)
        if (have_u2 && !single) {
            MRBF_HIP(ctx, hipStreamWaitEvent(P, evU2, 0));
            have_u2 = false;
        }
        const double *Pw = A + (c + CNB) + cw * lda;  // rows from block row j+1 on, columns of the window
        double *A22 = A + (c + CNB) + (c + CNB) * lda;
        hipLaunchKernelGGL((chol_update_kernel<64, UPD_COLUMN_SUB>), dim3((unsigned)(m / 64), 1), dim3(256), 0, P, Pw, lda, Pw, lda,
                           A22, lda, kpan, dinfo);
        if (last_in_window) {
            // U2: block columns >= j+2 get the whole window in one rank-kpan update, in two launches: U2a = the block
            // columns of the NEXT window (all the next window's U1 launches depend on), U2b = everything beyond, which
            // then runs under the next window's D / T / U1 chain.
            const int64_t mt2 = (ncols - c - 2 * CNB) / CNB;
            if (mt2 > 0) {
                if (!single) MRBF_HIP(ctx, hipStreamWaitEvent(M, evT, 0));
                const int64_t mx = (mrows - ncols) / CNB;  // extra row tiles below the square ride in the same launches
                const int64_t rem_next = ncols - c - CNB;
                int Wn = (rem_next > 4800) ? 4 : (rem_next > 2400 ? 2 : 1);
                if (ctx->chol_window > 0) Wn = ctx->chol_window;
                const int64_t na = std::min<int64_t>(Wn, mt2);  // block columns of U2a
                const double *P2 = A + (c + 2 * CNB) + cw * lda;  // window panels, rows from block row j+2 on
                double *C2 = A + (c + 2 * CNB) + (c + 2 * CNB) * lda;
                hipLaunchKernelGGL((chol_update_kernel<128, UPD_LOWER_SUB>),
                                   dim3((unsigned)(na * (na + 1) / 2 + (mt2 - na + mx) * na)), dim3(256), 0, M, P2, lda, P2, lda, C2, lda,
                                   kpan, dinfo, (int)na);
                if (!single) MRBF_HIP(ctx, hipEventRecord(evU2, M));
                have_u2 = true;
                const int64_t mb2 = mt2 - na;  // block columns of U2b
                if (mb2 > 0) {
                    const double *P3 = P2 + na * CNB;
                    double *C3 = C2 + na * CNB + na * CNB * lda;
                    const int64_t nt3 = mb2 * (mb2 + 1) / 2 + mx * mb2;
                    const int64_t g3 = (ctx->bulk_grid > 0 && nt3 > ctx->bulk_grid) ? ctx->bulk_grid : nt3;
                    hipLaunchKernelGGL((chol_update_kernel<128, UPD_LOWER_SUB>), dim3((unsigned)g3), dim3(256), 0, M, P3, lda, P3, lda,
                                       C3, lda, kpan, dinfo, (int)mb2, (int)nt3);
                }
            }
            w0 = j + 1;
        }
    }
    MRBF_HIP(ctx, hipGetLastError());
    // join: later work on the caller's stream must see both side streams' results
    if (!single) {
        MRBF_HIP(ctx, hipEventRecord(evEnd, P));
        MRBF_HIP(ctx, hipStreamWaitEvent(S, evEnd, 0));
        MRBF_HIP(ctx, hipEventRecord(evStart, M));
        MRBF_HIP(ctx, hipStreamWaitEvent(S, evStart, 0));
    }
    return 0;
}

int potrf_blocked(mrbf_ctx *ctx, int64_t n, double *A, int64_t lda, int *dinfo) {
    return potrf_blocked_tall(ctx, n, n, A, lda, dinfo, nullptr);
}

// ---- backward substitution  L' x = y  for k right-hand sides, using the stored diagonal-block inverses ----------
// Y: npad x k column-major (ld npad), overwritten by x.  Launch i (i = nb-1 .. 0) has i + 1 workgroups: workgroup
// j < i applies  y_j -= L(i,j)' x_i ; workgroup j == i - 1 then finishes  x_{i-1} = Linv_{i-1}' y_{i-1}  (all later
// updates of y_{i-1} were applied by earlier launches).  A prologue launch computes x_{nb-1}.
// res[l][c] = sum_r T(r, c) x_l[r] for one 128 x 128 column-major tile.  The tile is staged through LDS in two
// 64-row halves with coalesced 512-byte column reads (all loads of a half in flight together); thread (c, g)
// then sums rows 32g..32g+31 of column c straight from LDS (stride 65: conflict-free), and the two partial sums
// per column meet in LDS.  No cross-lane shuffles: 64 dependent shuffle chains per tile cost ~16 us.
constexpr int BS_LD = 65;
template <int KB>
__device__ __forceinline__ void tile_tdot(const double *__restrict__ Tl, int64_t ldt, const double *xs /* LDS KB x 128 */,
                                          double *res /* LDS KB x 128 */, double *stage /* LDS 128 x 65 */,
                                          double *part /* LDS KB x 256 */) {
    const int tid = threadIdx.x, c = tid & 127, g = tid >> 7;
    double acc[KB];
#pragma unroll
    for (int l = 0; l < KB; ++l) acc[l] = 0.0;
    for (int h = 0; h < 2; ++h) {
        double v[32];
#pragma unroll
        for (int p = 0; p < 32; ++p) v[p] = Tl[(int64_t)(p * 4 + (tid >> 6)) * ldt + 64 * h + (tid & 63)];
        __syncthreads();  // previous half consumed
#pragma unroll
        for (int p = 0; p < 32; ++p) stage[(p * 4 + (tid >> 6)) * BS_LD + (tid & 63)] = v[p];
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const double t = stage[c * BS_LD + 32 * g + r];
#pragma unroll
            for (int l = 0; l < KB; ++l) acc[l] = fma(t, xs[l * CNB + 64 * h + 32 * g + r], acc[l]);
        }
    }
#pragma unroll
    for (int l = 0; l < KB; ++l) part[l * 256 + tid] = acc[l];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int l = 0; l < KB; ++l) res[l * CNB + c] = part[l * 256 + c] + part[l * 256 + 128 + c];
    }
}

template <int KB>
__global__ __launch_bounds__(256) void chol_backsolve_kernel(const double *__restrict__ L, int64_t lda, const double *__restrict__ linv_all,
                                                             double *__restrict__ Y, int64_t ldy, int k0, int i /* block row */,
                                                             int prologue) {
    __shared__ double xs[KB * CNB];
    __shared__ double res[KB * CNB];
    __shared__ double part[KB * 256];
    __shared__ double stage[CNB * BS_LD];
    const int tid = threadIdx.x;
    const int j = blockIdx.x;
    if (!prologue) {
        // x_i (already final) -> LDS
        for (int t = tid; t < KB * CNB; t += 256) xs[t] = Y[(int64_t)(k0 + t / CNB) * ldy + (int64_t)i * CNB + (t % CNB)];
        __syncthreads();
        tile_tdot<KB>(L + (int64_t)i * CNB + (int64_t)j * CNB * lda, lda, xs, res, stage, part);  // tile L(i, j)
        __syncthreads();
        for (int t = tid; t < KB * CNB; t += 256) {
            double *yp = Y + (int64_t)(k0 + t / CNB) * ldy + (int64_t)j * CNB + (t % CNB);
            const double v = *yp - res[t];
            *yp = v;
            xs[t] = v;  // y_j after this update, kept for the finishing step below
        }
        if (j != i - 1) return;
        __syncthreads();
    } else {
        for (int t = tid; t < KB * CNB; t += 256) xs[t] = Y[(int64_t)(k0 + t / CNB) * ldy + (int64_t)i * CNB + (t % CNB)];
        __syncthreads();
    }
    // finish block jb: x = Linv' y  (Linv lower triangular, zeros above the diagonal are stored)
    const int jb = prologue ? i : i - 1;
    tile_tdot<KB>(linv_all + (size_t)jb * CNB * CNB, CNB, xs, res, stage, part);
    __syncthreads();
    for (int t = tid; t < KB * CNB; t += 256) Y[(int64_t)(k0 + t / CNB) * ldy + (int64_t)jb * CNB + (t % CNB)] = res[t];
}

int backsolve_blocked(mrbf_ctx *ctx, int64_t npad, const double *L, int64_t lda, const double *linv_all, double *Y, int64_t ldy, int k) {
    const int nb = (int)(npad / CNB);
    for (int k0 = 0; k0 < k; k0 += 4) {
        const int kb = std::min(4, k - k0);
#define MRBF_BS(KBV, grid, ii, pro)                                                                                         \
    hipLaunchKernelGGL((chol_backsolve_kernel<KBV>), dim3((unsigned)(grid)), dim3(256), 0, ctx->stream, L, lda, linv_all, Y, ldy, \
                       k0, ii, pro)
        for (int i = nb - 1; i >= 0; --i) {
            const bool pro = (i == nb - 1);
            if (pro) {
                if (kb == 1) MRBF_BS(1, 1, i, 1); else if (kb == 2) MRBF_BS(2, 1, i, 1); else if (kb == 3) MRBF_BS(3, 1, i, 1); else MRBF_BS(4, 1, i, 1);
            }
            if (i > 0) {
                if (kb == 1) MRBF_BS(1, i, i, 0); else if (kb == 2) MRBF_BS(2, i, i, 0); else if (kb == 3) MRBF_BS(3, i, i, 0); else MRBF_BS(4, i, i, 0);
            }
        }
#undef MRBF_BS
    }
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf
