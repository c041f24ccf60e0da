// Gram-matrix assembly  Phi[i,j] = phi(||c_i - c_j||)  -- replaces RBF.get_matrices /
// the assembly inside RBF.RBFInterpolationModel (/root/reference/src/models/RbfModel.jl:374-375, :759-763).
//
// Two kernels:
//  * gram_mfma_kernel  (default): GEMM form  s = |x_i|^2 + |x_j|^2 - 2 x_i.x_j  on CENTRED coordinates,
//    the inner products on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), radial function fused in the
//    epilogue, each lower-triangular 128x128 tile pair computed once and written twice (tile + mirrored
//    tile through an LDS transpose).  HBM-write bound at d <= ~96 (8 n^2 bytes), MFMA bound above.
//  * gram_diff_kernel  (MRBF_OPT_GRAM_MODE = 1): difference form  sum (x - c)^2  on the VALU, the
//    reference's own arithmetic (norm(x - c)); used for parity studies, not for speed.
#include "radial.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int GBM = 128;      // tile edge
constexpr int GBK = 16;       // k chunk staged in LDS
constexpr int GLD = GBK + 2;  // LDS row stride (doubles): 144 B rows -> conflict-free ds_read_b64 fragments

__device__ __forceinline__ void tri_decode(int bid, int &ti, int &tj) {
    int t = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) / 2 <= bid) ++t;
    while (t * (t + 1) / 2 > bid) --t;
    ti = t;
    tj = bid - t * (t + 1) / 2;
}

// Xc: npad x dpad centred + zero padded (npad % 128 == 0, dpad % 16 == 0); sq: npad
template <int KID, bool FAST, int OCC>
__global__ __launch_bounds__(256, OCC) void gram_mfma_kernel(const double *__restrict__ Xc, const double *__restrict__ sq,
                                                           int64_t n, int dpad, double *__restrict__ Phi, int64_t ld,
                                                           KP p, int aligned16) {
    __shared__ __attribute__((aligned(16))) double smem[2 * GBM * GLD];
    double *As = smem;
    double *Bs = smem + GBM * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    int ti, tj;
    tri_decode(blockIdx.x, ti, tj);
    const int64_t I0 = (int64_t)ti * GBM, J0 = (int64_t)tj * GBM;
    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int lr = tid >> 3, lc = (tid & 7) * 2;
    const double *Ap = Xc + (I0 + lr) * dpad + lc;
    const double *Bp = Xc + (J0 + lr) * dpad + lc;
    v2d ra[4], rb[4];
    if (OCC <= 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = *(const v2d *)(Ap + (int64_t)(32 * u) * dpad);
            rb[u] = *(const v2d *)(Bp + (int64_t)(32 * u) * dpad);
        }
    }
    const int nkc = dpad / GBK;
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
        if (OCC > 2) {  // no register prefetch: three workgroups per CU hide the load latency instead
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ra[u] = *(const v2d *)(Ap + (int64_t)(32 * u) * dpad + kc * GBK);
                rb[u] = *(const v2d *)(Bp + (int64_t)(32 * u) * dpad + kc * GBK);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *(v2d *)&As[(lr + 32 * u) * GLD + lc] = ra[u];
            *(v2d *)&Bs[(lr + 32 * u) * GLD + lc] = rb[u];
        }
        __syncthreads();
        if (OCC <= 2 && kc + 1 < nkc) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ra[u] = *(const v2d *)(Ap + (int64_t)(32 * u) * dpad + (kc + 1) * GBK);
                rb[u] = *(const v2d *)(Bp + (int64_t)(32 * u) * dpad + (kc + 1) * GBK);
            }
        }
#pragma unroll
        for (int kk = 0; kk < GBK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(wr * 64 + i * 16 + l15) * GLD + kk * 4 + l4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[(wc * 64 + j * 16 + l15) * GLD + kk * 4 + l4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: s = |xi|^2 + |xj|^2 - 2 g, radial function, direct store of block (I,J)
    // f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    double sqj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sqj[j] = sq[J0 + wc * 64 + j * 16 + l15];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = I0 + wr * 64 + i * 16 + l4 + 4 * r;
            const double sqi = sq[gi];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t gj = J0 + wc * 64 + j * 16 + l15;
                double s = fma(-2.0, acc[i][j][r], sqi + sqj[j]);
                s = s > 0.0 ? s : 0.0;
                if (gi == gj) s = 0.0;
                const double v = rbf_phi_t<KID, FAST>(s, p);
                acc[i][j][r] = v;
                if (gi < n && gj < n) Phi[gi * ld + gj] = v;
            }
        }
    }
    if (ti == tj) return;  // diagonal tile: the full square was computed, nothing to mirror

    // ---- mirrored block (J,I) through an LDS transpose, 16-row strips, 128-byte row segments
    double *T = smem + wave * (64 * GLD);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(j * 16 + l15) * GLD + l4 + 4 * r] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int jl = it * 8 + (lane >> 3), il = 2 * (lane & 7);
            const v2d v = *(const v2d *)&T[jl * GLD + il];
            const int64_t gj = J0 + wc * 64 + jl;
            const int64_t gi = I0 + wr * 64 + i * 16 + il;
            if (gj < n) {
                double *dst = Phi + gj * ld + gi;
                if (gi + 1 < n) {
                    if (aligned16) {
                        *(v2d *)dst = v;
                    } else {
                        dst[0] = v.x;
                        dst[1] = v.y;
                    }
                } else if (gi < n) {
                    dst[0] = v.x;
                }
            }
        }
    }
}

// ---- reference-arithmetic kernel: difference form on the VALU, 64x64 tiles, 4x4 per thread
template <int KID>
__global__ __launch_bounds__(256) void gram_diff_kernel(const double *__restrict__ C, int64_t n, int d,
                                                        double *__restrict__ Phi, int64_t ld, KP p) {
    __shared__ double As[64][17];
    __shared__ double Bs[64][17];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    int ti, tj;
    tri_decode(blockIdx.x, ti, tj);
    const int64_t I0 = (int64_t)ti * 64, J0 = (int64_t)tj * 64;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < d; k0 += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = ty + 16 * u, c = tx;
            const bool kin = (k0 + c) < d;
            As[r][c] = (I0 + r < n && kin) ? C[(I0 + r) * d + k0 + c] : 0.0;
            Bs[r][c] = (J0 + r < n && kin) ? C[(J0 + r) * d + k0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = As[ty + 16 * u][kk];
#pragma unroll
            for (int v = 0; v < 4; ++v) b[v] = Bs[tx + 16 * v][kk];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double df = a[u] - b[v];
                    acc[u][v] = fma(df, df, acc[u][v]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t gi = I0 + ty + 16 * u, gj = J0 + tx + 16 * v;
            if (gi < n && gj < n) {
                const double val = rbf_phi<KID>(gi == gj ? 0.0 : acc[u][v], p);
                Phi[gi * ld + gj] = val;
                if (ti != tj) Phi[gj * ld + gi] = val;
            }
        }
}

// ---- rectangular kernel block  K[i][j] = phi(||x_i - c_j||), difference form (the reference's kernels(xi) call of round 4,
// RbfModel.jl:421, for all candidates at once).  X: m x d, C: n x d row-major; K: m x n row-major.
template <int KID>
__global__ __launch_bounds__(256) void cross_gram_kernel(const double *__restrict__ X, int64_t m, const double *__restrict__ C, int64_t n,
                                                         int d, double *__restrict__ K, KP p) {
    __shared__ double As[64][17];
    __shared__ double Bs[64][17];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t I0 = (int64_t)blockIdx.y * 64, J0 = (int64_t)blockIdx.x * 64;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < d; k0 += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = ty + 16 * u, c = tx;
            const bool kin = (k0 + c) < d;
            As[r][c] = (I0 + r < m && kin) ? X[(I0 + r) * d + k0 + c] : 0.0;
            Bs[r][c] = (J0 + r < n && kin) ? C[(J0 + r) * d + k0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = As[ty + 16 * u][kk];
#pragma unroll
            for (int v = 0; v < 4; ++v) b[v] = Bs[tx + 16 * v][kk];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double df = a[u] - b[v];
                    acc[u][v] = fma(df, df, acc[u][v]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t gi = I0 + ty + 16 * u, gj = J0 + tx + 16 * v;
            if (gi < m && gj < n) K[gi * n + gj] = rbf_phi<KID>(acc[u][v], p);
        }
}

int launch_cross_gram(mrbf_ctx *ctx, const double *X, int64_t m, const double *C, int64_t n, int d, const KP &kp, double *K) {
    if (m <= 0 || n <= 0) return 0;
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 63) / 64));
    MRBF_DISPATCH_KID(kp.kid, hipLaunchKernelGGL((cross_gram_kernel<KID>), grid, dim3(256), 0, ctx->stream, X, m, C, n, d, K, kp));
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- 64-row variant: four workgroups per CU --------------------------------------------------------------------------
// A wave that issues stores into a saturated memory path stalls at issue, so a workgroup's store phase only overlaps with
// OTHER workgroups' MFMA / radial-function phases.  With 128 x 128 tiles (128 accumulator registers)
// two workgroups fit on a CU and the kernel reaches 3.5 TB/s; a pure-store kernel with the same tiling writes 5.3-5.5 TB/s.
// Here a workgroup computes rows [64 h, 64 h + 64) of tile pair (ti, tj) -- 64 accumulator registers, four workgroups per
// CU -- so that some workgroup is storing at (almost) any time.  Wave w owns columns 32 w .. 32 w + 31 of the tile.
template <int KID, bool FAST>
__global__ __launch_bounds__(256, 4) void gram_mfma64_kernel(const double *__restrict__ Xc, const double *__restrict__ sq, int64_t n,
                                                             int dpad, double *__restrict__ Phi, int64_t ld, KP p, int aligned16, int npairs,
                                                             int remap) {
    __shared__ __attribute__((aligned(16))) double smem[(64 + GBM) * GLD];
    double *As = smem;             // 64 rows
    double *Bs = smem + 64 * GLD;  // 128 rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8): keep the two halves of a tile pair, which stage the same
    // 128 centre rows J, on ONE XCD's L2 (blocks b and b + 8), and give an XCD every eighth pair of a tile row (shared rows I)
    int ti, tj;
    const int xb = blockIdx.x & 7, grp = blockIdx.x >> 3;
    int pair = xb + 8 * (grp >> 1), half = grp & 1;
    if (remap == 0 || pair >= npairs) {  // tail of a grid that is not a multiple of 16: the plain order
        pair = blockIdx.x >> 1;
        half = blockIdx.x & 1;
    }
    tri_decode(pair, ti, tj);
    const int64_t I0 = (int64_t)ti * GBM + 64 * half, J0 = (int64_t)tj * GBM;
    v4d acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int lr = tid >> 3, lc = (tid & 7) * 2;  // 32 rows x 8 column pairs per pass
    const double *Ap = Xc + (I0 + lr) * dpad + lc;
    const double *Bp = Xc + (J0 + lr) * dpad + lc;
    v2d ra[2], rb[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) ra[u] = *(const v2d *)(Ap + (int64_t)(32 * u) * dpad);
#pragma unroll
    for (int u = 0; u < 4; ++u) rb[u] = *(const v2d *)(Bp + (int64_t)(32 * u) * dpad);
    const int nkc = dpad / GBK;
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) *(v2d *)&As[(lr + 32 * u) * GLD + lc] = ra[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) *(v2d *)&Bs[(lr + 32 * u) * GLD + lc] = rb[u];
        __syncthreads();
        if (kc + 1 < nkc) {
#pragma unroll
            for (int u = 0; u < 2; ++u) ra[u] = *(const v2d *)(Ap + (int64_t)(32 * u) * dpad + (kc + 1) * GBK);
#pragma unroll
            for (int u = 0; u < 4; ++u) rb[u] = *(const v2d *)(Bp + (int64_t)(32 * u) * dpad + (kc + 1) * GBK);
        }
#pragma unroll
        for (int kk = 0; kk < GBK / 4; ++kk) {
            double a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(i * 16 + l15) * GLD + kk * 4 + l4];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(wave * 32 + j * 16 + l15) * GLD + kk * 4 + l4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue (f64 MFMA C/D layout: col = lane & 15 -> j, row = (lane >> 4) + 4 reg -> i)
    double sqj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) sqj[j] = sq[J0 + wave * 32 + j * 16 + l15];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = I0 + i * 16 + l4 + 4 * r;
            const double sqi = sq[gi];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t gj = J0 + wave * 32 + j * 16 + l15;
                double s = fma(-2.0, acc[i][j][r], sqi + sqj[j]);
                s = s > 0.0 ? s : 0.0;
                if (gi == gj) s = 0.0;
                const double v = rbf_phi_t<KID, FAST>(s, p);
                acc[i][j][r] = v;
                if (gi < n && gj < n) Phi[gi * ld + gj] = v;
            }
        }
    }
    if (ti == tj) return;  // diagonal tile: the full square was computed, nothing to mirror
    // mirrored block through an LDS transpose: wave-private 32 x 16 strips
    double *T = smem + wave * (32 * GLD);
    __syncthreads();  // every wave has finished reading As / Bs
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(j * 16 + l15) * GLD + l4 + 4 * r] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); the strip is private to the wave and LDS operations of a wave are in order
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int jl = it * 8 + (lane >> 3), il = 2 * (lane & 7);
            const v2d v = *(const v2d *)&T[jl * GLD + il];
            const int64_t gj = J0 + wave * 32 + jl;
            const int64_t gi = I0 + i * 16 + il;
            if (gj < n) {
                double *dst = Phi + gj * ld + gi;
                if (gi + 1 < n) {
                    if (aligned16) {
                        *(v2d *)dst = v;
                    } else {
                        dst[0] = v.x;
                        dst[1] = v.y;
                    }
                } else if (gi < n) {
                    dst[0] = v.x;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // reads of the strip done before the next pass overwrites it
    }
}

int launch_gram(mrbf_ctx *ctx, int mode, const double *C, const double *Xc, const double *sq, int64_t n, int64_t npad,
                int d, int dpad, const KP &kp, double *Phi, int64_t ld) {
    if (n <= 0) return 0;
    if (mode == 1) {
        const int64_t nt = (n + 63) / 64;
        const int64_t nb = nt * (nt + 1) / 2;
        MRBF_DISPATCH_KID(kp.kid, hipLaunchKernelGGL((gram_diff_kernel<KID>), dim3((unsigned)nb), dim3(256), 0,
                                                     ctx->stream, C, n, d, Phi, ld, kp));
    } else {
        if (npad % GBM != 0 || dpad % GBK != 0) return fail(ctx, MRBF_EHIP, "gram: bad padding npad=%lld dpad=%d", (long long)npad, dpad);
        const int64_t nt = (n + GBM - 1) / GBM;  // tiles that hold at least one real row
        const int64_t nb = nt * (nt + 1) / 2;
        const int aligned16 = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(Phi) & 15) == 0);
        static const int occ3 = mrbf_env("MRBF_GRAM_OCC3") ? atoi(mrbf_env("MRBF_GRAM_OCC3")) : 0;
        static const int rows64 = mrbf_env("MRBF_GRAM_ROWS64") ? atoi(mrbf_env("MRBF_GRAM_ROWS64")) : 1;
        if (kp.fast && rows64 && nb >= 512) {
            // the remap needs whole groups of 16 blocks (8 pairs x 2 halves); a ragged tail falls back to the plain order inside the kernel
            static const int remap_env = mrbf_env("MRBF_GRAM_REMAP") ? atoi(mrbf_env("MRBF_GRAM_REMAP")) : 1;
            const int remap = (remap_env && nb % 8 == 0) ? 1 : 0;
            MRBF_DISPATCH_KID(kp.kid, hipLaunchKernelGGL((gram_mfma64_kernel<KID, true>), dim3((unsigned)(2 * nb)), dim3(256), 0, ctx->stream,
                                                         Xc, sq, n, dpad, Phi, ld, kp, aligned16, (int)nb, remap));
        } else if (kp.fast && occ3) {
            MRBF_DISPATCH_KID(kp.kid, hipLaunchKernelGGL((gram_mfma_kernel<KID, true, 3>), dim3((unsigned)nb), dim3(256), 0,
                                                         ctx->stream, Xc, sq, n, dpad, Phi, ld, kp, aligned16));
        } else if (kp.fast) {
            MRBF_DISPATCH_KID(kp.kid, hipLaunchKernelGGL((gram_mfma_kernel<KID, true, 2>), dim3((unsigned)nb), dim3(256), 0,
                                                         ctx->stream, Xc, sq, n, dpad, Phi, ld, kp, aligned16));
        } else {
            MRBF_DISPATCH_KID(kp.kid, hipLaunchKernelGGL((gram_mfma_kernel<KID, false, 2>), dim3((unsigned)nb), dim3(256), 0,
                                                         ctx->stream, Xc, sq, n, dpad, Phi, ld, kp, aligned16));
        }
    }
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- debug: one f64 MFMA on exact integer data, to pin the operand / result lane maps in a test
__global__ void mfma_layout_kernel(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ D) {
    const int lane = threadIdx.x & 63;
    // A is 16x4 row-major, B is 4x16 row-major
    const double a = A[(lane & 15) * 4 + (lane >> 4)];
    const double b = B[(lane >> 4) * 16 + (lane & 15)];
    v4d c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = c[r];
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_debug_mfma_layout(mrbf_ctx *ctx, double *out, const double *A, const double *B) {
    if (!ctx) return -1;
    (void)hipSetDevice(ctx->device);
    const double *dA, *dB;
    double *dD;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, A, 64, &dA));
    MRBF_TRY(stage_in(ctx, S_STAGE_B, B, 64, &dB));
    MRBF_TRY(stage_out(ctx, S_OUT_A, out, 256, &dD));
    hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, ctx->stream, dA, dB, dD);
    MRBF_HIP(ctx, hipGetLastError());
    MRBF_TRY(finish_out(ctx, out, dD, 256));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MRBF_OK;
}
