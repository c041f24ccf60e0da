// Fit-side Gram assembly fused with the first product of the projection -- replaces, inside mrbf_fit, the pair
// "assemble Phi in full, read it back for W = Phi Q1" of the assembly + dense solve of RBF.RBFInterpolationModel
// (/root/reference/src/models/RbfModel.jl:759-763; RBF.get_matrices :374-375 keeps the full-matrix kernel of gram.hip).
//
//   gram_w_kernel      d <= 64.  A workgroup owns a 128-row strip I of Phi and walks a segment of column tiles J in 64-row
//                      half steps: G' = Xc_J Xc_I' on the f64 matrix cores (the strip's operands stay in registers for the
//                      whole walk), radial function in registers, the tile is stored ONLY when it lies in the lower triangle
//                      (J <= I: 4 n^2 bytes instead of 8 n^2, and nobody reads them back for the projection), and
//                      W'_I += Phi(I, J) Xc_J is accumulated straight from the accumulator registers -- the C/D layout of the
//                      f64 MFMA (row = 4 r + lane / 16) IS the B-operand layout of a k-slice, so no transposition is needed.
//                      The row sums (the product with the constant column of the tail basis) are taken along the way.
//                      Per strip and segment one partial panel; fixed-order reduction afterwards (bit-reproducible).
//                      Price: the mirrored tiles are computed twice (n^2 d more flops than the triangular assembly), which is
//                      what keeps every contribution to W_I inside one workgroup; the kernel is MFMA-bound
//                      (2 n^2 d + 2 n^2 d flops against 4 n^2 + 8 n d bytes: 31 flop / byte at d = 64).
//   proj_reduce_kernel W' -> W1 = Phi Q1 = [rowsum / sqrt n | W' Lx^-T] (Q1 = [1/sqrt n | Xc Lx^-T], Lx the Cholesky-QR factor)
//                      and the row-chunk partials of G = Q1' W1
//   proj_shift_kernel  G, trace, mu = (n phi0 - trace) / (n - q), M = G / 2 + mu / 2 I
//   proj_panels_kernel V = W1 - Q1 M and the operand panels [Q | V], [V | Q] of the rank-2q update K = Phi - Q1 V' - V Q1'
// (three launches where solve.hip's general path needs eight: reduce, G partial + reduce, trace, dgemm, shift, panels).
#include "radial.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int GW_LDT = 66;  // LDS row stride of a staged 64 x 64 block of centres (doubles)

template <int KID, bool FAST>
__global__ __launch_bounds__(256, 2) void gram_w_kernel(const double *__restrict__ Xc, const double *__restrict__ sq, int64_t n,
                                                        int64_t npad, double *__restrict__ Phi, int64_t ld, KP p, int nt, int nseg,
                                                        int seglen, double *__restrict__ Wpart, double *__restrict__ rspart, int dbg) {
    __shared__ __attribute__((aligned(16))) double T[2][64 * GW_LDT];
    __shared__ double Tsq[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int seg = blockIdx.x % nseg, I = nt - 1 - (int)(blockIdx.x / nseg);
    const int jt0 = seg * seglen;
    const int jt1 = (jt0 + seglen < nt) ? jt0 + seglen : nt;
    const int64_t iw0 = (int64_t)I * 128 + 32 * wave;  // this wave's 32 columns i of the strip
    // loop-invariant B operands of G' = Xc_J Xc_I': B[k = 4 ks + l4][n = i]
    double bI[2][16];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) bI[it][ks] = Xc[(iw0 + it * 16 + l15) * 64 + ks * 4 + l4];
    double sqi[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) sqi[it] = sq[iw0 + it * 16 + l15];
    v4d wacc[4][2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int it = 0; it < 2; ++it) wacc[tt][it] = (v4d){0.0, 0.0, 0.0, 0.0};
    double rs[2] = {0.0, 0.0};
    const int nh = 2 * (jt1 - jt0);  // half steps of 64 rows j
    if (nh > 0) {
        const double *src = Xc + (int64_t)jt0 * 128 * 64;  // a half tile of centres is 64 x 64 contiguous doubles
        const double *ssq = sq + (int64_t)jt0 * 128;
        v2d pf[4];
        double pfsq = 0.0;
        if (tid < 64) pfsq = ssq[tid];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pf[u] = *(const v2d *)(src + (tid + 256 * (4 * half + u)) * 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = (tid + 256 * (4 * half + u)) * 2;
                *(v2d *)&T[0][(e >> 6) * GW_LDT + (e & 63)] = pf[u];
            }
        }
        if (tid < 64) Tsq[0][tid] = pfsq;
        __syncthreads();
#pragma unroll 1
        for (int h = 0; h < nh; ++h) {
            const int buf = h & 1;
            const double *Tb = T[buf];
            const int64_t jb = (int64_t)jt0 * 128 + 64 * h;
            const bool store = (jt0 + (h >> 1)) <= I && !(dbg & 1);
            const bool full = jb + 64 <= n && (int64_t)I * 128 + 128 <= n;
            const bool edge = jb + 64 > n || (jt0 + (h >> 1)) == I;  // ragged rows or the diagonal tile
            const bool more = h + 1 < nh && !(dbg & 16);
            const double *s2 = src + (int64_t)(h + 1) * 64 * 64;
            if (more && tid < 64) pfsq = ssq[64 * (h + 1) + tid];
            // two sub-blocks of 32 rows j each (a rolled loop: 32 accumulator registers for G' instead of 64 keep the kernel at two
            // workgroups per compute unit without spills)
#pragma unroll 1
            for (int sub = 0; sub < 2; ++sub) {
                const double *Ts = Tb + sub * 32 * GW_LDT;
                // prefetch of the next half tile of centres, one half of it per sub-block (8 registers in flight, not 16)
                if (more) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) pf[u] = *(const v2d *)(s2 + (tid + 256 * (4 * sub + u)) * 2);
                }
                // ---- G' (rows j of the sub-block, columns i of the strip)
                v4d acc[2][2];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int it = 0; it < 2; ++it) acc[jt][it] = (v4d){0.0, 0.0, 0.0, 0.0};
                if (!(dbg & 8))
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    double a[2];
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) a[jt] = Ts[(jt * 16 + l15) * GW_LDT + ks * 4 + l4];
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                        for (int it = 0; it < 2; ++it)
                            acc[jt][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt], bI[it][ks], acc[jt][it], 0, 0, 0);
                }
                // the next half tile of centres goes to the other buffer (its readers finished before the barrier that ended the
                // previous step)
                if (more) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = (tid + 256 * (4 * sub + u)) * 2;
                        *(v2d *)&T[buf ^ 1][(e >> 6) * GW_LDT + (e & 63)] = pf[u];
                    }
                    if (sub == 0 && tid < 64) Tsq[buf ^ 1][tid] = pfsq;
                }
                // ---- radial function (C/D layout: column i = lane & 15, row j = lane / 16 + 4 r).  Tiles away from the diagonal and
                // from the ragged edge (uniform per step) skip the per-element tests
                if (edge || !FAST) {  // (the exp / log kernels keep ONE copy of the radial function: registers)
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int jl = sub * 32 + jt * 16 + l4 + 4 * r;
                            const int64_t gj = jb + jl;
                            const double sqj = Tsq[buf][jl];
#pragma unroll
                            for (int it = 0; it < 2; ++it) {
                                const int64_t gi = iw0 + it * 16 + l15;
                                double s = fma(-2.0, acc[jt][it][r], sqi[it] + sqj);
                                s = s > 0.0 ? s : 0.0;
                                if (gi == gj) s = 0.0;
                                double v = (dbg & 2) ? s : rbf_phi_t<KID, FAST>(s, p);
                                v = gj < n ? v : 0.0;  // padded rows take no part in the products
                                acc[jt][it][r] = v;
                                rs[it] += v;
                            }
                        }
                } else {
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double sqj = Tsq[buf][sub * 32 + jt * 16 + l4 + 4 * r];
#pragma unroll
                            for (int it = 0; it < 2; ++it) {
                                double s = fma(-2.0, acc[jt][it][r], sqi[it] + sqj);
                                s = s > 0.0 ? s : 0.0;
                                const double v = (dbg & 2) ? s : rbf_phi_t<KID, FAST>(s, p);
                                acc[jt][it][r] = v;
                                rs[it] += v;
                            }
                        }
                }
                if (store) {  // uniform per step; tiles away from the ragged edge store without per-element tests
                    double *dst = Phi + (jb + sub * 32 + l4) * ld + iw0 + l15;
                    if (full) {
#pragma unroll
                        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int it = 0; it < 2; ++it) dst[(int64_t)(jt * 16 + 4 * r) * ld + it * 16] = acc[jt][it][r];
                    } else {
#pragma unroll
                        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int it = 0; it < 2; ++it) {
                                    const int64_t gj = jb + sub * 32 + jt * 16 + l4 + 4 * r, gi = iw0 + it * 16 + l15;
                                    if (gi < n && gj < n) dst[(int64_t)(jt * 16 + 4 * r) * ld + it * 16] = acc[jt][it][r];
                                }
                    }
                }
                // ---- W'_I' (t x i) += Xc_J' (t x j) Phi' (j x i): the accumulator registers are the B operand
                if (!(dbg & 4))
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double aw[4];
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) aw[tt] = Ts[(jt * 16 + 4 * r + l4) * GW_LDT + tt * 16 + l15];
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                            for (int it = 0; it < 2; ++it)
                                wacc[tt][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[tt], acc[jt][it][r], wacc[tt][it], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---- partial panel of this (strip, segment): Wpart[seg][t][i], rspart[seg][i]
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = tt * 16 + l4 + 4 * r;
#pragma unroll
            for (int it = 0; it < 2; ++it) Wpart[((int64_t)seg * 64 + t) * npad + iw0 + it * 16 + l15] = wacc[tt][it][r];
        }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        double x = rs[it];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        if (l4 == 0) rspart[(int64_t)seg * npad + iw0 + it * 16 + l15] = x;
    }
}

// fixed-order sum of `cnt` partials spaced `stride` apart: batches of 32 loads in flight, eight interleaved chains
__device__ __forceinline__ double sum_partials(const double *__restrict__ p, int cnt, int64_t stride) {
    double sp[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int c = 0;
    for (; c + 32 <= cnt; c += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = p[(int64_t)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < 32; ++u) sp[u & 7] += v[u];
    }
    for (; c + 8 <= cnt; c += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sp[u] += v[u];
    }
    for (int u = 0; c + u < cnt; ++u) sp[u] += p[(int64_t)(c + u) * stride];
    return ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
}

// ---- the small products of the projection on the matrix cores.  One workgroup = 64 rows of the n x q panels; operands in LDS as
// [column][row] with a stride of 68 doubles (k-adjacent lanes 4-way apart: the f64 A / B fragment reads are conflict free).
// A fragment: lane (l & 15 -> m, l >> 4 -> k); B fragment: lane (l >> 4 -> k, l & 15 -> n); D: row = l / 16 + 4 r, col = l & 15.
constexpr int PJ_LS = 68;
constexpr int PJ_QP = 80;  // q <= 65 padded to five 16-wide tiles

// D (16 x 16 tile (mt, nt)) = sum_k A[m][k] B[n][k] over K (multiple of 4): A at As[m * PJ_LS + k], B at Bs[n * PJ_LS + k]
__device__ __forceinline__ v4d pj_tile(const double *As, const double *Bs, int mt, int nt, int K, int l15, int l4) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    const double *ap = As + (mt * 16 + l15) * PJ_LS + l4;
    const double *bp = Bs + (nt * 16 + l15) * PJ_LS + l4;
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k], bp[k], acc, 0, 0, 0);
    return acc;
}

// rows [64 b, 64 b + 64): W1 (npad x q, column-major) = [rowsum / sqrt n | W' Lx^-T] and Gp[b] = Q1(rows)' W1(rows) (q x q).
// linv: inverse of the Cholesky-QR factor Lx, 128 x 128 column-major, zero above the diagonal (q > 1 only).
// All global loads of a phase are issued before their first use (one wave per SIMD: a dependent load is a ~2 us round trip).
__global__ __launch_bounds__(256) void proj_reduce_kernel(const double *__restrict__ Wpart, const double *__restrict__ rspart, int nseg,
                                                          int64_t npad, int64_t n, int d, int q, const double *__restrict__ linv,
                                                          const double *__restrict__ Q1, double rsqrtn, double *__restrict__ W1,
                                                          double *__restrict__ Gp, double *__restrict__ trp) {
    __shared__ double Ws[64 * PJ_LS];      // [i][t]   W' rows of this chunk
    __shared__ double Ls[64 * PJ_LS];      // [c][t]   inv(Lx)
    __shared__ double W1s[PJ_QP * PJ_LS];  // [b][i]
    __shared__ double Qs[PJ_QP * PJ_LS];   // [a][i]
    const int tid = threadIdx.x, i = tid & 63, g = tid >> 6;
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int64_t R0 = (int64_t)blockIdx.x * 64;
    const bool live = R0 + i < n;
    constexpr int Q4 = PJ_QP / 4;
    double qv[Q4];
#pragma unroll
    for (int u = 0; u < Q4; ++u) {
        const int a = g + 4 * u;
        qv[u] = (a < q && live) ? Q1[R0 + i + (int64_t)a * npad] : 0.0;
    }
    const double rsv = (g == 0 && live) ? sum_partials(rspart + R0 + i, nseg, npad) * rsqrtn : 0.0;
    if (q > 1) {
        double lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u, c = e & 63, t = e >> 6;
            lv[u] = (c < d && t <= c) ? linv[c + (int64_t)t * 128] : 0.0;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // segments in batches of eight: 64 loads in flight, fixed summation order (small n has up to 16 segments -- one per
            // column tile --; adding those beyond the eighth one dependent load at a time cost 25 of this kernel's 52 us at n = 2048)
            double sv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int s0 = 0; s0 < nseg; s0 += 8) {
                double wv[8][8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int sgm = 0; sgm < 8; ++sgm) {
                        const int t = g * 16 + half * 8 + u;
                        wv[u][sgm] = s0 + sgm < nseg ? Wpart[((int64_t)(s0 + sgm) * 64 + t) * npad + R0 + i] : 0.0;
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    sv[u] += ((wv[u][0] + wv[u][1]) + (wv[u][2] + wv[u][3])) + ((wv[u][4] + wv[u][5]) + (wv[u][6] + wv[u][7]));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) Ws[i * PJ_LS + (g * 16 + half * 8 + u)] = live ? sv[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            Ls[(e & 63) * PJ_LS + (e >> 6)] = lv[u];
        }
    }
#pragma unroll
    for (int u = 0; u < Q4; ++u) {
        const int a = g + 4 * u;
        Qs[a * PJ_LS + i] = qv[u];       // rows q .. 79 are zero
        if (a > d || q == 1) W1s[a * PJ_LS + i] = 0.0;  // rows beyond the last column of W1 (written below: 0 and 1 .. d)
    }
    if (g == 0) {
        W1s[i] = rsv;
        W1[R0 + i] = rsv;
    }
    __syncthreads();
    if (q > 1) {
        // W1x' (c x i) = inv(Lx) (c x t) W'' (t x i): 4 x 4 tiles, four per wave
#pragma unroll 1
        for (int tile = wave; tile < 16; tile += 4) {
            const int ct = tile >> 2, it = tile & 3;
            const v4d acc = pj_tile(Ls, Ws, ct, it, 64, l15, l4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ct * 16 + l4 + 4 * r, ii = it * 16 + l15;
                if (c < d) {
                    W1s[(1 + c) * PJ_LS + ii] = acc[r];
                    W1[R0 + ii + (int64_t)(1 + c) * npad] = acc[r];
                }
            }
        }
    }
    __syncthreads();
    // Gp (a x b) = Q1' (a x i) W1 (i x b)
    double *out = Gp + (size_t)blockIdx.x * q * q;
    const int qt = (q + 15) >> 4;
#pragma unroll 1
    for (int tile = wave; tile < qt * qt; tile += 4) {
        const int at = tile % qt, bt = tile / qt;
        const v4d acc = pj_tile(Qs, W1s, at, bt, 64, l15, l4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = at * 16 + l4 + 4 * r, bb = bt * 16 + l15;
            if (a < q && bb < q) out[a + bb * q] = acc[r];
            if (at == bt && a == bb && a < q) Ls[a] = acc[r];  // the chunk's diagonal (Ls is free by now)
        }
    }
    __syncthreads();
    if (tid == 0) {  // the chunk's share of trace(G), summed in a fixed order
        double tr = 0.0;
        for (int a = 0; a < q; ++a) tr += Ls[a];
        trp[blockIdx.x] = tr;
    }
}

// G = sum of the chunk partials; trace (sum of the chunks' traces, the same order in every workgroup) -> mu; M = G / 2 + mu / 2 I
__global__ __launch_bounds__(256) void proj_shift_kernel(const double *__restrict__ Gp, const double *__restrict__ trp, int nchunks, int q,
                                                         double nphi0, double nmq, double *__restrict__ G, double *__restrict__ Mh,
                                                         double *__restrict__ scal, int *__restrict__ flags) {
    __shared__ double dg[256];
    __shared__ double smu;
    const int tid = threadIdx.x;
    const size_t qq = (size_t)q * q;
    const int idx = blockIdx.x * 256 + tid;
    const double *p0 = Gp + (idx < (int)qq ? idx : 0);
    double tv = 0.0;
    for (int c = tid; c < nchunks; c += 256) tv += trp[c];
    double s0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int c = 0;
    for (; c + 64 <= nchunks; c += 64) {  // 64 loads in flight
        double v0[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v0[u] = p0[(size_t)(c + u) * qq];
#pragma unroll
        for (int u = 0; u < 64; ++u) s0[u & 7] += v0[u];
    }
    for (; c < nchunks; ++c) s0[c & 7] += p0[(size_t)c * qq];
    const double gv = ((s0[0] + s0[1]) + (s0[2] + s0[3])) + ((s0[4] + s0[5]) + (s0[6] + s0[7]));
    dg[tid] = tv;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) dg[tid] += dg[tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        const double tr = dg[0];
        const double mu = (nphi0 - tr) / nmq;
        const bool ok = mu > 0.0 && mu < 1e300;
        smu = ok ? mu : 1.0;
        if (blockIdx.x == 0) {
            scal[0] = tr;
            scal[1] = smu;
            flags[2] = ok ? 0 : 1;
        }
    }
    __syncthreads();
    if (idx < (int)qq) {
        G[idx] = gv;
        Mh[idx] = 0.5 * gv + ((idx % q) == (idx / q) ? 0.5 * smu : 0.0);
    }
}

// rows [64 b, 64 b + 64): V = W1 - Q1 M; PA = [Q | V | 0], PB = [V | Q | 0] (K2 columns, columns `skip`.. of Q1 / V only),
// v0 = first column of V when the constant column is skipped (its rank-2 term goes through the update's epilogue)
__global__ __launch_bounds__(256) void proj_panels_kernel(const double *__restrict__ Q1, const double *__restrict__ W1,
                                                          const double *__restrict__ Mh, int64_t n, int64_t npad, int q, int K2, int skip,
                                                          double *__restrict__ PA, double *__restrict__ PB, double *__restrict__ v0) {
    __shared__ double Qs[64 * PJ_LS];      // [i][a], a < 68
    __shared__ double Ms[PJ_QP * PJ_LS];   // [b][a]
    const int tid = threadIdx.x, i = tid & 63, g = tid >> 6;
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int64_t R0 = (int64_t)blockIdx.x * 64;
    const bool live = R0 + i < n;
    constexpr int QA = 17;  // a = g + 4 u < 68
    double qv[QA];
#pragma unroll
    for (int u = 0; u < QA; ++u) {
        const int a = g + 4 * u;
        qv[u] = (a < q && live) ? Q1[R0 + i + (int64_t)a * npad] : 0.0;
    }
    constexpr int ME = (PJ_QP * PJ_LS + 255) / 256;
    double mv[ME];
#pragma unroll
    for (int u = 0; u < ME; ++u) {  // Ms[b][a] = M[a + b q], zero padded; all loads before the first LDS store
        const int e = tid + 256 * u, bb = e / PJ_LS, a = e % PJ_LS;
        mv[u] = (a < q && bb < q) ? Mh[a + bb * q] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < ME; ++u) {
        const int e = tid + 256 * u;
        if (e < PJ_QP * PJ_LS) Ms[e] = mv[u];
    }
#pragma unroll
    for (int u = 0; u < QA; ++u) Qs[i * PJ_LS + g + 4 * u] = qv[u];
    __syncthreads();
    const int qq = q - skip, qt = (q + 15) >> 4, K = (q + 3) & ~3;
    // (Q1 M)' (b x i) = M' (b x a) Q1' (a x i)
#pragma unroll 1
    for (int tile = wave; tile < qt * 4; tile += 4) {
        const int bt = tile >> 2, it = tile & 3;
        const v4d acc = pj_tile(Ms, Qs, bt, it, K, l15, l4);
        const int ii = it * 16 + l15;
        const bool lv = R0 + ii < n;
        double w1[4], qb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = bt * 16 + l4 + 4 * r;
            w1[r] = (b < q && lv) ? W1[R0 + ii + (int64_t)b * npad] : 0.0;
            qb[r] = b < q ? Qs[ii * PJ_LS + b] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = bt * 16 + l4 + 4 * r;
            if (b >= q) continue;
            const double v = lv ? w1[r] - acc[r] : 0.0;
            if (b >= skip) {
                const int c = b - skip;
                PA[R0 + ii + (int64_t)c * npad] = qb[r];
                PB[R0 + ii + (int64_t)c * npad] = v;
                PA[R0 + ii + (int64_t)(qq + c) * npad] = v;
                PB[R0 + ii + (int64_t)(qq + c) * npad] = qb[r];
            } else {
                v0[R0 + ii] = v;
            }
        }
    }
    for (int c = 2 * qq + g; c < K2; c += 4) {
        PA[R0 + i + (int64_t)c * npad] = 0.0;
        PB[R0 + i + (int64_t)c * npad] = 0.0;
    }
}

bool gram_w_applies(const mrbf_ctx *ctx, const mrbf_model *M) {
    static const int on = mrbf_env("MRBF_GRAM_FUSED") ? atoi(mrbf_env("MRBF_GRAM_FUSED")) : 1;
    if (!on || ctx->gram_mode != 0) return false;
    if (M->dpad != 64 || M->q < 1 || M->q > 65 || M->n < 1024) return false;
    return M->kp.fast || M->kp.kid == MRBF_GAUSSIAN || M->kp.kid == MRBF_THIN_PLATE_SPLINE;
}

// Phi (lower triangle, column-major, leading dimension ld) + the partial panels of W' = Phi Xc and of the row sums
int launch_gram_w(mrbf_ctx *ctx, const mrbf_model *M, double *Phi, int64_t ld, double **Wpart_out, double **rspart_out, int *nseg_out) {
    const int nt = (int)(M->npad / 128);
    static const int dbg = mrbf_env("MRBF_GW_DBG") ? atoi(mrbf_env("MRBF_GW_DBG")) : 0;
    static const int wgs = mrbf_env("MRBF_GW_WGS") ? atoi(mrbf_env("MRBF_GW_WGS")) : 512;
    int nseg = wgs / nt;  // two workgroups per compute unit
    if (nseg < 1) nseg = 1;
    if (nseg > nt) nseg = nt;
    const int seglen = (nt + nseg - 1) / nseg;
    nseg = (nt + seglen - 1) / seglen;  // no empty segment
    double *Wpart, *rspart;
    MRBF_TRY(get_buf(ctx, S_GW_PART, (size_t)nseg * 64 * M->npad, &Wpart));
    MRBF_TRY(get_buf(ctx, S_GW_RS, (size_t)nseg * M->npad, &rspart));
    const dim3 grid((unsigned)(nt * nseg));
    if (M->kp.fast) {
        MRBF_DISPATCH_KID(M->kp.kid, hipLaunchKernelGGL((gram_w_kernel<KID, true>), grid, dim3(256), 0, ctx->stream, M->Xc, M->sq, M->n,
                                                        M->npad, Phi, ld, M->kp, nt, nseg, seglen, Wpart, rspart, dbg));
    } else if (M->kp.kid == MRBF_GAUSSIAN) {
        hipLaunchKernelGGL((gram_w_kernel<MRBF_GAUSSIAN, false>), grid, dim3(256), 0, ctx->stream, M->Xc, M->sq, M->n, M->npad, Phi, ld,
                           M->kp, nt, nseg, seglen, Wpart, rspart, dbg);
    } else {
        hipLaunchKernelGGL((gram_w_kernel<MRBF_THIN_PLATE_SPLINE, false>), grid, dim3(256), 0, ctx->stream, M->Xc, M->sq, M->n, M->npad,
                           Phi, ld, M->kp, nt, nseg, seglen, Wpart, rspart, dbg);
    }
    MRBF_HIP(ctx, hipGetLastError());
    *Wpart_out = Wpart;
    *rspart_out = rspart;
    *nseg_out = nseg;
    return 0;
}

// W1 = Phi Q1 from the partial panels, G = Q1' W1, mu, M = G / 2 + mu / 2 I, the operand panels of the rank-2q update
int launch_projection_small(mrbf_ctx *ctx, const mrbf_model *M, const double *Wpart, const double *rspart, int nseg, const double *linv,
                            const double *Q1, double *W1, double *G, double *scal, int *flags, int K2, int skip, double *PA, double *PB,
                            double *v0) {
    const int64_t n = M->n, npad = M->npad;
    const int q = M->q, nch = (int)(npad / 64);
    double *Gp, *Mh;
    MRBF_TRY(get_buf(ctx, S_GW_GP, (size_t)nch * q * q + nch, &Gp));
    MRBF_TRY(get_buf(ctx, S_GW_M, (size_t)q * q, &Mh));
    double *trp = Gp + (size_t)nch * q * q;
    hipLaunchKernelGGL(proj_reduce_kernel, dim3((unsigned)nch), dim3(256), 0, ctx->stream, Wpart, rspart, nseg, npad, n, M->d, q, linv, Q1,
                       1.0 / std::sqrt((double)n), W1, Gp, trp);
    hipLaunchKernelGGL(proj_shift_kernel, dim3((unsigned)((q * q + 255) / 256)), dim3(256), 0, ctx->stream, Gp, trp, nch, q,
                       (double)n * M->kp.phi0, (double)std::max<int64_t>(n - q, 1), G, Mh, scal, flags);
    hipLaunchKernelGGL(proj_panels_kernel, dim3((unsigned)nch), dim3(256), 0, ctx->stream, Q1, W1, Mh, n, npad, q, K2, skip, PA, PB, v0);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf
