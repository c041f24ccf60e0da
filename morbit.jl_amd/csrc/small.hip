// Small problems (n <= 512 sites, d <= 128): the whole fit -- centring, Gram matrix, polynomial-tail basis, projection, Cholesky
// factorisation, the two triangular solves, tail coefficients -- in ONE launch, one workgroup per problem.
//
// This is the regime Morbit actually runs in (n <= (d+1)(d+2)/2 sites per model, /root/reference/src/models/RbfModel.jl:356) and the
// one its benchmark driver parallelises over (Threads.@threads over independent problems,
// /root/reference/examples/large_scale_benchmarks.jl:253; n = 2d + 1 sites per start, :157).  The per-problem launch chain of
// solve.hip (~60 launches, memsets and copies) is pure launch latency at these sizes: 0.87 ms for 5.7 MFLOP of factorisation at
// n = 257.  Here a batch of problems is one grid; a single mrbf_fit of such a size is a grid of one workgroup, so a batch and single
// calls run the same code on the same data and agree bit for bit.
//
// The arithmetic is that of solve.hip's Cholesky paths (mrbf_fit_info.path 1 and 2): K = P Phi P + mu Q1 Q1' on the lower triangle of
// Phi, Q1 from the Cholesky-QR of the centred coordinates, 128 x 128 diagonal blocks factored by the register-resident MFMA core of
// chol_diag_core.hpp, everything else by one generic workgroup GEMM on the fp64 matrix cores with its operands read from global
// memory (the working set of a problem, ~4 MB, lives in L2 / Infinity Cache).  A problem whose factorisation meets a non-positive
// pivot, a rank-deficient tail basis or a non-positive shift sets a flag and the caller takes the ordinary path (LU) for it.
#include "chol_diag_core.hpp"
#include "radial.hpp"
#include "small.hpp"

#include <utility>

namespace mrbf {
namespace smallfit {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double phi_rt(double s, const KP &p) {
    switch (p.kid) {
        case MRBF_CUBIC: return p.fast ? rbf_phi_t<MRBF_CUBIC, true>(s, p) : rbf_phi_t<MRBF_CUBIC, false>(s, p);
        case MRBF_INV_MULTIQUADRIC: return p.fast ? rbf_phi_t<MRBF_INV_MULTIQUADRIC, true>(s, p) : rbf_phi_t<MRBF_INV_MULTIQUADRIC, false>(s, p);
        case MRBF_MULTIQUADRIC: return p.fast ? rbf_phi_t<MRBF_MULTIQUADRIC, true>(s, p) : rbf_phi_t<MRBF_MULTIQUADRIC, false>(s, p);
        case MRBF_THIN_PLATE_SPLINE: return rbf_phi_t<MRBF_THIN_PLATE_SPLINE, false>(s, p);
        default: return rbf_phi_t<MRBF_GAUSSIAN, false>(s, p);
    }
}

// epilogue accesses through explicit global-address-space pointers: the workspace pointers come out of a struct, so the compiler
// would emit flat_ instructions (both counters, the LDS aperture check) for every element
typedef __attribute__((address_space(1))) double gdbl;
__device__ __forceinline__ void gst(double *p, double v) { *(gdbl *)p = v; }
__device__ __forceinline__ double gld(const double *p) { return *(const gdbl *)p; }

// elementwise pass over [0, count): f(e, loaded...) with the loads of eight elements per thread issued before the first store
template <class Load, class Store>
__device__ __forceinline__ void batched_pass(int count, Load load, Store store, int member = 0, int nc = 1) {
    for (int e0 = threadIdx.x + 8 * 256 * member; e0 < count; e0 += 8 * 256 * nc) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = load(e0 + 256 * u < count ? e0 + 256 * u : e0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + 256 * u < count) store(e0 + 256 * u, t[u]);
    }
}

// C(i, j) (op)= sum_k A'(i, k) B'(k, j) for one workgroup of 4 waves; A' = TA ? A^T : A, B' = TB ? B^T : B (A, B column-major with
// leading dimensions lda, ldb); M, N multiples of 16, K a multiple of 4.  Every wave takes 32 x 32 macro tiles (2 x 2 MFMA tiles)
// round-robin; the epilogue is pre(i, j) -> what it reads, then post(i, j, value, what pre returned).  LOWER: only 16 x 16 tiles on or below the diagonal.
// f64 MFMA lane maps (tests/test_gpu_parity.py::test_f64_mfma_lane_maps): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
// D[row = (l >> 4) + 4 r][col = l & 15].
template <bool TA, bool TB, bool LOWER, class Pre, class Post>
__device__ __forceinline__ void wg_gemm(int M, int N, int K, const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb, Pre pre,
                                        Post post, int member = 0, int nc = 1) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int mt = M >> 4, nt = N >> 4, MT = (mt + 1) >> 1, NT2 = (nt + 1) >> 1;
    for (int idx = wave + 4 * member; idx < MT * NT2; idx += 4 * nc) {  // macro tiles dealt over the waves of the cluster's workgroups
        const int I = idx % MT, J = idx / MT;
        if (LOWER && 2 * I + 1 < 2 * J) continue;
        const int i0 = 32 * I, j0 = 32 * J;
        const bool va1 = 2 * I + 1 < mt, vb1 = 2 * J + 1 < nt;
        v4d acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        // (explicit global address space: global_load, not flat_load; no guards on the loads -- K is a multiple of 16 at every call
        //  site and a missing second tile re-reads the first one: as `cond ? load : 0` every operand load sat in its own exec-mask
        //  branch)
        typedef const __attribute__((address_space(1))) double gcd;
        // Within a group of 16 columns of K, MFMA step s takes k = 4 l4 + s from lane group l4 (any assignment works as long as both
        // operands use the same one): an operand that is contiguous in k then gives every lane 32 contiguous bytes per burst and the
        // four lane groups one whole 128-byte line per row.  (With k = 4 s + l4 every 8-byte load took its own 32 bytes out of a line
        // that three later loads fetched again from the L2: 128 KB of line traffic per burst and CU, ~1.1 us, the pace of every
        // product here.)
        typedef const __attribute__((address_space(1))) v4d gcv4 __attribute__((aligned(8)));
        gcd *pa = (gcd *)(TA ? A + 4 * l4 + (int64_t)(i0 + l15) * lda : A + (i0 + l15) + (int64_t)(4 * l4) * lda);
        gcd *pb = (gcd *)(TB ? B + (j0 + l15) + (int64_t)(4 * l4) * ldb : B + 4 * l4 + (int64_t)(j0 + l15) * ldb);
        const int64_t oa = va1 ? (TA ? 16 * (int64_t)lda : 16) : 0, ob = vb1 ? (TB ? 16 : 16 * (int64_t)ldb) : 0;
        // Bursts of UK = 4 k-steps (one group of 16 columns) kept in a ring of NR: a memory round trip costs ~1.5 us here (a problem's
        // working set does not stay in the L2 once a batch shares it), the MFMAs of a burst 0.1 us.
        constexpr int UK = 4, NR = 4;
        double fa0[NR][UK], fa1[NR][UK], fb0[NR][UK], fb1[NR][UK];
        auto burst = [&](int buf) {
            if constexpr (TA) {
                const v4d x0 = *(gcv4 *)pa, x1 = *(gcv4 *)(pa + oa);
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    fa0[buf][u] = x0[u];
                    fa1[buf][u] = x1[u];
                }
                pa += 16;
            } else {
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    fa0[buf][u] = pa[(int64_t)u * lda];
                    fa1[buf][u] = pa[(int64_t)u * lda + oa];
                }
                pa += 16 * (int64_t)lda;
            }
            if constexpr (!TB) {
                const v4d y0 = *(gcv4 *)pb, y1 = *(gcv4 *)(pb + ob);
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    fb0[buf][u] = y0[u];
                    fb1[buf][u] = y1[u];
                }
                pb += 16;
            } else {
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    fb0[buf][u] = pb[(int64_t)u * ldb];
                    fb1[buf][u] = pb[(int64_t)u * ldb + ob];
                }
                pb += 16 * (int64_t)ldb;
            }
        };
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (4 * UK * r < K) burst(r);
        for (int k0 = 0; k0 < K; k0 += 4 * UK * NR) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (k0 + 4 * UK * r < K) {
#pragma unroll
                    for (int u = 0; u < UK; ++u) {
                        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb0[r][u], fa0[r][u], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb1[r][u], fa0[r][u], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb0[r][u], fa1[r][u], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb1[r][u], fa1[r][u], acc[1][1], 0, 0, 0);
                    }
                    if (k0 + 4 * UK * (r + NR) < K) burst(r);
                }
            }
        }
        // Epilogue in two passes: everything the epilogue READS for this macro tile first (pre: one burst of loads), then the stores
        // (post).  With a per-element read-modify-write the compiler must keep every load behind the previous element's store
        // (they may alias), and a 32 x 32 tile paid sixteen dependent memory round trips per lane: the Gram matrix of a 257-site
        // problem took 260 us whatever the dimension.  The products are formed transposed (B fragment as the MFMA's A operand), so
        // that the lane index runs along i, the contiguous index of every column-major output.
        double old[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                // (no branch around the reads: a missing second tile reads the first one's operands again, a tile above the diagonal
                //  is read like any other and not written -- inside a branch every read waited for memory on its own)
                const int ia = i0 + ((a == 1 && va1) ? 16 : 0), jb = j0 + ((b == 1 && vb1) ? 16 : 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) old[a][b][r] = pre(ia + l15, jb + l4 + 4 * r);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if ((a == 1 && !va1) || (b == 1 && !vb1)) continue;
                if (LOWER && 2 * I + a < 2 * J + b) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) post(i0 + 16 * a + l15, j0 + 16 * b + l4 + 4 * r, acc[a][b][r], old[a][b][r]);
            }
    }
}

struct NoPre {
    __device__ __forceinline__ double operator()(int, int) const { return 0.0; }
};

__device__ __forceinline__ double wg_sum(double v, double *red /* 4 doubles of LDS */) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// the 128 x 128 diagonal-block core as ONE out-of-line function: inlined at both call sites it pushed the whole kernel into scratch
__device__ __noinline__ int diag_block(double *__restrict__ A, int ld, double *__restrict__ Linv, diagcore::DiagV4Shared &sh) {
    diagcore::v4d acc[diagcore::NSLOT];
    return diagcore::diag_v4_core<false, false, false>(A, (int64_t)ld, Linv, sh, acc, nullptr, nullptr, 0);
}

// The same block by the persistent factorisation's core (chol_diag_core.hpp, diag_v6_core: one barrier per 16-column panel, 26 us against
// the 40 of the v4 core above).  It streams its leaf inverses through `itg` (8 x 256 doubles of global scratch) and counts panels in `prog`.
union DiagShared {
    diagcore::DiagV4Shared v4;
    diagcore::DiagV6Shared v6;
};
__device__ __noinline__ int diag_block6(double *__restrict__ A, int ld, double *__restrict__ Linv, diagcore::DiagV6Shared &sh, double *itg,
                                        unsigned *prog) {
    diagcore::v4d acc[diagcore::NSLOT6];
    diagcore::diag_v6_load(A, (int64_t)ld, acc);
    return diagcore::diag_v6_core(A, (int64_t)ld, Linv, sh, acc, itg, prog);
}

// A 128 x 128 diagonal block of which only the leading `real` x `real` part is not the identity (the last block of a padded matrix:
// n = 2d + 1 = 257 leaves 16 real rows in its third block; the tail basis' Gram matrix of a problem with d <= 32): factor and invert
// that part in LDS with one wave -- a few microseconds instead of the 40 of the register-resident 128 x 128 core.  Same contract as
// diag_block (L in the lower triangle of A, inv(L) as a full 128 x 128 block, the 1-based index of a bad pivot or 0).
__device__ __noinline__ int small_block(double *__restrict__ A, int ld, int real, double *__restrict__ Linv, double *ws /* >= 2 x 32 x 33 doubles of LDS */) {
    constexpr int LS = 33;
    double *L = ws, *X = ws + 32 * LS;
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e & 31, j = e >> 5;
        L[i * LS + j] = (i < real && j <= i) ? gld(&A[i + (int64_t)j * ld]) : (i == j ? 1.0 : 0.0);
        X[i * LS + j] = 0.0;
    }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (tid < 64) {
        const int i = tid;  // lane i owns row i
        for (int k = 0; k < real; ++k) {
            const double piv = L[k * LS + k];
            if (!(piv > 0.0)) {
                if (i == 0) s_bad = k + 1;
                break;  // uniform: every lane reads the same pivot
            }
            const double lkk = sqrt(piv);
            double lik = 0.0;
            if (i > k && i < real) {
                lik = L[i * LS + k] / lkk;
                L[i * LS + k] = lik;
            }
            if (i == k) L[k * LS + k] = lkk;
            __builtin_amdgcn_s_waitcnt(0xc07f);  // LDS writes of this wave are visible to its later reads in program order
            __builtin_amdgcn_wave_barrier();
            if (i > k && i < real)
                for (int j = k + 1; j <= i; ++j) L[i * LS + j] = fma(-lik, L[j * LS + k], L[i * LS + j]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
        if (s_bad == 0 && i < real) {
            // column i of inv(L): forward substitution  x_r = (delta_ri - sum_{i <= p < r} L_rp x_p) / L_rr
            for (int r = i; r < real; ++r) {
                double acc = r == i ? 1.0 : 0.0;
                for (int pp = i; pp < r; ++pp) acc = fma(-L[r * LS + pp], X[pp * LS + i], acc);
                X[r * LS + i] = acc / L[r * LS + r];
            }
        }
    }
    __syncthreads();
    const int bad = s_bad;
    if (bad) return bad;
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e & 127, j = e >> 7;
        double v = i == j ? 1.0 : 0.0;
        if (i < real && j < real) v = j <= i ? X[i * LS + j] : 0.0;
        gst(&Linv[e], v);
        if (i < real && j <= i && j < real) gst(&A[i + (int64_t)j * ld], L[i * LS + j]);
    }
    __syncthreads();
    return 0;
}
// Cholesky factor and its inverse of an m x m s.p.d. matrix, m <= 64, by one workgroup in registers: thread (i, w) = (tid & 63,
// tid >> 6) holds the entries (i, 4 u + w), u = 0 .. 15, of G and of E (starts as the identity).  Column k of [G ; E] is published
// unscaled through LDS (double buffered: ONE barrier per column), every thread scales what it reads with the pivot's reciprocal square
// root itself and applies the column operation col_j -= col_k L(j, k) to its 16 + 16 entries: G becomes L, E becomes L^-T (the leaf's
// trick of chol_diag_core.hpp at workgroup size).  ~0.25 us per column instead of the 1.5-3 us of small_block's one-wave loops
// over LDS (fine for the 1 .. 16 real rows it was written for, 105 us at m = 32).
// Returns the 1-based index of the first non-positive pivot or 0; on return g[u] = L(i, 4u + w) for 4u + w <= i, e[u] = inv(L)(4u + w, i).
template <int K>
__device__ __forceinline__ void chol64_step(double (&g)[16], double (&e)[16], int m, double *buf, int i, int w, int &bad) {
    if (K >= m) return;  // uniform
    constexpr int uk = K >> 2, wk = K & 3;
    double *cb = buf + (K & 1) * 128;
    if (w == wk) {
        cb[i] = g[uk];
        cb[64 + i] = e[uk];
    }
    __syncthreads();
    const double piv = cb[K];
    if (!(piv > 0.0) && bad == 0) bad = K + 1;
    const double rinv = diagcore::fast_rsqrt_v4(piv);
    const double ci = cb[i] * rinv, ei = cb[64 + i] * rinv;
    constexpr int u0 = (K + 1) >> 2;  // the first group of four columns that reaches beyond K
    double cj[16];
#pragma unroll
    for (int u = u0; u < 16; ++u) cj[u] = cb[4 * u + w];
#pragma unroll
    for (int u = u0; u < 16; ++u) {
        const int j = 4 * u + w;
        const double lj = cj[u] * rinv;
        const bool beyond = j > K;
        g[u] = (beyond && j <= i) ? fma(-ci, lj, g[u]) : g[u];
        e[u] = beyond ? fma(-ei, lj, e[u]) : e[u];
    }
    if (w == wk) {
        g[uk] = ci;
        e[uk] = ei;
    }
}
template <int... Ks>
__device__ __forceinline__ void chol64_steps(double (&g)[16], double (&e)[16], int m, double *buf, int i, int w, int &bad, std::integer_sequence<int, Ks...>) {
    (chol64_step<Ks>(g, e, m, buf, i, w, bad), ...);  // (a fold, not a loop: every register index is a constant whatever the unroller decides)
}
__device__ __forceinline__ int chol64_regs(double (&g)[16], double (&e)[16], int m, double *buf /* 2 x 2 x 64 doubles of LDS */) {
    const int i = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bad = 0;
    chol64_steps(g, e, m, buf, i, w, bad, std::make_integer_sequence<int, 64>());
    return bad;
}

// The same contract as small_block / diag_block for a 128 x 128 block whose leading `real` x `real` part (real <= 64) is not the identity:
// L into the lower triangle of A, inv(L) as a full 128 x 128 block, the 1-based index of a bad pivot or 0.  ws: 64 * 65 + 256 doubles of LDS.
__device__ __noinline__ int block64(double *__restrict__ A, int ld, int real, double *__restrict__ Linv, double *ws) {
    double *Li = ws, *colbuf = ws + 64 * 65;
    const int tid = threadIdx.x, i = tid & 63, w = tid >> 6;
    double g[16], e[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = 4 * u + w;
        g[u] = (i < real && j <= i) ? gld(&A[i + (int64_t)j * ld]) : (i == j ? 1.0 : 0.0);
        e[u] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    const int bad = chol64_regs(g, e, real, colbuf);
    if (bad) return bad;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = 4 * u + w;  // this thread holds L(i, j) and E(i, j) = inv(L)(j, i)
        Li[j * 65 + i] = (j < real && i <= j) ? e[u] : 0.0;
        if (i < real && j <= i) gst(&A[i + (int64_t)j * ld], g[u]);
    }
    __syncthreads();
    for (int idx = tid; idx < 128 * 128; idx += 256) {
        const int a = idx & 127, b = idx >> 7;
        gst(&Linv[idx], (a < real && b < real) ? Li[a * 65 + b] : (a == b ? 1.0 : 0.0));
    }
    __syncthreads();
    return 0;
}
__device__ __forceinline__ int diag_block_auto(double *__restrict__ A, int ld, int real, double *__restrict__ Linv, diagcore::DiagV4Shared &sh,
                                               double *itg6 = nullptr) {
    // (small_block's one-wave loops over LDS: 2 us at 8 real rows, 13 at 16, 105 at 32; block64: ~0.4 us per column + 10)
    if (real > 0 && real <= 12) return small_block(A, ld, real, Linv, sh.LT);
    if (real > 0 && real <= 64) return block64(A, ld, real, Linv, sh.LT);
    if (itg6) {  // (the scratch is only handed in when the problem asks for the v6 core)
        const int bad = diag_block6(A, ld, Linv, reinterpret_cast<DiagShared *>(&sh)->v6, itg6, reinterpret_cast<unsigned *>(itg6 + 8 * 256));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (its write-through stores of L and inv(L) have left the CU before the barrier that follows)
        return bad;
    }
    return diag_block(A, ld, Linv, sh);
}

// A problem is worked on by nc (1, 2, 4, 8 or 16) workgroups: every product and every elementwise pass deals its tiles / elements over
// them, the sequential pieces (the 128 x 128 diagonal blocks, the triangular solves, the tail) are member 0's, and the members meet at
// a barrier after every phase.  With one workgroup per problem a batch of 64 kept 64 of the 256 CUs busy for the whole fit.
// The members of a cluster are the blocks x + 8 (nc g + m), m = 0..nc-1: blocks are dealt round-robin over the 8 XCDs (observed, not
// promised), so a cluster shares ONE L2.
//
// Memory model of the barrier -- on purpose NOT an agent-scope release / acquire pair.  An agent-scope release on gfx950 is
// `buffer_wbl2 sc1` (write back the XCD's whole L2: the problem's ~4 MB working set is dirty there) and an agent-scope acquire drops
// the L2 as well; 25 barriers per fit would each pay that.  What the code relies on instead, and checks:
//   * the vector L1 of a CU is write-through: a store has reached the L2 when it is counted out of vmcnt, so `s_waitcnt vmcnt(0)`
//     + the workgroup barrier before the arrival means every member's stores ARE in the L2 when the arrival is visible;
//   * all members sit behind the SAME L2 (HW_REG_XCC_ID of every member is published and compared before anything is relied on;
//     a mismatch ends the launch with flag 2 and the host repeats it with one workgroup per problem, for good);
//   * the reader only has to drop its own L1: the acquire fence after the poll is `buffer_inv sc1`.
// The host gates clusters on the device being gfx950 with 256 CUs (context.hip), repeats a launch whose barrier timed out with one
// workgroup per problem (and returns to clusters afterwards: a time-out says nothing about visibility), and uses the interpolation
// residual every fit computes anyway as a tripwire: a clustered fit whose residual is not small is repeated with one workgroup,
// and if that changes the result the context stops using clusters (solve.hip: fit_model, batch.hip).
// Progress at any residency: blocks are dispatched in index order, so the lowest-numbered unfinished group of 8 nc blocks is always
// resident as a whole.  The spin is bounded by wall-clock time all the same.
struct Cluster {
    int member, nc, phase;
    int *words;  // [0] arrivals, [1] failure / bad-pivot word, [2 .. 2 + nc) XCD of the members (device memory, zero at launch)
    unsigned long long ticks;
    int *s_ok;   // LDS word
};
__device__ __forceinline__ int cl_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cl_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// false: the cluster has failed (a sibling did not arrive in time, or reported a failure)
__device__ __forceinline__ bool cl_barrier(Cluster &cl) {
    if (cl.nc == 1) {
        __syncthreads();
        return true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left the CU
    __syncthreads();
    ++cl.phase;
    if (threadIdx.x == 0) {
        const int target = cl.nc * cl.phase;
        __hip_atomic_fetch_add(cl.words, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (cl_load(cl.words) < target) {
            if ((++spins & 15u) == 0u) {
                if (cl_load(cl.words + 1) < 0) {
                    ok = false;
                    break;
                }
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > cl.ticks) {
                    cl_store(cl.words + 1, -9);
                    ok = false;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (ok && cl_load(cl.words + 1) < 0) ok = false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: what the siblings wrote is read from the L2, not from this CU's L1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *cl.s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return *cl.s_ok != 0;
}

// blocked Cholesky of the np x np matrix at A (lower, column-major, ld; np a multiple of 128) with the 128 x 128 MFMA core for the
// diagonal blocks (member 0); rows beyond `rows16` are known to be zero below the diagonal blocks (identity padding) and skipped in
// the panel and trailing products.  Linv: np x 128, block c at Linv + c * 128 * 128 holds inv(L_cc).  Returns 0, the 1-based index
// of a bad pivot, or -1 when the cluster failed.
__device__ int wg_potrf(double *__restrict__ A, int ld, int np, int rows16, double *__restrict__ Linv, double *__restrict__ Pt,
                        diagcore::DiagV4Shared &sh, Cluster &cl, int diag6 = 0) {
    const int nb = np / 128;
    for (int c = 0; c < nb; ++c) {
        double *Acc = A + (int64_t)c * 128 * (ld + 1);
        double *Lc = Linv + (int64_t)c * 128 * 128;
        if (cl.member == 0) {
            const int bad = diag_block_auto(Acc, ld, rows16 - 128 * c, Lc, sh, diag6 ? Pt : nullptr);  // (Pt is free until the panel product)
            __syncthreads();
            if (threadIdx.x == 0 && bad) cl_store(cl.words + 1, 128 * c + bad);  // positive: a bad pivot, every member leaves
        }
        if (!cl_barrier(cl)) return -1;
        {
            const int bad = cl_load(cl.words + 1);
            if (bad > 0) return bad;
        }
        const int r0 = 128 * (c + 1);
        const int mrows = rows16 - r0;  // real rows below this block
        if (mrows <= 0) continue;
        // panel: L(r, c) = A(r, c) inv(L_cc)'  -> Pt (mrows x 128), then back into A
        const double *Arc = A + r0 + (int64_t)c * 128 * ld;
        wg_gemm<false, true, false>(mrows, 128, 128, Arc, ld, Lc, 128, NoPre(), [&](int i, int j, double v, double) { gst(&Pt[i + (int64_t)j * np], v); },
                                    cl.member, cl.nc);
        if (!cl_barrier(cl)) return -1;
        batched_pass(mrows * 128, [&](int e) { return gld(&Pt[e % mrows + (int64_t)(e / mrows) * np]); },
                     [&](int e, double v) { gst(&A[(r0 + e % mrows) + (int64_t)(c * 128 + e / mrows) * ld], v); }, cl.member, cl.nc);
        // trailing update: A(r, s) -= L(r, c) L(s, c)'  for r >= s > c (lower tiles)
        double *Att = A + (int64_t)r0 * (ld + 1);
        wg_gemm<false, true, true>(mrows, mrows, 128, Pt, np, Pt, np, [&](int i, int j) { return gld(&Att[i + (int64_t)j * ld]); },
                                   [&](int i, int j, double v, double o) { gst(&Att[i + (int64_t)j * ld], o - v); }, cl.member, cl.nc);
        if (!cl_barrier(cl)) return -1;
    }
    return 0;
}

// column sums of the n x d sites -> s_mean[0 .. 127] (zero beyond d): wave w takes the rows w, w + 4, ..., eight rows (independent
// loads) per step, lanes over the columns; the four partial sums meet in LDS in a fixed order.  (One thread per column walking all n
// rows paid a memory round trip per row: 80 us of a 257-site fit.)  ONE function for the fit kernel and for small_mean_kernel: a
// batch (centroid computed up front) and a single fit agree bit for bit.
__device__ __forceinline__ void column_means(const double *C, int n, int d, double *part /* 4 x 128 doubles of LDS */, double *s_mean) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gdbl *Cg = (const gdbl *)C;
    double s0 = 0.0, s1 = 0.0;
    const int t0 = lane, t1 = lane + 64;
    for (int i0 = wave; i0 < n; i0 += 32) {
        double a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 4 * u;
            a0[u] = (i < n && t0 < d) ? Cg[(int64_t)i * d + t0] : 0.0;
            a1[u] = (i < n && t1 < d) ? Cg[(int64_t)i * d + t1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s0 += a0[u];
            s1 += a1[u];
        }
    }
    part[wave * 128 + t0] = s0;
    part[wave * 128 + t1] = s1;
    __syncthreads();
    if (tid < 128) s_mean[tid] = tid < d ? ((part[tid] + part[128 + tid]) + (part[256 + tid] + part[384 + tid])) / (double)n : 0.0;
}

__global__ __launch_bounds__(256) void small_mean_kernel(const Prob *__restrict__ many, int count) {
    __shared__ double part[512];
    __shared__ double s_mean[128];
    if ((int)blockIdx.x >= count) return;
    const Prob &P = many[blockIdx.x];
    column_means(P.C, P.n, P.d, part, s_mean);
    __syncthreads();
    for (int t = threadIdx.x; t < P.dpad; t += 256) P.mean[t] = t < 128 ? s_mean[t] : 0.0;
}

__global__ __launch_bounds__(256, 1) void small_fit_kernel(Prob one, const Prob *__restrict__ many, int count, int nc) {
    __shared__ __attribute__((aligned(16))) DiagShared shu;
    diagcore::DiagV4Shared &sh = shu.v4;
    __shared__ double red[4];
    __shared__ double s_mean[128];
    __shared__ int s_ok;
    // nc > 1: block x + 8 (nc g + m) is member m of the cluster of problem x + 8 g (see Cluster)
    const int bx = (int)blockIdx.x;
    const int prob = nc == 1 ? bx : (bx & 7) + 8 * ((bx >> 3) / nc);
    if (prob >= count) return;
    const Prob P = many ? many[prob] : one;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Cluster cl;
    cl.member = nc == 1 ? 0 : (bx >> 3) % nc;
    cl.nc = nc;
    cl.phase = 0;
    cl.words = P.cl;
    cl.ticks = P.spin_ticks;
    cl.s_ok = &s_ok;
    const int member = cl.member;
    const int n = P.n, d = P.d, k = P.k, q = P.q, n16 = P.n16, np = P.npad, dpad = P.dpad, q16 = P.q16;
    const Carve cv = carve(np, q16);
    double *ws = P.ws;
    double *Phi = ws + cv.Phi, *Q1 = ws + cv.Q1, *Wm = ws + cv.Wm, *V = ws + cv.V, *G = ws + cv.G, *Gx = ws + cv.Gx, *LinvX = ws + cv.LinvX,
           *Linv = ws + cv.Linv, *Pt = ws + cv.Pt, *Yc = ws + cv.Ycol, *Bm = ws + cv.B, *Fy = ws + cv.Fy, *Xs = ws + cv.Xs, *T1 = ws + cv.T1,
           *T2 = ws + cv.T2, *Z = ws + cv.Z;
    const int ldz = cv.ldz;
    if (member == 0 && tid < 4) P.flags[tid] = 0;
    int nstamp = 0;
#define MRBF_STAMP()                                                                           \
    do {                                                                                       \
        if (P.stamps && tid == 0 && member == 0) P.stamps[nstamp] = (long long)wall_clock64(); \
        ++nstamp;                                                                              \
    } while (0)
#define MRBF_CLB()                  \
    do {                            \
        if (!cl_barrier(cl)) {      \
            if (tid == 0) P.flags[3] = 1; /* the cluster failed: the host repeats the launch with one workgroup per problem */ \
            return;                 \
        }                           \
    } while (0)
    MRBF_STAMP();
    if (nc > 1) {
        // the XCD this member runs on; after the first barrier every member compares the four
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (tid == 0) cl_store(cl.words + 2 + member, (int)(xcc & 0xf) + 1);
    }

    // ---- centroid, centred + zero-padded coordinates, squared norms (the model's own arrays: the evaluation uses them later)
    if (P.mean_given) {
        if (tid < 128) s_mean[tid] = tid < d ? gld(&P.mean[tid]) : 0.0;
    } else {
        column_means(P.C, n, d, sh.LT, s_mean);
    }
    __syncthreads();
    if (member == 0 && !P.mean_given)
        for (int t = tid; t < dpad; t += 256) P.mean[t] = t < 128 ? s_mean[t] : 0.0;
    {
        const gdbl *Cg = (const gdbl *)P.C;
        gdbl *Xg = (gdbl *)P.Xc;
        for (int row0 = wave + 16 * member; row0 < np; row0 += 16 * nc) {  // four rows per wave and step
            double sacc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = row0 + 4 * u;
                double sr = 0.0;
                if (row < np) {
                    for (int t = lane; t < dpad; t += 64) {
                        double v = 0.0;
                        if (row < n && t < d) v = Cg[(int64_t)row * d + t] - s_mean[t];
                        Xg[(int64_t)row * dpad + t] = v;
                        sr = fma(v, v, sr);
                    }
                }
                sacc[u] = sr;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) sacc[u] += __shfl_xor(sacc[u], off);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (lane == 0 && row0 + 4 * u < np) P.sq[row0 + 4 * u] = sacc[u];
        }
    }
    // right-hand sides, column-major and zero padded: Yc (np x 16)
    for (int e = tid + 256 * member; e < np * 16; e += 256 * nc) {
        const int i = e % np, l = e / np;
        Yc[e] = (i < n && l < k) ? P.Y[(int64_t)i * k + l] : 0.0;
    }
    MRBF_CLB();
    if (nc > 1) {
        const int x0 = cl_load(cl.words + 2);
        bool same = x0 != 0;
        for (int m2 = 1; m2 < nc; ++m2) same = same && cl_load(cl.words + 2 + m2) == x0;
        if (!same) {  // not one L2: nothing has been relied on yet (every member sees the same four words and leaves)
            if (tid == 0 && member == 0) P.flags[3] = 2;
            return;
        }
    }

    if (P.fault && nc > 1 && member == 1) return;  // test hook: the siblings' next barrier times out, the host repeats with nc = 1

    MRBF_STAMP();  // 1: centred
    // ---- Phi = phi(|x_i - x_j|): GEMM form on the centred coordinates, radial function in the epilogue; rows / columns >= n: identity
    {
        const double *XcT = P.Xc;  // dpad x np column-major
        const KP kp = P.kp;
        const double *sq = P.sq;
        wg_gemm<true, false, false>(
            n16, n16, dpad, XcT, dpad, XcT, dpad, // (no branch around the loads: inside one, the sum waits for its two loads, sixteen dependent round trips per macro tile)
            [&](int i, int j) { return gld(&sq[i < n ? i : n - 1]) + gld(&sq[j < n ? j : n - 1]); },
            [&](int i, int j, double sdot, double sqsum) {
                // squared distances here; the radial function follows as a pass of its own with the kernel family chosen OUTSIDE the
                // loop: inlined into this epilogue the five-way switch stood sixteen times in every macro tile's code (100 KB of
                // instructions per tile, more than the instruction cache holds)
                double v;
                if (i < n && j < n) {
                    v = fma(-2.0, sdot, sqsum);
                    v = v > 0.0 ? v : 0.0;
                    if (i == j) v = 0.0;
                } else {
                    v = (i == j) ? 1.0 : 0.0;
                }
                gst(&Phi[(int64_t)j * np + i], v);
            }, member, nc);
        MRBF_CLB();
        if (P.stamps && tid == 0 && member == 0) P.stamps[11] = (long long)wall_clock64();  // product done
        {
            // eight columns at a time per wave: all loads of a batch before its first store (one element at a time, every iteration
            // paid a memory round trip: 110 us for 257 x 257 entries)
            auto radial = [&](auto f) {
                for (int j0 = 8 * wave + 32 * member; j0 < n; j0 += 32 * nc)
                    for (int i = lane; i < n; i += 64) {
                        double t[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) t[u] = gld(&Phi[(int64_t)(j0 + u < n ? j0 + u : n - 1) * np + i]);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (j0 + u < n) gst(&Phi[(int64_t)(j0 + u) * np + i], f(t[u]));
                    }
            };
            switch (kp.kid) {
                case MRBF_CUBIC:
                    if (kp.fast) radial([&](double t) { return rbf_phi_t<MRBF_CUBIC, true>(t, kp); });
                    else radial([&](double t) { return rbf_phi_t<MRBF_CUBIC, false>(t, kp); });
                    break;
                case MRBF_INV_MULTIQUADRIC:
                    if (kp.fast) radial([&](double t) { return rbf_phi_t<MRBF_INV_MULTIQUADRIC, true>(t, kp); });
                    else radial([&](double t) { return rbf_phi_t<MRBF_INV_MULTIQUADRIC, false>(t, kp); });
                    break;
                case MRBF_MULTIQUADRIC:
                    if (kp.fast) radial([&](double t) { return rbf_phi_t<MRBF_MULTIQUADRIC, true>(t, kp); });
                    else radial([&](double t) { return rbf_phi_t<MRBF_MULTIQUADRIC, false>(t, kp); });
                    break;
                case MRBF_THIN_PLATE_SPLINE: radial([&](double t) { return rbf_phi_t<MRBF_THIN_PLATE_SPLINE, false>(t, kp); }); break;
                default: radial([&](double t) { return rbf_phi_t<MRBF_GAUSSIAN, false>(t, kp); }); break;
            }
        }
        if (P.stamps && tid == 0 && member == 0) P.stamps[12] = (long long)wall_clock64();  // radial function done (this wave)
        // identity in the padding beyond n16 (rows / columns n16 .. np - 1 of the lower triangle are all the factorisation reads there)
        // (rows n16 .. np - 1 of every column, contiguous per column; the block above the diagonal is never read)
        for (int c = wave + 4 * member; c < np; c += 4 * nc)
            for (int r = n16 + lane; r < np; r += 64) gst(&Phi[r + (int64_t)c * np], (r == c) ? 1.0 : 0.0);
    }
    MRBF_CLB();

    MRBF_STAMP();  // 2: Gram
    const double rsn = 1.0 / sqrt((double)n);
    if (q > 0) {
        // ---- Q1 = [1/sqrt(n) | Xc Lx^-T]: orthonormal basis of the polynomial tail from the Cholesky-QR of the centred coordinates
        for (int e = tid + 256 * member; e < np * q16; e += 256 * nc) {
            const int i = e % np, t = e / np;
            Q1[e] = (t == 0 && i < n) ? rsn : 0.0;
        }
        if (q > 1) {
            const int d16 = (d + 15) & ~15;
            // Gx = Xc' Xc (d x d), identity padded to 128 x 128
            for (int e = tid + 256 * member; e < 128 * 128; e += 256 * nc) Gx[e] = (e % 128 == e / 128 && e % 128 >= d) ? 1.0 : 0.0;
            MRBF_CLB();
            wg_gemm<false, true, true>(d16, d16, n16, P.Xc, dpad, P.Xc, dpad, NoPre(), [&](int a, int b, double v, double) {
                if (a < d && b < d) gst(&Gx[a + b * 128], v);
            }, member, nc);
            MRBF_CLB();
            if (member == 0) {
                const int bad = diag_block_auto(Gx, 128, d, LinvX, sh);
                __syncthreads();
                if (tid == 0 && bad) P.flags[1] = bad;
            }
            MRBF_CLB();
            // Qx(i, a) = sum_b Xc(i, b) inv(Lx)(a, b)
            double *Qx = Q1 + np;
            wg_gemm<true, true, false>(n16, d16, d16, P.Xc, dpad, LinvX, 128, NoPre(), [&](int i, int a, double v, double) {
                if (i < n && a < d) gst(&Qx[i + (int64_t)a * np], v);
            }, member, nc);
        }
        MRBF_CLB();
        MRBF_STAMP();  // 3: Q1
        // ---- W1 = Phi Q1;  G = Q1' W1;  W = W1 - 1/2 Q1 G;  mu = (n phi0 - trace G) / (n - q);  V = W - mu/2 Q1
        wg_gemm<false, false, false>(n16, q16, n16, Phi, np, Q1, np, NoPre(), [&](int i, int t, double v, double) { gst(&Wm[i + (int64_t)t * np], v); },
                                     member, nc);
        for (int e = tid + 256 * member; e < (np - n16) * q16; e += 256 * nc) Wm[n16 + e % (np - n16) + (int64_t)(e / (np - n16)) * np] = 0.0;
        MRBF_CLB();
        wg_gemm<true, false, false>(q16, q16, n16, Q1, np, Wm, np, NoPre(), [&](int a, int b, double v, double) { gst(&G[a + b * q16], v); }, member, nc);
        MRBF_CLB();
        double tr = 0.0;
        for (int t = tid; t < q; t += 256) tr += gld(&G[t + t * q16]);  // (every member for itself)
        tr = wg_sum(tr, red);
        const double mu_raw = ((double)n * P.kp.phi0 - tr) / (double)max(n - q, 1);
        const bool mu_ok = mu_raw > 0.0 && mu_raw < 1e300;
        const double mu = mu_ok ? mu_raw : 1.0;
        if (tid == 0 && member == 0) {
            P.scal[0] = tr;
            P.scal[1] = mu;
            if (!mu_ok) P.flags[2] = 1;
        }
        wg_gemm<false, false, false>(n16, q16, q16, Q1, np, G, q16, [&](int i, int t) { return gld(&Wm[i + (int64_t)t * np]); },
                                     [&](int i, int t, double v, double o) { gst(&Wm[i + (int64_t)t * np], o - 0.5 * v); }, member, nc);
        MRBF_CLB();
        {
            const double hmu = -0.5 * mu;
            batched_pass(np * q16, [&](int e) { return fma(hmu, gld(&Q1[e]), gld(&Wm[e])); }, [&](int e, double v) { gst(&V[e], v); }, member, nc);
        }
        MRBF_CLB();
        MRBF_STAMP();  // 4: W, G, mu, V
        // ---- K = Phi - Q1 V' - V Q1' on the lower triangle (column-major: K(i, j), i >= j, at Phi[i + j * np])
        wg_gemm<false, true, true>(n16, n16, q16, Q1, np, V, np, [&](int i, int j) { return gld(&Phi[i + (int64_t)j * np]); },
                                   [&](int i, int j, double v, double o) { gst(&Phi[i + (int64_t)j * np], o - v); }, member, nc);
        __syncthreads();  // (the second product meets every tile in the wave that held it in the first)
        wg_gemm<false, true, true>(n16, n16, q16, V, np, Q1, np, [&](int i, int j) { return gld(&Phi[i + (int64_t)j * np]); },
                                   [&](int i, int j, double v, double o) { gst(&Phi[i + (int64_t)j * np], o - v); }, member, nc);
        MRBF_STAMP();  // 5: K update (issued)
        // ---- B = P Y = Y - Q1 (Q1' Y);  T1 keeps Q1' Y for the tail coefficients
        wg_gemm<true, false, false>(q16, 16, n16, Q1, np, Yc, np, NoPre(), [&](int a, int l, double v, double) { gst(&T1[a + l * ldz], v); }, member, nc);
        batched_pass(np * 16, [&](int e) { return gld(&Yc[e]); }, [&](int e, double v) { gst(&Bm[e], v); }, member, nc);
        MRBF_CLB();
        wg_gemm<false, false, false>(n16, 16, q16, Q1, np, T1, ldz, [&](int i, int l) { return gld(&Bm[i + l * np]); },
                                     [&](int i, int l, double v, double o) { gst(&Bm[i + l * np], o - v); }, member, nc);
    } else {
        batched_pass(np * 16, [&](int e) { return gld(&Yc[e]); }, [&](int e, double v) { gst(&Bm[e], v); }, member, nc);
        if (tid == 0 && member == 0) {
            P.scal[0] = 0.0;
            P.scal[1] = 0.0;
        }
    }
    MRBF_CLB();

    MRBF_STAMP();  // 6: rhs
    // ---- factorisation K = L L'
    {
        const int bad = wg_potrf(Phi, np, np, n16, Linv, Pt, sh, cl, P.diag6);
        if (bad) {
            if (tid == 0 && (member == 0 || bad < 0)) P.flags[bad > 0 ? 0 : 3] = bad > 0 ? bad : 1;
            return;
        }
        if (member != 0) return;  // the triangular solves and the tail are member 0's (one macro tile per wave: nothing to deal out)
    }
    __syncthreads();
    MRBF_STAMP();  // 7: potrf
    const int nb = np / 128;
    // ---- forward substitution L y = B (block rows; inv(L_cc) from the factorisation), then backward L' x = y
    for (int c = 0; c < nb; ++c) {
        if (c > 0) {
            wg_gemm<false, false, false>(128, 16, 128 * c, Phi + 128 * c, np, Fy, np, [&](int i, int l) { return gld(&Bm[128 * c + i + l * np]); },
                                         [&](int i, int l, double v, double o) { gst(&Bm[128 * c + i + l * np], o - v); });
            __syncthreads();
        }
        wg_gemm<false, false, false>(128, 16, 128, Linv + (int64_t)c * 128 * 128, 128, Bm + 128 * c, np, NoPre(),
                                     [&](int i, int l, double v, double) { gst(&Fy[128 * c + i + l * np], v); });
        __syncthreads();
    }
    for (int c = nb - 1; c >= 0; --c) {
        const int rest = np - 128 * (c + 1);
        if (rest > 0) {
            wg_gemm<true, false, false>(128, 16, rest, Phi + 128 * (c + 1) + (int64_t)128 * c * np, np, Xs + 128 * (c + 1), np,
                                        [&](int i, int l) { return gld(&Fy[128 * c + i + l * np]); },
                                        [&](int i, int l, double v, double o) { gst(&Fy[128 * c + i + l * np], o - v); });
            __syncthreads();
        }
        wg_gemm<true, false, false>(128, 16, 128, Linv + (int64_t)c * 128 * 128, 128, Fy + 128 * c, np, NoPre(),
                                    [&](int i, int l, double v, double) { gst(&Xs[128 * c + i + l * np], v); });
        __syncthreads();
    }
    MRBF_STAMP();  // 8: solves
    // ---- tail: re-project w (rounding hygiene), z = Q1' Y - W' w, lam = R^-1 z with R = [[sqrt n, sqrt n mean'], [0, Lx']]
    if (q > 0) {
        wg_gemm<true, false, false>(q16, 16, n16, Q1, np, Xs, np, NoPre(), [&](int a, int l, double v, double) { gst(&T2[a + l * ldz], v); });
        __syncthreads();
        wg_gemm<false, false, false>(n16, 16, q16, Q1, np, T2, ldz, [&](int i, int l) { return gld(&Xs[i + l * np]); },
                                     [&](int i, int l, double v, double o) { gst(&Xs[i + l * np], o - v); });
        __syncthreads();
        wg_gemm<true, false, false>(q16, 16, n16, Wm, np, Xs, np, [&](int a, int l) { return gld(&T1[a + l * ldz]); },
                                    [&](int a, int l, double v, double o) { gst(&Z[a + l * ldz], o - v); });
        for (int e = tid; e < 16 * 16; e += 256) Z[q16 + e % 16 + (e / 16) * ldz] = 0.0;  // the rows the next product may read beyond q16
        __syncthreads();
        if (q > 1) {
            const int d16 = (d + 15) & ~15;
            // lam_tail = Lx^-T z[1:]  =  inv(Lx)' z[1:]
            wg_gemm<true, false, false>(d16, 16, d16, LinvX, 128, Z + 1, ldz, NoPre(), [&](int a, int l, double v, double) { gst(&T2[1 + a + l * ldz], v); });
            __syncthreads();
        }
        // lam[0] = z0 / sqrt(n) - mean . lam[1:]
        for (int l = wave; l < k; l += 4) {
            double acc = 0.0;
            for (int t = 1 + lane; t < q; t += 64) acc = fma(s_mean[t - 1], T2[t + l * ldz], acc);
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0) P.lam[l] = Z[l * ldz] * rsn - acc;
        }
        for (int e = tid; e < (q - 1) * k; e += 256) {
            const int t = 1 + e / k, l = e % k;
            P.lam[(int64_t)t * k + l] = T2[t + l * ldz];
        }
    }
    MRBF_STAMP();  // 9: tail
    // ---- the model's weights: W (n x k row-major), Wc (npad x k column-major, zero padded)
    for (int e = tid; e < np * k; e += 256) {
        const int i = e % np, l = e / np;
        const double v = i < n ? Xs[i + l * np] : 0.0;
        P.Wc[e] = v;
        if (i < n) P.W[(int64_t)i * k + l] = v;
    }
}

// ---- tail basis of the launch chain (n > 512, d <= 64, k <= 16): three launches ------------------------------------------------------
// What solve.hip needs in front of the projection -- Q1 = [1/sqrt n | Xc Lx^-T] (Cholesky-QR of the centred coordinates), inv(Lx),
// T1 = Q1' Y, B = Y - Q1 T1 and the right-hand sides as extra rows of the matrix -- was a chain of twelve launches (memset, copy of Xc
// below its Gram matrix, Xc'Xc in two, identity padding, memset, 128-wide diagonal kernel, panel solve, Q1, Q1'Y, B, extra rows:
// 156 us at n = 2048, longer than the Gram kernel it is meant to hide under; every link a 3-40 us launch).  Here:
//   tailq_gram_kernel    partial sums over row chunks of  Xc' [Xc | Y | 1]          (np workgroups)
//   tailq_factor_kernel  fixed-order sum, Lx and inv(Lx) (the diagonal-block routines of the one-launch fit), T1 = [1'Y / sqrt n ;
//                        inv(Lx) Xc'Y] -- Q1' Y without Q1                                                     (one workgroup)
//   tailq_q1_kernel      per 64 rows: Qx = Xc inv(Lx)', Q1, B = Y - Q1 T1, the extra rows                      (npad / 64 workgroups)
struct TailQ {
    const double *Xc, *Y;
    int n, npad, d, dpad, q, k;
    int np, rows;       // partial sums: np <= 32 chunks of `rows` rows (a multiple of 16)
    double *part;       // [np][TQ_PS]: lower tiles of Xc'Xc as [a + 64 b] | Xc'Y as 4096 + [c + 64 l] | 1'Y as 4096 + 1024 + [l]
    double *Gx, *LinvX; // 128 x 128, identity padded
    double *T1;         // q x k, column-major
    double *Q1, *B;     // npad x q, npad x k
    double *rows_out;   // the matrix: right-hand sides as rows npad .. npad + xt - 1 (nullptr: none)
    int64_t ld;
    int xt;
    int *flags;
    double rsn;
};
constexpr int TQ_PS = 4096 + 1024 + 16;

__global__ __launch_bounds__(256) void tailq_gram_kernel(TailQ t) {
    __shared__ double red[4];
    __shared__ double part4[4][64];
    const int tid = threadIdx.x, p = blockIdx.x;
    const int n16 = (t.n + 15) & ~15, d16 = (t.d + 15) & ~15;
    const int r0 = p * t.rows, r1 = min(r0 + t.rows, n16), K = max(r1 - r0, 0);
    double *out = t.part + (size_t)p * TQ_PS;
    const double *X0 = t.Xc + (int64_t)r0 * t.dpad;
    if (K > 0) {
        wg_gemm<false, true, true>(d16, d16, K, X0, t.dpad, X0, t.dpad, NoPre(), [&](int a, int b, double v, double) { gst(&out[a + 64 * b], v); });
    } else {
        for (int e = tid; e < 4096; e += 256) gst(&out[e], 0.0);
    }
    // Xc' Y and 1' Y of the chunk: thread (c, g) takes rows r0 + g, r0 + g + 4, ..; the four groups are added in a fixed order
    const int c = tid & 63, g = tid >> 6;
    const int rend = min(r1, t.n);
    for (int l = 0; l < t.k; ++l) {
        double s = 0.0, sy = 0.0;
        for (int i0 = r0 + g; i0 < rend; i0 += 32) {  // eight rows' loads in flight (a load per iteration is a memory round trip each)
            double xv[8], yv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 4 * u < rend ? i0 + 4 * u : i0;
                xv[u] = gld(&X0[(int64_t)(i - r0) * t.dpad + c]);
                yv[u] = gld(&t.Y[(int64_t)i * t.k + l]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 4 * u < rend) {
                    s = fma(xv[u], yv[u], s);
                    sy += yv[u];
                }
        }
        __syncthreads();
        part4[g][c] = s;
        if (c == 0) red[g] = sy;
        __syncthreads();
        if (g == 0) gst(&out[4096 + c + 64 * l], (part4[0][c] + part4[1][c]) + (part4[2][c] + part4[3][c]));
        if (tid == 0) gst(&out[4096 + 1024 + l], (red[0] + red[1]) + (red[2] + red[3]));
    }
}

__global__ __launch_bounds__(256) void tailq_factor_kernel(TailQ t) {
    __shared__ double colbuf[2 * 128];
    __shared__ double Li[64 * 65];  // inv(Lx) [a][b]
    __shared__ double xty[64 * 16 + 16];
    const int tid = threadIdx.x, d = t.d, i = tid & 63, w = tid >> 6;
    // one workgroup on the critical path of the fit's front end, beside the Gram kernel's two workgroups per CU: first in line at the
    // SIMDs' issue ports (without: 37-55 us for 32 columns, a third of the issue slots)
    __builtin_amdgcn_s_setprio(3);
    // fixed-order sums of the partials (np <= 32: sixteen loads of an element in flight at once)
    double g[16], e[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) g[u] = 0.0;
    for (int p0 = 0; p0 < t.np; p0 += 4) {  // four partials x sixteen entries in flight per round trip; p ascending for every entry
        double x[4][16];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = 4 * u + w;
                x[p][u] = (p0 + p < t.np && i < d && j <= i) ? gld(&t.part[(size_t)(p0 + p) * TQ_PS + i + 64 * j]) : 0.0;
            }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int u = 0; u < 16; ++u) g[u] += x[p][u];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = 4 * u + w;
        if (!(i < d && j <= i)) g[u] = i == j ? 1.0 : 0.0;
        e[u] = i == j ? 1.0 : 0.0;
    }
    for (int idx = tid; idx < 64 * t.k + t.k; idx += 256) {
        const int src = idx < 64 * t.k ? 4096 + idx : 4096 + 1024 + (idx - 64 * t.k);
        double v = 0.0;
        for (int p0 = 0; p0 < t.np; p0 += 16) {
            double x[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) x[p] = p0 + p < t.np ? gld(&t.part[(size_t)(p0 + p) * TQ_PS + src]) : 0.0;
#pragma unroll
            for (int p = 0; p < 16; ++p) v += x[p];
        }
        xty[idx < 64 * t.k ? idx : 64 * 16 + (idx - 64 * t.k)] = v;
    }
    const int bad = chol64_regs(g, e, d, colbuf);
    if (tid == 0) t.flags[1] = bad;  // > 0: affinely dependent sites (Pi is rank deficient)
    // inv(Lx)(a, b) = E(b, a): to LDS for T1 and, identity padded to 128 x 128, to memory for the later launches
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = 4 * u + w;  // this thread holds E(i, j) = inv(Lx)(j, i)
        Li[j * 65 + i] = (j < d && i <= j) ? e[u] : 0.0;
    }
    __syncthreads();
    for (int idx = tid; idx < 128 * 128; idx += 256) {
        const int a = idx & 127, b = idx >> 7;
        gst(&t.LinvX[idx], (a < d && b < d) ? Li[a * 65 + b] : (a == b ? 1.0 : 0.0));
    }
    // T1 = Q1' Y = [1'Y / sqrt n ; inv(Lx) (Xc' Y)]
    for (int idx = tid; idx < t.q * t.k; idx += 256) {
        const int a = idx % t.q, l = idx / t.q;
        double v;
        if (a == 0) {
            v = xty[64 * 16 + l] * t.rsn;
        } else {
            v = 0.0;
            for (int b = 0; b < a; ++b) v = fma(Li[(a - 1) * 65 + b], xty[b + 64 * l], v);
        }
        gst(&t.T1[a + (int64_t)l * t.q], v);
    }
}

__global__ __launch_bounds__(256) void tailq_q1_kernel(TailQ t) {
    __shared__ double Qs[64 * 65];  // [a][i]
    __shared__ double part4[4][64];
    const int tid = threadIdx.x, d = t.d, d16 = (t.d + 15) & ~15;
    const int R0 = 64 * blockIdx.x;
    for (int e = tid; e < 64 * 65; e += 256) Qs[e] = 0.0;
    __syncthreads();
    // Qx(i, a) = sum_b Xc(i, b) inv(Lx)(a, b); rows beyond n are zero rows of Xc
    wg_gemm<true, true, false>(64, d16, d16, t.Xc + (int64_t)R0 * t.dpad, t.dpad, t.LinvX, 128, NoPre(), [&](int i, int a, double v, double) {
        if (a < d) {
            const double w = R0 + i < t.n ? v : 0.0;
            Qs[a * 65 + i] = w;
            gst(&t.Q1[(R0 + i) + (int64_t)(1 + a) * t.npad], w);
        }
    });
    if (tid < 64) gst(&t.Q1[R0 + tid], R0 + tid < t.n ? t.rsn : 0.0);
    __syncthreads();
    // B = Y - Q1 T1 (rows beyond n: zero), also as rows npad + l of the matrix
    const int i = tid & 63, g = tid >> 6;
    for (int l = 0; l < t.k; ++l) {
        double s = 0.0;
        for (int a = g; a < d; a += 4) s = fma(Qs[a * 65 + i], gld(&t.T1[1 + a + (int64_t)l * t.q]), s);
        __syncthreads();
        part4[g][i] = s;
        __syncthreads();
        if (g == 0) {
            double b = 0.0;
            if (R0 + i < t.n)
                b = gld(&t.Y[(int64_t)(R0 + i) * t.k + l]) - fma(t.rsn, gld(&t.T1[(int64_t)l * t.q]), (part4[0][i] + part4[1][i]) + (part4[2][i] + part4[3][i]));
            gst(&t.B[(R0 + i) + (int64_t)l * t.npad], b);
            if (t.rows_out) gst(&t.rows_out[(t.npad + l) + (int64_t)(R0 + i) * t.ld], b);
        }
    }
    if (t.rows_out)
        for (int e = tid; e < 64 * t.xt; e += 256) {
            const int l = e % t.xt, r = e / t.xt;
            if (l >= t.k) gst(&t.rows_out[(t.npad + l) + (int64_t)(R0 + r) * t.ld], 0.0);
        }
}

}  // namespace smallfit

bool small_fit_applies(const mrbf_ctx *ctx, int64_t n, int d, int k, int q, int path) {
    static const int off = mrbf_env("MRBF_SMALL_FIT") ? atoi(mrbf_env("MRBF_SMALL_FIT")) == 0 : 0;
    if (off || ctx->chol_impl == 1 || ctx->gram_mode == 1) return false;
    return n >= 1 && n <= 512 && d >= 1 && d <= 128 && k >= 1 && k <= 16 && n > q && (path == MRBF_PATH_CHOL || path == MRBF_PATH_PROJ_CHOL);
}

int small_fit_cluster(const mrbf_ctx *ctx, int count) {
    static const int env = mrbf_env("MRBF_SMALL_NC") ? atoi(mrbf_env("MRBF_SMALL_NC")) : 0;
    if (env == 1 || ctx->small_nc == 1 || !ctx->small_cluster_ok) return 1;
    const int groups = (std::max(count, 1) + 7) / 8;  // clusters are dealt in groups of eight problems (one per XCD)
    int nc = 1;
    while (2 * nc <= smallfit::MAX_CLUSTER && 8 * groups * 2 * nc <= ctx->ncu) nc *= 2;
    if (env > 1) nc = std::min(env, smallfit::MAX_CLUSTER);
    int p2 = 1;
    while (2 * p2 <= nc) p2 *= 2;
    return p2;
}

int launch_small_fit(mrbf_ctx *ctx, const smallfit::Prob *host_probs, int count, const smallfit::Prob *dev_probs, int nc) {
    if (count <= 0) return 0;
    const unsigned grid = nc == 1 ? (unsigned)count : 8u * (unsigned)nc * (unsigned)((count + 7) / 8);
    hipLaunchKernelGGL(smallfit::small_fit_kernel, dim3(grid), dim3(256), 0, ctx->stream, host_probs[0],
                       (count == 1 && !dev_probs) ? (const smallfit::Prob *)nullptr : dev_probs, count, nc);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

int launch_small_means(mrbf_ctx *ctx, const smallfit::Prob *dev_probs, int count) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(smallfit::small_mean_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, dev_probs, count);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

// The tail basis and the projected right-hand sides in three launches (see TailQ above); applies to d <= 64 (dpad 64), k <= 16, q = d + 1.
bool tail_basis_applies(const mrbf_model *M) {
    static const int on = mrbf_env("MRBF_TAILQ") ? atoi(mrbf_env("MRBF_TAILQ")) : 1;
    return on && M->dpad == 64 && M->d >= 1 && M->d <= 64 && M->q == M->d + 1 && M->k >= 1 && M->k <= 16 && M->npad % 64 == 0;
}
int launch_tail_basis(mrbf_ctx *ctx, const mrbf_model *M, const double *Y, double *scratch, double *LinvX, double *T1, double *Q1, double *B,
                      double *rows_out, int64_t ld, int xt, int *flags) {
    smallfit::TailQ t;
    t.Xc = M->Xc;
    t.Y = Y;
    t.n = (int)M->n;
    t.npad = (int)M->npad;
    t.d = M->d;
    t.dpad = M->dpad;
    t.q = M->q;
    t.k = M->k;
    const int n16 = (t.n + 15) & ~15;
    t.np = std::min(32, std::max(4, t.n / 128));
    t.rows = ((n16 + t.np - 1) / t.np + 15) & ~15;
    t.Gx = scratch;
    t.part = scratch + 128 * 128;
    t.LinvX = LinvX;
    t.T1 = T1;
    t.Q1 = Q1;
    t.B = B;
    t.rows_out = rows_out;
    t.ld = ld;
    t.xt = xt;
    t.flags = flags;
    t.rsn = 1.0 / std::sqrt((double)M->n);
    hipLaunchKernelGGL(smallfit::tailq_gram_kernel, dim3((unsigned)t.np), dim3(256), 0, ctx->stream, t);
    hipLaunchKernelGGL(smallfit::tailq_factor_kernel, dim3(1), dim3(256), 0, ctx->stream, t);
    hipLaunchKernelGGL(smallfit::tailq_q1_kernel, dim3((unsigned)(M->npad / 64)), dim3(256), 0, ctx->stream, t);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}
size_t tail_basis_scratch_doubles() { return (size_t)128 * 128 + (size_t)32 * smallfit::TQ_PS; }

}  // namespace mrbf
