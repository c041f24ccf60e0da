// The one-barrier decision kernel of the round-4 walk (included by round4.hip; tools/walklab times it stand-alone).
#pragma once
#include "radial.hpp"

namespace mrbf {
namespace r4 {
constexpr int SB = 128;  // block of candidates whose decisions are taken inside one kernel (also the tile of the block Schur complement)

// ---- Round 5, third form of the decision kernel.  The register kernel (round4.hip) still moved ~80 LDS reads per wave and candidate
// (column j of S re-read for every owned row, pi and g broadcast per owned column, 16 partial sums per row of G pi, 16 partial sums of
// pi' g) through the one LDS port of the CU -- 16 waves x 80 x 512 bytes at 128 bytes per clock = 2 of the 3.8 us per candidate -- and
// crossed three workgroup barriers.  Here, with 8 waves:
//   S: thread t owns the 4 x 8 tile rows 4 (t >> 4) .., columns 8 (t & 15) ..: a rank-1 step needs 4 + 8 entries of row j per thread,
//      six 16-byte LDS reads (a 16 x 2 strip per thread needed 18 values: as LDS broadcasts they saturated the LDS port, as v_readlane
//      the VALU); S is kept whole (both triangles), so row j -- sixteen consecutive lanes of ONE wave hold it -- is what gets published;
//   G: the 16-lane row rho of wave w owns rows w + 8 rho + 32 a, its lane l columns l + 16 b: (G pi)[t] is a sum over ONE 16-lane row
//      (four DPP rotations), the result is row-uniform -- exactly where the Sherman-Morrison update needs it -- and pi' g is the same
//      16-lane reduction repeated by every row on the same data (same order: same bits everywhere);
//   pi of the whole block is staged in LDS up front: the loop issues no global load.
// One barrier per candidate (s_waitcnt lgkmcnt(0) + s_barrier).  The accept test is a product (the quotient only when the two sides are
// within 1e-13 of each other), the rank-1 step uses one reciprocal of the pivot, nothing is masked (rows and columns <= j are dead: a row
// that was eliminated holds ~0 from its own step on, a rejected row stays a consistent row of the complement).  Same arithmetic per entry
// as the register kernel up to the order of the q-term sums and the rounding of l_r l_c against s_rj s_jc / s_jj.
// tools/walklab times the kernel stand-alone with per-phase cycle counts (PROF).
__device__ __forceinline__ unsigned dlo(double x) { return (unsigned)(unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ unsigned dhi(double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32); }
__device__ __forceinline__ double dmk(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double x) {  // (bound_ctrl with full masks: every lane is written, no `old` value to set up)
    return dmk((unsigned)__builtin_amdgcn_update_dpp(0, (int)dlo(x), CTRL, 0xf, 0xf, true),
               (unsigned)__builtin_amdgcn_update_dpp(0, (int)dhi(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double row_allreduce(double x) {  // every lane of a 16-lane row: the row's sum (rotations by 8, 4, 2, 1)
    x += dpp_d<0x128>(x);
    x += dpp_d<0x124>(x);
    x += dpp_d<0x122>(x);
    x += dpp_d<0x121>(x);
    return x;
}
__device__ __forceinline__ double readlane_d(double x, int l) {
    return dmk((unsigned)__builtin_amdgcn_readlane((int)dlo(x), l), (unsigned)__builtin_amdgcn_readlane((int)dhi(x), l));
}
template <int TW, int NA, int NB_, bool PROF = false>  // waves (8); rows of G per 16-lane row (q <= 32 NA); columns of G per lane (q <= 16 NB_)
__global__ __launch_bounds__(64 * TW) void select_block_walk_kernel(const double *__restrict__ Sg, int b, int64_t i0, int n0, int q, int max_points,
                                                                    int maxacc, double thr, const double *__restrict__ Prow,
                                                                    double *__restrict__ Ginv, int *__restrict__ acc, int *__restrict__ cnt,
                                                                    double *__restrict__ Lblk, int *__restrict__ blkidx,
                                                                    unsigned long long *__restrict__ prof = nullptr) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    static_assert(SB == 128 && TW == 8 && NB_ <= 12, "thread -> entry maps");
    constexpr int GL = (4 * TW * NA > 16 * NB_) ? 4 * TW * NA : 16 * NB_;  // rows of G of a 16-lane row: w + TW rho + 4 TW a
    __shared__ __attribute__((aligned(16))) double cj_s[2][SB];  // row j of S (double buffered: a slow wave may still read the previous candidate's)
    __shared__ double g_s[2][GL];                                  // g = G pi
    extern __shared__ double pi_s[];  // pi of the block's candidates, [b][16 NB_] (zero beyond q)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, rho = lane >> 4, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ri = tid >> 4, ci = tid & 15;  // S tile: rows 4 ri + a, columns 8 ci + c
    // (read through the transpose, 64 contiguous bytes per thread and row: the block is symmetric up to the rounding of the products that
    // formed it, either triangle is as good)
    double Sr[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const v2d t = *reinterpret_cast<const v2d *>(Sg + (8 * ci + c) + (4 * ri + a) * SB);
            Sr[a][c] = t.x;
            Sr[a][c + 1] = t.y;
        }
    double G[NA][NB_];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int bb = 0; bb < NB_; ++bb) {
            const int t = w + TW * rho + 4 * TW * a, u = l15 + 16 * bb;
            G[a][bb] = (t < q && u < q) ? Ginv[t + (int64_t)u * q] : 0.0;
        }
    int nacc = cnt[0], nblk = 0;
    for (int e0 = 0; e0 < b * 16 * NB_; e0 += 8 * 64 * TW) {  // (eight loads in flight per thread: one at a time is 20 L2 round trips)
        double t[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int e = e0 + tid + x * 64 * TW, jj = e / (16 * NB_), u = e % (16 * NB_);
            t[x] = (e < b * 16 * NB_ && u < q) ? Prow[(i0 + jj) * q + u] : 0.0;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int e = e0 + tid + x * 64 * TW;
            if (e < b * 16 * NB_) pi_s[e] = t[x];
        }
    }
    for (int e = tid; e < 2 * GL; e += 64 * TW) (&g_s[0][0])[e] = 0.0;
    // every load of the prologue has landed before the loop (the compiler otherwise merges the prologue's pending loads into the loop header
    // and drains the memory queue at the top of every step)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(Sr[a][c]));
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int bb = 0; bb < NB_; ++bb) asm volatile("" : "+v"(G[a][bb]));
    __syncthreads();
    unsigned long long tprof[4] = {0, 0, 0, 0}, tlast = 0;
    if constexpr (PROF) tlast = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int ph) {
        if constexpr (PROF) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            tprof[ph] += t - tlast;
            tlast = t;
        }
    };
    for (int j = 0; j < b; ++j) {
        if (n0 + nacc >= max_points || nacc >= maxacc) break;
        double p0[NB_];
#pragma unroll
        for (int bb = 0; bb < NB_; ++bb) p0[bb] = pi_s[j * 16 * NB_ + l15 + 16 * bb];
        const int par = j & 1;
        if (ri == (j >> 2)) {  // the sixteen lanes that hold row j
            v2d *dst = reinterpret_cast<v2d *>(&cj_s[par][8 * ci]);
            switch (j & 3) {
#define MRBF_WALK_ROW(A_)                                  \
    case A_:                                               \
        dst[0] = (v2d){Sr[A_][0], Sr[A_][1]};              \
        dst[1] = (v2d){Sr[A_][2], Sr[A_][3]};              \
        dst[2] = (v2d){Sr[A_][4], Sr[A_][5]};              \
        dst[3] = (v2d){Sr[A_][6], Sr[A_][7]};              \
        break;
                MRBF_WALK_ROW(0) MRBF_WALK_ROW(1) MRBF_WALK_ROW(2) MRBF_WALK_ROW(3)
#undef MRBF_WALK_ROW
            }
        }
        double v[NA];  // g at this 16-lane row's rows
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            double sp = 0.0;
#pragma unroll
            for (int bb = 0; bb < NB_; ++bb) sp = fma(G[a][bb], p0[bb], sp);
            v[a] = row_allreduce(sp);
        }
        if (l15 == 0) {
#pragma unroll
            for (int a = 0; a < NA; ++a) g_s[par][w + TW * rho + 4 * TW * a] = v[a];
        }
        stamp(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(1);
        double gu[NB_];
#pragma unroll
        for (int bb = 0; bb < NB_; ++bb) gu[bb] = g_s[par][l15 + 16 * bb];
        const double pk = cj_s[par][j];
        double ur[4], uc[8];  // row j at this tile's rows and columns
#pragma unroll
        for (int a = 0; a < 4; a += 2) {
            const v2d t = *reinterpret_cast<const v2d *>(&cj_s[par][4 * ri + a]);
            ur[a] = t.x;
            ur[a + 1] = t.y;
        }
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const v2d t = *reinterpret_cast<const v2d *>(&cj_s[par][8 * ci + c]);
            uc[c] = t.x;
            uc[c + 1] = t.y;
        }
        double pp = 0.0;
#pragma unroll
        for (int bb = 0; bb < NB_; ++bb) pp = fma(p0[bb], gu[bb], pp);
        const double ph = 1.0 + row_allreduce(pp);
        // tau^2 = pk / ph > (theta^2)^2, RbfModel.jl:370, :452 (NaN fails): decided by the product unless the two sides are within 1e-13
        const double tp = thr * ph;
        bool accept = pk > 0.0 && pk > tp && pk < 1e300 * ph;
        if (fabs(pk - tp) <= 1e-13 * tp) {
            const double tau2 = pk / ph;
            accept = pk > 0.0 && tau2 > thr && tau2 < 1e300;
        }
        stamp(2);
        if (!accept) continue;
        if (tid < SB) {  // the factor column (two waves): L(r, j) = S(r, j) / sqrt(pk)
            double sq_, rs;
            fast_sqrt_rsqrt(pk, sq_, rs);
            (void)sq_;
            Lblk[tid + nblk * SB] = (tid > j && tid < b) ? cj_s[par][tid] * rs : (tid == j ? pk * rs : 0.0);
        }
        // S(r, c) -= S(r, j) S(j, c) / pk for every entry, one reciprocal per candidate
        double rpk = __builtin_amdgcn_rcp(pk);
        rpk = fma(fma(-pk, rpk, 1.0), rpk, rpk);
        rpk = fma(fma(-pk, rpk, 1.0), rpk, rpk);
#pragma unroll
        for (int c = 0; c < 8; ++c) uc[c] *= rpk;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 8; ++c) Sr[a][c] = fma(-ur[a], uc[c], Sr[a][c]);
        // Ginv <- Ginv - g g' / s_H   (Sherman-Morrison for G + pi pi'), one reciprocal per candidate as in the register kernel
        double rph = __builtin_amdgcn_rcp(ph);
        rph = fma(fma(-ph, rph, 1.0), rph, rph);
        rph = fma(fma(-ph, rph, 1.0), rph, rph);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const double gt = v[a] * rph;
#pragma unroll
            for (int bb = 0; bb < NB_; ++bb) G[a][bb] = fma(-gt, gu[bb], G[a][bb]);
        }
        if (tid == 0) {
            acc[nacc] = (int)(i0 + j);
            blkidx[nblk] = j;
        }
        ++nacc;
        ++nblk;
        stamp(3);
    }
    if constexpr (PROF) {
        if (lane == 0 && prof)
            for (int ph = 0; ph < 4; ++ph) prof[w * 4 + ph] = tprof[ph];
    }
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int bb = 0; bb < NB_; ++bb) {
            const int t = w + TW * rho + 4 * TW * a, u = l15 + 16 * bb;
            if (t < q && u < q) Ginv[t + (int64_t)u * q] = G[a][bb];
        }
    if (tid == 0) {
        cnt[0] = nacc;
        cnt[1] = nblk;
        acc[maxacc] = nacc;
    }
}

// tau^2 = pk / ph > (theta^2)^2, RbfModel.jl:370, :452 (NaN fails): decided by the product unless the two sides are within 1e-13
__device__ __forceinline__ bool walk_accept(double pk, double ph, double thr) {
    const double tp = thr * ph;
    bool accept = pk > 0.0 && pk > tp && pk < 1e300 * ph;
    if (fabs(pk - tp) <= 1e-13 * tp) {
        const double tau2 = pk / ph;
        accept = pk > 0.0 && tau2 > thr && tau2 < 1e300;
    }
    return accept;
}

// ---- The same walk with the two halves of a step on different waves.  In the kernel above every wave does its share of both updates one
// after the other, and a step is one dependent chain: G pi -> row sums -> barrier -> pi' g -> decision -> S update -> G update (1.1 us per
// candidate at q = 65, 1.5 at q = 129, VALU issue and latency about even).  Here waves 0-3 own S (8 x 8 tiles), waves 4-7 own G (rows
// w + 4 rho + 16 a of the 16-lane row rho, columns l + 16 b): each SIMD holds one wave of either kind, the S update runs under the other
// wave's G update and row sums, and both kinds take the decision themselves from the same published g and row j (same sums in the same
// order: same bits).  One barrier per candidate, crossed by all eight waves.
template <int N, bool PROF = false>  // q <= 16 N
__global__ __launch_bounds__(512) void select_block_duo_kernel(const double *__restrict__ Sg, int b, int64_t i0, int n0, int q, int max_points, int maxacc,
                                                               double thr, const double *__restrict__ Prow, double *__restrict__ Ginv,
                                                               int *__restrict__ acc, int *__restrict__ cnt, double *__restrict__ Lblk,
                                                               int *__restrict__ blkidx, unsigned long long *__restrict__ prof = nullptr) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    static_assert(SB == 128 && N <= 9, "thread -> entry maps");
    constexpr int GL = 16 * N;
    __shared__ __attribute__((aligned(16))) double cj_s[2][SB];  // row j of S, double buffered
    __shared__ double g_s[2][GL];                                  // g = G pi
    extern __shared__ double pi_s[];                               // pi of the block's candidates, [b][16 N] (zero beyond q)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, rho = lane >> 4, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nacc = cnt[0], nblk = 0;
    for (int e0 = 0; e0 < b * GL; e0 += 8 * 512) {  // (eight loads in flight per thread: one at a time is 20 L2 round trips)
        double t[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int e = e0 + tid + x * 512, jj = e / GL, u = e % GL;
            t[x] = (e < b * GL && u < q) ? Prow[(i0 + jj) * q + u] : 0.0;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int e = e0 + tid + x * 512;
            if (e < b * GL) pi_s[e] = t[x];
        }
    }
    unsigned long long tprof[4] = {0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int ph) {
        if constexpr (PROF) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            tprof[ph] += t - tlast;
            tlast = t;
        }
    };
    if (w < 4) {
        // ---- S waves: thread t owns rows 8 (t >> 4) .., columns 8 (t & 15) .. (read through the transpose: 64 contiguous bytes per row)
        const int ri = tid >> 4, ci = tid & 15;
        double Sr[8][8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const v2d t = *reinterpret_cast<const v2d *>(Sg + (8 * ci + c) + (8 * ri + a) * SB);
                Sr[a][c] = t.x;
                Sr[a][c + 1] = t.y;
            }
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(Sr[a][c]));
        __syncthreads();
        if constexpr (PROF) tlast = __builtin_amdgcn_s_memtime();
        for (int j = 0; j < b; ++j) {
            if (n0 + nacc >= max_points || nacc >= maxacc) break;
            const int par = j & 1;
            if (ri == (j >> 3)) {  // the sixteen lanes that hold row j
                v2d *dst = reinterpret_cast<v2d *>(&cj_s[par][8 * ci]);
                switch (j & 7) {
#define MRBF_DUO_ROW(A_)                      \
    case A_:                                  \
        dst[0] = (v2d){Sr[A_][0], Sr[A_][1]}; \
        dst[1] = (v2d){Sr[A_][2], Sr[A_][3]}; \
        dst[2] = (v2d){Sr[A_][4], Sr[A_][5]}; \
        dst[3] = (v2d){Sr[A_][6], Sr[A_][7]}; \
        break;
                    MRBF_DUO_ROW(0) MRBF_DUO_ROW(1) MRBF_DUO_ROW(2) MRBF_DUO_ROW(3) MRBF_DUO_ROW(4) MRBF_DUO_ROW(5) MRBF_DUO_ROW(6) MRBF_DUO_ROW(7)
#undef MRBF_DUO_ROW
                }
            }
            stamp(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stamp(1);
            double pp = 0.0;
#pragma unroll
            for (int bb = 0; bb < N; ++bb) pp = fma(pi_s[j * GL + l15 + 16 * bb], g_s[par][l15 + 16 * bb], pp);
            const double pk = cj_s[par][j];
            double ur[8], uc[8];  // row j at this tile's rows and columns
#pragma unroll
            for (int a = 0; a < 8; a += 2) {
                const v2d t = *reinterpret_cast<const v2d *>(&cj_s[par][8 * ri + a]);
                ur[a] = t.x;
                ur[a + 1] = t.y;
            }
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const v2d t = *reinterpret_cast<const v2d *>(&cj_s[par][8 * ci + c]);
                uc[c] = t.x;
                uc[c + 1] = t.y;
            }
            const double ph = 1.0 + row_allreduce(pp);
            const bool accept = walk_accept(pk, ph, thr);
            stamp(2);
            if (!accept) continue;
            if (tid < SB) {  // the factor column (two waves): L(r, j) = S(r, j) / sqrt(pk)
                double sq_, rs;
                fast_sqrt_rsqrt(pk, sq_, rs);
                (void)sq_;
                Lblk[tid + nblk * SB] = (tid > j && tid < b) ? cj_s[par][tid] * rs : (tid == j ? pk * rs : 0.0);
            }
            // S(r, c) -= S(r, j) S(j, c) / pk for every entry, one reciprocal per candidate
            double rpk = __builtin_amdgcn_rcp(pk);
            rpk = fma(fma(-pk, rpk, 1.0), rpk, rpk);
            rpk = fma(fma(-pk, rpk, 1.0), rpk, rpk);
#pragma unroll
            for (int c = 0; c < 8; ++c) uc[c] *= rpk;
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int c = 0; c < 8; ++c) Sr[a][c] = fma(-ur[a], uc[c], Sr[a][c]);
            if (tid == 0) {
                acc[nacc] = (int)(i0 + j);
                blkidx[nblk] = j;
            }
            ++nacc;
            ++nblk;
            stamp(3);
        }
        if (tid == 0) {
            cnt[0] = nacc;
            cnt[1] = nblk;
            acc[maxacc] = nacc;
        }
    } else {
        // ---- G waves
        const int w2 = w - 4;
        double G[N][N];
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int bb = 0; bb < N; ++bb) {
                const int t = w2 + 4 * rho + 16 * a, u = l15 + 16 * bb;
                G[a][bb] = (t < q && u < q) ? Ginv[t + (int64_t)u * q] : 0.0;
            }
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int bb = 0; bb < N; ++bb) asm volatile("" : "+v"(G[a][bb]));
        __syncthreads();
        if constexpr (PROF) tlast = __builtin_amdgcn_s_memtime();
        double p0[N];
#pragma unroll
        for (int bb = 0; bb < N; ++bb) p0[bb] = pi_s[l15 + 16 * bb];
        for (int j = 0; j < b; ++j) {
            if (n0 + nacc >= max_points || nacc >= maxacc) break;
            const int par = j & 1;
            double v[N];  // g at this 16-lane row's rows
#pragma unroll
            for (int a = 0; a < N; ++a) {
                double sp = 0.0;
#pragma unroll
                for (int bb = 0; bb < N; ++bb) sp = fma(G[a][bb], p0[bb], sp);
                v[a] = sp;
            }
#pragma unroll
            for (int a = 0; a < N; ++a) v[a] = row_allreduce(v[a]);
            if (l15 == 0) {
#pragma unroll
                for (int a = 0; a < N; ++a) g_s[par][w2 + 4 * rho + 16 * a] = v[a];
            }
            stamp(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stamp(1);
            double gu[N];
#pragma unroll
            for (int bb = 0; bb < N; ++bb) gu[bb] = g_s[par][l15 + 16 * bb];
            const double pk = cj_s[par][j];
            double pp = 0.0;
#pragma unroll
            for (int bb = 0; bb < N; ++bb) pp = fma(p0[bb], gu[bb], pp);
            if (j + 1 < b) {  // (the next candidate's pi: in flight under the decision)
#pragma unroll
                for (int bb = 0; bb < N; ++bb) p0[bb] = pi_s[(j + 1) * GL + l15 + 16 * bb];
            }
            const double ph = 1.0 + row_allreduce(pp);
            const bool accept = walk_accept(pk, ph, thr);
            stamp(2);
            if (!accept) continue;
            // Ginv <- Ginv - g g' / s_H   (Sherman-Morrison for G + pi pi'), one reciprocal per candidate as in the register kernel
            double rph = __builtin_amdgcn_rcp(ph);
            rph = fma(fma(-ph, rph, 1.0), rph, rph);
            rph = fma(fma(-ph, rph, 1.0), rph, rph);
#pragma unroll
            for (int a = 0; a < N; ++a) {
                const double gt = v[a] * rph;
#pragma unroll
                for (int bb = 0; bb < N; ++bb) G[a][bb] = fma(-gt, gu[bb], G[a][bb]);
            }
            ++nacc;
            ++nblk;
            stamp(3);
        }
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int bb = 0; bb < N; ++bb) {
                const int t = w2 + 4 * rho + 16 * a, u = l15 + 16 * bb;
                if (t < q && u < q) Ginv[t + (int64_t)u * q] = G[a][bb];
            }
    }
    if constexpr (PROF) {
        if (lane == 0 && prof)
            for (int ph = 0; ph < 4; ++ph) prof[w * 4 + ph] = tprof[ph];
    }
}

}  // namespace r4
}  // namespace mrbf
