// Device core of the 128 x 128 diagonal-block factorisation (Cholesky + inverse of the factor), shared by the
// stand-alone kernel (chol_diag.hip) and the persistent factorisation kernel (chol_mega.hip).
//
// The block is cut into 16 x 16 tiles (36 lower tiles).  Every tile lives TRANSPOSED in the f64 MFMA C/D register
// layout of one of the 4 waves (lane l, register r hold A[16i + (l&15)][16j + (l>>4) + 4r]): in that layout a tile
// is directly the B operand of the next MFMA (its register r IS k-slice r), so the panel solve
//        P_i' = inv(L_bb) * A_ib'                (4 MFMAs per tile, A operand = the 16 x 16 leaf inverse from LDS)
// and the trailing update
//        A_ij' -= P_j * P_i'                     (4 MFMAs per tile, operands = finished panel tiles from LDS)
// never move data between lanes.  Only the 16 x 16 leaves are sequential: wave 0 factors them one column at a time
// with the rows of the leaf in lanes 0..15 and an identity block in lanes 16..31 (the same elimination turns the
// identity into L_bb^-T, i.e. the leaf inverse is free); columns are broadcast with v_readlane, no LDS, no barrier.
// After the 8 panels the full 128 x 128 inverse is assembled block column by block column with MFMAs
//        X_jj = inv(L_jj),   X_ij = -inv(L_ii) * sum_{p=j..i-1} L_ip X_pj.
#pragma once
#include "common.hpp"

namespace mrbf {
namespace diagcore {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gf64;  // explicit global address space: global_ (not flat_) loads / stores

constexpr int DNB = 128;
constexpr int NT = 8;      // 16 x 16 tiles per side
constexpr int TILE = 256;  // doubles per tile
// strictly-lower tile (i, j), j < i
#define MRBF_SIDX(i, j) ((i) * ((i)-1) / 2 + (j))
// lower tile incl. diagonal (register ownership map)
#define MRBF_TIDX(i, j) ((i) * ((i) + 1) / 2 + (j))

// Register ownership of the 36 lower tiles: wave 0 holds the 8 diagonal tiles (slot = i) and runs the sequential leaves;
// waves 1..3 hold the 28 off-diagonal tiles round-robin (slot = index / 3) and run the panel solves and trailing updates,
// which therefore overlap with the next leaf (look-ahead inside the block).
__host__ __device__ constexpr int tile_owner(int i, int j) { return i == j ? 0 : 1 + MRBF_SIDX(i, j) % 3; }
__host__ __device__ constexpr int tile_slot(int i, int j) { return i == j ? i : MRBF_SIDX(i, j) / 3; }
constexpr int NSLOT = 10;

struct DiagV4Shared {
    double LT[28 * TILE];  // finished strictly-lower L tiles, [tile][c][row]  (c = column inside the tile): MFMA operand order
    double IT[NT * TILE];  // leaf inverses transposed: IT[b][a2][a] = inv(L_bb)[a][a2]
    double Dt[TILE];       // diagonal tile handed to the leaf wave, [col a][row b']
    double Lb[TILE];       // factored diagonal leaf [col][row]: stored to global memory by another wave, off the leaf wave's path
    int bad;
};

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// v_rsq_f64 is good to ~2^-26 relative; one Newton step leaves ~1.5 * 2^-52 (under 2 ulp), two reach the last bit.  The pivot's
// reciprocal square root sits on the sequential chain of the whole factorisation (128 dependent leaf columns per block): the
// second step costs ~3 dependent f64 operations per column there and buys less than the rounding of the updates that follow
// (a 2-ulp perturbation of one pivot is a backward error of 2 ulp in that diagonal entry).  MRBF_DIAG_NEWTON=2 restores it.
#ifndef MRBF_DIAG_NEWTON
#define MRBF_DIAG_NEWTON 1
#endif
__device__ __forceinline__ double fast_rsqrt_v4(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
#pragma unroll
    for (int it = 0; it < MRBF_DIAG_NEWTON; ++it) y = y * fma(-h, y * y, 1.5);
    return y;
}

// operand fetch: element [4s + (l>>4)][l&15] of a 16 x 16 tile stored row-major ([k][c]); conflict-free ds_read_b64
__device__ __forceinline__ double opnd(const double *tile, int s, int l15, int l4) { return tile[(4 * s + l4) * 16 + l15]; }

// SC1: write-through stores (visible to other workgroups of the same launch after a flag hand-off)
template <bool SC1>
__device__ __forceinline__ void gstore(double *p, double v) {
    if constexpr (SC1)
        __hip_atomic_store((gf64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *(gf64 *)p = v;
}

// leaf: rows of the 16 x 16 s.p.d. block in lanes 0..15 (a[c], c <= lane valid), identity rows in lanes 16..31
template <int K>
__device__ __forceinline__ void leaf_step(double (&a)[16], int &bad, int col0) {
    // branch-free so that the 16 unrolled steps form one basic block: the scheduler can then start step K+1's
    // pivot -> rsqrt chain under the remaining column updates of step K
    const double piv = readlane_f64(a[K], K);
    const bool ok = piv > 0.0;
    const double rinv = ok ? fast_rsqrt_v4(ok ? piv : 1.0) : 0.0;
    bad = (!ok && bad == 0) ? col0 + K + 1 : bad;
    a[K] *= rinv;
#pragma unroll
    for (int j = K + 1; j < 16; ++j) {
        const double ljk = readlane_f64(a[K], j);
        a[j] = fma(-a[K], ljk, a[j]);
    }
}

// load phase (ownership: tile_owner / tile_slot)
__device__ __forceinline__ void diag_v4_load(const double *__restrict__ A, int64_t lda, v4d (&acc)[NSLOT]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            if (tile_owner(i, j) == wave) {
                v4d v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = *(const gf64 *)&A[(16 * i + l15) + (int64_t)(16 * j + l4 + 4 * r) * lda];
                acc[tile_slot(i, j)] = v;
            }
        }
}

// Factor the 128 x 128 block at A (lower part, column-major, leading dimension lda) in place and write the inverse
// of the factor to Linv (128 x 128 column-major, zeros above the diagonal).  256 threads.  Returns 0 or the 1-based
// index (inside the block) of the first non-positive pivot; in that case A / Linv are only partly written.
// PRELOADED: the caller hands the tile in registers (diag_v4_load layout).  STREAM: the 16 x 16 leaf inverses also go
// to itg (8 x 256 doubles, [b][k][j] = inv(L_bb)[j][k]) and *prog counts the 16-column panels whose entries of L (and
// leaf inverse) are visible to other workgroups -- a consumer can run the panel solve of the tiles below in step.
template <bool SC1, bool STREAM, bool PRELOADED>
__device__ __forceinline__ int diag_v4_core(double *__restrict__ A, int64_t lda, double *__restrict__ Linv, DiagV4Shared &sh,
                                            v4d (&acc)[NSLOT], double *__restrict__ itg, unsigned *prog, int dbg = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    if (tid == 0) sh.bad = 0;
    unsigned long long tseg[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;  // timing experiments (dbg & 4): wave 0's cycles per segment
#define MRBF_DSEG(k)                                         \
    do {                                                     \
        if (dbg & 4) {                                       \
            const unsigned long long now_ = __builtin_readcyclecounter(); \
            tseg[k] += now_ - tlast;                         \
            tlast = now_;                                    \
        }                                                    \
    } while (0)
    if (!PRELOADED) diag_v4_load(A, lda, acc);
    __syncthreads();
    if (dbg & 4) tlast = __builtin_readcyclecounter();

#pragma unroll 1
    for (int b = 0; b < NT; ++b) {
        if (wave == 0) {
            // ---- wave 0: finish diagonal tile (b,b) with panel b-1 (the only update the next leaf waits for), then the leaf
            if (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own write-through stores of the previous leaf
#pragma unroll
            for (int i = 1; i < NT; ++i)
                if (i == b) {
                    const double *pi_ = &sh.LT[MRBF_SIDX(i, i - 1) * TILE];
                    v4d c = acc[i];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(pi_, s, l15, l4), opnd(pi_, s, l15, l4), c, 0, 0, 0);
                    acc[i] = c;
                }
            // C/D layout -> leaf layout through LDS (wave-local: LDS operations of one wave complete in order)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (i == b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sh.Dt[(l4 + 4 * r) * 16 + l15] = acc[i][r];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double a[16];
            int bad = 0;
            MRBF_DSEG(0);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double v;
                if (lane < 16)
                    v = (c <= lane) ? sh.Dt[c * 16 + lane] : 0.0;  // A_bb[row = lane][col = c]
                else
                    v = (c == lane - 16) ? 1.0 : 0.0;
                a[c] = v;
            }
            if (!(dbg & 2)) {
            leaf_step<0>(a, bad, 16 * b);
            leaf_step<1>(a, bad, 16 * b);
            leaf_step<2>(a, bad, 16 * b);
            leaf_step<3>(a, bad, 16 * b);
            leaf_step<4>(a, bad, 16 * b);
            leaf_step<5>(a, bad, 16 * b);
            leaf_step<6>(a, bad, 16 * b);
            leaf_step<7>(a, bad, 16 * b);
            leaf_step<8>(a, bad, 16 * b);
            leaf_step<9>(a, bad, 16 * b);
            leaf_step<10>(a, bad, 16 * b);
            leaf_step<11>(a, bad, 16 * b);
            leaf_step<12>(a, bad, 16 * b);
            leaf_step<13>(a, bad, 16 * b);
            leaf_step<14>(a, bad, 16 * b);
            leaf_step<15>(a, bad, 16 * b);
            }
            MRBF_DSEG(1);
            if (bad && lane == 0) sh.bad = bad;
            // the leaf wave only writes LDS; waves 1 / 2 move the factored leaf and (STREAM) its inverse to global memory after
            // the barrier (32 predicated 8-byte global stores cost the leaf wave ~0.9 us per panel)
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) sh.Lb[c * 16 + lane] = a[c];
            } else if (lane < 32) {
                // lane 16 + r holds (L^-T)[r][c] = inv[c][r]:  IT[b][a2 = r][a = c]
                const int r = lane - 16;
#pragma unroll
                for (int c = 0; c < 16; ++c) sh.IT[b * TILE + r * 16 + c] = (c >= r) ? a[c] : 0.0;
            }
        } else {
            // ---- waves 1..3 (meanwhile): the rest of panel b-1's trailing update, off-diagonal tiles (i,j), i > j > b
            //      (block column b was brought up to date before the previous barrier)
#pragma unroll
            for (int i = 2; i < NT; ++i)
#pragma unroll
                for (int j = 1; j < i; ++j) {
                    if (b > 0 && j > b && tile_owner(i, j) == wave) {
                        const double *tj_ = &sh.LT[(j * (j - 1) / 2 + (b - 1)) * TILE];
                        const double *ti_ = &sh.LT[(i * (i - 1) / 2 + (b - 1)) * TILE];
                        v4d c = acc[tile_slot(i, j)];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            c = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(tj_, s, l15, l4), opnd(ti_, s, l15, l4), c, 0, 0, 0);
                        acc[tile_slot(i, j)] = c;
                    }
                }
            // panel b-1's write-through stores were issued a leaf ago: draining them here costs nothing
            if (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (wave == 0) MRBF_DSEG(2);
        __syncthreads();  // X: leaf inverse IT[b] visible
        if (wave == 0) MRBF_DSEG(3);
        if (sh.bad) break;
        if (STREAM && b > 0 && tid == 64)
            __hip_atomic_store((__attribute__((address_space(1))) unsigned *)prog, (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 1) {
            // factored leaf -> global (lower part of the diagonal tile), coalesced along rows
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = l4 + 4 * r;
                if (c <= l15) gstore<SC1>(&A[(16 * b + l15) + (int64_t)(16 * b + c) * lda], sh.Lb[c * 16 + l15]);
            }
        } else if (STREAM && wave == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gstore<true>(&itg[b * TILE + (l4 + 4 * r) * 16 + l15], sh.IT[b * TILE + (l4 + 4 * r) * 16 + l15]);
        }
        if (wave != 0) {
            // ---- panel: P_i' = inv(L_bb) * A_ib'  for the owned tiles of block column b
            const double *itb = &sh.IT[b * TILE];
            double ia[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ia[s] = opnd(itb, s, l15, l4);
#pragma unroll
            for (int i = 1; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    if (j == b && tile_owner(i, j) == wave) {
                        v4d p = {0.0, 0.0, 0.0, 0.0};
                        const v4d m = acc[tile_slot(i, j)];
#pragma unroll
                        for (int s = 0; s < 4; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[s], m[s], p, 0, 0, 0);
                        acc[tile_slot(i, j)] = p;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.LT[MRBF_SIDX(i, j) * TILE + (l4 + 4 * r) * 16 + l15] = p[r];
                    }
                }
        } else {
            // ---- wave 0 (meanwhile): panel b-1 on the diagonal tiles it has not reached yet, (j,j), j > b
#pragma unroll
            for (int j = 2; j < NT; ++j) {
                if (b > 0 && j > b) {
                    const double *pj_ = &sh.LT[(j * (j - 1) / 2 + (b - 1)) * TILE];
                    v4d c = acc[j];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(pj_, s, l15, l4), opnd(pj_, s, l15, l4), c, 0, 0, 0);
                    acc[j] = c;
                }
            }
        }
        if (wave == 0) MRBF_DSEG(4);
        __syncthreads();  // Y: panel tiles of block column b in LT
        if (wave == 0) MRBF_DSEG(5);
        if (wave != 0) {
            // L(i,b) -> global, behind the barrier the leaf wave waits at (coalesced along rows)
#pragma unroll
            for (int i = 1; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    if (j == b && tile_owner(i, j) == wave) {
                        const v4d pt = acc[tile_slot(i, j)];
#pragma unroll
                        for (int r = 0; r < 4; ++r) gstore<SC1>(&A[(16 * i + l15) + (int64_t)(16 * j + l4 + 4 * r) * lda], pt[r]);
                    }
                }
            // bring block column b+1 up to date first: the next panel solve needs it right after the next leaf
#pragma unroll
            for (int i = 2; i < NT; ++i)
#pragma unroll
                for (int j = 1; j < i; ++j) {
                    if (j == b + 1 && tile_owner(i, j) == wave) {
                        const double *tj_ = &sh.LT[(j * (j - 1) / 2 + b) * TILE];
                        const double *ti_ = &sh.LT[(i * (i - 1) / 2 + b) * TILE];
                        v4d c = acc[tile_slot(i, j)];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            c = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(tj_, s, l15, l4), opnd(ti_, s, l15, l4), c, 0, 0, 0);
                        acc[tile_slot(i, j)] = c;
                    }
                }
        }
    }
    if (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int bad_all = sh.bad;
    if (bad_all) return bad_all;
    if ((dbg & 4) && tid == 0) {
        for (int k = 0; k < 6; ++k) Linv[k] = (double)tseg[k];
    }
    if (dbg & 1) return 0;  // timing experiments: no inverse
    if (STREAM && tid == 64)
        __hip_atomic_store((__attribute__((address_space(1))) unsigned *)prog, (unsigned)NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- inverse: wave w assembles block columns w and 7 - w of X = L^-1 in registers, MFMA C/D layout
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = half == 0 ? wave : 7 - wave;
        v4d X[NT];
        // X_jj = inv(L_jj): C layout element [(l4 + 4r)][l15] = IT[j][l15][l4 + 4r]
#pragma unroll
        for (int r = 0; r < 4; ++r) X[0][r] = sh.IT[j * TILE + l15 * 16 + l4 + 4 * r];
#pragma unroll
        for (int r = 0; r < 4; ++r) gstore<SC1>(&Linv[(16 * j + l4 + 4 * r) + (16 * j + l15) * DNB], X[0][r]);
#pragma unroll
        for (int di = 1; di < NT; ++di) {
            const int i = j + di;
            if (i < NT) {
                v4d S = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int dp = 0; dp < di; ++dp) {
                    const int p = j + dp;
                    const double *lt = &sh.LT[MRBF_SIDX(i, p) * TILE];
#pragma unroll
                    for (int s = 0; s < 4; ++s) S = __builtin_amdgcn_mfma_f64_16x16x4f64(opnd(lt, s, l15, l4), X[dp][s], S, 0, 0, 0);
                }
                v4d Xi = {0.0, 0.0, 0.0, 0.0};
                const double *iti = &sh.IT[i * TILE];
#pragma unroll
                for (int s = 0; s < 4; ++s) Xi = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(iti, s, l15, l4), S[s], Xi, 0, 0, 0);
                X[di] = Xi;
#pragma unroll
                for (int r = 0; r < 4; ++r) gstore<SC1>(&Linv[(16 * i + l4 + 4 * r) + (16 * j + l15) * DNB], Xi[r]);
            }
        }
        // zeros above the diagonal of this block column (the panel GEMM reads the full 128 x 128 inverse)
        for (int e = lane; e < 16 * j * 16; e += 64) {
            const int row = e % (16 * j), col = e / (16 * j);
            gstore<SC1>(&Linv[row + (16 * j + col) * DNB], 0.0);
        }
    }
    return 0;
}

// =====================================================================================================================
// v5: the same factorisation with WAVE-SPECIALISED roles and LDS flags instead of workgroup barriers.
//
// In v4 every 16-column step passes two __syncthreads and the leaf wave also carries the deferred updates of the diagonal
// tiles: a step costs ~7 us (stand-alone 56 us per block, tools/diagbench) of which the sequential leaf is 2.4 us, and the
// 128-column block is the unit of the whole factorisation's critical chain (64 blocks at n = 8192: job log r02).  Here the
// critical chain of a step stays inside ONE wave:
//   wave 0        leaf b -> leaf inverse to LDS -> panel solve of tile (b+1,b) -> update of tile (b+1,b+1) with it (both MFMA
//                 operands are the solved tile's own accumulator registers) -> leaf b+1.  No other wave is waited for on this
//                 path: the two tiles arrive in LDS, already updated with the panels 0..b-1, while leaf b is running.
//   waves 1..3    own the 36 tiles by block ROW (row i -> wave 1 + i % 3, registers as in v4): the other panel solves, all
//                 trailing updates (nearest row first, so that row b+2's two leading tiles are staged for the leaf wave early),
//                 the global stores and the streamed publication of finished panels.
// Hand-offs are words in LDS (writer: data, s_waitcnt lgkmcnt(0), flag; reader: poll, then data -- LDS operations of a wave
// complete in order); every poll is bounded and ends in the `bad` exit, never in a hang.
struct DiagV5Shared {
    double LT[28 * TILE];  // strictly-lower tiles, [tile][c][row]: slot (b+1,b) first carries the staged (not yet solved) tile
    double IT[NT * TILE];  // leaf inverses transposed
    double Dt[2][TILE];    // by parity of b: tile (b,b) staged by its owner, then (same buffer) in leaf layout [col][row]
    double Lb[2][136];     // by parity of b: factored leaf, lower triangle packed by rows (stored to global memory by an update wave)
    int f_x[NT];           // tiles (b,b-1) and (b,b), updated with the panels 0..b-2, are staged in LT / Dt[b & 1]
    int f_leaf[NT];        // leaf b done: IT[b] valid
    int f_lbread[NT];      // Lb[b & 1] (leaf b, valid from f_leaf[b] on) has been read by its storing wave
    int f_p[NT][NT];       // [i][b]: panel tile L(i,b) is final in LT
    int done[NT];          // update waves whose global stores of panel b have drained
    int bad;
};
static_assert(sizeof(DiagV5Shared) <= 80 * 1024 - 256, "two workgroups per CU must fit into 160 KiB of LDS");

__host__ __device__ constexpr int owner5(int i) { return 1 + i % 3; }
__host__ __device__ constexpr int slot5(int i, int j) {
    int s = 0;
    for (int r = i % 3; r < i; r += 3) s += r + 1;
    return s + j;
}
constexpr int NSLOT5 = 15;  // rows 1, 4, 7

__device__ __forceinline__ bool spin_flag(const int *f, DiagV5Shared &sh) {
    const volatile int *vf = f;
    unsigned n = 0;
    while (*vf == 0) {
        if (++n > (1u << 22)) {  // ~0.5 s: a protocol bug must end as an error, not as a hung GPU
            sh.bad = 0x7ffffff0;
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ void set_flag(int *f) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's LDS writes issued before the flag have landed
    if ((threadIdx.x & 63) == 0) *(volatile int *)f = 1;
}

__device__ __forceinline__ void diag_v5_load(const double *__restrict__ A, int64_t lda, v4d (&acc)[NSLOT5]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            if (owner5(i) == wave) {
                v4d v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = *(const gf64 *)&A[(16 * i + l15) + (int64_t)(16 * j + l4 + 4 * r) * lda];
                acc[slot5(i, j)] = v;
            }
        }
}

// ---- the three pieces of v5 as inlinable helpers: the persistent kernel (chol_mega.hip) wraps the two wave roles into separate
// noinline functions so that each role gets its own register allocation (the leaf wave needs ~60 VGPRs, the update waves hold
// 15 tiles); the stand-alone kernel and tools/diagbench use diag_v5_core below.
__device__ __forceinline__ void diag5_init_flags(DiagV5Shared &sh) {  // threads 0..63 (one wave) zero all hand-off words
    const int tid = threadIdx.x;
    if (tid < NT) {
        sh.f_x[tid] = 0;
        sh.f_leaf[tid] = 0;
        sh.f_lbread[tid] = 0;
        sh.done[tid] = 0;
    }
    if (tid < NT * NT) (&sh.f_p[0][0])[tid] = 0;
    if (tid == 0) sh.bad = 0;
}

// update waves, before the second start-up barrier: tile (0,0) to the leaf wave, tiles (1,0), (1,1) staged
__device__ __forceinline__ void diag5_stage_first(DiagV5Shared &sh, const v4d (&acc)[NSLOT5]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    if (wave == owner5(0)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sh.Dt[0][(l4 + 4 * r) * 16 + l15] = acc[slot5(0, 0)][r];
    }
    if (wave == owner5(1)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sh.LT[MRBF_SIDX(1, 0) * TILE + (l4 + 4 * r) * 16 + l15] = acc[slot5(1, 0)][r];
            sh.Dt[1][(l4 + 4 * r) * 16 + l15] = acc[slot5(1, 1)][r];
        }
        set_flag(&sh.f_x[1]);
    }
}

__device__ __forceinline__ void diag5_leaf_loop(DiagV5Shared &sh, unsigned long long *ts) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    // ---------------------------------------------------------------- leaf wave: the whole critical chain
    // Every LDS round trip costs this wave ~250 cycles, so a step has exactly two: (1) leaf inverse / factored leaf out, the
    // solve's operands and the staged tiles in; (2) the finished diagonal tile out (accumulator layout) and in (leaf layout).
    double a[16];
    {
        const double *dt = sh.Dt[0];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = (lane < 16) ? ((c <= lane) ? dt[c * 16 + lane] : 0.0) : ((c == lane - 16) ? 1.0 : 0.0);
    }
    int lb_free = 1;  // Lb[b & 1] may be overwritten (its previous leaf, b - 2, has been picked up)
#pragma unroll 1
    for (int b = 0; b < NT; ++b) {
        if (ts && lane == 0) ts[4 * b + 0] = __builtin_readcyclecounter();
        int bad = 0;
        leaf_step<0>(a, bad, 16 * b);
        leaf_step<1>(a, bad, 16 * b);
        leaf_step<2>(a, bad, 16 * b);
        leaf_step<3>(a, bad, 16 * b);
        leaf_step<4>(a, bad, 16 * b);
        leaf_step<5>(a, bad, 16 * b);
        leaf_step<6>(a, bad, 16 * b);
        leaf_step<7>(a, bad, 16 * b);
        leaf_step<8>(a, bad, 16 * b);
        leaf_step<9>(a, bad, 16 * b);
        leaf_step<10>(a, bad, 16 * b);
        leaf_step<11>(a, bad, 16 * b);
        leaf_step<12>(a, bad, 16 * b);
        leaf_step<13>(a, bad, 16 * b);
        leaf_step<14>(a, bad, 16 * b);
        leaf_step<15>(a, bad, 16 * b);
        if (bad && lane == 0) sh.bad = bad;
        if (ts && lane == 0) ts[4 * b + 1] = __builtin_readcyclecounter();
        if (!lb_free && !spin_flag(&sh.f_lbread[b - 2], sh)) break;
        // ---- round trip 1: writes ...
        if (lane >= 16 && lane < 32) {
            const int r = lane - 16;  // lane 16 + r holds (L^-T)[r][c] = inv[c][r]:  IT[b][a2 = r][a = c]
#pragma unroll
            for (int c = 0; c < 16; ++c) sh.IT[b * TILE + r * 16 + c] = (c >= r) ? a[c] : 0.0;
        } else if (lane < 16) {
            // the factored leaf for the storing wave: row `lane`, columns 0..lane, packed
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c <= lane) sh.Lb[b & 1][lane * (lane + 1) / 2 + c] = a[c];
        }
        // ... and reads (LDS operations of a wave complete in order: the operand reads see the IT just written)
        const bool more = b + 1 < NT;
        double ia[4];
        v4d m, c;
        int fx = 1;
        double *xt = &sh.LT[MRBF_SIDX(more ? b + 1 : 1, more ? b : 0) * TILE];
        double *dn = sh.Dt[(b + 1) & 1];
        if (more) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ia[s] = opnd(&sh.IT[b * TILE], s, l15, l4);
            fx = *(volatile int *)&sh.f_x[b + 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                m[r] = xt[(l4 + 4 * r) * 16 + l15];
                c[r] = dn[(l4 + 4 * r) * 16 + l15];
            }
            if (b >= 1) lb_free = *(volatile int *)&sh.f_lbread[b - 1];
        }
        set_flag(&sh.f_leaf[b]);  // (one wait for everything above)
        if (__builtin_amdgcn_readfirstlane(bad)) break;
        if (!more) break;
        if (!fx) {  // the staged tiles were not there yet (rare): wait, read again
            if (!spin_flag(&sh.f_x[b + 1], sh)) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                m[r] = xt[(l4 + 4 * r) * 16 + l15];
                c[r] = dn[(l4 + 4 * r) * 16 + l15];
            }
        }
        if (ts && lane == 0) ts[4 * b + 2] = __builtin_readcyclecounter();
        // panel solve of tile (b+1,b) and the last update of tile (b+1,b+1), all inside this wave
        v4d p = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[s], m[s], p, 0, 0, 0);
        // A operand (row l15, k = 4s + l4) and B operand (k = 4s + l4, col l15) of P P' are both register s of p
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-p[s], p[s], c, 0, 0, 0);
        // ---- round trip 2: the finished tile out in accumulator layout, in again in leaf layout (same addressing [col][row])
#pragma unroll
        for (int r = 0; r < 4; ++r) dn[(l4 + 4 * r) * 16 + l15] = c[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) xt[(l4 + 4 * r) * 16 + l15] = p[r];  // final L(b+1,b) for everybody else
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) a[cc] = (lane < 16) ? ((cc <= lane) ? dn[cc * 16 + lane] : 0.0) : ((cc == lane - 16) ? 1.0 : 0.0);
        set_flag(&sh.f_p[b + 1][b]);  // (the wait the next leaf needs anyway)
        if (ts && lane == 0) ts[4 * b + 3] = __builtin_readcyclecounter();
    }
}

template <bool SC1, bool STREAM>
__device__ __forceinline__ void diag5_update_loop(double *__restrict__ A, int64_t lda, DiagV5Shared &sh, v4d (&acc)[NSLOT5],
                                                  double *__restrict__ itg, unsigned *prog, unsigned long long *ts) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // ---------------------------------------------------------------- update waves
#pragma unroll 1
    for (int b = 0; b < NT; ++b) {
        if (!spin_flag(&sh.f_leaf[b], sh)) break;
        if (*(volatile int *)&sh.bad) break;
        const double *itb = &sh.IT[b * TILE];
        double ia[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ia[s] = opnd(itb, s, l15, l4);
        // (B1) urgent: row b + 2 (if owned).  Its panel solve, then -- as soon as the leaf wave has published L(b+1,b) -- the
        //      updates of its two leading tiles, which are then complete through panel b and are staged for the leaf wave.
        //      No global store is issued here: a store in flight would make the drain below wait a full write-through.
#pragma unroll
        for (int i = 2; i < NT; ++i) {
            if (i == b + 2 && owner5(i) == wave) {
#pragma unroll
                for (int j = 0; j < i - 1; ++j) {
                    if (j == b) {
                        v4d p = {0.0, 0.0, 0.0, 0.0};
                        const v4d m = acc[slot5(i, j)];
#pragma unroll
                        for (int s = 0; s < 4; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[s], m[s], p, 0, 0, 0);
                        acc[slot5(i, j)] = p;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.LT[MRBF_SIDX(i, j) * TILE + (l4 + 4 * r) * 16 + l15] = p[r];
                    }
                }
                set_flag(&sh.f_p[i][b]);
                if (!spin_flag(&sh.f_p[i - 1][b], sh)) break;
                const double *ti_ = &sh.LT[(i * (i - 1) / 2 + b) * TILE];        // L(i,b)
                const double *tj_ = &sh.LT[((i - 1) * (i - 2) / 2 + b) * TILE];  // L(i-1,b)
                v4d c1 = acc[slot5(i, i - 1)], c2 = acc[slot5(i, i)];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double oi = opnd(ti_, s, l15, l4);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(tj_, s, l15, l4), oi, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-oi, oi, c2, 0, 0, 0);
                }
                acc[slot5(i, i - 1)] = c1;
                acc[slot5(i, i)] = c2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sh.LT[MRBF_SIDX(i, i - 1) * TILE + (l4 + 4 * r) * 16 + l15] = c1[r];
                    sh.Dt[i & 1][(l4 + 4 * r) * 16 + l15] = c2[r];
                }
                set_flag(&sh.f_x[i]);
            }
        }
        // (E) panel b - 1 is published once all three update waves have drained their stores of it (issued a leaf ago)
        if (STREAM && b >= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const int n = __hip_atomic_fetch_add(&sh.done[b - 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (n == 2)
                    __hip_atomic_store((__attribute__((address_space(1))) unsigned *)prog, (unsigned)b, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // (B2) the other panel solves of the owned rows, nearest row first, and the global stores of all of them
#pragma unroll
        for (int i = 2; i < NT; ++i) {
            if (i > b + 1 && owner5(i) == wave) {
#pragma unroll
                for (int j = 0; j < i - 1; ++j) {
                    if (j == b) {
                        v4d p = acc[slot5(i, j)];
                        if (i != b + 2) {
                            const v4d m = p;
                            p = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int s = 0; s < 4; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[s], m[s], p, 0, 0, 0);
                            acc[slot5(i, j)] = p;
#pragma unroll
                            for (int r = 0; r < 4; ++r) sh.LT[MRBF_SIDX(i, j) * TILE + (l4 + 4 * r) * 16 + l15] = p[r];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) gstore<SC1>(&A[(16 * i + l15) + (int64_t)(16 * j + l4 + 4 * r) * lda], p[r]);
                    }
                }
                if (i != b + 2) set_flag(&sh.f_p[i][b]);
            }
        }
        // (C) trailing update with panel b of the owned tiles (i,j), b < j <= i, i >= b + 3, nearest row first.  All panel
        //     tiles of column b are awaited once, up front: the updates below are then free of waits and the compiler can
        //     overlap one tile's operand loads with another tile's MFMAs.
        {
            bool okf = true;
#pragma unroll
            for (int j = 1; j < NT; ++j)
                if (j > b && okf) okf = spin_flag(&sh.f_p[j][b], sh);
            if (!okf) break;
        }
#pragma unroll
        for (int i = 3; i < NT; ++i) {
            if (i > b + 2 && owner5(i) == wave) {
#pragma unroll
                for (int j = 1; j <= i; ++j) {
                    if (j > b) {
                        const double *tj_ = &sh.LT[(j * (j - 1) / 2 + b) * TILE];  // L(j,b)   (j == i: the tile's own row)
                        const double *ti_ = &sh.LT[(i * (i - 1) / 2 + b) * TILE];  // L(i,b)
                        v4d c = acc[slot5(i, j)];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            c = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(tj_, s, l15, l4), opnd(ti_, s, l15, l4), c, 0, 0, 0);
                        acc[slot5(i, j)] = c;
                    }
                }
            }
        }
        // (D) global stores that are nobody's critical path: tile L(b+1,b) (solved by the leaf wave) by its row's owner, the
        //     factored leaf b and (streamed mode) its inverse by the owner of row b
        if (b + 1 < NT && wave == owner5(b + 1)) {
            if (!spin_flag(&sh.f_p[b + 1][b], sh)) break;
            const double *xt = &sh.LT[MRBF_SIDX(b + 1, b) * TILE];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gstore<SC1>(&A[(16 * (b + 1) + l15) + (int64_t)(16 * b + l4 + 4 * r) * lda], xt[(l4 + 4 * r) * 16 + l15]);
        }
        if (wave == owner5(b)) {
            double lv[4], iv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = l4 + 4 * r;
                lv[r] = (c <= l15) ? sh.Lb[b & 1][l15 * (l15 + 1) / 2 + c] : 0.0;
                iv[r] = sh.IT[b * TILE + (l4 + 4 * r) * 16 + l15];
            }
            set_flag(&sh.f_lbread[b]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = l4 + 4 * r;
                if (c <= l15) gstore<SC1>(&A[(16 * b + l15) + (int64_t)(16 * b + c) * lda], lv[r]);
                if (STREAM) gstore<true>(&itg[b * TILE + (l4 + 4 * r) * 16 + l15], iv[r]);
            }
        }
    }
}

// X = L^-1 assembled block column by block column (operands from LT / IT in LDS; independent of the register ownership)
template <bool SC1>
__device__ __forceinline__ void diag5_inverse(double *__restrict__ Linv, DiagV5Shared &sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = half == 0 ? wave : 7 - wave;
        v4d X[NT];
#pragma unroll
        for (int r = 0; r < 4; ++r) X[0][r] = sh.IT[j * TILE + l15 * 16 + l4 + 4 * r];
#pragma unroll
        for (int r = 0; r < 4; ++r) gstore<SC1>(&Linv[(16 * j + l4 + 4 * r) + (16 * j + l15) * DNB], X[0][r]);
#pragma unroll
        for (int di = 1; di < NT; ++di) {
            const int i = j + di;
            if (i < NT) {
                v4d S = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int dp = 0; dp < di; ++dp) {
                    const int p = j + dp;
                    const double *lt = &sh.LT[MRBF_SIDX(i, p) * TILE];
#pragma unroll
                    for (int s = 0; s < 4; ++s) S = __builtin_amdgcn_mfma_f64_16x16x4f64(opnd(lt, s, l15, l4), X[dp][s], S, 0, 0, 0);
                }
                v4d Xi = {0.0, 0.0, 0.0, 0.0};
                const double *iti = &sh.IT[i * TILE];
#pragma unroll
                for (int s = 0; s < 4; ++s) Xi = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(iti, s, l15, l4), S[s], Xi, 0, 0, 0);
                X[di] = Xi;
#pragma unroll
                for (int r = 0; r < 4; ++r) gstore<SC1>(&Linv[(16 * i + l4 + 4 * r) + (16 * j + l15) * DNB], Xi[r]);
            }
        }
        for (int e = lane; e < 16 * j * 16; e += 64) {
            const int row = e % (16 * j), col = e / (16 * j);
            gstore<SC1>(&Linv[row + (16 * j + col) * DNB], 0.0);
        }
    }
}

// same contract as diag_v4_core
template <bool SC1, bool STREAM, bool PRELOADED>
__device__ __forceinline__ int diag_v5_core(double *__restrict__ A, int64_t lda, double *__restrict__ Linv, DiagV5Shared &sh,
                                            v4d (&acc)[NSLOT5], double *__restrict__ itg, unsigned *prog,
                                            unsigned long long *ts = nullptr /* tools/diagbench: 4 cycle stamps per step */) {
    const int tid = threadIdx.x, wave = tid >> 6;
    if (wave == 0) diag5_init_flags(sh);
    if (!PRELOADED) diag_v5_load(A, lda, acc);
    __syncthreads();  // flags are zero before anybody sets one
    diag5_stage_first(sh, acc);
    __syncthreads();
    if (wave == 0)
        diag5_leaf_loop(sh, ts);
    else
        diag5_update_loop<SC1, STREAM>(A, lda, sh, acc, itg, prog, ts);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int bad_all = sh.bad;
    if (bad_all) return bad_all;
    if (STREAM && tid == 64)
        __hip_atomic_store((__attribute__((address_space(1))) unsigned *)prog, (unsigned)NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    diag5_inverse<SC1>(Linv, sh);
    return 0;
}

}  // namespace diagcore
}  // namespace mrbf
