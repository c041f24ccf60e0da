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
constexpr int ITLD = 17;   // row stride of a leaf inverse in LDS: the leaf wave writes rows, the panel waves read columns, both conflict-free
constexpr int ITS = 16 * ITLD;
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
    double IT[NT * ITS];   // leaf inverses transposed: IT[b][a2 * ITLD + a] = inv(L_bb)[a][a2]
    double Dt[2 * TILE];   // diagonal tile handed to the leaf wave, [col a][32 rows]: rows 0..15 the tile, rows 16..31 the identity
                           // (written once): the leaf wave's 32 lanes load their rows of [A_bb ; I] with one unconditional read per column
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
    // The pivot test stays off the pivot -> rsqrt -> scale chain: a non-positive pivot just lets NaNs through this block (every
    // later test then fails too, the first failure is the one recorded) and the caller aborts on the flag.
    const double piv = readlane_f64(a[K], K);
    const double rinv = fast_rsqrt_v4(piv);
    bad = (!(piv > 0.0) && bad == 0) ? col0 + K + 1 : bad;
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
                                            v4d (&acc)[NSLOT], double *__restrict__ itg, unsigned *prog, int dbg = 0,
                                            unsigned long long *segout = nullptr) {
    // the wave index is uniform: as a scalar the role branches below are s_cbranch instead of exec-mask juggling
    const int tid = threadIdx.x, lane_ = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) sh.bad = 0;
    // timing experiments (dbg & 4): cycles per segment of one wave -- wave 0 (segments: pre-leaf, leaf, post-leaf, barrier X,
    // look-ahead, barrier Y) or, with dbg & 8, wave (dbg >> 4) & 3 (rest of the trailing update, store drain, barrier X, leaf /
    // inverse stores + operand fetch, panel solve, barrier Y, panel stores + next column's update)
    unsigned long long tseg[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const int tw = (dbg & 8) ? (dbg >> 4) & 3 : 0;
#define MRBF_DSEG(k)                                         \
    do {                                                     \
        if ((dbg & 4) && wave == tw) {                       \
            const unsigned long long now_ = __builtin_readcyclecounter(); \
            tseg[k] += now_ - tlast;                         \
            tlast = now_;                                    \
        }                                                    \
    } while (0)
    if (!PRELOADED) diag_v4_load(A, lda, acc);
    if (tid < 256) sh.Dt[(tid >> 4) * 32 + 16 + (tid & 15)] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
    __syncthreads();
    if (dbg & 4) tlast = __builtin_readcyclecounter();

#pragma unroll 1
    for (int b = 0; b < NT; ++b) {
        // per-iteration opaque copy of the lane index: everything derived from it (LDS addresses of 28 tiles, lane masks, the
        // identity rows of the leaf) is otherwise hoisted out of this loop, spilled, and reloaded behind s_waitcnt vmcnt(0) --
        // i.e. behind the write-through stores of the previous panel -- in the middle of the sequential chain (measured: the
        // wave that stores the leaf spent ~2800 cycles per panel there)
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4;
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
                    for (int r = 0; r < 4; ++r) sh.Dt[(l4 + 4 * r) * 32 + l15] = acc[i][r];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double a[16];
            int bad = 0;
            MRBF_DSEG(0);
            // (entries above the diagonal of A_bb are whatever the tile held there: they stay in their own lanes and are never
            //  broadcast -- the pivots and multipliers are read from lanes >= the column index only)
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = sh.Dt[c * 32 + (lane & 31)];
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
                for (int c = 0; c < 16; ++c) sh.IT[b * ITS + r * ITLD + c] = (c >= r) ? a[c] : 0.0;
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
            MRBF_DSEG(0);
            // panel b-1's write-through stores were issued a leaf ago: draining them here costs nothing
            if (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MRBF_DSEG(1);
        }
        if (wave == 0) MRBF_DSEG(2);
        __syncthreads();  // X: leaf inverse IT[b] visible
        if (wave == 0)
            MRBF_DSEG(3);
        else
            MRBF_DSEG(2);
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
            for (int r = 0; r < 4; ++r) gstore<true>(&itg[b * TILE + (l4 + 4 * r) * 16 + l15], sh.IT[b * ITS + (l4 + 4 * r) * ITLD + l15]);
        }
        if (wave != 0) {
            // ---- panel: P_i' = inv(L_bb) * A_ib'  for the owned tiles of block column b
            const double *itb = &sh.IT[b * ITS];
            double ia[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ia[s] = itb[(4 * s + l4) * ITLD + l15];
            if (dbg & 8) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MRBF_DSEG(3);
            }
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
        MRBF_DSEG(4);
        __syncthreads();  // Y: panel tiles of block column b in LT
        MRBF_DSEG(5);
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
            MRBF_DSEG(6);
        }
    }
    if (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int lane = lane_, l15 = lane & 15, l4 = lane >> 4;
    const int bad_all = sh.bad;
    if (bad_all) return bad_all;
    if ((dbg & 4) && lane_ == 0 && wave == tw) {
        for (int k = 0; k < (segout ? 7 : 6); ++k) {
            if (segout)
                segout[k] = tseg[k];
            else
                Linv[k] = (double)tseg[k];
        }
    }
    if (dbg & 1) return 0;  // timing experiments: no inverse
    if (STREAM && tid == 64) {
        __hip_atomic_store((__attribute__((address_space(1))) unsigned *)prog, (unsigned)NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((dbg & 4) && segout) segout[7] = wall_clock64();  // all eight panels published
    }

    // ---- inverse: wave w assembles block columns w and 7 - w of X = L^-1 in registers, MFMA C/D layout
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = half == 0 ? wave : 7 - wave;
        v4d X[NT];
        // X_jj = inv(L_jj): C layout element [(l4 + 4r)][l15] = IT[j][l15][l4 + 4r]
#pragma unroll
        for (int r = 0; r < 4; ++r) X[0][r] = sh.IT[j * ITS + l15 * ITLD + l4 + 4 * r];
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
                const double *iti = &sh.IT[i * ITS];
#pragma unroll
                for (int s = 0; s < 4; ++s) Xi = __builtin_amdgcn_mfma_f64_16x16x4f64(-iti[(4 * s + l4) * ITLD + l15], S[s], Xi, 0, 0, 0);
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
    if ((dbg & 4) && tid == 0 && tw == 0 && segout) segout[6] = __builtin_readcyclecounter() - tlast;  // the inverse, wave 0's share
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// v6: the same factorisation with ONE workgroup barrier per 16-column panel and the leaf wave on a short leash.
//
// In v4 the leaf wave owns all eight diagonal tiles and meets the other waves twice per panel (X: leaf inverse visible, Y: panel
// tiles visible): per panel it spends 1.8 us in the leaf and 2.8 us around it (its own look-ahead updates of the later diagonal
// tiles, the panel solve of the other waves it waits for at Y, the update of the next diagonal tile from LDS).  Here the leaf
// wave only ever holds three tiles -- D_b = (b,b), S_b = (b+1,b), D_{b+1} -- and computes what its next leaf needs itself:
//     leaf(D_b) | barrier M_b | take S_b, D_{b+1} over from their owners (through LDS; they carry the panels <= b-2), apply
//     panel b-1 to both, S_b <- inv(L_bb) S_b = L(b+1,b), D_{b+1} -= L(b+1,b) L(b+1,b)' from registers, next leaf.
// The update waves (1..3) own every other tile; between M_b and M_{b+1} they apply panel b-1 to their tiles (column b first),
// solve column b's panel tiles (i >= b+2) with the leaf inverse, hand (b+2,b+1) and (b+2,b+2) over, store.  Whatever the leaf
// wave needs from them is one panel old, so it never waits for this step's work: ~3.2 us per panel instead of 4.6.
// STREAM semantics (itg, *prog) as in v4; the leaf inverses stay in LDS for two panels only and the final inverse is assembled
// from itg (global), which makes room for the double-buffered hand-over tiles.  Mega kernel only (PRELOADED, STREAM).
__host__ __device__ constexpr bool v6_leafwave(int i, int j) { return i <= 1; }  // (0,0), (1,0), (1,1) start in the leaf wave
__host__ __device__ constexpr int v6_index(int i, int j) { return MRBF_TIDX(i, j) - 3; }  // the other 33 lower tiles, row-major
__host__ __device__ constexpr int v6_owner(int i, int j) { return v6_leafwave(i, j) ? 0 : 1 + v6_index(i, j) % 3; }
__host__ __device__ constexpr int v6_slot(int i, int j) { return v6_leafwave(i, j) ? MRBF_TIDX(i, j) : v6_index(i, j) / 3; }
constexpr int NSLOT6 = 11;

struct DiagV6Shared {
    double LT[28 * TILE];  // strictly-lower tiles: final L (operand order) -- or, before that, the raw tile handed to the leaf wave
    double IT[2 * ITS];    // leaf inverses of the current and the previous panel (padded rows)
    double Dt[2 * TILE];   // leaf hand-over: [col][32 rows], rows 16..31 the identity
    double Lb[2 * TILE];   // factored leaves (double-buffered: the storing wave reads one while the next is written)
    double DH[2 * TILE];   // diagonal tiles on their way to the leaf wave (double-buffered)
    int bad[2];            // failure word of leaf b in slot b & 1: the leaf wave may write leaf b + 1's word before a slower wave has
                           // read leaf b's behind M_b -- with one word that wave would leave the loop one barrier early
};

__device__ __forceinline__ void diag_v6_load(const double *__restrict__ A, int64_t lda, v4d (&acc)[NSLOT6]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            if (v6_owner(i, j) == wave) {
                v4d v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = *(const gf64 *)&A[(16 * i + l15) + (int64_t)(16 * j + l4 + 4 * r) * lda];
                acc[v6_slot(i, j)] = v;
            }
        }
}

// tile (i,j) -= L(i,p) L(j,p)' with both operands from LT
__device__ __forceinline__ void v6_update(v4d &c, const double *LT, int i, int j, int p, int l15, int l4) {
    const double *ti_ = &LT[MRBF_SIDX(i, p) * TILE];
    const double *tj_ = &LT[MRBF_SIDX(j, p) * TILE];
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd(tj_, s, l15, l4), opnd(ti_, s, l15, l4), c, 0, 0, 0);
}

__device__ __forceinline__ int diag_v6_core(double *__restrict__ A, int64_t lda, double *__restrict__ Linv, DiagV6Shared &sh,
                                            v4d (&acc)[NSLOT6], double *__restrict__ itg, unsigned *prog,
                                            unsigned long long *pubstamp = nullptr) {
    const int tid = threadIdx.x, lane_ = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // No barrier on entry: the leaf wave prepares what only it uses (failure word, identity rows) and starts the first leaf while the
    // other waves may still be folding the caller's last panel into their tiles; they meet at M_0.  (The caller's LDS use ended
    // behind a barrier.)
    if (wave == 0) {
        if (tid == 0) sh.bad[0] = sh.bad[1] = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = lane_ + 64 * u;
            sh.Dt[(e >> 4) * 32 + 16 + (e & 15)] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        }
    }
    v4d Dc = acc[0], Sc = acc[1], Dn = acc[2];  // leaf wave: (0,0), (1,0), (1,1)

#pragma unroll 1
    for (int b = 0; b < NT; ++b) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));  // nothing derived from the lane index is hoisted out of the loop (and spilled)
        const int l15 = lane & 15, l4 = lane >> 4;
        const int bb = b & 1;
        if (wave == 0) {
            // ---- leaf b
#pragma unroll
            for (int r = 0; r < 4; ++r) sh.Dt[(l4 + 4 * r) * 32 + l15] = Dc[r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double a[16];
            int bad = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = sh.Dt[c * 32 + (lane & 31)];
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
            if (bad && lane == 0) sh.bad[bb] = bad;
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) sh.Lb[bb * TILE + c * 16 + lane] = a[c];
            } else if (lane < 32) {
                const int r = lane - 16;
#pragma unroll
                for (int c = 0; c < 16; ++c) sh.IT[bb * ITS + r * ITLD + c] = (c >= r) ? a[c] : 0.0;
            }
        }
        // every wave: its write-through stores of the previous panel (L tiles, leaf, inverse) are out before the panel is published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // M_b
        if (sh.bad[bb]) break;
        if (b > 0 && tid == 64)
            __hip_atomic_store((__attribute__((address_space(1))) unsigned *)prog, (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {
            if (b < NT - 1) {
                if (b > 0) {
                    // take (b+1,b) and (b+1,b+1) over (they carry the panels <= b-2) and apply panel b-1
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Sc[r] = sh.LT[MRBF_SIDX(b + 1, b) * TILE + (l4 + 4 * r) * 16 + l15];
                        Dn[r] = sh.DH[((b - 1) & 1) * TILE + (l4 + 4 * r) * 16 + l15];
                    }
                    v6_update(Sc, sh.LT, b + 1, b, b - 1, l15, l4);
                    v6_update(Dn, sh.LT, b + 1, b + 1, b - 1, l15, l4);
                }
                // L(b+1,b) = tile * inv(L_bb)'  (transposed layout: P' = inv(L_bb) * tile')
                const double *itb = &sh.IT[bb * ITS];
                v4d p = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(itb[(4 * s + l4) * ITLD + l15], Sc[s], p, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sh.LT[MRBF_SIDX(b + 1, b) * TILE + (l4 + 4 * r) * 16 + l15] = p[r];
                    gstore<true>(&A[(16 * (b + 1) + l15) + (int64_t)(16 * b + l4 + 4 * r) * lda], p[r]);
                }
                // next diagonal tile -= L(b+1,b) L(b+1,b)': both operands are the registers just computed
#pragma unroll
                for (int s = 0; s < 4; ++s) Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(-p[s], p[s], Dn, 0, 0, 0);
                Dc = Dn;
            }
        } else {
            // ---- update waves, phase b
            if (wave == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = l4 + 4 * r;
                    if (c <= l15) gstore<true>(&A[(16 * b + l15) + (int64_t)(16 * b + c) * lda], sh.Lb[bb * TILE + c * 16 + l15]);
                }
            } else if (wave == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gstore<true>(&itg[b * TILE + (l4 + 4 * r) * 16 + l15], sh.IT[bb * ITS + (l4 + 4 * r) * ITLD + l15]);
            }
            // panel b-1 on this wave's tiles of block column b, then their panel solve
            const double *itb = &sh.IT[bb * ITS];
            double ia[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ia[s] = itb[(4 * s + l4) * ITLD + l15];
#pragma unroll
            for (int i = 2; i < NT; ++i)
#pragma unroll
                for (int j = 0; j <= i - 2; ++j) {
                    if (j == b && v6_owner(i, j) == wave) {
                        v4d m = acc[v6_slot(i, j)];
                        if (b > 0) v6_update(m, sh.LT, i, j, b - 1, l15, l4);
                        v4d p = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[s], m[s], p, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sh.LT[MRBF_SIDX(i, j) * TILE + (l4 + 4 * r) * 16 + l15] = p[r];
                            gstore<true>(&A[(16 * i + l15) + (int64_t)(16 * j + l4 + 4 * r) * lda], p[r]);
                        }
                    }
                }
            // panel b-1 on the two tiles the leaf wave takes over after the next barrier, and their hand-over
#pragma unroll
            for (int i = 2; i < NT; ++i) {
                if (i == b + 2) {
                    if (v6_owner(i, i - 1) == wave) {  // (b+2, b+1)
                        v4d m = acc[v6_slot(i, i - 1)];
                        if (b > 0) v6_update(m, sh.LT, i, i - 1, b - 1, l15, l4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.LT[MRBF_SIDX(i, i - 1) * TILE + (l4 + 4 * r) * 16 + l15] = m[r];
                    }
                    if (v6_owner(i, i) == wave) {  // (b+2, b+2)
                        v4d m = acc[v6_slot(i, i)];
                        if (b > 0) v6_update(m, sh.LT, i, i, b - 1, l15, l4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.DH[bb * TILE + (l4 + 4 * r) * 16 + l15] = m[r];
                    }
                }
            }
            // panel b-1 on everything else this wave still owns: tiles (i,j) with j >= b+1 that are not on their way to the leaf wave
            if (b > 0) {
#pragma unroll
                for (int i = 3; i < NT; ++i)
#pragma unroll
                    for (int j = 1; j <= i; ++j) {
                        if (j >= b + 1 && i >= b + 3 && v6_owner(i, j) == wave) v6_update(acc[v6_slot(i, j)], sh.LT, i, j, b - 1, l15, l4);
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int lane = lane_, l15 = lane & 15, l4 = lane >> 4;
    const int bad_all = sh.bad[0] ? (sh.bad[1] && sh.bad[1] < sh.bad[0] ? sh.bad[1] : sh.bad[0]) : sh.bad[1];
    if (bad_all) return bad_all;
    if (tid == 64) {
        __hip_atomic_store((__attribute__((address_space(1))) unsigned *)prog, (unsigned)NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pubstamp) *pubstamp = wall_clock64();  // all eight panels published (diagnostic launches)
    }
    // ---- inverse: as in v4, the leaf inverses read back from itg (this workgroup's own write-through stores, drained above)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = half == 0 ? wave : 7 - wave;
        v4d X[NT];
#pragma unroll
        for (int r = 0; r < 4; ++r) X[0][r] = *(const gf64 *)&itg[j * TILE + l15 * 16 + l4 + 4 * r];
#pragma unroll
        for (int r = 0; r < 4; ++r) gstore<true>(&Linv[(16 * j + l4 + 4 * r) + (16 * j + l15) * DNB], X[0][r]);
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
                const double *iti = itg + i * TILE;
#pragma unroll
                for (int s = 0; s < 4; ++s) Xi = __builtin_amdgcn_mfma_f64_16x16x4f64(-*(const gf64 *)&iti[(4 * s + l4) * 16 + l15], S[s], Xi, 0, 0, 0);
                X[di] = Xi;
#pragma unroll
                for (int r = 0; r < 4; ++r) gstore<true>(&Linv[(16 * i + l4 + 4 * r) + (16 * j + l15) * DNB], Xi[r]);
            }
        }
        for (int e = lane; e < 16 * j * 16; e += 64) {
            const int row = e % (16 * j), col = e / (16 * j);
            gstore<true>(&Linv[row + (16 * j + col) * DNB], 0.0);
        }
    }
    return 0;
}

}  // namespace diagcore
}  // namespace mrbf
