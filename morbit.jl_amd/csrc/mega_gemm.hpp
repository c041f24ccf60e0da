// The operand pipeline of the persistent factorisation's GEMM jobs (chol_mega.hip), in a header of its own so that tools/gemmtest can
// run it alone against a plain reference.
#pragma once
#include "common.hpp"

namespace mrbf {
namespace mega {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2d gv2d;
constexpr int NB = 128, BK = 16, LDS_LD = NB + 16;  // (2*LD) % 64 == 32: k and k+1 rows of a chunk hit disjoint banks

// acc(j-part, i-part) += sum_k Bp(j, k) * Ap(i, k) over K columns (multiple of 8).  Ag / Bg point at row 0 of the TM-row / 128-row
// operand tiles, column 0 of the range (16-byte aligned, even leading dimensions).
// TM = 128: wave (w >> 1, w & 1) owns a 64 x 64 quadrant; TM = 64: wave w owns all 64 rows of columns 32w .. 32w+31.
//
// Round 4: operands go global -> LDS directly (LDS-DMA, global_load_lds_dwordx4: one instruction = one 128-row column = 1 KB, lane l
// brings rows 2l, 2l+1) into a ring of four 8-column stages, three stages in flight; ONE raw barrier per stage in the MIDDLE of the
// stage's MFMAs (the second k-step's fragments are in registers before it, the next stage's first fragments are read behind it under
// those MFMAs), counted vmcnt waits.  The round-1 loop (16-column chunks through registers, one chunk in flight, two barriers per
// chunk with the LDS write between them) left the matrix pipe idle ~30 % of the time when a workgroup had its compute unit to itself
// -- chain jobs always do, bulk jobs whenever the partner waits or stores: 115 us against 82 us of MFMA time for K = 768 (job log r04).
// Ordering rules (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"): a stage is read only
// behind [own vmcnt wait -> barrier]; a stage buffer is refilled only behind [lgkmcnt(0) -> barrier] of every reader.
#ifndef MRBF_GEMM_NSTG
#define MRBF_GEMM_NSTG 4
#endif
// The workgroup barrier of the operand ring.  `__builtin_amdgcn_s_barrier()` alone is IntrNoMem for the compiler: a plain LDS load that
// follows it in the source may be moved in front of it (it is ordered with the volatile asm waits, not with memory).  The asm form with
// a memory clobber is a barrier for the compiler as well as for the waves (round 5, profiles/r05_ldsdma_hazard.txt).
#ifndef MRBF_GEMM_RAW_BARRIER
#define MRBF_RING_BARRIER() asm volatile("s_barrier" ::: "memory")
#else
#define MRBF_RING_BARRIER() __builtin_amdgcn_s_barrier()
#endif
constexpr int SK = 8, NSTG = MRBF_GEMM_NSTG;  // columns per stage, stages in the ring
constexpr int STG = 2 * SK * LDS_LD;     // doubles per stage: A columns [k][LDS_LD], then B columns
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__device__ __forceinline__ void glds16(const double *g, double *l) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 16, 0, 0);
}
// at most 4 n of this wave's LDS-DMA loads may still be in flight
__device__ __forceinline__ void wait_stages(int n) {
    if (n >= 3)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int TM>
__device__ __forceinline__ void gemm_acc_v2(const double *__restrict__ Ag, int64_t lda, const double *__restrict__ Bg, int64_t ldb, int K,
                                         v4d (&acc)[TM / 32][4], double *smem) {
    constexpr int NJ = TM / 32;  // 16-wide j tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ioff = (TM == 128) ? (wave >> 1) * 64 : 0;
    const int joff = (TM == 128) ? (wave & 1) * 64 : wave * 32;
    const int nst = __builtin_amdgcn_readfirstlane(K / SK);
    // wave w stages the columns 2w, 2w+1 of both operands of every stage: four LDS-DMA instructions per wave and stage
    // A 64-row column is half a wave's worth: lanes 32 .. 63 fetch rows 0 .. 63 a second time (they land in rows 64 .. 127 of the
    // column's LDS slot, which nobody reads).  NOT `if (lane < 32)` around the instruction: with K and ldb known at compile time
    // (the panel solve: K = ldb = 128) the compiler unrolled the prologue, folded the identical B-column DMAs of the two sides of that
    // divergent branch into one instruction stream and ended up with ONE wave-wide DMA whose LDS base differed between the halves of the
    // wave -- it put the base through v_readfirstlane into M0, so the upper lanes' rows 64 .. 127 of the B columns went to the lower
    // lanes' address (round 5: profiles/r05_ldsdma_hazard.txt; what round 4 had put down to a stale read).  The LDS base of an LDS-DMA
    // must be wave-uniform: keep every LDS-DMA outside divergent control flow.
    const double *ap = Ag + (int64_t)(2 * wave) * lda + 2 * ((TM == 128) ? lane : (lane & 31));
    const double *bp = Bg + (int64_t)(2 * wave) * ldb + 2 * lane;
    constexpr bool a_on = true;
    auto issue = [&](int st) {
        double *sA = smem + (st % NSTG) * STG + 2 * wave * LDS_LD;
        double *sB = sA + SK * LDS_LD;
        const double *ga = ap + (int64_t)st * SK * lda;
        const double *gb = bp + (int64_t)st * SK * ldb;
        if (a_on) glds16(ga, sA);
        if (a_on) glds16(ga + lda, sA + LDS_LD);
        glds16(gb, sB);
        glds16(gb + ldb, sB + LDS_LD);
    };
    auto frag = [&](int st, int kk, double (&av)[4], double (&bv)[NJ]) {
        const double *sA = smem + (st % NSTG) * STG + (kk * 4 + l4) * LDS_LD;
        const double *sB = sA + SK * LDS_LD;
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = sA[ioff + i * 16 + l15];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = sB[joff + j * 16 + l15];
    };
    // D[row = j][col = i]: the lane index (l & 15) runs along i, contiguous in the column-major tile
    auto mma = [&](const double (&av)[4], const double (&bv)[NJ], int j0, int j1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (j >= j0 && j < j1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[j], av[i], acc[j][i], 0, 0, 0);
            }
    };
    const int npro = nst < NSTG ? nst : NSTG;
    for (int st = 0; st < npro; ++st) issue(st);
    wait_stages(npro - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MRBF_RING_BARRIER();
    double xa[4], xb[NJ], ya[4], yb[NJ];
    frag(0, 0, xa, xb);
    int st = 0;
    // steady state: stages st + 1 .. st + NSTG - 1 are in flight when stage st + 1 is awaited (the newer ones may stay out), stage st + NSTG is issued into
    // the buffer of stage st behind the barrier.  The issue order is pinned (sched_barrier): every fragment read sits behind the first
    // four MFMAs of the block in front of its consumer -- at that point nothing newer is outstanding on the LDS queue, so the
    // compiler's wait for those four MFMAs' own operands (it writes lgkmcnt(0), not a partial count) costs nothing, and the read has
    // twelve MFMAs (~770 cycles) to land; the barrier sits behind a full block of queued MFMAs.
#define MRBF_GEMM_HALF(FA, FB, NEXT_READ)                \
    mma(FA, FB, 0, 1);                                   \
    __builtin_amdgcn_sched_barrier(0);                   \
    NEXT_READ;                                           \
    __builtin_amdgcn_sched_barrier(0);                   \
    mma(FA, FB, 1, NJ);                                  \
    __builtin_amdgcn_sched_barrier(0)
#pragma unroll 1
    for (; st + NSTG < nst; ++st) {
        MRBF_GEMM_HALF(xa, xb, frag(st, 1, ya, yb));
        if constexpr (NSTG == 4)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // stage st + 1 of this wave has landed (the two behind it may stay out)
        else
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of stage st are in
        MRBF_RING_BARRIER();
        issue(st + NSTG);
        __builtin_amdgcn_sched_barrier(0);
        MRBF_GEMM_HALF(ya, yb, frag(st + 1, 0, xa, xb));
    }
#pragma unroll 1
    for (; st + 1 < nst; ++st) {  // the last stages: nothing left to issue
        MRBF_GEMM_HALF(xa, xb, frag(st, 1, ya, yb));
        wait_stages(nst - 2 - st);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MRBF_RING_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        MRBF_GEMM_HALF(ya, yb, frag(st + 1, 0, xa, xb));
    }
    MRBF_GEMM_HALF(xa, xb, frag(st, 1, ya, yb));
    mma(ya, yb, 0, NJ);
#undef MRBF_GEMM_HALF
    __syncthreads();  // the LDS stage buffers are free again (the caller may overlay them)
}

// The round-1 loop, kept for the panel jobs' triangular solve (below, gemm_acc_v1): operands through registers.
// acc(j-part, i-part) += sum_k Bp(j, k) * Ap(i, k) over K columns (multiple of 16): the LDS-tiled MFMA loop of
// chol_update_kernel<TM, *> (16-column chunks staged global -> registers -> LDS, next chunk's loads in flight under
// the current chunk's MFMAs).  Ag / Bg point at row 0 of the TM-row / 128-row operand tiles, column 0 of the range.
// TM = 128: wave (w >> 1, w & 1) owns a 64 x 64 quadrant; TM = 64: wave w owns all 64 rows of columns 32w .. 32w+31.
template <int TM>
__device__ __forceinline__ void gemm_acc_v1(const double *__restrict__ Ag, int64_t lda, const double *__restrict__ Bg, int64_t ldb, int K,
                                         v4d (&acc)[TM / 32][4], double *smem) {
    constexpr int NJ = TM / 32;              // 16-wide j tiles per wave
    constexpr int AL = TM / 32;              // v2d loads per thread per A chunk
    constexpr int AKS = (TM == 128) ? 4 : 8;  // k stride between a thread's A loads
    // global -> register prefetch depth in chunks.  The 128-row loop is MFMA-bound with one chunk in flight (and has no
    // registers to spare); the 64-row loop has half the MFMAs per chunk and was bound by the ~2 us load latency.
    constexpr int PF = (TM == 128) ? 1 : 2;
    double *As = smem;
    double *Bs = smem + BK * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ioff = (TM == 128) ? (wave >> 1) * 64 : 0;
    const int joff = (TM == 128) ? (wave & 1) * 64 : wave * 32;
    const int a_i2 = (TM == 128) ? (tid & 63) * 2 : (tid & 31) * 2;
    const int a_k0 = (TM == 128) ? (tid >> 6) : (tid >> 5);
    const int b_i2 = (tid & 63) * 2, b_k0 = tid >> 6;
    const double *Ap = Ag + a_i2 + (int64_t)a_k0 * lda;
    const double *Bp = Bg + b_i2 + (int64_t)b_k0 * ldb;
    v2d ra[PF][AL], rb[PF][4];
    const int nkc = K / BK;
#pragma unroll
    for (int f = 0; f < PF; ++f) {
        if (f < nkc) {
            const int64_t ko = (int64_t)f * BK;
#pragma unroll
            for (int u = 0; u < AL; ++u) ra[f][u] = *(const gv2d *)(Ap + (ko + AKS * u) * lda);
#pragma unroll
            for (int u = 0; u < 4; ++u) rb[f][u] = *(const gv2d *)(Bp + (ko + 4 * u) * ldb);
        }
    }
#pragma unroll 1
    for (int kc0 = 0; kc0 < nkc; kc0 += PF) {
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const int kc = kc0 + f;
            if (kc < nkc) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < AL; ++u) *(v2d *)&As[(a_k0 + AKS * u) * LDS_LD + a_i2] = ra[f][u];
#pragma unroll
                for (int u = 0; u < 4; ++u) *(v2d *)&Bs[(b_k0 + 4 * u) * LDS_LD + b_i2] = rb[f][u];
                __syncthreads();
                if (kc + PF < nkc) {
                    const int64_t ko = (int64_t)(kc + PF) * BK;
#pragma unroll
                    for (int u = 0; u < AL; ++u) ra[f][u] = *(const gv2d *)(Ap + (ko + AKS * u) * lda);
#pragma unroll
                    for (int u = 0; u < 4; ++u) rb[f][u] = *(const gv2d *)(Bp + (ko + 4 * u) * ldb);
                }
#pragma unroll
                for (int kk = 0; kk < BK / 4; ++kk) {
                    double av[4], bv[NJ];
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[i] = As[(kk * 4 + l4) * LDS_LD + ioff + i * 16 + l15];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bv[j] = Bs[(kk * 4 + l4) * LDS_LD + joff + j * 16 + l15];
                    // D[row = j][col = i]: the lane index (l & 15) runs along i, contiguous in the column-major tile
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[j], av[i], acc[j][i], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // the LDS chunk buffers are free again (the caller may overlay them)
}


template <int TM>
__device__ __forceinline__ void gemm_acc(const double *__restrict__ Ag, int64_t lda, const double *__restrict__ Bg, int64_t ldb, int K,
                                         v4d (&acc)[TM / 32][4], double *smem) {
    gemm_acc_v2<TM>(Ag, lda, Bg, ldb, K, acc, smem);
}

}  // namespace mega
}  // namespace mrbf
