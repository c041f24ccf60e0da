// Latency of one column step of the 16 x 16 leaf factorisation, one wave, variants (round 2 experiment, NOT kept in the product).
// Idea: in the MFMA C/D layout column K of the leaf is register K >> 2 of lane group K & 3 -- exactly k-slice K & 3 of an MFMA
// operand -- so the rank-1 update of the whole leaf is one v_mfma_f64_16x16x4 with the other k-slices zero and nothing crosses
// lanes but the pivot.  Measured on MI355X (cycles per column): MFMA leaf without inverse 185, with inverse 261-319, round-1
// leaf (v_readlane broadcasts, inverse included) 191: a dependent f64 MFMA costs ~117 cycles and occupies the SIMD's f64 pipe,
// so the one-instruction update does not beat 15 readlane + fma pairs.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/opt/rocm/include -I../../morbit.jl_amd/csrc -I../../include -o leafbench leafbench.hip
#include "chol_diag_core.hpp"

#ifndef DPPV
#define DPPV 0
#endif
#include <cmath>
#include <cstdio>
#include <vector>
using namespace mrbf;
using diagcore::v4d;

// round-1 leaf: rows in lanes 0..15, identity rows in lanes 16..31, one register per column, v_readlane broadcasts
template <int K>
__device__ __forceinline__ void old_step(double (&a)[16]) {
    const double piv = diagcore::readlane_f64(a[K], K);
    const double rinv = diagcore::fast_rsqrt_v4(piv);
    a[K] *= rinv;
#pragma unroll
    for (int j = K + 1; j < 16; ++j) {
        const double ljk = diagcore::readlane_f64(a[K], j);
        a[j] = fma(-a[K], ljk, a[j]);
    }
}
__global__ __launch_bounds__(64) void kold(const double *A, double *out, unsigned long long *ts, int reps) {
    const int lane = threadIdx.x;
    double a0[16], acc = 0.0;
    for (int c = 0; c < 16; ++c) a0[c] = lane < 16 ? (c <= lane ? A[lane + 16 * c] : 0.0) : (c == lane - 16 ? 1.0 : 0.0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
        double a[16];
        for (int c = 0; c < 16; ++c) a[c] = a0[c];
        old_step<0>(a); old_step<1>(a); old_step<2>(a); old_step<3>(a); old_step<4>(a); old_step<5>(a); old_step<6>(a); old_step<7>(a);
        old_step<8>(a); old_step<9>(a); old_step<10>(a); old_step<11>(a); old_step<12>(a); old_step<13>(a); old_step<14>(a); old_step<15>(a);
        if (it == reps - 1)
            for (int c = 0; c < 16; ++c) out[64 + lane * 16 + c] = a[c];
        for (int c = 0; c < 16; ++c) acc += a[c];
        a0[0] += acc * 1e-300;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = acc;
    if (lane == 0) ts[5] = t1 - t0;
}

// hand-scheduled variant of the round-1 leaf: the updates column K-1 still owes to the columns K+1.. are spread over the latency
// gaps of column K's pivot -> rsqrt -> Newton -> scale chain (the compiler puts them all in one place)
template <int KP, int J>
__device__ __forceinline__ void upd(double (&a)[16]) {
    const double l = diagcore::readlane_f64(a[KP], J);
    a[J] = fma(-a[KP], l, a[J]);
}
template <int K, int G>
__device__ __forceinline__ void pend(double (&a)[16]) {
    constexpr int n = K >= 1 ? 15 - K : 0;
    constexpr int lo = G * n / 6, hi = (G + 1) * n / 6;
    if constexpr (hi > lo + 0) upd<(K >= 1 ? K - 1 : 0), K + 1 + lo>(a);
    if constexpr (hi > lo + 1) upd<(K >= 1 ? K - 1 : 0), K + 2 + lo>(a);
    if constexpr (hi > lo + 2) upd<(K >= 1 ? K - 1 : 0), K + 3 + lo>(a);
}
#define SB __builtin_amdgcn_sched_barrier(0)
template <int K>
__device__ __forceinline__ void sched_step(double (&a)[16]) {
    const double piv = diagcore::readlane_f64(a[K], K);
    double y0 = __builtin_amdgcn_rsq(piv);
    SB; pend<K, 0>(a); SB;
    const double hh = -0.5 * piv;
    SB; pend<K, 1>(a); SB;
    const double t = y0 * y0;
    SB; pend<K, 2>(a); SB;
    const double e = fma(hh, t, 1.5);
    SB; pend<K, 3>(a); SB;
    const double rinv = y0 * e;
    SB; pend<K, 4>(a); SB;
    a[K] *= rinv;
    SB; pend<K, 5>(a); SB;
    if constexpr (K < 15) upd<K, K + 1>(a);
    SB;
}
__global__ __launch_bounds__(64) void ksched(const double *A, double *out, unsigned long long *ts, int reps) {
    const int lane = threadIdx.x;
    double a0[16], acc = 0.0;
    for (int c = 0; c < 16; ++c) a0[c] = lane < 16 ? (c <= lane ? A[lane + 16 * c] : 0.0) : (c == lane - 16 ? 1.0 : 0.0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
        double a[16];
        for (int c = 0; c < 16; ++c) a[c] = a0[c];
        sched_step<0>(a); sched_step<1>(a); sched_step<2>(a); sched_step<3>(a); sched_step<4>(a); sched_step<5>(a); sched_step<6>(a); sched_step<7>(a);
        sched_step<8>(a); sched_step<9>(a); sched_step<10>(a); sched_step<11>(a); sched_step<12>(a); sched_step<13>(a); sched_step<14>(a); sched_step<15>(a);
        for (int c = 0; c < 16; ++c) acc += a[c];
        a0[0] += acc * 1e-300;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = acc;
    if (lane == 0) ts[6] = t1 - t0;
}

// Round 4: multipliers through LDS -- 197 -> 156 cycles per column HERE (one wave alone on its CU), but slower inside the factorisation
// (n = 256 / 2048 / 8192: +14 / +4.5 / +1.5 % with this leaf in diag_v6_core / diag_v4_core: the leaf wave's LDS reads queue behind the
// three update waves' operand traffic on the same LDS), so the product keeps the readlane leaf.  Two things learnt on the way:
// (i) lanes that talk to each other through LDS need wavefront-scope fences (no instructions) -- without them the compiler treats
// the scratch words as private to a lane and reuses, on the lanes that do not store, the value they loaded two columns ago (the
// identity rows, i.e. the inverse, came out 0.5 % wrong); (ii) the compiler evaluates the column updates lazily (each column's chain of
// fmas right in front of its pivot) whatever the source order, and pinning them eagerly (empty asm on the values + sched_barrier)
// is slower (174 cycles): the leaf is bound by its instruction count, not by the pivot chain.
// The leaf is bound by its instruction count (120 (j, K) pairs x [2 v_readlane + s_nop + fma]); only
// the next column's multiplier is needed at once (one readlane pair: the critical chain), the others are written to LDS once per
// column (lanes 0..15: ds_write_b64) and read back as broadcast ds_read_b128 (two multipliers per instruction), applied one column
// step later under the next column's pivot chain.
template <int K>
__device__ __forceinline__ void lds_step(double (&a)[16], double (&m)[2][16], double *col, int lane) {
    const double piv = diagcore::readlane_f64(a[K], K);
    const double rinv = diagcore::fast_rsqrt_v4(piv);
    a[K] *= rinv;
    if (lane < 16) col[(K & 1) * 16 + lane] = a[K];
    // lanes talk to each other through LDS here: without the fences the compiler may (and does) treat the location as private to the
    // lane and reuse the value this lane loaded two columns ago on the lanes that do not store
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (K < 15) {
        const double l = diagcore::readlane_f64(a[K], K + 1);
        a[K + 1] = fma(-a[K], l, a[K + 1]);
    }
#pragma unroll
    for (int j = K + 2; j < 16; ++j) m[K & 1][j] = col[(K & 1) * 16 + j];
    if constexpr (K >= 1) {
#pragma unroll
        for (int j = K + 1; j < 16; ++j) a[j] = fma(-a[K - 1], m[(K - 1) & 1][j], a[j]);
    }
    // (without these the compiler sinks every update of a column to the step that needs it: a chain of K dependent fmas in front of
    //  column K's pivot)
#pragma unroll
    for (int j = K + 1; j < 16; ++j) asm volatile("" : "+v"(a[j]));
    __builtin_amdgcn_sched_barrier(0);
}
__global__ __launch_bounds__(64) void klds(const double *A, double *out, unsigned long long *ts, int reps) {
    __shared__ __attribute__((aligned(16))) double col[32];
    const int lane = threadIdx.x;
    double a0[16], acc = 0.0;
    for (int c = 0; c < 16; ++c) a0[c] = lane < 16 ? (c <= lane ? A[lane + 16 * c] : 0.0) : (c == lane - 16 ? 1.0 : 0.0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
        double a[16], m[2][16];
        for (int c = 0; c < 16; ++c) a[c] = a0[c];
        lds_step<0>(a, m, col, lane); lds_step<1>(a, m, col, lane); lds_step<2>(a, m, col, lane); lds_step<3>(a, m, col, lane);
        lds_step<4>(a, m, col, lane); lds_step<5>(a, m, col, lane); lds_step<6>(a, m, col, lane); lds_step<7>(a, m, col, lane);
        lds_step<8>(a, m, col, lane); lds_step<9>(a, m, col, lane); lds_step<10>(a, m, col, lane); lds_step<11>(a, m, col, lane);
        lds_step<12>(a, m, col, lane); lds_step<13>(a, m, col, lane); lds_step<14>(a, m, col, lane); lds_step<15>(a, m, col, lane);
        if (it == reps - 1)
            for (int c = 0; c < 16; ++c) out[64 + lane * 16 + c] = a[c];
        for (int c = 0; c < 16; ++c) acc += a[c];
        a0[0] += acc * 1e-300;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = acc;
    if (lane == 0) ts[7] = t1 - t0;
}

// Round 4 (second try): multipliers by DPP.  v_fmac_f64 has a DPP form on gfx90a+ whose only control is row_newbcast:J (lane J of each
// row of 16 lanes to the whole row): with column K of the leaf duplicated into lanes 16..31 (v_permlane16_swap_b32, gfx950) the update
// of column J for the leaf rows AND the identity rows is ONE instruction, a[J] += (-m[row_newbcast:J]) * a[K], instead of two
// v_readlane + s_nop + fma; the pivot comes the same way (v_mov_b64_dpp row_newbcast:K).  DPP reads need two wait states after the
// VALU write of their source: the producers below end with s_nop 1 (inline asm: nothing inserts it for us).
template <int K, int V = DPPV>
__device__ __forceinline__ void dpp_step(double (&a)[16], int &bad, int col0) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned alo = (unsigned)__double2loint(a[K]), ahi = (unsigned)__double2hiint(a[K]);
    const u2 slo = __builtin_amdgcn_permlane16_swap(alo, alo, false, false);  // [0]: rows 1, 3 <- rows 0, 2 of the second operand
    const u2 shi = __builtin_amdgcn_permlane16_swap(ahi, ahi, false, false);
    const double x = __hiloint2double((int)shi[0], (int)slo[0]);  // column K of the leaf rows, in lanes 0..15 and again in lanes 16..31
    double piv;
    if constexpr (V == 0)
        asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(piv) : "v"(x), "n"(K));
    else
        piv = diagcore::readlane_f64(a[K], K);  // (not behind the duplication)
    const double rinv = diagcore::fast_rsqrt_v4(piv);
    bad = (!(piv > 0.0) && bad == 0) ? col0 + K + 1 : bad;
    a[K] *= rinv;
    if constexpr (K < 15) {
        double m;
        asm("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(m) : "v"(x), "v"(rinv));
#pragma unroll
        for (int j = K + 1; j < 16; ++j)
            asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(m), "v"(a[K]), "n"(j));
    }
}
__global__ __launch_bounds__(64) void kdpp(const double *A, double *out, unsigned long long *ts, int reps) {
    const int lane = threadIdx.x;
    double a0[16], acc = 0.0;
    for (int c = 0; c < 16; ++c) a0[c] = lane < 16 ? (c <= lane ? A[lane + 16 * c] : 0.0) : (c == lane - 16 ? 1.0 : 0.0);
    int bad = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
        double a[16];
        for (int c = 0; c < 16; ++c) a[c] = a0[c];
        dpp_step<0>(a, bad, 0); dpp_step<1>(a, bad, 0); dpp_step<2>(a, bad, 0); dpp_step<3>(a, bad, 0);
        dpp_step<4>(a, bad, 0); dpp_step<5>(a, bad, 0); dpp_step<6>(a, bad, 0); dpp_step<7>(a, bad, 0);
        dpp_step<8>(a, bad, 0); dpp_step<9>(a, bad, 0); dpp_step<10>(a, bad, 0); dpp_step<11>(a, bad, 0);
        dpp_step<12>(a, bad, 0); dpp_step<13>(a, bad, 0); dpp_step<14>(a, bad, 0); dpp_step<15>(a, bad, 0);
        if (it == reps - 1)
            for (int c = 0; c < 16; ++c) out[64 + lane * 16 + c] = a[c];
        for (int c = 0; c < 16; ++c) acc += a[c];
        a0[0] += acc * 1e-300;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = acc + bad;
    if (lane == 0) ts[4] = t1 - t0;  // (slot of the 4x4x4 variant, which is not launched any more)
}

// DPP leaf, hand-ordered (every instruction an asm volatile, so the order below IS the issue order): a wave issues in order, and
// the compiler's order above puts the 15 - K updates column K-1 still owes between v_rsq and the Newton step of column K.  Here the owed
// updates fill the latency gaps of the pivot chain instead (latbench: dependent v_mul/v_fma_f64 9.5 cycles, v_rsq_f64 21, a DPP
// read 15 after the write; issue 6 per f64 instruction).
template <int KP, int J>
__device__ __forceinline__ void fm(double (&a)[16], const double &m) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[J]) : "v"(m), "v"(a[KP]), "n"(J));
}
// the IDX-th update column K-1 owes (to column K + 1 + IDX); returns nothing when there is none
template <int K, int IDX>
__device__ __forceinline__ void owed(double (&a)[16], const double &mprev) {
    if constexpr (K >= 1 && K + 1 + IDX <= 15) fm<(K >= 1 ? K - 1 : 0), (K + 1 + IDX <= 15 ? K + 1 + IDX : 15)>(a, mprev);
}
template <int K>
__device__ __forceinline__ void dpp2_step(double (&a)[16], double &mprev, int &bad, int col0, const double c15) {
    constexpr int nowed = K >= 1 ? 14 - K : 0;  // updates column K-1 owes to the columns K+1 .. 15
    int plo, phi;
    asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4" : "=s"(plo), "=s"(phi) : "v"(__double2loint(a[K])), "v"(__double2hiint(a[K])), "n"(K));
    int xlo = __double2loint(a[K]), xhi = __double2hiint(a[K]), ylo = xlo, yhi = xhi;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3" : "+v"(xlo), "+v"(xhi), "+v"(ylo), "+v"(yhi));
    const double x = __hiloint2double(xhi, xlo);  // column K of the leaf rows in lanes 0..15 and again in lanes 16..31
    const double piv = __hiloint2double(phi, plo);
    double r, h, r2, m0, ak0, c, m;
    asm volatile("v_rsq_f64 %0, %1" : "=v"(r) : "s"(piv));
    asm volatile("v_mul_f64 %0, %1, -0.5" : "=v"(h) : "s"(piv));
    owed<K, 0>(a, mprev);
    owed<K, 1>(a, mprev);
    owed<K, 2>(a, mprev);
    asm volatile("s_nop 0\n\tv_mul_f64 %0, %1, %1" : "=v"(r2) : "v"(r));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(m0) : "v"(x), "v"(r));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(ak0) : "v"(a[K]), "v"(r));
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(c) : "v"(h), "v"(r2), "v"(c15));
    owed<K, 3>(a, mprev);
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(m) : "v"(m0), "v"(c));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[K]) : "v"(ak0), "v"(c));
    if constexpr (nowed >= 6) {
        owed<K, 4>(a, mprev);
        owed<K, 5>(a, mprev);
    } else {
        asm volatile("s_nop 1");
    }
    if constexpr (K < 15) fm<K, (K < 15 ? K + 1 : 15)>(a, m);
    if constexpr (nowed < 6) {
        owed<K, 4>(a, mprev);
        owed<K, 5>(a, mprev);
    }
    owed<K, 6>(a, mprev); owed<K, 7>(a, mprev); owed<K, 8>(a, mprev); owed<K, 9>(a, mprev);
    owed<K, 10>(a, mprev); owed<K, 11>(a, mprev); owed<K, 12>(a, mprev); owed<K, 13>(a, mprev);
    bad = (!(piv > 0.0) && bad == 0) ? col0 + K + 1 : bad;
    mprev = m;
}
__global__ __launch_bounds__(64) void kdpp2(const double *A, double *out, unsigned long long *ts, int reps) {
    const int lane = threadIdx.x;
    double a0[16], acc = 0.0;
    for (int c = 0; c < 16; ++c) a0[c] = lane < 16 ? (c <= lane ? A[lane + 16 * c] : 0.0) : (c == lane - 16 ? 1.0 : 0.0);
    int bad = 0;
    double c15 = 1.5;
    asm volatile("" : "+v"(c15));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
        double a[16], mp = 0.0;
        for (int c = 0; c < 16; ++c) a[c] = a0[c];
        dpp2_step<0>(a, mp, bad, 0, c15); dpp2_step<1>(a, mp, bad, 0, c15); dpp2_step<2>(a, mp, bad, 0, c15); dpp2_step<3>(a, mp, bad, 0, c15);
        dpp2_step<4>(a, mp, bad, 0, c15); dpp2_step<5>(a, mp, bad, 0, c15); dpp2_step<6>(a, mp, bad, 0, c15); dpp2_step<7>(a, mp, bad, 0, c15);
        dpp2_step<8>(a, mp, bad, 0, c15); dpp2_step<9>(a, mp, bad, 0, c15); dpp2_step<10>(a, mp, bad, 0, c15); dpp2_step<11>(a, mp, bad, 0, c15);
        dpp2_step<12>(a, mp, bad, 0, c15); dpp2_step<13>(a, mp, bad, 0, c15); dpp2_step<14>(a, mp, bad, 0, c15); dpp2_step<15>(a, mp, bad, 0, c15);
        if (it == reps - 1)
            for (int c = 0; c < 16; ++c) out[64 + lane * 16 + c] = a[c];
        for (int c = 0; c < 16; ++c) acc += a[c];
        a0[0] += acc * 1e-300;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = acc + bad;
    if (lane == 0) ts[3] = t1 - t0;  // (slot of the 16x16x4 chain variant, not launched any more)
}

template <int V, int K>
__device__ __forceinline__ void step(v4d &T, v4d &Y, v4d &Lr, v4d &Yf, double &ps, double &pu, int &bad, int l15, int l4) {
    constexpr int g = K & 3, r = K >> 2;
    if constexpr (V == 0) {
        // T -= l l' and the same column operation on an identity tile (-> L^-T), the second MFMA held back by one column
        const double piv = diagcore::readlane_f64(T[r], 16 * g + K);
        __builtin_amdgcn_sched_barrier(0);
        if (K > 0) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(-ps, pu, Y, 0, 0, 0);
        const double rinv = diagcore::fast_rsqrt_v4(piv);
        const bool ing = l4 == g;
        const double lv = T[r] * rinv;
        const double ls = (ing && l15 > K) ? lv : 0.0;
        if (K < 15) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-ls, ls, T, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const double u = ing ? Y[r] * rinv : 0.0;
        Lr[r] = (ing && l15 >= K) ? lv : Lr[r];
        Yf[r] = ing ? u : Yf[r];
        ps = ls;
        pu = u;
    } else if constexpr (V == 1) {  // no inverse
        const double piv = diagcore::readlane_f64(T[r], 16 * g + K);
        const double rinv = diagcore::fast_rsqrt_v4(piv);
        const bool ing = l4 == g;
        const double lv = T[r] * rinv;
        Lr[r] = (ing && l15 >= K) ? lv : Lr[r];
        const double ls = (ing && l15 > K) ? lv : 0.0;
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(-ls, ls, T, 0, 0, 0);
    } else if constexpr (V == 2) {  // no rsqrt chain: readlane -> mul -> mfma
        const double piv = diagcore::readlane_f64(T[r], 16 * g + K);
        const bool ing = l4 == g;
        const double lv = T[r] * piv;
        const double ls = (ing && l15 > K) ? lv : 0.0;
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(-ls, ls, T, 0, 0, 0);
    } else if constexpr (V == 3) {  // mfma -> mfma through one VALU op (no readlane)
        const double lv = T[r] * 0.5;
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(-lv, lv, T, 0, 0, 0);
    } else if constexpr (V == 4) {  // 4x4x4 f64 mfma chain through one VALU op
        const double lv = T[0] * 0.5;
        const double o = __builtin_amdgcn_mfma_f64_4x4x4f64(-lv, lv, T[0], 0, 0, 0);
        T[0] = o;
    }
}

template <int V>
__global__ __launch_bounds__(64) void k(const double *A, double *out, unsigned long long *ts, int reps) {
    const int lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    v4d T0;
    for (int r = 0; r < 4; ++r) T0[r] = A[l15 + (l4 + 4 * r) * 16];
    v4d acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
        v4d T = T0, Y, Lr = {0, 0, 0, 0}, Yf = {0, 0, 0, 0};
        for (int r = 0; r < 4; ++r) Y[r] = (l15 == l4 + 4 * r) ? 1.0 : 0.0;
        int bad = 0;
        double ps = 0.0, pu = 0.0;
        step<V, 0>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 1>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 2>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 3>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 4>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 5>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 6>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 7>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 8>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 9>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 10>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 11>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 12>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 13>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 14>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        step<V, 15>(T, Y, Lr, Yf, ps, pu, bad, l15, l4);
        for (int r = 0; r < 4; ++r) acc[r] += Lr[r] + Yf[r] + T[r] * 1e-300;
        T0[0] += acc[0] * 1e-300;  // keep the iterations dependent
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
    if (lane == 0) ts[V] = t1 - t0;
}

int main() {
    std::vector<double> A(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) A[i + 16 * j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double *dA, *dout;
    unsigned long long *dts;
    hipMalloc(&dA, 256 * 8);
    hipMalloc(&dout, (64 + 64 * 16) * 8);
    hipMalloc(&dts, 64);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    hipMemset(dts, 0, 64);
    const int reps = 2000;
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    hipLaunchKernelGGL(kold, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    double ref[64], got[64];
    static double fref[64 * 16], fgot[64 * 16];
    hipDeviceSynchronize();
    hipMemcpy(ref, dout, 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(fref, dout + 64, 64 * 16 * 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(ksched, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    hipDeviceSynchronize();
    hipMemcpy(got, dout, 64 * 8, hipMemcpyDeviceToHost);
    double dmax = 0;
    for (int l = 0; l < 32; ++l) dmax = fmax(dmax, fabs(ref[l] - got[l]) / fmax(fabs(ref[l]), 1e-300));
    printf("hand-scheduled vs round-1 leaf: max relative difference of the lane sums %.2e\n", dmax);
    hipLaunchKernelGGL(klds, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    hipDeviceSynchronize();
    hipMemcpy(got, dout, 64 * 8, hipMemcpyDeviceToHost);
    dmax = 0;
    for (int l = 0; l < 32; ++l) dmax = fmax(dmax, fabs(ref[l] - got[l]) / fmax(fabs(ref[l]), 1e-300));
    printf("LDS-broadcast vs round-1 leaf: max relative difference of the lane sums %.2e\n", dmax);
    hipMemcpy(fgot, dout + 64, 64 * 16 * 8, hipMemcpyDeviceToHost);
    {
        int shown = 0;
        for (int l = 0; l < 32; ++l)
            for (int c = 0; c < 16; ++c)
                if (fabs(fref[l * 16 + c] - fgot[l * 16 + c]) > 1e-12 && shown++ < 12) printf("  lane %d col %d: %.6e vs %.6e\n", l, c, fref[l * 16 + c], fgot[l * 16 + c]);
    }
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kdpp, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    hipDeviceSynchronize();
    hipMemcpy(fgot, dout + 64, 64 * 16 * 8, hipMemcpyDeviceToHost);
    {
        int shown = 0;
        double worst = 0;
        for (int l = 0; l < 32; ++l)
            for (int c = 0; c < 16; ++c) {
                if (l < 16 && c > l) continue;  // (above the diagonal of the leaf rows: not part of the result)
                worst = fmax(worst, fabs(fref[l * 16 + c] - fgot[l * 16 + c]));
                if (!(fabs(fref[l * 16 + c] - fgot[l * 16 + c]) <= 1e-12) && shown++ < 12) printf("  lane %d col %d: %.6e vs %.6e\n", l, c, fref[l * 16 + c], fgot[l * 16 + c]);
            }
        printf("DPP leaf vs round-1 leaf: max absolute difference of L and its inverse %.2e\n", worst);
    }
    hipLaunchKernelGGL(kdpp2, dim3(1), dim3(64), 0, 0, dA, dout, dts, reps);
    hipDeviceSynchronize();
    hipMemcpy(fgot, dout + 64, 64 * 16 * 8, hipMemcpyDeviceToHost);
    {
        int shown = 0;
        double worst = 0;
        for (int l = 0; l < 32; ++l)
            for (int c = 0; c < 16; ++c) {
                if (l < 16 && c > l) continue;
                worst = fmax(worst, fabs(fref[l * 16 + c] - fgot[l * 16 + c]));
                if (!(fabs(fref[l * 16 + c] - fgot[l * 16 + c]) <= 1e-12) && shown++ < 12) printf("  lane %d col %d: %.6e vs %.6e\n", l, c, fref[l * 16 + c], fgot[l * 16 + c]);
            }
        printf("hand-ordered DPP leaf vs round-1 leaf: max absolute difference of L and its inverse %.2e\n", worst);
    }
    unsigned long long ts[8];
    hipMemcpy(ts, dts, 64, hipMemcpyDeviceToHost);
    const char *names[8] = {"full step (T and Y mfma)", "no inverse", "readlane -> mul -> mfma", "DPP leaf, hand-ordered (r4)", "leaf with DPP multipliers (r4)", "round-1 leaf (readlane, with inverse)", "round-1 leaf, hand-scheduled", "leaf with LDS-broadcast multipliers (r4)"};
    for (int v = 0; v < 8; ++v) printf("%-28s %.1f cycles per column step\n", names[v], (double)ts[v] / reps / 16);
    return 0;
}
