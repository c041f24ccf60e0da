// Full-matrix Gram assembly lab: does a kernel whose stores and MFMA / radial phases are FORCED to overlap beat
// gram_mfma64_kernel's "store time plus compute time"?  n % 128 == 0, d = 64 (C3), multiquadric; Phi row-major n x n with ld.
//
//   base   the shipped structure (copy of gram_mfma64_kernel: 64 x 128 half tiles, four workgroups per CU, free-running)
//   pp     ping-pong: ONE 512-thread workgroup per CU, two groups of four waves, 64 x 64 units; in every phase one group computes
//          its next unit (MFMA from register-resident row fragments and a wave-private LDS copy of its 16 centre rows, filled by
//          LDS-DMA one phase ahead: no barrier and no vector-memory instruction inside the compute phase) while the other group
//          applies the radial function and stores its previous unit (tile + mirrored tile); ONE workgroup barrier per phase.
//          MODE bits: 1 MFMAs on, 2 stores on, 4 phase barrier on (off: the two groups run free like two workgroups)
//   so     store-only kernels in the shipped grid shape with three block -> tile orders (DRAM / TLB locality of the mirrored tiles)
// Build: hipcc --offload-arch=gfx950 -O3 -o gramlab gramlab.hip      Run: ./gramlab [n] [ld]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <thread>
#include <chrono>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__device__ __forceinline__ void tri_decode(int bid, int &ti, int &tj) {
    int t = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) / 2 <= bid) ++t;
    while (t * (t + 1) / 2 > bid) --t;
    ti = t;
    tj = bid - t * (t + 1) / 2;
}
__device__ __forceinline__ double mq_phi(double s, double a2) {  // -sqrt(1 + a2 s): rsq estimate + two coupled Newton steps (radial.hpp)
    const double t = fma(a2, s, 1.0);
    double y = __builtin_amdgcn_rsq(t);
    double g = t * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double e = fma(-g, g, t);
    return -fma(e, h, g);
}

// ---------------------------------------------------------------- base: the shipped kernel's structure
constexpr int GBM = 128, GBK = 16, GLD = GBK + 2;
__global__ __launch_bounds__(256, 4) void gram_base(const double *__restrict__ Xc, const double *__restrict__ sq, int64_t n, int dpad,
                                                    double *__restrict__ Phi, int64_t ld, double a2, int npairs) {
    __shared__ __attribute__((aligned(16))) double smem[(64 + GBM) * GLD];
    double *As = smem, *Bs = smem + 64 * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int ti, tj;
    const int xb = blockIdx.x & 7, grp = blockIdx.x >> 3;
    int pair = xb + 8 * (grp >> 1), half = grp & 1;
    if (pair >= npairs || (npairs & 7)) {
        pair = blockIdx.x >> 1;
        half = blockIdx.x & 1;
    }
    tri_decode(pair, ti, tj);
    const int64_t I0 = (int64_t)ti * GBM + 64 * half, J0 = (int64_t)tj * GBM;
    v4d acc[4][2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int lr = tid >> 3, lc = (tid & 7) * 2;
    const double *Ap = Xc + (I0 + lr) * dpad + lc, *Bp = Xc + (J0 + lr) * dpad + lc;
    v2d ra[2], rb[4];
    for (int u = 0; u < 2; ++u) ra[u] = *(const v2d *)(Ap + (int64_t)(32 * u) * dpad);
    for (int u = 0; u < 4; ++u) rb[u] = *(const v2d *)(Bp + (int64_t)(32 * u) * dpad);
    const int nkc = dpad / GBK;
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
        for (int u = 0; u < 2; ++u) *(v2d *)&As[(lr + 32 * u) * GLD + lc] = ra[u];
        for (int u = 0; u < 4; ++u) *(v2d *)&Bs[(lr + 32 * u) * GLD + lc] = rb[u];
        __syncthreads();
        if (kc + 1 < nkc) {
            for (int u = 0; u < 2; ++u) ra[u] = *(const v2d *)(Ap + (int64_t)(32 * u) * dpad + (kc + 1) * GBK);
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
    double sqj[2];
    for (int j = 0; j < 2; ++j) sqj[j] = sq[J0 + wave * 32 + j * 16 + l15];
#pragma unroll
    for (int i = 0; i < 4; ++i)
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
                const double v = mq_phi(s, a2);
                acc[i][j][r] = v;
                Phi[gi * ld + gj] = v;
            }
        }
    if (ti == tj) return;
    double *T = smem + wave * (32 * GLD);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(j * 16 + l15) * GLD + l4 + 4 * r] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int jl = it * 8 + (lane >> 3), il = 2 * (lane & 7);
            const v2d v = *(const v2d *)&T[jl * GLD + il];
            *(v2d *)(Phi + (J0 + wave * 32 + jl) * ld + I0 + i * 16 + il) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// ---------------------------------------------------------------- pp: ping-pong groups
constexpr int PP_BW = 1024;        // doubles of a wave's LDS copy of its 16 centre rows (16 x 64)
constexpr int PP_TW = 16 * 18;     // doubles of a wave's transpose strip
constexpr int PP_WAVE = PP_BW + PP_TW;
constexpr size_t PP_LDS = (size_t)8 * PP_WAVE * sizeof(double);

template <int NST>
__device__ __forceinline__ void wait_vm() {
    if constexpr (NST == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (NST == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if constexpr (NST == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void gram_pp(const double *__restrict__ Xc, const double *__restrict__ sq, int64_t n,
                                                  double *__restrict__ Phi, int64_t ld, double a2, const int2 *__restrict__ ranges) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2, w = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    double *Bw = lds + wave * PP_WAVE, *T = Bw + PP_BW;
    const int u0 = ranges[blockIdx.x].x, nu = ranges[blockIdx.x].y - u0;
    v4d a[4][4];  // row fragments of the current 64-row strip: a[it][g] = Xc[I0 + 16 it + l15][16 g + 4 l4 .. + 3]
    v4d acc[4];
    double sqi[4][4], sqj = 0.0;
    int cur_ti = -1, my_ti = 0, my_tj = 0, pend = 0;  // pend: stores issued behind the newest B copy (0, 16, 24)
    auto issue_b = [&](int tj) {  // LDS-DMA: region R = 2 g + sp <- pairs (k = 16 g + 4 (l >> 4) + 2 sp, + 1) of row l & 15
        const double *src = Xc + ((int64_t)tj * 64 + 16 * w + l15) * 64 + 4 * l4;
#pragma unroll
        for (int R = 0; R < 8; ++R)
            __builtin_amdgcn_global_load_lds((glb_void *)(src + 16 * (R >> 1) + 2 * (R & 1)), (lds_void *)(Bw + R * 128), 16, 0, 0);
    };
    if (group < nu) {
        int ti, tj;
        tri_decode(u0 + group, ti, tj);
        issue_b(tj);
    }
    for (int p = 0; p <= nu; ++p) {
        if (p < nu && (p & 1) == group) {
            // ---------------- compute unit u0 + p
            tri_decode(u0 + p, my_ti, my_tj);
            if (pend == 24)
                wait_vm<24>();
            else if (pend == 16)
                wait_vm<16>();
            else
                wait_vm<0>();
            const int64_t I0 = (int64_t)my_ti * 64, J0 = (int64_t)my_tj * 64;
            if (my_ti != cur_ti) {
                cur_ti = my_ti;
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int g = 0; g < 4; ++g) a[it][g] = *(const v4d *)(Xc + (I0 + 16 * it + l15) * 64 + 16 * g + 4 * l4);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sqi[it][r] = sq[I0 + 16 * it + l4 + 4 * r];
            }
            sqj = sq[J0 + 16 * w + l15];
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (MODE & 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v2d b0 = *(const v2d *)(Bw + (2 * g) * 128 + 2 * lane);
                    const v2d b1 = *(const v2d *)(Bw + (2 * g + 1) * 128 + 2 * lane);
                    const double bs[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[it][g][s], bs[s], acc[it], 0, 0, 0);
                }
            }
        } else if (p >= 1 && ((p - 1) & 1) == group) {
            // ---------------- radial function + stores of unit u0 + p - 1 (my_ti, my_tj, acc from the previous phase)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // sqj (and a new strip's fragments) are in; the older stores have long left
            if (p + 1 < nu) {
                int ti, tj;
                tri_decode(u0 + p + 1, ti, tj);
                issue_b(tj);  // in front of this phase's stores in the queue
            }
            const int64_t I0 = (int64_t)my_ti * 64, J0 = (int64_t)my_tj * 64;
            const int64_t gj = J0 + 16 * w + l15;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = I0 + 16 * it + l4 + 4 * r;
                    double s = fma(-2.0, acc[it][r], sqi[it][r] + sqj);
                    s = s > 0.0 ? s : 0.0;
                    if (gi == gj) s = 0.0;
                    const double v = mq_phi(s, a2);
                    acc[it][r] = v;
                    if (MODE & 2) Phi[gi * ld + gj] = v;
                }
            pend = 16;
            if (my_ti != my_tj) {
                pend = 24;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[l15 * 18 + l4 + 4 * r] = acc[it][r];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int jl = t * 8 + (lane >> 3), il = 2 * (lane & 7);
                        const v2d v = *(const v2d *)&T[jl * 18 + il];
                        if (MODE & 2) *(v2d *)(Phi + (J0 + 16 * w + jl) * ld + I0 + 16 * it + il) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            if (!(MODE & 2)) pend = 0;
        }
        if (MODE & 4) __builtin_amdgcn_s_barrier();
    }
}


// ---------------------------------------------------------------- il: stores of the previous strip interleaved into the MFMAs of the next one, per wave
// tools/gramlab/ovl.hip: f64 MFMAs and stores issued by DIFFERENT waves of a CU do not overlap (61 us + 75 us -> 131 us), a wave that
// issues a store behind every four of its own MFMAs gets 91 us.  Hence: every wave is a stream of its own (no barrier, no shared LDS):
// it walks 64 x 16 strips (rows I0 .. I0+63 of a 64-row band held as MFMA A fragments in registers, 16 centre rows J as B fragments in a
// wave-private LDS copy filled by LDS-DMA one strip ahead), and between the MFMAs of strip k it applies the radial function to strip
// k-1 and stores it (tile + mirrored tile through a wave-private 16 x 16 LDS transpose).  |x_i|^2 + |x_j|^2 rides in the MFMA as one
// extra k step (A slot 0 = -|x_i|^2/2, slot 1 = 1; B slot 0 = 1, slot 1 = -|x_j|^2/2 from E2[n][4][2]), so acc = -s/2.
// All LDS traffic is inline asm: the compiler guards ds_reads behind LDS-DMA with s_waitcnt vmcnt(0), which would drain the stores.
constexpr int IL_BW = 1024 + 128;          // doubles per B buffer: 8 regions of 64 lanes x 2 doubles + the extra step's region
constexpr int IL_XS = 80;                  // row stride of Xa = [x (64) | 1, 0,0,0, -|x|^2/2, 0,0,0, 0 ...]: the extra k step rides in the rows
constexpr int IL_TW = 0;                   // (the mirrored tile is transposed across lanes, not through LDS)
constexpr int IL_WAVE = 2 * IL_BW + IL_TW;  // doubles of LDS per wave
constexpr size_t IL_LDS = (size_t)8 * IL_WAVE * sizeof(double);

__device__ __forceinline__ v2d lds_read128(unsigned addr) {
    v2d v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ double lds_read64(unsigned addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void lds_write64(unsigned addr, double v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v)); }

struct ILRange {
    int ti, c, count, pad;
};

// stores with a uniform 64-bit base in scalar registers and a 32-bit byte offset per lane (no vector address arithmetic)
__device__ __forceinline__ void gst64(const double *sbase, unsigned voff, double v) {
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gst128(const double *sbase, unsigned voff, v2d v) {
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
// LDS-DMA with a uniform base in scalar registers, a 32-bit byte offset per lane and the (uniform) LDS byte address through M0
__device__ __forceinline__ void dma16(const void *sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// 4 x 4 transpose of doubles across the four 16-lane rows of a wave: v[r] in row q  ->  v[q] in row r (two swap stages, gfx950)
__device__ __forceinline__ void rows_transpose4(double (&v)[4]) {
    unsigned lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v[r]);
        lo[r] = (unsigned)u;
        hi[r] = (unsigned)(u >> 32);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {  // rows 2, 3 of v[x] <-> rows 0, 1 of v[x + 2]
        auto p = __builtin_amdgcn_permlane32_swap(lo[x], lo[x + 2], false, false);
        lo[x] = p[0];
        lo[x + 2] = p[1];
        auto q = __builtin_amdgcn_permlane32_swap(hi[x], hi[x + 2], false, false);
        hi[x] = q[0];
        hi[x + 2] = q[1];
    }
#pragma unroll
    for (int x = 0; x < 4; x += 2) {  // rows 1, 3 of v[x] <-> rows 0, 2 of v[x + 1]
        auto p = __builtin_amdgcn_permlane16_swap(lo[x], lo[x + 1], false, false);
        lo[x] = p[0];
        lo[x + 1] = p[1];
        auto q = __builtin_amdgcn_permlane16_swap(hi[x], hi[x + 1], false, false);
        hi[x] = q[0];
        hi[x + 1] = q[1];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __longlong_as_double((long long)(((unsigned long long)hi[r] << 32) | lo[r]));
}

// one strip: MFMAs of the current strip (HAVE) into `acc`, radial function + stores of the previous one (`pv`, PREV) between them.
// PDIAG: the previous strip lies in a diagonal 64 x 64 tile (its diagonal entries are phi(0) exactly; no mirrored copy).
// Addresses: uniform bases in scalar registers + one 32-bit lane offset per store shape (doff, moff), so a store costs no vector
// address arithmetic.  Mirrored tile: the four values a lane holds for 16 rows i (i = l4 + 4 r) become, by a 4 x 4 transpose over
// the wave's four 16-lane rows, four CONSECUTIVE i of column j = lane & 15: two 16-byte stores per lane, no LDS.
template <bool HAVE, bool PREV, bool PDIAG, int MODE>
__device__ __forceinline__ void il_step(v4d (&acc)[4], const v4d (&pv)[4], const v4d (&a)[4][4], const double (&ax)[4], unsigned b_addr,
                                        double *__restrict__ Phi, int64_t ld, double a2, int pI, int pJ, int lane, unsigned doff, unsigned moff) {
    const int l15 = lane & 15, l4 = lane >> 4;
    double *const drow = Phi + (int64_t)pI * ld + pJ;  // direct tile: rows pI.., columns pJ..   (uniform)
    double *const mrow = Phi + (int64_t)pJ * ld + pI;  // mirrored tile: rows pJ.., columns pI.. (uniform)
    const int dl = l4 - l15;
    v2d bq0 = {0.0, 0.0}, bq1 = {0.0, 0.0}, bn0 = {0.0, 0.0}, bn1 = {0.0, 0.0};
    if (HAVE) {
        double bx = lds_read64(b_addr + 8 * 1024);
        bq0 = lds_read128(b_addr);
        bq1 = lds_read128(b_addr + 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bx), "+v"(bq0), "+v"(bq1));
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        // |x_i|^2, |x_j|^2 first: 0 + p + q is one rounded sum whichever of the two rows plays A (exact symmetry of the diagonal tiles)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[it] = (MODE & 1) ? __builtin_amdgcn_mfma_f64_16x16x4f64(ax[it], bx, zero, 0, 0, 0) : zero + ax[it] * bx;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (HAVE && g < 3) {
            bn0 = lds_read128(b_addr + (2 * g + 2) * 1024);
            bn1 = lds_read128(b_addr + (2 * g + 3) * 1024);
        }
        const double bs[4] = {bq0.x, bq0.y, bq1.x, bq1.y};
        double pc[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (HAVE && (MODE & 1)) {
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[it][g][s4], bs[s4], acc[it], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PREV) {  // piece (it' = g, r = s4) of the previous strip
                double sv = -2.0 * pv[g][s4];
                sv = sv > 0.0 ? sv : 0.0;
                if (PDIAG && dl == (pJ & 63) - (16 * g + 4 * s4)) sv = 0.0;  // row 16 g + l4 + 4 s4 == column (pJ & 63) + l15 of the tile
                const double v = mq_phi(sv, a2);
                if (MODE & 2) gst64(drow + (int64_t)(16 * g + 4 * s4) * ld, doff, v);
                pc[s4] = v;
                if (!PDIAG && s4 == 3) {
                    rows_transpose4(pc);
                    if (MODE & 2) {
                        gst128(mrow + 16 * g, moff, (v2d){pc[0], pc[1]});
                        gst128(mrow + 16 * g + 2, moff, (v2d){pc[2], pc[3]});
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bn0), "+v"(bn1));
        bq0 = bn0;
        bq1 = bn1;
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void gram_il(const double *__restrict__ Xc, const double *__restrict__ sq, const double *__restrict__ E2,
                                                  int64_t n, double *__restrict__ Phi, int64_t ld, double a2, const ILRange *__restrict__ ranges, int stride) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    double *Lw = lds + wave * IL_WAVE;
    const unsigned lw_addr = (unsigned)(unsigned long long)(lds_void *)Lw;
    const unsigned doff = (unsigned)(l4 * ld + l15) * 8u, moff = (unsigned)(l15 * ld + 4 * l4) * 8u;  // byte offsets
    const ILRange rg = ranges[blockIdx.x * 8 + wave];
    int ti = __builtin_amdgcn_readfirstlane(rg.ti), c = __builtin_amdgcn_readfirstlane(rg.c);
    const int count = __builtin_amdgcn_readfirstlane(rg.count);
    if (count <= 0) return;
    v4d a[4][4];
    double ax[4];
    v4d X[4], Y[4];
    auto load_a = [&](int t) {  // (drains this wave's vector-memory queue: a row change costs a round trip)
        const int64_t I0 = (int64_t)t * 64;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) a[it][g] = *(const v4d *)(Xc + (I0 + 16 * it + l15) * IL_XS + 16 * g + 4 * l4);
            const double q = sq[I0 + 16 * it + l15];
            ax[it] = l4 == 0 ? -0.5 * q : (l4 == 1 ? 1.0 : 0.0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(a[it][g]));  // the compiler's wait for these loads goes HERE, not in front of every MFMA block
            asm volatile("" : "+v"(ax[it]));
        }
    };
    const unsigned xoff = (unsigned)(l15 * IL_XS + 4 * l4) * 8u;  // lane byte offset into a strip's rows of Xa
    auto issue_b = [&](int cc, int buf) {  // 16 centre rows of strip cc -> buffer buf: 8 regions (R = 2 g + sp) + the extra step's region
        const char *xs = (const char *)(Xc + (int64_t)cc * 16 * IL_XS);  // uniform
        const unsigned dst = lw_addr + buf * IL_BW * 8;
#pragma unroll
        for (int R = 0; R < 8; ++R) dma16(xs + (16 * (R >> 1) + 2 * (R & 1)) * 8, xoff, dst + R * 1024);
        dma16(xs + 64 * 8, xoff, dst + 8192);  // the extra k step: Xa[j][64 + 4 l4] = {1, -|x_j|^2/2, 0, 0}[l4]
    };
    // strip bookkeeping, all wave-uniform
    int pI = 0, pJ = 0, pdiag = 0;   // previous strip
    int pend = 0;                    // stores this wave has issued behind its newest B copy (0, 16 or 24)
    auto advance = [&](int &nti, int &nc) {
        nti = ti;
        nc = c + stride;  // stride 1: a wave walks consecutive strips; stride 8: the eight waves of a workgroup walk a range together
        while (nc >= 4 * (nti + 1)) {
            nc -= 4 * (nti + 1);
            ++nti;
        }
    };
    auto wait_b = [&]() {
        if (pend == 24)
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (pend == 16)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    load_a(ti);
    issue_b(c, 0);
    int k = 0;
    // generic step: B(k) is in; issue B(k+1); MFMAs of strip k into CUR, epilogue of strip k-1 from PRV (if HASPREV)
#define IL_ITER(CUR, PRV, HASPREV)                                                                                              \
    {                                                                                                                            \
        wait_b();                                                                                                                \
        int nti, nc;                                                                                                             \
        advance(nti, nc);                                                                                                        \
        if (k + 1 < count) issue_b(nc, (k + 1) & 1);                                                                             \
        if (HASPREV && pdiag)                                                                                                    \
            il_step<true, HASPREV, true, MODE>(CUR, PRV, a, ax, lw_addr + (k & 1) * IL_BW * 8 + lane * 16, Phi, ld, a2, pI, pJ, lane, doff, moff); \
        else                                                                                                                     \
            il_step<true, HASPREV, false, MODE>(CUR, PRV, a, ax, lw_addr + (k & 1) * IL_BW * 8 + lane * 16, Phi, ld, a2, pI, pJ, lane, doff, moff); \
        pend = (HASPREV && (MODE & 2)) ? (pdiag ? 16 : 24) : 0;                                                                                \
        pI = ti * 64;                                                                                                            \
        pJ = c * 16;                                                                                                             \
        pdiag = c >= 4 * ti;                                                                                                     \
        if (k + 1 < count && nti != ti) {                                                                                        \
            load_a(nti);                                                                                                         \
            pend = 0;                                                                                                            \
        }                                                                                                                        \
        ti = nti;                                                                                                                \
        c = nc;                                                                                                                  \
        ++k;                                                                                                                     \
    }
    IL_ITER(X, Y, false)
    bool lastX = true;
    while (k < count) {
        IL_ITER(Y, X, true)
        lastX = false;
        if (k < count) {
            IL_ITER(X, Y, true)
            lastX = true;
        }
    }
#undef IL_ITER
    if (lastX) {
        if (pdiag)
            il_step<false, true, true, MODE>(Y, X, a, ax, 0u, Phi, ld, a2, pI, pJ, lane, doff, moff);
        else
            il_step<false, true, false, MODE>(Y, X, a, ax, 0u, Phi, ld, a2, pI, pJ, lane, doff, moff);
    } else {
        if (pdiag)
            il_step<false, true, true, MODE>(X, Y, a, ax, 0u, Phi, ld, a2, pI, pJ, lane, doff, moff);
        else
            il_step<false, true, false, MODE>(X, Y, a, ax, 0u, Phi, ld, a2, pI, pJ, lane, doff, moff);
    }
}

// ---------------------------------------------------------------- so: store-only, three block -> tile orders
template <int ORDER>
__global__ __launch_bounds__(256, 4) void store_only(double *__restrict__ Phi, long ld, double seed, int nt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int ti, tj;
    const int pair = blockIdx.x >> 1, half = blockIdx.x & 1;
    if (ORDER == 0) {
        tri_decode(pair, ti, tj);
    } else {
        // super-tiles of S x S tile pairs (S = 8 or 16), walked in triangular order; inside a super-tile row-major; a diagonal
        // super-tile holds S (S + 1) / 2 pairs -- the kernel is launched over nst (nst + 1) / 2 * S * S slots and empty slots return
        constexpr int S = (ORDER == 1) ? 8 : 16;
        const int sp = pair / (S * S), in = pair % (S * S);
        int si, sj;
        tri_decode(sp, si, sj);
        ti = si * S + in / S;
        tj = sj * S + in % S;
        if (tj > ti || ti >= nt) return;
    }
    const long I0 = (long)ti * 128 + 64 * half, J0 = (long)tj * 128;
    const double v = seed + blockIdx.x;
    const v2d vv = {v, v + 1.0};
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 4; ++r) *(v2d *)(Phi + (I0 + i * 16 + l4 + 4 * r) * ld + J0 + wave * 32 + 2 * l15) = vv;
    if (ti == tj) return;
    for (int p = 0; p < 2; ++p)
        for (int it = 0; it < 8; ++it) *(v2d *)(Phi + (J0 + wave * 32 + 4 * it + l4) * ld + I0 + 32 * p + 2 * l15) = vv;
}


int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 8192, ld = argc > 2 ? atol(argv[2]) : n;
    const int d = 64;
    const double a2 = 1.0;
    std::vector<double> X((size_t)n * d), sqh(n);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() {
        st ^= st << 13;
        st ^= st >> 7;
        st ^= st << 17;
        return (double)(st >> 11) / 9007199254740992.0;
    };
    for (auto &x : X) x = rnd();
    for (int c = 0; c < d; ++c) {
        double m = 0;
        for (long i = 0; i < n; ++i) m += X[i * d + c];
        m /= n;
        for (long i = 0; i < n; ++i) X[i * d + c] -= m;
    }
    for (long i = 0; i < n; ++i) {
        double s = 0;
        for (int c = 0; c < d; ++c) s += X[i * d + c] * X[i * d + c];
        sqh[i] = s;
    }
    double *dX, *dsq, *Phi, *Ref;
    CK(hipMalloc(&dX, X.size() * 8));
    CK(hipMalloc(&dsq, n * 8));
    CK(hipMalloc(&Phi, (size_t)ld * n * 8));
    CK(hipMalloc(&Ref, (size_t)ld * n * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsq, sqh.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = 8.0 * n * n + 8.0 * n * d;
    auto report = [&](const char *name, float ms) {
        printf("%-58s %8.1f us  %.2f TB/s  frac of 8 TB/s %.3f\n", name, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
        fflush(stdout);
    };
    // back-to-back average (20 launches) AND single launches from an idle device (median of 9; what mrbf_gram's own events see)
    auto measure = [&](const std::function<void()> &launch, const char *name) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipGetLastError());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        std::vector<float> one;
        for (int i = 0; i < 9; ++i) {
            CK(hipDeviceSynchronize());
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float u;
            CK(hipEventElapsedTime(&u, e0, e1));
            one.push_back(u);
        }
        std::sort(one.begin(), one.end());
        const float b2b = t / 20, med = one[4];
        printf("%-66s back-to-back %6.1f us (%.3f)   single %6.1f us (%.3f of 8 TB/s; min %.1f)\n", name, b2b * 1e3, bytes / b2b / 1e9 / 8.0,
               med * 1e3, bytes / med / 1e9 / 8.0, one[0] * 1e3);
        fflush(stdout);
    };
    const int reps = 20;
    // ---- base
    const long nt = n / 128, nb = nt * (nt + 1) / 2;
    auto run_base = [&](double *out) { hipLaunchKernelGGL(gram_base, dim3((unsigned)(2 * nb)), dim3(256), 0, 0, dX, dsq, n, d, out, ld, a2, (int)nb); };
    measure([&]() { run_base(Ref); }, "base (shipped structure, 64 x 128 halves, 4 wg / CU)");
    float ms;
    // ---- pp ranges: contiguous unit ranges of equal cost (a unit costs 1, the first unit of a 64-row strip 1 + ROWPEN)
    const long nt64 = n / 64, nunits = nt64 * (nt64 + 1) / 2;
    const int G = 256;
    auto make_ranges = [&](double rowpen) {
        std::vector<int2> r(G);
        const double total = (double)nunits + rowpen * nt64;
        double c = 0;
        long u = 0;
        int wgi = 0;
        r[0].x = 0;
        for (long ti = 0; ti < nt64; ++ti)
            for (long tj = 0; tj <= ti; ++tj, ++u) {
                c += 1.0 + (tj == 0 ? rowpen : 0.0);
                while (wgi + 1 < G && c > total * (wgi + 1) / G) {
                    r[wgi].y = (int)(u + 1);
                    ++wgi;
                    r[wgi].x = (int)(u + 1);
                }
            }
        r[wgi].y = (int)nunits;
        for (int k = wgi + 1; k < G; ++k) r[k].x = r[k].y = (int)nunits;
        return r;
    };
    int2 *dr;
    CK(hipMalloc(&dr, G * sizeof(int2)));
    auto time_pp = [&](auto kern, const char *name, double rowpen, bool check) {
        auto r = make_ranges(rowpen);
        CK(hipMemcpy(dr, r.data(), G * sizeof(int2), hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_LDS));
        CK(hipMemset(Phi, 0, (size_t)ld * n * 8));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(512), PP_LDS, 0, dX, dsq, n, Phi, ld, a2, dr);
        CK(hipGetLastError());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(512), PP_LDS, 0, dX, dsq, n, Phi, ld, a2, dr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        report(name, t / reps);
        if (check) {
            std::vector<double> hp((size_t)ld * n), hr((size_t)ld * n);
            CK(hipMemcpy(hp.data(), Phi, hp.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), Ref, hr.size() * 8, hipMemcpyDeviceToHost));
            double worst = 0, worst_ref = 0;
            long bad = 0;
            for (long i = 0; i < n; ++i)
                for (long j = 0; j < n; ++j) {
                    const double x = hp[i * ld + j], y = hr[i * ld + j];
                    const double e = fabs(x - y) / fmax(1.0, fabs(y));
                    if (!(e <= 1e-13)) ++bad;
                    if (e > worst) worst = e;
                }
            for (int smp = 0; smp < 20000; ++smp) {  // against the difference form on the host
                const long i = (long)(rnd() * n), j = (long)(rnd() * n);
                double s = 0;
                for (int c = 0; c < d; ++c) s += (X[i * d + c] - X[j * d + c]) * (X[i * d + c] - X[j * d + c]);
                const double y = -sqrt(1.0 + a2 * s);
                worst_ref = fmax(worst_ref, fabs(hp[i * ld + j] - y) / fabs(y));
            }
            printf("    check: max rel diff to base %.2e (entries beyond 1e-13: %ld), to the host difference form %.2e\n", worst, bad, worst_ref);
        }
    };
    time_pp(gram_pp<3>, "pp  MFMA + stores, groups free-running", 1.5, false);
    // ---- il: per-wave strip ranges of equal cost (a strip costs 1, a row change ROWPEN)
    {
        std::vector<double> xa((size_t)n * IL_XS, 0.0);
        for (long i = 0; i < n; ++i) {
            for (int c2 = 0; c2 < d; ++c2) xa[i * IL_XS + c2] = X[i * d + c2];
            xa[i * IL_XS + 64] = 1.0;
            xa[i * IL_XS + 68] = -0.5 * sqh[i];
        }
        double *dXa, *dE2 = nullptr;
        CK(hipMalloc(&dXa, xa.size() * 8));
        CK(hipMemcpy(dXa, xa.data(), xa.size() * 8, hipMemcpyHostToDevice));
        ILRange *dil;
        const int NW = 256 * 8;
        CK(hipMalloc(&dil, NW * sizeof(ILRange)));
        auto make_il = [&](double rowpen) {
            std::vector<ILRange> r(NW, ILRange{0, 0, 0, 0});
            const long nstrips = 2 * nt64 * (nt64 + 1);
            const double total = (double)nstrips + rowpen * nt64;
            double cst = 0;
            int wv = 0;
            r[0] = ILRange{0, 0, 0, 0};
            for (long ti = 0; ti < nt64; ++ti)
                for (long c = 0; c < 4 * (ti + 1); ++c) {
                    cst += 1.0 + (c == 0 ? rowpen : 0.0);
                    r[wv].count++;
                    if (wv + 1 < NW && cst >= total * (wv + 1) / NW) {
                        ++wv;
                        long nc = c + 1, nti = ti;
                        if (nc == 4 * (ti + 1)) {
                            nc = 0;
                            nti = ti + 1;
                        }
                        r[wv] = ILRange{(int)nti, (int)nc, 0, 0};
                    }
                }
            return r;
        };
        int il_stride = 1;
        auto run_il = [&](auto kern, const std::vector<ILRange> &r, const char *nm, bool check) {
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IL_LDS));
            CK(hipMemcpy(dil, r.data(), NW * sizeof(ILRange), hipMemcpyHostToDevice));
            if (check) CK(hipMemset(Phi, 0, (size_t)ld * n * 8));
            measure([&]() { hipLaunchKernelGGL(kern, dim3(256), dim3(512), IL_LDS, 0, dXa, dsq, dE2, n, Phi, ld, a2, dil, il_stride); }, nm);
            if (check) {
                std::vector<double> hp((size_t)ld * n), hr((size_t)ld * n);
                CK(hipMemcpy(hp.data(), Phi, hp.size() * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hr.data(), Ref, hr.size() * 8, hipMemcpyDeviceToHost));
                double worst = 0;
                long bad = 0, asym = 0;
                for (long i = 0; i < n; ++i)
                    for (long j = 0; j < n; ++j) {
                        const double e = fabs(hp[i * ld + j] - hr[i * ld + j]) / fmax(1.0, fabs(hr[i * ld + j]));
                        if (!(e <= 1e-13)) ++bad;
                        if (e > worst) worst = e;
                        if (hp[i * ld + j] != hp[j * ld + i]) ++asym;
                    }
                printf("    check: max rel diff to base %.2e (entries beyond 1e-13: %ld), asymmetric entries %ld\n", worst, bad, asym);
            }
        };
        for (double rowpen : {2.0, 1.0, 1.5, 2.5, 3.0}) {
            char nm[96];
            snprintf(nm, sizeof nm, "il  per-wave streams, stores inside the MFMAs, rowpen %.1f", rowpen);
            run_il(gram_il<3>, make_il(rowpen), nm, rowpen == 2.0);
        }
        // interleaved: a workgroup owns a contiguous range of strips (equal cost), its eight waves take every eighth strip of it -- at any
        // time the eight waves write eight ADJACENT 128-byte column strips of the same rows (1 KB runs instead of isolated lines)
        auto make_il8 = [&](double rowpen) {
            std::vector<ILRange> r(NW, ILRange{0, 0, 0, 0});
            const long nstrips = 2 * nt64 * (nt64 + 1);
            std::vector<long> cut(257, nstrips);
            const double total = (double)nstrips + rowpen * nt64;
            double cst = 0;
            long sidx = 0;
            int wg = 0;
            cut[0] = 0;
            for (long ti = 0; ti < nt64; ++ti)
                for (long c = 0; c < 4 * (ti + 1); ++c, ++sidx) {
                    cst += 1.0 + (c == 0 ? rowpen : 0.0);
                    if (wg + 1 < 256 && cst >= total * (wg + 1) / 256) cut[++wg] = sidx + 1;
                }
            for (int b = 0; b < 256; ++b)
                for (int w2 = 0; w2 < 8; ++w2) {
                    const long s0 = cut[b] + w2, s1 = cut[b + 1];
                    if (s0 >= s1) continue;
                    long ti = (long)((sqrt(1.0 + 2.0 * s0) - 1.0) / 2.0);
                    while (2 * (ti + 1) * (ti + 2) <= s0) ++ti;
                    while (2 * ti * (ti + 1) > s0) --ti;
                    r[b * 8 + w2] = ILRange{(int)ti, (int)(s0 - 2 * ti * (ti + 1)), (int)((s1 - s0 + 7) / 8), 0};
                }
            return r;
        };
        il_stride = 8;
        for (double rowpen : {2.0, 0.0, 4.0}) {
            char nm[96];
            snprintf(nm, sizeof nm, "il8 waves of a workgroup interleaved over its strips, rowpen %.1f", rowpen);
            run_il(gram_il<3>, make_il8(rowpen), nm, rowpen == 2.0);
        }
        run_il(gram_il<2>, make_il8(2.0), "il8 stores + radial function, no MFMAs, rowpen 2", false);
        il_stride = 1;
        run_il(gram_il<1>, make_il(2.0), "il  MFMA + radial function, no global stores, rowpen 2", false);
        run_il(gram_il<2>, make_il(2.0), "il  stores + radial function, no MFMAs, rowpen 2", false);
        {   // every wave 16 strips of one row (no row change, perfect balance; overwrites the same region -- timing only)
            std::vector<ILRange> r(NW);
            for (int wv = 0; wv < NW; ++wv) {
                const int ti = 8 + wv % (int)(nt64 - 8);
                r[wv] = ILRange{ti, (wv / 7) % (4 * (ti + 1) - 16), 16, 0};
            }
            run_il(gram_il<3>, r, "il  every wave 16 strips of ONE row (balance / row-change bound), 33k strips", false);
            run_il(gram_il<1>, r, "il  the same, no global stores", false);
            run_il(gram_il<2>, r, "il  the same, no MFMAs", false);
        }
    }
    // ---- store-only orders
    auto so = [&](auto kern, unsigned grid, const char *name) {
        measure([&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, Phi, ld, 1.0, (int)nt); }, name);
    };
    so(store_only<0>, (unsigned)(2 * nb), "so  store-only, shipped order (tile rows)");
    {
        const long ns8 = (nt + 7) / 8, ns16 = (nt + 15) / 16;
        so(store_only<1>, (unsigned)(2 * ns8 * (ns8 + 1) / 2 * 64), "so  store-only, 8 x 8 super-tiles");
        so(store_only<2>, (unsigned)(2 * ns16 * (ns16 + 1) / 2 * 256), "so  store-only, 16 x 16 super-tiles");
    }
    CK(hipMemsetAsync(Phi, 0, (size_t)n * n * 8, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(Phi, 0, (size_t)n * n * 8, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    report("hipMemsetAsync of 8 n^2 bytes", ms / 10);
    return 0;
}
