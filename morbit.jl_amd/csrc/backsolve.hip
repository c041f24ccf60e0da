// Backward substitution  L' x = y  for k right-hand sides as ONE persistent launch.
//
// chol_blocked.hip's backsolve_blocked issues one launch per block row (64 dependent launches x 11 us at n = 8192:
// profiles/r01_d_kernel_stats.csv).  Here, for block j of the unknowns,
//     x_j = inv(L_jj)' (y_j - sum_{i >= j+3} L(i,j)' x_i)  -  M2_j' x_{j+2}  -  M1_j' x_{j+1},     Mt_j = L(j+t,j) inv(L_jj).
// The two products that depend on the newest blocks use tiles PRE-MULTIPLIED with the inverse (premul_kernel, one launch before the
// solve: 2 (nb - 1) products of 128^3), so that between the arrival of x_{j+1} and the publication of x_j lies ONE tile product (round 2:
// two -- L(j+1,j)' x then inv(L_jj)' y -- with a barrier-separated LDS reduction each), and two workgroups share a block column: a
// streamer for the sum and the product with inv(L_jj)', a finisher for the two pre-multiplied products (see the kernel).
// What round 3's per-block stamps (MRBF_BSOLVE_STAMPS=1) and tools/pingpong showed, in the order found:
//   * an undisturbed one-word hand-off between two workgroups takes 0.6 us across XCDs (0.3 through one L2), not the 2-3 us a step
//     of round 2's chain seemed to lose: the 5.4 us per step were the FETCH of the tiles -- each load instruction took 16 bytes from
//     64 different lines, 8192 address cycles of the CU's L1 per 128 x 128 tile (3.4 us).  Eight consecutive lanes now read one line
//     (load_tile): 5.4 -> 2.8 us per step;
//   * a poll is a vector load and returns in order, behind every tile fetch issued before it by the same wave, so a workgroup that
//     prefetches and polls pays a memory latency per step: hence the two roles, the first look at an arrival two steps ahead of its
//     use, the fixed number of loads per step (exact vmcnt waits) and no fetch in flight during a streamer's last three steps;
//   * sums over lanes by DPP row shifts instead of LDS-routed shuffles (0.74 -> 0.45 us from barrier to publication).
// n = 8192, two right-hand sides: 343 us (round 2) -> 160 us + 25 us for the pre-multiplication.
// Hand-off (MI355X: XCD L2s are not coherent): the data is its own flag.  x_j / z_j go to exchange buffers that the host filled with
// an all-ones pattern (a NaN no arithmetic produces); the producer's lanes store their entries write-through (sc1) and are done -- no
// drain, no barrier, no flag store; each consumer thread polls ITS entry with sc1 loads until it is no longer the pattern.
#include "common.hpp"

namespace mrbf {

namespace bsolve {

constexpr int NB = 128, NTHR = 512;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) double gf64;
typedef __attribute__((address_space(1))) v2d gv2d;

template <int KB>
struct Shared {
    double xs[2][KB][NB];  // source vector of a step (the published x_i, or this workgroup's running right-hand side), by step parity
    int ok;
};

// M_t(j) = L(j+1+t, j) inv(L_jj), t = 0, 1 -> Mbuf[(2 j + t) 128^2], column-major, ld 128.  Four workgroups per product (32 columns
// each; inv(L_jj) is lower triangular: columns >= c0 only meet rows k >= c0).  f64 MFMA 16x16x4, operands straight from global
// memory in bursts of eight k-steps, the next burst in flight under this burst's MFMAs.
// (also resets the chain's flag lines and marks the exchange buffers "not published yet" for the first pass of right-hand sides: two
//  memset launches less in front of the persistent kernel)
__global__ __launch_bounds__(256) void premul_kernel(const double *__restrict__ L, int64_t lda, const double *__restrict__ linv_all,
                                                     double *__restrict__ Mbuf, int nb, unsigned *__restrict__ flags, int nflags,
                                                     unsigned long long *__restrict__ xb, int nxb) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nflags; e += gridDim.x * 256) flags[e] = 0u;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nxb; e += gridDim.x * 256) xb[e] = ~0ull;
    const int slab = blockIdx.x & 3, t = (blockIdx.x >> 2) & 1, j = blockIdx.x >> 3;
    const int i = j + 1 + t;
    if (i >= nb) return;
    const double *A = L + (int64_t)i * NB + (int64_t)j * NB * lda;
    const double *B = linv_all + (size_t)j * NB * NB;
    double *M = Mbuf + ((size_t)j * 2 + t) * NB * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int i0 = 32 * wave, j0 = 32 * slab;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double *pa = A + (i0 + l15) + (int64_t)(j0 + l4) * lda;
    const double *pb = B + (j0 + l4) + (int64_t)(j0 + l15) * NB;
    constexpr int UK = 8;
    double fa0[2][UK], fa1[2][UK], fb0[2][UK], fb1[2][UK];
    auto burst = [&](int buf) {
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            fa0[buf][u] = pa[(int64_t)4 * u * lda];
            fa1[buf][u] = pa[(int64_t)4 * u * lda + 16];
            fb0[buf][u] = pb[4 * u];
            fb1[buf][u] = pb[4 * u + 16 * NB];
        }
        pa += (int64_t)4 * UK * lda;
        pb += 4 * UK;
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[buf][u], fb0[buf][u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[buf][u], fb1[buf][u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[buf][u], fb0[buf][u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[buf][u], fb1[buf][u], acc[1][1], 0, 0, 0);
        }
    };
    const int nburst = (NB - j0) / (4 * UK);  // 4, 3, 2, 1
    burst(0);
    for (int b = 0; b < nburst; b += 2) {
        if (b + 1 < nburst) burst(1);
        mma(0);
        if (b + 1 >= nburst) break;
        if (b + 2 < nburst) burst(0);
        mma(1);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(i0 + 16 * a + l4 + 4 * r) + (size_t)(j0 + 16 * b + l15) * NB] = acc[a][b][r];
}

// Tile in registers: wave w owns columns 16w .. 16w+15; lane (q = l & 7, c8 = l >> 3) holds, of its two columns 16w + 2 c8 + cc, the
// rows 16u + 2q, 16u + 2q + 1 (u = 0..7).  Eight consecutive lanes read one whole 128-byte line, so a load instruction touches 8
// lines (round 2 / first version of this kernel: 64 lines with 16 bytes taken from each -- a tile then cost 8192 address cycles of the
// CU's L1, 3.4 us, and the FETCH of a tile, not the arithmetic, set the pace of the chain).
__device__ __forceinline__ void load_tile(const double *__restrict__ T, int64_t ldt, v2d (&r)[2][8]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double *p = T + (int64_t)(16 * w + 2 * (lane >> 3)) * ldt + 2 * (lane & 7);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int u = 0; u < 8; ++u) r[cc][u] = *(const gv2d *)(p + cc * ldt + 16 * u);
}

// lane i receives lane i + N of its row of 16 (DPP row_shl); lanes shifted in from outside the row keep their own value
template <int N>
__device__ __forceinline__ double shl_lanes(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, 0x100 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), 0x100 + N, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// out[cc][l] = (T' xs[l])[column 16w + 2 c8 + cc], valid in the lane q = 0 of the eight lanes that share the columns
template <int KB>
__device__ __forceinline__ void tile_dot(const double (*xs)[NB], const v2d (&r)[2][8], double (&out)[2][KB]) {
    const int q = threadIdx.x & 7;
    double acc[2][KB];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int l = 0; l < KB; ++l) acc[cc][l] = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int l = 0; l < KB; ++l) {
            const v2d x = *(const v2d *)&xs[l][16 * u + 2 * q];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) acc[cc][l] = fma(r[cc][u][1], x[1], fma(r[cc][u][0], x[0], acc[cc][l]));
        }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int l = 0; l < KB; ++l) {
            // sum over the eight lanes of a group, complete in the lane q = 0 only (the one that keeps the running sums and
            // publishes): three DPP row shifts instead of LDS-routed exchanges
            double v = acc[cc][l];
            v += shl_lanes<4>(v);
            v += shl_lanes<2>(v);
            v += shl_lanes<1>(v);
            out[cc][l] = v;
        }
}

// One arrival: the threads tid < KB * 128 wait for their entries of block i of an exchange buffer and put them into xs.  The buffer is
// written through to memory (sc1 stores) and read past the L1 (sc1 loads).  `spec`: the value of a first look issued earlier (or the
// fill pattern).  (Tried: a second copy written with plain stores, which stay in the producer's L2 where a consumer on the same XCD
// finds them after 0.3 us instead of 0.6 -- tools/pingpong -- with the finishers of eight consecutive blocks placed on one XCD.  Not
// kept: a line of that copy left in an L2 by an EARLIER solve is indistinguishable from a fresh one, and one run of the concurrency
// test returned a wrong right-hand side.)
template <int KB>
__device__ __forceinline__ void await_block(Shared<KB> &sh, double (*xs)[NB], const double *glob, int i, double spec, int backoff,
                                            unsigned *abortw, unsigned epoch, unsigned long long spin_ticks, int *status, int code) {
    const int tid = threadIdx.x;
    if (tid >= KB * NB) return;
    const gf64 *src = (const gf64 *)&glob[((size_t)i * KB + tid / NB) * NB + (tid % NB)];
    double v = spec;
    if (__double_as_longlong(v) == -1ll) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while (__double_as_longlong(v) == -1ll) {
        if ((++spins & 63u) == 0u) {
            // somebody else gave up: stop at once instead of timing out one dependant after the other
            if (__hip_atomic_load((const gu32 *)abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                sh.ok = 0;
                break;
            }
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > spin_ticks) {
                __hip_atomic_store((gu32 *)abortw, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(__attribute__((address_space(1))) int *)status = code + (i & 0xff);  // read back with the factorisation's flags
                sh.ok = 0;
                break;
            }
        }
        // only those next in the chain poll at full rate: a block is awaited by every workgroup below it at once
        for (int b = 0; b < backoff; ++b) __builtin_amdgcn_s_sleep(8);
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    xs[tid / NB][tid % NB] = v;
}

// a bare s_barrier behind the LDS wait (__syncthreads() also waits for vmcnt(0), i.e. for the tile fetches in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte write-through store (pollers read 8-byte halves; observed untorn on gfx950); the fill pattern is never published
__device__ __forceinline__ void publish(double *dst, v2d v) {
    if (__double_as_longlong(v[0]) == -1ll) v[0] = __longlong_as_double(0x7ff8000000000000ll);
    if (__double_as_longlong(v[1]) == -1ll) v[1] = __longlong_as_double(0x7ff8000000000000ll);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}

// Two workgroups per block column j:
//   the STREAMER walks L(nb-1,j) .. L(j+3,j) as the x_i arrive (two tiles ahead in registers), multiplies the remaining right-hand side by
//   inv(L_jj)' and publishes z_j;
//   the FINISHER holds M2_j, M1_j in registers from the start and does nothing but  x_j = (z_j - M2' x_{j+2}) - M1' x_{j+1}.
// A poll is a vector load and returns in order, i.e. BEHIND every tile fetch issued before it (one workgroup doing both paid a memory
// latency on every step of the chain: 2.9 us per block, measured with the per-block stamps below); the finisher has no fetch in flight
// when it waits, and the streamer takes its first look at the NEXT arrival before it issues the next fetch (when it lags, the value
// is there and it never waits for a fetch; when it is ahead it may wait, it has the time).
// Grid: block 2p is the streamer, block 2p + 1 the finisher of position p of the chain (p = 0: block column nb - 1): dispatch order is
// chain order, streamer before finisher, so every workgroup only waits for workgroups dispatched BEFORE it and the launch drains at any
// residency (other kernels holding CUs, several contexts on one GPU, more workgroups than CUs).
template <int KB>
__global__ __launch_bounds__(NTHR, 1) void backsolve_persistent_kernel(const double *__restrict__ L, int64_t lda, const double *__restrict__ linv_all,
                                                                        const double *__restrict__ Mbuf, double *__restrict__ Y, int64_t ldy, int k0,
                                                                        int nb, unsigned *flags, unsigned epoch, unsigned long long spin_ticks,
                                                                        int *status, int fault, double *__restrict__ xb, double *__restrict__ zb,
                                                                        long long *stamps) {
    __shared__ __attribute__((aligned(16))) Shared<KB> sh;
    const int p = (int)blockIdx.x >> 1;
    const bool finisher = blockIdx.x & 1;
    const int j = nb - 1 - p, tid = threadIdx.x;
    const int lane = tid & 63, col = 16 * (tid >> 6) + 2 * (lane >> 3), q = lane & 7;  // this lane's columns: col, col + 1
    unsigned *abortw = flags + (size_t)nb * 32;
    const int na = nb - 1 - j, npm = na < 2 ? na : 2, npl = na - npm;  // arrivals; those met by M2 / M1; those met by L tiles
    if (tid == 0) sh.ok = 1;
    __syncthreads();
    const double fillv = __longlong_as_double(-1ll);
    if (!finisher) {
        double w[2][KB];  // the running right-hand side (valid in the lanes q = 0)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int l = 0; l < KB; ++l) w[cc][l] = *(const gf64 *)&Y[(int64_t)(k0 + l) * ldy + (int64_t)j * NB + col + cc];
        // Three tiles in flight.  The steady part of the walk is straight-line code with a fixed number of loads per step, so that the
        // compiler's wait in front of a product counts exactly the loads issued after that tile's (with conditional fetches it falls
        // back to vmcnt(0) and every step pays a memory latency); the last three steps -- x_{j+4}, x_{j+3}, inv(L_jj): the ones the
        // finisher is waiting for -- run with their tiles in registers and no fetch in flight.
        const int nst = npl + 1;
        auto fetch = [&](int s, v2d(&r)[2][8]) {  // s < npl: L(nb-1-s, j); s >= npl: inv(L_jj)
            const double *src = s < npl ? L + (int64_t)(nb - 1 - s) * NB + (int64_t)j * NB * lda : linv_all + (size_t)j * NB * NB;
            load_tile(src, s < npl ? lda : (int64_t)NB, r);
        };
        auto look = [&](int s) {  // first look at the arrival of step s (clamped to the last one; every thread loads, tid >= KB * 128 for nothing)
            const int i = nb - 1 - (s < npl ? s : (npl > 0 ? npl - 1 : 0));
            const int e = tid < KB * NB ? tid : 0;
            return __hip_atomic_load((const gf64 *)&xb[((size_t)i * KB + e / NB) * NB + (e % NB)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        double spec = fillv, spec2 = fillv;
        auto step = [&](int s, const v2d(&r)[2][8]) -> bool {
            double(*xs)[NB] = sh.xs[s & 1];
            if (s < npl) {
                const int i = nb - 1 - s;
                const int backoff = (i - j - 3) < 24 ? (i - j - 3) : 24;  // x 8 x 64 cycles (0.2 us) per block of distance, at most 5 us
                await_block<KB>(sh, xs, xb, i, spec, backoff, abortw, epoch, spin_ticks, status, 0x700);
                if (stamps && tid == 0 && s == npl - 1) stamps[256 * 5 + j * 4 + 0] = (long long)wall_clock64();
            } else if (q == 0) {
#pragma unroll
                for (int l = 0; l < KB; ++l) *(v2d *)&xs[l][col] = (v2d){w[0][l], w[1][l]};
            }
            // the first look at the arrival after the next, BEFORE the next fetch is issued: a poll is a vector load and returns in
            // order, behind every fetch issued before it -- this one sits behind the tiles of the next two steps only, which the step
            // that uses its value needs anyway
            spec = spec2;
            spec2 = look(s + 2);
            if (s + 2 >= npl) spec2 = fillv;
            // the only barrier of a step: xs is double-buffered by step parity, and a wave passes the NEXT step's barrier only after its
            // reads of this step, so the buffer is free again when step s + 2 writes it
            lds_barrier();
            if (!sh.ok) return false;  // gave up: the abort word is set, the host re-runs the solve with backsolve_blocked
            double dsum[2][KB];
            tile_dot<KB>(xs, r, dsum);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int l = 0; l < KB; ++l) w[cc][l] = s == npl ? dsum[cc][l] : w[cc][l] - dsum[cc][l];
            return true;
        };
        v2d t0[2][8], t1[2][8], t2[2][8];
        // the walk in groups of three steps (one per buffer), phantom steps in front so that the last group is complete
        const int o = (3 - nst % 3) % 3, G = (nst + o) / 3;
        if (0 - o >= 0) fetch(0 - o, t0);
        if (1 - o >= 0) fetch(1 - o, t1);
        fetch(2 - o, t2);
        spec = look(0);
        if (npl == 0) spec = fillv;
        spec2 = look(1);
        if (npl <= 1) spec2 = fillv;
        int g = 0;
        if (G >= 2) {  // first group: may hold the phantoms
            const int s = -o;
            if (s >= 0 && !step(s, t0)) return;
            fetch(s + 3, t0);
            if (s + 1 >= 0 && !step(s + 1, t1)) return;
            fetch(s + 4, t1);
            if (!step(s + 2, t2)) return;
            fetch(s + 5, t2);
            g = 1;
        }
        for (; g < G - 1; ++g) {  // steady
            const int s = 3 * g - o;
            if (!step(s, t0)) return;
            fetch(s + 3, t0);
            if (!step(s + 1, t1)) return;
            fetch(s + 4, t1);
            if (!step(s + 2, t2)) return;
            fetch(s + 5, t2);
        }
        {  // last group: nothing fetched any more
            const int s = 3 * (G - 1) - o;
            if (s >= 0 && !step(s, t0)) return;
            if (s + 1 >= 0 && !step(s + 1, t1)) return;
            if (!step(s + 2, t2)) return;
        }
        if (stamps && tid == 0) stamps[256 * 5 + j * 4 + 1] = (long long)wall_clock64();
        if (q == 0) {
#pragma unroll
            for (int l = 0; l < KB; ++l) publish(&zb[((size_t)j * KB + l) * NB + col], (v2d){w[0][l], w[1][l]});
        }
        return;
    }
    // ---- finisher
    v2d tm2[2][8], tm1[2][8];
    if (npm == 2) load_tile(Mbuf + ((size_t)j * 2 + 1) * NB * NB, NB, tm2);
    if (npm >= 1) load_tile(Mbuf + (size_t)j * 2 * NB * NB, NB, tm1);
    double acc[2][KB];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int l = 0; l < KB; ++l) acc[cc][l] = 0.0;
    if (npm == 2) {
        await_block<KB>(sh, sh.xs[0], xb, j + 2, fillv, 1, abortw, epoch, spin_ticks, status, 0x700);
        lds_barrier();
        if (!sh.ok) return;
        if (stamps && tid == 0) stamps[256 * 5 + j * 4 + 3] = (long long)wall_clock64();
        tile_dot<KB>(sh.xs[0], tm2, acc);
    }
    // z_j: the lanes that publish read their own entries (16-byte loads past the L1)
    v2d zj[KB];
    if (q == 0) {
        unsigned spins = 0;
        unsigned long long t0 = 0;
#pragma unroll
        for (int l = 0; l < KB; ++l) {
            const double *src = &zb[((size_t)j * KB + l) * NB + col];
            while (true) {
                v2d v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
                if (__double_as_longlong(v[0]) != -1ll && __double_as_longlong(v[1]) != -1ll) {
                    zj[l] = v;
                    break;
                }
                if ((++spins & 63u) == 0u) {
                    if (__hip_atomic_load((const gu32 *)abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        sh.ok = 0;
                        break;
                    }
                    const unsigned long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > spin_ticks) {
                        __hip_atomic_store((gu32 *)abortw, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *(__attribute__((address_space(1))) int *)status = 0x600 + (j & 0xff);
                        sh.ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!sh.ok) break;
        }
    }
    if (stamps && tid == 0) stamps[256 * 5 + j * 4 + 2] = (long long)wall_clock64();
    double last[2][KB];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int l = 0; l < KB; ++l) last[cc][l] = 0.0;
    if (npm >= 1) {
        if (stamps && tid == 0) stamps[256 * 4 + j] = (long long)wall_clock64();
        await_block<KB>(sh, sh.xs[1], xb, j + 1, fillv, 0, abortw, epoch, spin_ticks, status, 0x700);
        if (stamps && tid == 0) stamps[j * 4 + 0] = (long long)wall_clock64();
        lds_barrier();
        if (!sh.ok) return;
        if (stamps && tid == 0) stamps[j * 4 + 1] = (long long)wall_clock64();
        tile_dot<KB>(sh.xs[1], tm1, last);
    } else {
        lds_barrier();  // the z_j wait may have given up
        if (!sh.ok) return;
    }
    if ((fault & 1) && j == nb / 2) return;  // test hook: this block's solution is never published
    if (stamps && tid == 0) stamps[j * 4 + 2] = (long long)wall_clock64();
    if (q == 0) {
#pragma unroll
        for (int l = 0; l < KB; ++l) {
            const v2d v = (v2d){(zj[l][0] - acc[0][l]) - last[0][l], (zj[l][1] - acc[1][l]) - last[1][l]};
            publish(&xb[((size_t)j * KB + l) * NB + col], v);
            *(v2d *)&Y[(int64_t)(k0 + l) * ldy + (int64_t)j * NB + col] = v;  // the result
        }
    }
}

}  // namespace bsolve

int backsolve_blocked(mrbf_ctx *ctx, int64_t npad, const double *L, int64_t lda, const double *linv_all, double *Y, int64_t ldy, int k);

// Same contract as backsolve_blocked (chol_blocked.hip): Y (npad x k, column-major, ld ldy) is overwritten by the solution.
// *status (device int, zeroed by the caller) receives a non-zero code when a workgroup gave up on a dependency; Y is then
// partly overwritten and the caller must restore the right-hand sides and use backsolve_blocked.
int backsolve_persistent(mrbf_ctx *ctx, int64_t npad, const double *L, int64_t lda, const double *linv_all, double *Y, int64_t ldy, int k,
                         int *status) {
    using namespace bsolve;
    const int nb = (int)(npad / NB);
    static const int force_old = mrbf_env("MRBF_BACKSOLVE_LAUNCHES") ? atoi(mrbf_env("MRBF_BACKSOLVE_LAUNCHES")) : 0;
    // two 512-thread workgroups per block column, one per CU; beyond the CU count the later positions of the chain start as the first
    // ones leave (dispatch order is chain order), up to twice the CU count
    const int ncu = ctx->ncu;
    if (force_old || nb < 3 || nb > ncu || (lda & 1) || (reinterpret_cast<uintptr_t>(L) & 15))
        return backsolve_blocked(ctx, npad, L, lda, linv_all, Y, ldy, k);
    unsigned *flags;
    MRBF_TRY(get_buf(ctx, S_BSOLVE_FLAGS, (size_t)(nb + 1) * 32, &flags));  // one 128-byte line per block + the abort word
    const unsigned long long spin_ticks = (unsigned long long)std::max(1, ctx->spin_ms) * 100000ull;  // wall_clock64: 100 MHz
    const int fault = (ctx->debug_fault & 2) ? 1 : 0;
    double *xb, *Mbuf;
    MRBF_TRY(get_buf(ctx, S_BSOLVE_X, (size_t)2 * nb * 4 * NB, &xb));  // xb | zb
    MRBF_TRY(get_buf(ctx, S_BSOLVE_M, (size_t)nb * 2 * NB * NB, &Mbuf));
    hipLaunchKernelGGL(premul_kernel, dim3((unsigned)(8 * (nb - 1))), dim3(256), 0, ctx->stream, L, lda, linv_all, Mbuf, nb, flags, (nb + 1) * 32,
                       reinterpret_cast<unsigned long long *>(xb), 2 * nb * 4 * NB);
    // debug (MRBF_BSOLVE_STAMPS=1): per block, wall_clock64 when x_{j+1} was seen, after the barrier, before the publication
    static const bool want_stamps = mrbf_env("MRBF_BSOLVE_STAMPS") && atoi(mrbf_env("MRBF_BSOLVE_STAMPS")) != 0;
    long long *stamps = nullptr;
    if (want_stamps) {
        static long long *dbg = nullptr;  // debug only: one allocation for the largest case, never freed
        if (!dbg) MRBF_HIP(ctx, hipMalloc(&dbg, (size_t)256 * 9 * sizeof(long long)));
        stamps = dbg;
    }
    unsigned epoch = 0;
    for (int k0 = 0; k0 < k; k0 += 4) {
        const int kb = std::min(4, k - k0);
        ++epoch;
        if (k0 > 0) MRBF_HIP(ctx, hipMemsetAsync(xb, 0xff, (size_t)2 * nb * 4 * NB * sizeof(double), ctx->stream));  // "not published yet" (first pass: premul_kernel)
#define MRBF_BSP(KBV)                                                                                                          \
    hipLaunchKernelGGL((backsolve_persistent_kernel<KBV>), dim3((unsigned)(2 * nb)), dim3(NTHR), 0, ctx->stream, L, lda, linv_all, Mbuf, Y, ldy, k0, nb, \
                       flags, epoch, spin_ticks, status, fault, xb, xb + (size_t)nb * 4 * NB, stamps)
        if (kb == 1) MRBF_BSP(1); else if (kb == 2) MRBF_BSP(2); else if (kb == 3) MRBF_BSP(3); else MRBF_BSP(4);
#undef MRBF_BSP
    }
    MRBF_HIP(ctx, hipGetLastError());
    if (want_stamps) {
        std::vector<long long> h((size_t)256 * 9);
        MRBF_HIP(ctx, hipMemcpyAsync(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double seen = 0, bar = 0, dot = 0, hop = 0, late = 0;
        int cnt = 0;
        for (int j = 1; j + 2 < nb; ++j) {  // block j saw x_{j+1}, which block j+1 published after its own third stamp
            seen += (double)(h[j * 4 + 0] - h[(j + 1) * 4 + 2]);
            bar += (double)(h[j * 4 + 1] - h[j * 4 + 0]);
            dot += (double)(h[j * 4 + 2] - h[j * 4 + 1]);
            hop += (double)(h[j * 4 + 2] - h[(j + 1) * 4 + 2]);
            late += (double)(h[256 * 4 + j] - h[(j + 1) * 4 + 2]);
            ++cnt;
        }
        double a3 = 0, zt = 0, zs = 0, x2 = 0;
        int c2 = 0;
        for (int j = 2; j + 4 < nb; ++j) {
            const long long *S = &h[256 * 5 + j * 4];
            a3 += (double)(S[0] - h[(j + 3) * 4 + 2]);   // streamer j sees x_{j+3} after its publication
            zt += (double)(S[1] - S[0]);                 // ... until it publishes z_j
            zs += (double)(S[2] - S[1]);                 // finisher j has z_j after that
            x2 += (double)(S[3] - h[(j + 2) * 4 + 2]);   // finisher j is through its barrier for x_{j+2} after that block's publication
            ++c2;
        }
        fprintf(stderr, "[backsolve stamps] streamer: x_{j+3} seen %.2f us after its publication, z_j published %.2f us later, finisher has z_j %.2f us "
                "later; finisher past x_{j+2} %.2f us after that block's publication\n", a3 / c2 / 100.0, zt / c2 / 100.0, zs / c2 / 100.0, x2 / c2 / 100.0);
        fprintf(stderr, "[backsolve stamps] nb %d: publish -> seen %.2f us, seen -> barrier %.2f us, barrier -> publish %.2f us, step %.2f us; "
                "the finisher starts polling %.2f us after the publication\n", nb, seen / cnt / 100.0, bar / cnt / 100.0, dot / cnt / 100.0,
                hop / cnt / 100.0, late / cnt / 100.0);
    }
    return 0;
}

}  // namespace mrbf
