// Backward substitution  L' x = y  for k right-hand sides as ONE persistent launch.
//
// chol_blocked.hip's backsolve_blocked issues one launch per block row (64 dependent launches x 11 us at n = 8192:
// profiles/r01_d_kernel_stats.csv).  Here workgroup j owns block j of the unknowns: it subtracts L(i,j)' x_i for
// i = nb-1 .. j+1 as the x_i are published, then finishes x_j = inv(L_jj)' y_j (stored block inverses) and publishes it.
// Only the last two products of a workgroup are on the critical path (x_{j+1} has just been published); their tiles,
// L(j+1,j) and inv(L_jj), are fetched into REGISTERS before the wait, each thread holding 32 consecutive rows of one
// column, so after the flag the product is 32 x k FMAs per thread and one LDS reduction -- a step of the chain costs a
// flag hand-off plus ~1 us instead of a kernel boundary plus two LDS-staged tile passes.
// Hand-off (MI355X: XCD L2s are not coherent): the data is its own flag.  x_j goes to an exchange buffer that the host
// filled with an all-ones pattern (a NaN no arithmetic produces); the producer's threads store their entries write-through
// (sc1) and are done -- no drain, no barrier, no flag store; each consumer thread polls ITS entry with sc1 loads until it is
// no longer the pattern.  A step of the chain loses the producer's ~1.5 us store drain and one ~1 us flag round trip
// (round 2: 6.2 -> ~4 us per step).
#include "common.hpp"

namespace mrbf {

namespace bsolve {

constexpr int NB = 128, NTHR = 512, RPT = 32;  // rows per thread: thread (c = t & 127, g = t >> 7) holds rows 32g .. 32g+31 of column c
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) double gf64;
typedef __attribute__((address_space(1))) v2d gv2d;

template <int KB>
struct Shared {
    double xs[KB][NB];       // the published x_i (or y_j for the final product)
    double y[KB][NB];        // this workgroup's running right-hand side
    double part[KB][4][NB];  // partial sums of the 4 row groups
    int ok;
};

__device__ __forceinline__ void load_tile(const double *__restrict__ T, int64_t ldt, v2d (&r)[RPT / 2]) {
    const int c = threadIdx.x & 127, g = threadIdx.x >> 7;
    const double *p = T + (int64_t)c * ldt + RPT * g;
#pragma unroll
    for (int u = 0; u < RPT / 2; ++u) r[u] = *(const gv2d *)(p + 2 * u);
}

// y -= T' xs  (all threads; T in registers)
template <int KB>
__device__ __forceinline__ void tile_product(Shared<KB> &sh, const v2d (&r)[RPT / 2]) {
    const int c = threadIdx.x & 127, g = threadIdx.x >> 7;
    double acc[KB];
#pragma unroll
    for (int l = 0; l < KB; ++l) acc[l] = 0.0;
#pragma unroll
    for (int u = 0; u < RPT / 2; ++u) {
#pragma unroll
        for (int l = 0; l < KB; ++l) {
            acc[l] = fma(r[u][0], sh.xs[l][RPT * g + 2 * u], acc[l]);
            acc[l] = fma(r[u][1], sh.xs[l][RPT * g + 2 * u + 1], acc[l]);
        }
    }
#pragma unroll
    for (int l = 0; l < KB; ++l) sh.part[l][g][c] = acc[l];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int l = 0; l < KB; ++l) sh.y[l][c] -= (sh.part[l][0][c] + sh.part[l][1][c]) + (sh.part[l][2][c] + sh.part[l][3][c]);
    }
}

template <int KB>
__global__ __launch_bounds__(NTHR, 1) void backsolve_persistent_kernel(const double *__restrict__ L, int64_t lda, const double *__restrict__ linv_all,
                                                                        double *__restrict__ Y, int64_t ldy, int k0, int nb, unsigned *flags,
                                                                        unsigned epoch, unsigned long long spin_ticks, int *status, int fault,
                                                                        double *__restrict__ xb) {
    __shared__ Shared<KB> sh;
    // block nb-1 (which depends on nothing) is dispatched first, block 0 last: every workgroup only waits for workgroups that were
    // dispatched BEFORE it, so the launch drains at any residency (other kernels holding CUs, several contexts on one GPU)
    const int j = nb - 1 - (int)blockIdx.x, tid = threadIdx.x;
    const int c = tid & 127, g = tid >> 7;
    unsigned *abortw = flags + (size_t)nb * 32;
    // y_j
    if (tid < KB * NB) sh.y[tid / NB][tid % NB] = *(const gf64 *)&Y[(int64_t)(k0 + tid / NB) * ldy + (int64_t)j * NB + (tid % NB)];
    // tiles of the critical path first, into registers
    v2d tnext[RPT / 2], tinv[RPT / 2];
    if (j + 1 < nb) load_tile(L + (int64_t)(j + 1) * NB + (int64_t)j * NB * lda, lda, tnext);
    load_tile(linv_all + (size_t)j * NB * NB, NB, tinv);
    if (tid == 0) sh.ok = 1;
    __syncthreads();
    for (int i = nb - 1; i > j; --i) {
        v2d ts[RPT / 2];
        if (i > j + 1) load_tile(L + (int64_t)i * NB + (int64_t)j * NB * lda, lda, ts);  // in flight under the wait
        // x_i: every thread of the first KB * 128 polls its own entry of the exchange buffer (write-through by the producer, read
        // past the L1) until it is no longer the fill pattern
        if (tid < KB * NB) {
            const gf64 *src = (const gf64 *)&xb[((size_t)i * KB + tid / NB) * NB + (tid % NB)];
            double v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                        *(__attribute__((address_space(1))) int *)status = 0x700 + (i & 0xff);  // read back with the factorisation's flags
                        sh.ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sh.xs[tid / NB][tid % NB] = v;
        }
        __syncthreads();
        if (!sh.ok) return;  // gave up: the abort word is set, the host re-runs the solve with backsolve_blocked
        if (i > j + 1)
            tile_product<KB>(sh, ts);
        else
            tile_product<KB>(sh, tnext);
        __syncthreads();
    }
    if (fault && j == nb / 2) return;  // test hook: this block's solution is never published
    // x_j = inv(L_jj)' y_j
    {
        double acc[KB];
#pragma unroll
        for (int l = 0; l < KB; ++l) acc[l] = 0.0;
#pragma unroll
        for (int u = 0; u < RPT / 2; ++u) {
#pragma unroll
            for (int l = 0; l < KB; ++l) {
                acc[l] = fma(tinv[u][0], sh.y[l][RPT * g + 2 * u], acc[l]);
                acc[l] = fma(tinv[u][1], sh.y[l][RPT * g + 2 * u + 1], acc[l]);
            }
        }
#pragma unroll
        for (int l = 0; l < KB; ++l) sh.part[l][g][c] = acc[l];
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int l = 0; l < KB; ++l) {
                double s = (sh.part[l][0][c] + sh.part[l][1][c]) + (sh.part[l][2][c] + sh.part[l][3][c]);
                if (__double_as_longlong(s) == -1ll) s = __longlong_as_double(0x7ff8000000000000ll);  // never publish the fill pattern
                __hip_atomic_store((gf64 *)&xb[((size_t)j * KB + l) * NB + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the hand-off
                Y[(int64_t)(k0 + l) * ldy + (int64_t)j * NB + c] = s;                                                         // the result
            }
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
    static const int force_old = getenv("MRBF_BACKSOLVE_LAUNCHES") ? atoi(getenv("MRBF_BACKSOLVE_LAUNCHES")) : 0;
    // every workgroup must be resident (one 512-thread workgroup per block column): fall back beyond the CU count
    const int ncu = ctx->ncu;
    if (force_old || nb < 3 || nb > ncu - 8 || (lda & 1) || (reinterpret_cast<uintptr_t>(L) & 15))
        return backsolve_blocked(ctx, npad, L, lda, linv_all, Y, ldy, k);
    unsigned *flags;
    MRBF_TRY(get_buf(ctx, S_BSOLVE_FLAGS, (size_t)(nb + 1) * 32, &flags));  // one 128-byte line per block + the abort word
    MRBF_HIP(ctx, hipMemsetAsync(flags, 0, (size_t)(nb + 1) * 32 * sizeof(unsigned), ctx->stream));
    const unsigned long long spin_ticks = (unsigned long long)std::max(1, ctx->spin_ms) * 100000ull;  // wall_clock64: 100 MHz
    const int fault = (ctx->debug_fault & 2) ? 1 : 0;
    double *xb;
    MRBF_TRY(get_buf(ctx, S_BSOLVE_X, (size_t)nb * 4 * NB, &xb));
    unsigned epoch = 0;
    for (int k0 = 0; k0 < k; k0 += 4) {
        const int kb = std::min(4, k - k0);
        ++epoch;
        MRBF_HIP(ctx, hipMemsetAsync(xb, 0xff, (size_t)nb * 4 * NB * sizeof(double), ctx->stream));  // "not published yet"
#define MRBF_BSP(KBV)                                                                                                          \
    hipLaunchKernelGGL((backsolve_persistent_kernel<KBV>), dim3((unsigned)nb), dim3(NTHR), 0, ctx->stream, L, lda, linv_all, Y, ldy, k0, nb, \
                       flags, epoch, spin_ticks, status, fault, xb)
        if (kb == 1) MRBF_BSP(1); else if (kb == 2) MRBF_BSP(2); else if (kb == 3) MRBF_BSP(3); else MRBF_BSP(4);
#undef MRBF_BSP
    }
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf
