// Persistent ("one launch") blocked Cholesky on the fp64 matrix cores.
//
// chol_blocked.hip drives the factorisation from the host: per 128-column step a diagonal kernel, a panel kernel, a
// block-column update and (per window) bulk updates on three streams.  At n = 8192 its run time IS the dependent
// chain of those launches (64 x ~128 us: profiles/r01_c_kernel_stats.csv) -- every hop pays a kernel boundary and a
// cross-stream event (~10 us each).  Here the whole factorisation is ONE launch of resident workgroups that pull
// 128 x 128 tile jobs from two in-order queues and hand tiles to each other through flags in memory (hop ~2-3 us):
//
//   P(c)      diagonal tile: subtract the in-window panels, Cholesky + inverse of the factor (chol_diag_core.hpp)
//   T(i,c)    panel tile: subtract the in-window panels, then  L(i,c) = X * inv(L_cc)'   (a GEMM with the inverse)
//   U(i,c,w)  bulk update of tile (i,c), c beyond window w, with the 4 panels of window w  (K = 512, read-modify-write)
//
// Windows are 4 block columns: a tile of block column c receives nb(c) = floor((c-1)/4) bulk updates (right-looking,
// rank 512) and its last 1..4 panels left-looking inside its T / P job (so the panel that has just been finished never
// has to pass through a bulk job before the next diagonal block can start); the bulk of the flops streams the trailing
// matrix once per 512 columns.  Panel jobs are claimed in a topological order (column by column, P first, rows
// ascending) and only wait for jobs claimed before them.  Bulk jobs sit in one queue per window (block columns
// ascending); a workgroup takes, among the heads of all queues, the READY job with the smallest block column (the one
// the diagonal chain needs first); a bulk job therefore (almost) never waits.  Every spin is
// bounded and gives up through the abort word; the launch drains for any number of resident workgroups.
//
// Hand-off between workgroups (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed): every store of a
// tile another workgroup will read is a write-through (sc1) store, every storing wave drains (s_waitcnt vmcnt(0)),
// the workgroup meets at a barrier, ONE lane publishes the flag / counter with an agent-scope atomic; the consumer
// polls that word with relaxed agent-scope loads, then ONE agent-scope acquire + s_waitcnt vmcnt(0) + barrier, then
// plain loads.
#include "chol_diag_core.hpp"
#include "common.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

namespace mega {

constexpr int NB = 128, BK = 16, LDS_LD = NB + 16;  // (2*LD) % 64 == 32: k and k+1 rows of a chunk hit disjoint banks
constexpr int WIN = 4;                              // block columns per window

enum { JOB_U = 0, JOB_T = 1, JOB_P = 2 };
struct Job {
    short kind, i, c, w;
};
// control words (each on its own 128-byte line)
enum { CTL_QP = 0, CTL_ABORT = 32, CTL_TIMEOUT = 64, CTL_PCOLS = 96, CTL_WORDS = 128 };
constexpr int QSTRIDE = 32;  // one bulk-queue head per 128-byte line
__host__ __device__ inline int nbulk_updates(int c) { return c < 1 ? 0 : (c - 1) / WIN; }

struct Args {
    double *A;
    int64_t lda;
    int NT, MT;  // block columns, block rows (MT >= NT: extra rows ride along as panel rows)
    double *linv;  // NT x (128 x 128) inverses of the diagonal blocks of L
    unsigned *tdone;  // [MT][NT]: tile (i,c) holds its final L entries (i == c: L_cc and its inverse)
    unsigned *ucnt;   // [MT][NT]: bulk updates applied to tile (i,c)
    unsigned *ctl;
    const Job *pjobs;
    int npanel;
    const Job *bjobs;
    int nbulk;
    const int *wq_start;  // [nwin + 1] first job of each window's queue in bjobs
    unsigned *wq_head;    // [nwin] x QSTRIDE claimed jobs per window
    unsigned *quiet;      // [512] x QSTRIDE per-CU count of chain-critical jobs in flight: the CU's other workgroup pauses
    int nwin;
    int *info;
    int ndedicated;  // workgroups [0, ndedicated) serve the panel queue only
    int look;        // general workgroups take a panel job of block column c once c < (finished diagonal blocks) + look
    unsigned spin_limit;
    int use_quiet;
    unsigned long long *trace;  // diagnostic launches only: 8 time stamps (10 ns units) per chain job (P(c), T(c+1,c))
};

struct Shared {
    union {
        double gemm[2 * BK * LDS_LD];
        diagcore::DiagV4Shared diag;
    } u;
    int ok;
    int jkind, jidx;
    int wlo;  // first window whose queue still holds jobs (monotone, per workgroup)
    int mycu;
};

// every shared word and tile is accessed through explicit global-address-space pointers: global_ instructions, never flat_
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) double gf64;
typedef __attribute__((address_space(1))) v2d gv2d;
__device__ __forceinline__ unsigned ldf(const unsigned *p) {
    return __hip_atomic_load((const gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stf(unsigned *p, unsigned v) { __hip_atomic_store((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned addf(unsigned *p, unsigned v) {
    return __hip_atomic_fetch_add((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store((gf64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// thread 0: spin until *f >= want (bounded); false on abort / timeout
__device__ __forceinline__ bool poll_ge(const unsigned *f, unsigned want, const Args &a, unsigned code) {
    unsigned spins = 0;
    while (ldf(f) < want) {
        if (ldf(a.ctl + CTL_ABORT)) return false;
        if (++spins > a.spin_limit) {
            stf(a.ctl + CTL_TIMEOUT, code);
            stf(a.ctl + CTL_ABORT, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}

// Workgroup-wide wait for up to three flags, then the acquire that makes the published tiles loadable.
__device__ __forceinline__ bool wg_wait(Shared &sh, const Args &a, const unsigned *f0, unsigned w0, const unsigned *f1, unsigned w1,
                                        const unsigned *f2, unsigned w2, unsigned code) {
    if (threadIdx.x == 0) {
        bool ok = true;
        if (f0) ok = poll_ge(f0, w0, a, code);
        if (ok && f1) ok = poll_ge(f1, w1, a, code + 1);
        if (ok && f2) ok = poll_ge(f2, w2, a, code + 2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh.ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = sh.ok != 0;
    __syncthreads();
    return ok;
}

// publish: every storing wave has drained its write-through stores, then one lane sets the word
__device__ __forceinline__ void wg_drain() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc(j-part, i-part) += sum_k Bp(j, k) * Ap(i, k) over K columns (multiple of 16): the LDS-tiled MFMA loop of
// chol_update_kernel<128, *> (16-column chunks staged global -> registers -> LDS, next chunk's loads in flight under
// the current chunk's 64 MFMAs per wave).  Ap / Bp point at row 0 of the 128-row operand tiles, column 0 of the range.
__device__ __forceinline__ void gemm_acc(const double *__restrict__ Ag, int64_t lda, const double *__restrict__ Bg, int64_t ldb, int K,
                                         v4d (&acc)[4][4], double *smem) {
    double *As = smem;
    double *Bs = smem + BK * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ioff = (wave >> 1) * 64, joff = (wave & 1) * 64;
    const int i2 = (tid & 63) * 2, k0 = tid >> 6;
    const double *Ap = Ag + i2 + (int64_t)k0 * lda;
    const double *Bp = Bg + i2 + (int64_t)k0 * ldb;
    v2d ra[4], rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ra[u] = *(const gv2d *)(Ap + (int64_t)(4 * u) * lda);
#pragma unroll
    for (int u = 0; u < 4; ++u) rb[u] = *(const gv2d *)(Bp + (int64_t)(4 * u) * ldb);
    const int nkc = K / BK;
#pragma unroll 1
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) *(v2d *)&As[(k0 + 4 * u) * LDS_LD + i2] = ra[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) *(v2d *)&Bs[(k0 + 4 * u) * LDS_LD + i2] = rb[u];
        __syncthreads();
        if (kc + 1 < nkc) {
            const int64_t ko = (int64_t)(kc + 1) * BK;
#pragma unroll
            for (int u = 0; u < 4; ++u) ra[u] = *(const gv2d *)(Ap + (ko + 4 * u) * lda);
#pragma unroll
            for (int u = 0; u < 4; ++u) rb[u] = *(const gv2d *)(Bp + (ko + 4 * u) * ldb);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = As[(kk * 4 + l4) * LDS_LD + ioff + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = Bs[(kk * 4 + l4) * LDS_LD + joff + j * 16 + l15];
            // D[row = j][col = i]: the lane index (l & 15) runs along i, contiguous in the column-major tile
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[j], av[i], acc[j][i], 0, 0, 0);
        }
    }
    __syncthreads();  // the LDS chunk buffers are free again (the caller may overlay them)
}

__device__ __forceinline__ void zero_acc(v4d (&acc)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// C tile epilogue.  SUB: C = C - acc (else C = acc).  LOWER: entries above the diagonal of the tile stay untouched
// (diagonal tiles).  SC1: write-through stores.  Batches of 16 loads before the first store of a batch (see
// chol_update_kernel: element-wise read-modify-write serialises 64 dependent round trips).
template <bool SUB, bool LOWER, bool SC1>
__device__ __forceinline__ void store_tile(double *__restrict__ C, int64_t ldc, const v4d (&acc)[4][4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ioff = (wave >> 1) * 64, joff = (wave & 1) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double cv[4][4];
        if (SUB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gj = joff + j * 16 + l4 + 4 * r;
#pragma unroll
                for (int i = 0; i < 4; ++i) cv[r][i] = *(const gf64 *)&C[(ioff + i * 16 + l15) + gj * ldc];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gj = joff + j * 16 + l4 + 4 * r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t gi = ioff + i * 16 + l15;
                const double v = SUB ? cv[r][i] - acc[j][i][r] : acc[j][i][r];
                if (!LOWER || gi >= gj) {
                    if (SC1)
                        st_sc1(C + gi + gj * ldc, v);
                    else
                        *(gf64 *)&C[gi + gj * ldc] = v;
                }
            }
        }
    }
}

// ---- jobs ------------------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) bool run_bulk(const Args &a, Shared &sh, const Job jb) {
    const int i = jb.i, c = jb.c, w = jb.w;
    const int pl = WIN * w + WIN - 1;  // last panel of the window: rows finish their panels in order
    if (!wg_wait(sh, a, a.tdone + (size_t)i * a.NT + pl, 1u, a.tdone + (size_t)c * a.NT + pl, 1u, a.ucnt + (size_t)i * a.NT + c,
                 (unsigned)w, 0x100u))
        return false;
    v4d acc[4][4];
    zero_acc(acc);
    const int64_t k0 = (int64_t)WIN * w * NB;
    gemm_acc(a.A + (int64_t)i * NB + k0 * a.lda, a.lda, a.A + (int64_t)c * NB + k0 * a.lda, a.lda, WIN * NB, acc, sh.u.gemm);
    double *C = a.A + (int64_t)i * NB + (int64_t)c * NB * a.lda;
    if (i == c)
        store_tile<true, true, true>(C, a.lda, acc);
    else
        store_tile<true, false, true>(C, a.lda, acc);
    wg_drain();
    if (threadIdx.x == 0) addf(a.ucnt + (size_t)i * a.NT + c, 1u);
    return true;
}

// left-looking part shared by T and P jobs: X = A(i,c) - sum_{p in [4 nb(c), c)} L(i,p) L(c,p)'  (the panels no bulk job applies)
// written back in place (plain stores: only this workgroup reads X again).  Returns false on abort.
__device__ __attribute__((noinline)) bool window_part(const Args &a, Shared &sh, int i, int c) {
    const int wc = nbulk_updates(c), p0 = wc * WIN;
    const unsigned *uc = a.ucnt + (size_t)i * a.NT + c;
    if (p0 == c) return wg_wait(sh, a, uc, (unsigned)wc, nullptr, 0, nullptr, 0, 0x200u);  // c == 0 only
    v4d acc[4][4];
    zero_acc(acc);
    for (int p = p0; p < c; ++p) {
        if (!wg_wait(sh, a, a.tdone + (size_t)i * a.NT + p, 1u, a.tdone + (size_t)c * a.NT + p, 1u, p == p0 ? uc : nullptr, (unsigned)wc,
                     0x210u))
            return false;
        const int64_t k0 = (int64_t)p * NB;
        gemm_acc(a.A + (int64_t)i * NB + k0 * a.lda, a.lda, a.A + (int64_t)c * NB + k0 * a.lda, a.lda, NB, acc, sh.u.gemm);
    }
    double *C = a.A + (int64_t)i * NB + (int64_t)c * NB * a.lda;
    if (i == c)
        store_tile<true, true, false>(C, a.lda, acc);
    else
        store_tile<true, false, false>(C, a.lda, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return true;
}

#define MEGA_STAMP(k)                                                          \
    do {                                                                       \
        if (tr && threadIdx.x == 0) tr[k] = wall_clock64();                    \
    } while (0)

__device__ __attribute__((noinline)) bool run_panel(const Args &a, Shared &sh, const Job jb) {
    const int i = jb.i, c = jb.c;
    unsigned long long *tr = (a.trace && i <= c + 1) ? a.trace + (size_t)(2 * c + (i - c)) * 8 : nullptr;
    MEGA_STAMP(0);
    if (!window_part(a, sh, i, c)) return false;
    MEGA_STAMP(1);
    double *C = a.A + (int64_t)i * NB + (int64_t)c * NB * a.lda;
    double *Linv = a.linv + (size_t)c * NB * NB;
    unsigned *flag = a.tdone + (size_t)i * a.NT + c;
    if (jb.kind == JOB_P) {
        __builtin_amdgcn_s_setprio(3);
        const int bad = diagcore::diag_v4_core<true>(C, a.lda, Linv, sh.u.diag);
        __builtin_amdgcn_s_setprio(1);
        if (bad) {
            if (threadIdx.x == 0) {
                *(__attribute__((address_space(1))) int *)a.info = c * NB + bad;
                stf(a.ctl + CTL_ABORT, 1u);
            }
            return false;
        }
        MEGA_STAMP(2);
        wg_drain();
        if (threadIdx.x == 0) {
            stf(flag, 1u);
            addf(a.ctl + CTL_PCOLS, 1u);
        }
        MEGA_STAMP(3);
        return true;
    }
    // T: L(i,c) = X * inv(L_cc)'
    if (!wg_wait(sh, a, a.tdone + (size_t)c * a.NT + c, 1u, nullptr, 0, nullptr, 0, 0x300u)) return false;
    MEGA_STAMP(2);
    v4d acc[4][4];
    zero_acc(acc);
    gemm_acc(C, a.lda, Linv, NB, NB, acc, sh.u.gemm);  // every wave's loads of X are complete behind the loop's last barrier
    MEGA_STAMP(3);
    store_tile<false, false, true>(C, a.lda, acc);
    wg_drain();
    if (threadIdx.x == 0) stf(flag, 1u);
    MEGA_STAMP(4);
    return true;
}

// wave 0: among the heads of the window queues [wlo, wlo + 64) find the ready job with the smallest block column and
// claim the next job of that queue.  Returns the job index in bjobs, -1 (nothing ready now) or -2 (every queue drained).
__device__ __forceinline__ int pick_bulk(const Args &a, Shared &sh) {
    const int lane = threadIdx.x & 63;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const int wlo = sh.wlo;
        if (wlo >= a.nwin) return -2;
        const int wq = wlo + lane;
        const bool valid = wq < a.nwin;
        unsigned h = 0;
        int cnt = 0, base = 0;
        if (valid) {
            base = a.wq_start[wq];
            cnt = a.wq_start[wq + 1] - base;
            h = ldf(a.wq_head + (size_t)wq * QSTRIDE);
        }
        const bool has = valid && h < (unsigned)cnt;
        // advance wlo past drained queues (lane 0's queue)
        const unsigned long long has_mask = __ballot(has);
        if (has_mask == 0ull) {
            if (wlo + 64 >= a.nwin) return -2;
            if (lane == 0) sh.wlo = wlo + 64;
            continue;
        }
        const int first = __ffsll((long long)has_mask) - 1;
        if (first > 0 && lane == 0) sh.wlo = wlo + first;
        unsigned key = 0xffffffffu;
        if (has) {
            const Job jb = a.bjobs[base + (int)h];
            const int pl = WIN * jb.w + WIN - 1;
            if (ldf(a.tdone + (size_t)jb.i * a.NT + pl) && ldf(a.tdone + (size_t)jb.c * a.NT + pl) &&
                ldf(a.ucnt + (size_t)jb.i * a.NT + jb.c) >= (unsigned)jb.w)
                key = ((unsigned)jb.c << 16) | (unsigned)lane;
        }
        unsigned best = key;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)best, off);
            best = o < best ? o : best;
        }
        if (best == 0xffffffffu) return -1;
        const int win_lane = (int)(best & 0xffffu);
        // claim by fetch-add (a compare-and-swap on the inspected head serialises all idle workgroups on one word).  The
        // job actually received may lie behind the inspected one and not be ready yet: run_bulk waits for it (bounded);
        // its predecessors are ready or claimed jobs of lower queues, which the same priority rule hands out first.
        int got = -1;
        if (lane == win_lane) {
            const unsigned g = addf(a.wq_head + (size_t)wq * QSTRIDE, 1u);
            if (g < (unsigned)cnt) got = base + (int)g;
        }
        got = __shfl(got, win_lane);
        if (got >= 0) return got;
    }
    return -1;
}

__global__ __launch_bounds__(256, 2) void potrf_mega_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const bool dedicated = (int)blockIdx.x < a.ndedicated;
    if (dedicated) __builtin_amdgcn_s_setprio(2);
    if (threadIdx.x == 0) {
        sh.wlo = 0;
        // (XCC, shader engine, CU) of this workgroup: a 512-workgroup launch puts exactly two workgroups on each of the 256 CUs
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        sh.mycu = (int)(((xcc & 7u) << 6) | (((hw >> 13) & 3u) << 4) | ((hw >> 8) & 15u));
    }
    __syncthreads();
    unsigned *myquiet = a.quiet + (size_t)sh.mycu * QSTRIDE;
    unsigned idle = 0;
    int nidle = 0;
    while (true) {
        if (threadIdx.x < 64) {
            int kind = -1, idx = 0;  // -1 idle, -2 exit
            bool panel_left = false;
            if (ldf(a.ctl + CTL_ABORT)) {
                kind = -2;
            } else if (!dedicated && a.use_quiet && ldf(myquiet) != 0) {
                kind = -1;  // the CU's other workgroup runs a chain-critical job: leave it the SIMDs
                idle = 0;
            } else {
                const unsigned ph = ldf(a.ctl + CTL_QP);
                panel_left = ph < (unsigned)a.npanel;
                bool want_panel = false;
                if (panel_left) {
                    if (dedicated || a.ndedicated == 0)
                        want_panel = true;
                    else
                        want_panel = a.pjobs[ph].c < (int)ldf(a.ctl + CTL_PCOLS) + a.look;
                }
                if (want_panel) {
                    unsigned got = 0;
                    if (threadIdx.x == 0) got = addf(a.ctl + CTL_QP, 1u);
                    got = (unsigned)__shfl((int)got, 0);
                    if (got < (unsigned)a.npanel) {
                        kind = 1;
                        idx = (int)got;
                    }
                } else if (!dedicated) {
                    const int r = pick_bulk(a, sh);
                    if (r >= 0) {
                        kind = 0;
                        idx = r;
                    } else if (r == -2 && !panel_left) {
                        kind = -2;  // every queue drained
                    }
                } else {
                    kind = -2;  // dedicated workgroup, panel queue drained
                }
                if (kind == -1 && ++idle > a.spin_limit) {
                    if (threadIdx.x == 0) {
                        stf(a.ctl + CTL_TIMEOUT, 0x400u);
                        stf(a.ctl + CTL_ABORT, 1u);
                    }
                    kind = -2;
                }
            }
            if (threadIdx.x == 0) {
                sh.jkind = kind;
                sh.jidx = idx;
            }
        }
        __syncthreads();
        const int kind = sh.jkind, idx = sh.jidx;
        __syncthreads();
        if (kind == -2) break;
        if (kind == -1) {
            // back off: an idle workgroup's scan costs ~5 wave loads; 400 of them polling flat out slow everyone's memory traffic
            ++nidle;
            const int reps = nidle < 8 ? 1 : (nidle < 32 ? 4 : 16);
            for (int r = 0; r < reps; ++r) __builtin_amdgcn_s_sleep(64);
            continue;
        }
        idle = 0;
        nidle = 0;
        bool ok;
        if (kind == 1) {
            const Job jb = a.pjobs[idx];
            const bool critical = a.use_quiet && jb.i <= jb.c + 1;  // the diagonal tile and the tile below it
            if (critical && threadIdx.x == 0) addf(myquiet, 1u);
            if (!dedicated) __builtin_amdgcn_s_setprio(1);
            ok = run_panel(a, sh, jb);
            if (!dedicated) __builtin_amdgcn_s_setprio(0);
            if (critical && threadIdx.x == 0) addf(myquiet, 0xffffffffu);
        } else {
            ok = run_bulk(a, sh, a.bjobs[idx]);
        }
        if (!ok) break;
    }
}

__global__ void mega_status_kernel(const unsigned *ctl, int *info) {
    if (ctl[CTL_TIMEOUT] != 0 && *info == 0) *info = -(int)ctl[CTL_TIMEOUT];
}

}  // namespace mega

// Same contract as potrf_blocked_tall (chol_blocked.hip): on return the leading ncols x ncols block holds L, the rows
// below hold A_below * L^-T, linv_all (optional) the inverses of the diagonal blocks; *dinfo = 0, the 1-based index of
// the first non-positive pivot, or a negative code when the launch gave up on a dependency (never observed; every spin is bounded).
int potrf_mega_tall(mrbf_ctx *ctx, int64_t ncols, int64_t mrows, double *A, int64_t lda, int *dinfo, double *linv_all) {
    using namespace mega;
    if (ncols % NB != 0 || mrows % NB != 0 || mrows < ncols || (lda & 1) || (reinterpret_cast<uintptr_t>(A) & 15))
        return fail(ctx, MRBF_EHIP, "potrf_mega needs 128-padded, 16-byte aligned storage (ncols=%lld mrows=%lld lda=%lld)",
                    (long long)ncols, (long long)mrows, (long long)lda);
    const int NT = (int)(ncols / NB), MT = (int)(mrows / NB);
    if (MT > 32000) return fail(ctx, MRBF_EHIP, "potrf_mega: too many block rows");
    Args a{};
    a.A = A;
    a.lda = lda;
    a.NT = NT;
    a.MT = MT;
    if (linv_all) {
        a.linv = linv_all;
    } else {
        MRBF_TRY(get_buf(ctx, S_CHOL_WS, (size_t)NT * NB * NB, &a.linv));
    }
    // job tables (cached per shape)
    if (ctx->mega_nt != NT || ctx->mega_mt != MT) {
        std::vector<Job> pj, bj;
        for (int c = 0; c < NT; ++c) {
            pj.push_back(Job{JOB_P, (short)c, (short)c, (short)nbulk_updates(c)});
            for (int i = c + 1; i < MT; ++i) pj.push_back(Job{JOB_T, (short)i, (short)c, (short)nbulk_updates(c)});
        }
        std::vector<int> wqs;
        for (int w = 0; nbulk_updates(NT - 1) > w; ++w) {
            wqs.push_back((int)bj.size());
            for (int c = WIN * (w + 1) + 1; c < NT; ++c)
                for (int i = c; i < MT; ++i) bj.push_back(Job{JOB_U, (short)i, (short)c, (short)w});
        }
        wqs.push_back((int)bj.size());
        int *dwq;
        MRBF_TRY(get_buf(ctx, S_MEGA_WQ, wqs.size(), &dwq));
        MRBF_HIP(ctx, hipMemcpyAsync(dwq, wqs.data(), wqs.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        ctx->mega_nwin = (int)wqs.size() - 1;
        Job *dj;
        MRBF_TRY(get_buf(ctx, S_MEGA_JOBS, pj.size() + bj.size() + 1, &dj));
        MRBF_HIP(ctx, hipMemcpyAsync(dj, pj.data(), pj.size() * sizeof(Job), hipMemcpyHostToDevice, ctx->stream));
        if (!bj.empty())
            MRBF_HIP(ctx, hipMemcpyAsync(dj + pj.size(), bj.data(), bj.size() * sizeof(Job), hipMemcpyHostToDevice, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host vectors die here
        ctx->mega_nt = NT;
        ctx->mega_mt = MT;
        ctx->mega_npanel = (int)pj.size();
        ctx->mega_nbulk = (int)bj.size();
    }
    Job *dj;
    MRBF_TRY(get_buf(ctx, S_MEGA_JOBS, (size_t)ctx->mega_npanel + ctx->mega_nbulk + 1, &dj));
    a.pjobs = dj;
    a.npanel = ctx->mega_npanel;
    a.bjobs = dj + ctx->mega_npanel;
    a.nbulk = ctx->mega_nbulk;
    a.nwin = ctx->mega_nwin;
    int *dwq;
    MRBF_TRY(get_buf(ctx, S_MEGA_WQ, (size_t)a.nwin + 1, &dwq));
    a.wq_start = dwq;
    // flags: one block, zeroed before every launch
    const size_t nfl = ((size_t)CTL_WORDS + (size_t)QSTRIDE * (a.nwin + 1) + (size_t)QSTRIDE * 512 + 2 * (size_t)MT * NT + 3) / 4 * 4;
    unsigned *fl;
    MRBF_TRY(get_buf(ctx, S_MEGA_FLAGS, nfl, &fl));
    MRBF_HIP(ctx, hipMemsetAsync(fl, 0, nfl * sizeof(unsigned), ctx->stream));
    MRBF_HIP(ctx, hipMemsetAsync(dinfo, 0, sizeof(int), ctx->stream));
    a.ctl = fl;
    a.wq_head = fl + CTL_WORDS;
    a.quiet = a.wq_head + (size_t)QSTRIDE * (a.nwin + 1);
    a.tdone = a.quiet + (size_t)QSTRIDE * 512;
    a.ucnt = a.tdone + (size_t)MT * NT;
    a.info = dinfo;
    a.ndedicated = ctx->mega_dedicated;
    a.look = ctx->mega_look;
    a.use_quiet = ctx->mega_quiet;
    a.spin_limit = 4000000u;  // x ~0.1-0.3 us per poll: gives up after ~1 s without progress
    const int grid = ctx->mega_grid;
    if (a.ndedicated >= grid) a.ndedicated = grid / 2;
    const char *trace_path = getenv("MRBF_MEGA_TRACE");
    if (trace_path) {
        MRBF_TRY(get_buf(ctx, S_MISC, (size_t)NT * 2 * 8 + 8, &a.trace));
        MRBF_HIP(ctx, hipMemsetAsync(a.trace, 0, ((size_t)NT * 2 * 8 + 8) * sizeof(unsigned long long), ctx->stream));
    }
    hipLaunchKernelGGL(potrf_mega_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    MRBF_HIP(ctx, hipGetLastError());
    if (trace_path) {
        std::vector<unsigned long long> h((size_t)NT * 2 * 8);
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MRBF_HIP(ctx, hipMemcpy(h.data(), a.trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(trace_path, "w")) {
            const unsigned long long t0 = h[0];
            for (int c = 0; c < NT; ++c) {
                fprintf(f, "%d", c);
                for (int k = 0; k < 16; ++k) fprintf(f, " %.2f", h[(size_t)c * 16 + k] ? (double)(h[(size_t)c * 16 + k] - t0) * 0.01 : -1.0);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    // a launch that gave up reports through dinfo as well
    hipLaunchKernelGGL(mega_status_kernel, dim3(1), dim3(1), 0, ctx->stream, (const unsigned *)fl, dinfo);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf
