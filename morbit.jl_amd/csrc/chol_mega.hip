// Persistent ("one launch") blocked Cholesky on the fp64 matrix cores.
//
// chol_blocked.hip drives the factorisation from the host: per 128-column step a diagonal kernel, a panel kernel, a
// block-column update and (per window) bulk updates on three streams.  At n = 8192 its run time IS the dependent
// chain of those launches (64 x ~128 us: profiles/r01_c_kernel_stats.csv) -- every hop pays a kernel boundary and a
// cross-stream event (~10 us each).  Here the whole factorisation is ONE launch of resident workgroups (two per CU) that
// pull 128 x 128 tile jobs from queues and hand tiles to each other through flags in memory (hop ~2-3 us):
//
//   P(c)      diagonal tile: left-looking part over the newest panels, the last one (on chain-bound sizes the last two) folded
//             in 16 columns at a time as S(c, c-1) / S(c, c-2) publish them, then Cholesky + inverse of the factor
//             (chol_diag_core.hpp, diag_v6_core: one barrier per 16-column panel), which itself publishes every finished
//             16-column panel of L_cc and its 16 x 16 leaf inverse
//   S(i,c)    the `srows` tiles right below the diagonal (i = c+1 ..): the panel solve run in step with P(c) -- each
//             published panel is applied to the 128 x 128 tile held in registers (MFMA, accumulator-as-operand), the tile's
//             own 16-column panel published in turn: the next diagonal factorisation starts ~10 us after the previous
//             one ends instead of after two dependent 128^3 GEMMs.  Like P, an S job takes the newest panel (c-1) of its
//             left-looking part 16 columns at a time from the S jobs of the previous block column as THEY publish it: a
//             128^3 update is 14 us of one CU's matrix pipe, and taken in one piece after the previous column had finished
//             it made every S job start ~25 us into P(c) and end ~14 us after it (job log r02, tools/mega_gap.py)
//   T(i,h,c)  the other panel tiles in 64-row halves: left-looking part, then  L(i,c) = X * inv(L_cc)'  (a GEMM with the
//             stored inverse).  Half height keeps a row's column-to-column recurrence faster than the diagonal chain.
//   U(i,c,w)  bulk update of tile (i,c) with the panels of window w  (K = 128 x window length, read-modify-write)
//
// Windows hold `win` block columns (the first one `first`): a tile of block column c receives the windows that were
// closed at least `slack` chain steps before c as bulk jobs (right-looking, rank 512..1024) and takes the newer panels
// left-looking inside its own panel job, so a panel that has just been finished never has to pass through a bulk job
// before the next diagonal blocks can start; the bulk of the flops streams the trailing matrix once per window.
// (`srows`, `nchain` and the chain tiles' slack are chosen by matrix size in potrf_mega_tall.)
// Three pools of workgroups: `nchain` serve the chain queue (P, S jobs, in order; their CU partner pauses), `ndedicated`
// serve the panel queue (T jobs, column by column, rows ascending) and may wait inside a job, all others take the head of
// the panel queue when it can run (diagonal block at most `look` steps away AND its bulk updates in), else a bulk job: bulk
// jobs sit in one queue per window (block columns ascending); among the heads of all queues the READY job with the smallest
// block column (the one the diagonal chain needs first, newer windows trailing by `wbias` columns) is taken.  Every job only
// waits for jobs that were claimed before it in its queue or that another pool is guaranteed to run; every spin is bounded
// and gives up through the abort word (negative *dinfo); the launch drains for any number of resident workgroups.
//
// Hand-off between workgroups (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed): every store of a
// tile another workgroup will read is a write-through (sc1) store, every storing wave drains (s_waitcnt vmcnt(0)),
// the workgroup meets at a barrier, ONE lane publishes the flag / counter with an agent-scope atomic; the consumer
// polls that word with relaxed agent-scope loads, then ONE agent-scope acquire + s_waitcnt vmcnt(0) + barrier, then
// plain loads.
#include <chrono>

#include "chol_diag_core.hpp"
#include "common.hpp"
#include "mega_gemm.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

namespace mega {

constexpr int WIN_DEFAULT = 4;                      // block columns per window (Args::win)

enum { JOB_U = 0, JOB_T = 1, JOB_P = 2, JOB_S = 3, JOB_UH = 4 };  // UH: 64-row half of a bulk update (w + 256 * half)
struct Job {
    short kind, i, c, w;
};
// control words (each on its own 128-byte line)
enum { CTL_QP = 0, CTL_ABORT = 32, CTL_TIMEOUT = 64, CTL_PCOLS = 96, CTL_QC = 128, CTL_TS = 144 /* two 64-bit words: ~(first start), last end (wall_clock64) */, CTL_WORDS = 160 };
constexpr int QSTRIDE = 32;  // one bulk-queue head per 128-byte line
// Window w covers the panels [wstart(w), wstart(w+1)): the first window is shorter (`first` panels) so that bulk work
// exists early in the launch, all others hold `win` panels.
__host__ __device__ inline int wstart(int w, int first, int win) { return w <= 0 ? 0 : first + win * (w - 1); }
// number of windows that reach tile (i,c) through bulk jobs: those closed at least `slack` chain steps before column c; the
// tiles of the chain jobs (block rows c .. c+2) rely on a window's bulk update later than the other panel tiles
__host__ __device__ inline int nbulk_updates(int i, int c, int slack, int slack_chain, int first, int win, int srows = 2) {
    const int t = c - ((i - c <= srows) ? slack_chain : slack);
    return t < first ? 0 : (t - first) / win + 1;
}

// Edge regime: the first `head` and the last block columns (from `tail_c0` on) are chain-bound whatever the matrix size (the machine
// is still filling / already emptying), the middle is throughput-bound.  Streamed rows below the diagonal and the diagonal job's
// stream depth are therefore chosen per block column: the chain-bound values at the edges, the size's own in between.
struct Edge {
    int head, tail_c0, srows_edge, pstream_edge;
    int shalf, sh_head, sh_tail_c0;  // shalf: streamed tiles of the block columns c < sh_head and c >= sh_tail_c0 as two 64-row jobs (see run_stream)
    int tfull1;  // > 0: panel tiles more than tfull1 - 1 block rows below the streamed ones are ONE 128-row job (Job::w = 2), not two halves
    int xhalf;   // the block rows below the square (i >= NT: right-hand sides riding along) hold at most 64 non-zero rows: rows 64 .. 127 of
                 // their tiles are zero and stay zero -- panel and bulk jobs work on the upper half only and publish for both (Job::w = 3 / half code 2)
};
__host__ __device__ inline bool edge_col(int c, const Edge &e) { return c < e.head || c >= e.tail_c0; }
__host__ __device__ inline int srows_at(int c, int srows, const Edge &e) { return edge_col(c, e) ? e.srows_edge : srows; }
__host__ __device__ inline bool shalf_at(int c, int srows, const Edge &e) { return e.shalf && (c < e.sh_head || c >= e.sh_tail_c0) && srows_at(c, srows, e) >= 5; }
__host__ __device__ inline bool tfull_at(int i, int c, int srows, const Edge &e) { return e.tfull1 > 0 && i - c - srows_at(c, srows, e) > e.tfull1 - 1; }
__host__ __device__ inline bool xhalf_at(int i, int nt, const Edge &e) { return e.xhalf && i >= nt; }

struct Args {
    double *A;
    int64_t lda;
    int NT, MT;  // block columns, block rows (MT >= NT: extra rows ride along as panel rows)
    double *linv;  // NT x (128 x 128) inverses of the diagonal blocks of L
    unsigned *tdone;  // [MT][NT]: 2 when tile (i,c) holds its final L entries (i == c: L_cc and its inverse); 64-row halves add 1 each
    unsigned *ucnt;   // [MT][NT]: bulk updates applied to tile (i,c)
    unsigned *ctl;
    const Job *pjobs;
    int npanel;
    const Job *bjobs;
    int nbulk;
    const int *wq_start;  // [2 nwin + 1] first job of each queue in bjobs: queues [0, nwin) the windows' jobs on ordinary tiles, [nwin, 2 nwin) on chain tiles
    unsigned *wq_head;    // [2 nwin] x QSTRIDE claimed jobs per queue
    double *itg;          // NT x 8 x 256: 16 x 16 leaf inverses of every diagonal block (streamed panel solves)
    unsigned *dprog;      // [NT] x QSTRIDE: 16-column panels of diagonal block c that are published
    unsigned *sprog;      // [srows][2][NT] x QSTRIDE: 16-column panels of the 64-row halves of tile (c + 1 + k, c) that are published (sprog_at)
    unsigned *quiet;      // [512] x QSTRIDE per-CU count of chain-critical jobs in flight: the CU's other workgroup pauses
    int nwin;
    int *info;
    const Job *cjobs;  // chain queue: P(c), S(c+1,c), S(c+2,c) column by column
    int nchainjobs;
    int nchain;      // workgroups [0, nchain) serve the chain queue only (their CU partner stays idle)
    int ndedicated;  // workgroups [nchain, nchain + ndedicated) serve the panel queue only
    int look;        // general workgroups take a panel job of block column c once c < (finished diagonal blocks) + look
    unsigned long long spin_ticks;  // wall_clock64 ticks (10 ns) one wait may last without the awaited word changing
    int fault;                      // test hook: the last block row's first panel job of column 0 never publishes its tile
    int use_quiet;
    int xchain;      // chain workgroups on so many XCDs (1: blocks 0, 8, 16, ...); 0: blocks 0 .. nchain-1
    int quiet_tail;  // a chain workgroup's CU partner pauses only while at most this many block columns are left
    int slack, slack_chain, first, win, wbias, srows;
    Edge edge;
    int cboost;       // the chain tiles' window jobs compete as if they lay `cboost` block columns further left
    int head_job1;    // first chain job behind the head regime
    int nreserve;     // workgroups [nchain, nchain + nreserve) of the chain's placement class: general workers until the chain queue
    int reserve_job0;  // reaches job `reserve_job0` (the tail), chain workgroups from then on
    unsigned long long *jlog;   // diagnostic launches only: 8 words per job (meta, claim, 5 stage stamps, end), jlog[0] = count
    int jlog_cap;
    int pstream;                // panels the diagonal job takes in step from streamed producers (1 or 2)
    int panel_dma;              // the panel jobs' triangular solve on the LDS-DMA operand ring (MRBF_MEGA_PANELDMA=0: the register-staged loop)
    int trace_dbg;              // diagnostic launches only: dbg word of the diagonal core (4 = time wave 0, 4 + 8 + 16 w = time wave w)
    unsigned long long *trace;  // diagnostic launches only: 8 time stamps (10 ns units) per chain job (P(c), T(c+1,c))
};

struct Shared {
    union {
        double gemm[4 * 2 * 8 * LDS_LD];  // gemm_acc's ring of 8-column stages of both operands (three used)
        diagcore::DiagV6Shared diag;
    } u;
    int ok;
    int jkind, jidx;
    unsigned long long *jrec;  // this job's log record (or null)
    int wlo;   // first window whose queue still holds jobs (monotone, per workgroup)
    int cwlo;  // the same for the chain tiles' queues
    int mycu;
};

// every shared word and tile is accessed through explicit global-address-space pointers: global_ instructions, never flat_
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) double gf64;
__device__ __forceinline__ unsigned ldf(const unsigned *p) {
    return __hip_atomic_load((const gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stf(unsigned *p, unsigned v) { __hip_atomic_store((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned addf(unsigned *p, unsigned v) {
    return __hip_atomic_fetch_add((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store((gf64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Args lives in memory (the job functions are not inlined): a field read after an atomic store is RE-LOADED, and the reload's
// s_waitcnt vmcnt(0) also waits for that store -- eight write-through stores of a streamed step each waited for the previous one
// to reach memory (~0.35 us apiece, 2.6 us of every 4.5-us step: per-step stamps, r02).  The hot fields are therefore copied into
// scalar registers once per job.
__device__ __forceinline__ int64_t uni64(int64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T *uni_ptr(T *q) {
    return reinterpret_cast<T *>(uni64(reinterpret_cast<int64_t>(q)));
}

// thread 0: spin until *f >= want (bounded); false on abort / timeout
// (the bound is wall-clock time since the awaited word last changed, read every 256 polls: a healthy launch that merely shares the
// chip with other kernels is slow, not stuck)
__device__ __forceinline__ bool poll_ge(const unsigned *f, unsigned want, const Args &a, unsigned code) {
    unsigned spins = 0, last = 0xffffffffu;
    unsigned long long t0 = 0;
    unsigned cur;
    while ((cur = ldf(f)) < want) {
        if (ldf(a.ctl + CTL_ABORT)) return false;
        if ((++spins & 255u) == 0u) {
            const unsigned long long now = wall_clock64();
            if (cur != last || t0 == 0) {
                last = cur;
                t0 = now;
            } else if (now - t0 > a.spin_ticks) {
                stf(a.ctl + CTL_TIMEOUT + 1, (unsigned)(f - a.ctl));  // which word (diagnostics)
                stf(a.ctl + CTL_TIMEOUT + 2, want);
                stf(a.ctl + CTL_TIMEOUT, code);
                stf(a.ctl + CTL_ABORT, 1u);
                return false;
            }
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

// Wait until *f >= want and return the value observed (flags only grow): a consumer that is behind its producer learns how
// far it may run without waiting again -- one poll, one acquire, two barriers for several 16-column steps.  0 on abort.
__device__ __forceinline__ unsigned wg_wait_val(Shared &sh, const Args &a, const unsigned *f, unsigned want, unsigned code) {
    if (threadIdx.x == 0) {
        unsigned v = 0;
        if (poll_ge(f, want, a, code)) v = ldf(f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh.ok = (int)v;
    }
    __syncthreads();
    const unsigned v = (unsigned)sh.ok;
    __syncthreads();
    return v;
}

// Same for two producers that are followed in step: waits until both words are >= want, returns the smaller of the two values.
__device__ __forceinline__ unsigned wg_wait_val2(Shared &sh, const Args &a, const unsigned *f0, const unsigned *f1, unsigned want, unsigned code) {
    if (threadIdx.x == 0) {
        unsigned v = 0;
        if (poll_ge(f0, want, a, code) && poll_ge(f1, want, a, code + 1)) {
            const unsigned v0 = ldf(f0), v1 = ldf(f1);
            v = v0 < v1 ? v0 : v1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh.ok = (int)v;
    }
    __syncthreads();
    const unsigned v = (unsigned)sh.ok;
    __syncthreads();
    return v;
}

// Same for up to four producers (null pointers are skipped): the smallest of the values observed, 0 on abort.
__device__ __forceinline__ unsigned wg_wait_val4(Shared &sh, const Args &a, const unsigned *f0, const unsigned *f1, const unsigned *f2, const unsigned *f3,
                                                 unsigned want, unsigned code) {
    if (threadIdx.x == 0) {
        const unsigned *f[4] = {f0, f1, f2, f3};
        unsigned v = 0xffffffffu;
        bool ok = true;
        for (int t = 0; t < 4 && ok; ++t)
            if (f[t]) ok = poll_ge(f[t], want, a, code + t);
        if (ok)
            for (int t = 0; t < 4; ++t)
                if (f[t]) {
                    const unsigned x = ldf(f[t]);
                    v = x < v ? x : v;
                }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh.ok = ok ? (int)v : 0;
    }
    __syncthreads();
    const unsigned v = (unsigned)sh.ok;
    __syncthreads();
    return v;
}
// progress word of half h of the k-th streamed tile below the diagonal of block column c
__device__ __forceinline__ unsigned *sprog_at(const Args &a, int k, int h, int c) { return a.sprog + (((size_t)k * 2 + h) * a.NT + c) * QSTRIDE; }

// publish: every storing wave has drained its write-through stores, then one lane sets the word
__device__ __forceinline__ void wg_drain() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int NJ>
__device__ __forceinline__ void zero_acc(v4d (&acc)[NJ][4]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// C tile epilogue (TM x 128).  SUB: C = C - acc (else C = acc).  LOWER: entries above the diagonal of the tile stay
// untouched (diagonal tiles).  SC1: write-through stores.  Batches of 16 loads before the first store of a batch (see
// chol_update_kernel: element-wise read-modify-write serialises 64 dependent round trips).
// (roff: row offset of a 64-row half inside its 128 x 128 tile, for LOWER)
template <int TM, bool SUB, bool LOWER, bool SC1>
__device__ __forceinline__ void store_tile(double *__restrict__ C, int64_t ldc, const v4d (&acc)[TM / 32][4], int roff = 0) {
    constexpr int NJ = TM / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ioff = (TM == 128) ? (wave >> 1) * 64 : 0;
    const int joff = (TM == 128) ? (wave & 1) * 64 : wave * 32;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
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
                if (!LOWER || gi + roff >= gj) {
                    if (SC1)
                        st_sc1(C + gi + gj * ldc, v);
                    else
                        *(gf64 *)&C[gi + gj * ldc] = v;
                }
            }
        }
    }
}

// acc = -C in the layout of store_tile (a bulk job starts from the tile instead of ending with a read-modify-write: the
// four dependent load / store round trips of the epilogue become one load burst under the pipeline's prologue)
template <int TM>
__device__ __forceinline__ void load_tile_neg(const double *__restrict__ C, int64_t ldc, v4d (&acc)[TM / 32][4]) {
    constexpr int NJ = TM / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ioff = (TM == 128) ? (wave >> 1) * 64 : 0;
    const int joff = (TM == 128) ? (wave & 1) * 64 : wave * 32;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gj = joff + j * 16 + l4 + 4 * r;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i][r] = -*(const gf64 *)&C[(ioff + i * 16 + l15) + gj * ldc];
        }
}

#define JLOG(k)                                                             \
    do {                                                                   \
        if (sh.jrec && threadIdx.x == 0) sh.jrec[k] = wall_clock64();      \
    } while (0)

// ---- jobs ------------------------------------------------------------------------------------------------------
// ucnt counts half tiles: a full 128-row job adds 2, a 64-row job 1; window w of tile (i,c) may start at ucnt >= 2w
template <int TM>
__device__ __attribute__((noinline)) bool run_bulk(const Args &a, Shared &sh, const Job jb) {
    const int64_t lda = uni64(a.lda);
    double *const A = uni_ptr(a.A);
    const int i = __builtin_amdgcn_readfirstlane(jb.i), c = __builtin_amdgcn_readfirstlane(jb.c), jw = __builtin_amdgcn_readfirstlane(jb.w);
    const int w = jw & 255, hcode = jw >> 8, roff = (TM == 64 && hcode == 1) ? 64 : 0;  // (hcode 2: rows 0 .. 63 of a tile whose rows 64 .. 127 are zero)
    const int pl = wstart(w + 1, a.first, a.win) - 1;  // last panel of the window: rows finish their panels in order
    if (!wg_wait(sh, a, a.tdone + (size_t)i * a.NT + pl, 2u, a.tdone + (size_t)c * a.NT + pl, 2u, a.ucnt + (size_t)i * a.NT + c,
                 (unsigned)(2 * w), 0x100u))
        return false;
    JLOG(2);
    v4d acc[TM / 32][4];
    double *C = A + (int64_t)i * NB + roff + (int64_t)c * NB * lda;
    load_tile_neg<TM>(C, lda, acc);  // the tile's earlier bulk updates are in (ucnt, awaited above)
    const int64_t k0 = (int64_t)wstart(w, a.first, a.win) * NB;
    gemm_acc<TM>(A + (int64_t)i * NB + roff + k0 * lda, lda, A + (int64_t)c * NB + k0 * lda, lda,
                 (wstart(w + 1, a.first, a.win) - wstart(w, a.first, a.win)) * NB, acc, sh.u.gemm);
    JLOG(3);
#pragma unroll
    for (int j = 0; j < TM / 32; ++j)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) acc[j][i2] = -acc[j][i2];
    if (i == c)
        store_tile<TM, false, true, true>(C, lda, acc, roff);  // (a half of a diagonal tile: rows roff .. roff + 63 against all 128 columns)
    else
        store_tile<TM, false, false, true>(C, lda, acc);
    wg_drain();
    if (threadIdx.x == 0) addf(a.ucnt + (size_t)i * a.NT + c, (TM == 128 || hcode == 2) ? 2u : 1u);
    return true;
}

// left-looking part shared by the panel jobs: X = A(i,c) - sum_{p in [4 nb(c), pend)} L(i,p) L(c,p)'  (panels no bulk job applies)
// written back in place (plain stores: only this workgroup reads X again).  TM = 64 handles rows [roff, roff + 64) of
// block row i.  Returns false on abort.
template <int TM>
__device__ __forceinline__ bool window_part(const Args &a, Shared &sh, int i, int c, int pend, int roff,
                                                      unsigned long long *tr) {
    const int64_t lda = uni64(a.lda);
    double *const A = uni_ptr(a.A);
    const int wc = nbulk_updates(i, c, a.slack, a.slack_chain, a.first, a.win, srows_at(c, a.srows, a.edge)), p0 = wstart(wc, a.first, a.win);
    const unsigned *uc = a.ucnt + (size_t)i * a.NT + c;
    if (p0 >= pend) return wg_wait(sh, a, uc, (unsigned)(2 * wc), nullptr, 0, nullptr, 0, 0x200u);
    v4d acc[TM / 32][4];
    zero_acc(acc);
    for (int p = p0; p < pend;) {
        // wait for panel p, then take every further panel that is already finished in the same GEMM call (contiguous columns):
        // one poll / acquire / pipeline start for the run instead of one per panel.  The tile's own bulk updates (ucnt) are NOT
        // awaited here: the products accumulate in registers and the tile in memory is only touched by the read-modify-write
        // below, so the GEMM part runs while the last bulk update of the tile is still in flight (measured: the diagonal tile's
        // last window update used to arrive ~100 us after the panels and held up the whole left-looking part, job log r02).
        if (threadIdx.x == 0) {
            bool ok = poll_ge(a.tdone + (size_t)i * a.NT + p, 2u, a, 0x210u) && poll_ge(a.tdone + (size_t)c * a.NT + p, 2u, a, 0x211u);
            int run = 0;
            if (ok) {
                run = 1;
                while (p + run < pend && ldf(a.tdone + (size_t)i * a.NT + p + run) >= 2u && ldf(a.tdone + (size_t)c * a.NT + p + run) >= 2u) ++run;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sh.ok = run;
        }
        __syncthreads();
        const int run = sh.ok;
        __syncthreads();
        if (run == 0) return false;
        if (tr && threadIdx.x == 0) tr[p == p0 ? 5 : 6] = wall_clock64();  // first / latest dependency satisfied
        JLOG(p == p0 ? 2 : 3);
        const int64_t k0 = (int64_t)p * NB;
        gemm_acc<TM>(A + (int64_t)i * NB + roff + k0 * lda, lda, A + (int64_t)c * NB + k0 * lda, lda, run * NB, acc, sh.u.gemm);
        p += run;
    }
    // the bulk updates of this tile must be in memory before it is read back
    if (tr && threadIdx.x == 0) tr[7] = wall_clock64();  // GEMM part done, waiting for the tile's bulk updates
    if (!wg_wait(sh, a, uc, (unsigned)(2 * wc), nullptr, 0, nullptr, 0, 0x212u)) return false;
    double *C = A + (int64_t)i * NB + roff + (int64_t)c * NB * lda;
    if (TM == 128 && i == c)
        store_tile<TM, true, true, false>(C, lda, acc);
    else
        store_tile<TM, true, false, false>(C, lda, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return true;
}

// S(i,c): panel solve of a tile right below the diagonal, in step with the factorisation of the diagonal block.
// Wave v owns rows 32v .. 32v+31 of the tile as 2 x 8 tiles of 16 x 16 in the MFMA C/D layout (lane l, register r:
// X[16u + (l & 15)][16q + (l >> 4) + 4r]), which is directly the B operand of the next MFMA.
//
// Round 4: the 16-column panels both phases take from their producers come in by LDS-DMA, one panel ahead when the producer is
// ahead.  A streamed job spends most of its life catching up (job log r04, n = 2048: S(c+1,c) started its fold of panel c-1 when
// its producers had finished, took 3.3 us per 16-column step -- 1.4 of it the operand loads' trip, issued behind the previous
// step's MFMAs because 208 of 256 registers hold the tile and the operands -- and reached the diagonal block's last panel 7 us
// late, which is what P(c+1) then waited for).  Through LDS the next step's operands travel under this step's MFMAs.
// HALF: the job owns rows [64 h, 64 h + 64) of the tile (jb.w = 1 + h), wave v rows 16v .. 16v+15 of them.  Chain-bound block columns
// are streamed as halves (shalf_at): a 128-row job carries 37 us of matrix-pipe time per block column (fold 8 x 64, solve 8 x 16..72
// MFMAs per wave at 34 ns) plus its left-looking GEMM -- more than the 35 us a chain-bound column takes --, so once behind (it
// starts behind: its last left-looking panel is final only when the previous-but-one column's streamed row has ended) it stays
// behind, and the diagonal job of the next column waits for it (trace r04, n = 2048: S(c+1,c) reached the last panel of P(c)
// 7 us late).  Two workgroups per tile halve every part of that.
template <bool HALF>
__device__ __attribute__((noinline)) bool run_stream(const Args &a, Shared &sh, const Job jb) {
    const int64_t lda = uni64(a.lda);
    double *const A = uni_ptr(a.A);
    const int i = __builtin_amdgcn_readfirstlane(jb.i), c = __builtin_amdgcn_readfirstlane(jb.c);  // uniform: addresses and flags in SGPRs
    const int hh = HALF ? __builtin_amdgcn_readfirstlane(jb.w) - 1 : 0, roff = 64 * hh;
    constexpr int RW = HALF ? 16 : 32, U = HALF ? 1 : 2;  // rows per wave, 16-row tiles per wave
    // all left-looking panels but the newest (c - 1) through the GEMM loop, in place
    if (HALF ? !window_part<64>(a, sh, i, c, c > 0 ? c - 1 : 0, roff, nullptr) : !window_part<128>(a, sh, i, c, c > 0 ? c - 1 : 0, 0, nullptr)) return false;
    JLOG(4);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    double *C = A + (int64_t)i * NB + roff + (int64_t)c * NB * lda;  // row 0 of this job's rows
    const double *Lcc = A + (int64_t)c * NB + (int64_t)c * NB * lda;
    const double *itg = uni_ptr(a.itg + (size_t)c * 8 * 256);
    const unsigned *dprog = uni_ptr(a.dprog + (size_t)c * QSTRIDE);
    unsigned *sprog0 = uni_ptr(sprog_at(a, i - c - 1, HALF ? hh : 0, c)), *sprog1 = uni_ptr(sprog_at(a, i - c - 1, HALF ? hh : 1, c));  // (a 128-row job: both halves' words)
    // two LDS slots of [128 x 16 panel | 128 x 16 panel]: [k][row] with the GEMM loop's row stride (conflict-free fragment reads)
    constexpr int PAN = 16 * LDS_LD, SLOT = 2 * PAN;
    double *const ring = sh.u.gemm;
    v4d x[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[u][q][r] = *(const gf64 *)&C[(RW * wave + 16 * u + l15) + (int64_t)(16 * q + l4 + 4 * r) * lda];
    if (c > 0) {
        // X -= L(i,c-1) L(c,c-1)', 16 columns at a time: L(c,c-1) is the first streamed tile of block column c-1, L(i,c-1) the
        // (i-c+1)-th one, or -- the last S row -- a T tile that is awaited whole
        const double *Lr = A + (int64_t)i * NB + roff + (int64_t)(c - 1) * NB * lda;  // this job's rows of tile (i, c-1)
        const double *Lc = A + (int64_t)c * NB + (int64_t)(c - 1) * NB * lda;
        const unsigned *fc0 = sprog_at(a, 0, 0, c - 1), *fc1 = sprog_at(a, 0, 1, c - 1);   // tile (c, c-1): all of its rows are the update's columns
        const bool row_streamed = i - (c - 1) <= srows_at(c - 1, a.srows, a.edge);  // was tile (i, c-1) a streamed one?
        const unsigned *fr0 = row_streamed ? sprog_at(a, i - c, HALF ? hh : 0, c - 1) : nullptr;      // this job's rows of tile (i, c-1)
        const unsigned *fr1 = (row_streamed && !HALF) ? sprog_at(a, i - c, 1, c - 1) : nullptr;
        if (!row_streamed && !wg_wait(sh, a, a.tdone + (size_t)i * a.NT + (c - 1), 2u, nullptr, 0, nullptr, 0, 0x520u)) return false;
        auto issue_fold = [&](int b) {  // wave w: columns 4w .. 4w+3 of both panels, eight LDS-DMA instructions
            double *sl = ring + (b & 1) * SLOT;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = 4 * wave + u;
                glds16(Lr + 2 * (HALF ? (lane & 31) : lane) + (int64_t)(16 * b + col) * lda, sl + col * LDS_LD);  // (64 rows: half a wave; the upper lanes repeat them into the slot's unused rows -- no LDS-DMA inside divergent control flow, see mega_gemm.hpp)
                glds16(Lc + 2 * lane + (int64_t)(16 * b + col) * lda, sl + PAN + col * LDS_LD);
            }
        };
        int got = 0, issued = 0;
#pragma unroll 1
        for (int b = 0; b < 8; ++b) {
            if (got < b + 1) {
                const unsigned g = wg_wait_val4(sh, a, fc0, fc1, fr0, fr1, (unsigned)(b + 1), 0x510u);
                if (!g) return false;
                got = __builtin_amdgcn_readfirstlane((int)g);
            }
            // panel b, and panel b + 1 when it is already published: its slot is that of panel b - 1, whose readers passed the barrier
            // at the end of the previous step
            const int upto = got < b + 2 ? got : b + 2;
            for (; issued < upto; ++issued) issue_fold(issued);
            if (issued > b + 1)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave's share of panel b has landed
            const double *sl = ring + (b & 1) * SLOT;
            double opr[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) opr[u][s2] = sl[(4 * s2 + l4) * LDS_LD + RW * wave + 16 * u + l15];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double opc[4];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) opc[s2] = sl[PAN + (4 * s2 + l4) * LDS_LD + 16 * q + l15];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) x[u][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-opc[s2], opr[u][s2], x[u][q], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // the slot may be refilled
        }
    }
    unsigned long long *const strbase = (a.trace && i == c + 1 && hh == 0) ? uni_ptr(a.trace + (size_t)a.NT * 16 + 4 * 1024 + (size_t)c * 64) : nullptr;
    unsigned *const tdone_ic = uni_ptr(a.tdone + (size_t)i * a.NT + c);
    auto issue_solve = [&](int b) {  // panel b of L_cc (waves: columns 4w .. 4w+3) and its leaf inverse (two 1-KB halves; waves 2, 3 repeat 0, 1)
        double *sl = ring + (b & 1) * SLOT;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = 4 * wave + u;
            glds16(Lcc + 2 * lane + (int64_t)(16 * b + col) * lda, sl + col * LDS_LD);
        }
        glds16(itg + b * 256 + (wave & 1) * 128 + 2 * lane, sl + PAN + (wave & 1) * 128);
    };
    int have = 0, sissued = 0;  // panels of the diagonal block known to be published / whose operands are on their way
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        if (have < b + 1) {
            const unsigned g = wg_wait_val(sh, a, dprog, (unsigned)(b + 1), 0x500u);
            if (!g) return false;
            have = __builtin_amdgcn_readfirstlane((int)g);
        }
        if (b == 0) JLOG(5);
        if (b == 7) JLOG(6);
        // operands of this step and, when the diagonal block is ahead, of the next one (its slot's readers met at the previous step's
        // drain barrier; a step that was issued a step ago has landed: that barrier sat behind vmcnt(0))
        {
            const int upto = have < b + 2 ? have : b + 2;
            for (; sissued < upto; ++sissued) issue_solve(sissued);
        }
        if (sissued > b + 1)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // per-step opaque copies of the lane coordinates: the store / operand addresses derived from them are otherwise computed
        // once for all eight (unrolled) steps, spilled, and reloaded between the write-through stores behind s_waitcnt vmcnt(0)
        int l15s = l15, l4s = l4;
        asm volatile("" : "+v"(l15s), "+v"(l4s));
        unsigned long long *str = strbase ? strbase + 8 * b : nullptr;
        if (str && threadIdx.x == 0) str[0] = wall_clock64();  // panel b of the diagonal block seen
        const double *sl = ring + (b & 1) * SLOT;
        double ia[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) ia[s2] = sl[PAN + (4 * s2 + l4s) * 16 + l15s];
        double lq[8][4];
#pragma unroll
        for (int q = b + 1; q < 8; ++q)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) lq[q][s2] = sl[(4 * s2 + l4s) * LDS_LD + 16 * q + l15s];
        v4d xs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v4d t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) t = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[s2], x[u][b][s2], t, 0, 0, 0);
            xs[u] = t;
#pragma unroll
            for (int r = 0; r < 4; ++r) st_sc1(&C[(RW * wave + 16 * u + l15s) + (int64_t)(16 * b + l4s + 4 * r) * lda], t[r]);
        }
        // the next step only needs block column b + 1 brought up to date: that one before the panel is published, the others after
        // (the early steps carry up to 56 MFMAs per wave, 1.6 us, which the consumers of this panel need not wait for)
        if (b < 7) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) x[u][b + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lq[b + 1][s2], xs[u][s2], x[u][b + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // publish the finished 16-column panel, also when this job is only catching up with a diagonal block that is already
        // complete: its consumers (the next diagonal job, the next column's streamed jobs) fold panel by panel at ~3 us each and
        // would otherwise start all eight after this job's end (seen as 17-23 us instead of 3 us between the end of S(c+1,c) and
        // the start of P(c+1): trace r02)
        wg_drain();  // (also: every wave's LDS reads of this step's slot are in -- lq lives in registers from here on)
        if (threadIdx.x == 0) {
            stf(sprog0, (unsigned)(b + 1));
            if (!HALF) stf(sprog1, (unsigned)(b + 1));
            if (b == 7) {
                if (HALF)
                    addf(tdone_ic, 1u);  // the tile is final when both halves are
                else
                    stf(tdone_ic, 2u);
            }
            if (str) str[5] = wall_clock64();  // own panel b published
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = b + 2; q < 8; ++q)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) x[u][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lq[q][s2], xs[u][s2], x[u][q], 0, 0, 0);
    }
    return true;
}

#define MEGA_STAMP(k)                                                          \
    do {                                                                       \
        if (tr && threadIdx.x == 0) tr[k] = wall_clock64();                    \
    } while (0)

// TRACED: diagnostic launches (time stamps of the job, cycle counts of one wave of the diagonal core)
template <bool TRACED>
__device__ __attribute__((noinline)) bool run_diag(const Args &a, Shared &sh, const Job jb) {
    const int64_t lda = uni64(a.lda);
    double *const A = uni_ptr(a.A);
    const int c = __builtin_amdgcn_readfirstlane(jb.c);  // uniform: tile / flag addresses stay in SGPRs (spilled VGPR addresses used to
                                                         // be reloaded behind s_waitcnt vmcnt(0), i.e. behind the write-through stores)
    unsigned long long *tr = TRACED ? a.trace + (size_t)(2 * c) * 8 : nullptr;
    MEGA_STAMP(0);
    // All panels but the last `pstream` through the GEMM loop; the last ones (tiles (c, c-1), (c, c-2): streamed tiles of their
    // block columns) are folded in 16 columns at a time as their producers publish them.  Two on chain-bound sizes: with one, tile
    // (c, c-2) -- final a few us after P(c-2) -- still went through a 128^3 GEMM, the read-modify-write and eight catch-up folds
    // (46 us) between the end of S(c, c-2) and the start of P(c), which paced the chain at 38 us per column once the diagonal
    // block itself took 27.  (On large matrices the early read-modify-write would wait for bulk updates: one there.)
    const int pstream_c = edge_col(c, a.edge) ? a.edge.pstream_edge : a.pstream;
    const int pstream = c < pstream_c ? c : pstream_c;
    if (!window_part<128>(a, sh, c, c, c - pstream, 0, tr)) return false;
    MEGA_STAMP(1);
    JLOG(4);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    double *C = A + (int64_t)c * NB + (int64_t)c * NB * lda;
    double *Linv = a.linv + (size_t)c * NB * NB;
    diagcore::v4d acc[diagcore::NSLOT6];
    diagcore::diag_v6_load(C, lda, acc);
    // (round 4: the 16-column panels arrive by LDS-DMA, one panel ahead when the producer is ahead -- see run_stream.  Two slots in
    //  the first 37 KB of the workgroup's LDS: nothing the diagonal core touches before its first barrier lies there, so the leaf
    //  wave, which leaves the last step without waiting for the others, may start the core while they finish their tiles.)
    {
        constexpr int PAN = 16 * LDS_LD;
        double *const ring = sh.u.gemm;
#pragma unroll 1
        for (int pp = c - pstream; pp < c; ++pp) {
            const double *Lp = A + (int64_t)c * NB + (int64_t)pp * NB * lda;                       // tile (c, pp), produced by S(c, pp),
            const unsigned *sprog0 = uni_ptr(sprog_at(a, c - pp - 1, 0, pp)), *sprog1 = uni_ptr(sprog_at(a, c - pp - 1, 1, pp));  // the (c-pp)-th streamed tile of column pp, both halves
            auto issue_fold = [&](int b) {  // wave w: columns 4w .. 4w+3 of panel b
                double *sl = ring + (b & 1) * PAN;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int col = 4 * wave + u;
                    glds16(Lp + 2 * lane + (int64_t)(16 * b + col) * lda, sl + col * LDS_LD);
                }
            };
            int have = 0, issued = 0;  // 16-column panels of tile (c, pp) known to be published / on their way
#pragma unroll 1
            for (int b = 0; b < 8; ++b) {
                if (have < b + 1) {
                    const unsigned g = wg_wait_val4(sh, a, sprog0, sprog1, nullptr, nullptr, (unsigned)(b + 1), 0x600u);
                    if (!g) return false;
                    have = __builtin_amdgcn_readfirstlane((int)g);
                }
                const int upto = have < b + 2 ? have : b + 2;
                for (; issued < upto; ++issued) issue_fold(issued);
                if (issued > b + 1)
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // every wave's share of panel b has landed
                const double *sl = ring + (b & 1) * PAN;
                double op[8][4];
                // (the leaf wave's three tiles only need the first two row blocks: it is through its fold -- and, after the last panel,
                //  into its first leaf -- while the others still work on their eleven tiles each)
#pragma unroll
                for (int xb = 0; xb < 8; ++xb)
                    if (xb < 2 || wave != 0) {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) op[xb][s2] = sl[(4 * s2 + l4) * LDS_LD + 16 * xb + l15];
                    }
#pragma unroll
                for (int ti = 0; ti < 8; ++ti)
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj) {
                        if (diagcore::v6_owner(ti, tj) == wave) {
#pragma unroll
                            for (int s2 = 0; s2 < 4; ++s2)
                                acc[diagcore::v6_slot(ti, tj)] =
                                    __builtin_amdgcn_mfma_f64_16x16x4f64(-op[tj][s2], op[ti][s2], acc[diagcore::v6_slot(ti, tj)], 0, 0, 0);
                        }
                    }
                if (b < 7 || pp + 1 < c) {  // (not behind the very last panel: nothing refills its slot)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();  // the slot may be refilled
                }
            }
        }
    }
    MEGA_STAMP(2);
    JLOG(5);
    __builtin_amdgcn_s_setprio(3);
    const int bad = diagcore::diag_v6_core(C, lda, Linv, sh.u.diag, acc, a.itg + (size_t)c * 8 * 256, a.dprog + (size_t)c * QSTRIDE,
                                           TRACED ? tr + 15 : nullptr);
    __builtin_amdgcn_s_setprio(1);
    if (bad) {
        if (threadIdx.x == 0) {
            *(__attribute__((address_space(1))) int *)a.info = c * NB + bad;
            stf(a.ctl + CTL_ABORT, 1u);
        }
        return false;
    }
    MEGA_STAMP(3);
    JLOG(6);
    wg_drain();
    if (threadIdx.x == 0) {
        stf(a.tdone + (size_t)c * a.NT + c, 2u);
        addf(a.ctl + CTL_PCOLS, 1u);
    }
    MEGA_STAMP(4);
    return true;
}

// T(i, half, c): rows [64 half, 64 half + 64) of tile (i,c).  Half-height tiles keep a row's column-to-column recurrence
// L(i,c-1) -> update -> solve -> L(i,c) (two dependent 64 x 128 x 128 GEMMs on one CU that is shared with a bulk job)
// faster than the diagonal chain; with 128-row tiles the rows fall behind it.
template <bool FULL>
__device__ __attribute__((noinline)) bool run_panel(const Args &a, Shared &sh, const Job jb) {
    constexpr int TM = FULL ? 128 : 64;
    const int64_t lda = uni64(a.lda);
    double *const A = uni_ptr(a.A);
    const int i = __builtin_amdgcn_readfirstlane(jb.i), c = __builtin_amdgcn_readfirstlane(jb.c), wcode = __builtin_amdgcn_readfirstlane(jb.w);
    const int roff = (!FULL && wcode == 1) ? 64 : 0;  // (wcode 3: rows 0 .. 63 of a tile whose rows 64 .. 127 are zero and stay zero)
    if (!window_part<TM>(a, sh, i, c, c, roff, nullptr)) return false;
    JLOG(4);
    double *C = A + (int64_t)i * NB + roff + (int64_t)c * NB * lda;
    double *Linv = a.linv + (size_t)c * NB * NB;
    // L(i,c) = X * inv(L_cc)'
    if (!wg_wait(sh, a, a.tdone + (size_t)c * a.NT + c, 2u, nullptr, 0, nullptr, 0, 0x300u)) return false;
    JLOG(5);
    v4d acc[TM / 32][4];
    zero_acc(acc);
    // (the register-staged loop: X was written by this workgroup's plain stores a moment ago, and LDS-DMA loads of it came back
    //  wrong -- every other operand of gemm_acc is another workgroup's write-through data behind an acquire, or older; r04)
    // (rounds 4 / 5: the LDS-DMA ring gave wrong results here and only here -- not a stale read of the tile this workgroup had just
    //  written, as first thought, but a half-wave LDS-DMA inside divergent control flow that the compiler, knowing K and ldb at this call
    //  site, merged with the full-wave ones behind a non-uniform LDS base; fixed in mega_gemm.hpp, profiles/r05_ldsdma_hazard.txt)
    if (__builtin_amdgcn_readfirstlane(a.panel_dma))
        gemm_acc_v2<TM>(C, lda, Linv, NB, NB, acc, sh.u.gemm);
    else
        gemm_acc_v1<TM>(C, lda, Linv, NB, NB, acc, sh.u.gemm);  // every wave's loads of X are complete behind the loop's last barrier
    store_tile<TM, false, false, true>(C, lda, acc);
    wg_drain();
    if (a.fault && i == a.MT - 1 && c == 0 && roff == 0) return true;  // test hook: this (half) tile is never published
    if (threadIdx.x == 0) addf(a.tdone + (size_t)i * a.NT + c, (FULL || wcode == 3) ? 2u : 1u);
    return true;
}

__device__ __forceinline__ void jlog_begin(const Args &a, Shared &sh, const Job jb) {
    if (threadIdx.x == 0) {
        unsigned long long *r = nullptr;
        if (a.jlog) {
            const unsigned long long n = __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned long long *)a.jlog, 1ull,
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n < (unsigned long long)a.jlog_cap) {
                r = a.jlog + 8 + n * 8;
                r[0] = (unsigned long long)(unsigned short)jb.kind | ((unsigned long long)(unsigned short)jb.i << 8) |
                       ((unsigned long long)(unsigned short)jb.c << 24) | ((unsigned long long)(unsigned short)jb.w << 40) |
                       ((unsigned long long)blockIdx.x << 52);
                r[1] = wall_clock64();
            }
        }
        sh.jrec = r;
    }
    __syncthreads();
}
__device__ __forceinline__ void jlog_end(Shared &sh) {
    if (sh.jrec && threadIdx.x == 0) sh.jrec[7] = wall_clock64();
}

// wave 0: among the heads of the window queues find the ready job with the smallest block column and claim the next job of that
// queue.  Lanes 0..31 look at the ordinary queues of the windows [wlo, wlo + 32), lanes 32..63 at the chain tiles' queues of the
// windows [cwlo, cwlo + 32).  Returns the job index in bjobs, -1 (nothing ready now) or -2 (every queue drained).
// The chain tiles (diagonal tile and the streamed tiles below it) have queues of their own because their window updates are
// serial per tile (each a K = 128 win GEMM of ~150-200 us, as long as `win` chain steps): taken in column order with everything
// else they reached a tile only when the bulk front did, its last three or four windows back to back, and the diagonal job waited
// 30-280 us for the newest one at every window boundary (job log r04).  In their own queues, `cboost` columns ahead of the front,
// they run soon after their window closes, spread over the time the chain needs to get there.
__device__ __forceinline__ int pick_bulk(const Args &a, Shared &sh) {
    const int lane = threadIdx.x & 63, cls = lane >> 5, l32 = lane & 31;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const int wlo = sh.wlo, cwlo = sh.cwlo;
        if (wlo >= a.nwin && cwlo >= a.nwin) return -2;
        const int wnd = (cls ? cwlo : wlo) + l32;
        const bool valid = wnd < a.nwin;
        const int wq = cls * a.nwin + wnd;
        unsigned h = 0;
        int cnt = 0, base = 0;
        if (valid) {
            base = a.wq_start[wq];
            cnt = a.wq_start[wq + 1] - base;
            h = ldf(a.wq_head + (size_t)wq * QSTRIDE);
        }
        const bool has = valid && h < (unsigned)cnt;
        // advance wlo / cwlo past drained queues
        const unsigned long long has_mask = __ballot(has);
        const unsigned m0 = (unsigned)has_mask, m1 = (unsigned)(has_mask >> 32);
        bool moved = false;
        if (wlo < a.nwin) {
            const int adv = m0 ? __ffs((int)m0) - 1 : 32;
            if (adv > 0) {
                if (lane == 0) sh.wlo = wlo + adv;
                moved = m0 == 0u;
            }
        }
        if (cwlo < a.nwin) {
            const int adv = m1 ? __ffs((int)m1) - 1 : 32;
            if (adv > 0) {
                if (lane == 0) sh.cwlo = cwlo + adv;
                moved = moved || m1 == 0u;
            }
        }
        if (has_mask == 0ull) {
            if (!moved) return -2;
            continue;  // a class moved on by 32 windows: look again
        }
        unsigned key = 0xffffffffu;
        if (has) {
            const Job jb = a.bjobs[base + (int)h];
            const int jw = jb.w & 255, pl = wstart(jw + 1, a.first, a.win) - 1;
            if (ldf(a.tdone + (size_t)jb.i * a.NT + pl) >= 2u && ldf(a.tdone + (size_t)jb.c * a.NT + pl) >= 2u &&
                ldf(a.ucnt + (size_t)jb.i * a.NT + jb.c) >= (unsigned)(2 * jw))
                key = ((unsigned)(1024 + jb.c + a.wbias * wnd - (cls ? a.cboost : 0)) << 8) | (unsigned)lane;  // newer windows trail the older ones by wbias columns
        }
        unsigned best = key;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)best, off);
            best = o < best ? o : best;
        }
        if (best == 0xffffffffu) return -1;
        const int win_lane = (int)(best & 0xffu);
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

__device__ __forceinline__ void mega_body(const Args &a) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    // The chain workgroups are the blocks 0, 8, 16, ...: blocks are dealt round-robin over the 8 XCDs (observed, not promised; nothing
    // depends on it but speed), so the whole chain sits behind ONE L2 and its hand-offs stay on one die (tools/pingpong: 0.47 against
    // 0.58 us for a write-through word; n = 2048 / 4096 / 8192: -1.2 / -2.8 / -0.5 % against the blocks 0 .. nchain-1).  Copies of
    // the chain's panels written with plain stores and read from the L2 (0.3 us) were tried on top: only the LAST panel's hand-off of a
    // block column is on the critical path, -1 us of 36 per column, not kept.  MRBF_MEGA_XCHAIN=0: the old placement.
    const int bx = (int)blockIdx.x;
    // xchain = 2 .. 8: the same on so many XCDs -- blocks = 0 .. xchain-1 (mod 8), the first workgroup of each of their CUs (blocks < 256)
    // first, so that up to 32 xchain chain workgroups have a CU each.  (One XCD holds 32: a 64-workgroup chain on one XCD put two
    // chain workgroups on every CU of it, and the half-height streamed jobs that need the 64 lost more to that than they gained.)
    const int nx = a.xchain, cls = nx > 0 && (bx & 7) < nx;
    const int cidx = !cls ? -1 : (bx < 256 ? (bx >> 3) * nx + (bx & 7) : 32 * nx + ((bx - 256) >> 3) * nx + (bx & 7));  // rank in the chain's placement class
    const int oidx = nx == 0 ? bx - a.nchain : (cls ? -1 : (bx < 256 ? bx - ((bx >> 3) * nx + nx) : bx - 32 * nx - (((bx - 256) >> 3) * nx + nx)));  // rank outside it
    const bool chain = nx ? (cls && cidx < a.nchain) : bx < a.nchain;
    const bool dedicated = !chain && oidx >= 0 && oidx < a.ndedicated;
    if (chain || dedicated) __builtin_amdgcn_s_setprio(2);
    if (threadIdx.x == 0) {
        sh.wlo = 0;
        sh.cwlo = 0;
        // (XCC, shader engine, CU) of this workgroup: a 512-workgroup launch puts exactly two workgroups on each of the 256 CUs
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        sh.mycu = (int)(((xcc & 7u) << 6) | (((hw >> 13) & 3u) << 4) | ((hw >> 8) & 15u));
    }
    __syncthreads();
    unsigned *myquiet = a.quiet + (size_t)sh.mycu * QSTRIDE;
    // reserve workgroups (same placement class as the chain): general workers in the throughput-bound middle, chain workgroups while the
    // chain queue is in the edge regime -- before job `head_job1` and from job `reserve_job0` on, where the machine has idle workgroups
    // anyway and five streamed rows per column need more hands than the middle's three
    const bool reserve = !chain && cls && cidx < a.nchain + a.nreserve;
    bool in_chain = chain;
    if (reserve && threadIdx.x == 0) {
        const unsigned qc = ldf(a.ctl + CTL_QC);
        sh.jidx = (qc < (unsigned)a.head_job1 || qc >= (unsigned)a.reserve_job0) ? 1 : 0;
    }
    if (reserve) {
        __syncthreads();
        in_chain = sh.jidx != 0;
        __syncthreads();
    }
    unsigned long long idle_t0 = 0;  // start of the current run of idle scans
    int nidle = 0;
    while (true) {
    if (in_chain) {
        // chain workgroups only run the diagonal / streamed jobs, in order; the CU's other workgroup (if it is a general
        // one) stays idle meanwhile, so the latency-bound chain never shares its SIMDs
        if (a.use_quiet && threadIdx.x == 0) addf(myquiet, 1u);
        if (reserve) __builtin_amdgcn_s_setprio(2);
        bool leave = false;
        while (true) {
            if (threadIdx.x == 0) {
                int idx = -1;
                if (!ldf(a.ctl + CTL_ABORT)) {
                    if (reserve) {
                        const unsigned qc = ldf(a.ctl + CTL_QC);
                        if (qc >= (unsigned)a.head_job1 && qc < (unsigned)a.reserve_job0) idx = -2;  // the middle: back to the general pool
                    }
                    if (idx != -2) {
                        const unsigned got = addf(a.ctl + CTL_QC, 1u);
                        if (got < (unsigned)a.nchainjobs) idx = (int)got;
                    }
                }
                sh.jidx = idx;
            }
            __syncthreads();
            const int idx = sh.jidx;
            __syncthreads();
            if (idx == -2) {
                leave = true;
                break;
            }
            if (idx < 0) break;
            const Job jb = a.cjobs[idx];
            jlog_begin(a, sh, jb);
            if (!(jb.kind == JOB_P ? (a.trace ? run_diag<true>(a, sh, jb) : run_diag<false>(a, sh, jb))
                                   : (jb.w == 0 ? run_stream<false>(a, sh, jb) : run_stream<true>(a, sh, jb))))
                break;
            jlog_end(sh);
        }
        if (a.use_quiet && threadIdx.x == 0) addf(myquiet, 0xffffffffu);
        if (!leave) return;
        __builtin_amdgcn_s_setprio(0);
        in_chain = false;
        idle_t0 = 0;
        nidle = 0;
        continue;
    }
    {
        if (threadIdx.x < 64) {
            int kind = -1, idx = 0;  // -1 idle, -2 exit, -3 reserve workgroup: the chain queue has reached the tail
            bool panel_left = false;
            if (ldf(a.ctl + CTL_ABORT)) {
                kind = -2;
            } else if (reserve && ldf(a.ctl + CTL_QC) >= (unsigned)a.reserve_job0 && ldf(a.ctl + CTL_QC) < (unsigned)a.nchainjobs) {
                kind = -3;
            } else if (!dedicated && a.use_quiet && ldf(myquiet) != 0 && a.NT - (int)ldf(a.ctl + CTL_PCOLS) <= a.quiet_tail) {
                kind = -1;  // the CU's other workgroup runs a chain-critical job: leave it the SIMDs
                idle_t0 = 0;
            } else {
                const unsigned ph = ldf(a.ctl + CTL_QP);
                panel_left = ph < (unsigned)a.npanel;
                bool want_panel = false;
                if (panel_left) {
                    if (dedicated || a.ndedicated == 0)
                        want_panel = true;
                    else {
                        // general workgroups only take the head panel job if its bulk updates are in: they are the ones that
                        // run bulk jobs, and must never all sit inside panel jobs waiting for bulk updates
                        const Job hj = a.pjobs[ph];
                        want_panel = hj.c < (int)ldf(a.ctl + CTL_PCOLS) + a.look &&
                                     ldf(a.ucnt + (size_t)hj.i * a.NT + hj.c) >=
                                         2u * (unsigned)nbulk_updates(hj.i, hj.c, a.slack, a.slack_chain, a.first, a.win, srows_at(hj.c, a.srows, a.edge));
                    }
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
                if (kind == -1) {
                    // nothing ready for this workgroup: give up only after 4 x the dependency limit of wall-clock time (others may
                    // be inside long jobs; a stuck launch is reported by the waiting job itself long before)
                    const unsigned long long now = wall_clock64();
                    if (idle_t0 == 0) idle_t0 = now;
                    if (now - idle_t0 > 4ull * a.spin_ticks) {
                        if (threadIdx.x == 0) {
                            stf(a.ctl + CTL_TIMEOUT, 0x400u);
                            stf(a.ctl + CTL_ABORT, 1u);
                        }
                        kind = -2;
                    }
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
        if (kind == -3) {
            in_chain = true;
            continue;
        }
        if (kind == -1) {
            // back off: an idle workgroup's scan costs ~5 wave loads; 400 of them polling flat out slow everyone's memory traffic
            ++nidle;
            const int reps = nidle < 8 ? 1 : (nidle < 32 ? 4 : 16);
            for (int r = 0; r < reps; ++r) __builtin_amdgcn_s_sleep(64);
            continue;
        }
        idle_t0 = 0;
        nidle = 0;
        bool ok;
        const unsigned long long tj0 = a.trace ? wall_clock64() : 0ull;
        jlog_begin(a, sh, kind == 1 ? a.pjobs[idx] : a.bjobs[idx]);
        if (kind == 1) {
            if (!dedicated) __builtin_amdgcn_s_setprio(1);
            ok = a.pjobs[idx].w == 2 ? run_panel<true>(a, sh, a.pjobs[idx]) : run_panel<false>(a, sh, a.pjobs[idx]);
            if (!dedicated) __builtin_amdgcn_s_setprio(0);
        } else {
            ok = a.bjobs[idx].kind == JOB_UH ? run_bulk<64>(a, sh, a.bjobs[idx]) : run_bulk<128>(a, sh, a.bjobs[idx]);
        }
        jlog_end(sh);
        if (a.trace && threadIdx.x == 0) {
            unsigned long long *u = a.trace + (size_t)a.NT * 16 + (size_t)blockIdx.x * 4;
            u[kind == 1 ? 1 : 0] += wall_clock64() - tj0;
            u[kind == 1 ? 3 : 2] += 1;
        }
        if (!ok) break;
    }
    }
}

// The launch's own clock: every workgroup stamps wall_clock64 (100 MHz) when it starts and when it leaves; first start and last end
// are kept by 64-bit atomic max (the start inverted, so that the zeroed control block is the neutral element).  The difference is
// the time the grid spent on the device, whatever the host did between the hipEvents that bracket the phase (a host thread that is
// descheduled between recording the first event and launching the kernel shows up in the events, not here).
__global__ __launch_bounds__(256, 2) void potrf_mega_kernel(const Args a) {
    if (threadIdx.x == 0)
        __hip_atomic_fetch_max((unsigned long long *)(a.ctl + CTL_TS), ~(unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mega_body(a);
    if (threadIdx.x == 0)
        __hip_atomic_fetch_max((unsigned long long *)(a.ctl + CTL_TS + 2), (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void mega_status_kernel(const unsigned *ctl, int *info, unsigned long long *stat) {
    if (ctl[CTL_TIMEOUT] != 0 && *info == 0) *info = -(int)ctl[CTL_TIMEOUT];
    const unsigned long long *ts = (const unsigned long long *)(ctl + CTL_TS);
    if (stat) stat[0] = (ts[0] != 0 && ts[1] != 0) ? ts[1] - ~ts[0] : 0ull;  // device time of the launch, 10-ns ticks
}

}  // namespace mega

// The job tables of one shape -- pure host code (no GPU call), also reachable through mrbf_debug_mega_tables so that the CPU container
// can check their invariants, with sanitizers (csrc/Makefile: `make asan`).
//   chain queue  P(c), S(c+1..c+srows, c) column by column;  panel queue  T(i, half, c) for the other rows;
//   bulk queues  two per window w (ordinary tiles: queue w, chain tiles: queue nwin + w, empty without `chainq`): U(i, c, w) for
//                every tile the window reaches through a bulk job (nbulk_updates), as two 64-row halves when block column c lies
//                within `slack + half_cols` columns behind the window's end.
static void build_job_tables(int NT, int MT, int slack, int slack_chain, int first, int win, int srows, const mega::Edge &edge, int half_cols,
                             int tail_half, int tail_half_w, bool chainq, std::vector<mega::Job> &pj, std::vector<mega::Job> &bj, std::vector<mega::Job> &cj, std::vector<int> &wqs) {
    using namespace mega;
    pj.clear();
    bj.clear();
    cj.clear();
    wqs.clear();
    for (int c = 0; c < NT; ++c) {
        cj.push_back(Job{JOB_P, (short)c, (short)c, (short)0});
        for (int i = c + 1; i < MT; ++i) {
            if (i <= c + srows_at(c, srows, edge)) {
                if (shalf_at(c, srows, edge)) {  // two 64-row jobs (w = 1 + half)
                    cj.push_back(Job{JOB_S, (short)i, (short)c, (short)1});
                    cj.push_back(Job{JOB_S, (short)i, (short)c, (short)2});
                } else {
                    cj.push_back(Job{JOB_S, (short)i, (short)c, (short)0});
                }
            }
            else
                if (xhalf_at(i, NT, edge))
                    pj.push_back(Job{JOB_T, (short)i, (short)c, (short)3});
                else if (tfull_at(i, c, srows, edge))
                    pj.push_back(Job{JOB_T, (short)i, (short)c, (short)2});
                else
                    for (int h = 0; h < 2; ++h) pj.push_back(Job{JOB_T, (short)i, (short)c, (short)h});
        }
    }
    int nwin_max = 0;  // windows that reach at least one tile through a bulk job
    for (int c = 0; c < NT; ++c)
        for (int i = c; i < MT; ++i) nwin_max = std::max(nwin_max, nbulk_updates(i, c, slack, slack_chain, first, win, srows_at(c, srows, edge)));
    for (int cls = 0; cls < 2; ++cls)
        for (int w = 0; w < nwin_max; ++w) {
            wqs.push_back((int)bj.size());
            for (int c = 0; c < NT; ++c)
                for (int i = c; i < MT; ++i) {
                    const int sr = srows_at(c, srows, edge);
                    if (nbulk_updates(i, c, slack, slack_chain, first, win, sr) <= w) continue;  // this window reaches the tile inside its panel job
                    if (((i - c <= sr) && chainq) != (cls == 1)) continue;  // chain tiles: queues of their own
                    const bool half = (i != c && c < wstart(w + 1, first, win) + slack + half_cols) ||
                                      (c >= NT - tail_half && w >= nbulk_updates(i, c, slack, slack_chain, first, win, sr) - tail_half_w);
                    if (xhalf_at(i, NT, edge)) {
                        bj.push_back(Job{JOB_UH, (short)i, (short)c, (short)(w + 512)});  // upper half, counts for both
                    } else if (half) {
                        bj.push_back(Job{JOB_UH, (short)i, (short)c, (short)w});
                        bj.push_back(Job{JOB_UH, (short)i, (short)c, (short)(w + 256)});
                    } else {
                        bj.push_back(Job{JOB_U, (short)i, (short)c, (short)w});
                    }
                }
        }
    wqs.push_back((int)bj.size());
}

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
    // MRBF_MEGA_HOSTTRACE=<ms>: host-side time of every runtime call of this launcher, printed when the call as a whole took longer
    static const double host_trace_ms = mrbf_env("MRBF_MEGA_HOSTTRACE") ? atof(mrbf_env("MRBF_MEGA_HOSTTRACE")) : 0.0;
    double hstamp[8];
    int nh = 0;
    auto hnow = [&]() {
        if (host_trace_ms > 0.0 && nh < 8) hstamp[nh++] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    hnow();
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
    MRBF_TRY(get_buf(ctx, S_MEGA_IT, (size_t)NT * 8 * 256, &a.itg));
    // job tables (cached per shape)
    // Streamed tile rows below the diagonal, chain workgroups and the chain tiles' slack by size (measured, tools/sweep_sizes.sh): a
    // streamed row takes its predecessor column's panel in step instead of in one 14-us piece after a T job, which shortens the
    // dependency loop T(c+2,c-1) -> S(c+2,c) -> S(c+2,c+1) -> P(c+2) that paces chain-bound sizes (n <= 6144: 5-8 % with five rows);
    // every streamed row is one more resident job per column in flight (and one more paused CU partner), which the saturated middle of
    // large matrices pays for (n = 8192: three rows, n >= 12288: two).  Environment / option values override.
    const int NTq = (int)(ncols / NB);
    const int srows_auto = NTq <= 48 ? 5 : (NTq <= 96 ? 3 : 2);
    // streamed tiles as 64-row halves where the whole factorisation is chain-bound (n <= 4096), with twice the chain workgroups (each
    // on a CU of its own: four XCDs).  Measured r04 (tools/sweep_shalf.sh): n = 1024 / 2048 / 3072 / 4096: -8.6 / -10 / -9 /
    // -6 %; n >= 6144: the halves cost more matrix-pipe time than the chain gains (in the last 16 block columns only: -2.6 .. 0 %).
    static const int env_shalf = mrbf_env("MRBF_MEGA_SHALF") ? atoi(mrbf_env("MRBF_MEGA_SHALF")) : -1;
    static const int env_shalf_head = mrbf_env("MRBF_MEGA_SHALF_HEAD") ? atoi(mrbf_env("MRBF_MEGA_SHALF_HEAD")) : -1;
    const int sh_tail = env_shalf >= 0 ? env_shalf : (NTq <= 40 ? 1 << 20 : 0), sh_head = env_shalf_head >= 0 ? env_shalf_head : 0;
    const bool shalf_all = sh_tail >= NTq;
    // (n = 4608 / 5120 with 48 chain workgroups on two XCDs: -9 / -8 %, with 64 on four: -7 / -6 %; n = 6144: +4 %, left alone)
    const int chain_auto = NTq <= 48 ? (shalf_all ? (NTq <= 32 ? 64 : 48) : 32) : (NTq <= 96 ? 20 : 12);
    const int slack_chain_auto = (NTq > 48 && NTq <= 96) ? 7 : 6;
    const int slack = std::max(1, ctx->mega_slack), slack_chain = std::max(slack, ctx->mega_slack_chain > 0 ? ctx->mega_slack_chain : slack_chain_auto);
    // longer windows for large matrices (measured: n = 16384: 33.8 / 31.7 / 30.8 ms with 4 / 6 / 8 panels per window; n = 8192: the same)
    const int win = std::max(1, ctx->mega_win > 0 ? ctx->mega_win : (NT >= 88 ? 8 : (NT >= 56 ? 6 : WIN_DEFAULT)));
    const int first = std::min(win, std::max(1, ctx->mega_first_window));
    const int srows = ctx->mega_srows > 0 ? ctx->mega_srows : srows_auto;  // streamed tiles below each diagonal block
    // the edge regime (first / last block columns: chain-bound at every size, see mega::Edge); only where the middle differs from it
    static const int env_head = mrbf_env("MRBF_MEGA_HEAD") ? atoi(mrbf_env("MRBF_MEGA_HEAD")) : -1;
    static const int env_tail = mrbf_env("MRBF_MEGA_TAIL") ? atoi(mrbf_env("MRBF_MEGA_TAIL")) : -1;
    static const int env_reserve = mrbf_env("MRBF_MEGA_RESERVE") ? atoi(mrbf_env("MRBF_MEGA_RESERVE")) : -1;
    Edge edge{};
    edge.srows_edge = std::max(srows, 5);
    edge.pstream_edge = 2;
    {
        // (measured r04, alternating A/B: sixteen tail columns in the edge regime + 64-row bulk jobs in the last twenty block columns
        //  -- tail_half below --: n = 6144 / 8192 / 12288 / 16384: -2.5 / -5.3 / -2.6 / -1.2 %; the head alone: no change)
        const int head = env_head >= 0 ? env_head : 0, tail = env_tail >= 0 ? env_tail : 16;
        edge.head = (srows < edge.srows_edge) ? std::min(head, NT) : 0;
        edge.tail_c0 = (srows < edge.srows_edge) ? std::max(edge.head, NT - tail) : NT;
    }
    {
        // streamed tiles as 64-row halves: the last `MRBF_MEGA_SHALF` block columns (and the first MRBF_MEGA_SHALF_HEAD)
        edge.shalf = (sh_tail > 0 || sh_head > 0) ? 1 : 0;
        edge.sh_head = std::min(sh_head, NT);
        edge.sh_tail_c0 = std::max(edge.sh_head, NT - sh_tail);
        // panel tiles far below the diagonal as 128-row jobs (MRBF_MEGA_TFULL = block rows below the streamed ones that stay halves; -1: all halves)
        static const int env_tfull = mrbf_env("MRBF_MEGA_TFULL") ? atoi(mrbf_env("MRBF_MEGA_TFULL")) : -1;
        // (64-row halves keep a block row's column-to-column recurrence ahead of the chain; at n >= 12288 the rows more than eight below
        //  the streamed ones have the slack for 128-row jobs, whose GEMM loop shares the B operand between twice the MFMAs:
        //  alternating A/B r04: n = 12288 / 16384: -0.9 / -1.1 %, n = 10240: 0; all panel tiles full at n <= 8192: +4 .. +28 %)
        edge.tfull1 = env_tfull >= 0 ? env_tfull + 1 : (mrbf_env("MRBF_MEGA_TFULL") ? 0 : (NTq >= 96 ? 9 : 0));
        // rows below the square that are known to be zero (the fit's right-hand sides: k of the 128 rows of the extra block row)
        static const int env_xhalf = mrbf_env("MRBF_MEGA_XHALF") ? atoi(mrbf_env("MRBF_MEGA_XHALF")) : 1;
        edge.xhalf = (env_xhalf && MT == NT + 1 && ctx->mega_xreal > 0 && ctx->mega_xreal <= 64) ? 1 : 0;
    }
    const int srows_max = std::max(srows, (edge.head > 0 || edge.tail_c0 < NT) ? edge.srows_edge : srows);
    // job tables: one set per (NT, MT, schedule parameters), kept in a small per-context LRU -- Morbit's training sets grow and shrink
    // by a few sites between iterations, so n keeps crossing 128-boundaries back and forth; rebuilding the tables on every change
    // cost three copies and a stream synchronisation inside the factorisation phase
    static const int chainq = mrbf_env("MRBF_MEGA_CHAINQ") ? atoi(mrbf_env("MRBF_MEGA_CHAINQ")) : 0;
    static const int cboost = mrbf_env("MRBF_MEGA_CBOOST") ? atoi(mrbf_env("MRBF_MEGA_CBOOST")) : 12;
    // bulk jobs of the last block columns as 64-row halves (diagonal tiles included): there the machine runs empty and what is left
    // are per-tile chains of window updates -- each a K = 128 win GEMM of one workgroup, one after the other on the same tile --
    // that the diagonal chain ends up waiting for (job log r04: P(56) at n = 8192 waited 146 us for the last three windows of its
    // tile); two workgroups per tile halve every link.  n = 4096: -1.8 %, smaller: no change.
    static const int env_tail_half = mrbf_env("MRBF_MEGA_TAILHALF") ? atoi(mrbf_env("MRBF_MEGA_TAILHALF")) : -1;
    const int tail_half = env_tail_half >= 0 ? env_tail_half : (NT >= 32 ? 20 : 0);
    static const int env_tail_half_w = mrbf_env("MRBF_MEGA_TAILHALF_W") ? atoi(mrbf_env("MRBF_MEGA_TAILHALF_W")) : -1;
    const int tail_half_w = env_tail_half_w >= 0 ? env_tail_half_w : 1000;  // only the last so many window updates of such a tile
    const long tab_key = (chainq ? 50 : 0) + slack + 100000000000000L * tail_half + 10000000000000000L * std::min(tail_half_w, 99) + 100 * ctx->mega_half_cols + 10000 * slack_chain + 1000000 * first + 10000000 * win + 100000000 * (long)srows +
                         1000000000L * edge.head + 1000000000000L * edge.tail_c0;
    const long tab_key2 = (edge.shalf ? 1 + edge.sh_head + 1000L * edge.sh_tail_c0 : 0) + 1000000L * edge.tfull1 + 1000000000L * edge.xhalf;
    MegaTables *tab = nullptr;
    for (auto &t : ctx->mega_tables)
        if (t.nt == NT && t.mt == MT && t.key == tab_key && t.key2 == tab_key2) tab = &t;
    if (!tab) {
        std::vector<Job> pj, bj, cj;
        std::vector<int> wqs;
        build_job_tables(NT, MT, slack, slack_chain, first, win, srows, edge, ctx->mega_half_cols, tail_half, tail_half_w, chainq != 0, pj, bj, cj, wqs);
        if (ctx->mega_tables.size() >= 8) {  // evict the least recently used set (nothing on the stream may still read it)
            MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            size_t lru = 0;
            for (size_t t = 1; t < ctx->mega_tables.size(); ++t)
                if (ctx->mega_tables[t].stamp < ctx->mega_tables[lru].stamp) lru = t;
            (void)hipFree(ctx->mega_tables[lru].block);
            ctx->mega_tables.erase(ctx->mega_tables.begin() + lru);
        }
        MegaTables t{};
        t.nt = NT;
        t.mt = MT;
        t.key = tab_key;
        t.key2 = tab_key2;
        t.npanel = (int)pj.size();
        t.nbulk = (int)bj.size();
        t.nchainjobs = (int)cj.size();
        t.nwin = ((int)wqs.size() - 1) / 2;
        const size_t job_bytes = (pj.size() + bj.size() + cj.size() + 1) * sizeof(Job), wq_bytes = wqs.size() * sizeof(int);
        const size_t wq_off = (job_bytes + 255) & ~size_t(255);
        MRBF_HIP(ctx, hipMalloc(&t.block, wq_off + wq_bytes));
        // one host image, one copy (a synchronous hipMemcpy from pageable memory: the image may die right after it)
        std::vector<char> img(wq_off + wq_bytes, 0);
        Job *hj = (Job *)img.data();
        std::copy(pj.begin(), pj.end(), hj);
        std::copy(bj.begin(), bj.end(), hj + pj.size());
        std::copy(cj.begin(), cj.end(), hj + pj.size() + bj.size());
        std::memcpy(img.data() + wq_off, wqs.data(), wq_bytes);
        MRBF_HIP(ctx, hipMemcpy(t.block, img.data(), img.size(), hipMemcpyHostToDevice));
        t.jobs = t.block;
        t.wq = (char *)t.block + wq_off;
        ctx->mega_tables.push_back(t);
        tab = &ctx->mega_tables.back();
    }
    tab->stamp = ++ctx->mega_table_clock;
    ctx->mega_nt = NT;
    ctx->mega_mt = MT;
    ctx->mega_npanel = tab->npanel;
    ctx->mega_nbulk = tab->nbulk;
    ctx->mega_nchainjobs = tab->nchainjobs;
    ctx->mega_nwin = tab->nwin;
    Job *dj = (Job *)tab->jobs;
    a.cjobs = dj + ctx->mega_npanel + ctx->mega_nbulk;
    a.nchainjobs = ctx->mega_nchainjobs;
    a.pjobs = dj;
    a.npanel = ctx->mega_npanel;
    a.bjobs = dj + ctx->mega_npanel;
    a.nbulk = ctx->mega_nbulk;
    a.nwin = ctx->mega_nwin;
    a.wq_start = (int *)tab->wq;
    // flags: one block, zeroed before every launch
    const size_t nfl = ((size_t)CTL_WORDS + (size_t)QSTRIDE * (2 * a.nwin + 1) + (size_t)QSTRIDE * 512 + (size_t)QSTRIDE * (1 + 2 * srows_max) * NT + 2 * (size_t)MT * NT + 3) / 4 * 4;
    unsigned *fl;
    MRBF_TRY(get_buf(ctx, S_MEGA_FLAGS, nfl, &fl));
    hnow();  // 1: buffers / job tables
    MRBF_HIP(ctx, hipMemsetAsync(fl, 0, nfl * sizeof(unsigned), ctx->stream));
    hnow();  // 2: first memset enqueued
    if (!ctx->mega_info_clean) MRBF_HIP(ctx, hipMemsetAsync(dinfo, 0, sizeof(int), ctx->stream));  // (the fit zeroes its flag words itself, in front of everything)
    hnow();  // 3: second memset enqueued
    a.ctl = fl;
    a.wq_head = fl + CTL_WORDS;
    a.quiet = a.wq_head + (size_t)QSTRIDE * (2 * a.nwin + 1);
    a.dprog = a.quiet + (size_t)QSTRIDE * 512;
    a.sprog = a.dprog + (size_t)QSTRIDE * NT;
    a.tdone = a.sprog + (size_t)QSTRIDE * 2 * srows_max * NT;
    a.ucnt = a.tdone + (size_t)MT * NT;
    a.info = dinfo;
    a.nchain = ctx->mega_chain > 0 ? ctx->mega_chain : chain_auto;
    a.ndedicated = ctx->mega_dedicated;
    a.look = ctx->mega_look;
    a.use_quiet = ctx->mega_quiet;
    a.slack = slack;
    a.slack_chain = slack_chain;
    a.first = first;
    a.win = win;
    a.wbias = ctx->mega_wbias;
    a.srows = srows;
    a.edge = edge;
    a.cboost = cboost;
    {
        // chain jobs in front of the block columns `head` and `tail_c0 - 3` (the reserves are in place when the tail begins)
        int job = 0;
        a.head_job1 = 0;
        a.reserve_job0 = 1 << 30;
        // (edge columns here: those of the edge regime and those whose streamed tiles are halved -- twice the chain jobs per column)
        const int e_head = std::max(edge.head, edge.shalf ? edge.sh_head : 0), e_tail_c0 = std::min(edge.tail_c0, edge.shalf ? edge.sh_tail_c0 : NT);
        const int lead_c0 = std::max(e_head, e_tail_c0 - 3);
        for (int c = 0; c <= NT; ++c) {
            if (c == e_head) a.head_job1 = job;
            if (c == lead_c0 && e_tail_c0 < NT) a.reserve_job0 = job;
            if (c < NT) job += 1 + (shalf_at(c, srows, edge) ? 2 : 1) * std::min(srows_at(c, srows, edge), MT - 1 - c);
        }
        a.nreserve = (e_head > 0 || e_tail_c0 < NT) ? (env_reserve >= 0 ? env_reserve : 12) : 0;
    }
    {
        static const int env_pdma = mrbf_env("MRBF_MEGA_PANELDMA") ? atoi(mrbf_env("MRBF_MEGA_PANELDMA")) : 1;
        a.panel_dma = env_pdma;
    }
    a.pstream = ctx->mega_pstream > 0 ? std::min(ctx->mega_pstream, srows) : (NTq <= 48 && srows >= 2 ? 2 : 1);
    a.spin_ticks = (unsigned long long)std::max(1, ctx->spin_ms) * 100000ull;  // wall_clock64 runs at 100 MHz
    a.fault = (ctx->debug_fault & 1) && MT > 1;
    // small matrices: one workgroup per CU is plenty (and leaves room for other contexts' launches: mrbf_batch_run)
    const int grid = (NT <= 16 && ctx->mega_grid > 256) ? 256 : ctx->mega_grid;
    if (a.nchain < 1) a.nchain = 1;
    if (a.nchain + a.ndedicated >= grid) a.ndedicated = std::max(0, grid / 2 - a.nchain);
    {
        static const int env_xc = mrbf_env("MRBF_MEGA_XCHAIN") ? atoi(mrbf_env("MRBF_MEGA_XCHAIN")) : -1;
        const int xc = env_xc >= 0 ? env_xc : (a.nchain > 48 ? 4 : (a.nchain > 32 ? 2 : 1));
        a.xchain = (xc > 0 && 8 * a.nchain <= std::min(xc, 8) * grid) ? std::min(xc, 8) : 0;  // enough blocks = 0 .. xc-1 (mod 8) for the chain
        // (n = 16384: the CU partners of the 12 chain workgroups join the bulk work until 32 block columns are left: 27.0 -> 26.5 ms;
        //  at n <= 8192 the chain is never far from critical and pausing the partners throughout is as good or better)
        static const int qt = mrbf_env("MRBF_MEGA_QUIET_TAIL") ? atoi(mrbf_env("MRBF_MEGA_QUIET_TAIL")) : -1;
        a.quiet_tail = qt >= 0 ? qt : (NTq > 96 ? 32 : 1 << 20);
    }
    const char *trace_path = mrbf_env("MRBF_MEGA_TRACE");
    if (trace_path) {
        a.trace_dbg = mrbf_env("MRBF_MEGA_TRACE_WAVE") ? 12 + 16 * (atoi(mrbf_env("MRBF_MEGA_TRACE_WAVE")) & 3) : 4;
        MRBF_TRY(get_buf(ctx, S_MEGA_TRACE, (size_t)NT * 80 + 4 * 1024, &a.trace));
        MRBF_HIP(ctx, hipMemsetAsync(a.trace, 0, ((size_t)NT * 80 + 4 * 1024) * sizeof(unsigned long long), ctx->stream));
    }
    const char *jlog_path = mrbf_env("MRBF_MEGA_JLOG");
    if (jlog_path) {
        a.jlog_cap = ctx->mega_npanel + ctx->mega_nbulk + ctx->mega_nchainjobs + 16;
        MRBF_TRY(get_buf(ctx, S_MEGA_JLOG, (size_t)8 * a.jlog_cap + 16, &a.jlog));
        MRBF_HIP(ctx, hipMemsetAsync(a.jlog, 0, ((size_t)8 * a.jlog_cap + 16) * sizeof(unsigned long long), ctx->stream));
    }
    hipLaunchKernelGGL(potrf_mega_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    hnow();  // 4: persistent kernel enqueued
    if (jlog_path) {
        std::vector<unsigned long long> h((size_t)8 * a.jlog_cap + 16);
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MRBF_HIP(ctx, hipMemcpy(h.data(), a.jlog, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(jlog_path, "w")) {
            const unsigned long long n = std::min<unsigned long long>(h[0], (unsigned long long)a.jlog_cap);
            unsigned long long t0 = ~0ull;
            for (unsigned long long j = 0; j < n; ++j) t0 = std::min(t0, h[8 + 8 * j + 1]);
            for (unsigned long long j = 0; j < n; ++j) {
                const unsigned long long *r = &h[8 + 8 * j];
                fprintf(f, "%llu %llu %llu %llu %llu", r[0] & 0xff, (r[0] >> 8) & 0xffff, (r[0] >> 24) & 0xffff, (r[0] >> 40) & 0xfff, r[0] >> 52);
                for (int k = 1; k < 8; ++k) fprintf(f, " %.2f", r[k] ? (double)(r[k] - t0) * 0.01 : -1.0);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    MRBF_HIP(ctx, hipGetLastError());
    if (trace_path) {
        std::vector<unsigned long long> h((size_t)NT * 2 * 8);
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MRBF_HIP(ctx, hipMemcpy(h.data(), a.trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(trace_path, "w")) {
            const unsigned long long t0 = h[0];
            for (int c = 0; c < NT; ++c) {
                fprintf(f, "%d", c);
                for (int k = 0; k < 8; ++k) fprintf(f, " %.2f", h[(size_t)c * 16 + k] ? (double)(h[(size_t)c * 16 + k] - t0) * 0.01 : -1.0);
                for (int k = 8; k < 15; ++k) fprintf(f, " %.0f", (double)h[(size_t)c * 16 + k]);  // cycle counts, not time stamps
                fprintf(f, " %.2f", h[(size_t)c * 16 + 15] ? (double)(h[(size_t)c * 16 + 15] - t0) * 0.01 : -1.0);  // last panel published
                fprintf(f, "\n");
            }
            std::vector<unsigned long long> u(4 * 1024);
            (void)hipMemcpy(u.data(), a.trace + (size_t)NT * 16, u.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double tb = 0, tp = 0;
            unsigned long long nb = 0, np = 0;
            for (int g = 0; g < grid && g < 1024; ++g) {
                tb += (double)u[4 * g] * 0.01;
                tp += (double)u[4 * g + 1] * 0.01;
                nb += u[4 * g + 2];
                np += u[4 * g + 3];
            }
            {
                std::vector<unsigned long long> ss((size_t)NT * 64);
                (void)hipMemcpy(ss.data(), a.trace + (size_t)NT * 16 + 4 * 1024, ss.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                for (int c2 = 0; c2 < NT; ++c2) {
                    fprintf(f, "#S %d", c2);
                    for (int k2 = 0; k2 < 64; ++k2) {
                        if ((k2 & 7) > 5) continue;  // per step: seen, operands in, -, MFMAs issued, drained, published
                        fprintf(f, " %.2f", ss[(size_t)c2 * 64 + k2] ? (double)(ss[(size_t)c2 * 64 + k2] - t0) * 0.01 : -1.0);
                    }
                    fprintf(f, "\n");
                }
            }
            fprintf(f, "# util: bulk %.1f us over %llu jobs (%.1f us/job), panel %.1f us over %llu jobs (%.1f us/job), grid %d\n", tb, nb,
                    nb ? tb / nb : 0.0, tp, np, np ? tp / np : 0.0, grid);
            fclose(f);
        }
    }
    // a launch that gave up reports through dinfo as well
    unsigned long long *dstat;
    MRBF_TRY(get_buf(ctx, S_MEGA_STAT, (size_t)2, &dstat));
    hipLaunchKernelGGL(mega_status_kernel, dim3(1), dim3(1), 0, ctx->stream, (const unsigned *)fl, dinfo, dstat);
    ctx->mega_stat_dev = dstat;
    ctx->mega_stat_shape = (long)NT * 100000 + MT;
    ctx->mega_stat_pending = 1;
    hnow();  // 5: status kernel enqueued
    if (host_trace_ms > 0.0 && nh == 6 && hstamp[5] - hstamp[0] > host_trace_ms)
        fprintf(stderr, "potrf_mega_tall host trace (NT %d): buffers/tables %.3f ms | memset flags %.3f | memset info %.3f | kernel launch %.3f | status launch %.3f\n", NT,
                hstamp[1] - hstamp[0], hstamp[2] - hstamp[1], hstamp[3] - hstamp[2], hstamp[4] - hstamp[3], hstamp[5] - hstamp[4]);
    if (mrbf_env("MRBF_MEGA_DEBUG")) {
        unsigned h[CTL_WORDS];
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MRBF_HIP(ctx, hipMemcpy(h, fl, sizeof(h), hipMemcpyDeviceToHost));
        if (h[CTL_TIMEOUT]) {
            const long off = (long)h[CTL_TIMEOUT + 1];
            const long t0 = a.tdone - a.ctl, u0 = a.ucnt - a.ctl, d0 = a.dprog - a.ctl, s0 = a.sprog - a.ctl;
            fprintf(stderr, "mega timeout code 0x%x word %ld want %u: ", h[CTL_TIMEOUT], off, h[CTL_TIMEOUT + 2]);
            if (off >= u0) fprintf(stderr, "ucnt[%ld][%ld]", (off - u0) / NT, (off - u0) % NT);
            else if (off >= t0) fprintf(stderr, "tdone[%ld][%ld]", (off - t0) / NT, (off - t0) % NT);
            else if (off >= s0) fprintf(stderr, "sprog[%ld][%ld]", (off - s0) / QSTRIDE / NT, (off - s0) / QSTRIDE % NT);
            else if (off >= d0) fprintf(stderr, "dprog[%ld]", (off - d0) / QSTRIDE);
            fprintf(stderr, "  (NT %d MT %d pcols %u qpanel %u/%d qchain %u/%d)\n", NT, MT, h[CTL_PCOLS], h[CTL_QP], a.npanel, h[CTL_QC], a.nchainjobs);
        }
    }
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace mrbf

// Host-only: builds the job tables of a shape and checks their invariants.  out[0..3] = panel / bulk / chain jobs, windows;
// out[4] = order-dependent checksum of all jobs; out[5] = violated invariants (0 for every valid parameter set):
//   every tile (i, c), i >= c, is finished by exactly one chain job or by exactly two panel halves; it receives every window
//   w < nbulk_updates(i, c) exactly once (one full job or both halves, over the ordinary and the chain tiles' queue of that window
//   together) and no other; every window only updates tiles of block columns behind its own last panel; a queue lists its jobs by
//   ascending block column; the chain tiles' queues hold chain tiles only (and all of them when they are in use).
static int32_t check_mega_tables(int nt, int mt, int slack, int slack_chain, int first, int win, int srows, const mrbf::mega::Edge &edge,
                                 int half_cols, int tail_half, int tail_half_w, bool chainq, int64_t *out) {
    using namespace mrbf;
    using namespace mrbf::mega;
    std::vector<Job> pj, bj, cj;
    std::vector<int> wqs;
    build_job_tables(nt, mt, slack, slack_chain, first, win, srows, edge, half_cols, tail_half, tail_half_w, chainq, pj, bj, cj, wqs);
    int64_t bad = 0;
    auto sr = [&](int c) { return srows_at(c, srows, edge); };
    std::vector<int> fin((size_t)mt * nt, 0);
    for (const Job &j : cj) {
        if (j.i < j.c || j.i >= mt || j.c >= nt) {
            ++bad;
            continue;
        }
        fin[(size_t)j.i * nt + j.c] += (j.kind == JOB_S && j.w != 0) ? 1 : 2;  // a streamed tile: one 128-row job or two 64-row ones
        if (j.kind == JOB_P ? j.i != j.c : (j.kind != JOB_S || j.i <= j.c || j.i > j.c + sr(j.c) || j.w < 0 || j.w > 2 ||
                                            (j.w != 0) != shalf_at(j.c, srows, edge)))
            ++bad;
    }
    for (const Job &j : pj) {
        const bool xh = xhalf_at(j.i, nt, edge);
        if (j.kind != JOB_T || j.c >= nt || j.i <= j.c + sr(j.c) || j.i >= mt || j.w < 0 || j.w > 3 || (j.w == 3) != xh ||
            (!xh && (j.w == 2) != tfull_at(j.i, j.c, srows, edge))) {
            ++bad;
            continue;
        }
        fin[(size_t)j.i * nt + j.c] += j.w >= 2 ? 2 : 1;  // a panel tile: one 128-row job, two halves, or the upper half alone (the lower one is zero)
    }
    for (int c = 0; c < nt; ++c)
        for (int i = 0; i < mt; ++i) bad += fin[(size_t)i * nt + c] != (i >= c ? 2 : 0);
    const int nwin = ((int)wqs.size() - 1) / 2;  // queues [0, nwin): ordinary tiles, [nwin, 2 nwin): chain tiles
    std::vector<int> upd((size_t)mt * nt, 0);
    for (int w = 0; w < nwin; ++w) {
        std::fill(upd.begin(), upd.end(), 0);
        for (int cls = 0; cls < 2; ++cls) {
            int last_c = -1;
            for (int q = wqs[cls * nwin + w]; q < wqs[cls * nwin + w + 1]; ++q) {
                const Job &j = bj[q];
                if (j.i < j.c || j.i >= mt || j.c >= nt || (j.w & 255) != w || (j.kind != JOB_U && j.kind != JOB_UH)) {
                    ++bad;
                    continue;
                }
                if (j.c < last_c) ++bad;  // ascending block columns inside a queue
                last_c = j.c;
                if (j.c < wstart(w + 1, first, win)) ++bad;  // a window never reaches a block column it still belongs to
                if (((j.i - j.c <= sr(j.c)) && chainq) != (cls == 1)) ++bad;  // chain tiles in their own queues, nothing else there
                const int hcode = j.w >> 8;  // bulk halves: 0 / 1 = rows 0..63 / 64..127, 2 = the upper half of a tile whose lower half is zero
                if (j.kind == JOB_U ? hcode != 0 : (hcode > 2 || (hcode == 2) != xhalf_at(j.i, nt, edge))) ++bad;
                if (j.kind == JOB_U && xhalf_at(j.i, nt, edge)) ++bad;
                upd[(size_t)j.i * nt + j.c] += (j.kind == JOB_U || hcode == 2) ? 2 : 1;
            }
        }
        for (int c = 0; c < nt; ++c)
            for (int i = c; i < mt; ++i) bad += upd[(size_t)i * nt + c] != (w < nbulk_updates(i, c, slack, slack_chain, first, win, sr(c)) ? 2 : 0);
    }
    for (size_t w = 0; w + 1 < wqs.size(); ++w) bad += wqs[w] > wqs[w + 1];
    int64_t sum = 0;
    int64_t pos = 1;
    for (const std::vector<Job> *v : {&pj, &bj, &cj})
        for (const Job &j : *v) sum += (pos++) * (int64_t)(1 + j.kind + 7 * j.i + 131 * j.c + 1009 * j.w);
    out[0] = (int64_t)pj.size();
    out[1] = (int64_t)bj.size();
    out[2] = (int64_t)cj.size();
    out[3] = nwin;
    out[4] = sum;
    out[5] = bad;
    return 0;
}

extern "C" int32_t mrbf_debug_mega_tables(int32_t nt, int32_t mt, int32_t slack, int32_t slack_chain, int32_t first, int32_t win, int32_t srows,
                                          int32_t half_cols, int64_t *out) {
    if (nt < 1 || mt < nt || mt > 32000) return -1;
    if (slack < 1 || slack_chain < slack || win < 1 || first < 1 || first > win || srows < 0 || half_cols < 0) return -3;
    if (!out) return -9;
    const mrbf::mega::Edge no_edge{0, nt, srows, 1, 0};
    return check_mega_tables(nt, mt, slack, slack_chain, first, win, srows, no_edge, half_cols, 0, 0, false, out);
}

// The same with the round-4 options: opt[0] = head columns, opt[1] = tail columns of the edge regime (five streamed rows there),
// opt[2] = block columns at the end whose bulk jobs are 64-row halves, opt[3] = only a tile's last so many windows (0: all),
// opt[4] = chain tiles' window jobs in queues of their own.
extern "C" int32_t mrbf_debug_mega_tables2(int32_t nt, int32_t mt, int32_t slack, int32_t slack_chain, int32_t first, int32_t win, int32_t srows,
                                           int32_t half_cols, const int32_t *opt5, int64_t *out) {
    if (nt < 1 || mt < nt || mt > 32000) return -1;
    if (slack < 1 || slack_chain < slack || win < 1 || first < 1 || first > win || srows < 0 || half_cols < 0) return -3;
    if (!opt5) return -9;
    if (!out) return -10;
    if (opt5[0] < 0 || opt5[1] < 0 || opt5[2] < 0 || opt5[3] < 0) return -9;
    mrbf::mega::Edge edge{};
    edge.srows_edge = std::max(srows, 5);
    edge.pstream_edge = 2;
    edge.head = srows < edge.srows_edge ? std::min(opt5[0], nt) : 0;
    edge.tail_c0 = srows < edge.srows_edge ? std::max(edge.head, nt - opt5[1]) : nt;
    edge.shalf = (opt5[4] & 2) ? 1 : 0;  // (bit 1 of the last option: streamed tiles of five-row block columns as 64-row halves;
    edge.sh_head = (opt5[4] >> 2) & 0xff;  //  bits 2..9 / 10..17: only the first / last so many block columns, 0 / 0 = all)
    edge.sh_tail_c0 = ((opt5[4] >> 10) & 0xff) ? std::max(edge.sh_head, nt - ((opt5[4] >> 10) & 0xff)) : (edge.sh_head ? nt : 0);
    edge.xhalf = (opt5[4] >> 26) & 1;  // bit 26: the block rows below the square hold at most 64 non-zero rows
    edge.tfull1 = (opt5[4] >> 18) & 0xff;  // bits 18..25: panel tiles more than this - 1 block rows below the streamed ones as one 128-row job
    return check_mega_tables(nt, mt, slack, slack_chain, first, win, srows, edge, half_cols, opt5[2], opt5[3] > 0 ? opt5[3] : 1000, (opt5[4] & 1) != 0, out);
}
