// Internal declarations of libmrbf (not part of the ABI; the ABI is include/mrbf.h).
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mrbf.h"

namespace mrbf {

// ---- kernel parameters handed to device code -------------------------------------------
struct KP {
    int kid;
    double a, b;
    double a2;    // alpha^2 (gaussian / multiquadrics)
    double sgn;   // sign convention of the conditionally p.d. kernels
    int fast;     // 1: beta == 1/2 (multiquadrics) or beta == 3 (cubic): sqrt path instead of pow
    int ik;       // thin plate spline k
    double phi0;  // phi(0) = diagonal of Phi
};
KP make_kp(int kid, double a, double b);
int cpd_order(int kid, double a, double b);
inline int poly_dim(int d, int deg) { return deg < 0 ? 0 : (deg == 0 ? 1 : d + 1); }

// ---- workspace arena: named grow-only device buffers -----------------------------------
enum Slot {
    S_XC = 0, S_SQ, S_MEAN, S_PHI, S_PI, S_Q1, S_W1, S_G, S_R, S_TAU, S_RHS, S_T1, S_T2, S_IPIV, S_INFO,
    S_STAGE_A, S_STAGE_B, S_STAGE_C, S_STAGE_D, S_EVAL_E, S_EVAL_A, S_EVAL_J, S_EVAL_SA, S_EVAL_XC, S_EVAL_XSQ,
    S_OUT_A, S_OUT_B, S_CHOL_WS, S_MISC, S_MEGA_JOBS, S_MEGA_FLAGS, S_MEGA_WQ, S_MEGA_IT, S_BSOLVE_FLAGS, S_MEGA_TRACE, S_MEGA_JLOG,
    S_T1W, S_PS_STATE, S_PS_STAT, S_PS_POLISH, S_BSOLVE_X, S_V0, S_QR_INV, S_DIAG_SCR, S_SMALL_WS, S_SMALL_DESC, S_SMALL_FLAGS, S_MEGA_STAT, S_BSOLVE_M, S_SMALL_CL, S_GW_PART, S_GW_RS, S_GW_GP, S_GW_M, S_CHECK_SCAL, S_TAIL_PART, S_PS_RANK, S_NSLOTS
};
struct Buf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace mrbf

namespace mrbf {
// job tables of the persistent factorisation for one shape (chol_mega.hip), device resident
struct MegaTables {
    int nt, mt;
    long key, key2 = 0;
    void *block, *jobs, *wq;
    int npanel, nbulk, nchainjobs, nwin;
    unsigned long long stamp;
};
}  // namespace mrbf

struct mrbf_ctx {
    int device = 0;
    int ncu = 256;  // compute units of the device (queried once at init)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    rocblas_handle blas = nullptr;
    mrbf::Buf slots[mrbf::S_NSLOTS];
    hipEvent_t ev[8] = {};
    hipStream_t panel_stream = nullptr;  // high-priority side stream: diagonal block + panel of step j+1 under step j's trailing update
    hipEvent_t evx[4] = {};              // cross-stream dependencies of the look-ahead Cholesky
    hipStream_t bulk_stream = nullptr;   // CU-masked stream for the aggregated trailing updates: leaves CUs free for the panel chain
    int bulk_masked = 0;
    int bulk_grid = 384;  // > 0: cap on the workgroups of a bulk trailing update (persistent tile loop)
    // persistent factorisation (chol_mega.hip): cached job tables + launch geometry
    int mega_nt = 0, mega_mt = 0, mega_npanel = 0, mega_nbulk = 0, mega_nwin = 0, mega_nchainjobs = 0;
    int mega_grid = 512, mega_dedicated = 64, mega_look = 2, mega_min = 256, mega_max = 32768, mega_quiet = 1, mega_chain = 0, mega_slack = 3, mega_slack_chain = 0, mega_tab_slack = -1, mega_half_cols = 0, mega_first_window = 1, mega_win = 0, mega_wbias = 4, mega_pstream = 0, mega_srows = 0;  // chain, slack_chain, srows: 0 = by matrix size (potrf_mega_tall)
    std::string err;
    std::vector<mrbf::Buf> model_pool;  // released model blocks, reused by the next model of similar size (hipMalloc/hipFree cost
                                        // ~0.1-0.3 ms each and serialise across host threads)
    // options
    int gram_mode = 0, residual = 1, force_path = 0, chol_impl = 0, eval_impl = 0, timing = 1, diag_impl = 0, chol_window = 0;
    // bounded spins of the persistent kernels: wall-clock limit without progress (ms); debug_fault makes ONE workgroup of the named
    // kernel skip a publish so that the give-up paths can be tested deterministically (bit 0: factorisation, bit 1: backward substitution)
    int spin_ms = 1000, debug_fault = 0;
    // the persistent factorisation's own clock (chol_mega.hip): device time of the last launch, the shortest seen per shape, and the
    // number of launches of this context that took more than twice that (a stalled persistent launch must be visible to callers)
    unsigned long long *mega_stat_dev = nullptr;
    long mega_stat_shape = 0;
    int mega_stat_pending = 0;
    // pinned staging for small host buffers handed to / expected from the API (Julia and NumPy arrays): uploads are copied here and sent
    // asynchronously, downloads land here and are copied out behind the call's one stream synchronisation (pin_flush) -- a transfer
    // from / to pageable memory is a host round trip of its own.  Reset at the top of every API entry (each one ends synchronised).
    char *pin_base = nullptr;
    size_t pin_off = 0;
    bool pin_armed = false;  // between pin_reset and pin_flush of an API entry that guarantees both
    struct PinOut {
        void *user;
        const void *pin;
        size_t bytes;
    };
    std::vector<PinOut> pin_out;
    unsigned long long *hpin = nullptr;  // 64 pinned host words: the small read-backs of a fit (flags, shift, device clock) land here in one round trip
    int mega_info_clean = 0;  // the caller of the tall factorisation has zeroed *dinfo on the same stream already
    int mega_xreal = 0;  // > 0: the caller of the tall factorisation knows that only so many of the rows below the square are non-zero (the fit's right-hand sides)
    float last_device_ms = 0.f;
    int slow_launches = 0;
    int small_nc = 4;  // 1 after a cluster failure on this context (XCD placement, or a result that one workgroup does not reproduce)
    int small_cluster_ok = 0;    // the device is what the clusters' visibility argument assumes: gfx950, 8 XCDs x 32 CUs (context.hip)
    int small_timeouts = 0;      // barrier time-outs of clustered launches in a row (three: clusters off for this context)
    const double *eval_pre_xq = nullptr;  // the PS solver's next population already centred / padded in the evaluation's own query buffers (eval_fused skips its centring launch when they are these)
    int eval_population = 0;     // set around the PS solver's population sweeps (values only): eval_nsplit may split small models there
    int eval_check_call = 0;     // set around the residual check's evaluation at the model's own sites: eval_nsplit keeps the pre-round-5 rule for it (batch.hip does the same)
    int live_models = 0, live_round4 = 0;  // handles created through this context and not yet released (MRBF_OPT_LIVE_HANDLES)
    unsigned ps_rank_epoch = 0;  // launches of the PS ranking's wave kernel so far (part of the tag its exchanged records carry)
    int ps_multi_off = 0;        // the several-workgroup PS ranking timed out on this context (a device shared with other work): one workgroup per run from then on
    std::map<long, float> mega_best_ms;
    std::vector<mrbf::MegaTables> mega_tables;  // LRU of job tables, one set per shape
    unsigned long long mega_table_clock = 0;
};

struct mrbf_model {
    int64_t n = 0, npad = 0;
    int d = 0, dpad = 0, k = 0, q = 0, deg = -1;
    mrbf::KP kp{};
    double *C = nullptr;     // n x d original centres (row-major)
    double *Xc = nullptr;    // npad x dpad centred, zero padded
    double *sq = nullptr;    // npad squared norms of the centred rows
    double *mean = nullptr;  // dpad centroid
    double *W = nullptr;     // n x k row-major weights
    double *Wc = nullptr;    // npad x k column-major weights (zero padded) for the GEMM-shaped contractions
    double *lam = nullptr;   // q x k row-major
    void *block = nullptr;   // one device allocation carved into the arrays above
    size_t block_bytes = 0;
};

namespace mrbf {

// ---- error plumbing ---------------------------------------------------------------------
int fail(mrbf_ctx *ctx, int code, const char *fmt, ...);
#define MRBF_HIP(ctx, call)                                                                                  \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess)                                                                                \
            return mrbf::fail(ctx, e__ == hipErrorOutOfMemory ? MRBF_ENOMEM : MRBF_EHIP, "%s: %s (%s:%d)", #call, \
                              hipGetErrorString(e__), __FILE__, __LINE__);                                    \
    } while (0)
#define MRBF_BLAS(ctx, call)                                                                                 \
    do {                                                                                                      \
        rocblas_status s__ = (call);                                                                          \
        if (s__ != rocblas_status_success)                                                                    \
            return mrbf::fail(ctx, MRBF_EBLAS, "%s: rocblas status %d (%s:%d)", #call, (int)s__, __FILE__, __LINE__); \
    } while (0)
#define MRBF_TRY(call)             \
    do {                           \
        int rc__ = (call);         \
        if (rc__ != 0) return rc__; \
    } while (0)

int get_buf(mrbf_ctx *ctx, Slot s, size_t bytes, void **out);
template <class T>
inline int get_buf(mrbf_ctx *ctx, Slot s, size_t count, T **out) {
    return get_buf(ctx, s, count * sizeof(T), (void **)out);
}
bool is_device_ptr(const void *p);
// device view of a caller buffer: the pointer itself when it is device memory, else a staged copy
int stage_in(mrbf_ctx *ctx, Slot s, const double *user, size_t count, const double **dev);
// device buffer to produce an output into (user's own when device memory, else staging)
int stage_out(mrbf_ctx *ctx, Slot s, double *user, size_t count, double **dev);
int finish_out(mrbf_ctx *ctx, double *user, const double *dev, size_t count);
void pin_reset(mrbf_ctx *ctx);   // top of an API entry (through PinGuard)
void pin_flush(mrbf_ctx *ctx);   // behind the stream synchronisation that follows finish_out (through PinGuard::flush)
void pin_discard(mrbf_ctx *ctx); // an entry that leaves early: queued outputs dropped, block disarmed
char *pin_take(mrbf_ctx *ctx, size_t bytes);  // `bytes` of the armed staging block (64-byte aligned) or nullptr when they do not fit
// One guard per API entry that stages through the pinned block: arms it on construction; flush() copies the queued outputs to the
// user's buffers (call it behind the stream synchronisation that follows the last finish_out); an entry that returns before that --
// whatever the path -- discards them in the destructor.  fail() does not touch the block: an error that an entry recovers from
// (ENOMEM -> halves) leaves its queued outputs intact.
struct PinGuard {
    mrbf_ctx *ctx;
    bool flushed = false;
    explicit PinGuard(mrbf_ctx *c) : ctx(c) { pin_reset(c); }
    void flush() {
        pin_flush(ctx);
        flushed = true;
    }
    ~PinGuard() {
        if (!flushed) pin_discard(ctx);
    }
    PinGuard(const PinGuard &) = delete;
    PinGuard &operator=(const PinGuard &) = delete;
};
// word offsets into mrbf_ctx::hpin (64 pinned host words for the small read-backs, each landing with its caller's one synchronisation)
enum HpinSlot : int {
    HPIN_FIT_FLAGS = 0,    // solve.hip fit_chol: 4 ints
    HPIN_FIT_SCAL = 4,     // trace, shift (2 doubles) + the factorisation's flags
    HPIN_SMALL_FLAGS = 8,  // solve.hip one-launch fit: 4 ints
    HPIN_SMALL_SCAL = 12,  // 2 doubles
    HPIN_SMALL_CHK = 16,   // 4 doubles
    HPIN_RESIDUAL = 20,    // residual check: 3 doubles
    HPIN_R4_COUNT = 24,    // round4.hip: accepted so far / in this block (2 ints)
    HPIN_MEGA_STAT = 63,   // the persistent factorisation's device clock
};

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// EVERY environment switch of the library (MRBF_MEGA_*, MRBF_R4_*, MRBF_PS_*, MRBF_EVAL_*, ... -- schedule experiments, A/B forms kept for
// the record, diagnostics) is read through this function and is honoured only while MRBF_EXPERIMENTS=1 is set: a drop-in library must
// not change algorithm because of a stray variable in a user's environment.  Not cached (the tests toggle switches inside one process).
inline const char *mrbf_env(const char *name) {
    const char *gate = std::getenv("MRBF_EXPERIMENTS");
    if (!gate || std::atoi(gate) == 0) return nullptr;
    return std::getenv(name);
}

// ---- launchers implemented in the .hip files (all asynchronous on ctx->stream) ----------
// prep.hip
int launch_center_pad(mrbf_ctx *ctx, const double *X, int64_t n, int d, const double *mean_or_null, double *mean_out,
                      double *Xc, int64_t npad, int dpad, double *sq);
int launch_poly_matrix(mrbf_ctx *ctx, const double *C, int64_t n, int d, int q, double *Pi, int64_t ldpi);
int launch_transpose(mrbf_ctx *ctx, const double *in, int64_t rows, int64_t cols, double *out);  // in rows x cols row-major -> out col-major ld=rows... see impl
// gram.hip
int launch_gram(mrbf_ctx *ctx, int mode, const double *C, const double *Xc, const double *sq, int64_t n, int64_t npad,
                int d, int dpad, const KP &kp, double *Phi, int64_t ld);
int launch_cross_gram(mrbf_ctx *ctx, const double *X, int64_t m, const double *C, int64_t n, int d, const KP &kp, double *K);
// gram_fused.hip: the fit's own assembly (lower triangle + partial panels of Phi Xc), and the small kernels of the projection
bool gram_w_applies(const mrbf_ctx *ctx, const mrbf_model *M);
int launch_gram_w(mrbf_ctx *ctx, const mrbf_model *M, double *Phi, int64_t ld, double **Wpart_out, double **rspart_out, int *nseg_out);
int launch_projection_small(mrbf_ctx *ctx, const mrbf_model *M, const double *Wpart, const double *rspart, int nseg, const double *linv,
                            const double *Q1, double *W1, double *G, double *scal, int *flags, int K2, int skip, double *PA, double *PB,
                            double *v0);
// eval.hip
int eval_model(mrbf_ctx *ctx, const mrbf_model *M, int64_t m, const double *Xdev, double *vals_dev, double *jac_dev,
               mrbf_eval_info *info);
// chol.hip
int potrf_lower(mrbf_ctx *ctx, int impl, int64_t n, double *A, int64_t lda, int *info_host);
// solve.hip
int fit_model(mrbf_ctx *ctx, mrbf_model *M, const double *Ydev, mrbf_fit_info *info);
int build_model_shell(mrbf_ctx *ctx, int64_t n, int d, int k, const double *Cdev, int kid, double a, double b, int deg,
                      mrbf_model **out);
void destroy_model(mrbf_ctx *ctx, mrbf_model *M);
int fit_check(mrbf_ctx *ctx, mrbf_model *M, const double *Y, mrbf_fit_info *info);
// chol_mega.hip's launch clock: call after the stream has been synchronised behind a persistent factorisation
int mega_collect_stat(mrbf_ctx *ctx);
// the same in two halves: the download enqueued on the context's stream into pinned memory (nothing pending: no-op), evaluated after the caller's synchronisation
int mega_stat_enqueue(mrbf_ctx *ctx);
int mega_stat_finish(mrbf_ctx *ctx);

}  // namespace mrbf
