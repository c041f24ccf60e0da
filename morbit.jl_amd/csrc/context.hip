// Context, workspace arena, host<->device staging, options.
#include <cstdarg>

#include "common.hpp"

#include <cstring>

namespace mrbf {

static thread_local std::string g_init_err;

int fail(mrbf_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        ctx->err = buf;  // (the pinned staging block is the entry's PinGuard's business, not this function's)
    } else {
        g_init_err = buf;
    }
    return code;
}

int cpd_order(int kid, double a, double b) {
    switch (kid) {
        case MRBF_CUBIC: return (int)std::ceil(a / 2.0);
        case MRBF_MULTIQUADRIC: return (int)std::ceil(b);
        case MRBF_THIN_PLATE_SPLINE: return (int)a + 1;
        default: return 0;
    }
}

KP make_kp(int kid, double a, double b) {
    KP p{};
    p.kid = kid;
    p.a = a;
    p.b = b;
    p.a2 = a * a;
    p.sgn = 1.0;
    p.fast = 0;
    p.ik = 0;
    switch (kid) {
        case MRBF_GAUSSIAN: p.phi0 = 1.0; break;
        case MRBF_MULTIQUADRIC:
            p.sgn = ((int)std::ceil(b) & 1) ? -1.0 : 1.0;
            p.fast = (b == 0.5);
            p.phi0 = p.sgn;
            break;
        case MRBF_INV_MULTIQUADRIC:
            p.fast = (b == 0.5);
            p.phi0 = 1.0;
            break;
        case MRBF_CUBIC:
            p.sgn = ((int)std::ceil(a / 2.0) & 1) ? -1.0 : 1.0;
            p.fast = (a == 3.0);
            p.a2 = 0.0;
            p.phi0 = 0.0;
            break;
        case MRBF_THIN_PLATE_SPLINE:
            p.ik = (int)a;
            p.sgn = ((p.ik + 1) & 1) ? -1.0 : 1.0;
            p.a2 = 0.0;
            p.phi0 = 0.0;
            break;
    }
    return p;
}

int get_buf(mrbf_ctx *ctx, Slot s, size_t bytes, void **out) {
    Buf &b = ctx->slots[s];
    if (bytes == 0) bytes = 16;
    if (b.bytes < bytes) {
        if (b.p) {
            // a previous launch on the stream may still be using the old buffer
            MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            MRBF_HIP(ctx, hipFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
        }
        size_t want = bytes + bytes / 4;  // slack so slowly growing problems (a model gains a site per iteration: n^2 grows by ~1 % a call) do not realloc every call
        want = std::max<size_t>((want + 255) & ~size_t(255), size_t(64) << 10);  // small work buffers (candidate lists, per-call scalars) never regrow below 64 KB
        MRBF_HIP(ctx, hipMalloc(&b.p, want));
        b.bytes = want;
    }
    *out = b.p;
    return 0;
}

bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof(attr));
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'ed memory on older runtimes: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

constexpr size_t PIN_BYTES = (size_t)8 << 20, PIN_MAX = (size_t)2 << 20;  // staging block, largest single transfer that goes through it

int stage_in(mrbf_ctx *ctx, Slot s, const double *user, size_t count, const double **dev) {
    if (is_device_ptr(user)) {
        *dev = user;
        return 0;
    }
    double *b = nullptr;
    MRBF_TRY(get_buf(ctx, s, count, &b));
    const size_t bytes = count * sizeof(double);
    if (ctx->pin_base && ctx->pin_armed && bytes <= PIN_MAX && ctx->pin_off + bytes <= PIN_BYTES) {
        char *pin = ctx->pin_base + ctx->pin_off;
        ctx->pin_off += (bytes + 63) & ~(size_t)63;
        std::memcpy(pin, user, bytes);
        MRBF_HIP(ctx, hipMemcpyAsync(b, pin, bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        MRBF_HIP(ctx, hipMemcpyAsync(b, user, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    *dev = b;
    return 0;
}

int stage_out(mrbf_ctx *ctx, Slot s, double *user, size_t count, double **dev) {
    if (is_device_ptr(user)) {
        *dev = user;
        return 0;
    }
    return get_buf(ctx, s, count, dev);
}

int finish_out(mrbf_ctx *ctx, double *user, const double *dev, size_t count) {
    if (user == dev || user == nullptr) return 0;
    const size_t bytes = count * sizeof(double);
    if (ctx->pin_base && ctx->pin_armed && !is_device_ptr(user) && bytes <= PIN_MAX && ctx->pin_off + bytes <= PIN_BYTES) {
        char *pin = ctx->pin_base + ctx->pin_off;
        ctx->pin_off += (bytes + 63) & ~(size_t)63;
        MRBF_HIP(ctx, hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ctx->pin_out.push_back({user, pin, bytes});
        return 0;
    }
    MRBF_HIP(ctx, hipMemcpyAsync(user, dev, bytes, is_device_ptr(user) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

void pin_reset(mrbf_ctx *ctx) {
    ctx->pin_off = 0;
    ctx->pin_out.clear();
    ctx->pin_armed = true;
}
void pin_flush(mrbf_ctx *ctx) {
    for (const auto &o : ctx->pin_out) std::memcpy(o.user, o.pin, o.bytes);
    ctx->pin_out.clear();
    ctx->pin_armed = false;
}
void pin_discard(mrbf_ctx *ctx) {
    ctx->pin_out.clear();
    ctx->pin_armed = false;
}
char *pin_take(mrbf_ctx *ctx, size_t bytes) {
    if (!ctx->pin_base || !ctx->pin_armed || ctx->pin_off + bytes > PIN_BYTES) return nullptr;
    char *p = ctx->pin_base + ctx->pin_off;
    ctx->pin_off += (bytes + 63) & ~(size_t)63;
    return p;
}

static void mega_stat_account(mrbf_ctx *ctx, unsigned long long ticks);
int mega_stat_enqueue(mrbf_ctx *ctx) {
    if (!ctx->mega_stat_pending || !ctx->mega_stat_dev || !ctx->hpin) return 0;
    ctx->hpin[HPIN_MEGA_STAT] = ~0ull;
    MRBF_HIP(ctx, hipMemcpyAsync(&ctx->hpin[HPIN_MEGA_STAT], ctx->mega_stat_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    ctx->mega_stat_pending = 2;  // on its way
    return 0;
}
int mega_stat_finish(mrbf_ctx *ctx) {
    if (ctx->mega_stat_pending != 2) return mega_collect_stat(ctx);
    ctx->mega_stat_pending = 0;
    mega_stat_account(ctx, ctx->hpin[HPIN_MEGA_STAT]);
    return 0;
}
int mega_collect_stat(mrbf_ctx *ctx) {
    if (!ctx->mega_stat_pending || !ctx->mega_stat_dev) return 0;
    ctx->mega_stat_pending = 0;
    unsigned long long ticks = 0;
    MRBF_HIP(ctx, hipMemcpy(&ticks, ctx->mega_stat_dev, sizeof(ticks), hipMemcpyDeviceToHost));
    mega_stat_account(ctx, ticks);
    return 0;
}
static void mega_stat_account(mrbf_ctx *ctx, unsigned long long ticks) {
    const float ms = (float)((double)ticks * 1e-5);  // wall_clock64 runs at 100 MHz
    ctx->last_device_ms = ms;
    if (ms > 0.f) {
        auto it = ctx->mega_best_ms.find(ctx->mega_stat_shape);
        if (it == ctx->mega_best_ms.end()) {
            if (ctx->mega_best_ms.size() > 256) ctx->mega_best_ms.clear();
            ctx->mega_best_ms[ctx->mega_stat_shape] = ms;
        } else {
            if (ms > 2.f * it->second) ++ctx->slow_launches;
            if (ms < it->second) it->second = ms;
        }
    }
}

}  // namespace mrbf

using namespace mrbf;

extern "C" {

const char *mrbf_version(void) { return "mrbf 0.1.0 (gfx950)"; }

const char *mrbf_last_error(const mrbf_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_err.c_str(); }

int32_t mrbf_init(int32_t device_id, mrbf_ctx **out) {
    if (!out) return -2;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, MRBF_ENODEVICE, "no HIP device visible (%s); libmrbf has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess) device_id = 0;
    }
    if (device_id >= ndev) return fail(nullptr, -1, "device_id %d out of range (%d devices)", device_id, ndev);
    mrbf_ctx *ctx = new mrbf_ctx();
    ctx->device = device_id;
    auto bail = [&](int code, const char *what) {
        std::string msg = std::string(what) + " failed";
        delete ctx;
        return fail(nullptr, code, "%s", msg.c_str());
    };
    if (hipSetDevice(device_id) != hipSuccess) return bail(MRBF_EHIP, "hipSetDevice");
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess)
        return bail(MRBF_EHIP, "hipStreamCreate");
    ctx->stream = ctx->own_stream;
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->hpin), 64 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->hpin = nullptr;  // (read-backs then go through pageable memory as before)
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->pin_base), PIN_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->pin_base = nullptr;
    }
    if (rocblas_create_handle(&ctx->blas) != rocblas_status_success) return bail(MRBF_EBLAS, "rocblas_create_handle");
    rocblas_set_stream(ctx->blas, ctx->stream);
    rocblas_set_pointer_mode(ctx->blas, rocblas_pointer_mode_host);
    for (auto &ev : ctx->ev)
        if (hipEventCreate(&ev) != hipSuccess) return bail(MRBF_EHIP, "hipEventCreate");
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&ctx->panel_stream, hipStreamNonBlocking, hi) != hipSuccess)
            return bail(MRBF_EHIP, "hipStreamCreateWithPriority");
    }
    for (auto &ev : ctx->evx)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return bail(MRBF_EHIP, "hipEventCreate");
    {
        // bulk stream: every XCD keeps its last 4 CUs (32 of 256) out of the mask, so the panel chain's kernels
        // (1 workgroup for D, ~m/64 for T / U1) always find idle CUs while a rank-512 trailing update is running
        if (const char *bg = mrbf_env("MRBF_BULK_GRID")) ctx->bulk_grid = atoi(bg);
        if (const char *e = mrbf_env("MRBF_CHOL_IMPL")) ctx->chol_impl = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_GRID")) ctx->mega_grid = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_DEDICATED")) ctx->mega_dedicated = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_LOOK")) ctx->mega_look = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_MIN")) ctx->mega_min = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_QUIET")) ctx->mega_quiet = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_CHAIN")) ctx->mega_chain = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_SLACK")) ctx->mega_slack = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_SLACK_CHAIN")) ctx->mega_slack_chain = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_HALF_COLS")) ctx->mega_half_cols = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_FIRST_WINDOW")) ctx->mega_first_window = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_WIN")) ctx->mega_win = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_WBIAS")) ctx->mega_wbias = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_SROWS")) ctx->mega_srows = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_PSTREAM")) ctx->mega_pstream = atoi(e);
        if (const char *e = mrbf_env("MRBF_MEGA_MAX")) ctx->mega_max = atoi(e);
        if (const char *e = mrbf_env("MRBF_SPIN_MS")) ctx->spin_ms = std::max(1, atoi(e));
        hipDeviceProp_t prop;
        int ncu = 256;
        bool gfx950 = false;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
            ncu = prop.multiProcessorCount;
            gfx950 = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        }
        ctx->ncu = ncu;
        // workgroup clusters of the small fit (small.hip) rest on 8 XCDs x 32 CUs with round-robin block placement: only there
        ctx->small_cluster_ok = (gfx950 && ncu == 256) ? 1 : 0;
        const char *env = mrbf_env("MRBF_BULK_RESERVE");
        const int reserve_per_32 = env ? atoi(env) : 0;  // measured: masking costs more than it buys
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int cu = 0; cu < ncu; ++cu)
            if ((cu % 32) < 32 - reserve_per_32) mask[cu / 32] |= (1u << (cu % 32));
        if (reserve_per_32 > 0 && hipExtStreamCreateWithCUMask(&ctx->bulk_stream, (uint32_t)mask.size(), mask.data()) == hipSuccess) {
            ctx->bulk_masked = 1;
        } else {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&ctx->bulk_stream, hipStreamNonBlocking) != hipSuccess) return bail(MRBF_EHIP, "hipStreamCreate");
        }
    }
    *out = ctx;
    return MRBF_OK;
}

int32_t mrbf_shutdown(mrbf_ctx *ctx) {
    if (!ctx) return -1;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->slots)
        if (b.p) (void)hipFree(b.p);
    for (auto &b : ctx->model_pool)
        if (b.p) (void)hipFree(b.p);
    for (auto &t : ctx->mega_tables)
        if (t.block) (void)hipFree(t.block);
    if (ctx->hpin) (void)hipHostFree(ctx->hpin);
    if (ctx->pin_base) (void)hipHostFree(ctx->pin_base);
    for (auto &ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->evx)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->panel_stream) {
        (void)hipStreamSynchronize(ctx->panel_stream);
        (void)hipStreamDestroy(ctx->panel_stream);
    }
    if (ctx->bulk_stream) {
        (void)hipStreamSynchronize(ctx->bulk_stream);
        (void)hipStreamDestroy(ctx->bulk_stream);
    }
    if (ctx->blas) rocblas_destroy_handle(ctx->blas);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MRBF_OK;
}

int32_t mrbf_set_option(mrbf_ctx *ctx, int32_t key, double value) {
    if (!ctx) return -1;
    int v = (int)value;
    switch (key) {
        case MRBF_OPT_GRAM_MODE: ctx->gram_mode = v; break;
        case MRBF_OPT_RESIDUAL: ctx->residual = v; break;
        case MRBF_OPT_FORCE_PATH: ctx->force_path = v; break;
        case MRBF_OPT_CHOL_IMPL: ctx->chol_impl = v; break;
        case MRBF_OPT_EVAL_IMPL: ctx->eval_impl = v; break;
        case MRBF_OPT_TIMING: ctx->timing = v; break;
        case MRBF_OPT_DIAG_IMPL: ctx->diag_impl = v; break;
        case MRBF_OPT_CHOL_WINDOW: ctx->chol_window = v; break;
        case MRBF_OPT_SPIN_MS: ctx->spin_ms = v > 0 ? v : 1000; break;
        case MRBF_OPT_DEBUG_FAULT: ctx->debug_fault = v; break;
        default: return fail(ctx, -2, "unknown option key %d", key);
    }
    return MRBF_OK;
}

int32_t mrbf_get_option(const mrbf_ctx *ctx, int32_t key, double *value) {
    if (!ctx) return -1;
    if (!value) return -3;
    switch (key) {
        case MRBF_OPT_GRAM_MODE: *value = ctx->gram_mode; break;
        case MRBF_OPT_RESIDUAL: *value = ctx->residual; break;
        case MRBF_OPT_FORCE_PATH: *value = ctx->force_path; break;
        case MRBF_OPT_CHOL_IMPL: *value = ctx->chol_impl; break;
        case MRBF_OPT_EVAL_IMPL: *value = ctx->eval_impl; break;
        case MRBF_OPT_TIMING: *value = ctx->timing; break;
        case MRBF_OPT_DIAG_IMPL: *value = ctx->diag_impl; break;
        case MRBF_OPT_CHOL_WINDOW: *value = ctx->chol_window; break;
        case MRBF_OPT_SPIN_MS: *value = ctx->spin_ms; break;
        case MRBF_OPT_DEBUG_FAULT: *value = ctx->debug_fault; break;
        case MRBF_OPT_LAST_DEVICE_MS: *value = ctx->last_device_ms; break;
        case MRBF_OPT_SLOW_LAUNCHES: *value = ctx->slow_launches; break;
        case MRBF_OPT_ARENA_BYTES: {
            double bytes = 0.0;
            for (int i = 0; i < S_NSLOTS; ++i) bytes += (double)ctx->slots[i].bytes;
            for (auto &b : ctx->model_pool) bytes += (double)b.bytes;
            *value = bytes;
            break;
        }
        case MRBF_OPT_LIVE_HANDLES: *value = ctx->live_models + ctx->live_round4; break;
        default: return -2;
    }
    return MRBF_OK;
}

int32_t mrbf_set_stream(mrbf_ctx *ctx, void *hip_stream) {
    if (!ctx) return -1;
    (void)hipSetDevice(ctx->device);
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    MRBF_BLAS(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
    return MRBF_OK;
}

int32_t mrbf_sync(mrbf_ctx *ctx) {
    if (!ctx) return -1;
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MRBF_OK;
}

}  // extern "C"
