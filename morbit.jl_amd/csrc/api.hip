// extern "C" entry points of include/mrbf.h (argument checking, staging of host buffers, dispatch).
#include <algorithm>
#include <mutex>
#include <thread>

#include "common.hpp"

using namespace mrbf;

static int check_kernel(mrbf_ctx *ctx, int kid, double a, double b, int deg, int argpos) {
    if (kid < 0 || kid > 4) return fail(ctx, -argpos, "kernel_id %d not in 0..4 (Morbit.RbfKernels)", kid);
    // same sanity checks as RbfConfig's @assert block (RbfModel.jl:102-111)
    if (kid == MRBF_CUBIC && !(a >= 1.0 && std::fmod(a, 2.0) == 1.0))
        return fail(ctx, -(argpos + 1), "cubic exponent must be an odd integer >= 1 (got %g)", a);
    if (kid == MRBF_THIN_PLATE_SPLINE && !(a >= 1.0 && std::fmod(a, 1.0) == 0.0 && a <= 16.0))
        return fail(ctx, -(argpos + 1), "thin plate spline k must be an integer >= 1 (got %g)", a);
    if ((kid == MRBF_GAUSSIAN || kid == MRBF_MULTIQUADRIC || kid == MRBF_INV_MULTIQUADRIC) && !(a > 0.0))
        return fail(ctx, -(argpos + 1), "shape parameter must be strictly positive (got %g)", a);
    if (kid == MRBF_MULTIQUADRIC && !(b > 0.0 && std::fmod(b, 1.0) != 0.0))
        return fail(ctx, -(argpos + 2), "multiquadric exponent must be positive and non-integer (got %g)", b);
    if (kid == MRBF_INV_MULTIQUADRIC && !(b > 0.0))
        return fail(ctx, -(argpos + 2), "inverse multiquadric exponent must be positive (got %g)", b);
    if (deg < -1 || deg > 1) return fail(ctx, -(argpos + 3), "polynomial_degree must be -1, 0 or 1 (RbfModel.jl:21), got %d", deg);
    return 0;
}

// contexts of mrbf_batch_run's worker threads are kept between calls (creating a rocBLAS handle and streams costs tens of
// milliseconds, more than a whole small problem)
static std::mutex g_pool_mutex;
static std::vector<mrbf_ctx *> g_ctx_pool;
static mrbf_ctx *pool_acquire(int device, int *rc) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        for (size_t i = 0; i < g_ctx_pool.size(); ++i)
            if (g_ctx_pool[i]->device == device) {
                mrbf_ctx *c = g_ctx_pool[i];
                g_ctx_pool.erase(g_ctx_pool.begin() + i);
                *rc = 0;
                return c;
            }
    }
    mrbf_ctx *c = nullptr;
    *rc = mrbf_init(device, &c);
    return c;
}
static void pool_release(mrbf_ctx *c) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (g_ctx_pool.size() < 64)
        g_ctx_pool.push_back(c);
    else
        mrbf_shutdown(c);
}

namespace mrbf {
// the per-problem launch chain for the problems listed in `which` (anything the batched small-problem path of batch.hip does not
// take: large n, d > 128, LU / minimum-norm paths): worker threads x contexts, problem p on GPU p % n_dev
int batch_run_chain(int n_dev, const int *devs, const std::vector<int64_t> &which, const mrbf_problem *problems, mrbf_result *results) {
    if (which.empty()) return MRBF_OK;
    // small problems are launch-latency bound: several host threads (each with its own context and streams) per GPU keep
    // more kernels in flight; large problems fill the chip on their own
    int64_t nmax = 0;
    for (int64_t p : which) nmax = std::max<int64_t>(nmax, problems[p].n);
    int per_dev = (nmax <= 2048) ? 4 : 1;
    if (const char *e = mrbf_env("MRBF_BATCH_WORKERS")) per_dev = std::max(1, atoi(e));
    per_dev = (int)std::max<int64_t>(1, std::min<int64_t>(per_dev, ((int64_t)which.size() + n_dev - 1) / n_dev));
    std::vector<std::vector<int64_t>> per_gpu(n_dev);
    for (int64_t p : which) per_gpu[p % n_dev].push_back(p);
    const int n_workers = n_dev * per_dev;
    std::vector<int> rcs(n_workers, 0);
    auto worker = [&](int w) {
        const int g = w % n_dev, lane = w / n_dev;
        const std::vector<int64_t> &mine = per_gpu[g];
        int rc = 0;
        mrbf_ctx *ctx = pool_acquire(devs[g], &rc);
        if (rc != 0) {
            rcs[w] = rc;
            for (size_t j = lane; j < mine.size(); j += per_dev) {
                std::memset(&results[mine[j]], 0, sizeof(mrbf_result));
                results[mine[j]].status = rc;
                results[mine[j]].device = devs[g];
            }
            return;
        }
        for (size_t j = lane; j < mine.size(); j += per_dev) {  // round-robin shard (problem p -> GPU p % n_dev), no exchange
            const int64_t p = mine[j];
            const mrbf_problem &pr = problems[p];
            mrbf_result &res = results[p];
            std::memset(&res, 0, sizeof(res));
            res.device = devs[g];
            mrbf_model *M = nullptr;
            std::vector<double> wtmp, vtmp;
            double *wout = pr.weights_out, *vout = pr.vals_out;
            if (!wout || is_device_ptr(wout)) {
                wtmp.resize((size_t)pr.n * pr.k);
                wout = wtmp.data();
            }
            res.status = mrbf_fit(ctx, pr.n, pr.d, pr.k, pr.centres, pr.values, pr.kernel_id, pr.a, pr.b, pr.poly_deg, &M,
                                  wout, pr.poly_out, &res.fit);
            if (res.status == 0 && wout != pr.weights_out && pr.weights_out)
                (void)hipMemcpy(pr.weights_out, wout, wtmp.size() * sizeof(double), hipMemcpyDefault);
            if (res.status == 0 && pr.m > 0) {
                if (!vout || is_device_ptr(vout)) {
                    vtmp.resize((size_t)pr.m * pr.k);
                    vout = vtmp.data();
                }
                mrbf_eval_info ei;
                res.status = mrbf_eval(ctx, M, pr.m, pr.X, vout, pr.jac_out, &ei);
                res.ms_eval = ei.ms_total;
                if (res.status == 0) {
                    for (size_t i = 0; i < (size_t)pr.m * pr.k; ++i) res.checksum_vals += vout[i];  // (host sums on this path)
                    if (vout != pr.vals_out && pr.vals_out) (void)hipMemcpy(pr.vals_out, vout, vtmp.size() * sizeof(double), hipMemcpyDefault);
                }
            }
            if (res.status == 0 || M)
                for (size_t i = 0; M && i < (size_t)pr.n * pr.k; ++i) res.checksum_w += wout[i];
            if (M) mrbf_free_model(ctx, M);
        }
        pool_release(ctx);
    };
    std::vector<std::thread> th;
    for (int w = 0; w < n_workers; ++w) th.emplace_back(worker, w);
    for (auto &t : th) t.join();
    for (int w = 0; w < n_workers; ++w)
        if (rcs[w] != 0) return rcs[w];
    return MRBF_OK;
}

mrbf_ctx *batch_pool_acquire(int device, int *rc) { return pool_acquire(device, rc); }
void batch_pool_release(mrbf_ctx *c) { pool_release(c); }
}  // namespace mrbf

extern "C" {

int32_t mrbf_gram(mrbf_ctx *ctx, int64_t n, int32_t d, const double *centres, int32_t kernel_id, double a, double b,
                  int32_t poly_deg, double *Phi_out, double *Pi_out, float *ms) {
    if (!ctx) return -1;
    if (n < 0 || n > 46000) return fail(ctx, -2, "n = %lld out of range", (long long)n);
    if (d < 1 || d > 4096) return fail(ctx, -3, "d = %d out of range", d);
    if (!centres && n > 0) return fail(ctx, -4, "centres is NULL");
    MRBF_TRY(check_kernel(ctx, kernel_id, a, b, poly_deg, 5));
    if (!Phi_out) return fail(ctx, -9, "Phi_out is NULL");
    if (n == 0) return MRBF_OK;
    (void)hipSetDevice(ctx->device);
    PinGuard pin(ctx);
    const int q = poly_dim(d, poly_deg);
    const int64_t npad = round_up(n, 128);
    const int dpad = (int)round_up(d, 16);
    const KP kp = make_kp(kernel_id, a, b);
    const double *C;
    double *Xc, *sq, *mean, *Phi, *Pi = nullptr;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, centres, (size_t)n * d, &C));
    MRBF_TRY(get_buf(ctx, S_XC, (size_t)npad * dpad, &Xc));
    MRBF_TRY(get_buf(ctx, S_SQ, (size_t)npad, &sq));
    MRBF_TRY(get_buf(ctx, S_MEAN, (size_t)dpad, &mean));
    MRBF_TRY(stage_out(ctx, S_PHI, Phi_out, (size_t)n * n, &Phi));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (ctx->gram_mode != 1) MRBF_TRY(launch_center_pad(ctx, C, n, d, nullptr, mean, Xc, npad, dpad, sq));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MRBF_TRY(launch_gram(ctx, ctx->gram_mode, C, Xc, sq, n, npad, d, dpad, kp, Phi, n));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (Pi_out && q > 0) {
        MRBF_TRY(stage_out(ctx, S_PI, Pi_out, (size_t)n * q, &Pi));
        MRBF_TRY(launch_poly_matrix(ctx, C, n, d, q, Pi, n));
    }
    MRBF_TRY(finish_out(ctx, Phi_out, Phi, (size_t)n * n));
    if (Pi) MRBF_TRY(finish_out(ctx, Pi_out, Pi, (size_t)n * q));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pin.flush();
    if (ms) MRBF_HIP(ctx, hipEventElapsedTime(ms, ctx->ev[1], ctx->ev[2]));
    return MRBF_OK;
}

int32_t mrbf_cross_gram(mrbf_ctx *ctx, int64_t m, int64_t n, int32_t d, const double *X, const double *centres, int32_t kernel_id,
                        double a, double b, double *K_out) {
    if (!ctx) return -1;
    if (m < 0 || m > (int64_t)1 << 24) return fail(ctx, -2, "m out of range");
    if (n < 0 || n > (int64_t)1 << 24) return fail(ctx, -3, "n out of range");
    if (d < 1 || d > 4096) return fail(ctx, -4, "d out of range");
    MRBF_TRY(check_kernel(ctx, kernel_id, a, b, 1, 7));
    if (m == 0 || n == 0) return MRBF_OK;
    if (!X) return fail(ctx, -5, "X is NULL");
    if (!centres) return fail(ctx, -6, "centres is NULL");
    if (!K_out) return fail(ctx, -10, "K_out is NULL");
    (void)hipSetDevice(ctx->device);
    PinGuard pin(ctx);
    const double *dX, *dC;
    double *dK;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, X, (size_t)m * d, &dX));
    MRBF_TRY(stage_in(ctx, S_STAGE_B, centres, (size_t)n * d, &dC));
    MRBF_TRY(stage_out(ctx, S_PHI, K_out, (size_t)m * n, &dK));
    MRBF_TRY(launch_cross_gram(ctx, dX, m, dC, n, d, make_kp(kernel_id, a, b), dK));
    MRBF_TRY(finish_out(ctx, K_out, dK, (size_t)m * n));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pin.flush();
    return MRBF_OK;
}

int32_t mrbf_fit(mrbf_ctx *ctx, int64_t n, int32_t d, int32_t k, const double *centres, const double *values,
                 int32_t kernel_id, double a, double b, int32_t poly_deg, mrbf_model **model, double *weights_out,
                 double *poly_out, mrbf_fit_info *info) {
    if (!ctx) return -1;
    if (n < 1 || n > 46000) return fail(ctx, -2, "n = %lld out of range", (long long)n);
    if (d < 1 || d > 4096) return fail(ctx, -3, "d = %d out of range", d);
    if (k < 1 || k > 1024) return fail(ctx, -4, "k = %d out of range", k);
    if (!centres) return fail(ctx, -5, "centres is NULL");
    if (!values) return fail(ctx, -6, "values is NULL");
    MRBF_TRY(check_kernel(ctx, kernel_id, a, b, poly_deg, 7));
    if (!model) return fail(ctx, -11, "model is NULL");
    *model = nullptr;
    (void)hipSetDevice(ctx->device);
    PinGuard pin(ctx);
    const double *C, *Y;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, centres, (size_t)n * d, &C));
    MRBF_TRY(stage_in(ctx, S_STAGE_B, values, (size_t)n * k, &Y));
    mrbf_model *M = nullptr;
    MRBF_TRY(build_model_shell(ctx, n, d, k, C, kernel_id, a, b, poly_deg, &M));
    int rc = fit_model(ctx, M, Y, info);
    if (rc != 0) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_model(ctx, M);
        return rc;
    }
    auto copy_out = [&]() -> int {
        if (weights_out) MRBF_TRY(finish_out(ctx, weights_out, M->W, (size_t)n * k));
        if (poly_out && M->q > 0) MRBF_TRY(finish_out(ctx, poly_out, M->lam, (size_t)M->q * k));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        pin.flush();
        return 0;
    };
    rc = copy_out();
    if (rc != 0) {  // the model is not handed out: release it
        destroy_model(ctx, M);
        return rc;
    }
    *model = M;
    return MRBF_OK;
}

int32_t mrbf_model_from_coeffs(mrbf_ctx *ctx, int64_t n, int32_t d, int32_t k, const double *centres,
                               const double *weights, const double *poly, int32_t kernel_id, double a, double b,
                               int32_t poly_deg, mrbf_model **model) {
    if (!ctx) return -1;
    if (n < 1 || n > 46000) return fail(ctx, -2, "n out of range");
    if (d < 1 || d > 4096) return fail(ctx, -3, "d out of range");
    if (k < 1 || k > 1024) return fail(ctx, -4, "k out of range");
    if (!centres) return fail(ctx, -5, "centres is NULL");
    if (!weights) return fail(ctx, -6, "weights is NULL");
    MRBF_TRY(check_kernel(ctx, kernel_id, a, b, poly_deg, 8));
    const int q = poly_dim(d, poly_deg);
    if (q > 0 && !poly) return fail(ctx, -7, "poly is NULL but polynomial_degree >= 0");
    if (!model) return fail(ctx, -12, "model is NULL");
    (void)hipSetDevice(ctx->device);
    const double *C, *W, *L = nullptr;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, centres, (size_t)n * d, &C));
    MRBF_TRY(stage_in(ctx, S_STAGE_B, weights, (size_t)n * k, &W));
    if (q > 0) MRBF_TRY(stage_in(ctx, S_STAGE_C, poly, (size_t)q * k, &L));
    mrbf_model *M = nullptr;
    MRBF_TRY(build_model_shell(ctx, n, d, k, C, kernel_id, a, b, poly_deg, &M));
    auto fill = [&]() -> int {
        MRBF_HIP(ctx, hipMemcpyAsync(M->W, W, (size_t)n * k * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        MRBF_HIP(ctx, hipMemsetAsync(M->Wc, 0, (size_t)M->npad * k * sizeof(double), ctx->stream));
        for (int l = 0; l < k; ++l)  // strided copy W[:, l] -> Wc[:, l]
            MRBF_HIP(ctx, hipMemcpy2DAsync(M->Wc + (size_t)l * M->npad, sizeof(double), W + l, (size_t)k * sizeof(double),
                                           sizeof(double), (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        if (q > 0) MRBF_HIP(ctx, hipMemcpyAsync(M->lam, L, (size_t)q * k * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    const int rc = fill();
    if (rc != 0) {
        destroy_model(ctx, M);
        return rc;
    }
    *model = M;
    return MRBF_OK;
}

int32_t mrbf_eval(mrbf_ctx *ctx, const mrbf_model *model, int64_t m, const double *X, double *vals_out, double *jac_out,
                  mrbf_eval_info *info) {
    if (!ctx) return -1;
    if (!model) return fail(ctx, -2, "model is NULL");
    if (m < 0) return fail(ctx, -3, "m < 0");
    if (info) std::memset(info, 0, sizeof(*info));
    if (m == 0) return MRBF_OK;
    if (!X) return fail(ctx, -4, "X is NULL");
    (void)hipSetDevice(ctx->device);
    PinGuard pin(ctx);
    const int d = model->d, k = model->k;
    const double *Xd;
    double *V = nullptr, *J = nullptr;
    MRBF_TRY(stage_in(ctx, S_STAGE_C, X, (size_t)m * d, &Xd));
    if (vals_out) MRBF_TRY(stage_out(ctx, S_OUT_A, vals_out, (size_t)m * k, &V));
    if (jac_out) MRBF_TRY(stage_out(ctx, S_OUT_B, jac_out, (size_t)m * k * d, &J));
    MRBF_TRY(eval_model(ctx, model, m, Xd, V, J, info));
    if (V) MRBF_TRY(finish_out(ctx, vals_out, V, (size_t)m * k));
    if (J) MRBF_TRY(finish_out(ctx, jac_out, J, (size_t)m * k * d));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pin.flush();
    return MRBF_OK;
}

int32_t mrbf_backtrack(mrbf_ctx *ctx, const mrbf_model *model, const double *x, const double *dir, double step0,
                       double omega, int32_t strict, double const_rhs, double shrink, double min_stepsize,
                       int32_t max_loops, double *x_plus, double *mx_plus, double *step, int32_t *n_loops) {
    if (!ctx) return -1;
    if (!model) return fail(ctx, -2, "model is NULL");
    if (!x) return fail(ctx, -3, "x is NULL");
    if (!dir) return fail(ctx, -4, "dir is NULL");
    if (!(shrink > 0.0 && shrink < 1.0)) return fail(ctx, -9, "shrink must be in (0,1)");
    if (max_loops < 0 || max_loops > 100000) return fail(ctx, -11, "max_loops out of range");
    (void)hipSetDevice(ctx->device);
    const int d = model->d, k = model->k;
    std::vector<double> hx(d), hd(d);
    auto fetch = [&](const double *src, double *dst) -> int {
        if (is_device_ptr(src)) {
            MRBF_HIP(ctx, hipMemcpy(dst, src, d * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::memcpy(dst, src, d * sizeof(double));
        }
        return 0;
    };
    MRBF_TRY(fetch(x, hx.data()));
    MRBF_TRY(fetch(dir, hd.data()));
    // step sizes exactly as the sequential loop produces them: step_size *= alpha (descent.jl:176)
    const int L = max_loops + 1;
    std::vector<double> steps(L);
    steps[0] = step0;
    for (int i = 1; i < L; ++i) steps[i] = steps[i - 1] * shrink;
    // batch: row 0 = x, row 1+i = x .+ steps[i] .* dir   (descent.jl:162, :177)
    std::vector<double> Xb((size_t)(L + 1) * d), Vb((size_t)(L + 1) * k);
    for (int t = 0; t < d; ++t) Xb[t] = hx[t];
    for (int i = 0; i < L; ++i)
        for (int t = 0; t < d; ++t) Xb[(size_t)(i + 1) * d + t] = hx[t] + steps[i] * hd[t];
    int rc = mrbf_eval(ctx, model, L + 1, Xb.data(), Vb.data(), nullptr, nullptr);
    if (rc != 0) return rc;
    // scan with the reference's loop logic (descent.jl:166-180)
    const double *mx = Vb.data();
    int i = 0;
    while (i < max_loops) {
        const double *mp = Vb.data() + (size_t)(i + 1) * k;
        bool ok;
        if (strict) {
            ok = true;
            for (int l = 0; l < k; ++l) ok = ok && ((mx[l] - mp[l]) >= steps[i] * const_rhs * omega);
        } else {
            ok = (*std::max_element(mx, mx + k) - *std::max_element(mp, mp + k)) >= steps[i] * const_rhs * omega;
        }
        if (ok) break;
        if (steps[i] <= min_stepsize) break;
        ++i;
    }
    std::vector<double> hs(d);
    for (int t = 0; t < d; ++t) hs[t] = steps[i] * hd[t];
    auto put = [&](double *dst, const double *src, size_t cnt) -> int {
        if (!dst) return 0;
        if (is_device_ptr(dst)) {
            MRBF_HIP(ctx, hipMemcpy(dst, src, cnt * sizeof(double), hipMemcpyHostToDevice));
        } else {
            std::memcpy(dst, src, cnt * sizeof(double));
        }
        return 0;
    };
    MRBF_TRY(put(x_plus, Xb.data() + (size_t)(i + 1) * d, d));
    MRBF_TRY(put(mx_plus, Vb.data() + (size_t)(i + 1) * k, k));
    MRBF_TRY(put(step, hs.data(), d));
    if (n_loops) *n_loops = i;
    return MRBF_OK;
}

int32_t mrbf_model_dims(const mrbf_model *model, int64_t *n, int32_t *d, int32_t *k, int32_t *q) {
    if (!model) return -1;
    if (n) *n = model->n;
    if (d) *d = model->d;
    if (k) *k = model->k;
    if (q) *q = model->q;
    return MRBF_OK;
}

int32_t mrbf_free_model(mrbf_ctx *ctx, mrbf_model *model) {
    if (!ctx) return -1;
    if (!model) return MRBF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    destroy_model(ctx, model);
    return MRBF_OK;
}

}  // extern "C"

extern "C" int32_t mrbf_stochastic_rank(int32_t lam, const double *f, const double *phi, const double *u, double pf, int32_t *idx_out) {
    if (lam <= 0) return -1;
    if (!f) return -2;
    if (!phi) return -3;
    if (!u) return -4;
    if (!idx_out) return -6;
    for (int32_t i = 0; i < lam; ++i) idx_out[i] = i;
    for (int32_t sweep = 0; sweep < lam; ++sweep) {
        bool swapped = false;
        const double *us = u + (size_t)sweep * (size_t)(lam - 1);
        for (int32_t j = 0; j + 1 < lam; ++j) {
            const int32_t a = idx_out[j], b = idx_out[j + 1];
            const bool by_f = (phi[a] == 0.0 && phi[b] == 0.0) || us[j] < pf;
            const bool worse = by_f ? (f[a] > f[b]) : (phi[a] > phi[b]);
            if (worse) {
                idx_out[j] = b;
                idx_out[j + 1] = a;
                swapped = true;
            }
        }
        if (!swapped) break;
    }
    return MRBF_OK;
}

// Host only: is the environment switch `name` honoured right now (set, and MRBF_EXPERIMENTS=1 opens the gate)?  1 / 0; -1 for NULL.
extern "C" int32_t mrbf_debug_env(const char *name) {
    if (!name) return -1;
    return mrbf::mrbf_env(name) ? 1 : 0;
}
