// mrbf_batch_run: the many-problem mode -- replaces the reference's Threads.@threads loop over independent problems
// (/root/reference/examples/large_scale_benchmarks.jl:253; Halton starts :102-109, n = 2d + 1 sites per start :157, 50 (d + 1)
// surrogate evaluations per subproblem :217).
//
// Problems are dealt round-robin over the GPUs (problem p -> GPU p % n_dev, one host thread + context per GPU, no exchange).  On each
// GPU the small problems (n <= 512, d <= 128, Cholesky paths -- the regime Morbit runs in) go through a handful of launches whose grids
// span ALL of them:
//     1   small_fit_kernel            one workgroup per problem: the whole fit (small.hip)
//     1   center_pad_batch_kernel     queries of all problems (and, for the residual, their sites) centred + padded
//   2 k'  eval_fused / eval_combine   the fused MFMA evaluation with the problem index in the grid (k' = output passes)
//     1   batch_check_kernel          residual through the evaluation, max |Pi' w|, order-fixed checksums, per problem
// instead of ~65 device operations per problem.  A batch and single calls (mrbf_fit + mrbf_eval) run the same kernels with the same
// split of the centre range, so their results agree bit for bit (tests/test_gpu_configs.py::test_c4_many_start_batch).  Problems of
// any other shape, and problems whose factorisation raised a flag, take the per-problem chain (api.hip: batch_run_chain).
#include <algorithm>
#include <thread>

#include "common.hpp"
#include "small.hpp"

namespace mrbf {

int batch_run_chain(int n_dev, const int *devs, const std::vector<int64_t> &which, const mrbf_problem *problems, mrbf_result *results);
mrbf_ctx *batch_pool_acquire(int device, int *rc);
void batch_pool_release(mrbf_ctx *c);
void fill_small_prob(const mrbf_ctx *ctx, const mrbf_model *M, const double *Y, double *ws, int *flags, double *scal, int *cl, smallfit::Prob *P);
int small_fit_verdict(const mrbf_model *M, const int *hflags, const double *hscal, mrbf_fit_info *info);

struct CheckDesc {
    const double *V, *Y;       // surrogate values at the sites / data, n x k
    const double *C, *Wc, *W;  // sites n x d, weights npad x k column-major, n x k row-major
    const double *vals;        // m x k values at the queries (may be NULL)
    int64_t n, npad, m;
    int d, k, q;
    double *out;               // [0] sum (V - Y)^2, [1] sum Y^2, [2] max |Pi' w|, [3] sum W, [4] sum vals
};

// one workgroup per problem, every reduction in a fixed order; every loop issues eight independent loads per thread before it
// uses them (a thread that walks its elements one load at a time pays a memory round trip each: the first version of this kernel
// took 166 us for 257 sites)
// part 0: what depends on the fit and on the evaluation at the sites only (runs beside the evaluation of the queries); part 1: the
// checksum of the values at the queries
__global__ __launch_bounds__(256) void batch_check_kernel(const CheckDesc *__restrict__ many, int part) {
    const CheckDesc &E = many[blockIdx.x];
    __shared__ double r0[256], r1[256];
    const int tid = threadIdx.x;
    auto tree = [&](double a, double b, bool use_max) {
        r0[tid] = a;
        r1[tid] = b;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) {
                r0[tid] = use_max ? fmax(r0[tid], r0[tid + w]) : r0[tid] + r0[tid + w];
                r1[tid] += r1[tid + w];
            }
            __syncthreads();
        }
    };
    // sum over i = tid, tid + 256, ... of f(i), eight terms in flight, added in index order
    auto strided_sum = [&](int64_t cnt, auto f) {
        double s = 0.0;
        int64_t i = tid;
        for (; i + 7 * 256 < cnt; i += 8 * 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = f(i + 256 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < cnt; i += 256) s += f(i);
        return s;
    };
    if (part == 1) {
        const double *vals = E.vals;
        const double sv = vals ? strided_sum(E.m * E.k, [&](int64_t i) { return vals[i]; }) : 0.0;
        tree(0.0, sv, false);
        if (tid == 0) E.out[4] = r1[0];
        return;
    }
    const double *V = E.V, *Y = E.Y;
    const double s0 = strided_sum(E.n * E.k, [&](int64_t i) { const double df = V[i] - Y[i]; return df * df; });
    const double s1 = strided_sum(E.n * E.k, [&](int64_t i) { return Y[i] * Y[i]; });
    tree(s0, s1, false);
    if (tid == 0) {
        E.out[0] = r0[0];
        E.out[1] = r1[0];
    }
    __syncthreads();
    // max |Pi' w|: entry (t, l) = sum_i pi_t(c_i) w_il, pi_0 = 1, pi_t = coordinate t - 1; thread e owns entry e (consecutive t:
    // coalesced rows of C), eight sites per step
    double mx = 0.0;
    for (int e = tid; e < E.q * E.k; e += 256) {
        const int t = e % E.q, l = e / E.q;
        const double *wl = E.Wc + (int64_t)l * E.npad;
        double acc = 0.0;
        int64_t i = 0;
        for (; i + 8 <= E.n; i += 8) {
            double c[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = t == 0 ? 1.0 : E.C[(i + u) * E.d + (t - 1)];
                w[u] = wl[i + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(c[u], w[u], acc);
        }
        for (; i < E.n; ++i) acc = fma(t == 0 ? 1.0 : E.C[i * E.d + (t - 1)], wl[i], acc);
        mx = fmax(mx, fabs(acc));
    }
    const double *W = E.W;
    const double sw = strided_sum(E.n * E.k, [&](int64_t i) { return W[i]; });
    tree(mx, sw, true);
    if (tid == 0) {
        E.out[2] = r0[0];
        E.out[3] = r1[0];
    }
}

static inline size_t al16(size_t c) { return (c + 15) & ~size_t(15); }

// all small problems of one GPU; `redo` receives the problems that have to take the per-problem chain after all (flags)
static int run_small_batch(mrbf_ctx *ctx, const std::vector<int64_t> &idx, const mrbf_problem *problems, mrbf_result *results,
                           std::vector<int64_t> *redo, int force_nc = 0) {
    const int P = (int)idx.size();
    if (P == 0) return 0;
    (void)hipSetDevice(ctx->device);
    struct Lay {
        size_t C, Y, X, Xc, sq, mean, W, Wc, lam, ws, Xq0, xsq0, vp0, gp0, V0, Xq1, xsq1, vp1, gp1, vals, jac, out;
        int nsplit0, nsplit1, D, q, KO;
        int64_t npad, mpad0, mpad1;
    };
    std::vector<Lay> lay(P);
    size_t total = 0;
    auto take = [&](size_t cnt) {
        const size_t at = total;
        total += al16(std::max<size_t>(cnt, 1));
        return at;
    };
    const bool check = ctx->residual != 0;
    // the per-problem result words (8 doubles each), contiguous, and the flag words (4 ints each) right behind them: one copy back
    // for the whole batch
    const size_t out0 = take((size_t)10 * P);
    for (int i = 0; i < P; ++i) {
        const mrbf_problem &pr = problems[idx[i]];
        Lay &L = lay[i];
        const int d = pr.d, k = pr.k;
        L.q = poly_dim(d, pr.poly_deg);
        L.npad = round_up(pr.n, 128);
        L.D = d <= 64 ? 64 : 128;
        L.KO = outputs_per_pass(k, L.D);
        const int q16 = (int)round_up(std::max(L.q, 1), 16);
        L.C = is_device_ptr(pr.centres) ? (size_t)-1 : take((size_t)pr.n * d);
        L.Y = is_device_ptr(pr.values) ? (size_t)-1 : take((size_t)pr.n * k);
        L.X = (pr.m == 0 || is_device_ptr(pr.X)) ? (size_t)-1 : take((size_t)pr.m * d);
        L.Xc = take((size_t)L.npad * L.D);
        L.sq = take((size_t)L.npad);
        L.mean = take((size_t)L.D);
        L.W = (pr.weights_out && is_device_ptr(pr.weights_out)) ? (size_t)-1 : take((size_t)pr.n * k);  // device output: written in place
        L.Wc = take((size_t)L.npad * k);
        L.lam = (pr.poly_out && L.q > 0 && is_device_ptr(pr.poly_out)) ? (size_t)-1 : take((size_t)std::max(L.q, 1) * k);
        L.ws = take(smallfit::carve((int)L.npad, q16).total);
        // evaluation 0: the fit's residual at the sites; evaluation 1: the caller's queries
        L.mpad0 = check ? round_up(pr.n, 64) : 0;
        L.nsplit0 = check ? eval_nsplit(ctx, pr.n, (int)((pr.n + 63) / 64), true) : 1;
        L.Xq0 = take((size_t)L.mpad0 * L.D);
        L.xsq0 = take((size_t)L.mpad0);
        L.vp0 = take(L.nsplit0 > 1 ? (size_t)L.nsplit0 * L.mpad0 * L.KO * 2 : 0);
        L.gp0 = take((size_t)0);
        L.V0 = take((size_t)pr.n * k);
        L.mpad1 = round_up(pr.m, 64);
        L.nsplit1 = pr.m > 0 ? eval_nsplit(ctx, pr.m, (int)((pr.n + 63) / 64)) : 1;
        L.Xq1 = take((size_t)L.mpad1 * L.D);
        L.xsq1 = take((size_t)L.mpad1);
        L.vp1 = take(L.nsplit1 > 1 ? (size_t)L.nsplit1 * L.mpad1 * L.KO * 2 : 0);
        L.gp1 = take((pr.jac_out && L.nsplit1 > 1) ? (size_t)L.nsplit1 * L.mpad1 * L.KO * L.D : 0);
        L.vals = (pr.m == 0 || (pr.vals_out && is_device_ptr(pr.vals_out))) ? (size_t)-1 : take((size_t)pr.m * k);
        L.jac = (pr.m == 0 || !pr.jac_out || is_device_ptr(pr.jac_out)) ? (size_t)-1 : take((size_t)pr.m * k * d);
        L.out = out0 + (size_t)8 * i;
    }
    // the arena of one launch group is bounded (MRBF_BATCH_ARENA_MB, default 8192): a batch that needs more -- 10^4 problems would ask
    // for tens of GB in one piece -- is worked off in halves, and an allocation that fails is retried in halves as well instead of
    // failing every problem of the device
    const size_t arena_budget = (size_t)(mrbf_env("MRBF_BATCH_ARENA_MB") ? atoll(mrbf_env("MRBF_BATCH_ARENA_MB")) : 8192) << 20;
    auto halves = [&]() -> int {
        const std::vector<int64_t> lo(idx.begin(), idx.begin() + P / 2), hi(idx.begin() + P / 2, idx.end());
        MRBF_TRY(run_small_batch(ctx, lo, problems, results, redo, force_nc));
        return run_small_batch(ctx, hi, problems, results, redo, force_nc);
    };
    if (P > 1 && total * sizeof(double) > arena_budget) return halves();
    double *base;
    int *flags = nullptr;
    {
        const int rc = get_buf(ctx, S_SMALL_WS, total, &base);
        if (rc == MRBF_ENOMEM && P > 1) return halves();
        if (rc != 0) return rc;
    }
    flags = reinterpret_cast<int *>(base + out0 + (size_t)8 * P);
    PinGuard pin(ctx);  // this batch's share of the pinned staging block (descriptor upload, result words); released on every return path
    // descriptors: [Prob x P | EvalDesc x 2P | CheckDesc x P | CL_WORDS cluster words x P (zero)] in one device buffer
    const size_t desc_bytes = (size_t)P * (sizeof(smallfit::Prob) + 2 * sizeof(EvalDesc) + sizeof(CheckDesc) + smallfit::CL_WORDS * sizeof(int)) + 256;
    char *ddesc;
    MRBF_TRY(get_buf(ctx, S_SMALL_DESC, desc_bytes, (void **)&ddesc));
    // (the device addresses of the three arrays and of the cluster words are known before the host copies are filled)
    smallfit::Prob *dprobs = (smallfit::Prob *)ddesc;
    EvalDesc *devs_ = (EvalDesc *)(ddesc + al16((size_t)P * sizeof(smallfit::Prob)));
    CheckDesc *dchk = (CheckDesc *)((char *)devs_ + al16((size_t)2 * P * sizeof(EvalDesc)));
    int *dcl = (int *)((char *)dchk + al16((size_t)P * sizeof(CheckDesc)));
    std::vector<char> hdesc;
    std::vector<smallfit::Prob> probs(P);
    std::vector<EvalDesc> evs((size_t)2 * P);
    std::vector<CheckDesc> chk(P);
    std::vector<mrbf_model> models(P);
    hipStream_t st = ctx->stream;
    auto dev_in = [&](const double *user, size_t off, size_t cnt, const double **out) -> int {
        if (off == (size_t)-1) {
            *out = user;
            return 0;
        }
        MRBF_HIP(ctx, hipMemcpyAsync(base + off, user, cnt * sizeof(double), hipMemcpyHostToDevice, st));
        *out = base + off;
        return 0;
    };
    for (int i = 0; i < P; ++i) {
        const mrbf_problem &pr = problems[idx[i]];
        const Lay &L = lay[i];
        const double *C, *Y, *X = nullptr;
        MRBF_TRY(dev_in(pr.centres, L.C, (size_t)pr.n * pr.d, &C));
        MRBF_TRY(dev_in(pr.values, L.Y, (size_t)pr.n * pr.k, &Y));
        if (pr.m > 0) MRBF_TRY(dev_in(pr.X, L.X, (size_t)pr.m * pr.d, &X));
        mrbf_model &M = models[i];
        M.n = pr.n;
        M.npad = L.npad;
        M.d = pr.d;
        M.dpad = L.D;
        M.k = pr.k;
        M.q = L.q;
        M.deg = pr.poly_deg;
        M.kp = make_kp(pr.kernel_id, pr.a, pr.b);
        M.C = const_cast<double *>(C);
        M.Xc = base + L.Xc;
        M.sq = base + L.sq;
        M.mean = base + L.mean;
        M.W = L.W == (size_t)-1 ? pr.weights_out : base + L.W;
        M.Wc = base + L.Wc;
        M.lam = L.lam == (size_t)-1 ? pr.poly_out : base + L.lam;
        fill_small_prob(ctx, &M, Y, base + L.ws, flags + 4 * i, base + L.out + 5, dcl + smallfit::CL_WORDS * i, &probs[i]);
        probs[i].mean_given = 1;  // small_mean_kernel below: the queries are centred beside the fit, not after it
        for (int e = 0; e < 2; ++e) {
            EvalDesc &E = evs[(size_t)e * P + i];  // [residual evaluations of all problems | query evaluations of all problems]
            const int64_t m = e == 0 ? (check ? pr.n : 0) : pr.m;
            E.X = e == 0 ? C : X;
            E.mean = M.mean;
            E.Xq = base + (e == 0 ? L.Xq0 : L.Xq1);
            E.xsq = base + (e == 0 ? L.xsq0 : L.xsq1);
            E.Cc = M.Xc;
            E.csq = M.sq;
            E.Wc = M.Wc;
            E.lam = M.lam;
            E.npad = L.npad;
            E.mpad = e == 0 ? L.mpad0 : L.mpad1;
            E.m = m;
            E.d = pr.d;
            E.k = pr.k;
            E.q = L.q;
            E.nsplit = e == 0 ? L.nsplit0 : L.nsplit1;
            const int ntiles = (int)((pr.n + 63) / 64);
            E.ntiles = ntiles;
            E.nsub = (int)((pr.n + 15) / 16);
            E.tiles_per_split = (ntiles + E.nsplit - 1) / E.nsplit;
            E.kp = M.kp;
            E.vpart = base + (e == 0 ? L.vp0 : L.vp1);
            E.sapart = E.vpart + (size_t)E.nsplit * E.mpad * L.KO;
            E.gpart = base + (e == 0 ? L.gp0 : L.gp1);
            E.vals = e == 0 ? base + L.V0 : (L.vals == (size_t)-1 ? pr.vals_out : base + L.vals);
            E.jac = e == 0 ? nullptr : (pr.jac_out ? (L.jac == (size_t)-1 ? pr.jac_out : base + L.jac) : nullptr);
        }
        CheckDesc &K = chk[i];
        K.V = base + L.V0;
        K.Y = Y;
        K.C = C;
        K.Wc = M.Wc;
        K.W = M.W;
        K.vals = pr.m > 0 ? evs[(size_t)P + i].vals : nullptr;
        K.n = check ? pr.n : 0;
        K.npad = L.npad;
        K.m = pr.m;
        K.d = pr.d;
        K.k = pr.k;
        K.q = check ? L.q : 0;
        K.out = base + L.out;
    }
    {
        // one upload for the three descriptor arrays and the zeroed cluster words (the staging vector lives until the stream is synchronised below)
        // (staged in the context's pinned block when it fits: the upload is then asynchronous and the result words below come back
        //  with the one synchronisation of the batch instead of a round trip of their own)
        char *hd = desc_bytes <= ((size_t)4 << 20) ? pin_take(ctx, desc_bytes) : nullptr;  // (this batch's guard armed the block: `pin` below)
        if (hd) {
            memset(hd, 0, desc_bytes);
        } else {
            hdesc.assign(desc_bytes, 0);
            hd = hdesc.data();
        }
        memcpy(hd, probs.data(), (size_t)P * sizeof(smallfit::Prob));
        memcpy(hd + ((char *)devs_ - ddesc), evs.data(), (size_t)2 * P * sizeof(EvalDesc));
        memcpy(hd + ((char *)dchk - ddesc), chk.data(), (size_t)P * sizeof(CheckDesc));
        MRBF_HIP(ctx, hipMemcpyAsync(ddesc, hd, desc_bytes, hipMemcpyHostToDevice, st));
    }
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    const int nc = force_nc > 0 ? force_nc : small_fit_cluster(ctx, P);
    hipStream_t side = (ctx->panel_stream && ctx->panel_stream != st) ? ctx->panel_stream : st;
    // the centroids first (the fit kernel's own arithmetic, small.hip: column_means), then -- on the side stream, under the
    // latency-bound fit launch -- the centring of all evaluation points of the batch (sites for the residual, queries): 423 MB of
    // queries at C4, 0.11 ms that used to stand between the fit and the evaluation
    MRBF_TRY(launch_small_means(ctx, dprobs, P));
    if (side != st) {
        MRBF_HIP(ctx, hipEventRecord(ctx->evx[0], st));
        MRBF_HIP(ctx, hipStreamWaitEvent(side, ctx->evx[0], 0));
    }
    // (the fit launch first: its workgroups -- one per CU, clusters that meet at barriers -- are all resident before the centring's
    //  1600 blocks arrive on the high-priority side stream; the other way round the fit took 0.83 or 0.96 ms from run to run)
    MRBF_TRY(launch_small_fit(ctx, probs.data(), P, dprobs, nc));
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    {
        int64_t max_mpad = 0;
        for (const EvalDesc &E : evs) max_mpad = std::max(max_mpad, E.mpad);
        ctx->stream = side;
        const int rc = max_mpad > 0 ? center_pad_batch(ctx, devs_, 2 * P, max_mpad) : 0;
        ctx->stream = st;
        if (rc != 0) return rc;
    }
    // evaluation launches per group of equal (kernel, fast flag, padded dimension, outputs, Jacobians wanted, centre range split or not): usually two groups,
    // the residual evaluations at the sites (no Jacobians; on the side stream, beside the evaluation of the queries, together with
    // part 0 of the check kernel) and the evaluations of the queries
    if (side != st) {
        MRBF_HIP(ctx, hipEventRecord(ctx->evx[0], st));
        MRBF_HIP(ctx, hipStreamWaitEvent(side, ctx->evx[0], 0));   // residual evaluations: after the fit
        MRBF_HIP(ctx, hipEventRecord(ctx->evx[2], side));
        MRBF_HIP(ctx, hipStreamWaitEvent(st, ctx->evx[2], 0));     // query evaluations: after the centring
    }
    {
        struct RestoreStream {
            mrbf_ctx *c;
            hipStream_t s;
            ~RestoreStream() { c->stream = s; }
        } restore{ctx, st};
        std::vector<char> done((size_t)2 * P, 0);
        for (size_t a = 0; a < evs.size(); ++a) {
            if (done[a]) continue;
            const mrbf_problem &pa = problems[idx[a % P]];
            const KP kpa = evs[a].kp;
            const bool ja = evs[a].jac != nullptr;
            const bool is_check = a < (size_t)P;
            std::vector<EvalDesc> grp;
            std::vector<size_t> members;
            for (size_t b = a; b < (is_check ? (size_t)P : evs.size()); ++b) {
                const mrbf_problem &pb = problems[idx[b % P]];
                // (members that split their centre range and members that do not -- few query points against few centre tiles are split since
                // the end of round 5, eval_nsplit -- go into different launches: the group's kernels either all write partials or none does)
                if (!done[b] && evs[b].kp.kid == kpa.kid && evs[b].kp.fast == kpa.fast && lay[b % P].D == lay[a % P].D && pb.k == pa.k &&
                    (evs[b].jac != nullptr) == ja && (evs[b].nsplit > 1) == (evs[a].nsplit > 1)) {
                    done[b] = 1;
                    if (evs[b].m > 0) members.push_back(b);
                }
            }
            if (members.empty()) continue;
            ctx->stream = is_check ? side : st;
            // contiguous runs of the device array are launched as they lie; scattered members are compacted into a second array
            bool contiguous = true;
            for (size_t j = 1; j < members.size(); ++j) contiguous = contiguous && members[j] == members[j - 1] + 1;
            const EvalDesc *dev = devs_ + members[0];
            const EvalDesc *host = &evs[members[0]];
            if (!contiguous) {
                for (size_t b : members) grp.push_back(evs[b]);
                EvalDesc *dgrp;
                MRBF_TRY(get_buf(ctx, is_check ? S_STAGE_C : S_STAGE_D, grp.size() * sizeof(EvalDesc) / sizeof(double) + 16, (double **)&dgrp));
                MRBF_HIP(ctx, hipMemcpyAsync(dgrp, grp.data(), grp.size() * sizeof(EvalDesc), hipMemcpyHostToDevice, ctx->stream));
                MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // grp is reused by the next group
                dev = dgrp;
                host = grp.data();
            }
            MRBF_TRY(eval_fused_batch(ctx, kpa, lay[a % P].D, pa.k, ja, host, dev, (int)members.size(), true));
            if (!contiguous) MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    hipLaunchKernelGGL(batch_check_kernel, dim3((unsigned)P), dim3(256), 0, side, dchk, 0);
    if (side != st) {
        MRBF_HIP(ctx, hipEventRecord(ctx->evx[1], side));
        MRBF_HIP(ctx, hipStreamWaitEvent(st, ctx->evx[1], 0));
    }
    hipLaunchKernelGGL(batch_check_kernel, dim3((unsigned)P), dim3(256), 0, st, dchk, 1);
    MRBF_HIP(ctx, hipGetLastError());
    MRBF_HIP(ctx, hipEventRecord(ctx->ev[2], st));
    // results
    std::vector<double> hout_v;
    double *hout;
    if (char *ph = (size_t)10 * P * sizeof(double) <= ((size_t)1 << 20) ? pin_take(ctx, (size_t)10 * P * sizeof(double)) : nullptr) {
        hout = reinterpret_cast<double *>(ph);  // (from the same allocator as the descriptor staging area)
    } else {
        hout_v.resize((size_t)10 * P);
        hout = hout_v.data();
    }
    MRBF_HIP(ctx, hipMemcpyAsync(hout, base + out0, (size_t)10 * P * sizeof(double), hipMemcpyDeviceToHost, st));
    const int *hflags = reinterpret_cast<const int *>(hout + (size_t)8 * P);
    for (int i = 0; i < P; ++i) {
        const mrbf_problem &pr = problems[idx[i]];
        const Lay &L = lay[i];
        if (pr.weights_out && L.W != (size_t)-1)
            MRBF_HIP(ctx, hipMemcpyAsync(pr.weights_out, base + L.W, (size_t)pr.n * pr.k * sizeof(double), hipMemcpyDeviceToHost, st));
        if (pr.poly_out && L.q > 0 && L.lam != (size_t)-1)
            MRBF_HIP(ctx, hipMemcpyAsync(pr.poly_out, base + L.lam, (size_t)L.q * pr.k * sizeof(double), hipMemcpyDeviceToHost, st));
        if (pr.m > 0 && pr.vals_out && L.vals != (size_t)-1)
            MRBF_HIP(ctx, hipMemcpyAsync(pr.vals_out, base + L.vals, (size_t)pr.m * pr.k * sizeof(double), hipMemcpyDeviceToHost, st));
        if (pr.m > 0 && pr.jac_out && L.jac != (size_t)-1)
            MRBF_HIP(ctx, hipMemcpyAsync(pr.jac_out, base + L.jac, (size_t)pr.m * pr.k * pr.d * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    MRBF_HIP(ctx, hipStreamSynchronize(st));
    if (nc > 1) {
        int worst = 0;
        std::vector<std::pair<int, double>> suspects;  // tripwire (small.hip, Cluster): clustered fits that do not interpolate, with their residual
        for (int i = 0; i < P; ++i) {
            worst = std::max(worst, hflags[(size_t)4 * i + 3]);
            const double *o = &hout[(size_t)8 * i];
            const bool flagged = hflags[(size_t)4 * i] != 0 || hflags[(size_t)4 * i + 1] != 0 || hflags[(size_t)4 * i + 2] != 0;
            const double rr = std::sqrt(o[0]) / std::max(std::sqrt(o[1]), 1e-300);
            if (check && !flagged && !(rr < 1e-6)) suspects.emplace_back(idx[i], rr);
        }
        if (worst != 0 || !suspects.empty()) {
            // members on different XCDs (2): clusters off for this context; a barrier that timed out (1): this batch again with one workgroup
            // per problem, clusters stay unless it keeps happening; a residual that is not small: one workgroup per problem has to confirm
            // it -- clusters are switched off only if that run does BETTER on a suspect problem (as fit_model does for single fits,
            // solve.hip).  A problem that is ill-conditioned in its own right (near-duplicate sites of a trust region) gives the same
            // residual either way and must not cost a pooled context its clusters for the rest of the process.
            if (worst == 2 || (worst == 1 && ++ctx->small_timeouts >= 3)) ctx->small_nc = 1;
            const int rc1 = run_small_batch(ctx, idx, problems, results, redo, 1);
            if (rc1 == 0 && worst == 0) {
                for (const auto &sp : suspects) {
                    const double r1 = results[sp.first].fit.rel_residual;
                    if (std::isfinite(r1) && (r1 < 1e-6 || r1 < 0.1 * sp.second)) ctx->small_nc = 1;  // one workgroup interpolates, the cluster did not
                }
            }
            return rc1;
        }
        ctx->small_timeouts = 0;
    }
    float ms_fit = 0.f, ms_eval = 0.f;
    MRBF_HIP(ctx, hipEventElapsedTime(&ms_fit, ctx->ev[0], ctx->ev[1]));
    MRBF_HIP(ctx, hipEventElapsedTime(&ms_eval, ctx->ev[1], ctx->ev[2]));
    for (int i = 0; i < P; ++i) {
        const mrbf_problem &pr = problems[idx[i]];
        mrbf_result &res = results[idx[i]];
        std::memset(&res, 0, sizeof(res));
        res.device = ctx->device;
        res.fit.n = (int32_t)pr.n;
        res.fit.q = lay[i].q;
        res.fit.rel_residual = NAN;
        res.fit.max_pitw = NAN;
        const double *o = &hout[(size_t)8 * i];
        if (small_fit_verdict(&models[i], &hflags[(size_t)4 * i], o + 5, &res.fit)) {
            redo->push_back(idx[i]);  // not positive definite / rank-deficient tail: the per-problem chain takes the LU path
            continue;
        }
        if (check) {
            res.fit.rel_residual = std::sqrt(o[0]) / std::max(std::sqrt(o[1]), 1e-300);
            res.fit.max_pitw = lay[i].q > 0 ? o[2] : 0.0;
        }
        // one launch fits the whole batch and a few more evaluate it: the times are those of the batch, shared by its members
        res.fit.ms_factor = ms_fit;
        res.fit.ms_total = ms_fit;
        res.ms_eval = ms_eval;
        res.checksum_w = o[3];
        res.checksum_vals = o[4];
    }
    return 0;
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_batch_run(int32_t n_dev, const int32_t *device_ids, int64_t n_problems, const mrbf_problem *problems,
                                  mrbf_result *results) {
    if (n_dev < 1) return -1;
    if (n_problems < 0) return -3;
    if (n_problems > 0 && (!problems || !results)) return -4;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return MRBF_ENODEVICE;
    std::vector<int> devs(n_dev);
    for (int g = 0; g < n_dev; ++g) {
        devs[g] = device_ids ? device_ids[g] : g;
        if (devs[g] < 0 || devs[g] >= visible) return -2;
    }
    // which problems the batched small-problem path takes (decided as fit_model decides the path of a single call)
    std::vector<std::vector<int64_t>> small(n_dev);
    std::vector<int64_t> chain;
    std::vector<int> rcs(n_dev, 0);
    std::vector<std::vector<int64_t>> redo(n_dev);
    auto worker = [&](int g) {
        if (small[g].empty()) return;
        int rc = 0;
        mrbf_ctx *ctx = batch_pool_acquire(devs[g], &rc);
        if (rc != 0) {
            rcs[g] = rc;
            for (int64_t p : small[g]) {
                std::memset(&results[p], 0, sizeof(mrbf_result));
                results[p].status = rc;
                results[p].device = devs[g];
            }
            return;
        }
        rcs[g] = run_small_batch(ctx, small[g], problems, results, &redo[g]);
        if (rcs[g] != 0)
            for (int64_t p : small[g]) {
                std::memset(&results[p], 0, sizeof(mrbf_result));
                results[p].status = rcs[g];
                results[p].device = devs[g];
            }
        batch_pool_release(ctx);
    };
    {
        // (the context only supplies the options small_fit_applies looks at; a pooled context of the first device)
        int rc = 0;
        mrbf_ctx *probe = batch_pool_acquire(devs[0], &rc);
        if (rc != 0) return rc;
        for (int64_t p = 0; p < n_problems; ++p) {
            const mrbf_problem &pr = problems[p];
            const int q = poly_dim(pr.d, pr.poly_deg);
            int path = MRBF_PATH_LU;
            if (pr.kernel_id >= 0 && pr.kernel_id <= 4 && pr.n > q && pr.poly_deg >= -1 && pr.poly_deg <= 1 &&
                cpd_order(pr.kernel_id, pr.a, pr.b) <= pr.poly_deg + 1)
                path = q > 0 ? MRBF_PATH_PROJ_CHOL : MRBF_PATH_CHOL;
            const bool ok = pr.centres && pr.values && (pr.m == 0 || pr.X) && pr.m >= 0 && probe->force_path == 0 && probe->eval_impl != 1 &&
                            small_fit_applies(probe, pr.n, pr.d, pr.k, q, path);
            if (ok)
                small[p % n_dev].push_back(p);
            else
                chain.push_back(p);
        }
        batch_pool_release(probe);
    }
    std::vector<std::thread> th;
    for (int g = 0; g < n_dev; ++g) th.emplace_back(worker, g);
    for (auto &t : th) t.join();
    for (int g = 0; g < n_dev; ++g) {
        if (rcs[g] != 0) return rcs[g];
        chain.insert(chain.end(), redo[g].begin(), redo[g].end());
    }
    return batch_run_chain(n_dev, devs.data(), chain, problems, results);
}
