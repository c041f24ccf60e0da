// Pascoletti-Serafini descent step with the whole subproblem solver on the device -- replaces the NLopt runs of
// compute_local_ideal_point / _ps_optimization inside get_criticality(::PascolettiSerafiniConfig, ...)
// (/root/reference/src/descent.jl:369-412, :434-510, :512-581) for objectives that share one grouped RBF model.
//
// The reference hands one-point closures to NLopt's GN_ISRES: every candidate costs k sweeps over all n centres.  Here the
// subproblems are solved by an ISRES-style (mu, lambda) evolution strategy (Runarsson & Yao; population 20 (dim + 1), mu =
// lambda / 7, as NLopt's defaults) whose state lives in device memory:
//   * the k single-objective runs of the local ideal point and the Pascoletti-Serafini run each own a population block;
//     all blocks of a generation are evaluated by ONE batched surrogate sweep (eval_model, values only);
//   * one workgroup per run then does everything else of the generation: objective / constraint violation, best-so-far,
//     stochastic ranking (odd-even transposition in LDS with the random comparison rule -- the parallel form of the bubble
//     sweeps), survivor selection, differential variation, log-normal self-adaptive mutation with re-draws inside the box,
//     from a counter-based generator (Philox 4x32-10, keyed by seed / run / generation: no state, any launch order);
//   * the host enqueues generations back to back and reads the runs' status words every few generations only.
// Parity with the reference is the contract of get_criticality (problem solved, budgets, start values, critical / failure
// short cuts, returned tuple), not NLopt's random trajectory; the returned point is always feasible for the subproblem.
#include "radial.hpp"

namespace mrbf {

int eval_model(mrbf_ctx *ctx, const mrbf_model *M, int64_t m, const double *Xdev, double *vals_dev, double *jac_dev,
               mrbf_eval_info *info);

namespace ps {

constexpr int LDSPOP = 6144;  // doubles of population / step sizes staged in LDS by the generation kernel
constexpr int MAXLAM = 2048;  // population limit of the device path (dim <= 100); larger problems use the host loop of the mirrors

struct Run {  // one (mu, lambda) run; all pointers into device arenas
    int nvar;       // 1 + d for the PS run (chi = [t; x]), d for an ideal-point run
    int lam, mu;
    int kind;       // 0: minimise output `obj` over the box; 1: Pascoletti-Serafini run
    int obj;
    int off;        // first row of this run's block in the evaluation batch
    int max_evals;
    double *X[2];   // lam x nvar, ping-pong
    double *S[2];   // lam x nvar step sizes
    double *best;   // nvar + 2: best_x, best_f, best_phi
    int *stat;      // [0] evals so far, [1] done, [2] generations run
};

struct Args {
    Run runs[9];
    int nruns, d, k;
    const double *lb, *ub;   // d
    const double *mx, *r;    // k (PS run)
    const double *F;         // evaluation batch results, rows x k
    double *Xeval;           // evaluation batch, rows x d
    unsigned long long seed;
    double xtol_rel;
    int gen;
    int dbg;  // timing experiments (MRBF_PS_DBG): 1 no ranking, 2 no breeding, 4 no best / stop bookkeeping
};

// ---- Philox 4x32-10 ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0;
        c[1] = n1;
        c[2] = n2;
        c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// two uniforms in (0,1) and two standard normals for (run, generation, individual, component, draw)
__device__ __forceinline__ void rng4(const Args &a, int run, int gen, int ind, int comp, int draw, double &u0, double &u1, double &z0, double &z1) {
    unsigned c[4] = {(unsigned)ind, (unsigned)comp, (unsigned)(gen * 16 + draw), (unsigned)run};
    philox(c, (unsigned)a.seed, (unsigned)(a.seed >> 32));
    u0 = ((double)c[0] + 0.5) * (1.0 / 4294967296.0);
    u1 = ((double)c[1] + 0.5) * (1.0 / 4294967296.0);
    // Box-Muller in single precision (hardware log / sin / cos): the draws only steer a random search
    const float v0 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), v1 = ((float)(c[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = __fsqrt_rn(-2.0f * __logf(v0));
    z0 = (double)(rad * __cosf(6.2831853f * v1));
    z1 = (double)(rad * __sinf(6.2831853f * v1));
}

__device__ __forceinline__ double lo_of(const Args &a, const Run &R, int j) { return R.kind == 1 ? (j == 0 ? -1.0 : a.lb[j - 1]) : a.lb[j]; }
__device__ __forceinline__ double hi_of(const Args &a, const Run &R, int j) { return R.kind == 1 ? (j == 0 ? 0.0 : a.ub[j - 1]) : a.ub[j]; }

// generation 0: uniform population in the box, individual 0 = the start point (PS: [t0; x_n]), PS individual 1 = [0; x_n]
// (always feasible: m(x_n) - m(x_n) - 0 r = 0, so the run can never end without a feasible point); step sizes (ub - lb) / sqrt(n)
__global__ __launch_bounds__(256) void ps_init_kernel(Args a, const double *xn, double t0) {
    const Run &R = a.runs[blockIdx.x];
    const int n = R.nvar;
    for (int e = threadIdx.x; e < R.lam * n; e += 256) {
        const int i = e / n, j = e % n;
        const double lo = lo_of(a, R, j), hi = hi_of(a, R, j);
        double u0, u1, z0, z1;
        rng4(a, blockIdx.x, 0, i, j, 15, u0, u1, z0, z1);
        double v = lo + u0 * (hi - lo);
        if (i == 0 || (i == 1 && R.kind == 1)) {
            v = (R.kind == 1) ? (j == 0 ? (i == 0 ? t0 : 0.0) : xn[j - 1]) : xn[j];
            v = fmin(fmax(v, lo), hi);
        }
        R.X[0][e] = v;
        R.S[0][e] = (hi - lo) / sqrt((double)n);
        const int skip = R.kind == 1 ? 1 : 0;  // the evaluation batch holds the x part of every individual of every run
        if (j >= skip) a.Xeval[(size_t)(R.off + i) * a.d + (j - skip)] = v;
    }
    if (threadIdx.x == 0) {
        R.best[n] = INFINITY;
        R.best[n + 1] = INFINITY;
        R.stat[0] = 0;
        R.stat[1] = 0;
        R.stat[2] = 0;
    }
}

__global__ __launch_bounds__(256) void ps_step_kernel(Args a) {
    __shared__ double sf[MAXLAM], sphi[MAXLAM];
    __shared__ int sidx[MAXLAM];
    __shared__ int s_swapped, s_stop, s_best;
    __shared__ int s_cand[256];
    const int run = blockIdx.x, tid = threadIdx.x;
    const Run &R = a.runs[run];
    if (R.stat[1]) return;  // finished earlier
    const int n = R.nvar, lam = R.lam, mu = R.mu, k = a.k, gen = a.gen;
    const double *X = R.X[gen & 1], *S = R.S[gen & 1];
    double *Xn = R.X[(gen + 1) & 1], *Sn = R.S[(gen + 1) & 1];
    const int evals0 = R.stat[0];
    // a single workgroup is latency-bound: every dependent global load costs ~1-2 us, so the population, the step sizes and the
    // box are staged in LDS once (coalesced) whenever they fit
    __shared__ double ldsX[LDSPOP], ldsS[LDSPOP], ldsLo[256], ldsHi[256];
    const bool staged = lam * n <= LDSPOP && n <= 256;
    if (staged) {
        for (int e = tid; e < lam * n; e += 256) {
            ldsX[e] = X[e];
            ldsS[e] = S[e];
        }
        for (int c = tid; c < n; c += 256) {
            ldsLo[c] = lo_of(a, R, c);
            ldsHi[c] = hi_of(a, R, c);
        }
        X = ldsX;
        S = ldsS;
    }
    auto lo_at = [&](int c) { return staged ? ldsLo[c] : lo_of(a, R, c); };
    auto hi_at = [&](int c) { return staged ? ldsHi[c] : hi_of(a, R, c); };
    const int m = min(lam, R.max_evals - evals0);  // individuals of this generation inside the budget
    // ---- objective and constraint violation
    for (int i = tid; i < lam; i += 256) {
        double f = INFINITY, phi = INFINITY;
        if (i < m) {
            const double *Fi = a.F + (size_t)(R.off + i) * k;
            if (R.kind == 0) {
                f = Fi[R.obj];
                phi = 0.0;
            } else {
                const double t = X[(size_t)i * n];
                f = t;
                phi = 0.0;
                for (int l = 0; l < k; ++l) {
                    const double g = Fi[l] - a.mx[l] - t * a.r[l];  // m_l(x) - m_l(x_n) - t r_l <= 0  (descent.jl:443)
                    phi += g > 0.0 ? g * g : 0.0;
                }
            }
            if (!(f == f) || !(phi == phi) || fabs(f) == INFINITY || fabs(phi) == INFINITY) {
                f = INFINITY;
                phi = INFINITY;
            }
        }
        sf[i] = f;
        sphi[i] = phi;
        sidx[i] = i;
    }
    __syncthreads();
    // ---- best so far: feasible beats infeasible, then the objective (strided scan per thread, 256 finalists by thread 0)
    {
        auto better = [&](int i, int j) {  // is individual i better than j
            if (j < 0) return true;
            const bool fi = sphi[i] == 0.0, fj = sphi[j] == 0.0;
            return (fi && !fj) || (fi && fj && sf[i] < sf[j]) || (!fi && !fj && sphi[i] < sphi[j]);
        };
        int mine = -1;
        for (int i = tid; i < m; i += 256)
            if (better(i, mine)) mine = i;
        s_cand[tid] = mine;
        __syncthreads();
        if (tid == 0) {
            int j = -1;
            for (int t = 0; t < 256; ++t)
                if (s_cand[t] >= 0 && (j < 0 || better(s_cand[t], j) || (!better(j, s_cand[t]) && s_cand[t] < j))) j = s_cand[t];
            s_best = -1;
            if (j >= 0) {
                const double bf = R.best[n], bphi = R.best[n + 1];
                const bool fj = sphi[j] == 0.0;
                if ((fj && (bphi > 0.0 || sf[j] < bf)) || (!fj && sphi[j] < bphi)) {
                    s_best = j;
                    R.best[n] = sf[j];
                    R.best[n + 1] = sphi[j];
                }
            }
            R.stat[0] = evals0 + max(m, 0);
            R.stat[2] = gen + 1;
            s_stop = (evals0 + m >= R.max_evals || m < lam) ? 1 : 0;
        }
        __syncthreads();
        if (s_best >= 0)
            for (int c = tid; c < n; c += 256) R.best[c] = X[(size_t)s_best * n + c];
    }
    if (s_stop) {
        if (tid == 0) R.stat[1] = 1;
        return;
    }
    // ---- stochastic ranking: lam phases of odd-even transposition; a pair is compared by f when both are feasible or with
    //      probability 0.45, else by the constraint violation (Runarsson & Yao)
    // (the records themselves are swapped -- one LDS round trip per phase instead of an index indirection; one Philox call feeds
    //  four phases of a pair; one barrier per phase; the no-swap exit is tested every 16 phases)
    // If no individual inside the budget violates a constraint (always so for the ideal-point runs, usually so late in a PS run)
    // every comparison of the pairwise rule is decided by (violation, objective) whatever is drawn, and lam phases of the stable
    // transposition sort end in THE sorted order (ties by index): a bitonic network over the padded array reaches the same
    // order in log2(N) (log2(N) + 1) / 2 phases (45 instead of 260 at lam = 260).
    __shared__ int s_infeas;
    if (tid == 0) s_infeas = 0;
    __syncthreads();
    for (int i = tid; i < lam; i += 256)
        if (sphi[i] > 0.0 && sphi[i] < INFINITY) s_infeas = 1;
    __syncthreads();
    const bool plain_sort = s_infeas == 0 && !(a.dbg & 4);
    if (plain_sort && !(a.dbg & 1)) {
        int N = 1;
        while (N < lam) N <<= 1;
        for (int i = lam + tid; i < N; i += 256) {
            sf[i] = INFINITY;
            sphi[i] = INFINITY;
            sidx[i] = 0x7fffffff;
        }
        __syncthreads();
        for (int kk = 2; kk <= N; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < N / 2; t += 256) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;  // the t-th pair of this stage
                    const double fa = sf[i], fb = sf[l], pa = sphi[i], pb = sphi[l];
                    const int ia = sidx[i], ib = sidx[l];
                    const bool greater = pa > pb || (pa == pb && (fa > fb || (fa == fb && ia > ib)));
                    const bool up = (i & kk) == 0;
                    if (greater == up) {
                        sf[i] = fb;
                        sf[l] = fa;
                        sphi[i] = pb;
                        sphi[l] = pa;
                        sidx[i] = ib;
                        sidx[l] = ia;
                    }
                }
                __syncthreads();
            }
    }
    if (tid == 0) s_swapped = 1;
    __syncthreads();
    for (int ph0 = 0; ph0 < (((a.dbg & 1) || plain_sort) ? 0 : lam); ph0 += 4) {
        if ((ph0 & 15) == 0) {
            const int sw = s_swapped;
            __syncthreads();
            if (!sw) break;  // sixteen phases without a swap: sorted under the drawn rules
            if (tid == 0) s_swapped = 0;
            __syncthreads();
        }
        unsigned c4[4] = {0u, 0u, 0u, 0u};
        bool drawn = false;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int ph = ph0 + q4;
            if (ph < lam) {
                for (int p = tid; 2 * p + (ph & 1) + 1 < lam; p += 256) {
                    const int j = 2 * p + (ph & 1);
                    const double fa = sf[j], fb = sf[j + 1], pa = sphi[j], pb = sphi[j + 1];
                    bool by_f = pa == 0.0 && pb == 0.0;
                    if (!by_f) {
                        if (!drawn || p >= 256) {  // (p >= 256 only for lam > 512: one call per pair and phase there)
                            c4[0] = (unsigned)p;
                            c4[1] = (unsigned)(p < 256 ? ph0 : ph);
                            c4[2] = (unsigned)(gen * 16 + 1);
                            c4[3] = (unsigned)run;
                            philox(c4, (unsigned)a.seed, (unsigned)(a.seed >> 32));
                            drawn = p < 256;
                        }
                        by_f = ((double)c4[p < 256 ? q4 : 0] + 0.5) * (1.0 / 4294967296.0) < 0.45;
                    }
                    const bool worse = by_f ? (fa > fb) : (pa > pb);
                    if (worse) {
                        const int ia = sidx[j], ib = sidx[j + 1];
                        sf[j] = fb;
                        sf[j + 1] = fa;
                        sphi[j] = pb;
                        sphi[j + 1] = pa;
                        sidx[j] = ib;
                        sidx[j + 1] = ia;
                        s_swapped = 1;
                    }
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- stop like NLopt's xtol_rel, on the survivors' spread (one thread per variable)
    if (tid == 0) s_swapped = 0;  // reused: number of variables whose spread is still above the tolerance
    __syncthreads();
    for (int c = tid; c < n; c += 256) {
        double lo = INFINITY, hi = -INFINITY;
        for (int s2 = 0; s2 < mu; ++s2) {
            const double v = X[(size_t)sidx[s2] * n + c];
            lo = fmin(lo, v);
            hi = fmax(hi, v);
        }
        if (hi - lo > a.xtol_rel * fmax(fabs(X[(size_t)sidx[0] * n + c]), 1e-300)) atomicAdd(&s_swapped, 1);
    }
    __syncthreads();
    if (s_swapped == 0) {
        if (tid == 0) R.stat[1] = 1;
        return;
    }
    // ---- next generation
    const double tau = 1.0 / sqrt(2.0 * sqrt((double)n)), taup = 1.0 / sqrt(2.0 * (double)n), alpha = 0.2, gamma = 0.85;
    const int nd = mu - 1;
    for (int o = tid; o < ((a.dbg & 2) ? 0 : lam); o += 256) {
        const int par = sidx[o % mu];
        const double *xp = X + (size_t)par * n, *sp = S + (size_t)par * n;
        double *xo = Xn + (size_t)o * n, *so = Sn + (size_t)o * n;
        const int skip = R.kind == 1 ? 1 : 0;
        double *xe = a.Xeval + (size_t)(R.off + o) * a.d;  // next generation's row of the evaluation batch (x part)
        if (o < nd) {
            // differential variation towards the best individual; kept only if it stays inside the box
            const double *xb = X + (size_t)sidx[0] * n, *xq = X + (size_t)sidx[o + 1] * n;
            bool inside = true;
            for (int c = 0; c < n; ++c) {
                const double v = xp[c] + gamma * (xb[c] - xq[c]);
                inside = inside && v >= lo_at(c) && v <= hi_at(c);
            }
            for (int c = 0; c < n; ++c) {
                const double v = inside ? xp[c] + gamma * (xb[c] - xq[c]) : xp[c];
                xo[c] = v;
                so[c] = sp[c];
                if (c >= skip) xe[c - skip] = v;
            }
        } else {
            double u0, u1, zg, z1;
            rng4(a, run, gen, o, n, 2, u0, u1, zg, z1);  // the individual's global factor
            for (int c = 0; c < n; ++c) {
                const double lo = lo_at(c), hi = hi_at(c);
                double z0, zz;
                rng4(a, run, gen, o, c, 3, u0, u1, z0, zz);
                double s = sp[c] * (double)__expf((float)(taup * zg + tau * z0));
                s = fmin(s, (hi - lo) / sqrt((double)n));
                double v = xp[c] + s * zz;
                for (int tr = 0; tr < 10 && (v < lo || v > hi); ++tr) {  // re-draw components that leave the box
                    double w0, w1;
                    rng4(a, run, gen, o, c, 4 + tr, u0, u1, w0, w1);
                    v = xp[c] + s * w0;
                }
                if (v < lo || v > hi) v = xp[c];
                xo[c] = v;
                if (c >= skip) xe[c - skip] = v;
                so[c] = sp[c] + alpha * (s - sp[c]);  // exponential smoothing
            }
        }
    }
}

}  // namespace ps

// host-side gradient polish of the PS solution (the reference hands a local NLopt algorithm, descent.jl:560-569): projected steps on
// chi = [t; x] that keep every iterate feasible; each trial costs one batched value (+ Jacobian) call
static int ps_polish(mrbf_ctx *ctx, const mrbf_model *M, const std::vector<double> &lb, const std::vector<double> &ub, const std::vector<double> &mx,
                     const std::vector<double> &r, int max_evals, double &tau, std::vector<double> &x, int *evals_out) {
    const int d = M->d, k = M->k;
    double *dX, *dV, *dJ;
    MRBF_TRY(get_buf(ctx, S_PS_POLISH, (size_t)d + k + (size_t)k * d, &dX));
    dV = dX + d;
    dJ = dV + k;
    std::vector<double> F(k), J((size_t)k * d), Ft(k), xt(d), dir(d);
    int evals = 0;
    auto eval = [&](const std::vector<double> &xx, std::vector<double> &out, bool jac) -> int {
        MRBF_HIP(ctx, hipMemcpyAsync(dX, xx.data(), d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        MRBF_TRY(eval_model(ctx, M, 1, dX, dV, jac ? dJ : nullptr, nullptr));
        MRBF_HIP(ctx, hipMemcpyAsync(out.data(), dV, k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (jac) MRBF_HIP(ctx, hipMemcpyAsync(J.data(), dJ, (size_t)k * d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ++evals;
        return 0;
    };
    while (evals < max_evals) {
        MRBF_TRY(eval(x, F, true));
        int act = 0;
        for (int l = 1; l < k; ++l)
            if ((F[l] - mx[l]) / r[l] > (F[act] - mx[act]) / r[act]) act = l;
        double nrm = 0.0;
        for (int t = 0; t < d; ++t) {
            dir[t] = -J[(size_t)t * k + act] / r[act];  // per point k x d column-major block
            nrm += dir[t] * dir[t];
        }
        if (nrm == 0.0) break;
        double step = 1.0;
        bool improved = false;
        for (int it = 0; it < 20 && evals < max_evals; ++it) {
            for (int t = 0; t < d; ++t) xt[t] = std::min(std::max(x[t] + step * dir[t], lb[t]), ub[t]);
            MRBF_TRY(eval(xt, Ft, false));
            double tt = -1.0;
            for (int l = 0; l < k; ++l) tt = std::max(tt, (Ft[l] - mx[l]) / r[l]);
            tt = std::min(std::max(tt, -1.0), 0.0);
            bool feas = true;
            for (int l = 0; l < k; ++l) feas = feas && (Ft[l] - mx[l] - tt * r[l] <= 1e-14);
            if (feas && tt < tau - 1e-12) {
                x = xt;
                tau = tt;
                improved = true;
                break;
            }
            step *= 0.5;
        }
        if (!improved) break;
    }
    *evals_out = evals;
    return 0;
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_ps_step(mrbf_ctx *ctx, const mrbf_model *model, const double *x_n, const double *lb_eff, const double *ub_eff,
                                const double *fx_n, const double *r_or_null, const mrbf_ps_options *opts, double *x_trial, double *mx_trial,
                                double *r_out, mrbf_ps_info *info) {
    if (!ctx) return -1;
    if (!model) return fail(ctx, -2, "model is NULL");
    if (!x_n) return fail(ctx, -3, "x_n is NULL");
    if (!lb_eff) return fail(ctx, -4, "lb_eff is NULL");
    if (!ub_eff) return fail(ctx, -5, "ub_eff is NULL");
    if (!r_or_null && !fx_n) return fail(ctx, -6, "fx_n is NULL but no direction was given");
    if (!opts) return fail(ctx, -8, "opts is NULL");
    if (opts->t0 < -1.0 || opts->t0 > 0.0) return fail(ctx, -8, "opts.t0 must lie in [-1, 0]");
    if (!x_trial) return fail(ctx, -9, "x_trial is NULL");
    if (!mx_trial) return fail(ctx, -10, "mx_trial is NULL");
    if (!info) return fail(ctx, -12, "info is NULL");
    (void)hipSetDevice(ctx->device);
    using namespace ps;
    const int d = model->d, k = model->k;
    std::memset(info, 0, sizeof(*info));
    if (20 * (d + 2) > MAXLAM || k > 8) return fail(ctx, -2, "mrbf_ps_step: d = %d (limit %d) or k = %d (limit 8) too large for the device path", d, MAXLAM / 20 - 2, k);
    for (int t = 0; t < d; ++t)
        if (!(lb_eff[t] <= ub_eff[t])) return fail(ctx, -4, "lb_eff[%d] > ub_eff[%d]", t, t);
    hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1];
    MRBF_HIP(ctx, hipEventRecord(e0, ctx->stream));
    const bool need_ideal = r_or_null == nullptr;
    const int lam_ip = 20 * (d + 1), lam_ps = 20 * (d + 2);
    const int rows = (need_ideal ? k * lam_ip : 0) + lam_ps;
    // device arena: box, x_n, mx, r, evaluation batch, results, per-run state
    const size_t per_ip = (size_t)4 * lam_ip * d + d + 2, per_ps = (size_t)4 * lam_ps * (d + 1) + d + 3;
    const size_t cnt = (size_t)3 * d + 2 * k + (size_t)rows * d + (size_t)rows * k + k * per_ip + per_ps + 64;
    double *base;
    int *stat;
    MRBF_TRY(get_buf(ctx, S_PS_STATE, cnt, &base));
    MRBF_TRY(get_buf(ctx, S_PS_STAT, (size_t)4 * 9, &stat));
    double *dlb = base, *dub = dlb + d, *dxn = dub + d, *dmx = dxn + d, *dr = dmx + k, *Xeval = dr + k, *F = Xeval + (size_t)rows * d;
    double *pool = F + (size_t)rows * k;
    MRBF_HIP(ctx, hipMemcpyAsync(dlb, lb_eff, d * sizeof(double), hipMemcpyDefault, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(dub, ub_eff, d * sizeof(double), hipMemcpyDefault, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(dxn, x_n, d * sizeof(double), hipMemcpyDefault, ctx->stream));
    // mx = m(x_n)
    MRBF_TRY(eval_model(ctx, model, 1, dxn, dmx, nullptr, nullptr));
    std::vector<double> mx(k), r(k), hlb(d), hub(d), hxn(d), hfx(k);
    if (fx_n) MRBF_HIP(ctx, hipMemcpyAsync(hfx.data(), fx_n, k * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (r_or_null) MRBF_HIP(ctx, hipMemcpyAsync(r.data(), r_or_null, k * sizeof(double), hipMemcpyDefault, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(mx.data(), dmx, k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(hlb.data(), dlb, d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(hub.data(), dub, d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(hxn.data(), dxn, d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int max_ip = opts->max_ideal_evals < 0 ? 500 * (d + 1) : opts->max_ideal_evals;   // descent.jl:527
    const int max_ps = opts->max_ps_evals < 0 ? 500 * (d + 1) : opts->max_ps_evals;          // descent.jl:416
    const double xtol = opts->xtol_rel > 0.0 ? opts->xtol_rel : 1e-3;                        // descent.jl:379, :485

    auto make_run = [&](Run &R, int kind, int obj, int lam, int nvar, int off, int max_evals, double *&p, int *st) {
        R.kind = kind;
        R.obj = obj;
        R.lam = lam;
        R.mu = (lam + 6) / 7;
        R.nvar = nvar;
        R.off = off;
        R.max_evals = max_evals;
        for (int b = 0; b < 2; ++b) {
            R.X[b] = p;
            p += (size_t)lam * nvar;
            R.S[b] = p;
            p += (size_t)lam * nvar;
        }
        R.best = p;
        p += nvar + 2;
        R.stat = st;
    };
    auto run_batch = [&](Args &a, const double *start, double t0, int max_gens) -> int {
        hipLaunchKernelGGL(ps_init_kernel, dim3((unsigned)a.nruns), dim3(256), 0, ctx->stream, a, start, t0);
        int rows_now = 0;
        for (int q2 = 0; q2 < a.nruns; ++q2) rows_now = std::max(rows_now, a.runs[q2].off + a.runs[q2].lam);
        std::vector<int> hstat((size_t)4 * a.nruns);
        for (int g = 0; g < max_gens; ++g) {
            a.gen = g;
            MRBF_TRY(eval_model(ctx, model, rows_now, a.Xeval, const_cast<double *>(a.F), nullptr, nullptr));
            hipLaunchKernelGGL(ps_step_kernel, dim3((unsigned)a.nruns), dim3(256), 0, ctx->stream, a);
            if ((g & 7) == 7 || g + 1 == max_gens) {  // status words every 8 generations: stop when every run is done
                MRBF_HIP(ctx, hipMemcpyAsync(hstat.data(), a.runs[0].stat, hstat.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
                bool all = true;
                for (int q2 = 0; q2 < a.nruns; ++q2) all = all && hstat[(size_t)4 * q2 + 1] != 0;
                if (all) break;
            }
        }
        MRBF_HIP(ctx, hipGetLastError());
        return 0;
    };

    Args a{};
    a.d = d;
    a.k = k;
    a.lb = dlb;
    a.ub = dub;
    a.mx = dmx;
    a.r = dr;
    a.F = F;
    a.Xeval = Xeval;
    a.seed = opts->seed;
    a.dbg = getenv("MRBF_PS_DBG") ? atoi(getenv("MRBF_PS_DBG")) : 0;
    a.xtol_rel = xtol;
    // ---- local ideal point: the k single-objective minimisations side by side (descent.jl:404-412)
    if (need_ideal) {
        double *p = pool;
        a.nruns = k;
        for (int l = 0; l < k; ++l) make_run(a.runs[l], 0, l, lam_ip, d, l * lam_ip, max_ip, p, stat + 4 * l);
        MRBF_TRY(run_batch(a, dxn, 0.0, (max_ip + lam_ip - 1) / lam_ip + 1));
        std::vector<double> bf(k);
        std::vector<int> hs((size_t)4 * k);
        for (int l = 0; l < k; ++l)
            MRBF_HIP(ctx, hipMemcpyAsync(&bf[l], a.runs[l].best + d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipMemcpyAsync(hs.data(), stat, hs.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int l = 0; l < k; ++l) {
            r[l] = hfx[l] - bf[l];  // r = f(x_n) - ideal point (descent.jl:536-538)
            info->evals_ideal += hs[(size_t)4 * l];
            info->generations += hs[(size_t)4 * l + 2];
        }
    }
    info->tau = 0.0;
    bool critical = false;
    for (int l = 0; l < k; ++l) critical = critical || !(r[l] > 0.0);
    std::vector<double> xt(hxn);
    if (critical) {
        info->status = MRBF_PS_CRITICAL;  // any(r .<= 0): omega = 0, the point itself (descent.jl:546-549)
    } else {
        MRBF_HIP(ctx, hipMemcpyAsync(dr, r.data(), k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        double *p = pool;
        a.nruns = 1;
        make_run(a.runs[0], 1, 0, lam_ps, d + 1, 0, max_ps, p, stat);
        MRBF_TRY(run_batch(a, dxn, opts->t0, (max_ps + lam_ps - 1) / lam_ps + 1));
        std::vector<double> best((size_t)d + 3);
        int hs[4];
        MRBF_HIP(ctx, hipMemcpyAsync(best.data(), a.runs[0].best, (d + 3) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipMemcpyAsync(hs, stat, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        info->evals_ps = hs[0];
        info->generations += hs[2];
        const double bf = best[d + 1], bphi = best[d + 2];
        if (!(bphi == 0.0) || !std::isfinite(bf)) {
            info->status = MRBF_PS_FAILURE;  // descent.jl:571-572
        } else {
            double tau = bf;
            for (int t = 0; t < d; ++t) xt[t] = best[1 + t];
            if (opts->max_polish_evals > 0) {
                int pe = 0;
                MRBF_TRY(ps_polish(ctx, model, hlb, hub, mx, r, opts->max_polish_evals, tau, xt, &pe));
                info->evals_polish = pe;
            }
            info->tau = tau;
            info->status = MRBF_PS_OK;
        }
    }
    // mx_trial = m(x_trial)
    if (info->status == MRBF_PS_OK) {
        MRBF_HIP(ctx, hipMemcpyAsync(dxn, xt.data(), d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        MRBF_TRY(eval_model(ctx, model, 1, dxn, dmx, nullptr, nullptr));
        MRBF_HIP(ctx, hipMemcpyAsync(mx.data(), dmx, k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    MRBF_HIP(ctx, hipEventRecord(e1, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MRBF_HIP(ctx, hipEventElapsedTime(&info->ms_total, e0, e1));
    auto put = [&](double *dst, const double *src, size_t c) -> int {
        MRBF_HIP(ctx, hipMemcpy(dst, src, c * sizeof(double), hipMemcpyDefault));
        return 0;
    };
    MRBF_TRY(put(x_trial, xt.data(), d));
    MRBF_TRY(put(mx_trial, mx.data(), k));
    if (r_out) MRBF_TRY(put(r_out, r.data(), k));
    return MRBF_OK;
}
