// Pascoletti-Serafini descent step with the whole subproblem solver on the device -- replaces the NLopt runs of
// compute_local_ideal_point / _ps_optimization inside get_criticality(::PascolettiSerafiniConfig, ...)
// (/root/reference/src/descent.jl:369-412, :434-510, :512-581).
//
// The reference hands one-point closures to NLopt's GN_ISRES: every candidate costs one sweep over all n centres PER OUTPUT.
// Here the subproblems are solved by an ISRES-style (mu, lambda) evolution strategy (Runarsson & Yao; population 20 (dim + 1),
// mu = lambda / 7, as NLopt's defaults) whose state lives in device memory:
//   * the k single-objective runs of the local ideal point and the Pascoletti-Serafini run each own a population block; all
//     blocks of a generation are evaluated by ONE batched surrogate sweep per grouped model (eval_model, values only);
//   * per generation three more launches do everything else, for all runs at once:
//       ps_score_kernel  one wave per individual: objective and constraint violation -- the PS constraints of descent.jl:443,
//                        the modelled (in)equality constraints (:448-466) and the MOP's linear constraints
//                        (AbstractMOPInterface.jl:487-495) -- into per-run arrays;
//       ps_rank_kernel   one 1024-thread workgroup per run: best so far, stop tests, stochastic ranking with the records in LDS
//                        (odd-even transposition with the random comparison rule -- the parallel form of the bubble sweeps --
//                        or, when no individual violates a constraint, a bitonic network that ends in the same order);
//       ps_breed_kernel  one wave per offspring: differential variation, log-normal self-adaptive mutation with re-draws inside
//                        the box, from a counter-based generator (Philox 4x32-10, keyed by seed / run / generation / individual /
//                        component: no state, any launch order);
//   * the host enqueues generations back to back and reads the runs' status words every few generations only.
// Populations up to MAXLAM individuals (d <= 356): the C4 / C5 dimensions of BASELINE.json (d = 128, 256; examples/example_zdt.jl:39)
// run here, not in a host loop.
// Parity with the reference is the contract of get_criticality (problem solved, budgets, start values, critical / failure
// short cuts, returned tuple), not NLopt's random trajectory; the returned point is always feasible for the subproblem.
#include "radial.hpp"

namespace mrbf {

int eval_model(mrbf_ctx *ctx, const mrbf_model *M, int64_t m, const double *Xdev, double *vals_dev, double *jac_dev,
               mrbf_eval_info *info);

namespace ps {

constexpr int MAXLAM = 7168;   // population limit: the ranking keeps one run's records in one workgroup's LDS
constexpr int MAXRUNS = 9;     // k <= 8 ideal-point runs side by side, or the PS run
constexpr int MAXMODELS = 8;   // grouped models of one container
constexpr int MAXOBJ = 8;
constexpr int MAXCON = 32;     // modelled (nonlinear) constraint rows
constexpr int RANK_THREADS = 1024;
// ranking of a large population on several compute units (ps_rank_sort_kernel): workgroups per run, phases per chunk (= halo width),
// smallest population that takes this path, longest window (owned part + two halos) of one workgroup
constexpr int RS_THREADS = 512, RS_W = 16, RS_B = 256, RS_MINLAM = 1024;
constexpr int RS_MAXWIN = ((MAXLAM + RS_W - 1) / RS_W + 2) + 2 * RS_B;
constexpr int RS_SYNC = 64;  // sync words per run: [0] arrivals, [1] failure, [2] the run asks for the sort, [3] buffer that holds the result, [4 + c] chunk c moved something

struct RankWs {  // device work space of the multi-workgroup ranking, run r at offset r * MAXLAM (r * RS_SYNC)
    double *f[2], *phi[2];
    int *idx[2];
    int *sync;
    int *cnt;  // ps_rank_wave_kernel's keys: per run MAXLAM counts for f, MAXLAM for phi
    int count_plain;  // ps_rank_prep_kernel is part of the launch sequence: a generation without violations is ranked by counting there
};

struct Run {  // one (mu, lambda) run; all pointers into device arenas
    int nvar;       // 1 + d for the PS run (chi = [t; x]), d for an ideal-point run
    int lam, mu;
    int kind;       // 0: minimise objective `obj` over the box; 1: Pascoletti-Serafini run
    int obj;
    int off;        // first row of this run's block in the evaluation batch
    int max_evals;
    double *X[2];   // lam x nvar, ping-pong
    double *S[2];   // lam x nvar step sizes
    double *best;   // nvar + 2: best_x, best_f, best_phi
    double *f, *phi;  // lam: objective / violation of the current generation
    int *order;     // lam: ranked individuals (best first); the first mu are the parents
    int *stat;      // [0] evals so far, [1] done, [2] generations run, [3] generation + 1 whose offspring are to be bred
};

struct Args {
    Run runs[MAXRUNS];
    int nruns, d, nobj, rows;
    const double *lb, *ub;   // d
    const double *mx, *r;    // nobj (PS run)
    // evaluation batch results: one rows x kf[j] block per grouped model
    const double *F[MAXMODELS];
    int kf[MAXMODELS];
    int nmodels;
    short obj_model[MAXOBJ], obj_col[MAXOBJ];   // where objective l lives
    int ncon;                                   // modelled constraints: (model, column, 1 = equality)
    short con_model[MAXCON], con_col[MAXCON], con_eq[MAXCON];
    int nlin_eq, nlin_ineq;                     // A x = b, A x <= b in scaled variables (row-major rows x d)
    const double *A_eq, *b_eq, *A_ineq, *b_ineq;
    double eq_tol;
    double *Xeval;           // evaluation batch, rows x d
    int score_fused;         // no linear constraints: ps_rank_kernel scores its own run's individuals (no ps_score_kernel launch)
    double *Xq, *xsq;        // non-null: the breeding kernel also writes the batch centred and padded (rows x xqD, squared norms) for the one model
    const double *xmean;     //           whose evaluation reads it (the arithmetic of center_pad_kernel, prep.hip)
    int xqD;
    unsigned long long seed;
    double xtol_rel;
    int gen;
    int dbg;  // experiments (MRBF_PS_DBG): 128 the plain sort as one pair per thread through LDS (the form before round 6), 64 the multi-workgroup ranking gives up at once (fallback test), 1 no ranking, 2 no breeding, 4 transposition phases also when a sort would do, 8 t stays a free
              // variable (no repair), 16 uniform start population
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
__device__ __forceinline__ int run_of_row(const Args &a, int row) {  // runs are laid out in row order
    int run = 0;
    while (run + 1 < a.nruns && row >= a.runs[run + 1].off) ++run;
    return run;
}

// generation 0: uniform population in the box, individual 0 = the start point (PS: [t0; x_n]), PS individual 1 = [0; x_n]
// (feasible for the PS constraints: m(x_n) - m(x_n) - 0 r = 0); step sizes (ub - lb) / sqrt(n)
__global__ __launch_bounds__(256) void ps_init_kernel(Args a, const double *xn, double t0) {
    const Run &R = a.runs[blockIdx.y];
    const int n = R.nvar;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < R.lam * n) {
        const int i = e / n, j = e % n;
        const double lo = lo_of(a, R, j), hi = hi_of(a, R, j);
        double u0, u1, z0, z1;
        rng4(a, blockIdx.y, 0, i, j, 15, u0, u1, z0, z1);
        double v = lo + u0 * (hi - lo);
        double step = (hi - lo) / sqrt((double)n);
        if (i == 0 || (i == 1 && R.kind == 1)) {
            v = (R.kind == 1) ? (j == 0 ? (i == 0 ? t0 : 0.0) : xn[j - 1]) : xn[j];
            v = fmin(fmax(v, lo), hi);
        } else if ((i & 1) && !(a.dbg & 16)) {
            // every second individual is a mutation of the start point with a scale from 1/4 of the box down to 2^-11 of it (NLopt's
            // ISRES starts its whole population at x0 too): at d >= 64 a uniform population holds no point that improves every
            // objective at once, i.e. no feasible chi with t < 0, and the run ended where it began (omega = 0 at d = 64 .. 256)
            const double sigma = 0.25 * exp2(-(double)(((i >> 1) - 1) % 10));
            const double xs = (R.kind == 1) ? (j == 0 ? 0.0 : xn[j - 1]) : xn[j];
            v = (R.kind == 1 && j == 0) ? 0.0 : fmin(fmax(xs + sigma * z0 * (hi - lo), lo), hi);
            step *= sigma;
        }
        R.X[0][e] = v;
        R.S[0][e] = step;
        const int skip = R.kind == 1 ? 1 : 0;  // the evaluation batch holds the x part of every individual of every run
        if (j >= skip) a.Xeval[(size_t)(R.off + i) * a.d + (j - skip)] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        R.best[n] = INFINITY;
        R.best[n + 1] = INFINITY;
        R.stat[0] = 0;
        R.stat[1] = 0;
        R.stat[2] = 0;
        R.stat[3] = 0;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- objective and constraint violation of individual i (row `row` of the evaluation batch) of run R.  WAVE: the 64 lanes of a wave
// work on the row together (the linear constraints' dot products are summed over the lanes); otherwise one thread does the row --
// problems without linear constraints only (ps_rank_kernel scores its own run that way: a launch per generation less).
template <bool WAVE>
__device__ __forceinline__ void score_row(const Args &a, const Run &R, int i, int row, int lane, double &f_out, double &phi_out) {
    const int m = min(R.lam, R.max_evals - R.stat[0]);  // individuals of this generation inside the budget
    double f = INFINITY, phi = INFINITY;
    if (i < m) {
        if (R.kind == 0) {
            f = a.F[a.obj_model[R.obj]][(size_t)row * a.kf[a.obj_model[R.obj]] + a.obj_col[R.obj]];
            phi = 0.0;
        } else {
            double t = R.X[a.gen & 1][(size_t)i * R.nvar];
            if (!(a.dbg & 8)) {
                // the subproblem is min_x max_l (m_l(x) - m_l(x_n)) / r_l in disguise: the best t an x admits is known once x has been
                // evaluated, so the individual carries THAT t (pulled a hair towards 0 so that rounding cannot make it infeasible);
                // an x that worsens some objective keeps t = 0 and is ranked by its violation
                double ts = -INFINITY;
                for (int l = 0; l < a.nobj; ++l)
                    ts = fmax(ts, (a.F[a.obj_model[l]][(size_t)row * a.kf[a.obj_model[l]] + a.obj_col[l]] - a.mx[l]) / a.r[l]);
                if (ts == ts) {
                    t = ts <= 0.0 ? fmax(ts * (1.0 - 1e-14), -1.0) : 0.0;
                    if (lane == 0) R.X[a.gen & 1][(size_t)i * R.nvar] = t;
                }
            }
            f = t;
            phi = 0.0;
            for (int l = 0; l < a.nobj; ++l) {
                const double g = a.F[a.obj_model[l]][(size_t)row * a.kf[a.obj_model[l]] + a.obj_col[l]] - a.mx[l] - t * a.r[l];
                phi += g > 0.0 ? g * g : 0.0;  // m_l(x) - m_l(x_n) - t r_l <= 0  (descent.jl:443)
            }
        }
        for (int c = 0; c < a.ncon; ++c) {  // modelled constraints (descent.jl:448-466): h(x) = 0, g(x) <= 0
            const double v = a.F[a.con_model[c]][(size_t)row * a.kf[a.con_model[c]] + a.con_col[c]];
            if (a.con_eq[c])
                phi += fabs(v) > a.eq_tol ? v * v : (v == v ? 0.0 : INFINITY);
            else
                phi += v > 0.0 ? v * v : (v == v ? 0.0 : INFINITY);
        }
        if constexpr (WAVE) {
            const double *x = a.Xeval + (size_t)row * a.d;
            for (int c = 0; c < a.nlin_eq + a.nlin_ineq; ++c) {  // linear constraints of the MOP in scaled variables
                const bool eq = c < a.nlin_eq;
                const double *Ar = eq ? a.A_eq + (size_t)c * a.d : a.A_ineq + (size_t)(c - a.nlin_eq) * a.d;
                double s = 0.0;
                for (int j = lane; j < a.d; j += 64) s = fma(Ar[j], x[j], s);
                s = wave_sum(s) - (eq ? a.b_eq[c] : a.b_ineq[c - a.nlin_eq]);
                if (eq)
                    phi += fabs(s) > a.eq_tol ? s * s : (s == s ? 0.0 : INFINITY);
                else
                    phi += s > 0.0 ? s * s : (s == s ? 0.0 : INFINITY);
            }
        }
        if (!(f == f) || !(phi == phi) || fabs(f) == INFINITY || fabs(phi) == INFINITY) {
            f = INFINITY;
            phi = INFINITY;
        }
    }
    f_out = f;
    phi_out = phi;
}

// one wave per row of the evaluation batch (problems with linear constraints; without, ps_rank_kernel scores its run itself)
__global__ __launch_bounds__(256) void ps_score_kernel(Args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int run = run_of_row(a, row);
    const Run &R = a.runs[run];
    if (R.stat[1]) return;  // finished earlier
    const int i = row - R.off;
    double f, phi;
    score_row<true>(a, R, i, row, lane, f, phi);
    if (lane == 0) {
        R.f[i] = f;
        R.phi[i] = phi;
    }
}

// record order of the best-so-far bookkeeping: feasible beats infeasible, then the objective (feasible) or the violation
// (infeasible), ties by index
struct Cand {
    double key;  // f when feasible, phi otherwise
    int feas;    // 1 feasible, 0 infeasible, -1 none
    int idx;
};
__device__ __forceinline__ bool cand_better(const Cand &x, const Cand &y) {
    if (y.feas < 0) return x.feas >= 0;
    if (x.feas < 0) return false;
    if (x.feas != y.feas) return x.feas > y.feas;
    if (x.key != y.key) return x.key < y.key;
    return x.idx < y.idx;
}

// ---- the bitonic network of the plain sort with E elements per thread in registers (position p = tid * E + e): compare-exchanges at
// distance j < E stay inside the thread, j < 64 E go through wave shuffles, only the rest through LDS with a barrier -- 10 of the 78
// stages at N = 4096 (the one-pair-per-thread form paid an LDS round trip and a sixteen-wave barrier per stage: 60-80 us per generation at
// d = 128, 2.4 ms per step).  Same network, same comparator (objective, ties by index): the same order.  On return sidx[p] holds the order.
__device__ __forceinline__ int npow2(int lam) {
    int N = 1;
    while (N < lam) N <<= 1;
    return N;
}
template <int J, int E>
__device__ __forceinline__ void bitonic_inthread(double (&f)[E], int (&ix)[E], int base, int kk) {
#pragma unroll
    for (int e = 0; e < E; ++e)
        if ((e & J) == 0) {
            const int l = e | J;
            const bool greater = f[e] > f[l] || (f[e] == f[l] && ix[e] > ix[l]);
            const bool up = ((base + e) & kk) == 0;
            if (greater == up) {
                const double tf = f[e];
                f[e] = f[l];
                f[l] = tf;
                const int ti = ix[e];
                ix[e] = ix[l];
                ix[l] = ti;
            }
        }
}
template <int E>
__device__ __forceinline__ void bitonic_regs(const double *__restrict__ fin, const int *__restrict__ iin, int lam, int N, double *sf, int *sidx) {
    const int tid = threadIdx.x, base = tid * E;
    double f[E];
    int ix[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int p = base + e;
        f[e] = p < lam ? fin[p] : INFINITY;
        ix[e] = p < lam ? (iin ? iin[p] : p) : 0x7fffffff;
    }
    if (iin) __syncthreads();  // (the inputs may live where the stages exchange their elements)
    for (int kk = 2; kk <= N; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j < E) {
                if constexpr (E > 1) {
                    if (j == 1) bitonic_inthread<1, E>(f, ix, base, kk);
                }
                if constexpr (E > 2) {
                    if (j == 2) bitonic_inthread<2, E>(f, ix, base, kk);
                }
                if constexpr (E > 4) {
                    if (j == 4) bitonic_inthread<4, E>(f, ix, base, kk);
                }
            } else if (j < 64 * E) {
                const int dl = j / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = base + e;
                    const double of = __shfl_xor(f[e], dl);
                    const int oi = __shfl_xor(ix[e], dl);
                    const bool keep_min = ((p & j) == 0) == ((p & kk) == 0);
                    const bool greater = f[e] > of || (f[e] == of && ix[e] > oi);
                    if (keep_min == greater) {
                        f[e] = of;
                        ix[e] = oi;
                    }
                }
            } else {
                __syncthreads();  // (the partners' reads of the stage before)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    sf[base + e] = f[e];
                    sidx[base + e] = ix[e];
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = base + e, q = p ^ j;
                    const double of = sf[q];
                    const int oi = sidx[q];
                    const bool keep_min = ((p & j) == 0) == ((p & kk) == 0);
                    const bool greater = f[e] > of || (f[e] == of && ix[e] > oi);
                    if (keep_min == greater) {
                        f[e] = of;
                        ix[e] = oi;
                    }
                }
            }
        }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) sidx[base + e] = ix[e];
    __syncthreads();
}

// ---- only the mu best individuals of a plain sort are ever read (the parents, and the spread test over them): find them without sorting
// the other six sevenths.  A threshold from blockDim evenly spaced samples (each sample ranks itself among the samples by counting:
// no sort), placed three standard deviations of the sample quantile above rank mu; one pass counts and compacts the individuals at or
// below it (ties included, so the compacted set always CONTAINS the mu best with their index order intact); if that set holds at least mu
// and at most blockDim individuals it is padded to blockDim and sorted one element per thread (shuffles up to distance 32, LDS beyond:
// 10 barrier stages of 55) -- else the caller sorts everything as before.  The parents and their order are those of the full sort.
// scratch: 2 blockDim doubles + 2 blockDim ints behind `sf` (the caller's N >= 2 blockDim doubles + N ints cover it).
__device__ __forceinline__ bool select_parents(const double *__restrict__ fin, int lam, int mu, double *sf, int *sidx_out) {
    __shared__ int s_cnt[32];
    __shared__ double s_thr;
    const int tid = threadIdx.x, NT = (int)blockDim.x, lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
    double *ssamp = sf, *csf = sf + NT;
    int *cidx = reinterpret_cast<int *>(sf + 2 * NT);
    // target rank: mu plus three standard deviations of the S-sample quantile estimate (in ranks)
    const int S = NT;
    const double p0 = (double)mu / lam;
    const double sd = lam * sqrt(fmax(p0 * (1.0 - p0), 0.0) / S);
    const double T = mu + 3.0 * sd + 2.0 * lam / S;
    if (!(T <= 0.9 * NT) || lam <= NT) return false;
    const int rstar = min(S - 1, (int)ceil(T * S / lam));
    const double mine = fin[(int)(((long long)tid * lam) / S)];
    ssamp[tid] = mine;
    __syncthreads();
    int rank = 0;
    for (int q = 0; q < S; ++q) {
        const double o = ssamp[q];  // (broadcast read)
        rank += (o < mine || (o == mine && q < tid)) ? 1 : 0;
    }
    if (rank == rstar) s_thr = mine;
    __syncthreads();
    const double thr = s_thr;
    int cnt = 0;
    for (int i = tid; i < lam; i += NT) cnt += fin[i] <= thr ? 1 : 0;
    // exclusive scan of the per-thread counts: inside the wave by shuffles, across the waves through LDS
    int incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_cnt[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w2 = 0; w2 < nw; ++w2) {
        if (w2 < wave) wbase += s_cnt[w2];
        total += s_cnt[w2];
    }
    if (total < mu || total > NT) return false;  // (uniform: every thread sees the same counts)
    int pos = wbase + incl - cnt;
    for (int i = tid; i < lam; i += NT) {
        const double v = fin[i];
        if (v <= thr) {
            csf[pos] = v;
            cidx[pos] = i;
            ++pos;
        }
    }
    for (int i = total + tid; i < NT; i += NT) {
        csf[i] = INFINITY;
        cidx[i] = 0x7fffffff;
    }
    __syncthreads();
    // (a thread's compacted elements are not in index order across threads -- thread t holds i = t, t + NT, .. -- which is fine: the
    // sort's comparator is (objective, index))
    bitonic_regs<1>(csf, cidx, NT, NT, ssamp, sidx_out);
    return true;
}

// ---- one workgroup per run: best so far, stop tests, ranking
// mode 0: everything in this launch.  Large populations with infeasible individuals (mode 1 / 2): mode 1 does the bookkeeping and, when
// the transposition phases are due, hands them to ps_rank_sort_kernel (sixteen workgroups per run); mode 2 picks the order up (or runs the
// phases here after all if that kernel gave up: it never touches f / phi) and finishes the generation.
constexpr int RW_OWN = 64, RW_H = 32;
constexpr int RW_CNT_T = 256;  // individuals per counting workgroup
__host__ __device__ inline int rw_waves(int lam) { return (lam + RW_OWN - 1) / RW_OWN; }
__host__ __device__ inline int rw_pitch(int lam) { return ((lam / 2 + RW_OWN + 63) / 64) * 64; }  // pair slots of a group, plus the last window's overhang
__host__ __device__ inline int rw_jsplit(int lam) { return lam > 2048 ? 32 : 8; }                 // the counting's split of the "other individual" loop

typedef unsigned long long u64;
// lane mask in a scalar register pair ? a : b
__device__ __forceinline__ unsigned msel(u64 m, unsigned a, unsigned b) {
    unsigned d;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m));
    return d;
}
// 16-bit compares of the two halves of a key, straight into a lane mask
__device__ __forceinline__ u64 gt_hi16(unsigned a, unsigned b) {
    u64 m;
    asm("v_cmp_gt_u16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_1" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ u64 gt_lo16(unsigned a, unsigned b) {
    u64 m;
    asm("v_cmp_gt_u16_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ u64 zero_lo16(unsigned a) {
    u64 m;
    asm("v_cmp_eq_u16_e64 %0, 0, %1" : "=s"(m) : "v"(a));
    return m;
}
// (bound_ctrl: the lane without a source reads 0 -- its pair is masked off -- and the move needs no initialised destination)
__device__ __forceinline__ unsigned dpp_from_next(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }  // wave_shl:1 -- lane l reads lane l + 1
__device__ __forceinline__ unsigned dpp_from_prev(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }  // wave_shr:1 -- lane l reads lane l - 1

// 4 NG phases on a window (KA / KB the keys, IA / IB the individuals of this lane's two positions).  TAIL: the ranking's last, short
// block -- phases from nph on do nothing.
template <bool TAIL, int NG>
__device__ __forceinline__ void rw_block(unsigned &KA, unsigned &KB, unsigned &IA, unsigned &IB, u64 dbits, int nph, u64 mPairE, u64 mPairO, u64 mOwnE,
                                         u64 mOwnO, u64 &moved) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const unsigned dg = (unsigned)(dbits >> (8 * g));  // (g < 4: low word, else high word -- resolved at compile time)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u64 live = (!TAIL || 4 * g + q < nph) ? ~0ull : 0ull;
            const u64 mDraw = __builtin_amdgcn_ballot_w64((dg & (1u << q)) != 0u);
            if ((q & 1) == 0) {
                const u64 gtf = gt_hi16(KA, KB), gtp = gt_lo16(KA, KB);
                const u64 byf = zero_lo16(KA | KB) | mDraw;
                u64 worse = mPairE & (gtp ^ (byf & (gtf ^ gtp)));
                if (TAIL) worse &= live;
                const unsigned tk = KA, ti = IA;
                KA = msel(worse, KB, KA);
                IA = msel(worse, IB, IA);
                KB = msel(worse, tk, KB);
                IB = msel(worse, ti, IB);
                moved |= worse & mOwnE;
            } else {
                const unsigned KN = dpp_from_next(KA);
                const u64 gtf = gt_hi16(KB, KN), gtp = gt_lo16(KB, KN);
                const u64 byf = zero_lo16(KB | KN) | mDraw;
                u64 worse = mPairO & (gtp ^ (byf & (gtf ^ gtp)));
                if (TAIL) worse &= live;
                const u64 wprev = worse << 1;  // lane l + 1 takes lane l's second record when lane l's pair swaps
                const unsigned IN = dpp_from_next(IA), KP = dpp_from_prev(KB), IP = dpp_from_prev(IB);
                KB = msel(worse, KN, KB);
                IB = msel(worse, IN, IB);
                KA = msel(wprev, KP, KA);
                IA = msel(wprev, IP, IA);
                moved |= worse & mOwnO;
            }
        }
    }
}


// ---- populations below RS_MINLAM: the same phases inside ps_rank_kernel's ONE workgroup (round 6).  The one-pair-per-thread loop
// through LDS costs 0.35 us per phase (a barrier of the whole workgroup per phase: 99 us per ranking at lam = 280, the largest item of
// a generation at d = 12); here wave w < ceil(lam / 96) holds a window of 128 two-word records in registers (its own 96, 16 of either
// neighbour), runs sixteen phases on them without a barrier (rw_block), and the waves exchange their parts through LDS at ONE
// workgroup barrier per sixteen phases -- which is also where this population size tests for sixteen phases without a swap.  Keys by
// counting and draws (one chunk of 256 phases at a time) are made by all threads of the workgroup.  Same comparisons, same draws,
// same exit rule as the loop it replaces (MRBF_PS_DBG 2048 keeps that loop; tests compare both with the NumPy oracle).
constexpr int RWS_OWN = 96, RWS_H = 16;
__host__ __device__ inline int rws_waves(int lam) { return (lam + RWS_OWN - 1) / RWS_OWN; }
__host__ __device__ inline int rws_pitch(int lam) { return ((lam / 2 + 64 + 7) / 8) * 8; }
__host__ __device__ inline size_t rws_smem_bytes(int lam) {  // f, phi | keys | two exchange buffers (key, index) | order | draws of a chunk | words
    const size_t lp = (size_t)((lam + 1) & ~1);
    return 16 * lp + 4 * lp + 16 * lp + 4 * lp + (size_t)64 * rws_pitch(lam) + 64;
}
// false: the generation holds a NaN (no rank): the caller's loop ranks it.  On success sidx_out[0 .. lam) is the order.
__device__ __forceinline__ bool rank_small_waves(const Args &a, const Run &R, int run, double *smem, int *&sidx_out) {
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int lam = R.lam, lp = (lam + 1) & ~1, nw = rws_waves(lam), pitch = rws_pitch(lam), nslots = lam / 2;
    double *sf = smem, *sphi = sf + lp;
    unsigned *skey = reinterpret_cast<unsigned *>(sphi + lp);
    unsigned *xk = skey + lp;        // [2][lp]
    unsigned *xi = xk + 2 * lp;      // [2][lp]
    int *sidx = reinterpret_cast<int *>(xi + 2 * lp);
    unsigned char *sdraw = reinterpret_cast<unsigned char *>(sidx + lp);
    int *s_moved = reinterpret_cast<int *>(sdraw + (size_t)64 * pitch);  // three words, used in turn
    bool nan = false;
    for (int i = tid; i < lam; i += NT) {
        const double fi = R.f[i], pi = R.phi[i];
        sf[i] = fi;
        sphi[i] = pi;
        nan = nan || fi != fi || pi != pi;
    }
    if (tid < 3) s_moved[tid] = 0;
    if (__syncthreads_or(nan ? 1 : 0)) return false;
    // keys: rank(f) << 16 | (0 if phi == 0, else 1 + rank(phi)), rank(x) = #{j : x_j < x}
    for (int i = tid; i < lam; i += NT) {
        const double fi = sf[i], pi = sphi[i];
        int cf = 0, cp = 0;
        for (int j = 0; j < lam; ++j) {
            cf += sf[j] < fi ? 1 : 0;
            cp += sphi[j] < pi ? 1 : 0;
        }
        skey[i] = ((unsigned)cf << 16) | (pi == 0.0 ? 0u : 1u + (unsigned)cp);
    }
    __syncthreads();
    const bool ranks = wave < nw;
    const int s0 = wave * RWS_OWN, e0 = min(lam, s0 + RWS_OWN), ws0 = s0 - RWS_H;
    const int gA = ws0 + 2 * lane, gB = gA + 1;
    const bool inA = ranks && gA >= 0 && gA < lam, inB = ranks && gB >= 0 && gB < lam;
    const bool ownA = ranks && gA >= s0 && gA < e0, ownB = ranks && gB >= s0 && gB < e0;
    const u64 mPairE = __builtin_amdgcn_ballot_w64(inA && inB), mPairO = __builtin_amdgcn_ballot_w64(inB && gB + 1 < lam && lane < 63);
    const u64 mOwnE = __builtin_amdgcn_ballot_w64(ownA), mOwnO = __builtin_amdgcn_ballot_w64(ownB);
    const int slot = max(0, gA >> 1);
    unsigned KA = inA ? skey[gA] : 0u, KB = inB ? skey[gB] : 0u, IA = (unsigned)gA, IB = (unsigned)gB;
    int blkno = 0;
    bool quiet_exit = false;
    for (int c = 0; c * 256 < lam && !quiet_exit; ++c) {
        // draws of this chunk: byte [group][pair slot], bit q = phase q of the group compares by objective
        const int ng = min(64, (lam - 256 * c + 3) / 4);
        for (int e = tid; e < ng * pitch; e += NT) {
            const int g = e / pitch, m = e % pitch;
            unsigned bits = 0;
            if (m < nslots) {
                unsigned c4[4] = {(unsigned)m, (unsigned)(256 * c + 4 * g), (unsigned)(a.gen * 16 + 1), (unsigned)run};
                philox(c4, (unsigned)a.seed, (unsigned)(a.seed >> 32));
#pragma unroll
                for (int q = 0; q < 4; ++q) bits |= (((double)c4[q] + 0.5) * (1.0 / 4294967296.0) < 0.45) ? (1u << q) : 0u;
            }
            sdraw[e] = (unsigned char)bits;
        }
        __syncthreads();
        for (int b = 0; b < 16; ++b, ++blkno) {
            const int phb = 256 * c + 16 * b;
            if (phb >= lam) break;
            const int par = blkno & 1;
            if (ranks) {
                u64 dbits = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (4 * b + g < ng) dbits |= (u64)sdraw[(size_t)(4 * b + g) * pitch + slot] << (8 * g);
                u64 moved = 0;
                if (phb + 16 <= lam)
                    rw_block<false, 4>(KA, KB, IA, IB, dbits, 16, mPairE, mPairO, mOwnE, mOwnO, moved);
                else
                    rw_block<true, 4>(KA, KB, IA, IB, dbits, lam - phb, mPairE, mPairO, mOwnE, mOwnO, moved);
                if (ownA) {
                    xk[par * lp + gA] = KA;
                    xi[par * lp + gA] = IA;
                }
                if (ownB) {
                    xk[par * lp + gB] = KB;
                    xi[par * lp + gB] = IB;
                }
                if (moved != 0 && lane == 0) s_moved[blkno % 3] = 1;
            }
            __syncthreads();
            const int any = s_moved[blkno % 3];
            if (tid == 0) s_moved[(blkno + 2) % 3] = 0;  // (its last readers passed this barrier's predecessor; its next writers come after this one)
            if (!any) {  // sixteen phases without a swap: ranked under the drawn rules
                quiet_exit = true;
                break;
            }
            if (inA && !ownA) {
                KA = xk[par * lp + gA];
                IA = xi[par * lp + gA];
            }
            if (inB && !ownB) {
                KB = xk[par * lp + gB];
                IB = xi[par * lp + gB];
            }
        }
    }
    if (ownA) sidx[gA] = (int)IA;
    if (ownB) sidx[gB] = (int)IB;
    __syncthreads();
    sidx_out = sidx;
    return true;
}

__global__ __launch_bounds__(RANK_THREADS) void ps_rank_kernel(Args a, int mode, RankWs ws) {
    extern __shared__ double smem[];
    __shared__ double s_red[2 * (RANK_THREADS / 64)];
    __shared__ int s_ired[2 * (RANK_THREADS / 64)];
    __shared__ int s_swapped, s_stop, s_best, s_infeas;
    const int run = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = (int)blockDim.x;  // 64 .. RANK_THREADS: a small population is ranked by as many waves as it has pairs (a barrier of three waves instead of sixteen)
    const Run &R = a.runs[run];
    if (R.stat[1]) return;
    int *const sy = ws.sync ? ws.sync + run * RS_SYNC : nullptr;
    if (mode == 2 && sy[2] == 0) return;  // (this run's generation was finished in mode 1)
    const int n = R.nvar, lam = R.lam, mu = R.mu, gen = a.gen;
    const double *X = R.X[gen & 1];
    const int evals0 = R.stat[0];
    const int m = min(lam, R.max_evals - evals0);
    if (tid == 0) {
        s_infeas = mode == 2 ? 1 : 0;
        s_stop = 0;
    }
    __syncthreads();
    // ---- (no linear constraints: the run's individuals are scored here, a thread per individual -- ps_score_kernel's arithmetic)
    if (mode != 2 && a.score_fused) {
        for (int i = tid; i < lam; i += NT) {
            double fi, pi;
            score_row<false>(a, R, i, R.off + i, 0, fi, pi);
            R.f[i] = fi;
            R.phi[i] = pi;
        }
        __syncthreads();  // f / phi of the run are read below (and by the launches that follow this one)
    }
    // ---- this generation's best individual and whether any individual inside the budget violates a constraint
    if (mode != 2) {
        Cand mine{0.0, -1, 0x7fffffff};
        bool infeas = false;
        for (int i = tid; i < m; i += NT) {
            const double fi = R.f[i], pi = R.phi[i];
            if (pi > 0.0 && pi < INFINITY) infeas = true;
            if (pi == INFINITY) continue;
            const Cand c{pi == 0.0 ? fi : pi, pi == 0.0 ? 1 : 0, i};
            if (cand_better(c, mine)) mine = c;
        }
        if (infeas) s_infeas = 1;
        for (int off = 32; off > 0; off >>= 1) {
            Cand o;
            o.key = __shfl_xor(mine.key, off);
            o.feas = __shfl_xor(mine.feas, off);
            o.idx = __shfl_xor(mine.idx, off);
            if (cand_better(o, mine)) mine = o;
        }
        if (lane == 0) {
            s_red[wave] = mine.key;
            s_ired[2 * wave] = mine.feas;
            s_ired[2 * wave + 1] = mine.idx;
        }
        __syncthreads();
        if (tid == 0) {
            Cand b{0.0, -1, 0x7fffffff};
            for (int w = 0; w < NT / 64; ++w) {
                const Cand c{s_red[w], s_ired[2 * w], s_ired[2 * w + 1]};
                if (cand_better(c, b)) b = c;
            }
            s_best = -1;
            if (b.feas >= 0) {
                const double bf = R.best[n], bphi = R.best[n + 1];
                if ((b.feas == 1 && (bphi > 0.0 || b.key < bf)) || (b.feas == 0 && b.key < bphi)) {
                    s_best = b.idx;
                    R.best[n] = b.feas == 1 ? b.key : R.f[b.idx];
                    R.best[n + 1] = b.feas == 1 ? 0.0 : b.key;
                }
            }
            R.stat[0] = evals0 + max(m, 0);
            R.stat[2] = gen + 1;
            s_stop = (evals0 + m >= R.max_evals || m < lam) ? 1 : 0;
        }
        __syncthreads();
        if (s_best >= 0)
            for (int c = tid; c < n; c += NT) R.best[c] = X[(size_t)s_best * n + c];
    }
    if (s_stop) {
        if (tid == 0) R.stat[1] = 1;
        return;
    }
    // ---- ranking.  If no individual inside the budget violates a constraint (always so for unconstrained ideal-point runs,
    // usually so late in a PS run) every comparison of the pairwise rule is decided by (violation, objective) whatever is drawn,
    // and lam phases of the stable transposition sort end in THE sorted order (ties by index): a bitonic network over the padded
    // array reaches the same order in log2(N) (log2(N) + 1) / 2 phases (91 instead of 5160 at lam = 5160).
    const bool plain_sort = s_infeas == 0 && !(a.dbg & 4);
    if (mode == 1) {
        // Populations of RS_MINLAM and more leave this workgroup here, either way:
        //   sy[2] = 1  violations inside the budget: the transposition phases on several compute units (ps_rank_wave_kernel)
        //   sy[2] = 2  none: the sorted order by COUNTING on the whole chip -- rank(i) = #{j : f_j < f_i, or f_j = f_i and j < i} is the
        //              position of i in the stable sort by f, lam^2 independent comparisons (ps_rank_prep_kernel, a few us) instead of
        //              a bitonic network in this one workgroup (34 / 61 / 71 us at lam = 1320 / 2600 / 5160, half of a generation of
        //              an ideal-point run at d = 128); the finishing launch (mode 2) turns the counts into the parents' list.
        // (a NaN has no rank, so a generation that holds one stays here)
        bool nan = false;
        if (lam >= RS_MINLAM)
            for (int i = tid; i < lam; i += NT) nan = nan || R.f[i] != R.f[i] || R.phi[i] != R.phi[i];
        const bool leave = !(a.dbg & 1) && lam >= RS_MINLAM && !__syncthreads_or(nan ? 1 : 0);
        const int kind = !leave ? 0 : (!plain_sort ? 1 : ((ws.count_plain && !(a.dbg & (128 | 256 | 512))) ? 2 : 0));
        for (int i = tid; i < RS_SYNC; i += NT) sy[i] = i == 2 ? kind : 0;
        if (kind) {
            int *cnt = ws.cnt + (size_t)run * 2 * MAXLAM;
            for (int i = tid; i < lam; i += NT) cnt[i] = cnt[MAXLAM + i] = 0;
            return;
        }
    }
    int *sidx;
    if (mode == 2 && sy[2] == 2) {  // the positions found by counting (ps_rank_prep_kernel)
        sidx = (int *)smem;
        const int *cnt = ws.cnt + (size_t)run * 2 * MAXLAM;
        for (int i = tid; i < lam; i += NT) sidx[cnt[i]] = i;
        __syncthreads();
    } else if (mode == 2 && sy[1] == 0) {  // the order found by ps_rank_wave_kernel / ps_rank_sort_kernel
        sidx = (int *)smem;
        const int *src = ws.idx[sy[3]] + (size_t)run * MAXLAM;
        for (int i = tid; i < lam; i += NT) sidx[i] = src[i];
        __syncthreads();
    } else if (a.dbg & 1) {
        sidx = (int *)smem;
        for (int i = tid; i < lam; i += NT) sidx[i] = i;
        __syncthreads();
    } else if (plain_sort && !(a.dbg & (128 | 256)) && NT >= 256 && mu * 4 <= lam && (npow2(lam) >= 8 * NT || (a.dbg & 512)) &&
               select_parents(R.f, lam, mu, smem, (int *)(smem + npow2(lam)))) {  // (measured: pays from lam > 4096 on -- 135 against 195 us per ranking at
                                                                                  // lam = 5160; below, its sampling pass costs what the shorter sort saves; dbg 512 forces it)
        sidx = (int *)(smem + npow2(lam));  // (parents found and ranked without sorting the rest: see select_parents)
    } else if (plain_sort && !(a.dbg & 128) && (npow2(lam) == NT || npow2(lam) == 2 * NT || npow2(lam) == 4 * NT || npow2(lam) == 8 * NT)) {
        // (the launch's thread count follows the LARGEST population: N / 2 up to N = 2048, 1024 beyond; a run whose padded size is not
        // 2, 4 or 8 elements per thread -- a much smaller run in the same launch -- takes the one-pair-per-thread form below)
        const int N = npow2(lam);
        double *sf = smem;      // N keys: violation is 0 or inf here, and inf comes with f = inf
        sidx = (int *)(sf + N);
        if (N == NT)
            bitonic_regs<1>(R.f, nullptr, lam, N, sf, sidx);
        else if (N == 2 * NT)
            bitonic_regs<2>(R.f, nullptr, lam, N, sf, sidx);
        else if (N == 4 * NT)
            bitonic_regs<4>(R.f, nullptr, lam, N, sf, sidx);
        else
            bitonic_regs<8>(R.f, nullptr, lam, N, sf, sidx);
    } else if (plain_sort) {
        int N = 1;
        while (N < lam) N <<= 1;
        double *sf = smem;      // N keys: violation is 0 or inf here, and inf comes with f = inf
        sidx = (int *)(sf + N);
        for (int i = tid; i < N; i += NT) {
            sf[i] = i < lam ? R.f[i] : INFINITY;
            sidx[i] = i < lam ? i : 0x7fffffff;
        }
        __syncthreads();
        for (int kk = 2; kk <= N; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < N / 2; t += NT) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;  // the t-th pair of this stage
                    const double fa = sf[i], fb = sf[l];
                    const int ia = sidx[i], ib = sidx[l];
                    const bool greater = fa > fb || (fa == fb && ia > ib);
                    const bool up = (i & kk) == 0;
                    if (greater == up) {
                        sf[i] = fb;
                        sf[l] = fa;
                        sidx[i] = ib;
                        sidx[l] = ia;
                    }
                }
                __syncthreads();
            }
    } else {
        // lam phases of odd-even transposition; a pair is compared by f when both are feasible or with probability 0.45, else by
        // the constraint violation (Runarsson & Yao).  The records themselves are swapped; one Philox call feeds four phases of a
        // pair; one barrier per phase.
        if (lam < RS_MINLAM && !(a.dbg & 2048) && NT >= 64 * rws_waves(lam) && rank_small_waves(a, R, run, smem, sidx)) {
            // (the order is in sidx: the records stayed in registers -- see rank_small_waves)
        } else {
        double *sf = smem, *sphi = sf + lam;
        sidx = (int *)(sphi + lam);
        for (int i = tid; i < lam; i += NT) {
            sf[i] = R.f[i];
            sphi[i] = R.phi[i];
            sidx[i] = i;
        }
        if (tid == 0) s_swapped = 1;
        __syncthreads();
        constexpr int PSLOTS = (MAXLAM / 2 + RANK_THREADS - 1) / RANK_THREADS;  // pairs per thread and phase
        // ONE exit rule for a population size, whichever kernel ranks it: with stochastic comparisons a stretch of phases without a
        // swap is not a fixed point (a pair that disagrees on f and phi can still swap on a later draw), so WHERE the no-swap test
        // sits decides the order that comes out.  Populations of RS_MINLAM and more -- the ones ps_rank_sort_kernel takes, which can
        // only test at its chunk boundaries -- are tested every RS_B phases here too (this code is also that kernel's fallback after
        // a time-out and the MRBF_PS_MULTI=0 path); smaller ones every sixteen.
        const int quiet = lam >= RS_MINLAM ? RS_B : 16;
        for (int ph0 = 0; ph0 < lam; ph0 += 4) {
            if (ph0 % quiet == 0) {
                const int sw = s_swapped;
                __syncthreads();
                if (!sw) break;  // `quiet` phases without a swap: ranked under the drawn rules
                if (tid == 0) s_swapped = 0;
                __syncthreads();
            }
            unsigned c4[PSLOTS][4];
            bool drawn[PSLOTS];
#pragma unroll
            for (int s = 0; s < PSLOTS; ++s) drawn[s] = false;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int ph = ph0 + q4;
                if (ph < lam) {
#pragma unroll
                    for (int s = 0; s < PSLOTS; ++s) {
                        const int p = tid + s * NT;
                        const int j = 2 * p + (ph & 1);
                        if (j + 1 < lam) {
                            const double fa = sf[j], fb = sf[j + 1], pa = sphi[j], pb = sphi[j + 1];
                            const int ia = sidx[j], ib = sidx[j + 1];  // (with the keys: a swap then costs no second LDS round trip)
                            bool by_f = pa == 0.0 && pb == 0.0;
                            if (!by_f) {
                                if (!drawn[s]) {
                                    c4[s][0] = (unsigned)p;
                                    c4[s][1] = (unsigned)ph0;
                                    c4[s][2] = (unsigned)(gen * 16 + 1);
                                    c4[s][3] = (unsigned)run;
                                    philox(c4[s], (unsigned)a.seed, (unsigned)(a.seed >> 32));
                                    drawn[s] = true;
                                }
                                by_f = ((double)c4[s][q4] + 0.5) * (1.0 / 4294967296.0) < 0.45;
                            }
                            const bool worse = by_f ? (fa > fb) : (pa > pb);
                            if (worse) {
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
                }
                __syncthreads();
            }
        }
        __syncthreads();
        }
    }
    // the parents, in rank order
    for (int i = tid; i < mu; i += NT) R.order[i] = sidx[i];
    // ---- stop like NLopt's xtol_rel, on the survivors' spread: variable by variable, leaving at the first one that still spreads
    bool converged = true;
    for (int c = 0; c < n; ++c) {
        double lo = INFINITY, hi = -INFINITY;
        for (int s2 = tid; s2 < mu; s2 += NT) {
            const double v = X[(size_t)sidx[s2] * n + c];
            lo = fmin(lo, v);
            hi = fmax(hi, v);
        }
        for (int off = 32; off > 0; off >>= 1) {
            lo = fmin(lo, __shfl_xor(lo, off));
            hi = fmax(hi, __shfl_xor(hi, off));
        }
        __syncthreads();  // s_red of the previous variable has been read
        if (lane == 0) {
            s_red[2 * wave] = lo;
            s_red[2 * wave + 1] = hi;
        }
        __syncthreads();
        for (int w = 0; w < NT / 64; ++w) {
            lo = fmin(lo, s_red[2 * w]);
            hi = fmax(hi, s_red[2 * w + 1]);
        }
        if (hi - lo > a.xtol_rel * fmax(fabs(X[(size_t)sidx[0] * n + c]), 1e-300)) {
            converged = false;
            break;  // uniform: every thread holds the same lo / hi
        }
    }
    if (tid == 0) {
        if (converged)
            R.stat[1] = 1;
        else
            R.stat[3] = gen + 1;  // breed the next generation
    }
}

// ---- the transposition phases of one run on RS_W workgroups.  A phase moves information by one position, so a workgroup that loads
// its part of the array plus RS_B positions on either side can run RS_B phases on its own and its part comes out exactly as if the
// whole array had been worked on (the halo positions go wrong from the window's edges inwards, one position per phase, and are thrown
// away); every comparison is a function of (pair, phase, values) -- the Philox counter is the pair and the four-phase group, as in
// ps_rank_kernel -- so the redundant comparisons in the halos agree with the owner's.  Between chunks the parts are exchanged through
// global memory (write-through stores, coherent loads: the workgroups sit behind different L2s) and the workgroups of a run meet at a
// counter.  A chunk in which no workgroup moved anything inside its own part ends the ranking (ps_rank_kernel applies the same test
// at the same phases for these population sizes: one exit rule, hence one order, whichever kernel runs).  The counter wait is bounded: on a time-out the failure word is set and ps_rank_kernel (mode 2) runs the phases
// itself -- this kernel never writes f / phi.
__device__ __forceinline__ unsigned long long rs_clock() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz
__global__ __launch_bounds__(RS_THREADS) void ps_rank_sort_kernel(Args a, RankWs ws) {
    __shared__ double sf[RS_MAXWIN], sphi[RS_MAXWIN];
    __shared__ int sidx[RS_MAXWIN];
    __shared__ int s_sw, s_ok, s_any;
    const int run = blockIdx.y, w = blockIdx.x, tid = threadIdx.x;
    const Run &R = a.runs[run];
    int *const sy = ws.sync + run * RS_SYNC;
    if (R.stat[1] || sy[2] != 1) return;
    if (a.dbg & 64) {  // (test switch: give up at once, as after a counter time-out)
        if (threadIdx.x == 0) __hip_atomic_store(sy + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int lam = R.lam, gen = a.gen;
    const int per = ((lam + RS_W - 1) / RS_W + 1) & ~1;  // even: windows start at even positions (a thread keeps its pair slot over both parities)
    const int s0 = min(lam, w * per), e0 = min(lam, s0 + per);
    const int ws0 = max(0, s0 - RS_B), we0 = min(lam, e0 + RS_B), wn = we0 - ws0;
    const size_t ro = (size_t)run * MAXLAM;
    const int nch = (lam + RS_B - 1) / RS_B;
    for (int c = 0; c < nch; ++c) {
        const int sb = c & 1, db = sb ^ 1;
        for (int i = tid; i < wn; i += RS_THREADS) {
            const int g = ws0 + i;
            if (c == 0) {
                sf[i] = R.f[g];
                sphi[i] = R.phi[g];
                sidx[i] = g;
            } else {
                sf[i] = __hip_atomic_load(ws.f[sb] + ro + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sphi[i] = __hip_atomic_load(ws.phi[sb] + ro + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sidx[i] = __hip_atomic_load(ws.idx[sb] + ro + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) s_sw = 0;
        __syncthreads();
        const int ph_end = min(lam, (c + 1) * RS_B);
        for (int ph0 = c * RS_B; ph0 < ph_end; ph0 += 4) {
            unsigned c4[4];
            bool drawn = false;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int ph = ph0 + q4;
                if (ph < lam) {
                    const int j = ws0 + (ph & 1) + 2 * tid;  // (ws0 even)
                    if (j + 1 < we0 && s0 < e0) {
                        const int li = j - ws0;
                        const double fa = sf[li], fb = sf[li + 1], pa = sphi[li], pb = sphi[li + 1];
                        const int ia = sidx[li], ib = sidx[li + 1];  // (with the keys: a swap then costs no second LDS round trip)
                        bool by_f = pa == 0.0 && pb == 0.0;
                        if (!by_f) {
                            if (!drawn) {
                                c4[0] = (unsigned)(j >> 1);
                                c4[1] = (unsigned)ph0;
                                c4[2] = (unsigned)(gen * 16 + 1);
                                c4[3] = (unsigned)run;
                                philox(c4, (unsigned)a.seed, (unsigned)(a.seed >> 32));
                                drawn = true;
                            }
                            by_f = ((double)c4[q4] + 0.5) * (1.0 / 4294967296.0) < 0.45;
                        }
                        const bool worse = by_f ? (fa > fb) : (pa > pb);
                        if (worse) {
                            sf[li] = fb;
                            sf[li + 1] = fa;
                            sphi[li] = pb;
                            sphi[li + 1] = pa;
                            sidx[li] = ib;
                            sidx[li + 1] = ia;
                            if (j >= s0 && j < e0) s_sw = 1;
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < e0 - s0; i += RS_THREADS) {
            const int li = s0 - ws0 + i;
            __hip_atomic_store(ws.f[db] + ro + s0 + i, sf[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ws.phi[db] + ro + s0 + i, sphi[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ws.idx[db] + ro + s0 + i, sidx[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left the CU
        __syncthreads();
        if (tid == 0) {
            if (s_sw) __hip_atomic_fetch_or(sy + 4 + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // arrival = RELEASE (orders the part's stores and the "moved" flag before the count, by the memory model and not only by
            // the s_waitcnt above), the wait ends with an ACQUIRE load of the same counter before s_any and the next chunk's loads
            __hip_atomic_fetch_add(sy, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const int target = RS_W * (c + 1);
            const unsigned long long t0 = rs_clock();
            bool ok = true;
            while (__hip_atomic_load(sy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (__hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || rs_clock() - t0 > 500000ull) {  // 5 ms
                    __hip_atomic_store(sy + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ws.sync + RS_SYNC * MAXRUNS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (sticky: the host stops using this kernel)
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            s_ok = ok ? 1 : 0;
            s_any = __hip_atomic_load(sy + 4 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_ok) return;
        if (w == 0 && tid == 0) sy[3] = db;  // (the result so far; read by the next kernel in the stream)
        if (!s_any) return;                  // nothing moved in RS_B phases: ranked under the drawn rules
    }
}

// ---- the same phases with one WAVE per part and the records in registers (round 6; ps_rank_sort_kernel above stays as the A/B form,
// MRBF_PS_MULTI=2).  The workgroup form pays an LDS round trip and a barrier of eight waves per phase (0.32 us: 425 us per ranking at
// lam = 1320) and works on a window seven times its own part.  Here a wave holds a window of 128 records, two per lane (its own
// RW_OWN = 64 in the middle, RW_H = 32 of either neighbour on each side), so an even phase is lane-local, an odd phase takes the
// neighbouring lane's record by DPP (wave_shl / wave_shr), and no phase needs a barrier; after RW_H phases the waves publish their parts
// (write-through stores of self-describing records: see the kernel) and take the neighbours' halves -- a wave waits for its two
// neighbours only.  Every RS_B phases all waves of the run meet at the arrival counter for the one exit rule.
// A wave alone on its SIMD issues one instruction every four cycles whatever its kind, so the instruction count of a phase IS its
// time (measured 0.11 us with records of two doubles + index: 22 vector + 14 scalar instructions).  Two things are therefore taken
// out of the phases and done beforehand on the whole chip by ps_rank_prep_kernel, since neither depends on the order so far:
//   * the draws (Philox is 40 quarter-rate multiplies per four phases): one byte per (four-phase group, pair slot), bit q = "phase
//     q of the group compares by objective";
//   * the keys: the pairwise rule only ever asks fa > fb, pa > pb and pa == pb == 0, and rank(x) = #{j : x_j < x} answers the first
//     two exactly (ties stay ties), so a record is TWO words -- (rank of f) << 16 | (0 if phi == 0, else 1 + rank of phi), and the
//     index -- instead of five, compared by 16-bit integer compares.  (NaN has no rank: ps_rank_kernel does not hand such a
//     generation over.)
// grid.x = [draw workgroups | counting workgroups]; the counts (ws.cnt: lam objective counts, then lam violation counts per run) are
// zeroed by ps_rank_kernel (mode 1) when it hands the generation over
__global__ __launch_bounds__(256) void ps_rank_prep_kernel(Args a, RankWs ws, u64 *draws, size_t per_run, int draw_blocks) {
    __shared__ double sf[RW_CNT_T], sp[RW_CNT_T];
    const int run = blockIdx.y;
    const Run &R = a.runs[run];
    const int *sy = ws.sync + run * RS_SYNC;
    const int kind = sy[2];  // 1: draws + keys for the transposition phases; 2: the sorted order of a generation without violations
    if (R.stat[1] || kind == 0) return;
    const int lam = R.lam;
    if ((int)blockIdx.x < draw_blocks) {
        if (kind != 1) return;
        const int pitch = rw_pitch(lam), nblk = (lam + RW_H - 1) / RW_H, npair = lam / 2;
        u64 *out = draws + (size_t)run * per_run;  // [block of RW_H phases][pair slot]: byte g = the block's g-th group of four phases
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)nblk * pitch; e += (int64_t)draw_blocks * 256) {
            const int blk = (int)(e / pitch), m = (int)(e % pitch);
            u64 bits = 0;
            if (m < npair) {
#pragma unroll
                for (int g = 0; g < RW_H / 4; ++g) {
                    unsigned c4[4] = {(unsigned)m, (unsigned)(blk * RW_H + 4 * g), (unsigned)(a.gen * 16 + 1), (unsigned)run};
                    philox(c4, (unsigned)a.seed, (unsigned)(a.seed >> 32));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bits |= (((double)c4[q] + 0.5) * (1.0 / 4294967296.0) < 0.45) ? (1ull << (8 * g + q)) : 0ull;
                }
            }
            out[e] = bits;
        }
        return;
    }
    // counting: workgroup (ib, js) counts, for its 256 individuals i, the j of its share with f_j < f_i and with phi_j < phi_i
    const int cb = (int)blockIdx.x - draw_blocks, nib = (lam + RW_CNT_T - 1) / RW_CNT_T, js = cb / nib, ib = cb % nib, nsplit = rw_jsplit(lam);
    if (js >= nsplit) return;
    const int i = ib * RW_CNT_T + threadIdx.x;
    const double fi = i < lam ? R.f[i] : 0.0, pi = i < lam ? R.phi[i] : 0.0;
    const int per = (lam + nsplit - 1) / nsplit, j0 = js * per, j1 = min(lam, j0 + per);
    int cf = 0, cp = 0;
    for (int jb = j0; jb < j1; jb += RW_CNT_T) {
        const int nj = min(RW_CNT_T, j1 - jb);
        __syncthreads();
        if ((int)threadIdx.x < nj) {
            sf[threadIdx.x] = R.f[jb + threadIdx.x];
            sp[threadIdx.x] = R.phi[jb + threadIdx.x];
        }
        __syncthreads();
        if (kind == 1) {
            for (int j = 0; j < nj; ++j) {
                cf += sf[j] < fi ? 1 : 0;
                cp += sp[j] < pi ? 1 : 0;
            }
        } else {  // position in the stable sort by f: ties by index
            for (int j = 0; j < nj; ++j) cf += (sf[j] < fi || (sf[j] == fi && jb + j < i)) ? 1 : 0;
        }
    }
    if (i < lam) {
        int *cnt = ws.cnt + (size_t)run * 2 * MAXLAM;
        atomicAdd(cnt + i, cf);
        atomicAdd(cnt + MAXLAM + i, cp);
    }
}

__global__ __launch_bounds__(64) void ps_rank_wave_kernel(Args a, RankWs ws, const u64 *draws, size_t per_run, unsigned epoch) {
    const int run = blockIdx.y, w = blockIdx.x, lane = threadIdx.x;
    const Run &R = a.runs[run];
    int *const sy = ws.sync + run * RS_SYNC;
    if (R.stat[1] || sy[2] != 1) return;
    const int lam = R.lam, nw = rw_waves(lam);
    if (w >= nw) return;
    if (a.dbg & 64) {  // (test switch: give up at once, as after a counter time-out)
        if (lane == 0) __hip_atomic_store(sy + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const size_t ro = (size_t)run * MAXLAM;
    // exchange buffers (the words of ws.f): a record travels as ONE 64-bit word -- key << 32 | tag << 16 | individual, tag = (launch, block)
    // -- so the word itself says when it has arrived: the reader polls the records it needs, there is no flag to write after the data
    // and no wait for the stores to be acknowledged (one trip through memory per exchange instead of three).  A buffer is rewritten
    // every second block and neighbours are never more than one block apart (a wave publishes block b + 1 only after it has taken
    // its neighbours' block b), so a reader cannot miss a record; records of earlier launches carry another launch number.
    u64 *const xr[2] = {reinterpret_cast<u64 *>(ws.f[0]) + ro, reinterpret_cast<u64 *>(ws.f[1]) + ro};
    const int s0 = w * RW_OWN, e0 = min(lam, s0 + RW_OWN), ws0 = s0 - RW_H;  // (ws0 even; negative for the first wave)
    const int gA = ws0 + 2 * lane, gB = gA + 1;                              // this lane's two positions
    const bool inA = gA >= 0 && gA < lam, inB = gB >= 0 && gB < lam;
    const bool ownA = gA >= s0 && gA < e0, ownB = gB >= s0 && gB < e0;
    const bool needA = inA && !ownA, needB = inB && !ownB;
    // lane masks (scalar registers): the decisions of a phase are mask arithmetic on the scalar unit, the vector unit compares and selects
    const u64 mPairE = __builtin_amdgcn_ballot_w64(inA && inB);                          // even phases: (gA, gB), lane-local
    const u64 mPairO = __builtin_amdgcn_ballot_w64(inB && gB + 1 < lam && lane < 63);    // odd phases: (gB, next lane's gA)
    const u64 mOwnE = __builtin_amdgcn_ballot_w64(ownA), mOwnO = __builtin_amdgcn_ballot_w64(ownB);  // the pair's left position lies in this wave's part
    const int slot = max(0, gA >> 1);  // pair slot of both of this lane's pairs
    const int pitch = rw_pitch(lam);
    const u64 *dr = draws + (size_t)run * per_run + slot;
    const int *cnt = ws.cnt + (size_t)run * 2 * MAXLAM;
    unsigned KA = 0, KB = 0, IA = (unsigned)gA, IB = (unsigned)gB;
    if (inA) KA = ((unsigned)cnt[gA] << 16) | (R.phi[gA] == 0.0 ? 0u : 1u + (unsigned)cnt[MAXLAM + gA]);
    if (inB) KB = ((unsigned)cnt[gB] << 16) | (R.phi[gB] == 0.0 ? 0u : 1u + (unsigned)cnt[MAXLAM + gB]);
    const int nblk = (lam + RW_H - 1) / RW_H;
    u64 moved = 0;
    u64 dnext = dr[0];
    for (int blk = 0; blk < nblk; ++blk) {
        const int phb = blk * RW_H;
        const u64 dbits = dnext;
        if (blk + 1 < nblk) dnext = dr[(size_t)(blk + 1) * pitch];  // (next block's draws: in flight over this block)
        if (phb + RW_H <= lam)
            rw_block<false, RW_H / 4>(KA, KB, IA, IB, dbits, RW_H, mPairE, mPairO, mOwnE, mOwnO, moved);
        else
            rw_block<true, RW_H / 4>(KA, KB, IA, IB, dbits, lam - phb, mPairE, mPairO, mOwnE, mOwnO, moved);
        // ---- publish this wave's part
        const int db = blk & 1;
        const unsigned tag = ((epoch & 0xffu) << 8) | (unsigned)(blk + 1);
        if (ownA) __hip_atomic_store(xr[db] + gA, ((u64)KA << 32) | (tag << 16) | IA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ownB) __hip_atomic_store(xr[db] + gB, ((u64)KB << 32) | (tag << 16) | IB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool chunk_end = ((blk + 1) * RW_H) % RS_B == 0 || blk + 1 == nblk;
        const int c = (blk * RW_H) / RS_B;
        const unsigned long long t0 = rs_clock();
        int ok = 1, any = 1;
        if (chunk_end) {  // ---- every RS_B phases: all waves of the run meet for the exit rule
            if (lane == 0) {
                if (moved != 0) __hip_atomic_fetch_or(sy + 4 + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(sy, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // (orders the "moved" flag before the count)
                int spins = 0;
                while (__hip_atomic_load(sy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nw * (c + 1)) {
                    if ((++spins & 15) == 0 && (__hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || rs_clock() - t0 > 500000ull)) {  // 5 ms
                        ok = 0;
                        break;
                    }
                }
                if (ok) any = __hip_atomic_load(sy + 4 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            any = __builtin_amdgcn_readfirstlane(any);
            moved = 0;
            if (ok && (!any || blk + 1 == nblk)) {  // nothing moved in RS_B phases (ranked under the drawn rules), or all phases done
                int *out = ws.idx[0] + ro;          // the order, for the finishing launch
                if (ownA) out[gA] = (int)IA;
                if (ownB) out[gB] = (int)IB;
                if (w == 0 && lane == 0) sy[3] = 0;
                return;
            }
        }
        // ---- take the neighbours' halves: poll the records until they carry this block's tag
        if (ok && !(a.dbg & 1024)) {  // (1024, a timing experiment: no wait for the neighbours -- the result is not a ranking)
            int spins = 0;
            for (;;) {
                const u64 ra = needA ? __hip_atomic_load(xr[db] + gA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                const u64 rb = needB ? __hip_atomic_load(xr[db] + gB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                const bool okA = !needA || (((unsigned)ra >> 16) == tag), okB = !needB || (((unsigned)rb >> 16) == tag);
                if (okA && needA) {
                    KA = (unsigned)(ra >> 32);
                    IA = (unsigned)ra & 0xffffu;
                }
                if (okB && needB) {
                    KB = (unsigned)(rb >> 32);
                    IB = (unsigned)rb & 0xffffu;
                }
                if (__builtin_amdgcn_ballot_w64(!(okA && okB)) == 0) break;
                if ((++spins & 15) == 0 && (__hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || rs_clock() - t0 > 500000ull)) {  // 5 ms
                    ok = 0;
                    break;
                }
            }
        }
        if (!ok) {
            if (lane == 0) {
                __hip_atomic_store(sy + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ws.sync + RS_SYNC * MAXRUNS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (sticky: the host stops using this kernel)
            }
            return;
        }
    }
}

// ---- next generation: one wave per offspring, lanes over the components
__global__ __launch_bounds__(256) void ps_breed_kernel(Args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int run = run_of_row(a, row);
    const Run &R = a.runs[run];
    const int gen = a.gen;
    if (R.stat[3] != gen + 1 || (a.dbg & 2)) return;
    const int n = R.nvar, mu = R.mu, o = row - R.off;
    const double *X = R.X[gen & 1], *S = R.S[gen & 1];
    double *Xn = R.X[(gen + 1) & 1], *Sn = R.S[(gen + 1) & 1];
    const double tau = 1.0 / sqrt(2.0 * sqrt((double)n)), taup = 1.0 / sqrt(2.0 * (double)n), alpha = 0.2, gamma = 0.85;
    const int nd = mu - 1;
    const int par = R.order[o % mu];
    const double *xp = X + (size_t)par * n, *sp = S + (size_t)par * n;
    double *xo = Xn + (size_t)o * n, *so = Sn + (size_t)o * n;
    const int skip = R.kind == 1 ? 1 : 0;
    double *xe = a.Xeval + (size_t)row * a.d;  // next generation's row of the evaluation batch (x part)
    __shared__ double s_row[4][360];           // the wave's new x part (a.Xq: centred below with the centring kernel's lane mapping)
    double *const srow = s_row[threadIdx.x >> 6];
    const bool keep = a.Xq != nullptr;
    if (o < nd) {
        // differential variation towards the best individual; kept only if it stays inside the box
        const double *xb = X + (size_t)R.order[0] * n, *xq = X + (size_t)R.order[o + 1] * n;
        bool inside = true;
        for (int c = lane; c < n; c += 64) {
            const double v = xp[c] + gamma * (xb[c] - xq[c]);
            inside = inside && v >= lo_of(a, R, c) && v <= hi_of(a, R, c);
        }
        inside = __all(inside);
        for (int c = lane; c < n; c += 64) {
            const double v = inside ? xp[c] + gamma * (xb[c] - xq[c]) : xp[c];
            xo[c] = v;
            so[c] = sp[c];
            if (c >= skip) {
                xe[c - skip] = v;
                if (keep) srow[c - skip] = v;
            }
        }
    } else {
        double u0, u1, zg, z1;
        rng4(a, run, gen, o, n, 2, u0, u1, zg, z1);  // the individual's global factor (the same draw in every lane)
        for (int c = lane; c < n; c += 64) {
            const double lo = lo_of(a, R, c), hi = hi_of(a, R, c);
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
            if (c >= skip) {
                xe[c - skip] = v;
                if (keep) srow[c - skip] = v;
            }
            so[c] = sp[c] + alpha * (s - sp[c]);  // exponential smoothing
        }
    }
    if (keep) {  // Xq[row][t] = x[t] - mean[t] (zero in the padding), xsq[row] = |Xq[row]|^2: center_pad_kernel's arithmetic, lane for lane
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (the wave's own LDS writes above are read by other lanes below)
        double ss = 0.0;
        for (int t = lane; t < a.xqD; t += 64) {
            double v = 0.0;
            if (t < a.d) v = srow[t] - a.xmean[t];
            a.Xq[(size_t)row * a.xqD + t] = v;
            ss = fma(v, v, ss);
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        if (lane == 0) a.xsq[row] = ss;
    }
}

// host view of the problem (function table, constraints) shared by the driver and the polish
struct Problem {
    int nmodels = 0, nobj = 0, nftot = 0, d = 0;
    const mrbf_model *models[MAXMODELS] = {};
    int foff[MAXMODELS] = {};  // first column of model j in the concatenated function vector
    int obj_model[MAXOBJ] = {}, obj_col[MAXOBJ] = {};
    int ncon = 0, con_model[MAXCON] = {}, con_col[MAXCON] = {}, con_eq[MAXCON] = {};
    int nlin_eq = 0, nlin_ineq = 0;
    std::vector<double> A_eq, b_eq, A_ineq, b_ineq;  // host copies (polish, final feasibility)
    const double *dA_eq = nullptr, *db_eq = nullptr, *dA_ineq = nullptr, *db_ineq = nullptr;
    double eq_tol = 1e-8;
    double objective(const std::vector<double> &allF, int l) const { return allF[foff[obj_model[l]] + obj_col[l]]; }
    // sum of squared violations of the modelled and linear constraints at x (allF = all model outputs at x)
    double violation(const std::vector<double> &allF, const double *x) const {
        double phi = 0.0;
        for (int c = 0; c < ncon; ++c) {
            const double v = allF[foff[con_model[c]] + con_col[c]];
            if (con_eq[c] ? std::fabs(v) > eq_tol : v > 0.0) phi += v * v;
            if (!(v == v)) phi = INFINITY;
        }
        for (int c = 0; c < nlin_eq + nlin_ineq; ++c) {
            const bool eq = c < nlin_eq;
            const double *Ar = eq ? &A_eq[(size_t)c * d] : &A_ineq[(size_t)(c - nlin_eq) * d];
            double s = 0.0;
            for (int j = 0; j < d; ++j) s = std::fma(Ar[j], x[j], s);
            s -= eq ? b_eq[c] : b_ineq[c - nlin_eq];
            if (eq ? std::fabs(s) > eq_tol : s > 0.0) phi += s * s;
        }
        return phi;
    }
};

}  // namespace ps

// all model outputs at m points (row-major m x nftot) and, optionally (m == 1), the objectives' Jacobian rows
// work space of the several-compute-unit ranking + the launches of its phases (form 1: one wave per part, ps_rank_wave_kernel, after
// the draws; form 2: the workgroup form of round 5, ps_rank_sort_kernel)
// work space of the several-compute-unit ranking + the launches of its phases (form 1: one wave per part, ps_rank_wave_kernel, after
// the draws and the keys; form 2: the workgroup form of round 5, ps_rank_sort_kernel)
static int rank_ws_setup(mrbf_ctx *ctx, int maxlam, ps::RankWs &rw, unsigned long long **draws, size_t *draws_per_run) {
    using namespace ps;
    double *wsb;
    const size_t per_run = (size_t)MAXLAM * 6 + RS_SYNC / 2;  // f, phi twice, idx twice and the two counts (as doubles: 4 x 1/2), sync words
    *draws_per_run = (size_t)((maxlam + RW_H - 1) / RW_H) * rw_pitch(maxlam);
    MRBF_TRY(get_buf(ctx, S_PS_RANK, per_run * MAXRUNS + 64 + *draws_per_run * MAXRUNS, &wsb));
    rw.f[0] = wsb;
    rw.f[1] = rw.f[0] + (size_t)MAXLAM * MAXRUNS;
    rw.phi[0] = rw.f[1] + (size_t)MAXLAM * MAXRUNS;
    rw.phi[1] = rw.phi[0] + (size_t)MAXLAM * MAXRUNS;
    rw.idx[0] = reinterpret_cast<int *>(rw.phi[1] + (size_t)MAXLAM * MAXRUNS);
    rw.idx[1] = rw.idx[0] + (size_t)MAXLAM * MAXRUNS;
    rw.cnt = rw.idx[1] + (size_t)MAXLAM * MAXRUNS;
    rw.sync = rw.cnt + (size_t)2 * MAXLAM * MAXRUNS;
    *draws = reinterpret_cast<unsigned long long *>(wsb + per_run * MAXRUNS + 64);
    MRBF_HIP(ctx, hipMemsetAsync(rw.sync + RS_SYNC * MAXRUNS, 0, sizeof(int), ctx->stream));
    return MRBF_OK;
}
static void rank_phases_launch(mrbf_ctx *ctx, const ps::Args &a, const ps::RankWs &rw, unsigned long long *draws, size_t draws_per_run, int maxlam, int form,
                               bool phases_possible) {
    using namespace ps;
    if (form == 2) {
        hipLaunchKernelGGL(ps_rank_sort_kernel, dim3(RS_W, (unsigned)a.nruns), dim3(RS_THREADS), 0, ctx->stream, a, rw);
        return;
    }
    const int draw_blocks = (int)std::min<size_t>(512, (draws_per_run + 255) / 256);
    const int count_blocks = ((maxlam + RW_CNT_T - 1) / RW_CNT_T) * rw_jsplit(maxlam);
    hipLaunchKernelGGL(ps_rank_prep_kernel, dim3((unsigned)(draw_blocks + count_blocks), (unsigned)a.nruns), dim3(256), 0, ctx->stream, a, rw, draws, draws_per_run, draw_blocks);
    if (phases_possible)  // (runs without modelled or linear constraints -- ideal-point runs of a box-constrained problem -- never violate anything)
        hipLaunchKernelGGL(ps_rank_wave_kernel, dim3((unsigned)rw_waves(maxlam), (unsigned)a.nruns), dim3(64), 0, ctx->stream, a, rw, draws, draws_per_run, ++ctx->ps_rank_epoch);
}

static int ps_eval_points(mrbf_ctx *ctx, const ps::Problem &P, const double *x_host, int m, std::vector<double> &allF, std::vector<double> *Jobj) {
    const int d = P.d;
    double *dX;
    size_t cnt = (size_t)m * d + (size_t)m * P.nftot;
    for (int j = 0; j < P.nmodels; ++j) cnt += (size_t)(Jobj ? m : 1) * P.models[j]->k * d;
    MRBF_TRY(get_buf(ctx, S_PS_POLISH, cnt, &dX));
    double *dV = dX + (size_t)m * d, *dJ = dV + (size_t)m * P.nftot;
    // The descent phase calls this some 140 times per step with a few KB each way.  Transfers from / to pageable memory are a
    // synchronous round trip of their own (round 6: 285 copies = a sixth of the d = 128 step): the points go up and the values /
    // Jacobians come down through the context's pinned block (free during a PS step: no entry point arms it here), enqueued
    // asynchronously in front of ONE stream synchronisation.
    const size_t up_cnt = (size_t)m * d, down_cnt = cnt - up_cnt;
    double *pin = (ctx->pin_base && !ctx->pin_armed && cnt * sizeof(double) <= ((size_t)1 << 20)) ? reinterpret_cast<double *>(ctx->pin_base) : nullptr;
    // Round 6, second half: no copies at all.  The pinned block is mapped into the device's address space, so the kernels read the
    // points from it and write values / Jacobians into it (a few KB over the link inside kernels that run anyway) -- two blit launches
    // and their gaps less per call, ~75 calls per step (MRBF_PS_ZEROCOPY=0: the staged copies).
    static const int zc_env = mrbf_env("MRBF_PS_ZEROCOPY") ? atoi(mrbf_env("MRBF_PS_ZEROCOPY")) : 1;
    const bool zero_copy = pin && zc_env;
    if (zero_copy) {
        std::memcpy(pin, x_host, up_cnt * sizeof(double));
        dX = pin;
        dV = pin + up_cnt;
        dJ = dV + (size_t)m * P.nftot;
    } else if (pin) {
        std::memcpy(pin, x_host, up_cnt * sizeof(double));
        MRBF_HIP(ctx, hipMemcpyAsync(dX, pin, up_cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    } else {
        MRBF_HIP(ctx, hipMemcpyAsync(dX, x_host, up_cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    std::vector<double *> jp(P.nmodels, nullptr), vp(P.nmodels, nullptr);
    for (int j = 0; j < P.nmodels; ++j) {
        bool need = false;
        for (int l = 0; Jobj && l < P.nobj; ++l) need = need || P.obj_model[l] == j;
        jp[j] = need ? dJ : nullptr;
        vp[j] = dV + (size_t)m * P.foff[j];  // model j's m x k_j block
        MRBF_TRY(eval_model(ctx, P.models[j], m, dX, vp[j], jp[j], nullptr));
        dJ += (size_t)(Jobj ? m : 1) * P.models[j]->k * d;
    }
    std::vector<double> blocks_v;
    std::vector<std::vector<double>> Jm_v(P.nmodels);
    const double *blocks;
    std::vector<const double *> Jm(P.nmodels, nullptr);
    if (pin) {
        double *down = pin + up_cnt;  // the values, then every model's Jacobian block at its device offset: one download
        if (!zero_copy) MRBF_HIP(ctx, hipMemcpyAsync(down, dV, down_cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        blocks = down;
        for (int j = 0; j < P.nmodels; ++j)
            if (jp[j]) Jm[j] = down + (jp[j] - dV);
    } else {
        blocks_v.resize((size_t)m * P.nftot);
        MRBF_HIP(ctx, hipMemcpyAsync(blocks_v.data(), dV, blocks_v.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        blocks = blocks_v.data();
        for (int j = 0; j < P.nmodels; ++j)
            if (jp[j]) {
                Jm_v[j].resize((size_t)m * P.models[j]->k * d);
                MRBF_HIP(ctx, hipMemcpyAsync(Jm_v[j].data(), jp[j], Jm_v[j].size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                Jm[j] = Jm_v[j].data();
            }
    }
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    allF.resize((size_t)m * P.nftot);
    for (int j = 0; j < P.nmodels; ++j) {
        const int kj = P.models[j]->k;
        for (int p = 0; p < m; ++p)
            for (int c = 0; c < kj; ++c) allF[(size_t)p * P.nftot + P.foff[j] + c] = blocks[(size_t)m * P.foff[j] + (size_t)p * kj + c];
    }
    if (Jobj) {  // [point][objective][coordinate]
        Jobj->resize((size_t)m * P.nobj * d);
        for (int p = 0; p < m; ++p)
            for (int l = 0; l < P.nobj; ++l) {
                const int j = P.obj_model[l], kj = P.models[j]->k;
                const double *Jp = Jm[j] + (size_t)p * kj * d;  // per point k x d column-major
                for (int t = 0; t < d; ++t) (*Jobj)[((size_t)p * P.nobj + l) * d + t] = Jp[(size_t)t * kj + P.obj_col[l]];
            }
    }
    return 0;
}
static int ps_eval_point(mrbf_ctx *ctx, const ps::Problem &P, const double *x_host, std::vector<double> &allF, std::vector<double> *Jobj) {
    return ps_eval_points(ctx, P, x_host, 1, allF, Jobj);
}

// weights of the proximal minimax step: minimise  sigma / 2 |sum_l lam_l g_l|^2 - sum_l lam_l gap_l  over the simplex (M = G G',
// gap_l = F_l - max F <= 0), Frank-Wolfe with exact line search.  sigma -> infinity: the minimum-norm point of the hull of ALL
// gradients (the common descent direction); sigma -> 0: the gradient of the active objective alone.  In between an objective
// counts as far as a step of that length can make it active -- no fixed "nearly active" tolerance.
static void prox_weights(int k, const std::vector<double> &M, const std::vector<double> &gap, double sigma, std::vector<double> &lam) {
    int b0 = 0;
    for (int a = 1; a < k; ++a)
        if (gap[a] > gap[b0]) b0 = a;
    lam.assign(k, 0.0);
    lam[b0] = 1.0;
    std::vector<double> Ml(k);
    for (int it = 0; it < 400; ++it) {
        double lMl = 0.0, lg = 0.0;
        for (int a = 0; a < k; ++a) {
            Ml[a] = 0.0;
            for (int b = 0; b < k; ++b) Ml[a] += M[(size_t)a * k + b] * lam[b];
            lMl += lam[a] * Ml[a];
            lg += lam[a] * gap[a];
        }
        int best = 0;
        double gbest = sigma * Ml[0] - gap[0];
        for (int a = 1; a < k; ++a) {
            const double ga = sigma * Ml[a] - gap[a];
            if (ga < gbest) {
                gbest = ga;
                best = a;
            }
        }
        const double slope = (sigma * lMl - lg) - gbest;  // -(directional derivative towards the vertex) >= 0
        if (slope <= 1e-14 * std::max(std::fabs(sigma * lMl) + std::fabs(lg), 1e-300)) break;
        const double curv = sigma * (lMl - 2.0 * Ml[best] + M[(size_t)best * k + best]);
        const double gamma = curv > 0.0 ? std::min(1.0, slope / curv) : 1.0;
        for (int a = 0; a < k; ++a) lam[a] *= (1.0 - gamma);
        lam[best] += gamma;
    }
}

// Gradient refinement of a point of the box: a proximal steepest-descent method for
//     min_x  max_{l in objs} (m_l(x) - off_l) / scale_l     subject to the problem's constraints and the box
// -- the Pascoletti-Serafini subproblem itself with objs = all objectives, off = m(x_n), scale = r (the value is tau, kept in
// [-1, 0]), one objective of the local ideal point with objs = {l}, off = 0, scale = 1.  (The reference hands a local NLopt
// algorithm for the polish, descent.jl:560-569; the paper benchmark uses :LD_MMA with 100 (d + 1) evaluations,
// examples/large_scale_benchmarks.jl:217-219.)  Per iteration one evaluation with Jacobians, then NSTEP trial points
// x - sigma_u sum_l lam_l(sigma_u) g_l for NSTEP proximity parameters a factor two apart (each with its own weights, see
// prox_weights), projected onto the box, evaluated as ONE batch; components that the step would push through an active bound are
// taken out and the weights recomputed.  The best feasible improving trial point is taken; the range of sigma follows the accepted
// steps.  It stops when the budget is used up or no step down to 1e-9 of the box improves (stationarity), NOT at the first line
// search without improvement.  Every accepted iterate is feasible.  Each iteration costs 1 + NSTEP evaluations.
// One proximal minimax descent as a state machine: `trials` writes the NSTEP trial points of the next iteration (false: the descent has
// ended), `consume` takes their values and Jacobians.  Several independent descents (the k ideal-point refinements) advance in
// lockstep through ONE evaluation per iteration (ps_descend_many) -- a round trip per iteration for all of them instead of one each.
struct Descent {
    static constexpr int NSTEP = 12;
    const ps::Problem *P = nullptr;
    const std::vector<double> *lb = nullptr, *ub = nullptr;
    std::vector<int> objs;
    std::vector<double> off, scale;
    bool clamp = false;
    int max_evals = 0;
    double xtol_rel = 0.0;
    double val = 0.0;
    std::vector<double> x;
    // state
    int evals = 0, nsmall = 0;
    double width = 0.0, frac = 0.25;  // the largest trial step moves the fastest component by `frac` of the box
    bool have = false, done = false;  // have: allF / J hold the values and the objectives' Jacobian rows at x
    std::vector<double> allF, J, XT;

    void start() {
        const int d = P->d;
        width = 0.0;
        for (int t = 0; t < d; ++t) width = std::max(width, (*ub)[t] - (*lb)[t]);
        XT.assign((size_t)NSTEP * d, 0.0);
        done = !(width > 0.0);
    }
    bool wants_start_eval() const { return !done && !have && evals + 1 + NSTEP <= max_evals; }
    void take_start(const double *F_row, const double *J_rows) {  // values (nftot) and Jacobian rows (nobj x d) at x
        allF.assign(F_row, F_row + P->nftot);
        J.assign(J_rows, J_rows + (size_t)P->nobj * P->d);
        ++evals;
        have = true;
    }
    // the trial points of the next iteration into XT; false: ended (budget, no gradient, range exhausted)
    bool trials() {
        if (done) return false;
        if (!have || evals + NSTEP > max_evals) {
            done = true;
            return false;
        }
        const int d = P->d, k = (int)objs.size();
        std::vector<double> F(k), gap(k), dir(d), M((size_t)k * k), lam, G((size_t)k * d);
        std::vector<char> fixed(d);
        double tmax = -INFINITY, gmax = 0.0;
        for (int l = 0; l < k; ++l) {
            F[l] = (P->objective(allF, objs[l]) - off[l]) / scale[l];
            tmax = std::max(tmax, F[l]);
        }
        for (int l = 0; l < k; ++l) {
            gap[l] = F[l] - tmax;
            for (int t = 0; t < d; ++t) {
                G[(size_t)l * d + t] = J[(size_t)objs[l] * d + t] / scale[l];
                gmax = std::max(gmax, std::fabs(G[(size_t)l * d + t]));
            }
        }
        if (!(gmax > 0.0) || !(gmax < INFINITY)) {
            done = true;
            return false;
        }
        const double sigma0 = frac * width / gmax;
        for (int u = 0; u < NSTEP; ++u) {
            const double sigma = sigma0 * std::ldexp(1.0, -u);
            std::fill(fixed.begin(), fixed.end(), 0);
            for (int pass = 0; pass < 2; ++pass) {
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b <= a; ++b) {
                        double sdot = 0.0;
                        for (int t = 0; t < d; ++t)
                            if (!fixed[t]) sdot += G[(size_t)a * d + t] * G[(size_t)b * d + t];
                        M[(size_t)a * k + b] = M[(size_t)b * k + a] = sdot;
                    }
                prox_weights(k, M, gap, sigma, lam);
                bool changed = false;
                for (int t = 0; t < d; ++t) {
                    double v = 0.0;
                    if (!fixed[t])
                        for (int a = 0; a < k; ++a) v -= sigma * lam[a] * G[(size_t)a * d + t];
                    dir[t] = v;
                    if (!fixed[t] && ((x[t] <= (*lb)[t] && v < 0.0) || (x[t] >= (*ub)[t] && v > 0.0))) {
                        fixed[t] = 1;
                        dir[t] = 0.0;
                        changed = true;
                    }
                }
                if (!changed) break;
            }
            for (int t = 0; t < d; ++t) XT[(size_t)u * d + t] = std::min(std::max(x[t] + dir[t], (*lb)[t]), (*ub)[t]);
        }
        return true;
    }
    // values (NSTEP x nftot) and Jacobian rows (NSTEP x nobj x d) of the trial points
    void consume(const double *allFT, const double *JT) {
        const int d = P->d, k = (int)objs.size();
        std::vector<double> row(P->nftot);
        evals += NSTEP;
        int bestj = -1;
        double bestv = val - 1e-13 * std::max(1.0, std::fabs(val));
        for (int j = 0; j < NSTEP; ++j) {
            for (int c = 0; c < P->nftot; ++c) row[c] = allFT[(size_t)j * P->nftot + c];
            double tt = -INFINITY;
            for (int l = 0; l < k; ++l) tt = std::max(tt, (P->objective(row, objs[l]) - off[l]) / scale[l]);
            if (!(tt == tt) || !(tt < INFINITY)) continue;
            bool feas = P->violation(row, &XT[(size_t)j * d]) == 0.0;
            if (clamp) {
                // tau = the t this x admits, a hair towards 0 so that rounding cannot make the PS constraints fail (descent.jl:443)
                tt = std::min(std::max(tt < 0.0 ? tt * (1.0 - 1e-14) : tt, -1.0), 0.0);
                for (int l = 0; l < k; ++l) feas = feas && (P->objective(row, objs[l]) - off[l] - tt * scale[l] <= 0.0);
            }
            if (feas && tt < bestv) {
                bestv = tt;
                bestj = j;
            }
        }
        if (bestj < 0) {
            frac *= std::ldexp(1.0, -NSTEP);  // nothing down to 2^-11 of the range improved: continue below it
            if (frac < 1e-9) done = true;
            return;
        }
        bool small = true;  // NLopt's xtol_rel test (descent.jl:379, :485), on two accepted steps in a row
        for (int t = 0; t < d; ++t) {
            const double xn2 = XT[(size_t)bestj * d + t];
            small = small && std::fabs(xn2 - x[t]) <= xtol_rel * std::max(std::fabs(xn2), 1e-300);
            x[t] = xn2;
        }
        allF.assign(allFT + (size_t)bestj * P->nftot, allFT + (size_t)(bestj + 1) * P->nftot);
        J.assign(JT + (size_t)bestj * P->nobj * d, JT + (size_t)(bestj + 1) * P->nobj * d);
        val = bestv;
        frac = std::min(1.0, frac * std::ldexp(4.0, -bestj));  // the accepted step sits two levels below the top of the next range
        if (clamp && val <= -1.0) done = true;
        nsmall = small ? nsmall + 1 : 0;
        if (nsmall >= 2) done = true;
    }
};

// the descents of `ds` in lockstep: one evaluation (values + Jacobians of every active descent's points) per iteration
static int ps_descend_many(mrbf_ctx *ctx, const ps::Problem &P, std::vector<Descent> &ds) {
    const int d = P.d;
    constexpr int NSTEP = Descent::NSTEP;
    std::vector<double> X, allF, J;
    std::vector<int> who;
    for (auto &D : ds) D.start();
    // the start points (values + Jacobian rows), one call
    for (size_t i = 0; i < ds.size(); ++i)
        if (ds[i].wants_start_eval()) who.push_back((int)i);
    if (!who.empty()) {
        X.resize(who.size() * (size_t)d);
        for (size_t w = 0; w < who.size(); ++w) std::copy(ds[who[w]].x.begin(), ds[who[w]].x.end(), X.begin() + w * (size_t)d);
        MRBF_TRY(ps_eval_points(ctx, P, X.data(), (int)who.size(), allF, &J));
        for (size_t w = 0; w < who.size(); ++w) ds[who[w]].take_start(&allF[w * (size_t)P.nftot], &J[w * (size_t)P.nobj * d]);
    }
    for (;;) {
        who.clear();
        for (size_t i = 0; i < ds.size(); ++i)
            if (ds[i].trials()) who.push_back((int)i);
        if (who.empty()) break;
        X.resize(who.size() * (size_t)NSTEP * d);
        for (size_t w = 0; w < who.size(); ++w) std::copy(ds[who[w]].XT.begin(), ds[who[w]].XT.end(), X.begin() + w * (size_t)NSTEP * d);
        MRBF_TRY(ps_eval_points(ctx, P, X.data(), (int)who.size() * NSTEP, allF, &J));
        for (size_t w = 0; w < who.size(); ++w)
            ds[who[w]].consume(&allF[w * (size_t)NSTEP * P.nftot], &J[w * (size_t)NSTEP * P.nobj * d]);
    }
    return 0;
}

static int ps_descend(mrbf_ctx *ctx, const ps::Problem &P, const std::vector<double> &lb, const std::vector<double> &ub, const std::vector<int> &objs,
                      const std::vector<double> &off, const std::vector<double> &scale, bool clamp, int max_evals, double xtol_rel, double &val,
                      std::vector<double> &x, int *evals_out) {
    std::vector<Descent> ds(1);
    Descent &D = ds[0];
    D.P = &P;
    D.lb = &lb;
    D.ub = &ub;
    D.objs = objs;
    D.off = off;
    D.scale = scale;
    D.clamp = clamp;
    D.max_evals = max_evals;
    D.xtol_rel = xtol_rel;
    D.val = val;
    D.x = x;
    MRBF_TRY(ps_descend_many(ctx, P, ds));
    val = D.val;
    x = D.x;
    *evals_out = D.evals;
    return 0;
}

static int fetch_host(mrbf_ctx *ctx, const double *src, size_t cnt, std::vector<double> &dst) {
    dst.resize(cnt);
    if (cnt) MRBF_HIP(ctx, hipMemcpy(dst.data(), src, cnt * sizeof(double), hipMemcpyDefault));
    return 0;
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_ps_step_problem(mrbf_ctx *ctx, const mrbf_ps_problem *prob, const double *x_n, const double *lb_eff, const double *ub_eff,
                                        const double *fx_n, const double *r_or_null, const mrbf_ps_options *opts, double *x_trial,
                                        double *mx_trial, double *r_out, mrbf_ps_info *info) {
    if (!ctx) return -1;
    if (!prob) return fail(ctx, -2, "problem is NULL");
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
    std::memset(info, 0, sizeof(*info));
    // ---- the function table
    if (prob->n_models < 1 || prob->n_models > MAXMODELS || !prob->models || !prob->roles)
        return fail(ctx, -2, "mrbf_ps_step: 1..%d grouped models with a roles table are required", MAXMODELS);
    Problem P;
    P.nmodels = prob->n_models;
    P.nobj = prob->n_objectives;
    if (P.nobj < 1 || P.nobj > MAXOBJ) return fail(ctx, -2, "mrbf_ps_step: %d objectives (device path: 1..%d)", P.nobj, MAXOBJ);
    std::vector<int> seen(P.nobj, 0);
    for (int j = 0, e = 0; j < P.nmodels; ++j) {
        const mrbf_model *M = prob->models[j];
        if (!M) return fail(ctx, -2, "mrbf_ps_step: model %d is NULL", j);
        if (j == 0) P.d = M->d;
        if (M->d != P.d) return fail(ctx, -2, "mrbf_ps_step: model %d has %d variables, model 0 has %d", j, M->d, P.d);
        P.models[j] = M;
        P.foff[j] = P.nftot;
        P.nftot += M->k;
        for (int c = 0; c < M->k; ++c, ++e) {
            const int role = prob->roles[e];
            if (role >= 0) {
                if (role >= P.nobj || seen[role]) return fail(ctx, -2, "mrbf_ps_step: roles[%d] = %d is not a (new) objective position", e, role);
                seen[role] = 1;
                P.obj_model[role] = j;
                P.obj_col[role] = c;
            } else if (role == MRBF_ROLE_EQ || role == MRBF_ROLE_INEQ) {
                if (P.ncon >= MAXCON) return fail(ctx, -2, "mrbf_ps_step: more than %d modelled constraints", MAXCON);
                P.con_model[P.ncon] = j;
                P.con_col[P.ncon] = c;
                P.con_eq[P.ncon] = role == MRBF_ROLE_EQ;
                ++P.ncon;
            } else if (role != MRBF_ROLE_NONE) {
                return fail(ctx, -2, "mrbf_ps_step: roles[%d] = %d is not a role", e, role);
            }
        }
    }
    for (int l = 0; l < P.nobj; ++l)
        if (!seen[l]) return fail(ctx, -2, "mrbf_ps_step: objective %d is not an output of any model", l);
    const int d = P.d, k = P.nobj;
    if (mrbf_dispatch_ps(d, k, P.nmodels, P.ncon, prob->n_lin_eq + prob->n_lin_ineq, 0) != MRBF_DISPATCH_DEVICE)
        return fail(ctx, -2, "mrbf_ps_step: d = %d / k = %d / %d constraints outside the device path (ask mrbf_dispatch_ps first)", d, k, P.ncon);
    P.nlin_eq = prob->n_lin_eq;
    P.nlin_ineq = prob->n_lin_ineq;
    if (P.nlin_eq < 0 || P.nlin_ineq < 0) return fail(ctx, -2, "mrbf_ps_step: negative constraint count");
    if ((P.nlin_eq && (!prob->A_eq || !prob->b_eq)) || (P.nlin_ineq && (!prob->A_ineq || !prob->b_ineq)))
        return fail(ctx, -2, "mrbf_ps_step: linear constraint matrices are NULL");
    P.eq_tol = prob->eq_tol >= 0.0 ? prob->eq_tol : 1e-8;
    // box, start point, direction: host copies first (the pointers may be host or device memory)
    std::vector<double> hlb, hub, hxn, hfx, r(k), mx(k), allF;
    MRBF_TRY(fetch_host(ctx, lb_eff, d, hlb));
    MRBF_TRY(fetch_host(ctx, ub_eff, d, hub));
    MRBF_TRY(fetch_host(ctx, x_n, d, hxn));
    if (fx_n) MRBF_TRY(fetch_host(ctx, fx_n, k, hfx));
    if (r_or_null) MRBF_TRY(fetch_host(ctx, r_or_null, k, r));
    for (int t = 0; t < d; ++t)
        if (!(hlb[t] <= hub[t])) return fail(ctx, -4, "lb_eff[%d] > ub_eff[%d]", t, t);
    MRBF_TRY(fetch_host(ctx, prob->A_eq, (size_t)P.nlin_eq * d, P.A_eq));
    MRBF_TRY(fetch_host(ctx, prob->b_eq, (size_t)P.nlin_eq, P.b_eq));
    MRBF_TRY(fetch_host(ctx, prob->A_ineq, (size_t)P.nlin_ineq * d, P.A_ineq));
    MRBF_TRY(fetch_host(ctx, prob->b_ineq, (size_t)P.nlin_ineq, P.b_ineq));

    hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1];
    MRBF_HIP(ctx, hipEventRecord(e0, ctx->stream));
    const bool need_ideal = r_or_null == nullptr;
    const int lam_ip = 20 * (d + 1), lam_ps = 20 * (d + 2);
    const int rows = std::max(need_ideal ? k * lam_ip : 0, lam_ps);
    // device arena: box, x_n, mx, r, linear constraints, evaluation batch, results, per-run state
    const size_t nlin = (size_t)P.nlin_eq + P.nlin_ineq;
    const size_t per_ip = (size_t)4 * lam_ip * d + d + 2 + (size_t)3 * lam_ip, per_ps = (size_t)4 * lam_ps * (d + 1) + d + 3 + (size_t)3 * lam_ps;
    const size_t cnt = (size_t)3 * d + 2 * k + nlin * (d + 1) + (size_t)rows * d + (size_t)rows * P.nftot + std::max(k * per_ip, per_ps) + 64;
    double *base;
    int *stat;
    MRBF_TRY(get_buf(ctx, S_PS_STATE, cnt, &base));
    MRBF_TRY(get_buf(ctx, S_PS_STAT, (size_t)4 * MAXRUNS, &stat));
    double *dlb = base, *dub = dlb + d, *dxn = dub + d, *dmx = dxn + d, *dr = dmx + k, *dlin = dr + k;
    double *Xeval = dlin + nlin * (d + 1), *F = Xeval + (size_t)rows * d;
    double *pool = F + (size_t)rows * P.nftot;
    MRBF_HIP(ctx, hipMemcpyAsync(dlb, hlb.data(), d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(dub, hub.data(), d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(dxn, hxn.data(), d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    {
        double *p = dlin;
        auto up = [&](const std::vector<double> &v, const double **dev) -> int {
            *dev = p;
            if (!v.empty()) MRBF_HIP(ctx, hipMemcpyAsync(p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            p += v.size();
            return 0;
        };
        MRBF_TRY(up(P.A_eq, &P.dA_eq));
        MRBF_TRY(up(P.b_eq, &P.db_eq));
        MRBF_TRY(up(P.A_ineq, &P.dA_ineq));
        MRBF_TRY(up(P.b_ineq, &P.db_ineq));
    }
    // mx = m(x_n)  (descent.jl:543)
    MRBF_TRY(ps_eval_point(ctx, P, hxn.data(), allF, nullptr));
    for (int l = 0; l < k; ++l) mx[l] = P.objective(allF, l);
    MRBF_HIP(ctx, hipMemcpyAsync(dmx, mx.data(), k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const int max_ip = opts->max_ideal_evals < 0 ? 500 * (d + 1) : opts->max_ideal_evals;   // descent.jl:527
    const int max_ps = opts->max_ps_evals < 0 ? 500 * (d + 1) : opts->max_ps_evals;          // descent.jl:416
    // The evolution strategy is a global method: in d + 1 >= 25 variables it locates a basin, it does not descend into it (with the
    // reference's defaults the step at d = 128 stayed at 7 % of what the subproblem allows).  A tenth of every global budget is
    // therefore kept back for gradient steps from the strategy's best point (ps_descend) -- the evaluations are counted against the
    // same budget, so the call never evaluates more than the configuration allows.  MRBF_PS_DBG bit 5 (32): the strategy alone.
    const int dbg_env = mrbf_env("MRBF_PS_DBG") ? atoi(mrbf_env("MRBF_PS_DBG")) : 0;
    const auto reserve = [&](int budget) { return (dbg_env & 32) || budget < 20 * 13 ? 0 : budget / 10; };
    const int res_ip = reserve(max_ip), res_ps = opts->max_polish_evals > 0 ? 0 : reserve(max_ps);
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
        R.f = p;
        p += lam;
        R.phi = p;
        p += lam;
        R.order = (int *)p;
        p += lam;
        R.stat = st;
    };
    auto run_batch = [&](Args &a, const double *start, double t0, int max_gens) -> int {
        int maxlam = 0, maxel = 0;
        a.rows = 0;
        for (int q2 = 0; q2 < a.nruns; ++q2) {
            a.rows = std::max(a.rows, a.runs[q2].off + a.runs[q2].lam);
            maxlam = std::max(maxlam, a.runs[q2].lam);
            maxel = std::max(maxel, a.runs[q2].lam * a.runs[q2].nvar);
        }
        int N = 1;
        while (N < maxlam) N <<= 1;
        // (populations below RS_MINLAM: room and waves for rank_small_waves -- a wave per 96 individuals, a chunk of draws in LDS;
        //  over ALL runs of the launch: a small run beside a large one must find its room too)
        size_t small_need = 0;
        int small_waves = 0;
        for (int q2 = 0; q2 < a.nruns; ++q2)
            if (a.runs[q2].lam < RS_MINLAM) {
                small_need = std::max(small_need, rws_smem_bytes(a.runs[q2].lam));
                small_waves = std::max(small_waves, rws_waves(a.runs[q2].lam));
            }
        const size_t shm = std::max(std::max((size_t)20 * maxlam, (size_t)12 * N), small_need);
        MRBF_HIP(ctx, hipFuncSetAttribute((const void *)ps_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        // one thread per pair of the largest population (the bitonic network's N / 2 pairs), whole waves, at most RANK_THREADS
        // (a small population that needs more waves than N / 2 threads hold takes N threads: the plain sort then runs as one element per thread)
        int rank_threads = (int)std::min<int64_t>(RANK_THREADS, std::max<int64_t>(64, round_up(N / 2, 64)));
        if (64 * small_waves > rank_threads) rank_threads = (int)std::min<int64_t>(RANK_THREADS, std::max<int64_t>(N, 64 * small_waves));
        // large populations: the transposition phases on RS_W workgroups per run (MRBF_PS_MULTI=0: one workgroup as in rounds 3 / 4)
        RankWs rw{};
        // (a counter wait that times out -- the workgroups of a run not resident together: a device shared with other work -- costs 5 ms; the
        // host sees the sticky failure word with the status words, every eight generations, and keeps to one workgroup per run from then on)
        const int multi_env = mrbf_env("MRBF_PS_MULTI") ? atoi(mrbf_env("MRBF_PS_MULTI")) : 1;  // 0: one workgroup per run; 2: the workgroup form of round 5
        bool multi = maxlam >= RS_MINLAM && ctx->ncu >= RS_W * a.nruns && !ctx->ps_multi_off && multi_env != 0;
        unsigned long long *draws = nullptr;
        size_t draws_per_run = 0;
        if (multi) MRBF_TRY(rank_ws_setup(ctx, maxlam, rw, &draws, &draws_per_run));
        rw.count_plain = (multi && multi_env != 2) ? 1 : 0;
        bool phases_possible = a.ncon + a.nlin_eq + a.nlin_ineq > 0 || (a.dbg & 4);
        for (int q2 = 0; q2 < a.nruns; ++q2) phases_possible = phases_possible || a.runs[q2].kind == 1;
        // one model, fused evaluation: the breeding kernel writes the next population centred and padded into the evaluation's own query
        // buffers (a launch per generation less; MRBF_PS_FUSEPAD=0: the evaluation's centring launch).  The buffers are the arena slots the
        // evaluation will ask for -- same size, same pointer; eval_fused checks that and centres itself otherwise.
        a.Xq = nullptr;
        a.xsq = nullptr;
        a.xmean = nullptr;
        a.xqD = 0;
        a.score_fused = (a.nlin_eq + a.nlin_ineq == 0 && !(mrbf_env("MRBF_PS_FUSESCORE") && atoi(mrbf_env("MRBF_PS_FUSESCORE")) == 0)) ? 1 : 0;
        {
            const int fusepad = mrbf_env("MRBF_PS_FUSEPAD") ? atoi(mrbf_env("MRBF_PS_FUSEPAD")) : 1;  // (read per call: the tests switch it inside one process)
            const mrbf_model *M0 = P.models[0];
            if (fusepad && P.nmodels == 1 && ctx->eval_impl != 1 && (M0->dpad == 64 || M0->dpad == 128 || M0->dpad == 256) && a.d <= 360) {
                const int64_t mpad = round_up((int64_t)a.rows, 64);
                MRBF_TRY(get_buf(ctx, S_EVAL_XC, (size_t)mpad * M0->dpad, &a.Xq));
                MRBF_TRY(get_buf(ctx, S_EVAL_XSQ, (size_t)mpad, &a.xsq));
                a.xmean = M0->mean;
                a.xqD = M0->dpad;
            }
        }
        hipLaunchKernelGGL(ps_init_kernel, dim3((unsigned)((maxel + 255) / 256), (unsigned)a.nruns), dim3(256), 0, ctx->stream, a, start, t0);
        std::vector<int> hstat((size_t)4 * a.nruns);
        const unsigned wave_blocks = (unsigned)((a.rows + 3) / 4);
        for (int g = 0; g < max_gens; ++g) {
            a.gen = g;
            ctx->eval_population = 1;
            ctx->eval_pre_xq = (g > 0 && a.Xq) ? a.Xq : nullptr;  // (generation 0 comes from ps_init_kernel: centred by the evaluation's own launch)
            int erc = 0;
            for (int j = 0; j < P.nmodels && !erc; ++j) erc = eval_model(ctx, P.models[j], a.rows, a.Xeval, const_cast<double *>(a.F[j]), nullptr, nullptr);
            ctx->eval_population = 0;
            ctx->eval_pre_xq = nullptr;
            MRBF_TRY(erc);
            if (!a.score_fused) hipLaunchKernelGGL(ps_score_kernel, dim3(wave_blocks), dim3(256), 0, ctx->stream, a);
            hipLaunchKernelGGL(ps_rank_kernel, dim3((unsigned)a.nruns), dim3(rank_threads), shm, ctx->stream, a, multi ? 1 : 0, rw);
            if (multi) {
                rank_phases_launch(ctx, a, rw, draws, draws_per_run, maxlam, multi_env == 2 ? 2 : 1, phases_possible);
                hipLaunchKernelGGL(ps_rank_kernel, dim3((unsigned)a.nruns), dim3(rank_threads), shm, ctx->stream, a, 2, rw);
            }
            hipLaunchKernelGGL(ps_breed_kernel, dim3(wave_blocks), dim3(256), 0, ctx->stream, a);
            if ((g & 7) == 7 || g + 1 == max_gens) {  // status words every 8 generations: stop when every run is done
                MRBF_HIP(ctx, hipMemcpyAsync(hstat.data(), a.runs[0].stat, hstat.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                int hfail = 0;
                if (multi) MRBF_HIP(ctx, hipMemcpyAsync(&hfail, rw.sync + RS_SYNC * MAXRUNS, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (hfail && !(a.dbg & 64)) {
                    multi = false;
                    ctx->ps_multi_off = 1;
                }
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
    a.nobj = k;
    a.lb = dlb;
    a.ub = dub;
    a.mx = dmx;
    a.r = dr;
    a.nmodels = P.nmodels;
    {
        double *fp = F;
        for (int j = 0; j < P.nmodels; ++j) {
            a.F[j] = fp;
            a.kf[j] = P.models[j]->k;
            fp += (size_t)rows * P.models[j]->k;
        }
    }
    for (int l = 0; l < k; ++l) {
        a.obj_model[l] = (short)P.obj_model[l];
        a.obj_col[l] = (short)P.obj_col[l];
    }
    a.ncon = P.ncon;
    for (int c = 0; c < P.ncon; ++c) {
        a.con_model[c] = (short)P.con_model[c];
        a.con_col[c] = (short)P.con_col[c];
        a.con_eq[c] = (short)P.con_eq[c];
    }
    a.nlin_eq = P.nlin_eq;
    a.nlin_ineq = P.nlin_ineq;
    a.A_eq = P.dA_eq;
    a.b_eq = P.db_eq;
    a.A_ineq = P.dA_ineq;
    a.b_ineq = P.db_ineq;
    a.eq_tol = P.eq_tol;
    a.Xeval = Xeval;
    a.seed = opts->seed;
    a.dbg = dbg_env;
    a.xtol_rel = xtol;
    // ---- local ideal point: the k single-objective minimisations side by side (descent.jl:404-412)
    if (need_ideal) {
        double *p = pool;
        a.nruns = k;
        for (int l = 0; l < k; ++l) make_run(a.runs[l], 0, l, lam_ip, d, l * lam_ip, max_ip - res_ip, p, stat + 4 * l);
        MRBF_TRY(run_batch(a, dxn, 0.0, (max_ip - res_ip + lam_ip - 1) / lam_ip + 1));
        std::vector<double> bx((size_t)k * (d + 2));
        std::vector<int> hs((size_t)4 * k);
        for (int l = 0; l < k; ++l)
            MRBF_HIP(ctx, hipMemcpyAsync(&bx[(size_t)l * (d + 2)], a.runs[l].best, (d + 2) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipMemcpyAsync(hs.data(), stat, hs.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // r = f(x_n) - ideal point (descent.jl:536-538); a run that never met a feasible point contributes its start value.  The runs'
        // best points are refined by gradient steps on their objective -- all k descents in lockstep, one evaluation per iteration
        std::vector<double> ideal(k);
        std::vector<Descent> ds;
        std::vector<int> dl;
        for (int l = 0; l < k; ++l) {
            const double *b = &bx[(size_t)l * (d + 2)];
            const bool found = b[d + 1] == 0.0 && std::isfinite(b[d]);
            ideal[l] = found ? b[d] : mx[l];
            info->evals_ideal += hs[(size_t)4 * l];
            info->generations += hs[(size_t)4 * l + 2];
            const int left = max_ip - hs[(size_t)4 * l];
            if (found && left >= 13 && !(dbg_env & 32)) {
                ds.emplace_back();
                Descent &D = ds.back();
                D.P = &P;
                D.lb = &hlb;
                D.ub = &hub;
                D.objs = std::vector<int>{l};
                D.off = std::vector<double>{0.0};
                D.scale = std::vector<double>{1.0};
                D.clamp = false;
                D.max_evals = left;
                D.xtol_rel = xtol;
                D.val = ideal[l];
                D.x.assign(b, b + d);
                dl.push_back(l);
            }
        }
        if (!ds.empty()) MRBF_TRY(ps_descend_many(ctx, P, ds));
        for (size_t i = 0; i < ds.size(); ++i) {
            ideal[dl[i]] = ds[i].val;
            info->evals_ideal += ds[i].evals;
        }
        for (int l = 0; l < k; ++l) r[l] = hfx[l] - ideal[l];
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
        make_run(a.runs[0], 1, 0, lam_ps, d + 1, 0, max_ps - res_ps, p, stat);
        MRBF_TRY(run_batch(a, dxn, opts->t0, (max_ps - res_ps + lam_ps - 1) / lam_ps + 1));
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
            // the polish the configuration asks for (its own budget, descent.jl:423-429); without one, the part of the global budget
            // that was kept back or left over
            std::vector<int> objs(k);
            for (int l = 0; l < k; ++l) objs[l] = l;
            const bool own = opts->max_polish_evals > 0;
            const int pbudget = own ? opts->max_polish_evals : ((dbg_env & 32) ? 0 : max_ps - hs[0]);
            if (pbudget >= 13) {
                int pe = 0;
                MRBF_TRY(ps_descend(ctx, P, hlb, hub, objs, mx, r, true, pbudget, xtol, tau, xt, &pe));
                if (own)
                    info->evals_polish = pe;
                else
                    info->evals_ps += pe;
            }
            info->tau = tau;
            info->status = MRBF_PS_OK;
        }
    }
    // mx_trial = m(x_trial)
    if (info->status == MRBF_PS_OK) {
        MRBF_TRY(ps_eval_point(ctx, P, xt.data(), allF, nullptr));
        for (int l = 0; l < k; ++l) mx[l] = P.objective(allF, l);
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

// one grouped model whose outputs are the objectives in order, no constraints: the common case (and the round-2 entry point)
extern "C" int32_t mrbf_ps_step(mrbf_ctx *ctx, const mrbf_model *model, const double *x_n, const double *lb_eff, const double *ub_eff,
                                const double *fx_n, const double *r_or_null, const mrbf_ps_options *opts, double *x_trial, double *mx_trial,
                                double *r_out, mrbf_ps_info *info) {
    if (!ctx) return -1;
    if (!model) return fail(ctx, -2, "model is NULL");
    if (model->k > ps::MAXOBJ) return fail(ctx, -2, "mrbf_ps_step: k = %d objectives (device path: <= %d)", model->k, ps::MAXOBJ);
    int32_t roles[ps::MAXOBJ];
    for (int l = 0; l < model->k; ++l) roles[l] = l;
    const mrbf_model *models[1] = {model};
    mrbf_ps_problem prob;
    std::memset(&prob, 0, sizeof(prob));
    prob.n_models = 1;
    prob.n_objectives = model->k;
    prob.models = models;
    prob.roles = roles;
    prob.eq_tol = -1.0;
    return mrbf_ps_step_problem(ctx, &prob, x_n, lb_eff, ub_eff, fx_n, r_or_null, opts, x_trial, mx_trial, r_out, info);
}

// Test hook: ONE ranking of a given generation (objective values f, violations phi; inf = outside the budget) by the kernels of the
// step above -- impl 0: ps_rank_kernel alone (one workgroup); 1: hand-over to ps_rank_wave_kernel (one wave per 64 individuals) and the
// finishing launch; 2: the same with the kernel giving up at once (what a counter time-out leaves behind); 6: hand-over to
// ps_rank_sort_kernel (sixteen workgroups, round 5); 3 - 5: see include/mrbf.h.  order_out[lam] = the
// individuals in rank order.  Lets the tests put crafted populations (nearly ranked, a few infeasible individuals: the no-swap exit
// is taken early) through both kernels and compare the orders entry by entry.
extern "C" int32_t mrbf_debug_ps_rank(mrbf_ctx *ctx, int32_t lam, const double *f, const double *phi, uint64_t seed, int32_t gen,
                                      int32_t impl, int32_t *order_out, int32_t *gave_up) {
    if (!ctx) return -1;
    using namespace ps;
    if (lam < 2 || lam > MAXLAM) return fail(ctx, -2, "mrbf_debug_ps_rank: lam = %d outside 2..%d", lam, MAXLAM);
    if (!f || !phi) return fail(ctx, -3, "f / phi is NULL");
    if (gen < 0) return fail(ctx, -6, "gen < 0");
    if (impl < 0 || impl > 9 || impl == 8) return fail(ctx, -7, "impl must be 0 .. 7 or 9");
    if ((impl == 1 || impl == 2 || impl == 6 || impl == 7) && lam < RS_MINLAM) return fail(ctx, -7, "mrbf_debug_ps_rank: the several-workgroup ranking takes populations >= %d", RS_MINLAM);
    if (!order_out) return fail(ctx, -8, "order_out is NULL");
    (void)hipSetDevice(ctx->device);
    double *base;
    int *stat;
    MRBF_TRY(get_buf(ctx, S_PS_STATE, (size_t)4 * lam + 16, &base));
    MRBF_TRY(get_buf(ctx, S_PS_STAT, (size_t)4 * MAXRUNS, &stat));
    Args a{};
    Run &R = a.runs[0];
    a.nruns = 1;
    a.seed = seed;
    a.gen = gen;
    a.xtol_rel = 1e-3;
    a.dbg = impl == 2 ? 64 : (impl == 3 ? 128 : (impl == 5 ? 256 : (impl == 4 ? 512 : (impl == 7 ? 1024 : (impl == 9 ? 2048 : 0)))));  // (3: the plain sort as one pair per thread through LDS; 5: no parent selection)
    R.nvar = 1;
    R.lam = lam;
    R.mu = (impl == 4 || impl == 5) ? (lam + 6) / 7 : lam;  // mu = lam: the whole order comes out; 4 / 5: the step's own mu (order_out beyond it: -1)
    R.max_evals = 1 << 30;
    R.X[0] = R.X[1] = base;      // one dummy variable per individual
    R.best = base + lam;         // [x, f, phi]
    R.f = base + lam + 4;
    R.phi = R.f + lam;
    R.order = reinterpret_cast<int *>(R.phi + lam);
    R.stat = stat;
    std::vector<double> h((size_t)lam + 4, 0.0);
    h[lam + 1] = h[lam + 2] = INFINITY;
    MRBF_HIP(ctx, hipMemcpyAsync(base, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(R.f, f, (size_t)lam * sizeof(double), hipMemcpyDefault, ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(R.phi, phi, (size_t)lam * sizeof(double), hipMemcpyDefault, ctx->stream));
    MRBF_HIP(ctx, hipMemsetAsync(stat, 0, (size_t)4 * MAXRUNS * sizeof(int), ctx->stream));
    int N = 1;
    while (N < lam) N <<= 1;
    const size_t shm = std::max(std::max((size_t)20 * lam, (size_t)12 * N), lam < RS_MINLAM ? rws_smem_bytes(lam) : (size_t)0);
    MRBF_HIP(ctx, hipFuncSetAttribute((const void *)ps_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    int rank_threads = (int)std::min<int64_t>(RANK_THREADS, std::max<int64_t>(64, round_up(N / 2, 64)));
    if (lam < RS_MINLAM && 64 * rws_waves(lam) > rank_threads) rank_threads = (int)std::min<int64_t>(RANK_THREADS, std::max<int64_t>(N, 64 * rws_waves(lam)));
    RankWs rw{};
    const bool several = impl == 1 || impl == 2 || impl == 6 || impl == 7;
    unsigned long long *draws = nullptr;
    size_t draws_per_run = 0;
    if (several) MRBF_TRY(rank_ws_setup(ctx, lam, rw, &draws, &draws_per_run));
    rw.count_plain = (several && impl != 6) ? 1 : 0;
    hipLaunchKernelGGL(ps_rank_kernel, dim3(1), dim3(rank_threads), shm, ctx->stream, a, several ? 1 : 0, rw);
    if (several) {
        rank_phases_launch(ctx, a, rw, draws, draws_per_run, lam, impl == 6 ? 2 : 1, true);
        hipLaunchKernelGGL(ps_rank_kernel, dim3(1), dim3(rank_threads), shm, ctx->stream, a, 2, rw);
    }
    MRBF_HIP(ctx, hipGetLastError());
    int hsync[2] = {0, 0};
    if (several) MRBF_HIP(ctx, hipMemcpyAsync(hsync, rw.sync, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipMemsetAsync(R.order + R.mu, 0xff, (size_t)(lam - R.mu) * sizeof(int), ctx->stream));
    MRBF_HIP(ctx, hipMemcpyAsync(order_out, R.order, (size_t)lam * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (gave_up) *gave_up = hsync[1];
    return MRBF_OK;
}
