// Problem descriptor and scratch layout of the one-launch small-problem fit (small.hip) and of the batched evaluation
// (eval_fused.hip): shared by the kernels and by the host code that carves the arena (solve.hip, api.hip).
#pragma once
#include "common.hpp"

namespace mrbf {
namespace smallfit {

struct Prob {
    int n, d, k, q, deg;
    int n16, npad, dpad, q16;  // n16 = round_up(n, 16), npad = round_up(n, 128), dpad in {64, 128}, q16 = round_up(max(q, 1), 16)
    KP kp;
    const double *C, *Y;                     // sites n x d, values n x k (row-major, device)
    double *Xc, *sq, *mean, *W, *Wc, *lam;   // the model's arrays (mrbf_model)
    double *ws;                              // scratch, carve(npad, q16).total doubles
    int *flags;                              // [0] bad pivot of K (1-based), [1] bad pivot of the tail's Cholesky-QR, [2] shift not positive,
                                             // [3] the cluster failed (1: a barrier timed out, 2: its members are not on one XCD): repeat with nc = 1
    double *scal;                            // [0] trace(Q1' Phi Q1), [1] mu
    long long *stamps;                       // debug (MRBF_SMALL_STAMPS): wall_clock64 at the phase boundaries, or NULL
    int *cl;                                 // CL_WORDS words of the problem's workgroup cluster (zero at launch): arrivals, failure word, XCDs
    int mean_given;                          // the centroid already lies in `mean` (small_mean_kernel, same arithmetic): the batch path centres its
                                             // queries beside the fit instead of after it
    unsigned long long spin_ticks;           // bound of a cluster barrier's spin (wall_clock64 ticks)
    int fault;                               // test hook (MRBF_OPT_DEBUG_FAULT bit 2 (value 4)): member 1 leaves before the third barrier
    int diag6;                               // full 128 x 128 diagonal blocks by the persistent factorisation's one-barrier-per-panel core (diag_v6_core)
};

constexpr int CL_WORDS = 24;   // arrivals, failure word, XCD of up to 16 members
constexpr int MAX_CLUSTER = 16;

struct Carve {
    size_t Phi, Q1, Wm, V, G, Gx, LinvX, Linv, Pt, Ycol, B, Fy, Xs, T1, T2, Z, total;
    int ldz;
};
__host__ __device__ inline Carve carve(int npad, int q16) {
    Carve c;
    size_t o = 0;
    auto take = [&](size_t cnt) {
        const size_t at = o;
        o += (cnt + 15) & ~size_t(15);
        return at;
    };
    c.ldz = q16 + 16;
    c.Phi = take((size_t)npad * npad);
    c.Q1 = take((size_t)npad * q16);
    c.Wm = take((size_t)npad * q16);
    c.V = take((size_t)npad * q16);
    c.G = take((size_t)q16 * q16);
    c.Gx = take((size_t)128 * 128);
    c.LinvX = take((size_t)128 * 128);
    c.Linv = take((size_t)npad * 128);
    c.Pt = take((size_t)npad * 128);
    c.Ycol = take((size_t)npad * 16);
    c.B = take((size_t)npad * 16);
    c.Fy = take((size_t)npad * 16);
    c.Xs = take((size_t)npad * 16);
    c.T1 = take((size_t)c.ldz * 16);
    c.T2 = take((size_t)c.ldz * 16);
    c.Z = take((size_t)c.ldz * 16 + 16);
    c.total = o;
    return c;
}

}  // namespace smallfit

// one problem of a batched evaluation (eval_fused.hip): the kernels of mrbf_eval with the problem index in the grid
struct EvalDesc {
    const double *X;      // m x d query points (row-major)
    const double *mean;   // the model's centroid
    double *Xq, *xsq;     // mpad x D centred + zero-padded queries, squared norms (scratch)
    const double *Cc, *csq, *Wc, *lam;  // the model: centred centres (npad x D), their norms, weights (npad x k col-major), tail
    int64_t npad, mpad, m;
    int d, k, q, tiles_per_split, nsplit;
    int ntiles;           // 64-centre tiles that hold real centres, ceil(n / 64) <= npad / 64: the padding tiles beyond are never walked
    int nsub;             // ... and 16-centre steps, ceil(n / 16): the last tile stops there (n = 2d + 1 = 257: 17 steps, not 20)
    KP kp;
    double *vpart, *sapart, *gpart;     // per-split partials (scratch)
    double *vals, *jac;                 // m x k, m x (k x d column-major); jac may be NULL for the whole batch only
};
// the centre-range split a single mrbf_eval of m points on a model with ntiles tiles of 64 centres uses (the batch takes the same one, so
// that a batch and single calls add up their partial sums in the same order)
int eval_nsplit(const mrbf_ctx *ctx, int64_t m, int ntiles, bool check_call = false);  // ntiles = ceil(n / 64); check_call: the residual check's evaluation at the sites
int outputs_per_pass(int k, int D, bool want_jac = true);  // outputs of a model the fused evaluation handles per pass
// all descriptors: same kernel id / fast flag / padded dimension D (64 or 128) / k; dev_descs = the same array in device memory
// centred: the descriptors' Xq / xsq are already filled (center_pad_batch on all descriptors of a batch, launched beside the fit)
int eval_fused_batch(mrbf_ctx *ctx, const KP &kp, int D, int k, bool want_jac, const EvalDesc *host_descs, const EvalDesc *dev_descs, int count,
                     bool centred = false);
// Xq = X - mean (zero padded to D = 64 / 128 by the descriptor's d), xsq = |Xq|^2 for `count` descriptors of any mix of dimensions
int center_pad_batch(mrbf_ctx *ctx, const EvalDesc *dev_descs, int count, int64_t max_mpad);

// does the one-launch path take a problem of this shape on this context?  (path: MRBF_PATH_* chosen by fit_model)
bool small_fit_applies(const mrbf_ctx *ctx, int64_t n, int d, int k, int q, int path);
// count == 1 and dev_probs == nullptr: the descriptor travels as a kernel argument; else one workgroup per descriptor of dev_probs
int launch_small_fit(mrbf_ctx *ctx, const smallfit::Prob *host_probs, int count, const smallfit::Prob *dev_probs, int nc);
// workgroups per problem for a launch of `count` problems: the power of two <= 256 / (problems rounded up to 8), at most 16 -- a batch of 8
// (what one of eight GPUs sees of 64 starts) spreads every problem over 16 compute units, a batch of 64 over 4; 1 after a cluster
// failure on this context, on devices that are not 8 XCDs x 32 CUs of gfx950, or with MRBF_SMALL_NC=1
int small_fit_cluster(const mrbf_ctx *ctx, int count);
// the centroids of `count` problems into their `mean` arrays (what small_fit_kernel computes itself unless Prob::mean_given)
int launch_small_means(mrbf_ctx *ctx, const smallfit::Prob *dev_probs, int count);

// The tail basis of the launch chain in three launches (small.hip, TailQ): Q1 = [1/sqrt n | Xc Lx^-T], inv(Lx) (128 x 128, identity
// padded), T1 = Q1' Y, B = Y - Q1 T1 (npad x k, zero rows beyond n) and, if rows_out, the right-hand sides as rows npad .. npad + xt - 1
// of the matrix; flags[1] = 1-based index of a non-positive pivot of Xc'Xc (affinely dependent sites) or 0.  d <= 64, k <= 16, q = d + 1.
bool tail_basis_applies(const mrbf_model *M);
size_t tail_basis_scratch_doubles();
int launch_tail_basis(mrbf_ctx *ctx, const mrbf_model *M, const double *Y, double *scratch, double *LinvX, double *T1, double *Q1, double *B,
                      double *rows_out, int64_t ld, int xt, int *flags);

}  // namespace mrbf
