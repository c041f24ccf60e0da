/* CPU oracle, plain C, "faithful mode" (TEST INFRASTRUCTURE ONLY -- never linked
 * into the product library; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, as the checker / the timed CPU baseline).
 *
 * Restates the call pattern of the reference's hot path, single-threaded:
 *   - Gram assembly with a per-pair norm(x - c) in difference form
 *       RBF.get_matrices / RBFInterpolationModel   src/models/RbfModel.jl:374-375, :759-763
 *   - dense LU with partial pivoting on the saddle system [Phi Pi; Pi' 0] (Julia `\`)
 *   - evaluation ONE point and ONE output at a time, value sweep and gradient
 *     sweep separate, an n-vector of kernel values allocated per call
 *       eval_models / get_gradient                  src/models/RbfModel.jl:783-795
 *       _get_optim_handle closures                  src/AbstractSurrogateInterface.jl:98-106
 *
 * PARITY UNPINNED by the reference (see oracle/rbf_oracle.py header): the
 * arithmetic lives in RadialBasisFunctionModels.jl 0.3.4, absent from
 * /root/reference; radial functions below restate its published formulas.
 *
 * Layouts: centres C n x d row-major; values Y, weights W n x k row-major;
 * poly coefficients Lam q x k row-major, basis [1, x_1..x_d]; Phi, Pi column-major.
 * kernel ids follow Morbit.RbfKernels (RbfModel.jl:48-54):
 *   0 cubic(a=beta) 1 inv_multiquadric(a=alpha,b=beta) 2 multiquadric(a,b)
 *   3 thin_plate_spline(a=k) 4 gaussian(a=alpha)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double sgn_pow(int e) { return (e & 1) ? -1.0 : 1.0; }

double orc_phi(int kid, double a, double b, double rho) {
    switch (kid) {
    case 4: return exp(-(a * rho) * (a * rho));
    case 2: return sgn_pow((int)ceil(b)) * pow(1.0 + (a * rho) * (a * rho), b);
    case 1: return pow(1.0 + (a * rho) * (a * rho), -b);
    case 0: return sgn_pow((int)ceil(a / 2.0)) * pow(rho, a);
    case 3: {
        int k = (int)a;
        if (rho == 0.0) return 0.0;
        return sgn_pow(k + 1) * pow(rho, 2.0 * k) * log(rho);
    }
    }
    return NAN;
}

/* phi'(rho) / rho with the rho = 0 term as in oracle/rbf_oracle.py */
double orc_psi(int kid, double a, double b, double rho) {
    switch (kid) {
    case 4: return -2.0 * a * a * exp(-(a * rho) * (a * rho));
    case 2: return sgn_pow((int)ceil(b)) * 2.0 * a * a * b * pow(1.0 + (a * rho) * (a * rho), b - 1.0);
    case 1: return -2.0 * a * a * b * pow(1.0 + (a * rho) * (a * rho), -b - 1.0);
    case 0:
        if (rho == 0.0 && a < 2.0) return 0.0;
        return sgn_pow((int)ceil(a / 2.0)) * a * pow(rho, a - 2.0);
    case 3: {
        int k = (int)a;
        if (rho == 0.0) return 0.0;
        return sgn_pow(k + 1) * pow(rho, 2.0 * k - 2.0) * (2.0 * k * log(rho) + 1.0);
    }
    }
    return NAN;
}

int orc_poly_dim(int d, int deg) { return deg < 0 ? 0 : (deg == 0 ? 1 : d + 1); }

static double dist(const double *x, const double *c, int d) {
    double s = 0.0;
    for (int t = 0; t < d; ++t) {
        double u = x[t] - c[t];
        s += u * u;
    }
    return sqrt(s);
}

/* Phi n x n and Pi n x q, both column-major */
void orc_gram(int n, int d, const double *C, int kid, double a, double b, int deg,
              double *Phi, double *Pi) {
    int q = orc_poly_dim(d, deg);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i)
            Phi[(size_t)j * n + i] = (i == j) ? orc_phi(kid, a, b, 0.0)
                                              : orc_phi(kid, a, b, dist(C + (size_t)i * d, C + (size_t)j * d, d));
    if (Pi)
        for (int i = 0; i < n; ++i)
            for (int t = 0; t < q; ++t)
                Pi[(size_t)t * n + i] = (t == 0) ? 1.0 : C[(size_t)i * d + (t - 1)];
}

/* the same per-pair loop spread over `threads` OpenMP threads (columns are independent; the arithmetic of every entry is
 * unchanged).  threads <= 1 is the faithful single-threaded mode above; the tests use more threads only to keep the
 * n = 16384 oracle affordable. */
void orc_gram_mt(int n, int d, const double *C, int kid, double a, double b, int deg, double *Phi, double *Pi, int threads) {
    if (threads <= 1) {
        orc_gram(n, d, C, kid, a, b, deg, Phi, Pi);
        return;
    }
    int q = orc_poly_dim(d, deg);
    const double phi0 = orc_phi(kid, a, b, 0.0);
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i)
            Phi[(size_t)j * n + i] = (i == j) ? phi0 : orc_phi(kid, a, b, dist(C + (size_t)i * d, C + (size_t)j * d, d));
    if (Pi)
        for (int i = 0; i < n; ++i)
            for (int t = 0; t < q; ++t)
                Pi[(size_t)t * n + i] = (t == 0) ? 1.0 : C[(size_t)i * d + (t - 1)];
}

/* the first `cols` columns of Phi (n x cols, column-major) with the faithful per-pair loop: a bounded sample of the assembly
 * for bench.py's cpu_baseline (cost is proportional to the columns computed) */
void orc_gram_cols(int n, int d, const double *C, int kid, double a, double b, int cols, double *out) {
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < n; ++i)
            out[(size_t)j * n + i] = (i == j) ? orc_phi(kid, a, b, 0.0) : orc_phi(kid, a, b, dist(C + (size_t)i * d, C + (size_t)j * d, d));
}

/* dgesv restated: LU with partial pivoting, A is N x N column-major, B is N x nrhs column-major.
 * returns 0, or j+1 when U(j,j) is exactly zero. */
int orc_lu_solve(int N, int nrhs, double *A, double *B) {
    for (int j = 0; j < N; ++j) {
        int p = j;
        double best = fabs(A[(size_t)j * N + j]);
        for (int i = j + 1; i < N; ++i)
            if (fabs(A[(size_t)j * N + i]) > best) { best = fabs(A[(size_t)j * N + i]); p = i; }
        if (best == 0.0) return j + 1;
        if (p != j) {
            for (int c = 0; c < N; ++c) { double t = A[(size_t)c * N + j]; A[(size_t)c * N + j] = A[(size_t)c * N + p]; A[(size_t)c * N + p] = t; }
            for (int c = 0; c < nrhs; ++c) { double t = B[(size_t)c * N + j]; B[(size_t)c * N + j] = B[(size_t)c * N + p]; B[(size_t)c * N + p] = t; }
        }
        double piv = 1.0 / A[(size_t)j * N + j];
        for (int i = j + 1; i < N; ++i) A[(size_t)j * N + i] *= piv;
        for (int c = j + 1; c < N; ++c) {
            double ajc = A[(size_t)c * N + j];
            if (ajc != 0.0)
                for (int i = j + 1; i < N; ++i) A[(size_t)c * N + i] -= A[(size_t)j * N + i] * ajc;
        }
    }
    for (int c = 0; c < nrhs; ++c) {
        double *x = B + (size_t)c * N;
        for (int j = 0; j < N; ++j)
            for (int i = j + 1; i < N; ++i) x[i] -= A[(size_t)j * N + i] * x[j];
        for (int j = N - 1; j >= 0; --j) {
            x[j] /= A[(size_t)j * N + j];
            for (int i = 0; i < j; ++i) x[i] -= A[(size_t)j * N + i] * x[j];
        }
    }
    return 0;
}

/* a3: assemble + solve. W n x k row-major, Lam q x k row-major. returns LU info. */
int orc_fit(int n, int d, int k, const double *C, const double *Y, int kid, double a, double b, int deg,
            double *W, double *Lam) {
    int q = orc_poly_dim(d, deg);
    int N = n + q;
    double *S = (double *)calloc((size_t)N * N, sizeof(double));
    double *Phi = (double *)malloc((size_t)n * n * sizeof(double));
    double *Pi = (double *)malloc((size_t)n * (q ? q : 1) * sizeof(double));
    double *B = (double *)calloc((size_t)N * k, sizeof(double));
    orc_gram(n, d, C, kid, a, b, deg, Phi, Pi);
    for (int j = 0; j < n; ++j) memcpy(S + (size_t)j * N, Phi + (size_t)j * n, (size_t)n * sizeof(double));
    for (int t = 0; t < q; ++t)
        for (int i = 0; i < n; ++i) {
            S[(size_t)(n + t) * N + i] = Pi[(size_t)t * n + i];
            S[(size_t)i * N + (n + t)] = Pi[(size_t)t * n + i];
        }
    for (int l = 0; l < k; ++l)
        for (int i = 0; i < n; ++i) B[(size_t)l * N + i] = Y[(size_t)i * k + l];
    int info = orc_lu_solve(N, k, S, B);
    for (int l = 0; l < k; ++l) {
        for (int i = 0; i < n; ++i) W[(size_t)i * k + l] = B[(size_t)l * N + i];
        for (int t = 0; t < q; ++t) Lam[(size_t)t * k + l] = B[(size_t)l * N + n + t];
    }
    free(S); free(Phi); free(Pi); free(B);
    return info;
}

/* a4: mod.model(x, ell) -- one point, one output */
double orc_eval_one(int n, int d, int k, const double *C, const double *W, const double *Lam,
                    int kid, double a, double b, int deg, const double *x, int ell) {
    double *kv = (double *)malloc((size_t)n * sizeof(double)); /* the per-call n-vector */
    for (int i = 0; i < n; ++i) kv[i] = orc_phi(kid, a, b, dist(x, C + (size_t)i * d, d));
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += W[(size_t)i * k + ell] * kv[i];
    free(kv);
    int q = orc_poly_dim(d, deg);
    if (q >= 1) s += Lam[ell];
    for (int t = 1; t < q; ++t) s += Lam[(size_t)t * k + ell] * x[t - 1];
    return s;
}

/* a5: RBF.grad(model, x, ell) -- one point, one output */
void orc_grad_one(int n, int d, int k, const double *C, const double *W, const double *Lam,
                  int kid, double a, double b, int deg, const double *x, int ell, double *g) {
    for (int t = 0; t < d; ++t) g[t] = 0.0;
    for (int i = 0; i < n; ++i) {
        const double *c = C + (size_t)i * d;
        double f = W[(size_t)i * k + ell] * orc_psi(kid, a, b, dist(x, c, d));
        for (int t = 0; t < d; ++t) g[t] += f * (x[t] - c[t]);
    }
    if (orc_poly_dim(d, deg) > 1)
        for (int t = 0; t < d; ++t) g[t] += Lam[(size_t)(t + 1) * k + ell];
}

/* the closure loop of section 3.3 of SURVEY.md: for every point, for every output, value then gradient.
 * vals m x k row-major; jac m x (k x d column-major) or NULL */
void orc_eval_loop(int n, int d, int k, const double *C, const double *W, const double *Lam,
                   int kid, double a, double b, int deg, int m, const double *X, double *vals, double *jac) {
    double *g = (double *)malloc((size_t)d * sizeof(double));
    for (int p = 0; p < m; ++p)
        for (int l = 0; l < k; ++l) {
            vals[(size_t)p * k + l] = orc_eval_one(n, d, k, C, W, Lam, kid, a, b, deg, X + (size_t)p * d, l);
            if (jac) {
                orc_grad_one(n, d, k, C, W, Lam, kid, a, b, deg, X + (size_t)p * d, l, g);
                for (int t = 0; t < d; ++t) jac[(size_t)p * k * d + (size_t)t * k + l] = g[t];
            }
        }
    free(g);
}
