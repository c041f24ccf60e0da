"""Synthetic workloads of the BASELINE.json configurations, shared by bench.py, tests/ and tools/.

All inputs are fp64, generated on the host with NumPy's PCG64 (seeds stated per config); nothing here touches the GPU.

  C2  d=32  n=2048  gaussian      k=1  single build+solve
  C3  d=64  n=8192  multiquadric  k=2  build+solve + 10 000 evals                      (the bench's default)
  C4  d=128 n=257   cubic         k=2  ZDT1, 64 starts = first 64 Halton points (examples/large_scale_benchmarks.jl:102-109),
                                       n = 2d+1 sites per start (:157), m = 50(d+1) = 6450 evals per start (:217)
  C5  d=256 n=16384 cubic + degree-1 tail (q = 257), k=2, m=1024, 256 problems (seeds 1000+p)
"""
import numpy as np

CONFIGS = {
    "C2": dict(kernel="gaussian", n=2048, d=32, k=1, m=0, deg=1, seed=2, problems=1,
               desc="C2: d=32 n=2048 gaussian deg1 k=1, single build+solve"),
    "C3": dict(kernel="multiquadric", n=8192, d=64, k=2, m=10000, deg=1, seed=3, problems=1,
               desc="C3: d=64 n=8192 multiquadric deg1 k=2, build+solve + 10000 evals (values+Jacobians)"),
    "C4": dict(kernel="cubic", n=257, d=128, k=2, m=6450, deg=1, seed=40, problems=64,
               desc="C4: ZDT1 d=128, 64 Halton starts, n=257 cubic deg1 k=2, build+solve + 6450 evals per start"),
    "C5": dict(kernel="cubic", n=16384, d=256, k=2, m=1024, deg=1, seed=1000, problems=256,
               desc="C5: d=256 n=16384 cubic deg1 (q=257) k=2, build+solve + 1024 evals, 256 problems"),
}

_PRIMES = None


def _first_primes(count):
    global _PRIMES
    if _PRIMES is None or len(_PRIMES) < count:
        out, c = [], 2
        while len(out) < max(count, 256):
            if all(c % p for p in out if p * p <= c):
                out.append(c)
            c += 1
        _PRIMES = out
    return _PRIMES[:count]


def halton(index, d):
    """index-th point (1-based, like HaltonPoint's iteration) of the d-dimensional Halton sequence"""
    x = np.empty(d)
    for t, base in enumerate(_first_primes(d)):
        f, r, i = 1.0, 0.0, index
        while i > 0:
            f /= base
            r += f * (i % base)
            i //= base
        x[t] = r
    return x


def zdt1(C):
    """ZDT1 (formulas of MultiObjectiveProblems.jl): f1 = x1, g = 1 + 9 sum(x[2:])/(d-1), f2 = g (1 - sqrt(f1/g))"""
    d = C.shape[1]
    f1 = C[:, 0]
    g = 1.0 + 9.0 * C[:, 1:].sum(axis=1) / (d - 1)
    return np.stack([f1, g * (1.0 - np.sqrt(f1 / g))], axis=1)


def two_parabolas(C):
    d = C.shape[1]
    return np.stack([((C - 1.0) ** 2).sum(axis=1) / d, ((C + 1.0) ** 2).sum(axis=1) / d], axis=1)


def problem(config, p=0):
    """(C, Y, X) of problem p of a configuration: sites n x d, values n x k, query points m x d"""
    cfg = CONFIGS[config]
    n, d, k, m = cfg["n"], cfg["d"], cfg["k"], cfg["m"]
    rng = np.random.Generator(np.random.PCG64(cfg["seed"] + 7919 * p))
    if config == "C4":
        # start p: the (p+1)-th Halton point; its training sites lie in the enlarged trust region (radius theta_enlarge_1 * Delta =
        # 2 * 0.1) around it, clipped to the feasible box [0,1]^d of ZDT1; the first site is the start point itself
        x0 = halton(p + 1, d)
        C = np.clip(x0[None, :] + 0.2 * (2.0 * rng.random((n, d)) - 1.0), 0.0, 1.0)
        C[0] = x0
        Y = zdt1(C)
        X = np.clip(x0[None, :] + 0.2 * (2.0 * rng.random((max(m, 1), d)) - 1.0), 0.0, 1.0)
        return C, Y, X
    C = rng.random((n, d))
    Y = two_parabolas(C)[:, :k]
    X = np.random.Generator(np.random.PCG64(cfg["seed"] + 1 + 7919 * p)).random((max(m, 1), d))
    return C, Y, X


def algorithmic(config):
    """SURVEY.md section 8d per-unit figures for one cycle (one problem) of a configuration"""
    cfg = CONFIGS[config]
    n, d, k, m = cfg["n"], cfg["d"], cfg["k"], cfg["m"]
    q = 0 if cfg["deg"] < 0 else (1 if cfg["deg"] == 0 else d + 1)
    return dict(
        gram_bytes=8.0 * n * d + 8.0 * n * n,                        # read centres once + write full Phi
        gram_flops=float(n) * n * d,                                 # GEMM form on the lower triangle (2 * n^2/2 * d)
        factor_flops=n ** 3 / 3.0,                                   # potrf
        project_flops=4.0 * n * n * q,                               # symm + syr2k (+ syrk 1 n^2 q not counted)
        project_bytes=3.0 * 8.0 * n * n,                             # Phi read once for Phi*Q1, read + written once by the rank-2q update
        solve_flops=2.0 * n * n * k,
        eval_flops=float(m) * n * (3 * d + 2 * k + 2 * k * d),
        eval_bytes=8.0 * (n * d + n * k + m * d + m * k + m * k * d),
    )
