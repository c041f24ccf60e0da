#!/usr/bin/env python3
"""Fuzz of mrbf_round4 (the right-looking walk on two streams) against the independent oracle (oracle/sampling_oracle.py): random dimension,
candidate count, kernel, tail degree, max_points and spread of the candidates (clustered candidates are rejected in runs: blocks without a
single accepted site, walks that end in the middle of a block, partial last blocks).  The accepted lists must be identical.
    python tools/r4_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morbit.jl_amd as pkg
from morbit.jl_amd import sampling
from oracle import sampling_oracle as so

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
dup_eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4   # distance of the near-duplicates (1e-9: the pivot of a duplicate is rounding noise,
                                                               # every implementation decides it differently -- reported, not counted)
kernels = ["cubic", "gaussian", "multiquadric", "inv_multiquadric", "thin_plate_spline"]
bad = 0
t00 = time.perf_counter()
for case in range(ncase):
    d = int(rng.choice([2, 3, 5, 8, 13, 21, 34, 47, 64, 70, 96, 130]))
    deg = int(rng.choice([1, 1, 1, 0, -1]))
    name = kernels[int(rng.integers(len(kernels)))]
    if name in ("cubic", "thin_plate_spline") and deg < 1:
        deg = 1  # (conditionally positive definite of order 2: the reference needs the linear tail)
    n = int(rng.integers(20, 700))
    spread = float(rng.choice([0.6, 0.3, 0.05]))          # small spread: many near-duplicates, long rejected stretches
    ndup = int(rng.choice([0, 0, n // 4]))                # exact duplicates of earlier candidates
    x = np.full(d, 0.5)
    pts = x + spread * (rng.random((d + n, d)) - 0.5)
    if ndup:
        src = rng.integers(0, d + n - ndup, size=ndup)
        pts[d + n - ndup:] = pts[src] + dup_eps * rng.standard_normal((ndup, d))
    sites = np.vstack([x, pts])
    idx, _, _ = so.affinely_independent_indices(x, sites[1:], d, 0.05)
    start = [0] + [i + 1 for i in idx]
    if deg == 1 and len(start) != d + 1:
        continue
    cands = [i for i in range(len(sites)) if i not in start]
    full = (d + 1) * (d + 2) // 2
    mp = int(rng.choice([-1, len(start) + 1 + int(rng.integers(0, max(2, len(cands)))), full]))
    cfg = pkg.RbfConfig(kernel=name, polynomial_degree=deg, max_model_points=mp)
    kidp, ap, bp = pkg.rbf_model._get_kernel_params(1.0, cfg)
    S0, Cc = sites[start], sites[cands]
    want = so.rbf_round4(S0, Cc, kidp, ap, bp, deg, max_points=mp if mp > 0 else full)
    try:
        got, st = sampling.rbf_round4_device(cfg, S0, Cc, 1.0, keep_state=True)
        st.free()
    except Exception as e:
        got = "error: %r" % (e,)
    ok = got == want
    bad += 0 if ok else 1
    print("%s case %2d: d=%3d %-17s deg %2d candidates %3d (spread %.2f, %d duplicates) max_points %5d -> %3d accepted%s"
          % ("ok " if ok else "BAD", case, d, name, deg, len(cands), spread, ndup, mp, len(want), "" if ok else "  device: %s" % (got if isinstance(got, str) else len(got))), flush=True)
print("%d cases, %d mismatches, %.1f s" % (ncase, bad, time.perf_counter() - t00))
sys.exit(1 if bad else 0)
