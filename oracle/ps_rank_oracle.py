"""CPU restatement of the stochastic ranking inside the device Pascoletti-Serafini solver (TEST INFRASTRUCTURE ONLY).

The reference hands the subproblem to NLopt's GN_ISRES (/root/reference/src/descent.jl:478-510, algorithm chosen at :339); NLopt is
a third-party dependency that is not in /root/reference (Project.toml: NLopt), and its ISRES ranks a generation by lambda bubble
sweeps with the pairwise rule of Runarsson & Yao (compare by objective when both are feasible or with probability 0.45, else by the
constraint violation).  The device solver runs the PARALLEL form of those sweeps -- lambda odd-even transposition phases with the
same pairwise rule, draws from a counter-based Philox 4x32-10 keyed by (pair, four-phase group, generation, run) -- and its
trajectory is by design not NLopt's ("parity unpinned" against NLopt: SURVEY.md section 7, "NLopt owns the loop").  What this file
pins is that the two device kernels that can rank a population (one workgroup: ps_rank_kernel; sixteen workgroups with halo
windows: ps_rank_sort_kernel, csrc/ps_solver.hip) compute THE SAME ranking as this plain NumPy loop, including where the no-swap
exit is taken: tests/test_pascoletti_serafini.py compares the orders entry by entry.

Only tests/ may import this module.
"""
import numpy as np

RS_B = 256        # phases per chunk of the several-workgroup kernel = distance between two no-swap tests for lam >= RS_MINLAM
RS_MINLAM = 1024  # smallest population the several-workgroup kernel takes
M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox 4x32-10 on arrays of counters (uint64 holding 32-bit values); returns the four output words"""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & M32 for c in (c0, c1, c2, c3))
    k0 = np.uint64(k0) & M32
    k1 = np.uint64(k1) & M32
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32
        n1 = p1 & M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c0, c1, c2, c3


def stochastic_rank(f, phi, seed, gen, run=0, quiet=None):
    """lam odd-even transposition phases over (f, phi) with the random pairwise rule; the no-swap test every `quiet` phases
    (default: the device rule -- RS_B for lam >= RS_MINLAM, 16 below).  Returns (order, phases run)."""
    f = np.array(f, dtype=np.float64)
    phi = np.array(phi, dtype=np.float64)
    lam = f.size
    idx = np.arange(lam)
    if quiet is None:
        quiet = RS_B if lam >= RS_MINLAM else 16
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    swapped = True
    ph_done = 0
    draws = None
    for ph0 in range(0, lam, 4):
        if ph0 % quiet == 0:
            if not swapped:
                break
            swapped = False
        npairs = lam // 2 + 1
        pr = np.arange(npairs, dtype=np.uint64)
        draws = philox4x32_10(pr, np.full(npairs, ph0, dtype=np.uint64), np.full(npairs, gen * 16 + 1, dtype=np.uint64),
                              np.full(npairs, run, dtype=np.uint64), k0, k1)
        for q4 in range(4):
            ph = ph0 + q4
            if ph >= lam:
                break
            j = np.arange(ph & 1, lam - 1, 2)
            fa, fb, pa, pb = f[j], f[j + 1], phi[j], phi[j + 1]
            u = (draws[q4][j >> 1].astype(np.float64) + 0.5) * (1.0 / 4294967296.0)
            by_f = ((pa == 0.0) & (pb == 0.0)) | (u < 0.45)
            worse = np.where(by_f, fa > fb, pa > pb)
            if worse.any():
                swapped = True
                a, b = j[worse], j[worse] + 1
                f[a], f[b] = f[b].copy(), f[a].copy()
                phi[a], phi[b] = phi[b].copy(), phi[a].copy()
                idx[a], idx[b] = idx[b].copy(), idx[a].copy()
            ph_done = ph + 1
    return idx, ph_done
