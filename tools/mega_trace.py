"""Print the chain breakdown of one traced persistent factorisation (MRBF_MEGA_TRACE=file).
Stamps per diagonal job P(c): 0 claimed, 1 GEMM-loop part done, 2 last panel folded in (factorisation starts), 3 factor + inverse done,
4 published."""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1])
c = rows[:, 0]
P = rows[:, 1:9]
def d(a, b):
    return np.where((a >= 0) & (b >= 0), b - a, np.nan)
step = np.diff(P[:, 2])
print("  c   wait+gemm  fold-in  factor+inv  publish | start-of-factor  step | first-dep-ready last-dep-ready (relative to start of factor)")
for k in range(len(c)):
    print(f"{int(c[k]):3d}  {d(P[k,0],P[k,1]):8.1f} {d(P[k,1],P[k,2]):8.1f} {d(P[k,2],P[k,3]):8.1f} {d(P[k,3],P[k,4]):6.1f} | {P[k,2]:9.1f} {step[k-1] if k else 0:7.1f} | {P[k,5]-P[k,2] if P[k,5]>=0 else float('nan'):8.1f} {P[k,6]-P[k,2] if P[k,6]>=0 else float('nan'):8.1f}")
if np.any(P[:, 7] > 0):
    w = d(P[:, 7], P[:, 1])
    print("left-looking GEMM done -> tile written back (wait for the bulk updates + read-modify-write): median %.1f us, max %.1f; per column:" % (np.nanmedian(w), np.nanmax(w)))
    print(" ".join(f"{x:.0f}" for x in w))
    print("tile written back -> factorisation starts (streamed fold of the last panel):")
    print(" ".join(f"{x:.0f}" for x in d(P[:, 1], P[:, 2])))
print("median step", np.nanmedian(step), "mean step", np.nanmean(step), "median factor+inv", np.nanmedian(d(P[:, 2], P[:, 3])), "total", P[-1, 4])

S = rows[:, 9:16]
if np.any(S > 0):
    # second half of the record: cycle counts of the leaf wave inside the diagonal core (sums over the 8 panels), raw ticks of the shader clock
    import os
    names = ["pre-leaf", "leaf", "post-leaf", "barrier X", "look-ahead", "barrier Y", "tail+inverse"]
    if os.environ.get("MRBF_MEGA_TRACE_WAVE"):
        names = ["trailing rest", "store drain", "barrier X", "stores+operand", "panel solve", "barrier Y", "panel stores+next col"]
    med = np.median(S[1:], axis=0)
    tot = med.sum()
    fac = np.nanmedian(d(P[:, 2], P[:, 3]))
    print("diag core, leaf wave, median cycles per block (x = share; factor+inv median %.1f us -> %.0f cycles/us):" % (fac, tot / fac))
    for n, m in zip(names, med):
        print(f"  {n:12s} {m:9.0f}  {m/tot:5.2f}  ~{m/tot*fac:5.1f} us")

    pub = rows[:, 16]
    if np.any(pub > 0):
        gap = P[1:, 2] - pub[:-1]    # last panel of P(c) published -> P(c+1) starts its factorisation
        core = pub - P[:, 2]          # start of the factorisation -> last panel published
        print("start of factorisation -> last panel published: median %.1f us;  published -> next block's factorisation starts: median %.1f us (min %.1f, max %.1f)"
              % (np.median(core), np.median(gap), gap.min(), gap.max()))
