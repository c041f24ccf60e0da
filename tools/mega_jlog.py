"""Analyse a job log of the persistent factorisation (MRBF_MEGA_JLOG=file).
columns: kind i c w wg claim s2 s3 s4 s5 s6 end   (us; -1 = not stamped)
kinds: 0 U (bulk), 1 T (half panel tile; w = half), 2 P (diag), 3 S (streamed panel tile), 4 UH"""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import sys
import numpy as np
L = np.loadtxt(sys.argv[1])
kind, i, c, w, wg = (L[:, k].astype(int) for k in range(5))
claim, s2, s3, s4, s5, s6, end = (L[:, k] for k in range(5, 12))
T_end = end.max()
print(f"jobs {len(L)}  makespan {T_end:.0f} us")
names = {0: "U", 1: "T", 2: "P", 3: "S", 4: "UH"}
for k in sorted(set(kind)):
    m = kind == k
    dur = end[m] - claim[m]
    print(f"kind {names[k]}: {m.sum()} jobs, mean duration {dur.mean():.1f} us, total {dur.sum()/1e3:.1f} ms-WG")
# bulk: wait inside (claim->s2), gemm (s2->s3), store (s3->end)
m = kind == 0
print(f"bulk: wait {np.mean(s2[m]-claim[m]):.1f}  gemm {np.mean(s3[m]-s2[m]):.1f}  store+publish {np.mean(end[m]-s3[m]):.1f}")
m = kind == 1
print(f"T: claim->first dep {np.nanmean(np.where(s2[m]>=0, s2[m]-claim[m], np.nan)):.1f}  ->window part done {np.mean(s4[m]-claim[m]):.1f}  wait Linv {np.mean(s5[m]-s4[m]):.1f}  trsm+store {np.mean(end[m]-s5[m]):.1f}")
# utilisation over time: number of WGs inside a job (excluding pure waits is not possible for all; report busy = in job)
bins = np.arange(0, T_end + 100, 100.0)
busy = np.zeros(len(bins) - 1)
comp = np.zeros(len(bins) - 1)
for a0, a1, arr in ((claim, end, busy),):
    for x0, x1 in zip(a0, a1):
        b0, b1 = int(x0 // 100), int(x1 // 100)
        for b in range(b0, min(b1, len(arr) - 1) + 1):
            arr[b] += max(0.0, min(x1, bins[b + 1]) - max(x0, bins[b])) / 100.0
# compute-only for bulk (s2..end) and T (approximations)
for x0, x1 in zip(np.where(kind == 0, s2, np.where(kind == 4, s2, claim)), end):
    b0, b1 = int(x0 // 100), int(x1 // 100)
    for b in range(b0, min(b1, len(comp) - 1) + 1):
        comp[b] += max(0.0, min(x1, bins[b + 1]) - max(x0, bins[b])) / 100.0
print("time(us)  WGs-in-job  (bulk compute + other jobs)")
for b in range(0, len(busy), 2):
    print(f"{bins[b]:7.0f} {busy[b]:6.0f} {comp[b]:6.0f}")
# per-column milestones
print("col: P factor start(s5) P end | S1 end | first T end  last T end | first U(c) ready->claim, last U of col c end")
for col in range(0, int(c.max()) + 1):
    mp = (kind == 2) & (c == col)
    ms = (kind == 3) & (c == col) & (i == col + 1)
    mt = (kind == 1) & (c == col)
    mu = ((kind == 0) | (kind == 4)) & (c == col)
    def g(a, m, f):
        return f(a[m]) if m.any() else float('nan')
    print(f"{col:3d}  {g(s5, mp, np.max):8.1f} {g(end, mp, np.max):8.1f} | {g(end, ms, np.max):8.1f} | {g(end, mt, np.min):8.1f} {g(end, mt, np.max):8.1f} | "
          f"{g(claim, mu, np.min):8.1f} {g(end, mu, np.max):8.1f}  nU={mu.sum()}")
