"""Cross-block hand-off of the diagonal chain: from one traced + logged launch (MRBF_MEGA_TRACE=t.txt MRBF_MEGA_JLOG=j.txt) print, per
block column c, when P(c) had published its last 16-column panel, when S(c+1,c) saw it / finished, and when P(c+1) started factoring."""
import os
os.environ.setdefault("MRBF_EXPERIMENTS", "1")  # the library honours its MRBF_* switches only behind this gate
import sys
import numpy as np
T = np.loadtxt(sys.argv[1])
L = np.loadtxt(sys.argv[2])
kind, i, c = (L[:, k].astype(int) for k in range(3))
claim, s2, s3, s4, s5, s6, end = (L[:, k] for k in range(5, 12))
P = {int(cc): k for k, cc in enumerate(c) if kind[k] == 2}
S = {int(cc): k for k, cc in enumerate(c) if kind[k] == 3 and i[k] == cc + 1}
off = np.median([s5[P[int(r[0])]] - r[3] for r in T if int(r[0]) in P])  # jlog time - trace time
print("  c | core loop | pub->S sees last panel | S last step | S end->P(c+1) core start | total gap | S: window done before pub, first panel seen before pub")
rows = []
for r in T:
    cc = int(r[0])
    if cc + 1 not in P or cc not in S:
        continue
    pub = r[16] + off
    core0 = r[3] + off
    k = S[cc]
    nxt = s5[P[cc + 1]]
    rows.append((pub - core0, s6[k] - pub, end[k] - s6[k], nxt - end[k], nxt - pub, pub - s4[k], pub - s5[k]))
    print(f"{cc:3d} | {rows[-1][0]:7.1f} | {rows[-1][1]:7.1f} | {rows[-1][2]:7.1f} | {rows[-1][3]:7.1f} | {rows[-1][4]:7.1f} | {rows[-1][5]:8.1f} {rows[-1][6]:8.1f}")
R = np.array(rows)
print("median:", " ".join(f"{x:7.1f}" for x in np.median(R, axis=0)))
