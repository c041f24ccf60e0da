"""Print the chain breakdown of one traced persistent factorisation (MRBF_MEGA_TRACE=file)."""
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1])
c = rows[:, 0]
P = rows[:, 1:9]
T = rows[:, 9:17]
def d(a, b):
    return np.where((a >= 0) & (b >= 0), b - a, np.nan)
print("P: wait+window_part  diag  publish | T: window_part  wait_P  gemm  store+publish | step")
step = np.diff(P[:, 3])
for k in range(len(c)):
    print(f"{int(c[k]):3d}  P {d(P[k,0],P[k,1]):7.1f} {d(P[k,1],P[k,2]):6.1f} {d(P[k,2],P[k,3]):5.1f} |"
          f" T {d(T[k,0],T[k,1]):7.1f} {d(T[k,1],T[k,2]):6.1f} {d(T[k,2],T[k,3]):6.1f} {d(T[k,3],T[k,4]):5.1f} |"
          f" Pdone {P[k,3]:8.1f} Tdone {T[k,4]:8.1f} Pstart {P[k,0]:8.1f} Tstart {T[k,0]:8.1f} step {step[k-1] if k else 0:6.1f}")
print("median step", np.nanmedian(step), "median diag", np.nanmedian(d(P[:,1],P[:,2])), "median Tgemm", np.nanmedian(d(T[:,2],T[:,3])),
      "Tdone->next P after-window", np.nanmedian(P[1:,1]-T[:-1,4]), "Pdone->T gemm start", np.nanmedian(T[:,2]-P[:,3]))
