"""Pascoletti-Serafini step on the device: evaluations per second of the population-batched solver vs one-point calls
(what the reference's NLopt callbacks amount to), C4-shaped surrogate (d = 12, n = 512 centres, k = 2)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import morbit, importlib
pkg = importlib.import_module("morbit.jl_amd")
ps = importlib.import_module("morbit.jl_amd.pascoletti_serafini")
rng = np.random.default_rng(0)
d, n = 12, 512
C = rng.random((n, d))
Y = np.stack([np.sum((C - 0.3) ** 2, axis=1), np.sum((C - 0.7) ** 2, axis=1) + 0.1 * np.sin(5 * C[:, 0])], axis=1)
mod = pkg.update_model(pkg.RbfConfig(kernel="cubic"), C, Y)
sc = pkg.surrogates.SurrogateContainer(objectives=[pkg.surrogates.RefSurrogate(mod, [0, 1])])
calls = {"n": 0, "pts": 0, "t": 0.0}
def ev(X):
    t0 = time.perf_counter()
    V = pkg.surrogates.eval_container_objectives_at_scaled_sites(sc, None, X)
    calls["t"] += time.perf_counter() - t0; calls["n"] += 1; calls["pts"] += X.shape[0]
    return V
x = np.full(d, 0.5); x[1] = 0.85; x[3] = 0.2
lb, ub = x - 0.1, x + 0.1
fx = ev(x[None, :])[0]
for k in calls: calls[k] = 0
stats = {}
t0 = time.perf_counter()
omega, rest = ps.get_criticality(ps.PascolettiSerafiniConfig(), x, x, fx, lb, ub, ev, rng=np.random.default_rng(1), stats=stats)[:2]
wall = time.perf_counter() - t0
print(f"omega {omega:.4f}; {calls['pts']} surrogate evaluations in {calls['n']} batched calls; wall {wall:.3f} s "
      f"(device calls {calls['t']:.3f} s, host ranking/breeding {wall - calls['t']:.3f} s) -> {calls['pts'] / wall:.0f} evaluations/s")
# one-point calls, as NLopt's callbacks would issue them (k closures per candidate share one sweep here: lower bound on their cost)
m = 300
t0 = time.perf_counter()
for j in range(m):
    ev(x[None, :] + 1e-3 * j)
t1 = time.perf_counter() - t0
print(f"one-point calls: {m / t1:.0f} evaluations/s -> batched / one-point = {calls['pts'] / wall / (m / t1):.1f}x")
# the same step with the subproblem solver on the device (mrbf_ps_step): population state, ranking and breeding in device memory
for rep in range(3):
    st = {}
    t0 = time.perf_counter()
    om = ps.get_criticality_device(ps.PascolettiSerafiniConfig(), mod, x, x, fx, lb, ub, seed=rep, stats=st)[0]
    wall_d = time.perf_counter() - t0
    ne = st["evals_ideal"] + st["evals_ps"]
    print(f"device solver: omega {om:.4f}; {ne} evaluations in {st['generations']} generations; wall {wall_d * 1e3:.2f} ms (device events {st['ms_total']:.2f} ms) "
          f"-> {ne / wall_d:.0f} evaluations/s")
