"""The n_dev > 1 branch of mrbf_batch_run, rehearsed on the one GPU there is (run with -m gpu).

mrbf_batch_run(n_dev, device_ids, ...) is the in-library twin of the reference's `Threads.@threads` loop over independent
problems (/root/reference/examples/large_scale_benchmarks.jl:253, Halton starts :102-109): problem p goes to device
device_ids[p % n_dev], one host thread + pooled context per device (csrc/batch.hip, csrc/api.hip).  device_ids = {0, 0, ...} is
accepted, so everything but the physical second GPU runs here: the p % n_dev dealing, one worker thread and one context per
"device" working on the card at the same time, the re-deal of problems the one-launch fit hands back to the per-problem chain,
per-record statuses, the context pool.

Asserted: every result bit-identical to the n_dev = 1 call (host and device buffers), the record's device as dealt, a singular
problem and an invalid one surface as a status on their own records and nowhere else, and a repeated call allocates nothing.
"""
import ctypes

import numpy as np
import pytest

from tests.conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import morbit.jl_amd as pkg
    from morbit.jl_amd import _lib
from morbit.jl_amd import workloads as wl
from oracle import rbf_oracle as orc


def _synthetic(n, d, k, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    C = rng.random((n, d))
    Y = np.stack([((C - 1.0) ** 2).sum(axis=1), ((C + 1.0) ** 2).sum(axis=1), np.sin(C.sum(axis=1))][:k], axis=1) / d
    return C, Y


MIXED = [  # kernel, deg, n, d, k, m, want_jac -- the list of test_batch_run_mixed_shapes_matches_single_calls
    ("cubic", 1, 150, 6, 2, 20, True), ("multiquadric", 1, 257, 100, 2, 70, True), ("cubic", 1, 90, 6, 1, 33, False),
    ("gaussian", -1, 200, 12, 3, 40, True), ("multiquadric", 1, 300, 100, 2, 64, False), ("cubic", 1, 160, 6, 2, 0, False),
    ("inv_multiquadric", 0, 120, 70, 2, 25, True), ("cubic", 1, 700, 10, 2, 30, True), ("cubic", 1, 140, 6, 2, 50, True)]


def _problem_set(n_c4):
    """(kid, a, b, deg, C, Y, X or None, want_jac) per problem: the nine mixed problems, n_c4 of the C4 starts, then -- spread
    over different residues mod 2 and mod 8 -- a singular problem (all sites identical, cubic without tail: Phi = 0 exactly, the LU
    meets a zero pivot), an invalid one (kernel id 9) and a problem beyond the small shape that needs the per-problem chain"""
    out = []
    for p, (kernel, deg, n, d, k, m, wj) in enumerate(MIXED):
        cfg = pkg.RbfConfig(kernel=kernel, polynomial_degree=deg)
        kid, a, b = pkg.rbf_model._get_kernel_params(1.0, cfg)
        C, Y = _synthetic(n, d, k, seed=500 + p)
        X = np.random.Generator(np.random.PCG64(600 + p)).random((m, d)) if m > 0 else None
        out.append((kid, a, b, deg, C, Y, X, wj))
    kid, a, b = pkg.rbf_model._get_kernel_params(1.0, pkg.RbfConfig(kernel="cubic", polynomial_degree=1))
    for p in range(n_c4):
        C, Y, X = wl.problem("C4", p)
        out.append((kid, a, b, 1, C, Y, X, p in (0, 17)))
    special = {}
    Cs = np.tile(np.random.Generator(np.random.PCG64(1)).random((1, 3)), (20, 1))
    special["singular"] = len(out)
    out.append((0, 3.0, 0.0, -1, Cs, np.arange(20.0).reshape(20, 1), Cs[:4].copy(), False))
    C, Y = _synthetic(60, 4, 1, seed=3)
    special["invalid"] = len(out)
    out.append((9, 1.0, 0.0, 1, C, Y, C[:5].copy(), False))
    C, Y = _synthetic(1100, 16, 2, seed=4)
    special["chain"] = len(out)
    out.append((2, 1.0, 0.5, 1, C, Y, np.random.Generator(np.random.PCG64(5)).random((50, 16)), True))
    return out, special


class _Run:
    """one mrbf_batch_run over the problem set; outputs in fresh buffers (host arrays, or torch tensors on the card)"""

    def __init__(self, probs, n_dev, on_device=False):
        import torch

        P = len(probs)
        self.arr = (_lib.Problem * P)()
        self.res = (_lib.Result * P)()
        self.keep, self.out = [], []
        for p, (kid, a, b, deg, C, Y, X, wj) in enumerate(probs):
            n, d = C.shape
            k = Y.shape[1]
            m = 0 if X is None else X.shape[0]
            q = orc.poly_dim(d, deg)
            if on_device:
                mk = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float64, device="cuda")
                tin = [torch.from_numpy(np.ascontiguousarray(t)).cuda() if t is not None else None for t in (C, Y, X)]
                W, L, V = mk(n, k), mk(max(q, 1), k), (mk(m, k) if m else None)
                J = mk(m, d, k) if (m and wj) else None
                ptr = lambda t: ctypes.cast(t.data_ptr(), _lib.c_dp) if t is not None else None
            else:
                tin = [np.ascontiguousarray(t) if t is not None else None for t in (C, Y, X)]
                mk = lambda *shape: np.full(shape, np.nan)
                W, L, V = mk(n, k), mk(max(q, 1), k), (mk(m, k) if m else None)
                J = mk(m, d, k) if (m and wj) else None
                ptr = lambda t: t.ctypes.data_as(_lib.c_dp) if t is not None else None
            self.keep.append(tin)
            self.out.append((W, L, V, J))
            self.arr[p] = _lib.Problem(n, m, d, k, kid, deg, a, b, ptr(tin[0]), ptr(tin[1]), ptr(tin[2]), ptr(W), ptr(L), ptr(V), ptr(J))
        self.P, self.n_dev, self.on_device = P, n_dev, on_device
        self.ids = (ctypes.c_int32 * n_dev)(*([0] * n_dev))

    def __call__(self):
        import torch

        rc = _lib.load().mrbf_batch_run(self.n_dev, self.ids, self.P, self.arr, self.res)
        if self.on_device:
            torch.cuda.synchronize()
        return rc

    def host(self, p):
        return [None if t is None else (t.cpu().numpy() if self.on_device else t) for t in self.out[p]]


def _same(a, b):
    return (a is None and b is None) or np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("n_c4", [64])
def test_batch_run_on_several_devices_equals_one_device(n_c4):
    import torch

    probs, special = _problem_set(n_c4)
    P = len(probs)
    ref = _Run(probs, 1)
    assert ref() == 0
    bad = {special["singular"]: _lib.MRBF_ESINGULAR, special["invalid"]: None}
    for p in range(P):
        st = ref.res[p].status
        if p in bad:
            assert st != 0 and (bad[p] is None or st == bad[p]), (p, st)
            if bad[p] is None:
                assert st < 0, st               # invalid argument: negative code
        else:
            assert st == 0, (p, st)
    assert ref.res[special["chain"]].fit.path == _lib.PATH_PROJ_CHOL and ref.res[special["chain"]].fit.n == 1100
    for n_dev in (2, 8):
        for on_device in (False, True):
            run = _Run(probs, n_dev, on_device)
            assert run() == 0, (n_dev, on_device)
            for p in range(P):
                r, r0 = run.res[p], ref.res[p]
                # a failing problem shows on its own record, and nowhere else
                assert r.status == r0.status, (n_dev, on_device, p, r.status, r0.status)
                assert r.device == run.ids[p % n_dev] == 0
                if r0.status != 0:
                    continue
                assert r.fit.path == r0.fit.path and r.fit.fallbacks == r0.fit.fallbacks and r.fit.n == r0.fit.n and r.fit.q == r0.fit.q
                for got, want, what in zip(run.host(p), ref.host(p), "WLVJ"):
                    if what == "L" and probs[p][3] < 0:
                        continue                # no tail: nothing is written
                    assert _same(got, want), (n_dev, on_device, p, what)
                assert r.checksum_w == r0.checksum_w and r.checksum_vals == r0.checksum_vals, (n_dev, p)
            # pooled contexts are returned and re-used: in steady state a call allocates nothing on the card (which pooled context
            # a worker thread draws is a race, so a context may still grow its arena to the largest shard on an early repeat: the
            # free-memory reading has to stand still within a few calls, and then stay)
            torch.cuda.synchronize()
            frees = []
            for _ in range(12):
                assert run() == 0
                frees.append(torch.cuda.mem_get_info()[0])
                if len(frees) >= 3 and frees[-1] == frees[-2] == frees[-3]:
                    break
            assert len(frees) < 12, (n_dev, on_device, frees)
            for p in (0, 9, special["chain"]):
                for got, want in zip(run.host(p), ref.host(p)):
                    assert _same(got, want) or probs[p][3] < 0
    # a device id the process cannot see is refused as argument 2, before any work
    two = (ctypes.c_int32 * 2)(0, torch.cuda.device_count())
    assert _lib.load().mrbf_batch_run(2, two, P, ref.arr, ref.res) == -2
