"""world_size-2 gloo test (CPU) of the many-problem sharding + record gather.  The per-problem solver is
injected (an oracle-backed stand-in) so the test needs no GPU; the sharding / gather code is the product's."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problems(P):
    out = []
    for p in range(P):
        rng = np.random.default_rng(1000 + p)
        C = rng.random((12 + p, 3))
        out.append(dict(id=p, sites=C, values=(C ** 2).sum(axis=1, keepdims=True), X=rng.random((4, 3))))
    return out


def _oracle_solver(problem):
    from oracle import rbf_oracle as orc

    mod = orc.fit(problem["sites"], problem["values"], 0, 3.0, 0.0, 1)
    if problem["id"] == 3:
        raise RuntimeError("injected failure")  # must surface as a status, not kill the batch
    return [float(problem["id"]), 0.0, 3.0, orc.rel_residual(mod, problem["values"]), float(mod.w.sum()),
            float(mod.values(problem["X"]).sum()), 0.0, 0.0]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from morbit.jl_amd import manystart

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    probs = _problems(7)
    table = manystart.run_manystart(probs, rank, world, device="cpu", solve=_oracle_solver)
    q.put((rank, manystart.shard_indices(7, rank, world), table))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    from morbit.jl_amd import manystart

    assert manystart.shard_indices(7, 0, 2) == [0, 2, 4, 6] and manystart.shard_indices(7, 1, 2) == [1, 3, 5]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = manystart.run_manystart(_problems(7), 0, 1, solve=_oracle_solver)
    for rank, shard, table in got:
        assert table.shape == (7, manystart.RECORD_LEN)
        assert np.array_equal(table[:, 0], np.arange(7.0))          # every rank holds the full table, ordered by id
        ok = table[:, 1] == 0
        assert list(np.where(~ok)[0]) == [3]                        # the injected failure is a status
        assert np.array_equal(table[ok], single[ok])                # identical to the unsharded run
