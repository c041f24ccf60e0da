"""Many-start / many-problem mode: independent RBF build+solve+eval problems sharded over GPUs.

The reference parallelises exactly this way, on CPU threads: `Threads.@threads` over the rows of the
benchmark table (/root/reference/examples/large_scale_benchmarks.jl:253), Halton start points (:102-109).
Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), problems dealt
round-robin, NO data-path collective (no problem is split across GPUs); one all_gather of small
fixed-size result records at the end of the batch (latency-bound, tens of KB).
"""
import numpy as np

RECORD_FIELDS = ("problem_id", "status", "path", "rel_residual", "checksum_w", "checksum_vals", "ms_fit", "ms_eval")
RECORD_LEN = len(RECORD_FIELDS)


def shard_indices(n_problems, rank, world_size):
    """round-robin: problem p runs on rank p % world_size"""
    return list(range(rank, n_problems, world_size))


def gpu_solve_problem(problem, ctx=None):
    """fit + eval of one problem dict on this rank's GPU through the C ABI -> record (list of floats)"""
    from . import rbf_model as rm

    cfg = problem["cfg"]
    mod = rm.update_model(cfg, problem["sites"], problem["values"], problem.get("delta", 1.0), ctx=ctx)
    info = {}
    vals = jac = None
    if problem.get("X") is not None and len(problem["X"]):
        vals, jac = mod.eval_sites(problem["X"], want_values=True, want_jac=problem.get("want_jac", True), info=info)
    rec = [float(problem["id"]), 0.0, float(mod.info["path"]), float(mod.info["rel_residual"]),
           float(np.sum(mod.weights)), float(np.sum(vals)) if vals is not None else 0.0,
           float(mod.info["ms_total"]), float(info.get("ms_total", 0.0))]
    mod.free()
    return rec


def run_local(problems, rank, world_size, solve=gpu_solve_problem, **kw):
    """solve this rank's shard; returns an (n_local x RECORD_LEN) float64 array"""
    recs = []
    for p in shard_indices(len(problems), rank, world_size):
        try:
            recs.append(solve(problems[p], **kw))
        except Exception as e:  # a failed factorisation must surface as a status, never take the batch down
            code = float(getattr(e, "code", -999))
            recs.append([float(problems[p]["id"]), code if code != 0 else -999.0] + [float("nan")] * (RECORD_LEN - 2))
    return np.asarray(recs, dtype=np.float64).reshape(-1, RECORD_LEN)


def gather_records(local, n_problems, device="cpu", group=None):
    """all_gather the per-rank record tables -> (n_problems x RECORD_LEN) table ordered by problem id on every rank.
    Works on any initialised torch.distributed backend (nccl on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        table = local
    else:
        world = dist.get_world_size(group)
        cap = (n_problems + world - 1) // world
        buf = torch.full((cap, RECORD_LEN), float("nan"), dtype=torch.float64, device=device)
        if len(local):
            buf[: len(local)] = torch.as_tensor(local, dtype=torch.float64, device=device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf, group=group)
        table = torch.cat(out).cpu().numpy()
        table = table[~np.isnan(table[:, 0])]
    order = np.argsort(table[:, 0], kind="stable")
    return table[order]


def run_manystart(problems, rank=0, world_size=1, device="cpu", solve=gpu_solve_problem, group=None, **kw):
    local = run_local(problems, rank, world_size, solve=solve, **kw)
    return gather_records(local, len(problems), device=device, group=group)
