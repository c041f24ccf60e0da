"""CPU rehearsal of bench.py's own launcher (no GPU): `python bench.py --gpus 2` must start two ranks itself, shard the problems
p % world, gather the records of the many-start workloads and print ONE JSON line that says n_gpus = 2 -- never a 1-GPU number
for an N-GPU request (ADVICE bench.py:124)."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT


def _run(*flags, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=300, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_spawns_its_own_ranks_dry_run():
    r, lines = _run("--gpus", "2", "--dry-run", "--config", "C4", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["config"]["problems_per_step"] == 64 and out["scaling"] == "strong"
    assert out["records"] == {"gathered": 64, "failed": 0, "worst_rel_residual": 0.0}
    assert len(out["rank_seconds"]) == 2


def test_bench_default_workload_is_c3_weak_scaling():
    r, lines = _run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["problems_per_step"] == 2
    assert out["config"]["n"] == 8192 and out["config"]["d"] == 64 and out["config"]["m"] == 10000


def test_bench_refuses_a_mismatched_world():
    # under an external launcher with the wrong rank count the script must fail, not report a smaller job
    r, lines = _run("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not lines
    assert "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_needs_a_gpu_for_real_runs():
    import torch

    if torch.cuda.is_available():
        return
    r, lines = _run("--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and not lines
    assert "GPU" in r.stderr
