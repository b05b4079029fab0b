"""CPU suite: `bench.py --gpus N` starts N ranks by itself, the ranks join one process group and rank 0 prints one
line for the whole job.  No GPU here, so the ranks run the launcher's dry run (gloo) or stop with the clean
"needs a GPU" message."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(*argv, timeout=300, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env.update(extra_env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("gpus", [1, 2, 3])
def test_bench_starts_its_own_ranks(gpus):
    r = _run("--gpus", str(gpus), "--dry-run", "--queries", "1024")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # one line for the whole job, from rank 0
    line = lines[0]
    assert line["n_gpus"] == gpus and line["ranks_joined"] == gpus
    assert line["scaling"] == "weak" and line["queries_total"] == 1024.0 * gpus  # every rank its own 1024 queries


def test_bench_strong_scaling_splits_one_batch():
    r = _run("--gpus", "2", "--dry-run", "--scaling", "strong", "--queries", "65537")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["queries_total"] == 65537.0


def test_bench_without_gpu_fails_on_every_rank():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "ranks failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no line without a measurement


def test_one_failing_rank_ends_the_others():
    """Rank 1 exits before the rendezvous; rank 0 would wait in it for minutes.  The launcher must notice, stop rank 0
    and report both: well inside the rendezvous timeout, and without a line on stdout."""
    import time

    t0 = time.monotonic()
    r = _run("--gpus", "2", "--dry-run", timeout=120, HYAS_DRY_RUN_FAIL_RANK="1")
    assert time.monotonic() - t0 < 60
    assert r.returncode != 0 and "ranks failed" in r.stderr
    assert "(1, 3)" in r.stderr  # rank 1's own exit code; rank 0 was terminated (negative code)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
