"""`python bench.py --gpus N` must itself start N ranks: the driver's N > 1 line is only as good as
the launcher. No GPU here, so the ranks run bench.py's --dry-run leg: gloo instead of RCCL, each rank contributes its
sample COUNT instead of a render, and rank 0 checks that the reduced sum is the image's spp. What is exercised is the
real launcher (child processes, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*), the real split and the reduce call order."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr)  # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_flag_starts_that_many_ranks_strong_split():
    p = _run("--gpus", "2", "--dry-run", "--spp", "1024")
    assert p.returncode == 0, p.stderr
    d = _json_line(p)
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["scaling"] == "strong"
    assert d["spp_total"] == 1024 and d["sample_ranges"] == [[0, 512], [512, 512]]


def test_uneven_split_and_weak_scaling():
    d = _json_line(_run("--gpus", "3", "--dry-run", "--spp", "7"))
    assert d["n_gpus"] == 3 and d["sample_ranges"] == [[0, 3], [3, 2], [5, 2]]
    d = _json_line(_run("--gpus", "2", "--dry-run", "--spp", "8", "--scaling", "weak"))
    assert d["spp_total"] == 16 and d["sample_ranges"] == [[0, 8], [8, 8]]


def test_single_rank_needs_no_launcher():
    d = _json_line(_run("--dry-run"))
    assert d["n_gpus"] == 1 and d["sample_ranges"] == [[0, 1024]]


def test_ranks_given_by_the_environment_are_used_as_they_are():
    # the driver's N > 1 form: torch.distributed.run has already set the rank variables — no second level of children
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}
    d = _json_line(_run("--gpus", "1", "--dry-run", env_extra=env))
    assert d["n_gpus"] == 1


def test_failing_rank_fails_the_launcher():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs: the ranks must fail")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")  # no --dry-run: a rank without a GPU exits non-zero
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
