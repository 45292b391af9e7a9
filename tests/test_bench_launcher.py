"""`python bench.py --gpus N` without a launcher around it starts its own ranks (the driver's N = 1
command shape with --gpus 8 died at an assert).  CPU only: the ranks form a gloo group and all-reduce, no GPU work."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True,
                         text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout  # ONE line, rank 0's, relayed by the parent
    got = json.loads(lines[0])
    assert got == {"launcher_selftest": True, "world_size": 2, "gpus_asked": 2, "sum_of_ranks_plus_one": 3}


@pytest.mark.timeout(120)
def test_a_rank_count_that_does_not_match_fails_loudly():
    """Under a launcher (WORLD_SIZE set) nothing is started; a world that is not --gpus is an error code, not a silent run."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True,
                         text=True, timeout=100)
    assert run.returncode == 5
