"""`python bench.py --gpus N` must start its N ranks by itself (the driver calls it that way).  The --dry leg
runs the whole launcher path - child torch.distributed.run, rendezvous on 127.0.0.1, one collective, the
relay of rank 0's line - over gloo, so it can be checked without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=300)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


@pytest.mark.parametrize("n", [2, 4])
def test_bench_starts_its_own_ranks(n):
    rc, out, err = _run("--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry")
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["ranks_seen"] == n and line["max_rank_plus_one"] == float(n)
    assert line["steps"] == 3 and line["warmup"] == 1 and line["dry"] is True


def test_single_rank_needs_no_launcher():
    rc, out, err = _run("--gpus", "1", "--dry")
    assert rc == 0, err[-2000:]
    line = json.loads(out.strip())
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1


def test_a_failing_rank_fails_the_launcher():
    # a rank count that contradicts the environment it is started in must not look like success
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode != 0 and not p.stdout.strip()
