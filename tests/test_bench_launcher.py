"""bench.py's process handling, without a GPU: `--gpus N` must really produce an N-rank job (the parent starts the
ranks itself when no launcher did), must agree with WORLD_SIZE when a launcher did, and must fail when a rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, OMP_NUM_THREADS="1", **kw)
    return env


def test_world_resolution():
    assert bench.resolve_world(1, {}) == ("rank", 1)
    assert bench.resolve_world(8, {}) == ("launch", 8)
    assert bench.resolve_world(4, {"WORLD_SIZE": "4"}) == ("rank", 4)
    assert bench.resolve_world(1, {"WORLD_SIZE": "1"}) == ("rank", 1)
    with pytest.raises(SystemExit, match="must agree"):
        bench.resolve_world(8, {"WORLD_SIZE": "2"})
    with pytest.raises(SystemExit):
        bench.resolve_world(0, {})
    a = bench.parse_args(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert (a.gpus, a.steps, a.warmup, a.workload, a.precision) == (4, 3, 1, "view", "bf16")


@pytest.mark.timeout(600)
def test_gpus_flag_starts_that_many_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=_env(),
                         capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line == {"dry_run": True, "n_gpus": 2, "backend": "gloo", "gpus_arg": 2}


@pytest.mark.timeout(600)
def test_under_a_launcher_the_flag_must_match():
    env = _env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    ok = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], env=env,
                        capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert ok.returncode == 0 and json.loads(ok.stdout.strip().splitlines()[-1])["n_gpus"] == 1, ok.stdout + ok.stderr
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert bad.returncode != 0 and "must agree" in bad.stderr


@pytest.mark.timeout(600)
def test_a_failing_rank_fails_the_job():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU: there every rank of a real run refuses to start")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=_env(), capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode != 0 and "ranks failed" in out.stderr


@pytest.mark.timeout(120)
def test_one_dead_rank_stops_the_others_at_once():
    """Rank 1 exits before the rendezvous; rank 0 would wait in it for minutes.  The launcher has to notice, stop rank
    0 and return non-zero within seconds."""
    import time
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=_env(NERF_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=100, cwd=ROOT)
    took = time.time() - t0
    assert out.returncode != 0 and "ranks failed" in out.stderr and "(1, " in out.stderr, out.stdout + out.stderr
    assert took < 60, took
