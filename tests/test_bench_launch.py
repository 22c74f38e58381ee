"""`python bench.py --gpus N` starts N ranks itself (no external launcher) and refuses a launcher whose world size
disagrees with --gpus. CPU: the --dry-run leg does the rendezvous and the one all-gather over gloo, no decode."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_starts_two_ranks_and_all_gathers():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, res.stdout          # rank 0 only
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["tokens"] == 3 and out["dry_run"] is True


def test_gpus_8_dry_run_rehearses_the_node_sized_job():
    """The driver's 8-GPU run, rehearsed on the CPU: 8 ranks rendezvous on 127.0.0.1, one all-gather of the per-rank struct,
    rank 0 reports all eight — with every rank's wall time, NUMA node and CPU count (each rank keeps >= 1 CPU)."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run"], env=_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, res.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 8 and out["tokens"] == sum(range(1, 9))
    r = out["ranks"]
    assert len(r["numa_node"]) == 8 and r["numa_node"] == [i % 2 for i in range(8)]
    assert all(c >= 1 for c in r["cpus"]) and r["ms_per_step_min"] <= r["ms_per_step_max"]


def test_gpus_1_runs_in_process():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], env=_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    assert json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_must_equal_gpus():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr
