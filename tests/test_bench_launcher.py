"""bench.py's own N-rank launcher (`python bench.py --gpus N` without torchrun used to benchmark ONE GPU):
exercised on CPU with the dry-run ranks (gloo group, ranks counted by an all-reduce of ones, no compute)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*extra, env=None):
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, BENCH, "--dry-run-launcher", *extra], capture_output=True, text=True, timeout=280, env=e)


@pytest.mark.timeout(300)
def test_launcher_starts_n_ranks_and_relays_one_line():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry_run"] is True
    # every rank's host copy pool is told its share of the CPU quota: budget // (2 * ranks), at least one thread
    assert out["host_copy_threads_env"] == str(max(1, out["cpu_budget"] // 4))


@pytest.mark.timeout(300)
def test_a_failed_rank_is_a_failed_run():
    r = _run("--gpus", "2", "--fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(120)
def test_single_rank_needs_no_launcher_and_world_mismatch_is_an_error():
    r = _run("--gpus", "1")
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 1
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--dry-run-launcher", "--gpus", "4"], capture_output=True, text=True, timeout=100, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in r.stderr


def test_parent_imports_neither_torch_nor_the_hip_library():
    code = ("import sys, types; sys.path.insert(0, %r); import bench, subprocess\n"
            "subprocess.run = lambda *a, **k: types.SimpleNamespace(returncode=0, stdout='{\"n_gpus\": 2}\\n')\n"
            "rc = bench.launch_ranks(bench.parse_args(['--gpus', '2']), ['--gpus', '2'])\n"
            "assert rc == 0 and 'torch' not in sys.modules and 'subspaceinference_jl_amd' not in sys.modules and 'numpy' not in sys.modules\n"
            % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip()) == {"n_gpus": 2}
