"""The N-rank flow of bench.py on real GPU contexts (tools/bench_rehearsal.py: the ranks share the visible GPU over gloo -- a
rehearsal of what the driver starts on an 8-GPU node, everything but RCCL itself): the self-launcher, one rank per
process, barrier + max-over-ranks timing, exactly ONE JSON line from rank 0 with n_gpus = ranks counted by the backend."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", [["--mode", "chains"], ["--mode", "data-sharded"]])
def test_two_ranks_share_the_gpu(gpu_ctx, mode):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_rehearsal.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"] + mode, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["higher_is_better"] is True
    assert "rehearsal" in d and "roofline" in d
    if mode[1] == "chains":
        assert d["scaling"] == "weak" and d["unit"] == "samples/s"
