"""RWMH transitions of the cfg4 CNN alone at B = 4096 (for rocprofv3 --kernel-trace --stats): nothing else on the device
apart from the set-up, so the per-kernel averages are per-layer times at ONE batch size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["CFG4_SETUP_ONLY"] = "1"
import cfg4_cnn_bench as m  # noqa: E402

m.ctx.sample_rwmh(3, 0.01, seed=1)
m.ctx.sample_rwmh(int(os.environ.get("STEPS", "20")), 0.01, seed=2)
