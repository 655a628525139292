"""docs/src/nn_example.md's MLP (2-200-50-50-50-1, B = 1000, M = 20): 600 RWMH transitions, one chain -- a target for
rocprofv3 --kernel-trace (per-kernel durations against the wall-clock of a transition: how much is dispatch latency)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

dims, acts, b, m = [2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 20
table, off = [], 0
for fin, fout, act in zip(dims[:-1], dims[1:], acts):
    table.append((fin, fout, act, off, off + fin * fout))
    off += fin * fout + fout
rng = np.random.default_rng(0)
ctx = si.Context(0)
ctx.infer_setup(table, off, m, 0.3 * rng.standard_normal(off), 0.05 * rng.standard_normal((off, m)),
                rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b)), 1.0)
ctx.sample_rwmh(50, 0.1, seed=1)
t0 = time.perf_counter()
ctx.sample_rwmh(600, 0.1, seed=1)
print("%.1f us per transition (wall)" % ((time.perf_counter() - t0) / 600 * 1e6))
ctx.close()
