"""does the device-memory placement of the loop's buffers move its time?  (nn_example model, one chain, 20 000 transitions; a
construction of varying size is allocated first, which shifts every later allocation)"""
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
args = (table, off, m, 0.3 * rng.standard_normal(off), 0.05 * rng.standard_normal((off, m)), rng.standard_normal((2, b)), rng.standard_normal((1, b)), 1.0)
for shift in (0, 1000, 37001, 300007, 2000003, 9000001):
    ctx = si.Context(0)
    if shift:
        ctx.construct_begin(shift, 3)
        ctx.construct_push(np.zeros(shift, dtype=np.float32), 1.0)
    ctx.infer_setup(*args)
    ctx.sample_rwmh(20, 0.1, seed=1)
    t0 = time.perf_counter()
    ctx.sample_rwmh(20000, 0.1, seed=1)
    print("shift %8d: %.2f us per transition" % (shift, (time.perf_counter() - t0) / 20000 * 1e6), flush=True)
    ctx.close()
