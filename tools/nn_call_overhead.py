"""where the wall clock of one si_sample_rwmh call goes beyond its one kernel (nn_example model, one chain)"""
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
ctx.infer_setup(table, off, m, 0.3 * rng.standard_normal(off), 0.05 * rng.standard_normal((off, m)), rng.standard_normal((2, b)), rng.standard_normal((1, b)), 1.0)
ctx.sample_rwmh(20, 0.1, seed=1)
for itr in (2000, 20000, 100000):
    for want_z in (True, False):
        t0 = time.perf_counter()
        ctx.sample_rwmh(itr, 0.1, seed=1, want_z=want_z)
        dt = time.perf_counter() - t0
        print("itr %6d want_z %5s: %.2f us per transition (%.1f ms)" % (itr, want_z, dt / itr * 1e6, dt * 1e3), flush=True)
ctx.close()
