"""Per-step time of the RWMH loop for launch-bound (small) models: README toy and docs/src/nn_example.md MLP."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402


def table_of(dims, acts):
    t, off = [], 0
    for fin, fout, act in zip(dims[:-1], dims[1:], acts):
        t.append((fin, fout, act, off, off + fin * fout))
        off += fin * fout + fout
    return t, off


ctx = si.Context(0)
for name, dims, acts, b, m in (("README toy", [10, 20, 20, 2], [0, 0, 0], 100, 3),
                               ("nn_example MLP", [2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 20)):
    table, n = table_of(dims, acts)
    rng = np.random.default_rng(0)
    ctx.infer_setup(table, n, m, 0.3 * rng.standard_normal(n), 0.05 * rng.standard_normal((n, m)),
                    rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b)), 1.0)
    ctx.sample_rwmh(50, 0.1, seed=1)
    itr = 20000   # (a call carries ~2 ms of fixed cost: set-up of the loop, the copies of the samples, the first launch)
    t0 = time.perf_counter()
    z, lp, acc = ctx.sample_rwmh(itr, 0.1, seed=1)
    dt = time.perf_counter() - t0
    print("%-16s N=%6d B=%5d: %.1f us per step, %.0f samples/s, accept %.2f, lp[-1]=%.3f" % (name, n, b, dt / itr * 1e6, itr / dt, acc[0], lp[-1, 0]))
    # independent chains stacked in grid.y of every launch (capi_infer.hip eval_density / ChainBatch)
    for nch in (8, 64, 512):
        ctx.sample_rwmh(20, 0.1, seed=1, nchains=nch)
        itc = 500
        t0 = time.perf_counter()
        ctx.sample_rwmh(itc, 0.1, seed=1, nchains=nch)
        dt = time.perf_counter() - t0
        print("    %4d chains: %.1f us per transition of all chains, %.0f samples/s" % (nch, dt / itc * 1e6, itc * nch / dt))
