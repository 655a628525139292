"""docs/src/nn_example.md's MLP (2-200-50-50-50-1, B = 1000, M = 20) under the three sampler paths: one launch per layer
(mode 0), the one-launch density stacked over chains (mode 2), the persistent grid loop where it applies (mode 1).
Usage: nn_example_modes.py [modes e.g. 012] [chains e.g. 1,8,64,512]   -- a target for rocprofv3 --kernel-trace as well."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

modes = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "021")]
chains = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "1,8,64,512").split(",")]
dims, acts, b, m = [2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 20
table, off = [], 0
for fin, fout, act in zip(dims[:-1], dims[1:], acts):
    table.append((fin, fout, act, off, off + fin * fout))
    off += fin * fout + fout
rng = np.random.default_rng(0)
ctx = si.Context(0)
ctx.infer_setup(table, off, m, 0.3 * rng.standard_normal(off), 0.05 * rng.standard_normal((off, m)),
                rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b)), 1.0)
flop = 2.0 * off * b
for nch in chains:
    itr = (20000 if nch == 1 else 2000) if nch <= 8 else (400 if nch <= 64 else 100)
    for mode in modes:
        ctx.set_chain_loop(mode)
        ctx.sample_rwmh(20, 0.1, seed=1, nchains=nch)
        t0 = time.perf_counter()
        z, lp, acc = ctx.sample_rwmh(itr, 0.1, seed=1, nchains=nch)
        dt = time.perf_counter() - t0
        print("mode %d %4d chains: %8.1f us per transition of all chains, %10.0f samples/s, %6.2f TFLOP/s, lp[-1]=%.6f  specialised (density, loop) %s" %
              (mode, nch, dt / itr * 1e6, itr * nch / dt, flop * nch * itr / dt / 1e12, lp[-1, 0], ctx.chain_kernel_info()[:2]), flush=True)
ctx.close()
