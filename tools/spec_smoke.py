"""run-time specialised narrow-chain kernels: one call of each path, with prints in between (a crash locator)"""
import faulthandler
import os
import sys

import numpy as np

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402
from oracle import subspace_oracle as so  # noqa: E402

dims, acts, b, m = [2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 20
table, n = so.layer_table(dims, acts)
rng = np.random.default_rng(0)
ctx = si.Context(0)
print("ctx", flush=True)
ctx.infer_setup(table, n, m, 0.3 * rng.standard_normal(n), 0.05 * rng.standard_normal((n, m)), rng.standard_normal((2, b)), rng.standard_normal((1, b)), 1.0)
print("setup", flush=True)
for mode in (4, 2, 3, 1):
    ctx.set_chain_loop(mode)
    lp = ctx.logdensity(np.zeros((m, 5)))
    print("mode", mode, "logdensity", lp[:2], ctx.chain_kernel_info(), flush=True)
    z, lpp, acc = ctx.sample_rwmh(50, 0.05, seed=1, nchains=2)
    print("mode", mode, "sample", lpp[-1], ctx.chain_kernel_info(), flush=True)
ctx.close()
print("done", flush=True)
