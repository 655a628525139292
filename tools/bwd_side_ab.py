"""cfg2-sized value + gradient (si_logdensity_grad) and full-batch training step: wall time per call.
usage: python3 tools/bwd_side_ab.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):
    si._capi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin",
                                     "libsubspace_hip_dev.so")
dims, acts, b, m = [128, 960, 960, 1], [1, 1, 0], 100000, 20
table, off = [], 0
for fin, fout, act in zip(dims[:-1], dims[1:], acts):
    table.append((fin, fout, act, off, off + fin * fout))
    off += fin * fout + fout
rng = np.random.default_rng(0)
w_swa = 0.03 * rng.standard_normal(off)
p = np.asfortranarray(0.01 * rng.standard_normal((off, m)))
x = np.asfortranarray(rng.standard_normal((dims[0], b)))
y = np.asfortranarray(rng.standard_normal((1, b)))
with si.Context(0) as ctx:
    ctx.infer_setup(table, off, m, w_swa, p, x, y, 1.0)
    z = 0.1 * rng.standard_normal(m)
    for _ in range(3):
        lp, g = ctx.logdensity_grad(z)
    t0 = time.perf_counter()
    for _ in range(10):
        lp, g = ctx.logdensity_grad(z)
    t_grad = (time.perf_counter() - t0) / 10 * 1e3
    ctx.train_setup(table, off, w_swa.astype(np.float32), x, y, b, 2, 0.001, 0.9, 0.999)
    ids = np.arange(b)
    ctx.train_step(ids)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.train_step(ids, want_loss=False)
    ctx.synchronize()
    t_train = (time.perf_counter() - t0) / 10 * 1e3
    print("logdensity_grad %.3f ms   train_step %.3f ms   lp %.12g  g0 %.12g  w[7] %.9g"
          % (t_grad, t_train, lp, g[0], ctx.train_get_weights()[7]), flush=True)
