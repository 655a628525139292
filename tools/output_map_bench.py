"""Row a13 (the reference materialises every weight sample on the host, space_inference.jl:125): time si_reconstruct at the
cfg2 size for 1, 8, 64 and 256 samples into a fresh (never touched) and into an already touched NumPy array."""
import os
import sys
import time

# NumPy's BLAS would start one spinning thread per visible CPU (256 on the GPU box) for the reference product below and burn
# the container's CPU quota: the cgroup then throttles the whole process for ~80 ms at a time, inside the timed calls
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402
from subspaceinference_jl_amd import _capi  # noqa: E402

N, M = 1047361, 20
if os.environ.get("TORCH_INIT"):   # development: does an initialised torch runtime in the process change the copy-out?
    import torch
    torch.zeros(1, device="cuda")
BX = int(os.environ.get("BX", "64"))
rng = np.random.default_rng(0)
ctx = si.Context(0)
table = [(128, 960, 1, 0, 128 * 960), (960, 960, 1, 128 * 960 + 960, 128 * 960 + 960 + 960 * 960),
         (960, 1, 0, 128 * 960 + 960 + 960 * 960 + 960, 128 * 960 + 960 + 960 * 960 + 960 + 960)]
w = rng.standard_normal(N)
p = np.asfortranarray(rng.standard_normal((N, M)))
x = np.asfortranarray(rng.standard_normal((128, BX)))
y = np.asfortranarray(rng.standard_normal((1, BX)))
ctx.infer_setup(table, N, M, w, p, x, y, 1.0)
if os.environ.get("SAMPLE"):       # development: a chain first, as bench.py does
    ctx.sample_rwmh(int(os.environ["SAMPLE"]), 0.1, seed=1)
if os.environ.get("PROF"):
    ctx.set_profiling(True)
    ctx.reset_stats()
    ctx.sample_rwmh(5, 0.1, seed=1)
    ctx.synchronize()
    ctx.stats()
    ctx.set_profiling(False)
for c in [int(a) for a in sys.argv[1:]] or (1, 8, 32, 64, 256):
    z = np.asfortranarray(rng.standard_normal((M, c)))
    ctx.reconstruct(z[:, :1])
    fresh, touched, ok = [], [], True
    for rep in range(int(os.environ.get('REPS', '5'))):
        t0 = time.perf_counter()
        out = ctx.reconstruct(z)
        fresh.append((time.perf_counter() - t0) / c * 1e3)
        ok = ok and np.allclose(out[:, -1], w + p @ z[:, -1], rtol=1e-12, atol=1e-12) and np.allclose(out[::9973, 0], (w + p @ z[:, 0])[::9973])
        t0 = time.perf_counter()
        ctx.reconstruct(z, out=out)      # the same array again: its pages exist now
        touched.append((time.perf_counter() - t0) / c * 1e3)
        del out
    fmt = lambda v: " ".join("%.3f" % t for t in v)
    print("C = %3d samples, ms per sample: fresh array [%s]  touched array [%s]  (8.38 MB per sample; correct: %s)" % (c, fmt(fresh), fmt(touched), ok))
