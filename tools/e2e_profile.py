"""Where the end-to-end drop-in call of bench.py (subspace_construction with the training step on the device, cfg2, T = 100)
spends its wall-clock beyond the 100 training steps: cProfile of the call, top entries by cumulative time."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402
from subspaceinference_jl_amd import flux  # noqa: E402

DIMS, ACTS, B, M, T = [128, 960, 960, 1], [flux.relu, flux.relu, flux.identity], 100000, 20, 100
rng = np.random.default_rng(0)
x = np.asfortranarray(rng.standard_normal((DIMS[0], B)))
y = np.asfortranarray(rng.standard_normal((1, B)))


def run():
    wr = np.random.default_rng(1)
    mdl = flux.Chain(*[flux.Dense(i, o, a, rng=wr) for i, o, a in zip(DIMS[:-1], DIMS[1:], ACTS)])
    data = flux.DataLoader(x, y, batchsize=B)
    with si.Context(0) as solo:
        t0 = time.perf_counter()
        si.subspace_construction(mdl, flux.mse, data, flux.ADAM(1e-3), T=T, c=1, M=M, ctx=solo, verbose=False,
                                 device_training=True, keep_on_device=True, data_parallel=False)
        solo.synchronize()
        return (time.perf_counter() - t0) * 1e3


print("warm-up run: %.1f ms" % run())
print("second run : %.1f ms" % run())
pr = cProfile.Profile()
pr.enable()
ms = run()
pr.disable()
print("profiled run: %.1f ms" % ms)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
