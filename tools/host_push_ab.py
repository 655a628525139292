"""cfg2's 100 host pushes (pageable Float32 vectors of 4.19 MB -> si_construct_push) + finish: wall time of the push phase.
usage: [SI_PROBE_LIB=<other build of the library>] python3 tools/host_push_ab.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402

if os.environ.get("SI_PROBE_LIB"):
    si._capi.LIB_PATH = os.path.abspath(os.environ["SI_PROBE_LIB"])
n, k, m = 1047361, 100, 20
rng = np.random.default_rng(0)
snaps = (0.03 * rng.standard_normal((k, n))).astype(np.float32)
snaps = np.cumsum(snaps, axis=0, dtype=np.float32)
with si.Context(0) as ctx:
    runs = []
    for rep in range(6):
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.construct_begin(n, k)
        for j in range(k):
            ctx.construct_push(snaps[j], float(j + 1))
        ctx.synchronize()
        tp = time.perf_counter() - t0
        s = ctx.construct_finish(m, want_swa=False, want_p=False)[2]
        runs.append(tp * 1e3)
    runs = runs[1:]
    print("host push phase: median %.2f ms (%.1f GB/s), runs %s, s[0] %.9g" % (
        float(np.median(runs)), k * n * 4 / (np.median(runs) * 1e-3) / 1e9, [round(r, 2) for r in runs], s[0]), flush=True)
