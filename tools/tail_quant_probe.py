"""Does the last, partly filled round of workgroups cost the dominant kernel time?  cfg2's 960x960 layer at B = 100 000 is
782 column panels x 10 row tiles = 7820 workgroups on 512 slots (two per CU) = 15.27 rounds.  Runs the cfg2 chain at batch
sizes whose tile counts are whole rounds and at ones just above, and prints time per tile.
usage: python3 tools/tail_quant_probe.py [B ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):   # the development build (python subspaceinference.jl_amd/build.py --dev): SI_GEMM_NO_NARROW=1 A/B
    si._capi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin",
                                     "libsubspace_hip_dev.so")
if os.environ.get("SI_PROBE_LIB"):   # any other build of the library (A/B against an older tree)
    si._capi.LIB_PATH = os.path.abspath(os.environ["SI_PROBE_LIB"])
sizes = [int(v) for v in sys.argv[1:]] or [98304, 100000, 100352, 104448, 131072, 65536, 66816]

dims, acts, m = [128, 960, 960, 1], [1, 1, 0], 20
table, off = [], 0
for fin, fout, act in zip(dims[:-1], dims[1:], acts):
    table.append((fin, fout, act, off, off + fin * fout))
    off += fin * fout + fout
rng = np.random.default_rng(0)
w_swa = 0.03 * rng.standard_normal(off)
p = np.asfortranarray(0.01 * rng.standard_normal((off, m)))
for b in sizes:
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((1, b)))
    with si.Context(0) as ctx:
        ctx.infer_setup(table, off, m, w_swa, p, x, y, 1.0)
        ctx.sample_rwmh(20, 0.1, seed=1)
        ctx.set_profiling(True, ["dense"])
        ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.sample_rwmh(60, 0.1, seed=2)
        wall = (time.perf_counter() - t0) / 60 * 1e3
        st = ctx.stats()["dense"]
        tiles = (b + 127) // 128 * 10
        ms = st["ms"] / 60
        print("B %7d  tiles %5d = %6.2f rounds  dense %.4f ms/step  %.2f TFLOP/s  %.4f us per tile-slot  step %.4f ms"
              % (b, tiles, tiles / 512, ms, st["flops"] / 60 / ms / 1e9, ms * 1e3 / (tiles / 512), wall), flush=True)
