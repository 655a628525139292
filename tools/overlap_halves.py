"""one chain, the batch in two halves on two streams (SI_OVERLAP_HALVES=1) against the single-stream
transition at cfg2.  Prints ms per transition and checks that lp is bit-identical."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import subspaceinference_jl_amd as si
    dims, acts, b, m = [128, 960, 960, 1], [1, 1, 0], 100000, 20
    table, off = [], 0
    for fin, fout, act in zip(dims[:-1], dims[1:], acts):
        table.append((fin, fout, act, off, off + fin * fout))
        off += fin * fout + fout
    rng = np.random.default_rng(0)
    w = 0.05 * rng.standard_normal(off)
    p = np.asfortranarray(0.01 * rng.standard_normal((off, m)))
    x = np.asfortranarray(rng.standard_normal((128, b)))
    y = np.asfortranarray(rng.standard_normal((1, b)))
    ctx = si.Context(0)
    ctx.infer_setup(table, off, m, w, p, x, y, 1.0)
    ctx.sample_rwmh(10, 0.1, seed=1)
    ctx.synchronize()
    t0 = time.perf_counter()
    z, lp, acc = ctx.sample_rwmh(200, 0.1, seed=1)
    dt = (time.perf_counter() - t0) / 200
    np.save(sys.argv[2], lp)
    print("SI_OVERLAP_HALVES=%s: %.4f ms per transition, %.1f samples/s" % (os.environ.get("SI_OVERLAP_HALVES", "0"), dt * 1e3, 1 / dt))
else:
    for v in ("0", "1"):
        subprocess.run([sys.executable, __file__, "child", "/tmp/lp_%s.npy" % v], env=dict(os.environ, SI_OVERLAP_HALVES=v), check=True)
    a, b2 = np.load("/tmp/lp_0.npy"), np.load("/tmp/lp_1.npy")
    print("lp bit-identical:", bool(np.array_equal(a, b2)))
