"""Do two independent chains on ONE GPU (two contexts = two HIP streams, driven from two host threads) overlap the
store-bound first layer of one with the MFMA-bound second layer of the other?  cfg2 workload."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

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
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctxs = [si.Context(0) for _ in range(nctx)]
for c in ctxs:
    c.infer_setup(table, off, m, w_swa, p, x, y, 1.0)
    c.sample_rwmh(5, 0.1, seed=1)
itr = 100


def run(c, cid):
    c.sample_rwmh(itr, 0.1, seed=1, chain_id0=cid)


for k in (1, nctx):
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(ctxs[i], i)) for i in range(k)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("%d concurrent chain(s): %.1f samples/s total (%.3f ms per transition per chain)" % (k, k * itr / dt, dt / itr * 1e3))
