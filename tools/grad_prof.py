"""Profile helper: cfg2-sized log-density gradient (forward + reverse sweep) a few times.  Usage:
   rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/grad_prof.py"""
import os
import sys

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
ctx = si.Context(0)
ctx.infer_setup(table, off, m, w_swa, p, x, y, 1.0)
z = 0.1 * rng.standard_normal(m)
for _ in range(6):
    lp, g = ctx.logdensity_grad(z)
print(lp, g[:3])
