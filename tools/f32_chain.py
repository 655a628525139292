"""Target for rocprofv3 passes: a short cfg2 chain with compute_dtype = SI_F32 (or f64 with argv[1] == f64); nothing else."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from subspaceinference_jl_amd import _capi  # noqa: E402

DIMS, ACTS, B, M = [128, 960, 960, 1], [1, 1, 0], 100000, 20
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
table, off = [], 0
for fin, fout, act in zip(DIMS[:-1], DIMS[1:], ACTS):
    table.append((fin, fout, act, off, off + fin * fout))
    off += fin * fout + fout
n = off
rng = np.random.default_rng(0)
x = rng.standard_normal((DIMS[0], B))
y = rng.standard_normal((1, B))
w = np.concatenate([np.concatenate([(rng.uniform(-1, 1, (fo, fi)) * np.sqrt(6.0 / (fi + fo))).reshape(-1, order="F"), np.zeros(fo)])
                    for fi, fo in zip(DIMS[:-1], DIMS[1:])])
p = 0.01 * rng.standard_normal((n, M))
with si.Context(0) as ctx:
    ctx.infer_setup(table, n, M, w, p, x, y, 1.0, compute_dtype=_capi.SI_F32 if mode == "f32" else _capi.SI_F64)
    z, lp, acc = ctx.sample_rwmh(steps, 0.1, seed=1)
    print(mode, "lp[-1] = %.10e" % lp[-1, 0], "accept", acc[0])
