"""Random Dense chains through forward / log-density (chain-batched) / gradient / training gradient / construction, meant to
run over the DEVELOPMENT library with the guard-page allocator (csrc/guard_alloc.hip), where an out-of-bounds access of
any kernel faults at once.  The widths favour the tile edges of the kernels (96 / 128 / 64-row tiles, 16-deep k tiles).
usage: SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin python3 tools/guard_fuzz.py [cases] [seed]
Every case is printed BEFORE it runs (a fault names its shape); values are checked against the oracle as well."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from oracle import subspace_oracle as so  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):
    si._capi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libsubspace_hip_dev.so")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
WIDTHS = [1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 95, 96, 97, 100, 127, 128, 129, 160, 191, 192, 193, 200, 256, 288]
BATCH = [1, 2, 7, 63, 64, 65, 127, 128, 129, 255, 300, 511, 513, 1000, 2049]
ctx = si.Context(0)
for case in range(cases):
    nl = int(rng.integers(1, 5))
    dims = [int(rng.choice([1, 2, 3, 7, 12, 15, 16, 17, 20, 31, 33, 40]))] + [int(rng.choice(WIDTHS)) for _ in range(nl - 1)] + \
           [int(rng.choice([1, 1, 2, 3, 4, 5, 8, 17, 96]))]
    acts = [int(rng.integers(0, 8)) for _ in range(nl)]
    b = int(rng.choice(BATCH))
    m = int(rng.integers(1, 9))
    print("case %d dims %s acts %s B %d M %d" % (case, dims, acts, b, m), flush=True)
    table, n = so.layer_table(dims, acts)
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, m)))
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)))
    ctx.infer_setup(table, n, m, w_swa, p, x, y, 0.8)
    zs = np.asfortranarray(0.5 * rng.standard_normal((m, 3)))
    yref = so.forward(table, so.reconstruct(w_swa, p, zs[:, 0]), x)
    assert np.allclose(ctx.forward(zs[:, 0]), yref, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(yref).max()))
    lp_ref = np.array([so.logdensity(table, w_swa, p, x, y, 0.8, zs[:, j]) for j in range(3)])
    assert np.allclose(ctx.logdensity(zs), lp_ref, rtol=1e-9)
    lp, g = ctx.logdensity_grad(zs[:, 1])
    lpr, gr, _ = so.logdensity_grad(table, w_swa, p, x, y, 0.8, zs[:, 1])
    assert np.isclose(lp, lpr, rtol=1e-9) and np.allclose(g, gr, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(gr).max()))
    zc, lpc, _ = ctx.sample_rwmh(4, 0.1, seed=case, nchains=int(rng.integers(1, 4)))
    assert np.all(np.isfinite(lpc))
    # training step on a random batch (values: the step must leave finite weights; the gradient itself is compared in tests/)
    nb = int(rng.integers(1, b + 1))
    ctx.train_setup(table, n, w_swa.astype(np.float32), x, y, b, int(rng.integers(0, 3)), 0.01, 0.9, 0.999)
    ids = rng.choice(b, nb, replace=False).astype(np.int64)
    ctx.train_step(ids)
    assert np.all(np.isfinite(ctx.train_get_weights()))
    # construction on the same N: K pushes of fp32 / fp64 vectors, ragged K
    k = int(rng.integers(2, 40))
    mm = int(rng.integers(1, min(k, 8) + 1))
    ctx.construct_begin(n, k)
    dt = np.float32 if rng.random() < 0.7 else np.float64
    base = w_swa.copy()
    for j in range(k):
        base = base + 0.05 * rng.standard_normal(n)
        ctx.construct_push(base.astype(dt), float(1 + j))
    try:
        ctx.construct_finish(mm)
    except si.BoundsError:
        pass
print("guard_fuzz: %d cases done" % cases, flush=True)
ctx.close()
