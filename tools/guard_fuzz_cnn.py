"""Random Conv / MaxPool / flatten / Dense chains through forward / log-density / gradient / training step, meant to run
over the DEVELOPMENT library with the guard-page allocator (see tools/guard_fuzz.py).
usage: SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin python3 tools/guard_fuzz_cnn.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from oracle import subspace_oracle as so  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):
    si._capi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libsubspace_hip_dev.so")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
osz = si._capi.conv_out_size


BIG = bool(os.environ.get("SI_FUZZ_BIG"))   # larger images and channel counts (cfg4-like layers at a reduced batch)


def random_spec():
    if BIG:
        w, h, c = int(rng.integers(12, 25)), int(rng.integers(12, 25)), int(rng.choice([3, 16]))
    else:
        w, h, c = int(rng.integers(3, 15)), int(rng.integers(3, 15)), int(rng.choice([1, 2, 3, 4, 16]))
    whc = (w, h, c)
    spec = []
    for _ in range(int(rng.integers(1, 4))):
        for _try in range(20):
            k = (int(rng.choice([1, 2, 3, 5])), int(rng.choice([1, 2, 3, 5])))
            s = (int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2])))
            pd = (int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 2])))
            d = (int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2])))
            wo, ho = osz(w, k[0], s[0], pd[0], d[0]), osz(h, k[1], s[1], pd[1], d[1])
            if wo >= 1 and ho >= 1:
                break
        else:
            break
        cout = int(rng.choice([32, 64, 96, 128])) if BIG else int(rng.choice([1, 2, 5, 16, 18, 32, 64, 70]))
        spec.append(("conv", k, cout, int(rng.integers(0, 8)), s, pd, d))
        w, h = wo, ho
        if rng.random() < 0.5:
            pk = (2, 2) if rng.random() < 0.7 else (3, 2)
            if w >= pk[0] and h >= pk[1]:
                spec.append(("maxpool", pk))
                w, h = osz(w, pk[0], pk[0], 0, 1), osz(h, pk[1], pk[1], 0, 1)
    spec.append(("flatten",))
    for _ in range(int(rng.integers(0, 2))):
        spec.append(("dense", int(rng.choice([3, 16, 33, 96])), int(rng.integers(0, 8))))
    spec.append(("dense", int(rng.choice([1, 2, 3, 10])), 0))
    return whc, spec


ctx = si.Context(0)
for case in range(cases):
    whc, spec = random_spec()
    b = int(rng.choice([16, 48])) if BIG else int(rng.choice([1, 5, 37, 130]))
    m = int(rng.integers(1, 6))
    print("case %d input %s B %d M %d spec %s" % (case, whc, b, m, spec), flush=True)
    table, n = so.conv_table(spec, whc)
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, m)))
    x = np.asfortranarray(rng.standard_normal((whc[0] * whc[1] * whc[2], b)))
    yref0 = so.forward(table, w_swa, x)
    y = np.asfortranarray(rng.standard_normal(yref0.shape))
    ctx.infer_setup(table, n, m, w_swa, p, x, y, 0.8)
    zs = np.asfortranarray(0.3 * rng.standard_normal((m, 2)))
    yref = so.forward(table, so.reconstruct(w_swa, p, zs[:, 0]), x)
    assert np.allclose(ctx.forward(zs[:, 0]), yref, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(yref).max()))
    lp_ref = np.array([so.logdensity(table, w_swa, p, x, y, 0.8, zs[:, j]) for j in range(2)])
    assert np.allclose(ctx.logdensity(zs), lp_ref, rtol=1e-9)
    lp, g = ctx.logdensity_grad(zs[:, 1])
    lpr, gr, _ = so.logdensity_grad(table, w_swa, p, x, y, 0.8, zs[:, 1])
    assert np.isclose(lp, lp_ref[1], rtol=1e-9) and np.allclose(g, gr, rtol=1e-6, atol=2e-7 * max(1.0, np.abs(gr).max()))
    if os.environ.get("SI_FUZZ_REPEAT"):   # same inputs, same bits
        assert np.array_equal(ctx.forward(zs[:, 0]), ctx.forward(zs[:, 0])) and np.array_equal(ctx.logdensity(zs), ctx.logdensity(zs))
        lp_b, g_b = ctx.logdensity_grad(zs[:, 1])
        assert lp_b == lp and np.array_equal(g_b, g)
    ctx.sample_rwmh(3, 0.05, seed=case)
    nb = int(rng.integers(1, b + 1))
    w32 = w_swa.astype(np.float32)
    ctx.train_setup(table, n, w32, x, y, b, int(rng.integers(0, 3)), 0.01, 0.9, 0.999)
    ids = rng.choice(b, nb, replace=False).astype(np.int64)
    sse = ctx.train_grad(ids, nb)
    loss, gref = so.mse_value_and_grad(table, w32.astype(np.float64), x[:, ids], y[:, ids])
    assert np.isclose(sse / (y.shape[0] * nb), loss, rtol=1e-9)
    # (atol: the fused conv + pool route evaluates act' at the window maximum where NNlib's rule picks an EARLIER input within
    # sqrt(eps) of it -- saturated tanh / sigmoid windows -- up to ~1e-8 of the pooled gradient per entry: kernels_conv.hip)
    assert np.allclose(ctx.train_grad_get(), gref, rtol=1e-7, atol=2e-7 * max(1.0, np.abs(gref).max()))
    if os.environ.get("SI_FUZZ_REPEAT"):
        g_a = ctx.train_grad_get()
        assert ctx.train_grad(ids, nb) == sse and np.array_equal(ctx.train_grad_get(), g_a)
    ctx.train_apply()
    ctx.train_step(ids)
    assert np.all(np.isfinite(ctx.train_get_weights()))
print("guard_fuzz_cnn: %d cases done" % cases, flush=True)
ctx.close()
