"""Random (N, K, M) through the construction kernels (K1 push from host and device, batched push with a leading dimension,
column shift, K2 Gram for every tile count incl. the ragged last tile and the wide-K panel kernels, K3 projection), values
against the oracle / LAPACK; meant for the guard-page development library (see tools/guard_fuzz.py).
usage: SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin python3 tools/guard_fuzz_gram.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from oracle import subspace_oracle as so  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):
    si._capi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libsubspace_hip_dev.so")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
NS = [1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 129, 1000, 4097, 20001]
ctx = si.Context(0)
for case in range(cases):
    n = int(rng.choice(NS))
    k = int(rng.integers(1, 261)) if rng.random() < 0.8 else int(rng.choice([16, 32, 96, 100, 112, 128, 144, 200, 208, 209, 256]))
    if rng.random() < 0.12:   # the README toy's regime: more deviation columns than weights (K > N route of si_construct_finish)
        n = int(rng.choice([3, 33, 127, 129, 682, 1000]))
        k = int(rng.choice([500, 1000, 1500]))
    if n * k > 3_000_000:
        k = max(1, 3_000_000 // n)
    dt = np.float32 if rng.random() < 0.6 else np.float64
    how = int(rng.integers(0, 3))            # 0 host pushes, 1 device pushes, 2 one batched device push with a padded ld
    max_cols = int(rng.integers(1, k + 1)) if rng.random() < 0.2 else 0
    a32 = rng.random() < 0.25               # the opt-in fp32 storage of the deviation matrix (si_construct_set_storage)
    print("case %d N %d K %d %s how %d max_cols %d%s" % (case, n, k, dt.__name__, how, max_cols, " A fp32" if a32 else ""), flush=True)
    base = 0.3 * rng.standard_normal(n)
    snaps = []
    for j in range(k):
        base = base + 0.05 * rng.standard_normal(n)
        snaps.append(base.astype(dt))
    ns = [float(1 + j // 2) for j in range(k)]
    ctx.construct_begin(n, k, max_cols)
    if a32:
        ctx.construct_set_storage(0)
    code = 0 if dt == np.float32 else 1
    if how == 0:
        for w, nn in zip(snaps, ns):
            ctx.construct_push(w, nn)
    else:
        ld = n + int(rng.integers(0, 5)) if how == 2 else n
        ld += (ld * snaps[0].itemsize) % 8 and 1   # rows of the batch start 8-byte aligned
        host = np.zeros((k, ld), dtype=dt)
        host[:, :n] = np.stack(snaps)
        dev = torch.from_numpy(host).cuda()
        if how == 2:
            ctx.construct_push_batch_dev(dev.data_ptr(), code, ld, ns)
        else:
            for j in range(k):
                ctx.construct_push_dev(dev.data_ptr() + j * ld * host.itemsize, code, ns[j])
        ctx.synchronize()
        del dev
    w_ref, a_ref = so.construct_stream(snaps, ns)
    if a32:
        a_ref = a_ref.astype(np.float32).astype(np.float64)   # columns formed in fp64, stored rounded once
    if max_cols:
        a_ref = a_ref[:, -max_cols:]     # the paper's column shift: the newest max_cols deviation columns
    kk = a_ref.shape[1]
    if kk <= n or rng.random() < 0.5:   # (K > N: half of the cases leave the choice of the route to si_construct_finish)
        ctx.construct_gram()
        g = ctx.construct_gram_get()
        if os.environ.get("SI_FUZZ_REPEAT"):   # the Gram reduction is fixed-order: same bits on a second pass
            ctx.construct_gram()
            assert np.array_equal(ctx.construct_gram_get(), g)
        g_ref = a_ref.T @ a_ref
        if max_cols:   # the ring keeps the columns in slot order: compare as sets through the eigenvalues
            assert np.allclose(np.linalg.eigvalsh(g), np.linalg.eigvalsh(g_ref), rtol=1e-9, atol=1e-11 * max(1e-300, np.abs(g_ref).max()))
        else:
            assert np.allclose(g, g_ref, rtol=1e-10, atol=1e-12 * max(1e-300, np.abs(g_ref).max()))
    m = int(rng.integers(1, min(kk, 12) + 1))
    if kk > 40 and rng.random() < 0.3:
        m = int(rng.integers(33, min(kk, 80) + 1))   # wide subspaces: the projection on the matrix cores (M > 32)
    s_ref = np.linalg.svd(a_ref, compute_uv=False)
    try:
        w_got, p_got, s_got, _ = ctx.construct_finish(m)
        assert np.array_equal(w_got, w_ref)
        assert np.allclose(s_got, s_ref[:m], rtol=1e-5, atol=1e-9 * s_ref[0]), (s_got, s_ref[:m])
        assert np.allclose(p_got.T @ p_got, np.diag(s_got ** 2), rtol=1e-6, atol=1e-7 * s_got[0] ** 2)
    except si.BoundsError:
        assert m > min(n, kk) or s_ref[m - 1] < 1e-6 * s_ref[0]
print("guard_fuzz_gram: %d cases done" % cases, flush=True)
ctx.close()
