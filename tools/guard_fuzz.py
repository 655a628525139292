"""Random Dense chains through forward / log-density (chain-batched) / gradient / training gradient / construction, meant to
run over the DEVELOPMENT library with the guard-page allocator (csrc/guard_alloc.hip), where an out-of-bounds access of
any kernel faults at once.  The widths favour the tile edges of the kernels (96 / 128 / 64-row tiles, 16-deep k tiles).
usage: SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin [SI_FUZZ_BIG=1] [SI_FUZZ_REPEAT=1] python3 tools/guard_fuzz.py [cases] [seed]
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
if os.environ.get("SI_FUZZ_BIG"):   # wider layers and longer batches (more row tiles, the XCD-grouped maps, other split-K plans)
    WIDTHS += [384, 480, 512, 960, 1000]
    BATCH += [4096, 10000]
ctx = si.Context(0)
for case in range(cases):
    nl = int(rng.integers(1, 5))
    dims = [int(rng.choice([1, 2, 3, 7, 12, 15, 16, 17, 20, 31, 33, 40]))] + [int(rng.choice(WIDTHS)) for _ in range(nl - 1)] + \
           [int(rng.choice([1, 1, 2, 3, 4, 5, 8, 17, 96]))]
    acts = [int(rng.integers(0, 8)) for _ in range(nl)]
    b = int(rng.choice(BATCH)) if rng.random() < 0.93 else int(rng.choice([7000, 22001, 33001, 66003]))   # a few with more than 512 tiles per layer (a second round of workgroups)
    m = int(rng.integers(1, 9)) if rng.random() < 0.85 else int(rng.integers(20, 71))
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
    if os.environ.get("SI_FUZZ_REPEAT"):   # same inputs, same bits: a race inside a kernel would show as a rare difference
        assert np.array_equal(ctx.forward(zs[:, 0]), ctx.forward(zs[:, 0]))
        assert np.array_equal(ctx.logdensity(zs), ctx.logdensity(zs))
        lp_b, g_b = ctx.logdensity_grad(zs[:, 1])
        assert lp_b == lp and np.array_equal(g_b, g)
    # chains stacked in one launch == one chain at a time, bit for bit; the streamed output map == si_reconstruct
    nch = int(rng.integers(1, 4))
    zc, lpc, _ = ctx.sample_rwmh(4, 0.1, seed=case, nchains=nch)
    z1, lp1, _ = ctx.sample_rwmh(4, 0.1, seed=case, chain_id0=nch - 1, nchains=1)
    assert np.array_equal(zc[:, :, nch - 1], z1[:, :, 0]) and np.array_equal(lpc[:, nch - 1], lp1[:, 0])
    zw, lpw, _, wmap = ctx.sample_rwmh_weights(5, 0.1, seed=case + 1, nchains=nch)
    assert np.array_equal(wmap[:, :, 0], ctx.reconstruct(np.asfortranarray(zw[:, :, 0])))
    # the device-resident transition loop (small chains: one launch for all transitions) == the launch-per-step loop, bit for bit
    ctx.set_chain_loop(False)
    z_off, lp_off, acc_off = ctx.sample_rwmh(6, 0.1, seed=case + 2, nchains=nch)
    ctx.set_chain_loop(True)
    z_on, lp_on, acc_on = ctx.sample_rwmh(6, 0.1, seed=case + 2, nchains=nch)
    assert np.array_equal(z_on, z_off) and np.array_equal(lp_on, lp_off) and np.array_equal(acc_on, acc_off)
    # compute_dtype = SI_F32 (round 4): the same density with fp32 layers -- outputs and lp against the fp64 oracle at the
    # tolerance the header states, stacked chains == one chain at a time
    ctx.infer_setup(table, n, m, w_swa, p, x, y, 0.8, compute_dtype=si._capi.SI_F32)
    y32 = ctx.forward(zs[:, 0])
    assert np.max(np.abs(y32 - yref)) <= 5e-5 * max(1.0, np.abs(yref).max()), np.max(np.abs(y32 - yref))
    lp32 = ctx.logdensity(zs)
    assert np.all(np.abs(lp32 - lp_ref) <= 1e-4 * np.abs(lp_ref)), (lp32, lp_ref)
    zc32, lpc32, _ = ctx.sample_rwmh(4, 0.1, seed=case, nchains=nch)
    z132, lp132, _ = ctx.sample_rwmh(4, 0.1, seed=case, chain_id0=nch - 1, nchains=1)
    assert np.array_equal(zc32[:, :, nch - 1], z132[:, :, 0]) and np.array_equal(lpc32[:, nch - 1], lp132[:, 0])
    if os.environ.get("SI_FUZZ_REPEAT"):
        assert np.array_equal(ctx.logdensity(zs), lp32)
    # training gradient on a random batch against the restatement of Zygote's, then one optimiser step
    # (half of the cases with a batch of whole 16-deep k tiles: the LDS-DMA weight-gradient kernel where the widths are even)
    nb = int(rng.integers(1, b + 1))
    if b >= 16 and rng.random() < 0.5:
        nb = max(16, nb // 16 * 16)
    opt_kind = int(rng.integers(0, 3))
    w32 = w_swa.astype(np.float32)
    ctx.train_setup(table, n, w32, x, y, b, opt_kind, 0.01, 0.9, 0.999)
    ids = rng.choice(b, nb, replace=False).astype(np.int64)
    sse = ctx.train_grad(ids, nb)
    loss, gref = so.mse_value_and_grad(table, w32.astype(np.float64), x[:, ids], y[:, ids])
    assert np.isclose(sse / (dims[-1] * nb), loss, rtol=1e-9)
    assert np.allclose(ctx.train_grad_get(), gref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(gref).max()))
    if os.environ.get("SI_FUZZ_REPEAT"):
        g_a = ctx.train_grad_get()
        assert ctx.train_grad(ids, nb) == sse and np.array_equal(ctx.train_grad_get(), g_a)
    ctx.train_apply()
    ctx.train_step(ids)
    assert np.all(np.isfinite(ctx.train_get_weights()))
    # construction on the same N: K pushes of fp32 / fp64 vectors, ragged K; W_swa bit-exact, P'P = diag(s^2), s vs LAPACK
    k = int(rng.integers(2, 40))
    mm = int(rng.integers(1, min(k, 8) + 1))
    ctx.construct_begin(n, k)
    dt = np.float32 if rng.random() < 0.7 else np.float64
    base = w_swa.copy()
    snaps = []
    for j in range(k):
        base = base + 0.05 * rng.standard_normal(n)
        snaps.append(base.astype(dt))
        ctx.construct_push(snaps[-1], float(1 + j))
    w_ref, a_ref = so.construct_stream(snaps, [float(1 + j) for j in range(k)])
    try:
        w_got, p_got, s_got, _ = ctx.construct_finish(mm)
        assert np.array_equal(w_got, w_ref)
        s_ref = np.linalg.svd(a_ref, compute_uv=False)[:mm]
        assert np.allclose(s_got, s_ref, rtol=1e-5, atol=1e-9 * s_ref[0]), (s_got, s_ref)
        assert np.allclose(p_got.T @ p_got, np.diag(s_got ** 2), rtol=1e-6, atol=1e-7 * s_got[0] ** 2)
    except si.BoundsError:
        assert min(n, k) < mm or np.linalg.svd(a_ref, compute_uv=False)[mm - 1] < 1e-6 * np.linalg.norm(a_ref, 2)
print("guard_fuzz: %d cases done" % cases, flush=True)
ctx.close()
