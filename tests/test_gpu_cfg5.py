"""BASELINE.json config 5 at its REAL shapes on one MI355X (one rank's share of the 8-GPU job):

  * construction at N = 51,138,049, K = 128, M = 64 -- the MFMA projection (M > 32) and the full eigen route at full
    size; A (52 GB) and P (26 GB) never leave the device, checks go through size-independent properties, host-fp64
    spot checks on pulled columns and torch-fp64 reductions over the library's own device buffers;
  * one rank's share of the data-sharded density, Chain(Dense(1024,6656,relu),Dense(6656,6656,relu),Dense(6656,1)),
    B/8 = 16,384 observations, M = 64, on the W_swa / P handed over ON THE DEVICE from that construction: lp against
    the oracle's forward on the pulled-back weight vector (rtol 1e-10), reconstruct rows against w_swa + P z.

Reference lines: src/subspace_construction.jl:45-52,61-65; src/space_inference.jl:90-95; BASELINE.md cfg5 row."""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu

N5, K5, M5, B5 = 51138049, 128, 64, 16384
DIMS5, ACTS5 = [1024, 6656, 6656, 1], [1, 1, 0]


@pytest.mark.timeout(900)
def test_cfg5_real_shapes_construct_then_data_shard_density(si):
    import torch
    from gpu_helpers import dev_view as _dev_view
    table, n = so.layer_table(DIMS5, ACTS5)
    assert n == N5
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~100 GB of free HBM (A 52 GB + P 26 GB + workspaces)")
    ctx = si.Context(0)
    try:
        gen = torch.Generator(device="cuda").manual_seed(55)
        cur = 0.02 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)   # glorot-scale weights
        keep = {}
        ctx.construct_begin(n, K5)
        for i in range(K5):   # random-walk snapshots (full-rank A), generated and pushed one at a time
            cur = cur + 2e-4 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)
            w32 = cur.to(torch.float32)
            torch.cuda.synchronize()
            ctx.construct_push_dev(w32.data_ptr(), 0, float(1 + i // 2))
            ctx.synchronize()
            if i in (0, K5 // 2, K5 - 1):
                keep[i] = w32.cpu().numpy()
        del w32, cur
        ctx.construct_gram()
        g = ctx.construct_gram_get()
        assert g.shape == (K5, K5) and np.array_equal(g, g.T)
        cols = [0, K5 // 2, K5 - 1]
        a_cols = [ctx.construct_get_A(c, 1)[:, 0] for c in cols]
        for i, ci in enumerate(cols):       # Gram entries of three pulled columns, recomputed on the host in fp64
            for j, cj in enumerate(cols):
                ref = float(np.dot(a_cols[i], a_cols[j]))
                assert abs(g[ci, cj] - ref) <= 1e-10 * np.sqrt(g[ci, ci] * g[cj, cj])
        w_swa, _, s, kk = ctx.construct_finish(M5, want_swa=True, want_p=False)
        assert kk == K5
        # K1 at full size, bit-exact: last deviation column = w_K - W_swa from the same fp32 snapshot
        assert np.array_equal(a_cols[2], keep[K5 - 1].astype(np.float64) - w_swa)
        # H1: singular values = sqrt of the top eigenvalues of G (LAPACK on the host)
        lam, v = np.linalg.eigh(g)
        lam, v = lam[::-1][:M5], v[:, ::-1][:, :M5]
        assert np.all(np.diff(s) <= 0) and np.allclose(s ** 2, lam, rtol=1e-9)
        # K3 (MFMA projection): P'P = diag(s^2), reduced in fp64 over the library's own 26 GB device buffer
        wptr, pptr, ld, mm = ctx.construct_result_ptr()
        assert mm == M5 and ld >= n
        p_t = _dev_view(pptr, (M5, ld))
        ptp = (p_t @ p_t.T).cpu().numpy()
        assert np.allclose(ptp, np.diag(s ** 2), atol=1e-8 * s[0] ** 2)
        assert float(p_t[:, n:].abs().max()) == 0.0                   # padding rows stay zero
        # A'P = G V = V diag(s^2): rows k of that identity for the three pulled columns a_k
        at = torch.from_numpy(np.stack(a_cols, axis=1)).cuda()        # n x 3
        atp = (p_t[:, :n] @ at).cpu().numpy().T                       # 3 x M
        ref = (v * lam[None, :])[cols, :]
        sign = np.sign(np.sum(atp * ref, axis=0))
        assert np.allclose(atp * sign[None, :], ref, rtol=1e-7, atol=1e-9 * lam[0])
        del at
        # ---- one rank's share of the data-sharded density on the device-resident (W_swa, P)
        rng = np.random.default_rng(5)
        x = np.asfortranarray(rng.standard_normal((DIMS5[0], B5)))
        y = np.asfortranarray(rng.standard_normal((DIMS5[-1], B5)))
        ctx.infer_setup(table, n, M5, None, None, x, y, 1.0)          # hand-over: no W_swa / P crosses PCIe
        z = np.asfortranarray(0.5 * rng.standard_normal((M5, 2)) / s[:, None] * s[-1])   # moves the weights by ~s_M per direction
        lp = ctx.logdensity(z)
        w = ctx.reconstruct(z[:, :1])[:, 0]                            # 409 MB
        r = y - so.forward(table, w, x)                                # ONE CPU forward: 1.7 TFLOP of fp64 BLAS
        lp_ref = so.lp_from_sse(float(np.sum(r * r)), y.size, 1.0)
        assert np.isclose(lp[0], lp_ref, rtol=1e-10), (lp[0], lp_ref)
        assert np.array_equal(ctx.logdensity(z), lp)                   # fixed-order reductions: same bits
        # data-sharded form: d_total covers all 8 ranks; this rank's SSE enters lp through the step-wise ABI
        ctx.rwmh_begin(2, 0.01, seed=3, d_total=8 * y.size)
        sse0 = ctx.rwmh_step_eval()
        assert np.isfinite(sse0[0]) and sse0[0] > 0.0
        ctx.rwmh_abort()
        # K4 over the 26 GB P: rows against w_swa + P z from the device P pulled for those rows only
        rows = np.concatenate([np.arange(0, 64), rng.integers(0, n, 256), np.arange(n - 64, n)])
        p_rows = p_t[:, torch.from_numpy(rows).cuda()].cpu().numpy().T
        assert np.allclose(w[rows], w_swa[rows] + p_rows @ z[:, 0], rtol=1e-13, atol=1e-15)
        del p_t
    finally:
        ctx.close()
        torch.cuda.empty_cache()
