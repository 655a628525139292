"""K6 as a device-resident loop (reference src/space_inference.jl:111-116 for README.md:52-79 sized models):
one launch runs every transition of every chain.  It must give the SAME BITS as the launch-per-step loop (same Philox
stream, same order of every floating-point operation), which in turn is held against the oracle elsewhere."""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu

CASES = [
    ([10, 20, 20, 2], [0, 0, 0], 100, 3),          # README toy
    ([10, 20, 2], [1, 0], 37, 2),                   # two layers: the head hangs on the first layer
    ([7, 33, 18, 40, 3], [2, 1, 3, 0], 130, 5),     # odd widths: slots of 48 / 32 features, ragged tiles, tanh / relu / sigmoid
    ([4, 100, 1], [1, 0], 500, 4),                  # out = 100: feature tile 128 -> slots of 64 features; B = 500
    ([5, 70, 4], [2, 2], 64, 6),                    # out = 70: feature tile 96 -> slots of 48; a head of 4
]


@pytest.mark.parametrize("dims,acts,b,m", CASES)
def test_chain_loop_equals_launch_per_step_bit_for_bit(si, gpu_ctx, dims, acts, b, m):
    table, n = so.layer_table(dims, acts)
    rng = np.random.default_rng(sum(dims) + b)
    x, y = rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b))
    w, p = 0.3 * rng.standard_normal(n), 0.05 * rng.standard_normal((n, m))
    gpu_ctx.infer_setup(table, n, m, w, p, x, y, 0.8)
    try:
        out = {}
        for on in (False, True):
            gpu_ctx.set_chain_loop(on)
            out[on] = gpu_ctx.sample_rwmh(300, 0.07, seed=11, chain_id0=2, nchains=5)
        for a, bb in zip(out[False], out[True]):
            assert np.array_equal(a, bb)
        # and against the oracle's chain on the same stream (chain id 2 + 1)
        zr, lpr, _, _ = so.sub_inference(table, x, y, w, p, 0.07, 0.8, 300, seed=11, chain=3)
        assert np.allclose(out[True][0][:, :, 1], zr, rtol=1e-9, atol=1e-12) and np.allclose(out[True][1][:, 1], lpr, rtol=1e-9)
    finally:
        gpu_ctx.set_chain_loop(True)


def test_chain_loop_falls_back_where_it_does_not_apply(si, gpu_ctx):
    """prior term on, weights streamed out, a model too large for one workgroup: the launch-per-step loop runs (same API, same bits)"""
    table, n = so.layer_table([10, 20, 20, 2], [0, 0, 0])
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal((10, 100)), rng.standard_normal((2, 100))
    w, p = 0.3 * rng.standard_normal(n), 0.05 * rng.standard_normal((n, 3))
    gpu_ctx.infer_setup(table, n, 3, w, p, x, y, 1.0)
    z0, lp0, a0 = gpu_ctx.sample_rwmh(50, 0.1, seed=1)
    zw, lpw, aw, ww = gpu_ctx.sample_rwmh_weights(50, 0.1, seed=1, nchains=3)   # the loop + one K4 pass over all samples
    assert np.array_equal(z0[:, :, 0], zw[:, :, 0]) and np.array_equal(lp0[:, 0], lpw[:, 0])
    gpu_ctx.set_chain_loop(False)
    zs, lps, _, ws = gpu_ctx.sample_rwmh_weights(50, 0.1, seed=1, nchains=3)   # streamed output map of the launch path
    gpu_ctx.set_chain_loop(True)
    assert np.array_equal(zs, zw) and np.array_equal(lps, lpw) and np.array_equal(ws, ww)
    assert np.array_equal(ww[:, :, 1], gpu_ctx.reconstruct(zw[:, :, 1]))
    gpu_ctx.set_prior(2.0)
    zp, lpp, _ = gpu_ctx.sample_rwmh(50, 0.1, seed=1)
    assert np.all(np.isfinite(lpp)) and not np.array_equal(lpp, lp0)
    gpu_ctx.set_prior(0.0)
    table2, n2 = so.layer_table([64, 256, 256, 1], [1, 1, 0])   # 2 N B > 3 MFLOP, and far more LDS than a CU has
    gpu_ctx.infer_setup(table2, n2, 3, 0.1 * rng.standard_normal(n2), 0.01 * rng.standard_normal((n2, 3)),
                        rng.standard_normal((64, 2000)), rng.standard_normal((1, 2000)), 1.0)
    zb, lpb, _ = gpu_ctx.sample_rwmh(20, 0.1, seed=1)
    gpu_ctx.set_chain_loop(False)
    zc, lpc, _ = gpu_ctx.sample_rwmh(20, 0.1, seed=1)
    gpu_ctx.set_chain_loop(True)
    assert np.array_equal(zb, zc) and np.array_equal(lpb, lpc)
