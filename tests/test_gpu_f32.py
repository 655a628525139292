"""compute_dtype = SI_F32 (SURVEY section 0 Q6 / 8(b),(d): "forward fp64 (exact parity) with fp32 as a measured option"):
the Dense-chain density on the fp32 matrix instruction -- X, the per-step weights (W_swa + P z formed in fp64, rounded
once) and the activations in fp32; the narrow head, the last bias and the sum of squared errors in fp64.  Checked through
the C ABI against the fp64 ORACLE; the tolerances below are the MEASURED error levels with headroom, north_star's bound
is rtol 1e-4 on lp."""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu

# (dims, acts, B): generic kernel (toy: in = 10, out = 20; unaligned layer offsets), LDS-DMA kernel (in % 16 == 0,
# out % 4 == 0) with ragged feature / batch edges, fused narrow head and a wide last layer, every activation
CASES = [
    ([10, 20, 20, 2], [0, 0, 0], 100),                  # README toy
    ([2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000),   # docs/src/nn_example.md:101 (w_off of layer 3 is not 16-byte aligned)
    ([16, 192, 1], [1, 0], 128),                        # one exact tile of the DMA kernel
    ([32, 960, 960, 1], [1, 1, 0], 1000),               # cfg2's widths, ragged batch edge (1000 = 7 x 128 + 104)
    ([48, 100, 36, 1], [2, 3, 0], 333),                 # out % 4 == 0 but not % 32: clamped feature rows, tanh / sigmoid
    ([64, 128, 64, 8], [1, 1, 0], 257),                 # wide last layer: un-fused fp32 SSE
    ([16, 64, 4], [4, 5], 70),                          # leakyrelu / elu (elementwise pass) into a fused head of 4
    ([16, 32, 3], [6, 7], 65),                          # softplus / selu
]


def _model(dims, acts, b, seed):
    table, n = so.layer_table(dims, acts)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((dims[0], b))
    y = rng.standard_normal((dims[-1], b))
    w = np.concatenate([np.concatenate([(rng.uniform(-1, 1, (fo, fi)) * np.sqrt(6.0 / (fi + fo))).reshape(-1, order="F"),
                                        0.1 * rng.standard_normal(fo)]) for fi, fo in zip(dims[:-1], dims[1:])])
    m = 5
    p = 0.05 * rng.standard_normal((n, m))
    return table, n, m, w, p, x, y


@pytest.mark.parametrize("dims,acts,b", CASES)
def test_f32_forward_and_lp_vs_fp64_oracle(si, gpu_ctx, dims, acts, b):
    from subspaceinference_jl_amd import _capi
    table, n, m, w, p, x, y = _model(dims, acts, b, seed=sum(dims) + b)
    gpu_ctx.infer_setup(table, n, m, w, p, x, y, 0.7, compute_dtype=_capi.SI_F32)
    rng = np.random.default_rng(1)
    z = 0.3 * rng.standard_normal((m, 3))
    lp = gpu_ctx.logdensity(z)
    for c in range(3):
        wz = w + p @ z[:, c]
        yhat_ref = so.forward(table, wz, x)
        yhat = gpu_ctx.forward(z[:, c])
        scale = np.max(np.abs(yhat_ref)) + 1e-30
        # fp32 inputs (2^-24 relative each) and fp32 accumulation over <= 960 terms: measured <= 3e-6 of the output scale
        assert np.max(np.abs(yhat - yhat_ref)) <= 2e-5 * scale
        lp_ref = so.logdensity(table, w, p, x, y, 0.7, z[:, c])
        assert abs(lp[c] - lp_ref) <= 1e-5 * abs(lp_ref)   # north_star: 1e-4
    # chain-batched evaluation (grid.y) == one at a time, bit for bit
    one = np.array([gpu_ctx.logdensity(z[:, c:c + 1])[0] for c in range(3)])
    assert np.array_equal(one, lp)
    # the gradient and the predictive forward stay in fp64 in this mode (documented): they match the fp64 oracle tightly
    lpg, g = gpu_ctx.logdensity_grad(z[:, 0])
    lp_ref, g_ref, _ = so.logdensity_grad(table, w, p, x, y, 0.7, z[:, 0])
    assert abs(lpg - lp_ref) <= 1e-10 * abs(lp_ref) and np.allclose(g, g_ref, rtol=1e-7, atol=1e-9 * np.max(np.abs(g_ref)))
    xn = np.random.default_rng(2).standard_normal((dims[0], 37))
    yp = gpu_ctx.predict(z[:, :2], xn)
    assert np.allclose(yp[:, :, 1], so.forward(table, w + p @ z[:, 1], xn), rtol=1e-9, atol=1e-12)
    # and the fp32 density is still what si_logdensity returns afterwards
    assert np.array_equal(gpu_ctx.logdensity(z), lp)


def test_f32_refused_for_conv_chains(si, gpu_ctx):
    from subspaceinference_jl_amd import _capi
    spec = [("conv", (3, 3), 4, 1, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("flatten",), ("dense", 3, 0)]
    table, n = so.conv_table(spec, (8, 8, 2))
    rng = np.random.default_rng(0)
    with pytest.raises(si.SubspaceError, match="SI_F32"):
        gpu_ctx.infer_setup(table, n, 2, rng.standard_normal(n), rng.standard_normal((n, 2)), rng.standard_normal((128, 5)),
                            rng.standard_normal((3, 5)), 1.0, compute_dtype=_capi.SI_F32)


def test_f32_rwmh_chain_follows_the_f64_chain_on_a_small_model(si, gpu_ctx):
    """Same Philox stream in both precisions.  On a small model (lp ~ -1e2) the fp32 error of lp (~1e-5) flips an accept
    decision only when |lp' - lp + Exp(1)| falls under it: the chains agree for the whole run here; at cfg2 (lp ~ -7e5)
    they do not -- bench.py reports that divergence."""
    from subspaceinference_jl_amd import _capi
    table, n, m, w, p, x, y = _model([10, 20, 20, 2], [0, 0, 0], 100, seed=7)
    out = {}
    for name, dt in (("f64", _capi.SI_F64), ("f32", _capi.SI_F32)):
        gpu_ctx.infer_setup(table, n, m, w, p, x, y, 1.0, compute_dtype=dt)
        out[name] = gpu_ctx.sample_rwmh(200, 0.05, seed=3, nchains=4)
    z64, lp64, acc64 = out["f64"]
    z32, lp32, acc32 = out["f32"]
    assert np.array_equal(z64, z32) and np.array_equal(acc64, acc32)   # identical decisions => identical states
    assert np.allclose(lp32, lp64, rtol=1e-5)
    # the streamed output map delivers the fp64 W_swa + P z in either mode
    zs, lps, accs, ws = gpu_ctx.sample_rwmh_weights(12, 0.05, seed=3)
    assert np.allclose(ws[:, :, 0], w[:, None] + p @ zs[:, :, 0], rtol=1e-13, atol=1e-15)
