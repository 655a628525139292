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
    # the gradient runs on the same fp32 arithmetic since round 5 (fp32 forward with kept activations + the fp32 reverse sweep of
    # the training step, P' g in fp64): its lp IS the fp32 density's, the gradient within fp32 rounding of the fp64 oracle's
    # (measured <= 4e-6 of its scale over these cases); the predictive forward stays fp64 (documented)
    lpg, g = gpu_ctx.logdensity_grad(z[:, 0])
    lp_ref, g_ref, _ = so.logdensity_grad(table, w, p, x, y, 0.7, z[:, 0])
    gerr = np.max(np.abs(g - g_ref)) / np.max(np.abs(g_ref))
    print("f32 gradient %s B=%d: max |g - g_ref| / max |g_ref| = %.2e, lp rel %.2e" % (dims, b, gerr, abs(lpg - lp_ref) / abs(lp_ref)))
    assert abs(lpg - lp_ref) <= 1e-5 * abs(lp_ref) and gerr <= 2e-5
    assert abs(lpg - lp[0]) <= 1e-9 * abs(lp[0])
    xn = np.random.default_rng(2).standard_normal((dims[0], 37))
    yp = gpu_ctx.predict(z[:, :2], xn)
    assert np.allclose(yp[:, :, 1], so.forward(table, w + p @ z[:, 1], xn), rtol=1e-9, atol=1e-12)
    # and the fp32 density is still what si_logdensity returns afterwards
    assert np.array_equal(gpu_ctx.logdensity(z), lp)


def _conv_cases():
    from tests.test_gpu_conv import CASES
    return CASES


@pytest.mark.parametrize("case", range(15))
def test_f32_conv_chains_vs_fp64_oracle(si, gpu_ctx, case):
    """compute_dtype = SI_F32 on Conv / MaxPool / flatten chains (round 5; kernels_conv.hip compiled for fp32 operands on
    v_mfma_f32_16x16x4_f32, the Dense layers behind flatten on kernels_gemm_f32.hip; squared errors in fp64): model outputs within
    5e-5 of their scale and lp within rtol 1e-5 of the fp64 oracle on every CNN case of tests/test_gpu_conv.py -- strided, dilated,
    padded convolutions, fused and un-fused pools, odd sizes, ragged tiles, every activation.  The reverse sweep stays fp64: a
    gradient on such a set-up is refused with a message."""
    from subspaceinference_jl_amd import _capi
    from tests.test_gpu_conv import CASES, _problem
    if case >= len(CASES):
        pytest.skip("no such case")
    whc, spec, b = CASES[case]
    m = 3
    table, n, w_swa, p, x, y = _problem(whc, spec, b, m, seed=case)
    gpu_ctx.infer_setup(table, n, m, w_swa, p, x, y, 0.8, compute_dtype=_capi.SI_F32)
    z = np.asfortranarray(0.2 * np.random.default_rng(99).standard_normal((m, 2)))
    yh = gpu_ctx.forward(z[:, 0])
    ref = so.forward(table, w_swa + p @ z[:, 0], x)
    err = np.abs(yh - ref).max() / max(1.0, np.abs(ref).max())
    lp = gpu_ctx.logdensity(z)
    lp_ref = np.array([so.logdensity(table, w_swa, p, x, y, 0.8, z[:, c]) for c in range(2)])
    rel = np.abs(lp / lp_ref - 1.0).max()
    print("f32 conv case %d: max |y - y_ref| / scale = %.2e, lp rel %.2e" % (case, err, rel))
    assert err <= 5e-5 and rel <= 1e-5
    assert np.array_equal(lp, gpu_ctx.logdensity(z))         # same inputs, same bits
    zc, lpc, _ = gpu_ctx.sample_rwmh(6, 0.05, seed=3, nchains=2)
    z1, lp1, _ = gpu_ctx.sample_rwmh(6, 0.05, seed=3, chain_id0=1, nchains=1)
    assert np.array_equal(zc[:, :, 1], z1[:, :, 0]) and np.array_equal(lpc[:, 1], lp1[:, 0])
    with pytest.raises(si.SubspaceError, match="SI_F32"):
        gpu_ctx.logdensity_grad(z[:, 0])


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


def test_f32_accept_decisions_at_cfg2(si, capsys):
    """BASELINE cfg2 at full size (N = 1 047 361, B = 100 000, M = 20, lp ~ -1.5e5): how many accept decisions of a
    1000-transition chain change when the density is evaluated in fp32?  The fp32 context is made to FOLLOW the fp64 chain
    (both are driven step-wise, both accept with the fp64 sum of squared errors) while its own evaluations are recorded;
    the decision each precision would take at every transition is then recomputed on the host from the library's Philox
    stream (oracle/philox.py, test infrastructure).  The rounding errors of ~1e5 squared residuals average out: the
    measured |lp32 - lp64| is a few 1e-5 ABSOLUTE on lp ~ -1.5e5 (rtol ~ 3e-10), so a flip needs |lp' - lp + Exp(1)|
    below that -- none in 1000 transitions here; the assertion allows a handful."""
    from oracle import philox
    from subspaceinference_jl_amd import _capi
    dims, acts, b, m, itr, sigma_z, seed = [128, 960, 960, 1], [1, 1, 0], 100000, 20, 1000, 0.1, 100
    table, n = so.layer_table(dims, acts)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((dims[0], b))
    y = rng.standard_normal((1, b))
    w = np.concatenate([np.concatenate([(rng.uniform(-1, 1, (fo, fi)) * np.sqrt(6.0 / (fi + fo))).reshape(-1, order="F"), np.zeros(fo)])
                        for fi, fo in zip(dims[:-1], dims[1:])])
    p = 0.01 * rng.standard_normal((n, m))
    d = float(b)
    with si.Context(0) as c64, si.Context(0) as c32:
        c64.infer_setup(table, n, m, w, p, x, y, 1.0, compute_dtype=_capi.SI_F64)
        c32.infer_setup(table, n, m, w, p, x, y, 1.0, compute_dtype=_capi.SI_F32)
        c64.rwmh_begin(itr, sigma_z, seed)
        c32.rwmh_begin(itr, sigma_z, seed)
        cur64 = cur32 = -np.inf
        flips, max_abs, acc = 0, 0.0, 0
        for t in range(itr):
            s64 = c64.rwmh_step_eval()
            s32 = c32.rwmh_step_eval()
            c64.rwmh_step_accept(s64)
            c32.rwmh_step_accept(s64)            # the fp32 context follows the fp64 chain's states
            lp64, lp32 = so.lp_from_sse(float(s64[0]), d, 1.0), so.lp_from_sse(float(s32[0]), d, 1.0)
            max_abs = max(max_abs, abs(lp64 - lp32))
            if t == 0:
                a64 = a32 = True
            else:
                e = philox.randexp(seed, 0, t)
                a64, a32 = (-e < lp64 - cur64), (-e < lp32 - cur32)
            flips += int(a64 != a32)
            if a64:
                cur64, cur32 = lp64, lp32
                acc += int(t > 0)
        z64, lps64, acc64 = c64.rwmh_end()
        z32, lps32, acc32 = c32.rwmh_end()
    assert np.array_equal(z64, z32)                                    # it did follow
    assert abs(acc64[0] - acc / (itr - 1)) < 1e-12                       # the host recomputation of the decisions is the library's
    with capsys.disabled():
        print("\n[f32 accept decisions, cfg2, %d transitions] accepted %d; decisions that differ fp32 vs fp64: %d; "
              "max |lp32 - lp64| = %.3e on lp ~ %.4e (rtol %.2e)" % (itr, acc, flips, max_abs, cur64, max_abs / abs(cur64)))
    assert max_abs <= 1e-5 * abs(cur64)
    assert flips <= 5


def test_f32_through_the_api_mirror(si):
    """sub_inference(...; compute_dtype = "f32") -- the reference's call with the non-default precision option"""
    from subspaceinference_jl_amd import flux
    rng = np.random.default_rng(5)
    x, y = rng.random((16, 200)), rng.random((1, 200))
    mdl = flux.Chain(flux.Dense(16, 64, flux.relu, rng=rng), flux.Dense(64, 32, flux.relu, rng=rng), flux.Dense(32, 1, rng=rng))
    data = flux.DataLoader(x, y, batchsize=50)
    table, n = flux.layer_table(mdl, None)
    w = rng.standard_normal(n) * 0.3
    p = 0.05 * rng.standard_normal((n, 4))
    z64, lp64 = si.sub_inference(mdl, data, w, p, σ_z=0.1, itr=50, M=4, seed=9, return_z=True)
    z32, lp32 = si.sub_inference(mdl, data, w, p, σ_z=0.1, itr=50, M=4, seed=9, return_z=True, compute_dtype="f32")
    assert np.array_equal(z64, z32) and np.allclose(lp32, lp64, rtol=1e-5)
    with pytest.raises(si.SubspaceError):
        si.sub_inference(mdl, data, w, p, itr=5, M=4, compute_dtype="bf16")
