"""GPU parity of the Conv / MaxPool / flatten layers through the C ABI: forward, log-density, the
gradient of the log-density, the training gradient, posterior predictive -- against the NumPy oracle, whose conv is
itself checked against scipy.signal.convolve2d and brute force (tests/test_oracle.py).  Shapes are deliberately ragged:
odd channel counts (channel pitch padding), stride 2, dilation, asymmetric windows, pad larger than the kernel reach, a
first layer with 3 channels (non-uniform tap tiles) and layers with a multiple of 16 channels (uniform tap tiles).
BASELINE config 4's 5.2 M-parameter CNN is sampled end to end at the end."""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu

R, T, S, I = so.ACT_RELU, so.ACT_TANH, so.ACT_SIGMOID, so.ACT_IDENTITY

CASES = [
    # (input (W, H, C), spec, B)
    ((6, 6, 2), [("conv", (3, 3), 4, T, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (2, 2), 3, S, (2, 1), (0, 1)),
                 ("flatten",), ("dense", 5, T), ("dense", 2, I)], 7),
    ((9, 8, 3), [("conv", (3, 2), 5, R, (1, 2), (2, 0), (2, 1)), ("maxpool", (3, 2), (2, 1)), ("flatten",), ("dense", 3, I)], 11),
    ((8, 8, 3), [("conv", (3, 3), 16, R, (1, 1), (1, 1)), ("conv", (3, 3), 32, T, (1, 1), (1, 1)), ("maxpool", (2, 2)),
                 ("conv", (1, 1), 7, I), ("flatten",), ("dense", 4, I)], 33),
    ((5, 7, 1), [("conv", (5, 3), 2, I, (1, 1), (4, 2)), ("flatten",), ("dense", 1, I)], 130),
    ((12, 12, 4), [("conv", (3, 3), 64, R, (2, 2), (1, 1)), ("conv", (3, 3), 64, R, (1, 1), (1, 1), (2, 2)), ("maxpool", (2, 2)),
                   ("flatten",), ("dense", 10, I)], 50),
    ((4, 4, 2), [("maxpool", (2, 2)), ("conv", (2, 2), 3, T), ("flatten",), ("dense", 2, I)], 5),      # chain starting on a pool
    ((3, 3, 2), [("flatten",), ("dense", 4, T), ("dense", 2, I)], 9),                                   # flatten only
    # a pool behind a pool (the second one takes the un-fused MaxPool gradient; the first one the pass fused with the
    # conv layer's act' and bias sum), 70 channels (two 64-row blocks of the fused pass), sigmoid (act'(pad channel) != 0)
    ((8, 8, 3), [("conv", (3, 3), 70, S, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("maxpool", (2, 2)), ("flatten",), ("dense", 3, I)], 19),
    # MaxPool((2, 2)) on ODD sizes (7 x 5 -> 3 x 2: the last row / column is dropped): the sampling path must not take the
    # fused conv + pool kernel here, the reverse sweep routes no gradient to the dropped pixels
    ((7, 5, 2), [("conv", (3, 3), 5, R, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("flatten",), ("dense", 2, I)], 13),
    # ... and on even sizes with a batch that leaves the last 128-position tile ragged (6 * 4 * 37 = 888 positions)
    ((6, 4, 3), [("conv", (3, 3), 18, T, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("flatten",), ("dense", 2, I)], 37),
    # window counts that are NOT a multiple of 4 through the general fused conv + pool kernel (70 channels keep the direct
    # first-layer kernel out): 3 * 3 * 5 = 45 and 5 * 2 * 5 = 50 windows -- the last 16-position slice holds 1 / 2 valid windows
    # whose inputs 1..3 lie past the last position (found by tools/guard_fuzz_cnn.py: wrong maxima in those windows)
    ((8, 8, 2), [("conv", (3, 3), 70, T), ("maxpool", (2, 2)), ("flatten",), ("dense", 3, I)], 5),
    ((8, 8, 2), [("conv", (3, 3), 70, T, (1, 2), (2, 1)), ("maxpool", (2, 2)), ("flatten",), ("dense", 3, I)], 5),
    # the four later activations (leakyrelu 4, elu 5, softplus 6, selu 7) in conv layers, fused and un-fused pools, a Dense layer
    ((8, 8, 2), [("conv", (3, 3), 6, so.ACT_ELU, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (3, 3), 5, so.ACT_SOFTPLUS, (1, 1), (1, 1)),
                 ("conv", (2, 2), 4, so.ACT_SELU, (2, 2)), ("flatten",), ("dense", 7, so.ACT_LEAKYRELU), ("dense", 2, I)], 21),
]


def test_golden_cnn_fixture(gpu_ctx):
    """the committed CNN vectors (tests/golden/cnn_density_rwmh.npz, generated from the oracle by make_golden.py): log-density
    rtol 1e-11, forward rtol 1e-10, gradient rtol 1e-8, the RWMH chain on the shared Philox stream rtol 1e-10."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_density_rwmh.npz"))
    spec = [("conv", (3, 3), 4, R, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (2, 2), 3, T, (2, 2)), ("flatten",), ("dense", 3, I)]
    table, n = so.conv_table(spec, (6, 6, 2))
    x, y = np.asfortranarray(d["X"]), np.asfortranarray(d["Y"])
    gpu_ctx.infer_setup(table, n, 4, d["W_swa"], np.asfortranarray(d["P"]), x, y, 0.7)
    assert np.allclose(gpu_ctx.logdensity(np.asfortranarray(d["Z"])), d["lp"], rtol=1e-11)
    assert np.allclose(gpu_ctx.forward(np.ascontiguousarray(d["Z"][:, 0])), d["Yhat0"], rtol=1e-10, atol=1e-12)
    lp1, g1 = gpu_ctx.logdensity_grad(np.ascontiguousarray(d["Z"][:, 1]))
    assert np.isclose(lp1, d["lp"][1], rtol=1e-11) and np.allclose(g1, d["grad1"], rtol=1e-8)
    z, lp, acc = gpu_ctx.sample_rwmh(12, 0.05, seed=77)
    assert np.allclose(z[:, :, 0], d["Z_chain"], rtol=1e-10, atol=1e-13) and np.allclose(lp[:, 0], d["lp_chain"], rtol=1e-10)


def _problem(whc, spec, b, m, seed):
    rng = np.random.default_rng(seed)
    table, n = so.conv_table(spec, whc)
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, m)))
    x = np.asfortranarray(rng.standard_normal((whc[0] * whc[1] * whc[2], b)))
    out = so.forward(table, w_swa, x)
    y = np.asfortranarray(rng.standard_normal(out.shape))
    return table, n, w_swa, p, x, y


@pytest.mark.parametrize("case", range(len(CASES)))
def test_conv_forward_logdensity_gradient(gpu_ctx, case):
    whc, spec, b = CASES[case]
    m = 3
    table, n, w_swa, p, x, y = _problem(whc, spec, b, m, seed=case)
    gpu_ctx.infer_setup(table, n, m, w_swa, p, x, y, 0.8)
    z = np.asfortranarray(0.2 * np.random.default_rng(99).standard_normal((m, 2)))
    yh = gpu_ctx.forward(z[:, 0])
    ref = so.forward(table, w_swa + p @ z[:, 0], x)
    assert np.allclose(yh, ref, rtol=1e-10, atol=1e-12), np.abs(yh - ref).max()
    lp = gpu_ctx.logdensity(z)                               # two chains: evaluated one after the other on conv chains
    lp_ref = [so.logdensity(table, w_swa, p, x, y, 0.8, z[:, c]) for c in range(2)]
    assert np.allclose(lp, lp_ref, rtol=1e-11)
    assert np.array_equal(lp, gpu_ctx.logdensity(z))         # same inputs, same bits
    lpg, g = gpu_ctx.logdensity_grad(z[:, 1])
    lpo, go, _ = so.logdensity_grad(table, w_swa, p, x, y, 0.8, z[:, 1])
    assert np.isclose(lpg, lpo, rtol=1e-11)
    assert np.allclose(g, go, rtol=1e-8, atol=1e-10 * np.abs(go).max()), (g, go)
    lpg2, g2 = gpu_ctx.logdensity_grad(z[:, 1])
    assert lpg2 == lpg and np.array_equal(g, g2)             # split-K partials are reduced in a fixed order


def test_conv_predict_and_rwmh_chain(gpu_ctx):
    whc, spec, b = CASES[0]
    m = 3
    table, n, w_swa, p, x, y = _problem(whc, spec, b, m, seed=21)
    gpu_ctx.infer_setup(table, n, m, w_swa, p, x, y, 1.1)
    zs, lps, acc = gpu_ctx.sample_rwmh(25, 0.05, seed=4, chain_id0=2, nchains=2)
    for c in range(2):
        zr, lpr, _, _ = so.sub_inference(table, x, y, w_swa, p, 0.05, 1.1, 25, seed=4, chain=2 + c)
        assert np.allclose(zs[:, :, c], zr, rtol=1e-9, atol=1e-12) and np.allclose(lps[:, c], lpr, rtol=1e-10)
    xnew = np.asfortranarray(np.random.default_rng(3).standard_normal((x.shape[0], 13)))
    yh = gpu_ctx.predict(zs[:, -3:, 0], xnew)
    for j in range(3):
        assert np.allclose(yh[:, :, j], so.forward(table, w_swa + p @ zs[:, -3 + j, 0], xnew), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("case", [0, 1, 4])
def test_conv_training_gradient_and_step(gpu_ctx, case):
    """si_train_grad on a CNN: d mse / d w against the oracle's reverse sweep (mse = -2 sigma^2 lp / d + const)."""
    whc, spec, b = CASES[case]
    table, n, w_swa, _, x, y = _problem(whc, spec, b, 1, seed=40 + case)
    w0 = w_swa.astype(np.float32)
    gpu_ctx.train_setup(table, n, w0, x, y, b, 0, 0.05)      # Descent(0.05)
    idx = np.random.default_rng(1).permutation(b)[: max(2, b // 2)]
    sse = gpu_ctx.train_grad(idx, idx.size)
    g = gpu_ctx.train_grad_get()
    w64 = w0.astype(np.float64)
    xs, ys = np.asfortranarray(x[:, idx]), np.asfortranarray(y[:, idx])
    _, _, gw = so.logdensity_grad(table, w64, np.zeros((n, 1)), xs, ys, 1.0, np.zeros(1))
    g_ref = -2.0 / ys.size * gw
    r = ys - so.forward(table, w64, xs)
    assert np.isclose(sse, float(np.sum(r * r)), rtol=1e-11)
    assert np.allclose(g, g_ref, rtol=1e-8, atol=1e-10 * np.abs(g_ref).max())
    gpu_ctx.train_apply()
    w1 = gpu_ctx.train_get_weights()
    assert np.allclose(w1, (w64 - 0.05 * g_ref).astype(np.float32), rtol=1e-6, atol=1e-7)


def test_conv_layer_tables_are_validated(si, gpu_ctx):
    whc, spec, b = CASES[0]
    table, n, w_swa, p, x, y = _problem(whc, spec, b, 2, seed=5)
    bad = list(table)
    bad[0] = ("conv", (3, 3, 2, 4), (7, 6), (1, 1), (1, 1), (1, 1), T, 0, 72)           # wrong input width
    with pytest.raises(si.SubspaceError):
        gpu_ctx.infer_setup(bad, n, 2, w_swa, p, x, y, 1.0)
    with pytest.raises(si.SubspaceError):                                                  # Dense straight after a Conv
        gpu_ctx.infer_setup([table[0], (4 * 6 * 6, 2, I, 100, 101)], n, 2, w_swa, p, x, np.zeros((2, b)), 1.0)


def test_api_chain_with_conv_layers(si, gpu_ctx):
    """The reference's API names on a Flux-style CNN: construction with the training step on the device, then RWMH."""
    from subspaceinference_jl_amd import flux
    rng = np.random.default_rng(0)
    x4 = rng.random((8, 8, 3, 40))
    y = rng.random((2, 40))
    wr = np.random.default_rng(2)
    model = flux.Chain(flux.Conv((3, 3), 3, 4, flux.relu, pad=1, rng=wr), flux.MaxPool((2, 2)),
                       flux.Conv((3, 3), 4, 6, flux.tanh, rng=wr), flux.flatten, flux.Dense(24, 2, rng=wr))
    data = flux.DataLoader(x4, y, batchsize=10, shuffle=True, rng=np.random.default_rng(7))
    assert model(x4).shape == (2, 40)
    w_swa, p = si.subspace_construction(model, flux.mse, data, flux.ADAM(0.01), T=4, c=1, M=3, ctx=gpu_ctx, verbose=False)
    n = sum(a.size for a in flux.params(model))
    assert w_swa.shape == (n,) and p.shape == (n, 3)
    chn, lp = si.sub_inference(model, data, w_swa, p, σ_z=0.1, itr=6, M=3, ctx=gpu_ctx, seed=1)
    # lp of the first sample against the host model restructured from that sample (`re(chn[1])(x)`)
    flux.load_flat(model, chn[0])
    r = y - model(x4)
    assert np.isclose(lp[0], so.lp_from_sse(float(np.sum(r * r)), y.size, 1.0), rtol=1e-9)


CFG4 = [("conv", (3, 3), 64, R, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (3, 3), 128, R, (1, 1), (1, 1)), ("maxpool", (2, 2)),
        ("conv", (3, 3), 256, R, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (3, 3), 256, R, (1, 1), (1, 1)), ("maxpool", (2, 2)),
        ("flatten",), ("dense", 4096, R), ("dense", 10, I)]


@pytest.mark.timeout(600)
def test_cfg4_cnn_sampled_end_to_end(gpu_ctx):
    """BASELINE config 4: conv3x3 3->64->128->256->256 (+ MaxPool 2 each) + fc 1024->4096->10 = 5,200,266 parameters on
    32x32x3 images, M = 20: construction on the flattened weight vector, then RWMH sampling of the CNN itself."""
    import torch
    table, n = so.conv_table(CFG4, (32, 32, 3))
    assert n == 5200266
    b, m, k = 256, 20, 24
    rng = np.random.default_rng(4)
    gen = torch.Generator(device="cuda").manual_seed(4)
    # He-scale start, random-walk snapshots (device-resident), K = 24 pushes (the full K = 200 construct is timed in bench.py)
    w = np.zeros(n)
    for r in table:
        if r[0] == "conv":
            cnt = int(np.prod(r[1]))
            w[r[7]:r[7] + cnt] = rng.standard_normal(cnt) * np.sqrt(2.0 / (r[1][0] * r[1][1] * r[1][2]))
        elif not isinstance(r[0], str):
            w[r[3]:r[3] + r[0] * r[1]] = rng.standard_normal(r[0] * r[1]) * np.sqrt(2.0 / r[0])
    cur = torch.from_numpy(w).cuda()
    gpu_ctx.construct_begin(n, k)
    for i in range(k):
        cur = cur + 1e-3 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)
        w32 = cur.to(torch.float32)
        torch.cuda.synchronize()
        gpu_ctx.construct_push_dev(w32.data_ptr(), 0, float(1 + i))
        gpu_ctx.synchronize()
    w_swa, p, s, _ = gpu_ctx.construct_finish(m)
    x = np.asfortranarray(rng.standard_normal((32 * 32 * 3, b)))
    y = np.asfortranarray(rng.standard_normal((10, b)))
    gpu_ctx.infer_setup(table, n, m, None, None, x, y, 1.0)
    zs, lps, acc = gpu_ctx.sample_rwmh(8, 0.02, seed=11)
    assert np.all(np.isfinite(lps))
    lp_ref = so.logdensity(table, w_swa, p, x, y, 1.0, zs[:, 0, 0])
    assert np.isclose(lps[0, 0], lp_ref, rtol=1e-10), (lps[0, 0], lp_ref)
    lp_last = so.logdensity(table, w_swa, p, x, y, 1.0, zs[:, -1, 0])
    assert np.isclose(lps[-1, 0], lp_last, rtol=1e-10)
    lpg, g = gpu_ctx.logdensity_grad(zs[:, -1, 0])
    lpo, go, _ = so.logdensity_grad(table, w_swa, p, x, y, 1.0, zs[:, -1, 0])
    assert np.isclose(lpg, lpo, rtol=1e-10) and np.allclose(g, go, rtol=1e-7, atol=1e-9 * np.abs(go).max())
