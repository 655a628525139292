"""CPU tests of the oracle.  The reference has no tests or fixtures for this path (PARITY UNPINNED), so the
oracle is cross-checked by identities that do not depend on it being right (SURVEY.md 8c i-v) and against the
committed golden vectors (drift guard)."""
import math
import os

import numpy as np
import pytest
import scipy.stats

from oracle import philox
from oracle import subspace_oracle as so

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    def h(c, k):
        return [int(v) for v in philox.philox4x32(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))]
    assert h([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert h([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert h([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_moments():
    z = np.concatenate([philox.normals(7, 0, t, 20) for t in range(2000)])
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1.0) < 0.03
    e = np.array([philox.randexp(7, 1, t) for t in range(4000)])
    assert abs(e.mean() - 1.0) < 0.05 and e.min() > 0


def test_swa_closed_form_and_quirks():
    # Q1: zero start, first n = 1 -> W/2.  Q2: same n repeated inside an epoch = EMA with factor n/(n+1)
    w = np.array([2.0, -4.0, 8.0], dtype=np.float32)
    s, dev = so.swa_dev_push(np.zeros(3), w, 1.0)
    assert np.array_equal(s, w.astype(np.float64) / 2) and np.array_equal(dev, w / 2)
    rng = np.random.default_rng(0)
    ws = [rng.standard_normal(5).astype(np.float32) for _ in range(6)]
    ns = [1.0, 1.0, 2.0, 2.0, 3.0, 3.0]
    s_ref, a = so.construct_stream(ws, ns)
    s = np.zeros(5)
    for w_, n in zip(ws, ns):
        s = (n / (n + 1.0)) * s + w_.astype(np.float64) / (n + 1.0)  # algebraically equal EMA form
    assert np.allclose(s, s_ref, rtol=1e-14, atol=0)
    assert a.shape == (5, 6)  # Q3: every column kept
    assert np.array_equal(a[:, -1], ws[-1].astype(np.float64) - s_ref)  # deviation from the UPDATED mean


def test_projection_identities():
    rng = np.random.default_rng(1)
    a = np.asfortranarray(rng.standard_normal((300, 12)) * np.logspace(0, -2, 12)[None, :])
    p, s = so.projection_from_A(a, 4)
    u, sv, vt = np.linalg.svd(a, full_matrices=False)
    assert np.allclose(p.T @ p, np.diag(sv[:4] ** 2), atol=1e-10)          # P'P = diag(s^2)
    assert np.allclose(np.abs(p), np.abs(u[:, :4] * sv[:4]), atol=1e-12)    # |P_j| = |U_j s_j|
    assert np.allclose(np.abs(a @ vt[:4].T), np.abs(p), atol=1e-10)         # P = A V_M up to sign
    with pytest.raises(IndexError):
        so.projection_from_A(a, 13)


def test_logpdf_against_scipy():
    rng = np.random.default_rng(2)
    y, mu = rng.standard_normal(50), rng.standard_normal(50)
    for sigma in (0.3, 1.0, 2.5):
        lp, sse = so.mvnormal_logpdf_iso(y, mu, sigma)
        ref = scipy.stats.multivariate_normal(mean=mu, cov=sigma ** 2 * np.eye(50)).logpdf(y)
        assert math.isclose(lp, ref, rel_tol=1e-12)
        assert math.isclose(so.lp_from_sse(sse, 50, sigma), lp, rel_tol=0, abs_tol=0)


def test_forward_hand_computed():
    # 2 -> 2 (relu) -> 1 on 3 points, flat layout [vec(W1) col-major; b1; vec(W2); b2]
    table, n = so.layer_table([2, 2, 1], [so.ACT_RELU, so.ACT_IDENTITY])
    assert n == 9 and table[1][3] == 6 and table[1][4] == 8
    w1 = np.array([[1.0, -2.0], [0.5, 3.0]])  # out x in
    flat = np.concatenate([w1.reshape(-1, order="F"), [0.1, -0.2], [2.0, -1.0], [0.05]])
    x = np.array([[1.0, 0.0, -1.0], [2.0, 1.0, 0.5]])
    h = np.maximum(w1 @ x + np.array([[0.1], [-0.2]]), 0)
    exp = np.array([[2.0, -1.0]]) @ h + 0.05
    assert np.allclose(so.forward(table, flat, x), exp, atol=1e-15)
    # prior term is dead code (Q4): lp is the likelihood alone
    y = np.zeros((1, 3))
    lp = so.logdensity(table, flat, np.zeros((9, 1)), x, y, 1.0, np.zeros(1))
    assert math.isclose(lp, -(3 * math.log(2 * math.pi)) / 2 - float((exp ** 2).sum()) / 2, rel_tol=1e-14)


def test_gradient_against_finite_differences():
    rng = np.random.default_rng(5)
    for dims, acts in (([4, 9, 3], [so.ACT_TANH, so.ACT_IDENTITY]), ([3, 8, 6, 1], [so.ACT_SIGMOID, so.ACT_TANH, so.ACT_SIGMOID])):
        table, n = so.layer_table(dims, acts)
        w_swa, p = 0.4 * rng.standard_normal(n), 0.3 * rng.standard_normal((n, 4))
        x, y = rng.standard_normal((dims[0], 17)), rng.standard_normal((dims[-1], 17))
        z = rng.standard_normal(4)
        lp, gz, gw = so.logdensity_grad(table, w_swa, p, x, y, 0.8, z)
        assert math.isclose(lp, so.logdensity(table, w_swa, p, x, y, 0.8, z), rel_tol=1e-14)
        for m in range(4):
            e = np.zeros(4)
            e[m] = 1e-6
            fd = (so.logdensity(table, w_swa, p, x, y, 0.8, z + e) - so.logdensity(table, w_swa, p, x, y, 0.8, z - e)) / 2e-6
            assert math.isclose(gz[m], fd, rel_tol=1e-6, abs_tol=1e-7)
        assert np.allclose(gz, p.T @ gw)


def test_rwmh_gaussian_target():
    # stationary moments of N(0,1) in 2-D; `itr` samples include the initial draw
    dens = lambda z: -0.5 * float(z @ z)
    zs, lps, nacc = so.rwmh(dens, 2, 6000, 1.0, seed=11)
    assert zs.shape == (2, 6000) and lps.shape == (6000,)
    assert np.array_equal(zs[:, 0], philox.normals(11, 0, 0, 2))
    burn = zs[:, 500:]
    assert np.all(np.abs(burn.mean(axis=1)) < 0.15) and np.all(np.abs(burn.var(axis=1) - 1.0) < 0.2)
    assert 0.3 < nacc / 5999 < 0.8
    # rejected steps repeat the previous sample
    rep = np.all(zs[:, 1:] == zs[:, :-1], axis=0)
    assert rep.sum() == 5999 - nacc


def test_golden_vectors_match_oracle():
    g = np.load(os.path.join(GOLD, "toy_construct_k12.npz"))
    w_swa, a = so.construct_stream(list(g["snapshots"]), list(g["ns"]))
    assert np.array_equal(w_swa, g["W_swa"]) and np.array_equal(a, g["A"])
    p, s = so.projection_from_A(a, 3)
    assert np.allclose(s[:3], g["s"], rtol=1e-12)
    sign = np.sign(np.sum(p * g["P"], axis=0))
    assert np.allclose(p * sign, g["P"], rtol=1e-9, atol=1e-12)
    d = np.load(os.path.join(GOLD, "toy_density_rwmh.npz"))
    table, _ = so.layer_table([10, 20, 20, 2], [0, 0, 0])
    lps = [so.logdensity(table, d["W_swa"], d["P"], d["X"], d["Y"], 1.0, d["Z"][:, j]) for j in range(5)]
    assert np.allclose(lps, d["lp"], rtol=1e-12)
    z, lp, w, nacc = so.sub_inference(table, d["X"], d["Y"], d["W_swa"], d["P"], 1.0, 1.0, 10, seed=1234)
    assert np.allclose(z, d["Z_chain"], rtol=1e-12) and np.allclose(lp, d["lp_chain"], rtol=1e-12)
    assert nacc == int(d["nacc"])


def test_golden_toy_at_its_real_shape_k1000():
    """README.md:52-79: batchsize 1 x 100 observations x T = 10 => K = 1000 deviation columns of N = 682 weights.
    The fixture pins the oracle at K > N; identities that do not depend on the oracle being right: LAPACK's SVD of A."""
    g = np.load(os.path.join(GOLD, "toy_construct_k1000.npz"))
    assert g["snapshots"].shape == (1000, 682) and g["snapshots"].dtype == np.float32
    w_swa, a = so.construct_stream(list(g["snapshots"]), list(g["ns"]))
    assert np.array_equal(w_swa, g["W_swa"]) and a.shape == (682, 1000)
    p, s = so.projection_from_A(a, 3)
    assert np.allclose(s[:20], g["s"], rtol=1e-12)
    assert np.allclose(p * np.sign(np.sum(p * g["P"], axis=0)), g["P"], rtol=1e-9, atol=1e-12)
    sv = np.linalg.svd(a, compute_uv=False)
    assert np.allclose(sv[:20], g["s"], rtol=1e-10)
    assert np.allclose(g["P"].T @ g["P"], np.diag(g["s"][:3] ** 2), rtol=1e-9, atol=1e-9 * g["s"][0] ** 2)


GOLD_CNN_WHC = (6, 6, 2)   # tests/golden/make_golden.py CNN_WHC / CNN_SPEC
GOLD_CNN_SPEC = [("conv", (3, 3), 4, so.ACT_RELU, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (2, 2), 3, so.ACT_TANH, (2, 2)),
                 ("flatten",), ("dense", 3, so.ACT_IDENTITY)]


def test_golden_cnn_vectors_match_oracle():
    d = np.load(os.path.join(GOLD, "cnn_density_rwmh.npz"))
    table, n = so.conv_table(GOLD_CNN_SPEC, GOLD_CNN_WHC)
    assert n == d["W_swa"].shape[0]
    lps = [so.logdensity(table, d["W_swa"], d["P"], d["X"], d["Y"], 0.7, d["Z"][:, j]) for j in range(5)]
    assert np.allclose(lps, d["lp"], rtol=1e-12)
    assert np.allclose(so.forward(table, so.reconstruct(d["W_swa"], d["P"], d["Z"][:, 0]), d["X"]), d["Yhat0"], rtol=1e-12, atol=1e-14)
    _, gz, _ = so.logdensity_grad(table, d["W_swa"], d["P"], d["X"], d["Y"], 0.7, d["Z"][:, 1])
    assert np.allclose(gz, d["grad1"], rtol=1e-11)
    z, lp, _, nacc = so.sub_inference(table, d["X"], d["Y"], d["W_swa"], d["P"], 0.05, 0.7, 12, seed=77)
    assert np.allclose(z, d["Z_chain"], rtol=1e-12) and np.allclose(lp, d["lp_chain"], rtol=1e-12) and nacc == int(d["nacc"])


def test_later_activations_definitions_and_derivatives():
    """leakyrelu / elu / softplus / selu as Flux 0.11.2 / NNlib 0.7.23 define them, and their derivative rebuilt from the
    OUTPUT (what the reverse sweep does) against central differences of the definition."""
    x = np.concatenate([np.linspace(-6.0, 6.0, 241), [-40.0, 40.0, 1e-9, -1e-9]])
    x = x[x != 0.0]
    lam, alpha = 1.0507009873554805, 1.6732632423543772
    ref = {so.ACT_LEAKYRELU: np.maximum(0.01 * x, x), so.ACT_ELU: np.where(x >= 0, x, np.exp(x) - 1.0),
           so.ACT_SOFTPLUS: np.logaddexp(0.0, x), so.ACT_SELU: lam * np.where(x > 0, x, alpha * (np.exp(x) - 1.0))}
    for kind, r in ref.items():
        h = so._act(x, kind)
        assert np.allclose(h, r, rtol=1e-13, atol=1e-300)
        eps = 1e-6 * np.maximum(1.0, np.abs(x))
        keep = np.abs(x) > 2e-6                               # central differences must not straddle the kink at 0
        fd = (so._act(x + eps, kind) - so._act(x - eps, kind)) / (2 * eps)
        assert np.allclose(so._dact(h, kind)[keep], fd[keep], rtol=1e-6, atol=1e-9)


def test_c_port_equals_numpy_port():
    """oracle/subspace_oracle_c.c (the compiled CPU-baseline leg of bench.py) against the NumPy restatement: forward and
    log-density on ragged shapes (row / column edges of the 16x12 and 8x6 micro-kernels, out = 1 heads, every activation)."""
    from oracle import c_port
    rng = np.random.default_rng(4)
    for dims, acts, b in [([10, 20, 20, 2], [0, 0, 0], 100), ([7, 33, 50, 3], [1, 2, 3], 211), ([5, 17, 1], [2, 0], 13), ([6, 19, 23, 30, 2], [4, 5, 6, 7], 57),
                          ([128, 96, 64, 1], [1, 1, 0], 1000)]:
        table, n = so.layer_table(dims, acts)
        w = 0.2 * rng.standard_normal(n)
        p = 0.05 * rng.standard_normal((n, 4))
        x, y, z = rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b)), rng.standard_normal(4)
        for threads in (1, 0):
            assert np.allclose(c_port.forward(table, w + p @ z, x, threads), so.forward(table, w + p @ z, x), rtol=1e-12, atol=1e-13)
            assert np.isclose(c_port.logdensity(table, w, p, x, y, 0.8, z, threads), so.logdensity(table, w, p, x, y, 0.8, z), rtol=1e-12)


def test_conv_restatement_against_scipy_and_brute_force():
    """Flux 0.11.2 Conv = NNlib's TRUE convolution (flipped kernel) + bias; MaxPool; flatten -- the restatement against
    scipy.signal.convolve2d ('valid' and 'full'), a brute-force multi-channel loop with stride / pad / dilation, and the
    reverse sweep against central finite differences."""
    from scipy.signal import convolve2d
    rng = np.random.default_rng(0)
    x = rng.standard_normal((9, 7, 1, 1))
    w = rng.standard_normal((3, 2, 1, 1))
    y = so.conv_forward(x, w, np.array([0.3]), (1, 1), (0, 0), (1, 1))
    assert np.allclose(y[:, :, 0, 0], convolve2d(x[:, :, 0, 0], w[:, :, 0, 0], mode="valid") + 0.3, rtol=1e-13, atol=1e-14)
    yf = so.conv_forward(x, w, np.zeros(1), (1, 1), (2, 1), (1, 1))      # pad = K - 1 is the 'full' convolution
    assert np.allclose(yf[:, :, 0, 0], convolve2d(x[:, :, 0, 0], w[:, :, 0, 0], mode="full"), rtol=1e-13, atol=1e-14)
    x = rng.standard_normal((8, 8, 3, 2))
    w = rng.standard_normal((3, 3, 3, 4))
    b = rng.standard_normal(4)
    y = so.conv_forward(x, w, b, (2, 1), (1, 0), (1, 2))
    assert y.shape == (so.conv_out_size(8, 3, 2, 1, 1), so.conv_out_size(8, 3, 1, 0, 2), 4, 2)
    for (n, o) in ((0, 0), (1, 2), (1, 3)):
        for a in range(y.shape[0]):
            for c in range(y.shape[1]):
                acc = b[o]
                for kw in range(3):
                    for kh in range(3):
                        xi, yi = a * 2 - 1 + kw, c + 2 * kh
                        if 0 <= xi < 8 and 0 <= yi < 8:
                            acc += float(x[xi, yi, :, n] @ w[2 - kw, 2 - kh, :, o])
                assert abs(acc - y[a, c, o, n]) < 1e-12
    mp = so.maxpool_forward(x, (3, 2), (2, 1))
    assert mp.shape == (3, 7, 3, 2) and mp[1, 4, 2, 1] == x[2:5, 4:6, 2, 1].max()
    # whole-chain gradient (conv / pool / flatten / dense, every activation) against finite differences
    for spec, whc in (([("conv", (3, 3), 4, so.ACT_TANH, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (2, 2), 3, so.ACT_SIGMOID, (2, 1), (0, 1)),
                        ("flatten",), ("dense", 5, so.ACT_TANH), ("dense", 2, so.ACT_IDENTITY)], (6, 6, 2)),
                      ([("conv", (3, 2), 5, so.ACT_RELU, (1, 2), (2, 0), (2, 1)), ("maxpool", (3, 2), (2, 1)), ("flatten",),
                        ("dense", 3, so.ACT_IDENTITY)], (9, 8, 3))):
        table, n = so.conv_table(spec, whc)
        m, bsz = 3, 7
        w_swa, p = 0.3 * rng.standard_normal(n), 0.2 * rng.standard_normal((n, m))
        xx = rng.standard_normal((whc[0] * whc[1] * whc[2], bsz))
        yy = rng.standard_normal(so.forward(table, w_swa, xx).shape)
        z = 0.1 * rng.standard_normal(m)
        lp, gz, gw = so.logdensity_grad(table, w_swa, p, xx, yy, 0.7, z)
        assert np.isclose(lp, so.logdensity(table, w_swa, p, xx, yy, 0.7, z), rtol=1e-13)
        eps = 1e-6
        fd = np.array([(so.logdensity(table, w_swa, p, xx, yy, 0.7, z + eps * np.eye(m)[i]) -
                        so.logdensity(table, w_swa, p, xx, yy, 0.7, z - eps * np.eye(m)[i])) / (2 * eps) for i in range(m)])
        assert np.allclose(gz, fd, rtol=1e-6, atol=1e-8)
    # the flat-vector layout of BASELINE config 4's CNN
    cfg4 = [("conv", (3, 3), 64, 1, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (3, 3), 128, 1, (1, 1), (1, 1)), ("maxpool", (2, 2)),
            ("conv", (3, 3), 256, 1, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (3, 3), 256, 1, (1, 1), (1, 1)), ("maxpool", (2, 2)),
            ("flatten",), ("dense", 4096, 1), ("dense", 10, 0)]
    assert so.conv_table(cfg4, (32, 32, 3))[1] == 5200266


def test_flux_conv_standins_match_the_oracle():
    """flux.Conv / MaxPool / flatten (the caller-side stand-ins) == the oracle's restatement, and their layer table is the
    one the oracle derives from the same spec."""
    from subspaceinference_jl_amd import flux
    rng = np.random.default_rng(1)
    wr = np.random.default_rng(2)
    model = flux.Chain(flux.Conv((3, 2), 3, 5, flux.relu, stride=(1, 2), pad=(2, 0), dilation=(2, 1), rng=wr), flux.MaxPool((3, 2), stride=(2, 1)),
                       flux.flatten, flux.Dense(5 * 4 * 3, 3, rng=wr))
    x4 = rng.standard_normal((9, 8, 3, 6))
    spec = [("conv", (3, 2), 5, so.ACT_RELU, (1, 2), (2, 0), (2, 1)), ("maxpool", (3, 2), (2, 1)), ("flatten",), ("dense", 3, so.ACT_IDENTITY)]
    table, n = so.conv_table(spec, (9, 8, 3))
    tab2, n2 = flux.layer_table(model, (9, 8, 3))
    assert n == n2 and [tuple(r) for r in table] == [tuple(r) for r in tab2]
    wflat = flux.extract_params(flux.params(model)).astype(np.float64)
    assert np.allclose(model(x4), so.forward(table, wflat, x4.reshape((-1, 6), order="F")), rtol=1e-6, atol=1e-6)


TRAIN_DIMS, TRAIN_ACTS = [10, 20, 20, 2], [so.ACT_TANH, so.ACT_RELU, so.ACT_IDENTITY]
TRAIN_OPTS = {"descent": ("descent", 0.1), "momentum": ("momentum", 0.01, 0.9), "adam": ("adam", 0.001, 0.9, 0.999)}


def _train_fixture():
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "toy_train_steps.npz"))
    batches = [row[row >= 0] for row in d["batches"]]
    return d, batches


def test_training_step_restatement():
    """oracle.train_step (src/subspace_construction.jl:39-43; Flux 0.11.2 Descent / Momentum / ADAM): the committed fixture,
    the gradient against finite differences, closed forms of the first steps, Float32 state with Float64 arithmetic."""
    d, batches = _train_fixture()
    table, n = so.layer_table(TRAIN_DIMS, TRAIN_ACTS)
    x, y = d["X"], d["Y"]
    for name, opt in TRAIN_OPTS.items():
        w, st = d["w0"].copy(), so.optimiser_state(n, opt)
        losses = [so.train_step(table, w, st, x[:, ids], y[:, ids], opt) for ids in batches]
        assert w.dtype == np.float32 and st["m"].dtype == np.float32
        assert np.array_equal(w, d[name + "_w"]) and np.array_equal(np.array(losses), d[name + "_loss"])
        assert np.array_equal(st["m"], d[name + "_m"]) and np.array_equal(st["v"], d[name + "_v"])
        assert losses[-2] < losses[0]                                           # it trains
    assert np.allclose(d["adam_bp"], [0.9 ** 13, 0.999 ** 13], rtol=1e-13)      # beta powers start at beta: beta^(t+1) after t steps
    # gradient of the mse cost against central differences
    w64 = d["w0"].astype(np.float64)
    ids = batches[0]
    loss, g = so.mse_value_and_grad(table, w64, x[:, ids], y[:, ids])
    assert np.isclose(loss, np.mean((so.forward(table, w64, x[:, ids]) - y[:, ids]) ** 2), rtol=1e-14)
    for i in (0, 57, 230, 640, 681):
        e = np.zeros(n)
        e[i] = 1e-6
        fd = (so.mse_value_and_grad(table, w64 + e, x[:, ids], y[:, ids])[0] - so.mse_value_and_grad(table, w64 - e, x[:, ids], y[:, ids])[0]) / 2e-6
        assert abs(fd - g[i]) < 1e-8 * max(1.0, abs(g[i]))
    # closed forms: Descent x - eta g; Momentum's first step eta g; ADAM's first step eta * g / (|g| + eps') ~ eta sign(g)
    g32 = np.array([0.5, -2.0, 1e-3])
    for opt, want in ((("descent", 0.1), -0.1 * g32), (("momentum", 0.01, 0.9), -0.01 * g32),
                      (("adam", 0.001, 0.9, 0.999), -0.001 * np.sign(g32))):
        w, st = np.zeros(3, dtype=np.float32), so.optimiser_state(3, opt)
        so.apply_update(w, st, g32.copy(), opt)
        assert np.allclose(w, want, rtol=1e-4)
    # second Momentum step: v2 = rho v1 - eta g with v1 ROUNDED to Float32
    w, st = np.zeros(3, dtype=np.float32), so.optimiser_state(3, ("momentum", 0.01, 0.9))
    so.apply_update(w, st, g32.copy(), ("momentum", 0.01, 0.9))
    v1 = st["m"].copy()
    so.apply_update(w, st, g32.copy(), ("momentum", 0.01, 0.9))
    assert np.array_equal(st["m"], (0.9 * v1.astype(np.float64) - 0.01 * g32).astype(np.float32))


def test_flux_standin_optimisers_are_pinned_to_the_oracle():
    """the Python host's stand-ins for Flux's optimisers (product code, flux.py) against the oracle's restatement on the
    committed fixture: same Float32 weights after 12 steps, bit for bit."""
    from subspaceinference_jl_amd import flux
    d, batches = _train_fixture()
    x, y = d["X"], d["Y"]
    mk = {"descent": lambda: flux.Descent(0.1), "momentum": lambda: flux.Momentum(0.01, 0.9), "adam": lambda: flux.ADAM(0.001, (0.9, 0.999))}
    for name in TRAIN_OPTS:
        m = flux.Chain(flux.Dense(10, 20, flux.tanh), flux.Dense(20, 20, flux.relu), flux.Dense(20, 2))
        flux.load_flat(m, d["w0"])
        opt, ps = mk[name](), flux.params(m)
        losses = []
        for ids in batches:
            loss, gs = flux.gradient(flux.mse, m, x[:, ids], y[:, ids])
            flux.update(opt, ps, gs)
            losses.append(loss)
        assert np.allclose(losses, d[name + "_loss"], rtol=1e-12)
        assert np.allclose(flux.extract_params(ps), d[name + "_w"], rtol=0, atol=2e-7), name   # <= an ulp or two of Float32


def test_maxpool_gradient_goes_to_one_element_per_window():
    """NNlib's ∇maxpool routes a window's gradient to the FIRST input that is ≈ its maximum (kw
    fastest), not to every tied input.  Checked against an independent scalar loop, on inputs with exact ties."""
    rng = np.random.default_rng(0)
    wi, hi, c, b = 6, 4, 2, 3
    x4 = rng.integers(0, 3, size=(wi, hi, c, b)).astype(np.float64)      # many exact ties
    x4[:, :, 0, 0] = 0.25                                                 # a constant image channel
    row = ("maxpool", (2, 2), c, (wi, hi), (2, 2))
    h_in = x4.reshape((-1, b), order="F")
    h_out = so._layer_forward(row, None, h_in)
    g_out = rng.standard_normal(h_out.shape)
    gx = so._layer_backward(row, None, h_in, h_out, g_out, None).reshape(x4.shape, order="F")
    y4, g4 = h_out.reshape((3, 2, c, b), order="F"), g_out.reshape((3, 2, c, b), order="F")
    ref = np.zeros_like(x4)
    for n in range(b):
        for ch in range(c):
            for ho in range(2):
                for wo in range(3):
                    done = False
                    for kh in range(2):
                        for kw in range(2):
                            if not done and np.isclose(y4[wo, ho, ch, n], x4[2 * wo + kw, 2 * ho + kh, ch, n], rtol=1.4901161193847656e-08, atol=0):
                                ref[2 * wo + kw, 2 * ho + kh, ch, n] += g4[wo, ho, ch, n]
                                done = True
    assert np.array_equal(gx, ref)
    assert np.isclose(gx.sum(), g_out.sum())                              # every window's gradient lands exactly once
    assert np.array_equal(gx[0::2, 0::2, 0, 0], g4[:, :, 0, 0]) and gx[1::2, :, 0, 0].sum() == 0.0   # constant channel: first element


def test_training_step_float32_data():
    """Float32 (X, Y) with the Float32 model: the oracle's pass stays Float32 line by line (as Julia's would), reproduces the
    committed fixture, lands within Float32 rounding of the Float64 pass, and the optimisers round their step into the
    Float32 gradient array before the Float32 subtraction."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "toy_train_steps_f32.npz"))
    d64, batches = _train_fixture()
    table, n = so.layer_table(TRAIN_DIMS, TRAIN_ACTS)
    x, y = d["X"], d["Y"]
    assert x.dtype == np.float32 and y.dtype == np.float32 and np.array_equal(d["batches"], d64["batches"])
    loss, g = so.mse_value_and_grad(table, d["w0"], x[:, batches[0]], y[:, batches[0]])
    assert g.dtype == np.float32 and isinstance(loss, float)
    _, g64 = so.mse_value_and_grad(table, d["w0"].astype(np.float64), x[:, batches[0]].astype(np.float64), y[:, batches[0]].astype(np.float64))
    assert np.allclose(g, g64, rtol=0, atol=2e-6 * np.abs(g64).max())
    for name, opt in TRAIN_OPTS.items():
        w, st = d["w0"].copy(), so.optimiser_state(n, opt)
        losses = [so.train_step(table, w, st, x[:, ids], y[:, ids], opt) for ids in batches]
        assert np.array_equal(w, d[name + "_w"]) and np.array_equal(np.array(losses), d[name + "_loss"])
        # the two precisions train the same model: 12 steps apart by Float32 rounding only
        assert np.allclose(w, d64[name + "_w"], rtol=0, atol=2e-5) and np.allclose(losses, d64[name + "_loss"], rtol=1e-5)
    # the rounding rule of `x .-= apply!(...)` with a Float32 gradient: one Descent step by hand
    w = d["w0"].copy()
    _, g = so.mse_value_and_grad(table, w, x[:, batches[0]], y[:, batches[0]])
    so.apply_update(w, so.optimiser_state(n, ("descent", 0.1)), g, ("descent", 0.1))
    assert np.array_equal(w, d["w0"] - (g.astype(np.float64) * 0.1).astype(np.float32))


def test_reference_fixtures_when_present():
    """`tests/golden/make_golden_reference.jl` (UNEXECUTED here: no Julia) writes `*_reference.npz` from the REAL package.
    When a maintainer has run it and committed the files, this test holds the oracle against them -- the moment the parity
    of this repository stops being 'unpinned'.  Without the files it is skipped."""
    gold = os.path.join(os.path.dirname(__file__), "golden")
    names = ["toy_construct_k12_reference.npz", "toy_density_rwmh_reference.npz", "toy_train_steps_reference.npz",
             "toy_construct_k1000_reference.npz", "toy_train_steps_f32_reference.npz"]
    present = [n for n in names if os.path.exists(os.path.join(gold, n))]
    if not present:
        pytest.skip("no *_reference.npz committed (the reference cannot be run in this environment)")
    if names[0] in present:
        ref, mine = np.load(os.path.join(gold, names[0])), np.load(os.path.join(gold, "toy_construct_k12.npz"))
        assert np.array_equal(ref["W_swa"], mine["W_swa"]) and np.array_equal(ref["A"], mine["A"])     # same three rounded ops
        assert np.allclose(ref["s"], mine["s"], rtol=1e-8)
        sign = np.sign(np.sum(ref["P"] * mine["P"], axis=0))
        assert np.allclose(ref["P"] * sign, mine["P"], rtol=1e-4, atol=1e-10)                           # north_star: rtol 1e-4, up to sign
    if names[1] in present:
        ref, mine = np.load(os.path.join(gold, names[1])), np.load(os.path.join(gold, "toy_density_rwmh.npz"))
        assert np.allclose(ref["lp"], mine["lp"], rtol=1e-10) and np.allclose(ref["Yhat0"], mine["Yhat0"], rtol=1e-10, atol=1e-12)
        table, _ = so.layer_table([10, 20, 20, 2], [0, 0, 0])
        _, g1, _ = so.logdensity_grad(table, mine["W_swa"], mine["P"], mine["X"], mine["Y"], 1.0, mine["Z"][:, 1])
        assert np.allclose(ref["grad1"], g1, rtol=1e-8)
    if names[3] in present:   # the README toy at its real shape, K = 1000 > N = 682
        ref, mine = np.load(os.path.join(gold, names[3])), np.load(os.path.join(gold, "toy_construct_k1000.npz"))
        assert np.array_equal(ref["W_swa"], mine["W_swa"]) and np.allclose(ref["s"], mine["s"][:3], rtol=1e-8)
        sign = np.sign(np.sum(ref["P"] * mine["P"], axis=0))
        assert np.allclose(ref["P"] * sign, mine["P"], rtol=1e-4, atol=1e-10)
    if names[4] in present:   # the all-Float32 pass: sgemm / mean sum in the BLAS's and Julia's order, not NumPy's
        ref, mine = np.load(os.path.join(gold, names[4])), np.load(os.path.join(gold, "toy_train_steps_f32.npz"))
        for name in ("descent", "momentum", "adam"):
            assert np.allclose(ref[name + "_loss"], mine[name + "_loss"], rtol=1e-5)
            assert np.allclose(ref[name + "_w"], mine[name + "_w"], rtol=0, atol=5e-6)
    if names[2] in present:
        ref, mine = np.load(os.path.join(gold, names[2])), np.load(os.path.join(gold, "toy_train_steps.npz"))
        for name in ("descent", "momentum", "adam"):
            assert np.allclose(ref[name + "_loss"], mine[name + "_loss"], rtol=1e-6)
            assert np.allclose(ref[name + "_w"], mine[name + "_w"], rtol=0, atol=4e-7)   # Float32 weights, BLAS summation order
