"""Round-3 boundary work, against the oracle and the entry points it replaces:
  * si_construct_push is pipelined (pinned double buffer, no synchronisation per push): W_swa / A stay BIT-EXACT
    (src/subspace_construction.jl:45-52), also when the caller overwrites its buffer right after the call returns;
  * si_sample_rwmh_weights streams the output map (src/space_inference.jl:125) while the chain runs: every weight sample
    equals si_reconstruct of the returned z, bit for bit, and the chain itself is unchanged."""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,dtype", [(682, np.float32), (1047361, np.float32), (300001, np.float64), (5, np.float32)])
def test_pipelined_host_push_is_bit_exact(gpu_ctx, n, dtype):
    rng = np.random.default_rng(n)
    k = 9
    snaps = [rng.standard_normal(n).astype(dtype) for _ in range(k)]
    ns = [float(1 + i // 2) for i in range(k)]
    buf = np.empty(n, dtype=dtype)           # ONE caller buffer, overwritten between pushes like Julia's extract_params result
    gpu_ctx.construct_begin(n, k)
    for w, nn in zip(snaps, ns):
        buf[:] = w
        gpu_ctx.construct_push(buf, nn)
        buf[:] = np.nan                      # the call has returned: the library must not read the caller's memory any more
    a = gpu_ctx.construct_get_A(0, k)
    w_swa, _, _, kk = gpu_ctx.construct_finish(min(3, n))
    w_ref, a_ref = so.construct_stream(snaps, ns)
    assert kk == k and np.array_equal(w_swa, w_ref) and np.array_equal(a, a_ref)
    with pytest.raises(Exception):
        gpu_ctx.construct_push(buf, 1.0)     # K_capacity exhausted: refused before anything is queued
    # a different dtype / size right after: the staging is re-made
    gpu_ctx.construct_begin(7, 2)
    gpu_ctx.construct_push(np.arange(7, dtype=np.float64), 1.0)
    gpu_ctx.construct_push(np.ones(7, dtype=np.float32), 1.0)
    assert np.array_equal(gpu_ctx.construct_get_A(1, 1)[:, 0], 1.0 - (np.arange(7) / 2 + 1.0) / 2)


def _problem(dims, acts, b, m, seed):
    rng = np.random.default_rng(seed)
    table, n = so.layer_table(dims, acts)
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, m)))
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)))
    return table, n, w_swa, p, x, y


@pytest.mark.parametrize("dims,b,m,itr,nchains", [([10, 20, 20, 2], 100, 3, 10, 1), ([7, 33, 1], 257, 5, 37, 3),
                                                  ([6, 16, 3], 64, 4, 3, 2), ([5, 9, 2], 31, 2, 1, 1)])
def test_streamed_output_map_equals_reconstruct(si, gpu_ctx, dims, b, m, itr, nchains):
    acts = [1] * (len(dims) - 2) + [0]
    table, n, w_swa, p, x, y = _problem(dims, acts, b, m, seed=itr)
    gpu_ctx.infer_setup(table, n, m, w_swa, p, x, y, 0.8)
    z0, lp0, acc0 = gpu_ctx.sample_rwmh(itr, 0.2, seed=11, chain_id0=2, nchains=nchains)
    z, lp, acc, w = gpu_ctx.sample_rwmh_weights(itr, 0.2, seed=11, chain_id0=2, nchains=nchains)
    assert np.array_equal(z, z0) and np.array_equal(lp, lp0) and np.array_equal(acc, acc0)   # the chain is unchanged
    assert w.shape == (n, itr, nchains)
    for c in range(nchains):
        assert np.array_equal(w[:, :, c], gpu_ctx.reconstruct(z[:, :, c]))                   # same bits as the K4 pass
        assert np.allclose(w[:, :, c], w_swa[:, None] + p @ z[:, :, c], rtol=1e-13, atol=1e-15)   # and the oracle's :125
    # rejected proposals repeat the previous sample (the select path, not a recomputation)
    if itr > 5 and nchains == 1:
        rep = [t for t in range(1, itr) if np.array_equal(z[:, t, 0], z[:, t - 1, 0])]
        assert all(np.array_equal(w[:, t, 0], w[:, t - 1, 0]) for t in rep)
    # the API mirror returns the reference's Vector{Vector{Float64}}
    from subspaceinference_jl_amd import flux
    mdl = flux.Chain(*[flux.Dense(i, o, flux.relu if a else flux.identity) for i, o, a in zip(dims[:-1], dims[1:], acts)])
    data = flux.DataLoader(x, y)
    chn, lpa = si.sub_inference(mdl, data, w_swa, p, σ_z=0.2, σ_m=0.8, itr=itr, M=m, ctx=gpu_ctx, seed=11, chain_id=2)
    assert len(chn) == itr and np.array_equal(np.stack(chn, axis=1), w[:, :, 0]) and np.array_equal(lpa, lp[:, 0])


def test_streamed_output_map_many_chains_falls_back_to_k4(gpu_ctx):
    """more chains than one pass of launches carries (cfg2-sized activations keep ONE slot): the current states are
    reconstructed by K4 into the ring instead -- same kernel, same bits."""
    dims, acts, b, m = [128, 960, 960, 1], [1, 1, 0], 20000, 20
    table, n, w_swa, p, x, y = _problem(dims, acts, b, m, seed=1)
    gpu_ctx.infer_setup(table, n, m, w_swa, p * 0.01, x, y, 1.0)
    z, lp, acc, w = gpu_ctx.sample_rwmh_weights(6, 0.05, seed=3, nchains=2)
    for c in range(2):
        assert np.array_equal(w[:, :, c], gpu_ctx.reconstruct(z[:, :, c]))
    z1, lp1, _, w1 = gpu_ctx.sample_rwmh_weights(6, 0.05, seed=3, nchains=1)
    assert np.array_equal(z1[:, :, 0], z[:, :, 0]) and np.array_equal(w1[:, :, 0], w[:, :, 0])   # select path == K4 path


TRAIN_DIMS, TRAIN_ACTS = [10, 20, 20, 2], [so.ACT_TANH, so.ACT_RELU, so.ACT_IDENTITY]
TRAIN_OPTS = {"descent": (0, ("descent", 0.1), (0.1, 0.0, 0.0)), "momentum": (1, ("momentum", 0.01, 0.9), (0.01, 0.9, 0.0)),
              "adam": (2, ("adam", 0.001, 0.9, 0.999), (0.001, 0.9, 0.999))}


@pytest.mark.parametrize("name", ["descent", "momentum", "adam"])
def test_device_training_step_against_the_oracle_optimiser(gpu_ctx, name):
    """si_train_step / si_train_get_opt_state (src/subspace_construction.jl:39-43 on the device) against the ORACLE's
    restatement of Flux 0.11.2's Descent / Momentum / ADAM (not against the package's own flux.py): the
    committed 12-step fixture (tests/golden/toy_train_steps.npz) and a live oracle run on other batches."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "toy_train_steps.npz"))
    batches = [row[row >= 0] for row in d["batches"]]
    kind, opt, (eta, p1, p2) = TRAIN_OPTS[name]
    table, n = so.layer_table(TRAIN_DIMS, TRAIN_ACTS)
    x, y = np.asfortranarray(d["X"]), np.asfortranarray(d["Y"])
    gpu_ctx.train_setup(table, n, d["w0"], x, y, 25, kind, eta, p1, p2)
    losses = [gpu_ctx.train_step(ids) for ids in batches]
    assert np.allclose(losses, d[name + "_loss"], rtol=1e-9)                 # the loss sees the weights of every earlier step
    w = gpu_ctx.train_get_weights()
    # Float32 weights from Float64 arithmetic: the device's gradient sums differ from NumPy's in the last bits of fp64, which
    # can move a Float32 rounding by one ulp now and then
    assert w.dtype == np.float32 and np.allclose(w, d[name + "_w"], rtol=0, atol=4e-7)
    m, v, bp = gpu_ctx.train_get_opt_state()
    if name != "descent":
        assert np.allclose(m, d[name + "_m"], rtol=1e-5, atol=1e-9)
    if name == "adam":
        assert np.allclose(v, d[name + "_v"], rtol=1e-5, atol=1e-12) and np.allclose(bp, d["adam_bp"], rtol=1e-13)
    # live: continue both from the device's state for 5 more steps on fresh batches
    rng = np.random.default_rng(77)
    wo = w.copy()
    st = {"m": m.copy(), "v": v.copy(), "bp": [bp[0], bp[1]] if name == "adam" else None}
    for _ in range(5):
        ids = rng.permutation(100)[:25]
        lo = so.train_step(table, wo, st, x[:, ids], y[:, ids], opt)
        assert np.isclose(gpu_ctx.train_step(ids), lo, rtol=1e-9)
    assert np.allclose(gpu_ctx.train_get_weights(), wo, rtol=0, atol=4e-7)


def test_maxpool_gradient_with_exact_ties(gpu_ctx):
    """constant image regions make the conv output equal at every pixel of a pooling window; NNlib's
    ∇maxpool gives the window's gradient to ONE input (the first ≈ maximum, kw fastest).  Both device routes -- the pass
    fused with act' / bias sum behind a Conv, and the plain MaxPool gradient -- against the oracle, through the density
    gradient and the training gradient."""
    whc = (8, 8, 2)
    spec = [("conv", (3, 3), 6, so.ACT_IDENTITY, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("maxpool", (2, 2)), ("flatten",),
            ("dense", 3, so.ACT_IDENTITY)]
    table, n = so.conv_table(spec, whc)
    rng = np.random.default_rng(3)
    b = 6
    x4 = rng.standard_normal(whc + (b,))
    x4[:, :, :, 0] = 0.5                    # a constant image: interior conv outputs are all equal -> ties in every window
    x4[:, 4:, :, 1] = 0.0                   # half of another one is zero: conv output = bias there
    x4[:, :, :, 2] = np.round(x4[:, :, :, 2])   # coarse values
    x = np.asfortranarray(x4.reshape((-1, b), order="F"))
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, 2)))
    y = np.asfortranarray(rng.standard_normal((3, b)))
    gpu_ctx.infer_setup(table, n, 2, w_swa, p, x, y, 0.9)
    z = np.array([0.3, -0.2])
    lpg, g = gpu_ctx.logdensity_grad(z)
    lpo, go, gwo = so.logdensity_grad(table, w_swa, p, x, y, 0.9, z)
    assert np.isclose(lpg, lpo, rtol=1e-11) and np.allclose(g, go, rtol=1e-8, atol=1e-10 * np.abs(go).max())
    # the old "every tied input" rule gives a different answer on this input (the test would not see a regression otherwise)
    hs = [x]
    for row in table:
        hs.append(so._layer_forward(row, w_swa + p @ z, hs[-1]))
    x1 = hs[1].reshape((8, 8, 6, b), order="F")
    ties = sum(int(np.sum(x1[a::2, d::2] == hs[2].reshape((4, 4, 6, b), order="F"))) for a in range(2) for d in range(2))
    assert ties > 4 * 4 * 6 * b                                   # more maxima than windows: ties exist
    w0 = w_swa.astype(np.float32)
    gpu_ctx.train_setup(table, n, w0, x, y, b, 0, 0.05)
    gpu_ctx.train_grad(np.arange(b), b)
    gt = gpu_ctx.train_grad_get()
    _, gref = so.mse_value_and_grad(table, w0.astype(np.float64), x, y)
    assert np.allclose(gt, gref, rtol=1e-8, atol=1e-10 * np.abs(gref).max())
