"""The on-device training step in the CALLER's precision (si_train_setup_ex; reference src/subspace_construction.jl:39-43).

With a Float32 Flux model and Float32 (X, Y) the reference's Zygote pass is Float32 throughout; with Float64 data it is promoted
to Float64.  compute_dtype = SI_DTYPE_OF_DATA picks accordingly, SI_F32 / SI_F64 override.  The fp32 step (fp32 operands on
v_mfma_f32_32x32x2_f32, fp64 loss / head partials / sums over the batch) is held against the oracle's Float32 pass -- NumPy
float32 GEMMs, whose summation order is OpenBLAS's, as the reference's would be Julia's BLAS's: not bit-comparable by
construction.  MEASURED on MI355X (this file prints it): weights after 12 steps within 3e-7 absolute of the oracle's Float32
weights (their scale is 0.5: ~1 ulp of Float32), losses within 2e-7 relative; asserted at 2e-6 / 1e-5 as the header states.
The fp64 path must not change at all: si_train_setup_ex(Float64 data) and si_train_setup give the same bits."""
import os

import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu

DIMS, ACTS = [10, 20, 20, 2], [so.ACT_TANH, so.ACT_RELU, so.ACT_IDENTITY]
OPTS = {"descent": (0, ("descent", 0.1), (0.1, 0.0, 0.0)), "momentum": (1, ("momentum", 0.01, 0.9), (0.01, 0.9, 0.0)),
        "adam": (2, ("adam", 0.001, 0.9, 0.999), (0.001, 0.9, 0.999))}


def _fixture(name):
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    return d, [row[row >= 0] for row in d["batches"]]


@pytest.mark.parametrize("name", ["descent", "momentum", "adam"])
def test_f32_step_against_the_oracles_float32_pass(si, gpu_ctx, name):
    from subspaceinference_jl_amd import _capi
    d, batches = _fixture("toy_train_steps_f32.npz")
    kind, opt, (eta, p1, p2) = OPTS[name]
    table, n = so.layer_table(DIMS, ACTS)
    x, y = np.asfortranarray(d["X"]), np.asfortranarray(d["Y"])
    assert x.dtype == np.float32
    gpu_ctx.train_setup(table, n, d["w0"], x, y, 25, kind, eta, p1, p2)   # Float32 data => the Float32 pass
    assert gpu_ctx.train_compute_dtype() == _capi.SI_F32
    losses = np.array([gpu_ctx.train_step(ids) for ids in batches])
    w = gpu_ctx.train_get_weights()
    dw, dl = np.abs(w - d[name + "_w"]).max(), np.abs(losses / d[name + "_loss"] - 1.0).max()
    print("f32 step, %s: max |w - w_oracle32| = %.2e (scale %.2f), max rel loss difference %.2e" % (name, dw, np.abs(w).max(), dl))
    assert dw <= 2e-6 and dl <= 1e-5
    m, v, bp = gpu_ctx.train_get_opt_state()
    if name != "descent":
        assert np.allclose(m, d[name + "_m"], rtol=1e-3, atol=1e-7)
    if name == "adam":
        assert np.allclose(bp, d["adam_bp"], rtol=1e-13)
    # live: five more steps of both from the device's state, other batches
    rng = np.random.default_rng(5)
    wo = w.copy()
    st = {"m": m.copy(), "v": v.copy(), "bp": [bp[0], bp[1]] if name == "adam" else None}
    for _ in range(5):
        ids = rng.permutation(100)[:25]
        lo = so.train_step(table, wo, st, x[:, ids], y[:, ids], opt)
        assert np.isclose(gpu_ctx.train_step(ids), lo, rtol=1e-5)
    assert np.abs(gpu_ctx.train_get_weights() - wo).max() <= 3e-6


def test_precision_follows_the_data_and_can_be_overridden(si, gpu_ctx):
    """Float64 data: today's fp64 step, bit for bit (si_train_setup_ex == si_train_setup == the fp64 fixture's tolerances);
    SI_F32 on Float64 data rounds X once; SI_F64 on Float32 data widens it."""
    from subspaceinference_jl_amd import _capi
    d64, batches = _fixture("toy_train_steps.npz")
    d32, _ = _fixture("toy_train_steps_f32.npz")
    table, n = so.layer_table(DIMS, ACTS)
    kind, opt, (eta, p1, p2) = OPTS["adam"]
    x64, y64 = np.asfortranarray(d64["X"]), np.asfortranarray(d64["Y"])

    def run(x, y, cd):
        gpu_ctx.train_setup(table, n, d64["w0"], x, y, 25, kind, eta, p1, p2, compute_dtype=cd)
        dt = gpu_ctx.train_compute_dtype()
        return dt, np.array([gpu_ctx.train_step(ids) for ids in batches]), gpu_ctx.train_get_weights()

    dt_a, l_a, w_a = run(x64, y64, None)
    assert dt_a == _capi.SI_F64 and np.allclose(l_a, d64["adam_loss"], rtol=1e-9) and np.allclose(w_a, d64["adam_w"], rtol=0, atol=4e-7)
    # the plain entry point (what rounds 1-4 shipped): same bits
    arr = _capi._layer_array(table)
    w0 = np.ascontiguousarray(d64["w0"], dtype=np.float32)
    gpu_ctx._check(gpu_ctx.lib.si_train_setup(gpu_ctx.h, arr, len(table), n, _capi._ptr(w0), _capi._ptr(x64), _capi._ptr(y64), 10, 2, 100,
                                              25, kind, eta, p1, p2))
    l_b = np.array([gpu_ctx.train_step(ids) for ids in batches])
    assert np.array_equal(l_a, l_b) and np.array_equal(w_a, gpu_ctx.train_get_weights())
    # overrides
    dt_c, l_c, w_c = run(x64, y64, _capi.SI_F32)                                       # Float64 data, fp32 step: X rounded once
    assert dt_c == _capi.SI_F32 and np.abs(w_c - d32["adam_w"]).max() <= 2e-6 and np.allclose(l_c, d32["adam_loss"], rtol=1e-5)
    dt_d, l_d, w_d = run(x64.astype(np.float32), y64.astype(np.float32), _capi.SI_F64)   # Float32 data, fp64 step: widened
    assert dt_d == _capi.SI_F64 and np.allclose(l_d, d64["adam_loss"], rtol=1e-6) and np.abs(w_d - d64["adam_w"]).max() <= 2e-6
    assert not np.array_equal(l_c, l_a)


@pytest.mark.parametrize("dims,acts,b,bmax", [
    ([16, 64, 32, 1], [1, 2, 0], 512, 512),          # narrow head behind a tanh layer; whole k tiles: the LDS-DMA weight gradient
    ([7, 33, 18, 3], [2, 1, 3], 130, 100),           # odd widths, a ragged second batch: the generic weight-gradient kernel
    ([12, 40, 24, 9], [1, 1, 0], 256, 256),          # a WIDE last layer (no fused head)
    ([128, 192, 128, 1], [1, 1, 0], 4096, 4096),     # several 128 x 128 output tiles, 16 splits of the batch
    ([5, 8], [3], 64, 64),                           # a single sigmoid layer
])
def test_f32_gradient_on_other_shapes(si, gpu_ctx, dims, acts, b, bmax):
    """one Descent step = w - eta * gradient: the device's fp32 gradient against the oracle's Float32 and Float64 gradients"""
    table, n = so.layer_table(dims, acts)
    rng = np.random.default_rng(sum(dims) + b)
    x = np.asfortranarray(rng.standard_normal((dims[0], b)).astype(np.float32))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)).astype(np.float32))
    w0 = (0.3 * rng.standard_normal(n)).astype(np.float32)
    gpu_ctx.train_setup(table, n, w0, x, y, bmax, 0, 1.0)
    for ids in (np.arange(bmax), rng.permutation(b)[: max(1, bmax // 3)]):
        gpu_ctx.train_setup(table, n, w0, x, y, bmax, 0, 1.0)
        loss = gpu_ctx.train_step(ids)
        g_dev = w0.astype(np.float64) - gpu_ctx.train_get_weights().astype(np.float64)   # eta = 1: the step IS the gradient (rounded twice)
        l32, g32 = so.mse_value_and_grad(table, w0, x[:, ids], y[:, ids])
        l64, g64 = so.mse_value_and_grad(table, w0.astype(np.float64), x[:, ids].astype(np.float64), y[:, ids].astype(np.float64))
        scale = np.abs(g64).max()
        assert np.isclose(loss, l64, rtol=2e-6), (loss, l32, l64)
        # against the exact gradient: Float32 arithmetic of a pass this deep; against NumPy's Float32 pass: two Float32 passes apart
        assert np.abs(g_dev - g64).max() <= 3e-6 * scale + 1.2e-7 * np.abs(w0).max()
        assert np.abs(g_dev - g32).max() <= 6e-6 * scale + 1.2e-7 * np.abs(w0).max()


def test_f32_refused_for_conv_chains_in_training(si, gpu_ctx):
    from subspaceinference_jl_amd import _capi
    table, n = so.conv_table([("conv", (3, 3), 4, so.ACT_RELU, (1, 1), (1, 1)), ("flatten",), ("dense", 2, so.ACT_IDENTITY)], (6, 6, 1))
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal((36, 8)).astype(np.float32), rng.standard_normal((2, 8)).astype(np.float32)
    with pytest.raises(si.SubspaceError):
        gpu_ctx.train_setup(table, n, np.zeros(n, np.float32), x, y, 8, 0, 0.1)          # Float32 data would mean SI_F32
    gpu_ctx.train_setup(table, n, np.zeros(n, np.float32), x, y, 8, 0, 0.1, compute_dtype=_capi.SI_F64)
    assert gpu_ctx.train_compute_dtype() == _capi.SI_F64


def test_api_picks_the_precision_from_the_data(si):
    """subspace_construction (api.py; the Julia wrapper has the same keyword): Float32 DataLoader => the fp32 device step, and the
    result agrees with the host step of the same data (NumPy's promotion = Julia's) to Float32 accuracy"""
    from subspaceinference_jl_amd import flux
    rng = np.random.default_rng(2)
    x, y = rng.random((10, 96)).astype(np.float32), rng.random((2, 96)).astype(np.float32)

    def run(device_training, compute_dtype="auto"):
        r = np.random.default_rng(7)
        m = flux.Chain(flux.Dense(10, 20, flux.tanh, rng=r), flux.Dense(20, 20, flux.relu, rng=r), flux.Dense(20, 2, rng=r))
        data = flux.DataLoader(x, y, batchsize=32)
        return si.subspace_construction(m, flux.mse, data, flux.ADAM(0.01), T=4, c=1, M=3, verbose=False,
                                        device_training=device_training, compute_dtype=compute_dtype)

    w_host, p_host = run(False)
    w_dev, p_dev = run(True)
    w_dev64, _ = run(True, "f64")
    assert np.allclose(w_dev, w_host, rtol=0, atol=5e-6) and np.allclose(w_dev64, w_host, rtol=0, atol=5e-6)
    assert not np.array_equal(w_dev, w_dev64)
    sign = np.sign(np.sum(p_dev * p_host, axis=0))
    assert np.allclose(p_dev * sign, p_host, rtol=0, atol=2e-4 * np.abs(p_host).max())
