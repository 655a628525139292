"""K5 as one launch and K6 as a persistent grid loop for NARROW Dense chains (csrc/kernels_chain_grid.hip): the model of the
reference's worked example, docs/src/nn_example.md:112-118 (2-200-50-50-50-1 on 1000 observations, sampled at :188-194 by
src/space_inference.jl:111-116), and its class.  Both forms must give the SAME BITS as the launch-per-step loop with one
launch per layer (si_set_chain_loop(ctx, 0)), which is held against the oracle here and elsewhere.

  mode 0  one launch per layer and per step            (the reference point)
  mode 2  every layer in one launch, one pass of launches per transition (chains stacked in the grid)
  mode 1  automatic: the persistent grid loop where ceil(B / tile) * nchains workgroups are resident, else mode 2's path
  modes 3 / 4  = 1 / 2 with the generic kernels only.  In modes 1 / 2 chains with a narrow head whose weight fragments fit a wave's
          registers run kernels compiled for their shapes at run time (csrc/chain_spec.inc through hiprtc; the loop then takes ONE
          grid barrier per transition and hands the weights over in fragment order): same bits again.
"""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu

NN_EXAMPLE = ([2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 3)   # docs/src/nn_example.md:112-118, M = 3 at :188

CASES = [
    NN_EXAMPLE,
    ([2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 20),   # the same model at the largest M of the docs' sweep (:236)
    ([7, 33, 18, 40, 3], [2, 1, 3, 0], 1300, 5),            # odd widths, ragged tiles, tanh / relu / sigmoid, a head of 3
    ([4, 100, 1], [1, 0], 5000, 4),                          # two layers: head slots of 64 features, 313 tiles
    ([5, 70, 4], [2, 2], 777, 6),                            # head of 4 with an activation of its own, slots of 48
    ([12, 256, 130, 2], [1, 2, 0], 2500, 7),                 # the widest member of the class; 130 -> slots past the last feature
    ([6, 40, 24, 9], [1, 1, 3], 900, 4),                     # a WIDE last layer (out = 9 > 4): no fused head, sigmoid output
    ([3, 5], [0], 4000, 2),                                  # a single layer
]


def _setup(gpu_ctx, dims, acts, b, m, seed=0, sigma=0.8):
    table, n = so.layer_table(dims, acts)
    rng = np.random.default_rng(sum(dims) + b + seed)
    x, y = rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b))
    w, p = 0.3 * rng.standard_normal(n), 0.05 * rng.standard_normal((n, m))
    gpu_ctx.infer_setup(table, n, m, w, p, x, y, sigma)
    return table, n, x, y, w, p


@pytest.mark.parametrize("dims,acts,b,m", CASES)
@pytest.mark.parametrize("nchains", [1, 8, 64])
def test_fused_and_grid_loop_equal_launch_per_step_bit_for_bit(si, gpu_ctx, dims, acts, b, m, nchains):
    _setup(gpu_ctx, dims, acts, b, m)
    itr = 60 if nchains > 8 else 200
    try:
        out, spec = {}, {}
        for mode in (0, 2, 1, 4, 3):
            gpu_ctx.set_chain_loop(mode)
            out[mode] = gpu_ctx.sample_rwmh(itr, 0.07, seed=11, chain_id0=2, nchains=nchains)
            spec[mode] = gpu_ctx.chain_kernel_info()
        for mode in (2, 1, 4, 3):
            for a, bb in zip(out[0], out[mode]):
                assert np.array_equal(a, bb), (mode, dims, nchains, spec[mode])
        assert 0.0 < out[1][2].mean() < 1.0   # the chains move and reject: both branches of the accept step ran
        assert not spec[3][0] and not spec[3][1] and not spec[4][0] and not spec[4][1]   # generic kernels when asked for
        # a narrow head and fragments that fit a wave's registers: the kernels compiled for the chain's shapes ran in modes 1 / 2
        in_class = dims[-1] <= 4 and len(dims) >= 3 and sum(-(-((o + 15) // 16) // 4) * ((i + 3) // 4 + 1) for i, o in zip(dims[:-2], dims[1:-1])) <= 96
        assert spec[2][0] == in_class, (dims, spec, in_class)   # (mode 1's LOOP is specialised only up to 1024 model outputs: four blocks of the SSE tree)
    finally:
        gpu_ctx.set_chain_loop(1)


def test_nn_example_against_the_oracle(si, gpu_ctx):
    """the worked example's model: the device-resident chain equals the oracle's on the same Philox stream"""
    dims, acts, b, m = NN_EXAMPLE
    table, n, x, y, w, p = _setup(gpu_ctx, dims, acts, b, m, sigma=1.0)
    z, lp, acc = gpu_ctx.sample_rwmh(100, 0.1, seed=5, chain_id0=0, nchains=3)
    for c in range(3):
        zr, lpr, _, _ = so.sub_inference(table, x, y, w, p, 0.1, 1.0, 100, seed=5, chain=c)
        assert np.allclose(z[:, :, c], zr, rtol=1e-9, atol=1e-12) and np.allclose(lp[:, c], lpr, rtol=1e-10)
    # the density alone, one launch for every layer, on 33 points at once
    zz = 0.3 * np.random.default_rng(1).standard_normal((m, 33))
    got = gpu_ctx.logdensity(zz)
    ref = np.array([so.logdensity(table, w, p, x, y, 1.0, zz[:, j]) for j in range(33)])
    assert np.allclose(got, ref, rtol=1e-11)
    gpu_ctx.set_chain_loop(0)
    try:
        assert np.array_equal(got, gpu_ctx.logdensity(zz))
    finally:
        gpu_ctx.set_chain_loop(1)


def test_the_specialised_kernels_are_the_ones_that_run(si, gpu_ctx):
    """nn_example: the persistent loop (few chains) and the stacked density (many) run the kernels compiled for the chain's shapes;
    a chain outside the class (wide head) or too wide for the register preload keeps the generic ones and says why"""
    dims, acts, b, m = NN_EXAMPLE
    _setup(gpu_ctx, dims, acts, b, 20, seed=1)
    gpu_ctx.set_chain_loop(1)
    gpu_ctx.sample_rwmh(30, 0.05, seed=1, nchains=2)
    d, l, msg = gpu_ctx.chain_kernel_info()
    assert l, msg
    gpu_ctx.sample_rwmh(5, 0.05, seed=1, nchains=300)     # more chains than the grid loop holds: one pass of launches per transition
    d, l, msg = gpu_ctx.chain_kernel_info()
    assert d and not l, msg
    gpu_ctx.logdensity(np.zeros((20, 7)))
    assert gpu_ctx.chain_kernel_info()[0]
    _setup(gpu_ctx, [12, 256, 130, 2], [1, 2, 0], 2500, 7)   # 256 x 130: the fragments of a wave do not fit its registers
    gpu_ctx.sample_rwmh(5, 0.05, seed=1, nchains=1)
    d, l, msg = gpu_ctx.chain_kernel_info()
    assert not d and not l and "class" in msg


def test_long_chain_stays_identical(si, gpu_ctx):
    """20 000 transitions = 40 000 grid barriers and as many hand-offs of the weight / output buffers between workgroups on
    different XCDs: one stale read anywhere changes an lp and every sample after it"""
    dims, acts, b, m = NN_EXAMPLE
    _setup(gpu_ctx, dims, acts, b, m, seed=3)
    try:
        gpu_ctx.set_chain_loop(1)
        z1, lp1, a1 = gpu_ctx.sample_rwmh(20000, 0.05, seed=2, nchains=4)
        gpu_ctx.set_chain_loop(2)
        z2, lp2, a2 = gpu_ctx.sample_rwmh(20000, 0.05, seed=2, nchains=4)
        assert np.array_equal(z1, z2) and np.array_equal(lp1, lp2) and np.array_equal(a1, a2)
    finally:
        gpu_ctx.set_chain_loop(1)


def test_output_map_and_fallbacks(si, gpu_ctx):
    """the weight samples of the loop (one K4 pass over all samples) equal the streamed output map of the launch path; with
    the prior term on, the loop steps aside (same API, the launch-per-step path with the fused density)"""
    dims, acts, b, m = NN_EXAMPLE
    _setup(gpu_ctx, dims, acts, b, m, seed=9)
    try:
        zw, lpw, aw, ww = gpu_ctx.sample_rwmh_weights(40, 0.1, seed=1, nchains=2)
        gpu_ctx.set_chain_loop(0)
        zs, lps, _, ws = gpu_ctx.sample_rwmh_weights(40, 0.1, seed=1, nchains=2)
        assert np.array_equal(zs, zw) and np.array_equal(lps, lpw) and np.array_equal(ws, ww)
        assert np.array_equal(ww[:, :, 1], gpu_ctx.reconstruct(zw[:, :, 1]))
        gpu_ctx.set_prior(2.0)
        z0, lp0, _ = gpu_ctx.sample_rwmh(30, 0.1, seed=1)
        gpu_ctx.set_chain_loop(1)
        z1, lp1, _ = gpu_ctx.sample_rwmh(30, 0.1, seed=1)
        assert np.array_equal(z0, z1) and np.array_equal(lp0, lp1) and not np.array_equal(lp0, lpw[:30, :1])
        gpu_ctx.set_prior(0.0)
    finally:
        gpu_ctx.set_prior(0.0)
        gpu_ctx.set_chain_loop(1)


def test_many_chains_stack_in_one_pass(si, gpu_ctx):
    """512 chains: K4 for all of them is one launch (grid.y), the density one launch; same bits as per-layer launches"""
    dims, acts, b, m = NN_EXAMPLE
    _setup(gpu_ctx, dims, acts, b, m, seed=4)
    try:
        gpu_ctx.set_chain_loop(1)
        z1, lp1, a1 = gpu_ctx.sample_rwmh(12, 0.05, seed=8, nchains=515)
        gpu_ctx.set_chain_loop(0)
        z0, lp0, a0 = gpu_ctx.sample_rwmh(12, 0.05, seed=8, nchains=515)
        assert np.array_equal(z1, z0) and np.array_equal(lp1, lp0) and np.array_equal(a1, a0)
    finally:
        gpu_ctx.set_chain_loop(1)


def test_chain_loop_mode_is_validated(si, gpu_ctx):
    for bad in (5, -1):
        with pytest.raises(si.SubspaceError):
            gpu_ctx.set_chain_loop(bad)
    gpu_ctx.set_chain_loop(1)
