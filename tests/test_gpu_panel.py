"""K5 for a wide first layer with a short reduction (csrc/kernels_gemm_panel.hip): in <= 128, persistent workgroups that keep
their X operands in registers and stream W through an LDS ring.  launch_dense_f64 routes single-chain layers of that class to it;
the same layer of STACKED chains (grid.y = chain slot) runs dense_f64_kernel.  Both must give the same bits, and the oracle's number
(reference src/space_inference.jl:92-94 + the density of :88-96).  (tools/panel_bench.hip compares whole layer outputs of the two
kernels: profiles/r05_panel_edge_shapes.log.)
"""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu

# (dims, acts, B): every first layer is inside dense_panel_applies -- ceil(out / 16) * ceil(B / 128) >= 16 * 512 tiles
CASES = [
    ([128, 960, 64, 1], [1, 1, 0], 20000),    # cfg2's first layer (BASELINE.json configs[1]) on a fifth of its batch
    ([48, 130, 64, 1], [2, 1, 0], 120001),     # a ragged last feature tile (130 = 8 x 16 + 2), a ragged last panel, tanh
    ([16, 1024, 32, 2], [3, 1, 0], 17000),     # the shortest reduction and the widest layer of the class, sigmoid
    ([112, 256, 40, 1], [0, 2, 0], 66000),    # identity epilogue, 7 k tiles
    ([13, 512, 24, 1], [1, 1, 0], 33000),     # a reduction that is not a multiple of 4: zero-padded k rows (dense_f64_kernel: its ragged k tile)
    ([100, 300, 32, 1], [2, 1, 0], 56000),    # 100 = 6 k tiles + 4, 300 = 18 feature tiles + 12
    ([1, 128, 16, 1], [1, 1, 0], 132000),      # a single input feature
]


@pytest.mark.parametrize("dims,acts,b", CASES)
def test_panel_first_layer_same_bits_as_the_tile_kernel(si, gpu_ctx, dims, acts, b):
    m = 5
    table, n = so.layer_table(dims, acts)
    rng = np.random.default_rng(sum(dims) + b)
    x, y = rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b))
    w, p = 0.3 * rng.standard_normal(n), 0.05 * rng.standard_normal((n, m))
    gpu_ctx.infer_setup(table, n, m, w, p, x, y, 0.9)
    zz = 0.3 * np.random.default_rng(1).standard_normal((m, 2))
    single = gpu_ctx.logdensity(zz[:, :1])     # one chain: the panel kernel takes layer 1
    stacked = gpu_ctx.logdensity(zz)           # two chains in one launch: dense_f64_kernel takes it
    assert single[0] == stacked[0]
    ref = np.array([so.logdensity(table, w, p, x, y, 0.9, zz[:, j]) for j in range(2)])
    assert np.allclose(stacked, ref, rtol=1e-11)


def test_panel_layer_in_a_chain(si, gpu_ctx):
    """a short RWMH chain on a model whose first layer is in the class equals the oracle's chain on the same Philox stream"""
    dims, acts, b, m = [32, 512, 48, 1], [1, 1, 0], 33000, 4
    table, n = so.layer_table(dims, acts)
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal((dims[0], b)), rng.standard_normal((dims[-1], b))
    w, p = 0.3 * rng.standard_normal(n), 0.05 * rng.standard_normal((n, m))
    gpu_ctx.infer_setup(table, n, m, w, p, x, y, 1.0)
    z, lp, acc = gpu_ctx.sample_rwmh(12, 0.05, seed=9)
    zr, lpr, _, _ = so.sub_inference(table, x, y, w, p, 0.05, 1.0, 12, seed=9, chain=0)
    assert np.allclose(np.asarray(z).reshape(zr.shape), zr, rtol=1e-9, atol=1e-12) and np.allclose(np.asarray(lp).ravel(), lpr, rtol=1e-10)
