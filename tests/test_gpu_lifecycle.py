"""Context life cycle: create / use every family of entry points / destroy, many times -- the device memory the library
holds (incl. the buffers it caches between calls: staging of the output map, Gram partials, conv scratch) goes back."""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu


def test_no_device_memory_is_left_behind(si):
    import torch
    rng = np.random.default_rng(0)

    def free_mb():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0] / 2 ** 20

    spec = [("conv", (3, 3), 8, so.ACT_RELU, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("flatten",), ("dense", 10, so.ACT_IDENTITY)]
    table, n = so.conv_table(spec, (8, 8, 3))
    w = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, 4)))
    x = np.asfortranarray(rng.standard_normal((8 * 8 * 3, 64)))
    y = np.asfortranarray(rng.standard_normal((10, 64)))
    dims = [(64, 300, 1, 0, 64 * 300), (300, 1, 0, 64 * 300 + 300, 64 * 300 + 300 + 300)]
    nd = 64 * 300 + 300 + 300 + 1
    wd = rng.standard_normal(nd)
    pd = np.asfortranarray(rng.standard_normal((nd, 5)))
    xd = np.asfortranarray(rng.standard_normal((64, 500)))
    yd = np.asfortranarray(rng.standard_normal((1, 500)))
    torch.zeros(1, device="cuda")
    base = None
    for it in range(24):
        ctx = si.Context(0)
        ctx.infer_setup(table, n, 4, w, p, x, y, 1.0)
        ctx.sample_rwmh(3, 0.1, seed=it)
        ctx.logdensity_grad(np.zeros(4))
        ctx.reconstruct(np.asfortranarray(rng.standard_normal((4, 70))))
        ctx.infer_setup(dims, nd, 5, wd, pd, xd, yd, 1.0)
        ctx.sample_rwmh(3, 0.1, seed=it, nchains=3)
        ctx.reconstruct(np.asfortranarray(rng.standard_normal((5, 9))))
        ctx.construct_begin(nd, 12)
        for k in range(12):
            ctx.construct_push(rng.standard_normal(nd).astype(np.float32), float(k))
        ctx.construct_finish(5)
        ctx.close()
        if it == 3:
            base = free_mb()
    assert abs(free_mb() - base) < 64.0
