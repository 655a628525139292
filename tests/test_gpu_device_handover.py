"""GPU tests of the device-resident hand-over (si_infer_setup_dev, si_construct_result_ptr): the host-staged paths of
round 1 are the reference here -- the device paths must give the SAME BITS (same kernels, same order; only the staging
differs).  The in-library RCCL collectives are in tests/test_gpu_comm.py."""
import os

import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(dims, acts, b, m, seed):
    rng = np.random.default_rng(seed)
    table, n = so.layer_table(dims, acts)
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, m)))
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)))
    return table, n, w_swa, p, x, y


@pytest.mark.parametrize("n_odd", [False, True])
def test_infer_setup_dev_copy_and_borrow(si, gpu_ctx, n_odd):
    import torch
    dims = [7, 32, 3] if n_odd else [6, 32, 2]
    table, n, w_swa, p, x, y = _problem(dims, [1, 0], 301, 5, seed=3)
    assert (n % 2 == 1) == n_odd
    m = 5
    z = np.asfortranarray(0.2 * np.random.default_rng(1).standard_normal((m, 3)))
    gpu_ctx.infer_setup(table, n, m, w_swa, p, x, y, 1.3)
    lp_host = gpu_ctx.logdensity(z)
    assert np.allclose(lp_host, [so.logdensity(table, w_swa, p, x, y, 1.3, z[:, c]) for c in range(3)], rtol=1e-11)
    # device copies of everything; P with a padded leading dimension
    ld = (n + 63) // 64 * 64
    p_t = torch.zeros((m, ld), dtype=torch.float64, device="cuda")
    p_t[:, :n] = torch.from_numpy(np.ascontiguousarray(p.T)).cuda()
    w_t = torch.zeros(ld, dtype=torch.float64, device="cuda")
    w_t[:n] = torch.from_numpy(w_swa).cuda()
    x_t = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()   # [B, in] row-major == in x B column-major
    y_t = torch.from_numpy(np.ascontiguousarray(y.T)).cuda()
    torch.cuda.synchronize()
    for borrow in (False, True):
        gpu_ctx.infer_setup_dev(table, n, m, w_t.data_ptr(), p_t.data_ptr(), ld, x_t.data_ptr(), y_t.data_ptr(), dims[0],
                                dims[-1], 301, 1.3, borrow=borrow)
        assert np.array_equal(gpu_ctx.logdensity(z), lp_host)
        assert np.array_equal(gpu_ctx.reconstruct(z[:, :1])[:, 0], w_swa + p @ z[:, 0]) or \
            np.allclose(gpu_ctx.reconstruct(z[:, :1])[:, 0], w_swa + p @ z[:, 0], rtol=1e-13, atol=1e-15)
    # a borrowed P with an odd leading dimension would make the 16-byte row pairs misaligned: refused, loudly
    with pytest.raises(si.SubspaceError):
        gpu_ctx.infer_setup_dev(table, n, m, w_t.data_ptr(), p_t.data_ptr(), n | 1, x_t.data_ptr(), y_t.data_ptr(), dims[0],
                                dims[-1], 301, 1.3, borrow=True)
    gpu_ctx.infer_setup(table, n, m, w_swa, p, x, y, 1.3)   # leave no borrowed pointers behind in the shared ctx


def test_construct_result_ptr_hand_over(si, gpu_ctx):
    """construction -> device pointers -> a SECOND context set up from them device-to-device == host round trip."""
    import torch
    dims, acts, b, m, k = [5, 24, 1], [2, 0], 97, 3, 9
    table, n, _, _, x, y = _problem(dims, acts, b, m, seed=5)
    rng = np.random.default_rng(8)
    snaps = [(0.2 * rng.standard_normal(n)).astype(np.float32) for _ in range(k)]
    gpu_ctx.construct_begin(n, k)
    for i, w in enumerate(snaps):
        gpu_ctx.construct_push(w, float(1 + i // 2))
    w_swa, p, s, _ = gpu_ctx.construct_finish(m)
    wptr, pptr, ld, mm = gpu_ctx.construct_result_ptr()
    assert mm == m and ld >= n and ld % 64 == 0
    wg, pg, sg = gpu_ctx.construct_get_result()   # the same device buffers, read back through the ABI
    assert np.array_equal(pg, p) and np.array_equal(wg, w_swa) and np.array_equal(sg, s)
    z = np.asfortranarray(0.1 * rng.standard_normal((m, 2)))
    x_t = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    y_t = torch.from_numpy(np.ascontiguousarray(y.T)).cuda()
    torch.cuda.synchronize()
    with si.Context(0) as other:
        other.infer_setup(table, n, m, w_swa, p, x, y, 0.7)
        lp_host = other.logdensity(z)
        other.infer_setup_dev(table, n, m, wptr, pptr, ld, x_t.data_ptr(), y_t.data_ptr(), dims[0], dims[-1], b, 0.7, borrow=True)
        assert np.array_equal(other.logdensity(z), lp_host)


def test_open_session_blocks_the_shared_buffers(si, gpu_ctx):
    """a density call between step_eval and step_accept would overwrite the pending proposal / SSE."""
    table, n, w_swa, p, x, y = _problem([4, 10, 1], [1, 0], 50, 2, seed=9)
    gpu_ctx.infer_setup(table, n, 2, w_swa, p, x, y, 1.0)
    z = np.zeros((2, 1), order="F")
    gpu_ctx.rwmh_begin(3, 0.1, 1)
    gpu_ctx.rwmh_step_eval()
    for call in (lambda: gpu_ctx.logdensity(z), lambda: gpu_ctx.logdensity_grad(z[:, 0]), lambda: gpu_ctx.forward(z[:, 0]),
                 lambda: gpu_ctx.predict(z, x), lambda: gpu_ctx.sample_rwmh(2, 0.1, 1)):
        with pytest.raises(si.SubspaceError) as e:
            call()
        assert e.value.code == si._capi.SI_ERR_STATE
    gpu_ctx.rwmh_abort()
    assert np.isfinite(gpu_ctx.logdensity(z)[0])
    with pytest.raises(si.SubspaceError):
        gpu_ctx.rwmh_step_eval()   # aborted: nothing open
