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
