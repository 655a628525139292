"""si_construct_set_storage(SI_F32) -- the opt-in fp32 storage of the deviation matrix ("A/P fp64 by
default; fp32 storage is an opt-in bandwidth optimisation that must still meet rtol 1e-4").  The columns w - W_swa are formed
in fp64 and rounded once; W_swa, the Gram matrix, the eigen-decomposition and P stay fp64.  Held against the oracle twice:
tightly against the oracle run on the ROUNDED matrix (the kernels do exactly that), and against the exact fp64 oracle with the
measured tolerance (the rounding of A perturbs s by ~1e-8 and P by ~1e-7 of its scale; north_star asks 1e-4)."""
import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu


def _snaps(n, k, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    w0 = rng.standard_normal(n)
    return [(w0 + c).astype(dtype) for c in np.cumsum(0.01 * rng.standard_normal((k, n)), axis=0)]


def _align(p, ref):
    return p * np.sign(np.sum(p * ref, axis=0))[None, :]


@pytest.mark.parametrize("n,k,m,how", [(682, 12, 3, "host"), (4099, 100, 20, "host"), (20001, 200, 40, "batch"),
                                       (1047361, 30, 20, "dev"), (300, 500, 5, "host"), (5000, 37, 7, "batch"),
                                       # wide subspaces, K <= 128: the fp32 slab-stream projection on the matrix cores
                                       (5003, 128, 64, "batch"), (4099, 100, 40, "host"), (33, 50, 33, "host"),
                                       (70001, 73, 70, "batch"), (1000, 128, 100, "batch")])
def test_fp32_stored_deviation_matrix(si, gpu_ctx, n, k, m, how):
    import torch
    from subspaceinference_jl_amd import _capi
    snaps = _snaps(n, k, seed=n + k)
    ns = [float(1 + j // 3) for j in range(k)]
    w_ref, a_ref = so.construct_stream(snaps, ns)
    a32 = a_ref.astype(np.float32).astype(np.float64)        # what the device stores
    gpu_ctx.construct_begin(n, k)
    gpu_ctx.construct_set_storage(_capi.SI_F32)
    if how == "host":
        for w, nn in zip(snaps, ns):
            gpu_ctx.construct_push(w, nn)
    else:
        ld = n + (n & 1)
        host = np.zeros((k, ld), dtype=np.float32)
        host[:, :n] = np.stack(snaps)
        dev = torch.from_numpy(host).cuda()
        if how == "batch":
            gpu_ctx.construct_push_batch_dev(dev.data_ptr(), 0, ld, ns)
        else:
            for j in range(k):
                gpu_ctx.construct_push_dev(dev.data_ptr() + 4 * ld * j, 0, ns[j])
        gpu_ctx.synchronize()
    assert np.array_equal(gpu_ctx.construct_get_A(0, k), a32)          # one rounding of the fp64 column, bit for bit
    w_swa, p, s, kk = gpu_ctx.construct_finish(m)
    assert kk == k and np.array_equal(w_swa, w_ref)                    # the mean is untouched by the storage option
    p32, s32 = so.projection_from_A(a32, m)
    assert np.allclose(s, s32[:m], rtol=1e-9)
    assert np.allclose(_align(p, p32), p32, rtol=1e-6, atol=1e-9 * np.abs(p32).max())
    p64, s64 = so.projection_from_A(a_ref, m)
    assert np.allclose(s, s64[:m], rtol=1e-6)                          # measured ~1e-8
    assert np.max(np.abs(_align(p, p64) - p64)) <= 1e-5 * np.abs(p64).max()   # measured ~1e-7; north_star 1e-4
    # the next construction on the same ctx is fp64 again unless asked otherwise
    gpu_ctx.construct_begin(n, min(k, 8))
    for w, nn in zip(snaps[:min(k, 8)], ns):
        gpu_ctx.construct_push(w, nn)
    assert np.array_equal(gpu_ctx.construct_get_A(0, min(k, 8)), a_ref[:, :min(k, 8)])


def test_fp32_storage_column_shift_and_errors(si, gpu_ctx):
    from subspaceinference_jl_amd import _capi
    n, k, mc = 3001, 20, 6
    snaps = _snaps(n, k, seed=9)
    ns = [float(1 + j // 2) for j in range(k)]
    _, a_ref = so.construct_stream(snaps, ns)
    gpu_ctx.construct_begin(n, k, mc)
    gpu_ctx.construct_set_storage(_capi.SI_F32)
    for w, nn in zip(snaps, ns):
        gpu_ctx.construct_push(w, nn)
    with pytest.raises(si.SubspaceError):
        gpu_ctx.construct_set_storage(_capi.SI_F64)                    # only before the first push
    _, p, s, kk = gpu_ctx.construct_finish(3)
    a_last = a_ref[:, -mc:].astype(np.float32).astype(np.float64)
    assert kk == mc and np.allclose(s, np.linalg.svd(a_last, compute_uv=False)[:3], rtol=1e-9)


def test_fp32_storage_through_the_api(si, gpu_ctx):
    from subspaceinference_jl_amd import flux
    rng = np.random.default_rng(0)
    x, y = rng.random((10, 100)), rng.random((2, 100))
    res = {}
    for st in ("f64", "f32"):
        wr = np.random.default_rng(2)
        m = flux.Chain(flux.Dense(10, 20, flux.tanh, rng=wr), flux.Dense(20, 20, flux.relu, rng=wr), flux.Dense(20, 2, rng=wr))
        data = flux.DataLoader(x, y, batchsize=25, shuffle=True, rng=np.random.default_rng(7))
        res[st] = si.subspace_construction(m, flux.mse, data, flux.ADAM(0.01), T=4, c=1, M=3, ctx=gpu_ctx, verbose=False, a_storage=st)
    assert np.array_equal(res["f64"][0], res["f32"][0])
    p64, p32 = res["f64"][1], res["f32"][1]
    assert np.max(np.abs(_align(p32, p64) - p64)) <= 1e-5 * np.abs(p64).max()


def test_padding_rows_stay_zero_across_storage_types_and_capacities(si, gpu_ctx):
    """N % 32 != 0; an fp32 construction at K = 100 writes over the fp64 padding rows of columns 0 .. 49; a small
    fp64 construction then re-zeroes 20 columns only; the next fp64 construction at K = 100 must not read dirty padding
    (the Gram kernels read whole 32-row slabs)."""
    from subspaceinference_jl_amd import _capi
    n, m = 1003, 5
    ctx = gpu_ctx
    s32 = _snaps(n, 100, seed=1)
    ctx.construct_begin(n, 100)
    ctx.construct_set_storage(_capi.SI_F32)
    for i, w in enumerate(s32):
        ctx.construct_push(w, float(1 + i // 7))
    ctx.construct_finish(m)
    small = _snaps(n, 20, seed=2)
    ctx.construct_begin(n, 20)
    for i, w in enumerate(small):
        ctx.construct_push(w, float(1 + i // 7))
    ctx.construct_finish(m)
    big = _snaps(n, 100, seed=3)
    ns = [float(1 + i // 7) for i in range(100)]
    ctx.construct_begin(n, 100)
    for w, nn in zip(big, ns):
        ctx.construct_push(w, nn)
    w_swa, p, s, k = ctx.construct_finish(m)
    w_ref, a_ref = so.construct_stream(big, ns)
    p_ref, s_ref = so.projection_from_A(a_ref, m)
    assert np.array_equal(w_swa, w_ref)
    assert np.allclose(s, np.asarray(s_ref)[:m], rtol=1e-10)
    assert np.allclose(_align(p, p_ref), p_ref, rtol=1e-7, atol=1e-9 * float(np.abs(p_ref).max()))


def test_fp32_storage_allocates_the_fp32_size(si, gpu_ctx):
    """the deviation matrix is allocated by its first use, at the size of the storage type in use -- a fresh
    context's fp32 construction takes half the device memory of the fp64 one"""
    import torch
    from subspaceinference_jl_amd import _capi
    n, k = 2_000_003, 64   # fp64: 1.02 GB, fp32: 0.51 GB
    w = np.zeros(n, dtype=np.float32)

    def grown(storage):
        with si.Context(0) as c:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            c.construct_begin(n, k)
            if storage is not None:
                c.construct_set_storage(storage)
            c.construct_push(w, 1.0)
            c.synchronize()
            return free0 - torch.cuda.mem_get_info()[0]

    g64, g32 = grown(None), grown(_capi.SI_F32)
    a64 = 8 * k * ((n + 63) // 64 * 64)
    assert g64 >= a64 and g32 < g64 - 0.4 * a64, (g64, g32, a64)
