"""GPU tests of R1, the RCCL communicator BEHIND THE C ABI (csrc/comm.hip; include/subspace_hip.h `si_comm_*`): no torch
in the collective path.  The test box has one GPU and RCCL refuses two ranks on one device, so these run communicators of
world 1 (several in one process, each its own id): every in-library collective must then be the identity and the
sharded flows must give the SAME BITS as the single-GPU entry points -- same kernels, same order, only the exchange
differs.  World 2 of the same host flows runs over the gloo test double in tests/test_dist_cpu.py."""

import numpy as np
import pytest

from oracle import subspace_oracle as so

pytestmark = pytest.mark.gpu


def _with_comm(si, tmp_path=None):
    from subspaceinference_jl_amd import dist as sd
    c = si.Context(0)
    if tmp_path is None:
        sd.comm_init(c, rank=0, world=1)
    else:   # the torch-free start-up: the id travels through a file
        sd.comm_init(c, rank=0, world=1, id_file=str(tmp_path / "rccl_id"))
    return c


def _pushed(c, n, k, snaps):
    c.construct_begin(n, k)
    for i, w in enumerate(snaps):
        c.construct_push(w, float(1 + i // 3))
    return c


def test_comm_lifecycle_and_host_values(si, tmp_path):
    c = si.Context(0)
    assert c.comm_info()[0] == 0
    for call in (c.construct_allreduce_gram, c.rwmh_allreduce_sse, c.comm_barrier, lambda: c.bcast_subspace(0, 10, 2),
                 lambda: c.construct_allgather(10), lambda: c.comm_allreduce_host([1.0])):
        with pytest.raises(si.SubspaceError) as e:
            call()
        assert e.value.code == si._capi.SI_ERR_STATE and "no communicator" in str(e.value)
    uid = si._capi.comm_unique_id()
    assert len(uid) == 128 and uid != si._capi.comm_unique_id()
    with pytest.raises(si.SubspaceError):
        c.comm_init_rank(1, 1, uid)                 # rank out of range
    c.comm_init_rank(1, 0, uid)
    w, r, ver = c.comm_info()
    assert (w, r) == (1, 0) and ver >= 20000        # NCCL-style version code of the RCCL that was bound
    with pytest.raises(si.SubspaceError):
        c.comm_init_rank(1, 0, uid)                 # already has one
    v = np.array([3.5, -1.25, 7.0])
    assert np.array_equal(c.comm_allreduce_host(v, "sum"), v) and np.array_equal(c.comm_allreduce_host(v, "max"), v)
    assert np.array_equal(c.comm_allgather_host(v), v[None, :])
    c.comm_barrier()
    c.comm_destroy()
    assert c.comm_info()[0] == 0
    c.close()
    c2 = _with_comm(si, tmp_path)                   # file rendezvous
    assert c2.comm_info()[:2] == (1, 0)
    c2.close()                                      # si_destroy releases the communicator


def test_sharded_construction_in_library_equals_single_gpu(si):
    from subspaceinference_jl_amd import dist as sd
    rng = np.random.default_rng(0)
    dims, acts, b, m, k = [6, 40, 3], [1, 0], 500, 4, 11
    table, n = so.layer_table(dims, acts)
    snaps = [(0.3 * rng.standard_normal(n)).astype(np.float32) for _ in range(k)]
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)))
    a = _pushed(si.Context(0), n, k, snaps)
    bb = _pushed(_with_comm(si), n, k, snaps)
    cc = _pushed(_with_comm(si), n, k, snaps)
    d = _with_comm(si)
    try:
        # (1) Gram all-reduced in place by the library + all-gather of the row blocks == plain finish, bit for bit
        w0, p0, s0, _ = a.construct_finish(m)
        w1, p1, s1, _ = sd.sharded_construct_finish(bb, m, n_total=n, gather=True)
        assert np.array_equal(w0, w1) and np.array_equal(p0, p1) and np.array_equal(s0, s1)
        # (2) the all-device variant: the ctx ends up holding the full finished construction
        s2 = sd.sharded_construct_finish_dev(cc, m, n)
        wg, pg, sg = cc.construct_get_result()
        assert np.array_equal(wg, w0) and np.array_equal(pg, p0) and np.array_equal(s2, s0) and np.array_equal(sg, s0)
        with pytest.raises(si.SubspaceError):
            cc.construct_push(snaps[0], 1.0)        # an assembled construction has no deviation matrix to push into
        # (3) device-to-device broadcast: root == self at world 1 (identity); a receiver-shaped ctx adopts the buffers
        bb2 = _pushed(_with_comm(si), n, k, snaps)
        bb2.construct_finish(m, want_swa=False, want_p=False)
        sd.replicate_subspace_dev(bb2, n, m, src=0)
        wb, pb, sb = bb2.construct_get_result()
        assert np.array_equal(wb, w0) and np.array_equal(pb, p0) and np.array_equal(sb, s0)
        bb2.close()
        with pytest.raises(si.SubspaceError):
            d.bcast_subspace(0, n, m)               # the root must hold a finished construction
        # (4) inference on the assembled subspace, in place; data-sharded chain through the library's own loop
        cc.infer_setup(table, n, m, None, None, x, y, 2.0)
        a.infer_setup(table, n, m, w0, p0, x, y, 2.0)
        z_ref, lp_ref, acc_ref = a.sample_rwmh(40, 0.05, seed=7, chain_id0=1, nchains=2)
        z_d, lp_d, acc_d = cc.sample_rwmh(40, 0.05, seed=7, chain_id0=1, nchains=2)
        assert np.array_equal(z_ref, z_d) and np.array_equal(lp_ref, lp_d)
        z_s, lp_s, acc_s = sd.sample_data_sharded(cc, 40, 0.05, seed=7, d_total=dims[-1] * b, chain_id0=1, nchains=2)
        assert np.array_equal(z_ref, z_s) and np.array_equal(lp_ref, lp_s) and np.array_equal(acc_ref, acc_s)
        # the step-wise form with the explicit collective between eval and accept
        cc.rwmh_begin(5, 0.05, 7, 1, 2, dims[-1] * b)
        with pytest.raises(si.SubspaceError):
            cc.rwmh_allreduce_sse()                 # nothing evaluated yet
        for _ in range(5):
            cc.rwmh_step_eval(on_device=True)
            cc.rwmh_allreduce_sse()
            cc.rwmh_step_accept(None)
        z5, lp5, _ = cc.rwmh_end()
        assert np.array_equal(z5, z_ref[:, :5]) and np.array_equal(lp5, lp_ref[:5])
        # independent chains gathered through the communicator
        zc, lpc, accc = sd.sample_chains(cc, 2, 40, 0.05, seed=7)
        z01, lp01, _ = a.sample_rwmh(40, 0.05, seed=7, chain_id0=0, nchains=2)
        assert np.array_equal(zc, z01) and np.array_equal(lpc, lp01)
    finally:
        for c in (a, bb, cc, d):
            c.close()


def test_data_parallel_step_in_library(si):
    """si_train_step_dp (gradient + SSE all-reduced by the library, one grouped launch) == si_train_step at world 1."""
    from subspaceinference_jl_amd import dist as sd, flux
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal((6, 64)), rng.standard_normal((1, 64))
    wr = np.random.default_rng(1)
    m = flux.Chain(flux.Dense(6, 40, flux.relu, rng=wr), flux.Dense(40, 1, rng=wr))
    table, n = flux.layer_table(m)
    w0 = flux.extract_params(flux.params(m))
    a, b = si.Context(0), _with_comm(si)
    try:
        for c in (a, b):
            c.train_setup(table, n, w0, x, y, 64, *flux.device_optimiser(flux.Momentum(0.01, 0.9)))
        with pytest.raises(si.SubspaceError):
            b.train_allreduce_grad()                # no gradient pending
        for ids in (np.arange(0, 64), np.arange(10, 50)):
            la = a.train_step(ids)
            lb = sd.train_step_data_parallel(b, ids, ids.size)
            assert np.isclose(la, lb, rtol=1e-12), (la, lb)
        assert np.array_equal(a.train_get_weights(), b.train_get_weights())
        # the pieces: grad -> all-reduce (returns the summed SSE) -> apply
        sse_loc = b.train_grad(np.arange(5, 25), 20)
        assert b.train_allreduce_grad() == sse_loc
        b.train_apply()
        a.train_step(np.arange(5, 25))
        assert np.array_equal(a.train_get_weights(), b.train_get_weights())
    finally:
        a.close()
        b.close()


def test_api_construction_uses_the_communicator(si):
    """api.subspace_construction with a ctx that carries a communicator takes the data-parallel training step through
    si_train_step_dp and returns the single-process result (world 1)."""
    from subspaceinference_jl_amd import flux
    rng = np.random.default_rng(0)
    x, y = rng.random((10, 100)), rng.random((2, 100))

    def run(ctx):
        wr = np.random.default_rng(2)
        m = flux.Chain(flux.Dense(10, 20, flux.tanh, rng=wr), flux.Dense(20, 20, flux.relu, rng=wr), flux.Dense(20, 2, rng=wr))
        data = flux.DataLoader(x, y, batchsize=25, shuffle=True, rng=np.random.default_rng(7))
        return si.subspace_construction(m, flux.mse, data, flux.ADAM(0.01), T=4, c=2, M=3, ctx=ctx, verbose=False,
                                        device_training=True, data_parallel=si.dist._has_comm(ctx))
    a, b = si.Context(0), _with_comm(si)
    try:
        (w0, p0), (w1, p1) = run(a), run(b)
        assert np.array_equal(w0, w1) and np.array_equal(p0, p1)
    finally:
        a.close()
        b.close()
