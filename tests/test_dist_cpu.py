"""world_size-2 `gloo` tests (CPU) of the multi-GPU host path: chain partitioning, the Gram all-reduce of the
row-sharded construction and the final gathers.  The GPU context is replaced by a test double that computes the
rank-local pieces with the ORACLE (tests may use it as the checker); the collective plumbing under test is the
product's own (subspaceinference.jl_amd/dist.py)."""
import os
import socket
import time
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeCtx:
    """Rank-local stand-in for _capi.Context: same method names, NumPy arithmetic on this rank's row block."""

    def __init__(self, a_local, w_local, eig):
        self.a, self.w, self.eig, self.g, self.refined = a_local, w_local, eig, None, False

    def construct_gram(self):
        self.g = self.a.T @ self.a

    def construct_gram_get(self):
        return self.g

    def construct_gram_set(self, g):
        self.g = np.array(g)

    def construct_needs_refine(self, m):
        lam = np.linalg.eigvalsh(self.g)[::-1]
        return (not self.refined) and lam[m - 1] <= 1e-9 * lam[0]

    def construct_refine(self):
        """second stage of the ill-conditioned route on this rank's rows: B = A V_full, the Gram of B replaces G"""
        _, v = np.linalg.eigh(self.g)
        self.vfull = v[:, ::-1]
        self.b = self.a @ self.vfull
        self.g = self.b.T @ self.b
        self.refined = True

    def construct_finish(self, m, **kw):
        if self.refined:
            import subspaceinference_jl_amd as si
            lam, w = si._capi.host_jacobi_eig_psd(self.g)
            return self.w, self.b @ w[:, :m], np.sqrt(lam[:m]), self.a.shape[1]
        lam, v = self.eig(self.g)
        v = v[:, ::-1][:, :m]
        v = v * np.sign(v[np.argmax(np.abs(v), axis=0), np.arange(m)])[None, :]
        return self.w, self.a @ v, np.sqrt(lam[::-1][:m]), self.a.shape[1]

    # --- step-wise RWMH on this rank's data block (oracle arithmetic), mirrors si_rwmh_begin .. si_rwmh_end
    def set_density(self, table, w_swa, p, x_blk, y_blk, sigma_m):
        self.den = (table, w_swa, p, x_blk, y_blk, sigma_m)

    def rwmh_begin(self, itr, sigma_z, seed, chain_id0=0, nchains=1, d_total=0):
        from oracle import philox
        self.sw = dict(itr=itr, sz=sigma_z, seed=seed, c0=chain_id0, d=d_total, t=0, z=None, lp=-np.inf, nacc=0,
                       zs=np.empty((self.den[2].shape[1], itr, 1), order="F"), lps=np.empty((itr, 1), order="F"))
        self.philox = philox

    def rwmh_step_eval(self):
        from oracle import subspace_oracle as so
        table, w_swa, p, x, y, _ = self.den
        sw = self.sw
        m = p.shape[1]
        base = np.zeros(m) if sw["z"] is None else sw["z"]
        sw["zp"] = base + sw["sz"] * self.philox.normals(sw["seed"], sw["c0"], sw["t"], m)
        r = y - so.forward(table, so.reconstruct(w_swa, p, sw["zp"]), x)
        return np.array([float(np.sum(r * r))])

    def rwmh_step_accept(self, sse_total):
        from oracle import subspace_oracle as so
        sw = self.sw
        lpp = so.lp_from_sse(float(sse_total[0]), sw["d"], self.den[5])
        if sw["t"] == 0 or -self.philox.randexp(sw["seed"], sw["c0"], sw["t"]) < lpp - sw["lp"]:
            if sw["t"] > 0:
                sw["nacc"] += 1
            sw["z"], sw["lp"] = sw["zp"], lpp
        sw["zs"][:, sw["t"], 0], sw["lps"][sw["t"], 0] = sw["z"], sw["lp"]
        sw["t"] += 1

    def rwmh_end(self):
        return self.sw["zs"], self.sw["lps"], np.array([self.sw["nacc"] / max(1, self.sw["itr"] - 1)])

    # --- data-parallel training step (mirrors si_train_grad / _grad_get / _grad_set / _apply) on a linear model
    # yhat = W x with plain gradient descent; the full-batch result is checked in closed form by the caller
    def set_linear(self, w, x_all, y_all, eta):
        self.lw, self.lx, self.ly, self.eta, self.lg = np.array(w, dtype=np.float64), x_all, y_all, eta, None

    def train_out_dim(self):
        return self.ly.shape[0]

    def train_grad(self, idx, nb_total):
        x, y = self.lx[:, idx], self.ly[:, idx]
        r = self.lw @ x - y
        self.lg = (2.0 / (y.shape[0] * nb_total)) * (r @ x.T)
        return float(np.sum(r * r))

    def train_grad_get(self):
        return self.lg.ravel(order="F").copy()

    def train_grad_set(self, g):
        self.lg = np.asarray(g).reshape(self.lw.shape, order="F")

    def train_apply(self):
        self.lw = self.lw - self.eta * self.lg
        self.lg = None

    def sample_rwmh(self, itr, sigma_z, seed, chain_id0=0, nchains=1):
        from oracle import philox
        z = np.stack([np.stack([sigma_z * philox.normals(seed, chain_id0 + c, t, 3) for t in range(itr)], axis=1)
                      for c in range(nchains)], axis=2)
        lp = -0.5 * (z ** 2).sum(axis=0)
        return z, lp, np.full(nchains, 0.5)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import subspaceinference_jl_amd as si
    from subspaceinference_jl_amd import dist as sd
    from oracle import subspace_oracle as so
    try:
        r, w = sd.init(backend="gloo")
        assert (r, w) == (rank, world)
        # the RCCL id of the in-library communicator travels through the process group: every rank ends up with rank 0's
        class Rec:
            def comm_init_rank(self, world, rank, uid):
                self.got = (world, rank, uid)
        rec = Rec()
        assert sd.comm_init(rec) == (rank, world) and rec.got[:2] == (world, rank) and len(rec.got[2]) == 128
        # a rank 0 that cannot create the id (no RCCL to dlopen) must not leave the others waiting in the broadcast:
        # every rank gets the error
        real_id = si._capi.comm_unique_id
        if rank == 0:
            def broken():
                raise si.SubspaceError("librccl.so.1 not found (test)", si._capi.SI_ERR_COMM)
            sd._capi.comm_unique_id = broken
        try:
            sd.comm_init(Rec())
            raise AssertionError("comm_init must fail on every rank")
        except si.SubspaceError as e:
            assert "rank 0 could not create the RCCL id" in str(e) and e.code == si._capi.SI_ERR_COMM
        finally:
            sd._capi.comm_unique_id = real_id
        import torch.distributed as td
        ids = [None] * world
        td.all_gather_object(ids, rec.got[2])
        assert len(set(ids)) == 1
        n, k, m = 1000 + 37, 12, 3
        rng = np.random.default_rng(0)  # same data on every rank
        snaps = [rng.standard_normal(n).astype(np.float32) for _ in range(k)]
        ns = [float(1 + i // 3) for i in range(k)]
        w_swa, a = so.construct_stream(snaps, ns)
        p_ref, s_ref = so.projection_from_A(a, m)
        r0, r1 = sd.row_shard(n, rank, world)
        ctx = FakeCtx(a[r0:r1], w_swa[r0:r1], si.host_sym_eig)
        w_full, p_full, s, kk = sd.sharded_construct_finish(ctx, m, n_total=n, gather=True)
        ok = kk == k and np.array_equal(w_full, w_swa) and np.allclose(s, s_ref[:m], rtol=1e-9)
        sign = np.sign(np.sum(p_full * p_ref, axis=0))
        ok = ok and np.allclose(p_full * sign, p_ref, rtol=1e-7, atol=1e-10)
        # the all-reduced Gram is the full one on every rank
        ok = ok and np.allclose(ctx.g, a.T @ a, rtol=1e-12)
        # ill-conditioned A (ten decades of singular-value spread): the second-stage Gram matrix is all-reduced too
        rq = np.random.default_rng(3)
        uu, _ = np.linalg.qr(rq.standard_normal((n, 8)))
        vv, _ = np.linalg.qr(rq.standard_normal((8, 8)))
        sv = np.logspace(0, -10, 8)
        a_ill = (uu * sv[None, :]) @ vv.T
        ci = FakeCtx(a_ill[r0:r1], w_swa[r0:r1], si.host_sym_eig)
        _, p_ill, s_ill, _ = sd.sharded_construct_finish(ci, 8, n_total=n, gather=True)
        ok = ok and ci.refined and np.allclose(s_ill, sv, rtol=1e-4)
        p_true = uu * sv[None, :]
        sg = np.sign(np.sum(p_ill * p_true, axis=0))
        ok = ok and np.allclose(p_ill * sg, p_true, rtol=0, atol=1e-4 * np.abs(p_true).max(axis=0)[None, :])
        # chains: 5 chains over 2 ranks, gathered == one rank running all five
        z, lp, acc = sd.sample_chains(ctx, 5, 7, 0.3, seed=9)
        zr, lpr, _ = FakeCtx(None, None, None).sample_rwmh(7, 0.3, 9, 0, 5)
        ok = ok and z.shape == (3, 7, 5) and np.array_equal(z, zr) and np.array_equal(lp, lpr) and acc.shape == (5,)
        # data-sharded density: each rank holds a column block of (X, Y); partial SSEs are all-reduced per step
        dims, acts, b, mm = [5, 12, 2], [so.ACT_TANH, so.ACT_IDENTITY], 41, 3
        table, npar = so.layer_table(dims, acts)
        r2 = np.random.default_rng(1)
        ws2, p2 = 0.3 * r2.standard_normal(npar), 0.2 * r2.standard_normal((npar, mm))
        x2, y2 = r2.standard_normal((dims[0], b)), r2.standard_normal((dims[-1], b))
        b0, b1 = sd.col_shard(b, rank, world)
        fc = FakeCtx(None, None, None)
        fc.set_density(table, ws2, p2, x2[:, b0:b1], y2[:, b0:b1], 0.9)
        zsh, lpsh, accsh = sd.sample_data_sharded(fc, 25, 0.1, seed=5, d_total=dims[-1] * b)
        zfull, lpfull, _, naccfull = so.sub_inference(table, x2, y2, ws2, p2, 0.1, 0.9, 25, seed=5)
        ok = ok and np.allclose(zsh[:, :, 0], zfull, rtol=1e-10, atol=1e-13) and np.allclose(lpsh[:, 0], lpfull, rtol=1e-11)
        ok = ok and abs(accsh[0] - naccfull / 24) < 1e-12
        # one-off broadcast of the subspace from the constructing rank
        wb, pb = sd.replicate_subspace(w_swa if rank == 0 else None, p_ref if rank == 0 else None, n, m, src=0)
        ok = ok and np.array_equal(wb, w_swa) and np.array_equal(pb, p_ref) and pb.flags.f_contiguous
        # data-parallel training: each rank takes its share of every batch; one gradient all-reduce per step
        r3 = np.random.default_rng(2)
        xl, yl, w0 = r3.standard_normal((4, 30)), r3.standard_normal((2, 30)), r3.standard_normal((2, 4))
        ft = FakeCtx(None, None, None)
        ft.set_linear(w0, xl, yl, 0.05)
        wref = w0.copy()
        for batch in (np.arange(0, 17), np.arange(17, 30), np.array([3, 9, 1])):
            c0, c1 = sd.col_shard(batch.size, rank, world)
            loss = sd.train_step_data_parallel(ft, batch[c0:c1], batch.size)
            rr = wref @ xl[:, batch] - yl[:, batch]
            ok = ok and np.isclose(loss, np.mean(rr * rr), rtol=1e-12)
            wref = wref - 0.05 * (2.0 / rr.size) * (rr @ xl[:, batch].T)
            ok = ok and np.allclose(ft.lw, wref, rtol=1e-12, atol=1e-14)
        q.put((rank, bool(ok), ""))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def test_partitions():
    sys.path.insert(0, ROOT)
    from subspaceinference_jl_amd import dist as sd
    for n in (1, 31, 32, 33, 1047361, 51138049):
        for ws in (1, 2, 3, 8):
            blocks = [sd.row_shard(n, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(ws - 1))
            assert all(b[0] % 32 == 0 or b[0] == b[1] for b in blocks)  # non-empty shards start 256-B aligned
            # the library's own partition (si_row_shard: what the Julia wrapper and the sharded entry points use) agrees
            import subspaceinference_jl_amd as si
            assert blocks == [si._capi.row_shard(n, r, ws) for r in range(ws)]
    assert [sd.chain_ids(8, r, 8) for r in range(8)] == [[i] for i in range(8)]
    assert sd.chain_ids(5, 0, 2) == [0, 1, 2] and sd.chain_ids(5, 1, 2) == [3, 4]
    assert sd.chain_ids(1, 1, 2) == []
    assert [sd.col_shard(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert sd.world() == (0, 1)
    assert np.array_equal(sd.allreduce_sum(np.eye(3)), np.eye(3))  # no process group: identity


def test_rccl_is_bound_at_run_time_and_ids_are_unique(tmp_path):
    """R1 start-up without a GPU: the library has no link-time RCCL dependency (dlopen at the first si_comm_* call); the id
    needs no ctx; the file rendezvous of dist.comm_init delivers rank 0's id to a late rank."""
    sys.path.insert(0, ROOT)
    import subprocess
    import subspaceinference_jl_amd as si
    needed = subprocess.run(["readelf", "-d", si._capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed and "libtorch" not in needed
    a, b = si._capi.comm_unique_id(), si._capi.comm_unique_id()
    assert len(a) == len(b) == si._capi.SI_COMM_ID_BYTES == 128 and a != b

    import threading
    joined = threading.Barrier(2)   # ncclCommInitRank returns only once every rank has joined: the stand-in does the same

    class Rec:
        def comm_init_rank(self, world, rank, uid):
            self.got = (world, rank, uid)
            joined.wait(timeout=20)
    from subspaceinference_jl_amd import dist as sd
    path = str(tmp_path / "id")
    # a file an EARLIER run left behind (right size, old): must not be taken for this run's id
    with open(path, "wb") as f:
        f.write(bytes([0]) + b"\x07" * 128)
    os.utime(path, (1.0, 1.0))
    r0, r1 = Rec(), Rec()
    out = {}
    t1 = threading.Thread(target=lambda: out.setdefault(1, sd.comm_init(r1, rank=1, world=2, id_file=path, timeout_s=20)))
    t1.start()
    time.sleep(0.3)                 # rank 1 is polling and has NOT accepted the stale file
    assert not hasattr(r1, "got")
    out[0] = sd.comm_init(r0, rank=0, world=2, id_file=path)
    t1.join(20)
    assert out == {0: (0, 2), 1: (1, 2)}
    assert r0.got[2] == r1.got[2] and len(r1.got[2]) == 128 and r1.got[2] != b"\x07" * 128
    assert (r0.got[:2], r1.got[:2]) == ((2, 0), (2, 1))
    assert not os.path.exists(path)   # rank 0 removed it after the join
    joined = threading.Barrier(1)
    with pytest.raises(si.SubspaceError):
        sd.comm_init(Rec(), rank=1, world=2, id_file=str(tmp_path / "never"), timeout_s=0.2)
    with pytest.raises(si.SubspaceError):
        sd.comm_init(Rec(), rank=0, world=2)          # two ranks and no way to share the id


@pytest.mark.timeout(300)
def test_gloo_world2_sharded_construct_and_chains():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, ok, tb in res:
        assert ok, "rank %d failed\n%s" % (rank, tb)
