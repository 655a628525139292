"""One process per GPU.  The collectives of the hot path live in the LIBRARY (csrc/comm.hip: an RCCL communicator per
si_ctx behind the C ABI, `si_comm_*`, in place on the library's device buffers, stream-ordered) -- this module only
calls them, exactly as the Julia wrapper does.  The reference is single-process (zero collectives); what
is sharded is what the hot path allows:

  * independent chains  -- chain c runs on rank c mod world with W_swa / P / X / Y replicated (one si_bcast_subspace);
                           NO per-step collective; one gather of (Z, lp) at the end.  bench.py's weak-scaling mode.
  * row-sharded construction -- every rank holds a row block of w, W_swa, A and P.  K1 and K3 are row-local; the only
                           exchange is ONE all-reduce(sum) of the K x K fp64 Gram matrix (si_construct_allreduce_gram).
  * data-sharded density -- every rank holds a column block of (X, Y); ONE all-reduce of 8 * nchains bytes per transition
                           (si_sample_rwmh_sharded).
  * data-parallel training step -- ONE all-reduce of the N-double gradient per step (si_train_step_dp).

torch.distributed appears in two roles only: (1) shipping the 128-byte RCCL id when the ranks were started by torchrun
(`comm_init`; a file rendezvous works without torch), and (2) as the CPU TEST DOUBLE: under a `gloo` group and a ctx
without a communicator (tests/test_dist_cpu.py's stand-in contexts, tools/bench_rehearsal.py) the same
functions stage the small arrays through the host.  No arithmetic happens here.
"""
import os
import time

import numpy as np

from . import _capi


def init(backend=None, device=None, force=False):
    """Join the torch process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (the control plane of a
    torchrun launch: it ships the RCCL id; `gloo` is enough).  Returns (rank, world).  A world of one joins no group
    unless `force`."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend or "gloo", rank=rank, world_size=world)
    return rank, world


def _dist():
    import sys
    if "torch.distributed" not in sys.modules:  # never imported => no process group; keeps torch out of 1-GPU runs
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def _has_comm(ctx):
    """True when `ctx` carries an in-library RCCL communicator (the product path)."""
    return ctx is not None and hasattr(ctx, "comm_world") and ctx.comm_world() > 0


def comm_init(ctx, rank=None, world=None, id_file=None, timeout_s=120.0):
    """Give `ctx` its RCCL communicator (si_comm_init_rank).  Rank 0 creates the id (si_comm_unique_id); it travels
    either through the torch process group when one is up (torchrun launches), or through `id_file` (a path every rank
    can see: rank 0 writes it atomically, the others poll) -- no torch needed.  Returns (rank, world)."""
    d = _dist()
    if rank is None:
        rank = d.get_rank() if d else int(os.environ.get("RANK", "0"))
    if world is None:
        world = d.get_world_size() if d else int(os.environ.get("WORLD_SIZE", "1"))
    if d is not None and id_file is None:
        box = [None]
        if rank == 0:   # a failure on rank 0 must still reach the broadcast: the other ranks are waiting in it
            try:
                box = [_capi.comm_unique_id()]
            except _capi.SubspaceError as e:
                box = [("error", str(e))]
        d.broadcast_object_list(box, src=0)
        uid = box[0]
        if isinstance(uid, tuple):
            raise _capi.SubspaceError("comm_init: rank 0 could not create the RCCL id: %s" % uid[1], _capi.SI_ERR_COMM)
    elif id_file is not None:
        # File rendezvous.  A file left behind by an EARLIER run must never be taken for this run's id (ncclCommInitRank on
        # a dead id hangs): every rank is handed the same `nonce` by its launcher (SI_COMM_NONCE, e.g. the launch time or
        # the launcher's pid), rank 0 writes nonce + id, the others accept only a file that carries their nonce.  Without a
        # nonce a file older than this process is ignored.  Rank 0 removes the file once it has joined the communicator.
        nonce = os.environ.get("SI_COMM_NONCE", "").encode()[:64]
        started = _process_start_time()
        if rank == 0:
            try:
                os.unlink(id_file)          # whatever an earlier run left there
            except OSError:
                pass
            uid = _capi.comm_unique_id()
            tmp = "%s.tmp.%d" % (id_file, os.getpid())
            with open(tmp, "wb") as f:
                f.write(bytes([len(nonce)]) + nonce + uid)
            os.replace(tmp, id_file)
        else:
            t0 = time.time()
            uid = None
            while uid is None:
                try:
                    st = os.stat(id_file)
                    with open(id_file, "rb") as f:
                        blob = f.read()
                    ok = len(blob) >= 1 and len(blob) == 1 + blob[0] + _capi.SI_COMM_ID_BYTES and blob[1:1 + blob[0]] == nonce
                    if ok and (nonce or st.st_mtime >= started - 1.0):
                        uid = blob[1 + blob[0]:]
                except OSError:
                    pass
                if uid is None:
                    if time.time() - t0 > timeout_s:
                        raise _capi.SubspaceError("comm_init: no RCCL id of this run appeared at %s within %.0f s" % (id_file, timeout_s),
                                                  _capi.SI_ERR_COMM)
                    time.sleep(0.01)
    elif world == 1:
        uid = _capi.comm_unique_id()
    else:
        raise _capi.SubspaceError("comm_init: %d ranks need a way to share the RCCL id (a torch process group or id_file)" % world)
    try:
        ctx.comm_init_rank(world, rank, uid)
    finally:
        # every rank has read the id once ncclCommInitRank returns anywhere; a bring-up that RAISES must not leave the file
        # behind either (the next run would find a stale id there)
        if id_file is not None and rank == 0:
            try:
                os.unlink(id_file)
            except OSError:
                pass
    return rank, world


def _process_start_time():
    """wall-clock start of this process (seconds since the epoch): /proc/self/stat field 22 in clock ticks since boot"""
    try:
        with open("/proc/self/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - up + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        return time.time()


def world(ctx=None):
    """(rank, world) of the ctx's communicator, else of the torch process group, else (0, 1)."""
    if _has_comm(ctx):
        w, r, _ = ctx.comm_info()
        return r, w
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def chain_ids(nchains_total, rank, world_size):
    """Contiguous block partition of chain ids 0..nchains_total-1 (chain id = Philox stream id, so the union over
    ranks is exactly the set of chains a single GPU would have run)."""
    base, rem = divmod(nchains_total, world_size)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def row_shard(n, rank, world_size, align=32):
    """Row block [r0, r1) of the flattened weight vector for this rank; boundaries aligned to `align` elements
    (256 B) so every shard keeps the kernels' 16-B access alignment.  Same arithmetic as the library's si_row_shard
    (asserted in tests/test_dist_cpu.py)."""
    blocks = (n + align - 1) // align
    base, rem = divmod(blocks, world_size)
    b0 = rank * base + min(rank, rem)
    b1 = b0 + base + (1 if rank < rem else 0)
    return min(b0 * align, n), min(b1 * align, n)


def col_shard(b, rank, world_size):
    """Observation block [b0, b1) of (X, Y) for this rank in the data-sharded density."""
    base, rem = divmod(b, world_size)
    b0 = rank * base + min(rank, rem)
    return b0, b0 + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------- CPU test double (gloo)
def _tensor(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def allreduce_sum(a, ctx=None):
    """Sum a small fp64 array over all ranks: through the ctx's communicator when it has one, else (test double) gloo."""
    if _has_comm(ctx) and np.size(a) <= 4096:
        return ctx.comm_allreduce_host(np.asarray(a, dtype=np.float64), "sum")
    d = _dist()
    if d is None:
        return np.array(a, dtype=np.float64)
    t = _tensor(np.asarray(a, dtype=np.float64))
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return t.numpy().reshape(np.shape(a))


def broadcast(a, src=0, shape=None, dtype=np.float64):
    """HOST arrays from rank `src` to every rank over the torch group (test double / host-resident callers).  Non-source
    ranks pass a=None and the `shape`.  Device-resident subspaces travel with `replicate_subspace_dev` instead."""
    d = _dist()
    if d is None:
        return np.array(a, dtype=dtype)
    if d.get_rank() == src:
        buf = np.ascontiguousarray(np.asfortranarray(a, dtype=dtype).ravel(order="F"))
        shape = np.shape(a)
    else:
        if shape is None:
            raise ValueError("broadcast: receiving ranks need the shape")
        buf = np.empty(int(np.prod(shape)), dtype=dtype)
    t = _tensor(buf)
    d.broadcast(t, src=src)
    return np.asfortranarray(t.numpy().reshape(shape, order="F"))


def replicate_subspace(w_swa, p, n, m, src=0):
    """(W_swa, P) as HOST arrays from rank `src` to every rank (test double; see replicate_subspace_dev)."""
    return broadcast(w_swa, src, (n,)), broadcast(p, src, (n, m))


def allgather_rows(local, n_total):
    """Concatenate row blocks (row_shard order) of a host vector / matrix held one block per rank (test double)."""
    d = _dist()
    if d is None:
        return np.array(local)
    import torch
    rank, ws = d.get_rank(), d.get_world_size()
    local = np.ascontiguousarray(local, dtype=np.float64)
    cols = 1 if local.ndim == 1 else local.shape[1]
    sizes = [row_shard(n_total, r, ws)[1] - row_shard(n_total, r, ws)[0] for r in range(ws)]
    mx = max(sizes)
    buf = np.zeros((mx, cols), dtype=np.float64)
    buf[:sizes[rank]] = local.reshape(sizes[rank], cols)
    t = _tensor(buf)
    outs = [torch.empty_like(t) for _ in range(ws)]
    d.all_gather(outs, t)
    full = np.concatenate([o.numpy()[:sizes[r]] for r, o in enumerate(outs)], axis=0)
    return full[:, 0] if local.ndim == 1 else np.asfortranarray(full)


# ---------------------------------------------------------------------------------------------- the sharded hot path
def _allreduce_gram(ctx):
    if _has_comm(ctx):
        ctx.construct_allreduce_gram()      # in place on the library's K x K device buffer, on its stream
    elif _dist() is not None:
        ctx.construct_gram_set(allreduce_sum(ctx.construct_gram_get()))


def sharded_construct_finish(ctx, m, n_total=None, gather=True):
    """Row-sharded `U,s,V = psvd(A); P = U[:,1:M]*Diagonal(s[1:M])` (src/subspace_construction.jl:63,65).

    `ctx` holds this rank's row block (pushed with si_construct_push).  G = sum over ranks of the local A'A
    (si_construct_allreduce_gram: RCCL, in place on the device), the K x K eigensolve is replicated, P rows are local.
    With gather=True the full W_swa / P are assembled on every rank (si_construct_allgather, device to device) and
    returned as host arrays."""
    ctx.construct_gram()
    _allreduce_gram(ctx)
    if ctx.construct_needs_refine(m):   # ill-conditioned A: every rank sees the same G, hence the same answer
        ctx.construct_refine()          # B = A V on this rank's rows; the second-stage Gram matrix replaces G
        _allreduce_gram(ctx)
    w_loc, p_loc, s, k = ctx.construct_finish(m)
    if not gather or n_total is None:
        return w_loc, p_loc, s, k
    if _has_comm(ctx):
        ctx.construct_allgather(n_total)
        w_full, p_full, _ = ctx.construct_get_result()
        return w_full, p_full, s, k
    return allgather_rows(w_loc, n_total), allgather_rows(p_loc, n_total), s, k


def sharded_construct_finish_dev(ctx, m, n_total):
    """The same with the results kept on the DEVICE: after it `ctx` holds a finished construction of all n_total rows
    (W_swa [ld], P [ld x M], ld = n_total rounded up to 64, padding rows zero), assembled by si_construct_allgather --
    what `infer_setup(table, n_total, m, None, None, ...)` then uses in place.  Nothing crosses PCIe.  Returns s."""
    ctx.construct_gram()
    _allreduce_gram(ctx)
    if ctx.construct_needs_refine(m):
        ctx.construct_refine()
        _allreduce_gram(ctx)
    _, _, s, _ = ctx.construct_finish(m, want_swa=False, want_p=False)
    if _has_comm(ctx):
        ctx.construct_allgather(n_total)
    return s


def replicate_subspace_dev(ctx, n, m, src=0):
    """(W_swa, P, s) of the construction finished on rank `src` -> every rank's ctx, DEVICE TO DEVICE (cfg3: 168 MB at
    cfg2, once; si_bcast_subspace).  Receivers then hold a finished construction: `infer_setup(..., None, None, ...)`."""
    ctx.bcast_subspace(src, n, m)


def sample_data_sharded(ctx, itr, sigma_z, seed, d_total, chain_id0=0, nchains=1):
    """RWMH with the DATA split over the ranks: `ctx` was set up with this rank's column block of
    (X, Y) and the full W_swa / P.  Per transition every rank evaluates the SAME proposal on its block and the partial
    sums of squared errors are all-reduced (8*nchains bytes) before the accept step; all ranks return the same chain.
    d_total = out_dim * (observations over all ranks).  With a communicator the whole loop is ONE library call
    (si_sample_rwmh_sharded: eval -> RCCL -> accept on one stream, no host round trip)."""
    if _has_comm(ctx):
        return ctx.sample_rwmh_sharded(itr, sigma_z, seed, d_total, chain_id0, nchains)
    ctx.rwmh_begin(itr, sigma_z, seed, chain_id0, nchains, d_total)
    for _ in range(itr):
        ctx.rwmh_step_accept(allreduce_sum(ctx.rwmh_step_eval()))
    return ctx.rwmh_end()


def train_step_data_parallel(ctx, idx_local, nb_total):
    """One data-parallel `gradient(ps) do cost(model, d...) end; Flux.update!(opt, ps, gs)` (src/subspace_construction.jl:
    39-43, SURVEY 8e last paragraph): `idx_local` are this rank's observations of the batch of `nb_total`; every rank
    holds the same weights and optimiser state before and after.  The only exchange is ONE all-reduce(sum) of the
    N-double gradient plus the 8-byte SSE (si_train_step_dp; host staging under the gloo test double).  Returns the mse
    of the whole batch before the update."""
    if _has_comm(ctx):
        return ctx.train_step_dp(idx_local, nb_total)
    sse = ctx.train_grad(idx_local, nb_total)
    if _dist() is not None:
        ctx.train_grad_set(allreduce_sum(ctx.train_grad_get()))
        sse = float(allreduce_sum(np.array([sse]))[0])
    ctx.train_apply()
    return sse / (ctx.train_out_dim() * nb_total)


def sample_chains(ctx, nchains_total, itr, sigma_z, seed):
    """Independent RWMH chains spread over the ranks; returns (Z (M x itr x nchains_total), lp, accept) on every rank."""
    rank, ws = world(ctx)
    ids = chain_ids(nchains_total, rank, ws)
    if ids:
        z, lp, acc = ctx.sample_rwmh(itr, sigma_z, seed, chain_id0=ids[0], nchains=len(ids))
    else:
        z = lp = acc = None
    if ws == 1:
        return z, lp, acc
    if _has_comm(ctx):
        # equal-sized padded blocks through the communicator (M*itr*8 B per chain: tiny)
        per = max(len(chain_ids(nchains_total, r, ws)) for r in range(ws))
        m = ctx._m
        blk = np.zeros((m + 1) * itr * per + per, dtype=np.float64)
        if ids:
            nz = m * itr * len(ids)
            blk[:nz] = z.reshape(-1, order="F")
            blk[m * itr * per:m * itr * per + itr * len(ids)] = lp.reshape(-1, order="F")
            blk[(m + 1) * itr * per:(m + 1) * itr * per + len(ids)] = acc
        allb = ctx.comm_allgather_host(blk)
        zs, lps, accs = [], [], []
        for r in range(ws):
            nr = len(chain_ids(nchains_total, r, ws))
            if nr == 0:
                continue
            b = allb[r]
            zs.append(b[:m * itr * nr].reshape((m, itr, nr), order="F"))
            lps.append(b[m * itr * per:m * itr * per + itr * nr].reshape((itr, nr), order="F"))
            accs.append(b[(m + 1) * itr * per:(m + 1) * itr * per + nr])
        return np.concatenate(zs, axis=2), np.concatenate(lps, axis=1), np.concatenate(accs)
    d = _dist()
    parts = [None] * ws
    d.all_gather_object(parts, (ids, z, lp, acc))
    zs = np.concatenate([p[1] for p in parts if p[0]], axis=2)
    lps = np.concatenate([p[2] for p in parts if p[0]], axis=1)
    accs = np.concatenate([p[3] for p in parts if p[0]])
    return zs, lps, accs
