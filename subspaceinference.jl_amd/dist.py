"""One-process-per-GPU plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the
CPU tests).  The reference is single-process (zero collectives); what is sharded here is what the hot
path allows to be sharded:

  * independent chains  -- chain c runs on rank c mod world with W_swa / P / X / Y replicated; NO per-step collective;
                           one gather of (Z, lp) at the end.  This is bench.py's weak-scaling mode.
  * row-sharded construction -- every rank holds a row block of w, W_swa, A and P.  K1 (SWA/deviation push) and K3
                           (projection) are row-local; the only exchange is ONE all-reduce(sum) of the K x K fp64 Gram
                           matrix (80 KB at K=100: latency-bound) before the replicated K x K eigensolve.

torch is used for the process group and the collective only; all arithmetic is in the HIP library.
"""
import os

import numpy as np


def init(backend=None, device=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns (rank, world).
    A world of one joins no group unless `force` (used to rehearse the RCCL path on a single GPU)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            import torch
            kw["device_id"] = torch.device("cuda", device)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def _dist():
    import sys
    if "torch.distributed" not in sys.modules:  # never imported => no process group; keeps torch out of 1-GPU runs
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
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
    (256 B) so every shard keeps the kernels' 16-B access alignment."""
    blocks = (n + align - 1) // align
    base, rem = divmod(blocks, world_size)
    b0 = rank * base + min(rank, rem)
    b1 = b0 + base + (1 if rank < rem else 0)
    return min(b0 * align, n), min(b1 * align, n)


def _tensor(a):
    import torch
    d = _dist()
    t = torch.from_numpy(np.ascontiguousarray(a))
    if d is not None and d.get_backend() == "nccl":
        t = t.cuda()
    return t


def allreduce_sum(a):
    """Sum a small fp64 array over all ranks (the K x K Gram matrix / SWA moments)."""
    d = _dist()
    if d is None:
        return np.array(a, dtype=np.float64)
    t = _tensor(np.asarray(a, dtype=np.float64))
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return t.cpu().numpy().reshape(np.shape(a))


def broadcast(a, src=0, shape=None, dtype=np.float64):
    """Replicate an array from rank `src` on every rank (independent chains: W_swa / P / X / Y are broadcast
    ONCE at setup -- 168 MB of P at cfg2 -- and nothing is exchanged per step).  Non-source ranks pass a=None and the
    `shape` to receive into."""
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
    return np.asfortranarray(t.cpu().numpy().reshape(shape, order="F"))


def replicate_subspace(w_swa, p, n, m, src=0):
    """(W_swa, P) constructed on rank `src` -> every rank (what the chains of cfg3 start from)."""
    return broadcast(w_swa, src, (n,)), broadcast(p, src, (n, m))


def allgather_rows(local, n_total):
    """Concatenate row blocks (row_shard order) of a vector / matrix held one block per rank."""
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
    full = np.concatenate([o.cpu().numpy()[:sizes[r]] for r, o in enumerate(outs)], axis=0)
    return full[:, 0] if local.ndim == 1 else np.asfortranarray(full)


def sharded_construct_finish(ctx, m, n_total=None, gather=True):
    """Row-sharded `U,s,V = psvd(A); P = U[:,1:M]*Diagonal(s[1:M])` (src/subspace_construction.jl:63,65).

    `ctx` holds this rank's row block (pushed with si_construct_push).  G = sum over ranks of the local A'A, the
    eigensolve is replicated, P rows are local.  With gather=True the full W_swa / P are assembled on every rank."""
    ctx.construct_gram()
    g = allreduce_sum(ctx.construct_gram_get())
    ctx.construct_gram_set(g)
    w_loc, p_loc, s, k = ctx.construct_finish(m)
    if not gather or n_total is None:
        return w_loc, p_loc, s, k
    return allgather_rows(w_loc, n_total), allgather_rows(p_loc, n_total), s, k


def col_shard(b, rank, world_size):
    """Observation block [b0, b1) of (X, Y) for this rank in the data-sharded density."""
    base, rem = divmod(b, world_size)
    b0 = rank * base + min(rank, rem)
    return b0, b0 + base + (1 if rank < rem else 0)


def sample_data_sharded(ctx, itr, sigma_z, seed, d_total, chain_id0=0, nchains=1):
    """RWMH with the DATA split over the ranks: `ctx` was set up with this rank's column block of
    (X, Y) and the full W_swa / P.  Per transition every rank evaluates the SAME proposal on its block and the partial
    sums of squared errors are all-reduced (8*nchains bytes) before the accept step; all ranks return the same chain.
    d_total = out_dim * (observations over all ranks)."""
    ctx.rwmh_begin(itr, sigma_z, seed, chain_id0, nchains, d_total)
    for _ in range(itr):
        ctx.rwmh_step_accept(allreduce_sum(ctx.rwmh_step_eval()))
    return ctx.rwmh_end()


class _DeviceBuffer:
    """Zero-copy view of a device buffer owned by the HIP library (`__cuda_array_interface__`), so that
    torch.distributed can all-reduce it in place over RCCL."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def train_step_data_parallel(ctx, idx_local, nb_total):
    """One data-parallel `gradient(ps) do cost(model, d...) end; Flux.update!(opt, ps, gs)` (src/subspace_construction.jl:
    39-43, SURVEY 8e last paragraph): `idx_local` are this rank's observations of the batch of `nb_total`; every rank
    holds the same weights and optimiser state before and after.  The only exchange is ONE all-reduce(sum) of the
    N-double gradient -- in place on the library's device buffer over RCCL, through host staging under gloo -- plus the
    8-byte SSE.  Returns the mse of the whole batch before the update."""
    d = _dist()
    sse = ctx.train_grad(idx_local, nb_total)
    if d is not None:
        if d.get_backend() == "nccl":
            import torch
            ptr, n = ctx.train_grad_ptr()
            g = torch.as_tensor(_DeviceBuffer(ptr, n), device="cuda")  # si_train_grad has synchronised the library's stream
            d.all_reduce(g, op=d.ReduceOp.SUM)
            torch.cuda.current_stream().synchronize()                 # RCCL ran on torch's stream
        else:
            ctx.train_grad_set(allreduce_sum(ctx.train_grad_get()))
        sse = float(allreduce_sum(np.array([sse]))[0])
    ctx.train_apply()
    return sse / (ctx.train_out_dim() * nb_total)


def sample_chains(ctx, nchains_total, itr, sigma_z, seed):
    """Independent RWMH chains spread over the ranks; returns (Z (M x itr x nchains_total), lp, accept) on every rank."""
    d = _dist()
    rank, ws = world()
    ids = chain_ids(nchains_total, rank, ws)
    if ids:
        z, lp, acc = ctx.sample_rwmh(itr, sigma_z, seed, chain_id0=ids[0], nchains=len(ids))
    else:
        z = lp = acc = None
    if d is None:
        return z, lp, acc
    parts = [None] * ws
    d.all_gather_object(parts, (ids, z, lp, acc))
    zs = np.concatenate([p[1] for p in parts if p[0]], axis=2)
    lps = np.concatenate([p[2] for p in parts if p[0]], axis=1)
    accs = np.concatenate([p[3] for p in parts if p[0]])
    return zs, lps, accs
