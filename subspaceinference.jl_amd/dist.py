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


def _on_rccl():
    d = _dist()
    return d is not None and d.get_backend() == "nccl"


_side_streams = {}


def bind_stream(ctx):
    """Under RCCL: run the library's kernels on a torch side stream (torch's default stream is the NULL stream, which
    the library maps to its own) and return that stream.  Collectives issued inside `with torch.cuda.stream(s)` are then
    stream-ordered with the kernels before and after them: no hipStreamSynchronize, no host staging around an in-place
    all-reduce of a library buffer."""
    import torch
    dev = getattr(ctx, "device", 0)
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(device=dev)
    if getattr(ctx, "_bound_stream", None) is not s:
        ctx.set_stream(s.cuda_stream)
        ctx._bound_stream = s
    return s


def _dev_view(ptr, shape, strides_elems=None):
    """torch view of library-owned device memory (fp64).  shape / strides in elements, torch (row-major) index order."""
    import torch
    iface = {"shape": tuple(int(v) for v in shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    if strides_elems is not None:
        iface["strides"] = tuple(8 * int(v) for v in strides_elems)

    class _Buf:
        __cuda_array_interface__ = iface
    return torch.as_tensor(_Buf(), device="cuda")


def allreduce_inplace(ctx, ptr, n):
    """all-reduce(sum) of n doubles at device address `ptr` (a buffer owned by `ctx`'s library) IN PLACE over RCCL,
    stream-ordered on the stream the library runs on."""
    import torch
    d = _dist()
    s = bind_stream(ctx)
    with torch.cuda.stream(s):
        d.all_reduce(_dev_view(ptr, (n,)), op=d.ReduceOp.SUM)


def _allreduce_gram(ctx):
    if _on_rccl():
        ptr, k = ctx.construct_gram_ptr()
        allreduce_inplace(ctx, ptr, k * k)
    elif _dist() is not None:
        ctx.construct_gram_set(allreduce_sum(ctx.construct_gram_get()))


def sharded_construct_finish(ctx, m, n_total=None, gather=True):
    """Row-sharded `U,s,V = psvd(A); P = U[:,1:M]*Diagonal(s[1:M])` (src/subspace_construction.jl:63,65).

    `ctx` holds this rank's row block (pushed with si_construct_push).  G = sum over ranks of the local A'A, the
    eigensolve is replicated, P rows are local.  With gather=True the full W_swa / P are assembled on every rank.

    Under RCCL the K x K Gram matrix is all-reduced in place on the library's device buffer (si_construct_gram_ptr),
    ordered on the library's stream: no D2H / H2D of G and no synchronisation of its own.  gloo (CPU tests) stages it
    through the host."""
    ctx.construct_gram()
    _allreduce_gram(ctx)
    if ctx.construct_needs_refine(m):   # ill-conditioned A: every rank sees the same G, hence the same answer
        ctx.construct_refine()          # B = A V on this rank's rows; the second-stage Gram matrix replaces G
        _allreduce_gram(ctx)
    w_loc, p_loc, s, k = ctx.construct_finish(m)
    if not gather or n_total is None:
        return w_loc, p_loc, s, k
    return allgather_rows(w_loc, n_total), allgather_rows(p_loc, n_total), s, k


def sharded_construct_finish_dev(ctx, m, n_total):
    """The same, results kept on the DEVICE (RCCL only): every rank ends with torch tensors (W_swa [ld], P [M, ld]
    = column-major ld x M, ld = n_total rounded up to 64, padding rows zero) assembled by ONE all-gather of the padded
    row blocks -- what si_infer_setup_dev(borrow=1) takes.  Nothing crosses PCIe.  Returns (w_swa_t, p_t, ld, s)."""
    import torch
    d = _dist()
    rank, ws = world()
    ctx.construct_gram()
    _allreduce_gram(ctx)
    if ctx.construct_needs_refine(m):
        ctx.construct_refine()
        _allreduce_gram(ctx)
    _, _, s, _ = ctx.construct_finish(m, want_swa=False, want_p=False)
    wptr, pptr, ld_loc, _ = ctx.construct_result_ptr()
    sizes = [row_shard(n_total, r, ws)[1] - row_shard(n_total, r, ws)[0] for r in range(ws)]
    n_loc, mx = sizes[rank], max(sizes)
    ld = (n_total + 63) // 64 * 64
    st = bind_stream(ctx)
    with torch.cuda.stream(st):
        loc = torch.zeros((m + 1, mx), dtype=torch.float64, device="cuda")
        loc[:m, :n_loc] = _dev_view(pptr, (m, n_loc), (ld_loc, 1))
        loc[m, :n_loc] = _dev_view(wptr, (n_loc,))
        if d is not None:
            allg = torch.empty((ws, m + 1, mx), dtype=torch.float64, device="cuda")
            d.all_gather_into_tensor(allg, loc)
        else:
            allg = loc[None]
        p_t = torch.zeros((m, ld), dtype=torch.float64, device="cuda")
        w_t = torch.zeros((ld,), dtype=torch.float64, device="cuda")
        for r in range(ws):
            r0, r1 = row_shard(n_total, r, ws)
            p_t[:, r0:r1] = allg[r, :m, :r1 - r0]
            w_t[r0:r1] = allg[r, m, :r1 - r0]
        del allg, loc
    return w_t, p_t, ld, s


def replicate_subspace_dev(ctx, n, m, src=0):
    """(W_swa, P) of the construction finished on rank `src` -> every rank, DEVICE TO DEVICE (cfg3: 168 MB at cfg2, once;
    cfg5: 26 GB): the source broadcasts straight out of the library's buffers (si_construct_result_ptr), the others
    receive into torch tensors that si_infer_setup_dev(borrow=1) then uses in place.  Returns (w_swa_t, p_t, ld)."""
    import torch
    d = _dist()
    rank, _ = world()
    ld = (n + 63) // 64 * 64
    st = bind_stream(ctx)
    with torch.cuda.stream(st):
        if rank == src:
            wptr, pptr, ld_src, m_src = ctx.construct_result_ptr()
            assert ld_src == ld and m_src == m
            w_t, p_t = _dev_view(wptr, (ld,)), _dev_view(pptr, (m, ld))
        else:
            w_t = torch.empty((ld,), dtype=torch.float64, device="cuda")
            p_t = torch.empty((m, ld), dtype=torch.float64, device="cuda")
        if d is not None:
            d.broadcast(w_t, src=src)
            d.broadcast(p_t, src=src)
    return w_t, p_t, ld


def infer_setup_from_tensors(ctx, table, n, m, w_t, p_t, ld, x_t, y_t, sigma_m):
    """si_infer_setup_dev on torch tensors: W_swa / P used in place (kept alive on the ctx), X / Y copied D2D.
    x_t is [B, in] row-major == in x B column-major (likewise y_t)."""
    import torch
    st = bind_stream(ctx)
    st.synchronize()
    torch.cuda.current_stream().synchronize()
    ctx._borrowed = (w_t, p_t)
    b = x_t.shape[0]
    ctx.infer_setup_dev(table, n, m, w_t.data_ptr(), p_t.data_ptr(), ld, x_t.data_ptr(), y_t.data_ptr(),
                        x_t.shape[1], y_t.shape[1], b, sigma_m, borrow=True)


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
    if _on_rccl():
        # the partial sums never leave the device: eval -> in-place RCCL all-reduce -> accept, all on one stream
        import torch
        d = _dist()
        st = bind_stream(ctx)
        ptr, c = ctx.rwmh_sse_ptr()
        sse = _dev_view(ptr, (c,))
        with torch.cuda.stream(st):
            for _ in range(itr):
                ctx.rwmh_step_eval(on_device=True)
                d.all_reduce(sse, op=d.ReduceOp.SUM)
                ctx.rwmh_step_accept(None)
    else:
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
            ptr, n = ctx.train_grad_ptr()
            allreduce_inplace(ctx, ptr, n)  # ordered on the library's stream: si_train_apply follows without a host sync
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
