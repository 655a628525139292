"""Host-side mirror of the reference's exported API for the hot path (src/SubspaceInference.jl:27-34):

    subspace_construction(model, cost, data, opt; T=10, c=1, M=3, print_freq=1)        -> (W_swa, P)
    subspace_inference(model, cost, data, opt; σ_z, σ_m, σ_p, itr, T, c, M, ...)        -> (chn, lp, W_swa)
    sub_inference(in_model, data, W_swa, P; σ_z, σ_m, σ_p, itr, M, alg, backend)        -> (chn, lp)
    inference(...)  README-only alias of sub_inference (README.md:153-154); alg=:mh == :rwmh

Same names, argument meaning, defaults and error behaviour (`throw(String)` -> SubspaceError) as the
reference.  Everything numerical goes through the C ABI (include/subspace_hip.h) to the HIP kernels; the
only host arithmetic here is the caller-side training step (flux.py, see its header).  Extra keyword
arguments (device, ctx, seed, max_cols, ...) are additions and default to the reference's behaviour.
Python identifiers may be Greek, so `σ_z=` works as in Julia; `sigma_z=` is accepted too.
"""
import numpy as np

from . import _capi, dist, flux, samplers
from ._capi import Context, SubspaceError

_RWMH_ALGS = ("rwmh", "mh")


def _get_ctx(ctx, device):
    return (ctx, False) if ctx is not None else (Context(device), True)


def _alg_name(alg):
    return str(alg).lstrip(":")


def subspace_construction(model, cost, data, opt, T=10, c=1, M=3, print_freq=1, *, device=0, ctx=None,
                          max_cols=0, verbose=True, keep_on_device=False, device_training="auto", init="zeros",
                          data_parallel=False, a_storage="f64", compute_dtype="auto"):
    """src/subspace_construction.jl:26-67.

    Per batch the host does `gradient` + `update!` (:39-43, caller side) and hands the flattened weights
    (`extract_params`, src/libs.jl:19-22, Float32) plus `n = i/c` to the device, which applies the SWA update,
    forms the deviation column and appends it (:45-52, kernel K1).  After the loop the device forms A'A
    (K2), the host solves the K x K eigenproblem (H1) and the device projects P = A*V_M (K3) == :61-65.

    device_training ("auto" | True | False): when the cost is `flux.mse` and the optimiser is a fresh Descent /
    Momentum / ADAM, the training step itself (:39-43) also runs on the GPU (si_train_step: forward, reverse sweep,
    optimiser) and the weights are pushed in place (si_train_push) -- no weight vector crosses PCIe.  The model's
    arrays receive the trained weights at the end, like Flux's in-place `update!`.

    init ("zeros" | "pretrained"): "zeros" is what the reference's CODE does (W_swa = zeros, :31 -- quirk Q1, the default);
    "pretrained" starts the running mean at the model's weights, as the reference's docs describe (nn_example.md:44).

    compute_dtype ("auto" | "f32" | "f64") of the DEVICE training step: "auto" is the reference's own arithmetic for the data
    handed in -- a Float32 model on Float32 (X, Y) is a Float32 Zygote pass (:39-43), Float64 data (every example of the
    reference: `rand(10, 100)`) promote the pass to Float64; "f32" / "f64" override (Float64 data are then rounded once /
    Float32 data widened).  The host step (device_training=False) follows NumPy's promotion, which is Julia's.

    a_storage ("f64" | "f32"): "f64" keeps the deviation matrix in Float64 like the reference (:33,51-52); "f32" (opt-in,
    SURVEY section 0 Q6) stores every column rounded once to Float32 -- half the memory and half the bytes of the Gram /
    projection kernels, P within 1e-5 of the Float64 result (tests/test_gpu_a32.py).

    data_parallel=True (opt-in; needs a ctx with an RCCL communicator -- one process per GPU, dist.comm_init -- or a torch
    process group) makes the device training step data-parallel: EVERY rank must make this call with the same model, data
    and DataLoader seed; each takes its share of every batch and the gradient is all-reduced once per step inside the
    library (si_train_step_dp), so all ranks return the same (W_swa, P).  The default keeps the step local whatever the
    ctx carries: the documented cfg3 flow (construct on rank 0, then si_bcast_subspace) must not find rank 0 alone inside a
    gradient all-reduce.
    """
    ps = flux.params(model)
    n_par = int(sum(p.size for p in ps))
    nb = len(data)
    n_push = sum(1 for i in range(1, T + 1) if i % c == 0) * nb
    if n_push == 0:
        # reference: reshape of an empty A, then psvd / U[:,1:M] fails
        raise SubspaceError("BoundsError: no snapshot was collected (mod(i,c) never 0 for i in 1:T)")
    dev_opt = flux.device_optimiser(opt) if isinstance(cost, flux.MSE) and hasattr(data, "index_batches") else None
    # si_train_* keeps Float32 weights and optimiser state (Flux's default): a Float64 model stays on the host step,
    # otherwise "auto" and device_training=False would return different (W_swa, P) for the same inputs
    all_f32 = all(p.dtype == np.float32 for p in ps)
    if device_training is True and (dev_opt is None or not all_f32):
        raise SubspaceError("device_training needs cost = flux.mse, a fresh Descent / Momentum / ADAM optimiser and "
                            "Float32 parameters (the device step keeps Float32 weights and optimiser state)")
    use_dev = dev_opt is not None and all_f32 and device_training in ("auto", True)
    ctx, own = _get_ctx(ctx, device)
    try:
        ctx.construct_begin(n_par, n_push, max_cols)
        if a_storage not in ("f64", "f32"):
            raise SubspaceError("a_storage must be \"f64\" (the reference) or \"f32\"")
        if a_storage == "f32":
            ctx.construct_set_storage(_capi.SI_F32)
        if _alg_name(init) == "pretrained":
            ctx.construct_set_mean(flux.extract_params(ps))
        elif _alg_name(init) != "zeros":
            raise SubspaceError("init must be :zeros (the reference's behaviour) or :pretrained")
        training_loss = 0.0
        if data_parallel not in (True, False):
            raise SubspaceError("data_parallel must be True or False")
        if data_parallel and not use_dev:
            raise SubspaceError("data_parallel=True needs the device training step (cost = flux.mse, Float32 model, a fresh optimiser)")
        use_dp = use_dev and data_parallel is True
        dp_rank, dp_world = dist.world(ctx) if use_dp else (0, 1)
        if use_dev:
            xm, ym, in_size = flux.data_matrices(data)
            table, _ = flux.layer_table(model, in_size)
            bmax = min(data.batchsize, data.nobs)
            if compute_dtype not in ("auto", "f32", "f64"):
                raise SubspaceError("compute_dtype must be \"auto\" (the data's element type, as in the reference), \"f32\" or \"f64\"")
            ctx.train_setup(table, n_par, flux.extract_params(ps), xm, ym, bmax, *dev_opt,
                            compute_dtype={"auto": None, "f32": _capi.SI_F32, "f64": _capi.SI_F64}[compute_dtype])
        for i in range(1, T + 1):
            if use_dev:
                last = (i % print_freq == 0) or (i == T)
                batches = list(data.index_batches())
                for j, ids in enumerate(batches):
                    if use_dp and (dp_world > 1 or dist._has_comm(ctx)):
                        # data-parallel step: this rank's share of the batch, one gradient all-reduce (dist.py)
                        c0, c1 = dist.col_shard(len(ids), dp_rank, dp_world)
                        loss = dist.train_step_data_parallel(ctx, np.asarray(ids)[c0:c1], len(ids))
                    else:
                        loss = ctx.train_step(ids, want_loss=last and j == len(batches) - 1)
                    if loss is not None:
                        training_loss = loss
                    if i % c == 0:
                        ctx.train_push(i / c)
            else:
                for d in data:
                    training_loss, gs = flux.gradient(cost, model, *d)
                    flux.update(opt, ps, gs)
                    if i % c == 0:
                        ctx.construct_push(flux.extract_params(ps), i / c)
            if (i % print_freq == 0) or (i == T):
                if verbose:
                    print("Traing loss: ", training_loss, " Epoch: ", i)  # [sic] reference :57
        if use_dev:
            flux.load_flat(model, ctx.train_get_weights())
            flux.store_device_state(opt, model, *ctx.train_get_opt_state())  # opt continues where the device left it
        w_swa, p, _, _ = ctx.construct_finish(M, want_swa=True, want_p=not keep_on_device)
        return w_swa, p
    finally:
        if own and not keep_on_device:
            ctx.close()


def sub_inference(in_model, data, W_swa, P, σ_z=1.0, σ_m=1.0, σ_p=1.0, itr=100, M=3, alg="rwmh",
                  backend="forwarddiff", *, sigma_z=None, sigma_m=None, sigma_p=None, device=0, ctx=None,
                  seed=0, chain_id=0, return_z=False, nchains=1, include_prior=False, compute_dtype="f64"):
    """src/space_inference.jl:82-164 for a Chain model and alg = :rwmh.

    `density(z)` (:90-95: W_swa + P*z -> model_re -> forward over the FULL data -> Gaussian log-likelihood,
    prior term dead) and the RWMH loop (:111-116) run on the device; the output map (:125) materialises
    `W_swa + P*z` for every sample like the reference unless `return_z=True` (then chn is the M x itr matrix
    of subspace samples).  σ_p is accepted and unused, exactly as in the reference (quirk Q4), unless
    `include_prior=True` (non-default) asks for the term the reference's source writes after its `return`.

    `nchains > 1` (RWMH only; not in the reference, which runs one chain per call) runs the independent chains
    chain_id .. chain_id+nchains-1 stacked in every launch of the forward pass: chn becomes a list over chains
    (or the M x itr x nchains array with return_z) and lp is itr x nchains.

    `compute_dtype="f32"` (non-default; SURVEY section 0 Q6: "forward fp64 with fp32 as a measured option") evaluates the
    density of a Dense chain on the fp32 matrix instruction: X rounded once, W_swa + P*z formed in fp64 and rounded once per
    transition, fp32 activations, head + sum of squared errors in fp64 (lp within 1e-5 of the fp64 value, tests/test_gpu_f32.py).
    The gradient samplers (:mala / :hmc / :nuts) keep the fp64 reverse sweep either way.
    """
    σ_z = σ_z if sigma_z is None else sigma_z
    σ_m = σ_m if sigma_m is None else sigma_m
    σ_p = σ_p if sigma_p is None else sigma_p
    a = _alg_name(alg)
    if a == "advi":
        raise SubspaceError("advi is outside what this build accelerates (SURVEY section 2)")
    if a not in _RWMH_ALGS and a not in ("mala", "hmc", "nuts"):
        raise SubspaceError("%s is not available" % a)  # reference :162
    if not isinstance(in_model, flux.Chain):
        raise SubspaceError("Error: density function is not avaliable for this model")  # [sic] reference :103
    x, y, in_size = flux.data_matrices(data)  # split_data, src/libs.jl:75-77 (full data, not the batches)
    table, n_par = flux.layer_table(in_model, in_size)
    if compute_dtype not in ("f64", "f32"):
        raise SubspaceError("compute_dtype must be \"f64\" (the reference's arithmetic) or \"f32\"")
    cdt = _capi.SI_F32 if compute_dtype == "f32" else _capi.SI_F64
    ctx, own = _get_ctx(ctx, device)
    try:
        if W_swa is None:
            ctx.infer_setup(table, n_par, M, None, None, x, y, σ_m, compute_dtype=cdt)
        else:
            W_swa = np.asarray(W_swa, dtype=np.float64)
            P = np.asarray(P, dtype=np.float64)
            if P.shape[1] != M:
                # reference: MvNormal(zeros(M), σ_z) proposal against an N x size(P,2) matrix -> DimensionMismatch in P*z
                raise SubspaceError("DimensionMismatch: P has %d columns but M = %d" % (P.shape[1], M))
            ctx.infer_setup(table, n_par, M, W_swa, P, x, y, σ_m, compute_dtype=cdt)
        # include_prior=True adds the term the reference leaves dead after its `return` (quirk Q4); default: as the reference
        ctx.set_prior(σ_p if include_prior else 0.0)
        if nchains != 1 and a not in _RWMH_ALGS:
            raise SubspaceError("nchains > 1 is available for alg = :rwmh / :mh only")
        if a in _RWMH_ALGS:
            if return_z:
                z, lp, _ = ctx.sample_rwmh(itr, σ_z, seed, chain_id, nchains)
                return (z, lp) if nchains > 1 else (z[:, :, 0], lp[:, 0])
            # :125 map(z -> W_swa + P*z.params, chm): the weight samples stream out of the device WHILE the chain runs
            # (si_sample_rwmh_weights: K4's own output, selected on accept; DMA + host copy hidden under the next transitions)
            _, lp, _, w = ctx.sample_rwmh_weights(itr, σ_z, seed, chain_id, nchains)
            if nchains > 1:
                return [[w[:, t, c] for t in range(itr)] for c in range(nchains)], lp
            return [w[:, t, 0] for t in range(itr)], lp[:, 0]
        else:
            # :mala (:117-120) / :hmc, :nuts (:139-160): the sampler logic is host control flow, every density + gradient
            # evaluation is the device reverse sweep (si_logdensity_grad) instead of M-wide ForwardDiff duals (:107)
            rng = np.random.default_rng([int(seed), int(chain_id)])
            fn = {"mala": samplers.mala, "hmc": samplers.hmc, "nuts": samplers.nuts}[a]
            z, lp, _ = fn(ctx.logdensity_grad, M, itr, σ_z, rng)
        if return_z:
            return z, lp
        w = ctx.reconstruct(z)  # :125  map(z -> W_swa + P*z.params, chm)
        return [w[:, t] for t in range(w.shape[1])], lp
    finally:
        if own:
            ctx.close()


def inference(*args, **kwargs):
    """README.md:153-154 name for `sub_inference`."""
    return sub_inference(*args, **kwargs)


def subspace_inference(model, cost, data, opt, σ_z=1.0, σ_m=1.0, σ_p=1.0, itr=1000, T=25, c=1, M=20,
                       print_freq=1, alg="rwmh", backend="forwarddiff", method="subspace", *, sigma_z=None,
                       sigma_m=None, sigma_p=None, device=0, ctx=None, seed=0, verbose=True, return_z=False,
                       compute_dtype="f64"):
    """src/space_inference.jl:33-54: construction, then sampling; returns (chn, lp, W_swa).
    W_swa and P stay on the device between the two stages (no host round trip)."""
    m = _alg_name(method)
    if m == "diffusion":
        raise SubspaceError("method :diffusion is outside the accelerated path (SURVEY section 2)")
    if m != "subspace":
        raise SubspaceError("Error: No method found")  # reference :42
    a = _alg_name(alg)
    if a in ("turing_mh", "turing_nuts"):
        raise SubspaceError("Turing samplers are outside the accelerated path (SURVEY section 2)")
    ctx, own = _get_ctx(ctx, device)
    try:
        w_swa, _ = subspace_construction(model, cost, data, opt, T=T, c=c, M=M, print_freq=print_freq, ctx=ctx,
                                         verbose=verbose, keep_on_device=True)
        chn, lp = sub_inference(model, data, None, None, σ_z=σ_z, σ_m=σ_m, σ_p=σ_p, itr=itr, M=M, alg=alg,
                                backend=backend, sigma_z=sigma_z, sigma_m=sigma_m, ctx=ctx, seed=seed,
                                return_z=return_z, compute_dtype=compute_dtype)
        return chn, lp, w_swa
    finally:
        if own:
            ctx.close()
