# SubspaceInferenceHIP.jl -- the `ccall` side of the drop-in boundary (include/subspace_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: Julia is absent from the build image and from the GPU boxes, so this file is
# the binding a maintainer of efmanu/SubspaceInference.jl would add (INTEGRATION.md).  What CAN be checked without
# Julia is checked: tests/test_julia_binding.py parses every `ccall((:si_..., LIB), ret, (argtypes...), ...)` below and
# every prototype of include/subspace_hip.h and asserts symbol, arity and type-by-type agreement (Int32 <-> int32_t,
# Ptr{Float64} <-> double*, ...), the SiLayer field layout against `si_layer`, and that the exported names and keyword
# defaults are the reference's.  Every executable test goes through the Python ctypes binding (_capi.py) over the
# IDENTICAL C ABI.
#
# It keeps the reference's exported names and keyword arguments (src/SubspaceInference.jl:27-34):
#   subspace_construction(model, cost, data, opt; T, c, M, print_freq)              -> W_swa, P
#   subspace_inference(model, cost, data, opt; σ_z, σ_m, σ_p, itr, T, c, M, ...)      -> chn, lp, W_swa
#   sub_inference(in_model, data, W_swa, P; σ_z, σ_m, σ_p, itr, M, alg, backend)      -> chn, lp
#   inference(...)  (README.md:153-154 name), alg = :mh ≡ :rwmh
#
# What runs where:
#   * src/subspace_construction.jl:45-52,61-65 (SWA update, deviation, append, psvd, P)        -> GPU, always
#   * src/subspace_construction.jl:39-43 (gradient + Flux.update!)                               -> Julia (Zygote + Flux)
#       by default: an arbitrary `cost` closure and optimiser cannot cross a C ABI;
#       with device_training = true (the caller asserts cost(m, x, y) == Flux.Losses.mse(m(x), y) and opt is a fresh
#       Descent / Momentum / ADAM)                                                                -> GPU (si_train_*)
#   * src/space_inference.jl:90-95 `density(z)` and :107 `ℓπ_grad(θ)`                          -> GPU (si_logdensity,
#       si_logdensity_grad: ONE reverse sweep instead of M-wide ForwardDiff duals; `backend` is accepted and unused)
#   * :rwmh / :mh sampler loop (:111-116)                                                        -> GPU (si_sample_rwmh)
#   * :mala / :hmc / :nuts sampler logic (:117-120, :139-160)                                    -> the reference's own
#       AdvancedMH / AdvancedHMC calls, unchanged, driven by the two device callbacks above
#   * output map :125                                                                            -> GPU (si_reconstruct)
#   * more than one GPU: one Julia process per GPU (Distributed workers), one Ctx each, joined by the RCCL communicator
#       INSIDE the library (si_comm_*): `init_gpus()`, then `subspace_inference(...; ngpu = 8, nchains = 8)`,
#       `sub_inference_chains`, `sharded_construct_finish!`, `sample_data_sharded` (section at the end of this file)
module SubspaceInferenceHIP

using Flux, Zygote, Random, Distributed
using Distributions: MvNormal
using AdvancedMH, AdvancedHMC
export subspace_construction, subspace_inference, sub_inference, inference, predict
export init_gpus, sub_inference_chains, sharded_construct_finish!, sample_data_sharded, bcast_subspace!

const LIB = get(ENV, "SUBSPACE_HIP_LIB", joinpath(@__DIR__, "..", "libsubspace_hip.so"))
const SI_F32, SI_F64 = Int32(0), Int32(1)

# mirrors `si_layer` (include/subspace_hip.h) field by field
struct SiLayer
    kind::Int32
    in::Int32
    out::Int32
    act::Int32
    w_off::Int64
    b_off::Int64
    kw::Int32
    kh::Int32
    cin::Int32
    cout::Int32
    wi::Int32
    hi::Int32
    sw::Int32
    sh::Int32
    pw::Int32
    ph::Int32
    dw::Int32
    dh::Int32
end
dense_layer(i, o, act, w_off, b_off) = SiLayer(0, i, o, act, w_off, b_off, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)

mutable struct Ctx
    h::Ptr{Cvoid}
    function Ctx(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:si_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32), r, device)
        rc == 0 || throw(unsafe_string(ccall((:si_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        c = new(r[])
        finalizer(x -> (x.h != C_NULL && ccall((:si_destroy, LIB), Int32, (Ptr{Cvoid},), x.h); x.h = C_NULL), c)
        return c
    end
end

# reference error convention: throw(::String) (src/space_inference.jl:42,103,162); -5 is the BoundsError of U[:,1:M]
function check(c::Ctx, rc::Int32)
    rc == 0 && return
    msg = unsafe_string(ccall((:si_last_error, LIB), Cstring, (Ptr{Cvoid},), c.h))
    rc == -5 ? throw(BoundsError(msg)) : throw(msg)
end

# same flattening as the reference (src/libs.jl:19-22), kept here so the wrapper does not depend on its internals
extract_params(ps) = mapreduce(p -> vec(p), vcat, ps)

act_id(f) = f === identity ? Int32(0) : f === relu ? Int32(1) : f === tanh ? Int32(2) :
            (f === σ || f === sigmoid) ? Int32(3) : f === leakyrelu ? Int32(4) : f === elu ? Int32(5) :
            f === softplus ? Int32(6) : f === selu ? Int32(7) :
            throw("Error: activation $f is not available on the device")   # (gelu / swish: not invertible from the output)

conv_out(wi, k, s, p, d) = div(wi + 2p - d * (k - 1) - 1, s) + 1

# static layer-offset table replacing the per-call Flux.destructure/re of model_re (src/libs.jl:55-57): Dense, Conv,
# MaxPool and flatten layers of any Chain; `insize` = (W, H, C) of one observation when the chain starts on images
# [upstream Flux 0.11.2 field names: Dense.W/.b/.σ; Conv.weight/.bias/.σ/.stride/.pad/.dilation; MaxPool.k/.pad/.stride]
function layer_table(model, insize = nothing)
    model isa Chain || throw("Error: model_re function is not available for this model")
    tbl, off = SiLayer[], 0
    whc = insize
    for l in model.layers
        if l isa Dense
            o, i = size(l.W)
            push!(tbl, dense_layer(i, o, act_id(l.σ), off, off + i * o))
            off += i * o + o
            whc = nothing
            continue
        end
        whc === nothing && throw("DimensionMismatch: Conv / MaxPool / flatten need (W, H, C, N) data in front of the Dense layers")
        wi, hi, c = whc
        if l isa Conv
            kw, kh, cin, cout = size(l.weight)
            (sw, sh), (dw, dh) = l.stride, l.dilation
            pw, ph = l.pad[1], l.pad[end - 1]          # Flux stores (lo, hi) per dimension; symmetric padding assumed
            (all(l.pad[1:2:end] .== l.pad[2:2:end]) || length(l.pad) == 2) || throw("Error: asymmetric Conv padding is not available on the device")
            wo, ho = conv_out(wi, kw, sw, pw, dw), conv_out(hi, kh, sh, ph, dh)
            push!(tbl, SiLayer(1, wi * hi * cin, wo * ho * cout, act_id(l.σ), off, off + length(l.weight),
                               kw, kh, cin, cout, wi, hi, sw, sh, pw, ph, dw, dh))
            off += length(l.weight) + cout
            whc = (wo, ho, cout)
        elseif l isa MaxPool
            (kw, kh), (sw, sh) = l.k, l.stride
            all(l.pad .== 0) || throw("Error: MaxPool padding is not available on the device")
            wo, ho = div(wi - kw, sw) + 1, div(hi - kh, sh) + 1
            push!(tbl, SiLayer(2, wi * hi * c, wo * ho * c, 0, 0, 0, kw, kh, c, c, wi, hi, sw, sh, 0, 0, 1, 1))
            whc = (wo, ho, c)
        elseif l === Flux.flatten
            push!(tbl, SiLayer(3, wi * hi * c, wi * hi * c, 0, 0, 0, 0, 0, c, c, wi, hi, 1, 1, 0, 0, 1, 1))
            whc = nothing
        else
            throw("Error: model_re function is not available for this model")
        end
    end
    return tbl, off
end

# split_data (src/libs.jl:75-77) as the (features x B) matrices the C ABI takes; a (W, H, C, N) array is the same memory
function data_matrices(data)
    X, Y = data.data[1], data.data[2]
    insize = ndims(X) == 4 ? size(X)[1:3] : nothing
    return Float64.(reshape(X, :, size(X)[end])), Float64.(reshape(Y, :, size(Y)[end])), insize
end
# the same for the training step, in the element type that decides its arithmetic: Float32 (X, Y) stay Float32 -- with a
# Float32 model that is an all-Float32 Zygote pass in the reference (src/subspace_construction.jl:39-43) -- anything else is
# promoted to Float64, as `W * x` would promote it
function train_matrices(data)
    X, Y = data.data[1], data.data[2]
    insize = ndims(X) == 4 ? size(X)[1:3] : nothing
    T = (eltype(X) == Float32 && eltype(Y) == Float32) ? Float32 : Float64
    return Array{T}(reshape(X, :, size(X)[end])), Array{T}(reshape(Y, :, size(Y)[end])), insize
end
const SI_DTYPE_OF_DATA = Int32(-1)
si_dtype(::Type{Float32}) = Int32(0)
si_dtype(::Type{Float64}) = Int32(1)

# ---- on-device training step (SURVEY 8 f1): src/subspace_construction.jl:39-43 for cost = mse ----------------------
# (kind, η, p1, p2) of a fresh Flux 0.11.2 optimiser [upstream field names: Descent.eta; Momentum.eta/.rho; ADAM.eta/.beta]
function device_optimiser(opt)
    opt isa Descent && return (Int32(0), Float64(opt.eta), 0.0, 0.0)
    opt isa Momentum && isempty(opt.velocity) && return (Int32(1), Float64(opt.eta), Float64(opt.rho), 0.0)
    opt isa ADAM && isempty(opt.state) && return (Int32(2), Float64(opt.eta), Float64(opt.beta[1]), Float64(opt.beta[2]))
    throw("Error: device_training needs a fresh Descent / Momentum / ADAM optimiser")
end

# the observation indices of every batch of one epoch, as the DataLoader would deliver them
# [upstream Flux 0.11.2 DataLoader: fields data, batchsize, nobs, partial, imax, indices, shuffle; shuffle!(indices) at i == 0]
function index_batches(d)
    d.shuffle && shuffle!(d.indices)
    out = Vector{Vector{Int64}}()
    i = 0
    while i < d.imax
        nexti = min(i + d.batchsize, d.nobs)
        push!(out, Int64.(d.indices[i+1:nexti]) .- 1)        # 0-based for the C ABI
        i += d.batchsize
    end
    return out
end

# data_parallel = true (opt-in): every rank of the ctx's communicator makes this call with the same model / data / DataLoader
# seed and takes its share of every batch.  The default keeps the step local whatever the ctx carries: a construction on one
# rank followed by bcast_subspace! (the cfg3 flow) must not wait alone inside a gradient all-reduce.
# compute_dtype: SI_DTYPE_OF_DATA (default: Float32 data => the Float32 pass, Float64 data => the Float64 pass, as in the
# reference), or si_dtype(Float32) / si_dtype(Float64) to override
function train_on_device!(ctx::Ctx, model, data, opt, T, c, print_freq; data_parallel = false, compute_dtype = SI_DTYPE_OF_DATA)
    X, Y, insize = train_matrices(data)
    tbl, N = layer_table(model, insize)
    kind, η, p1, p2 = device_optimiser(opt)
    ps = Flux.params(model)
    w0 = Float32.(extract_params(ps))
    GC.@preserve tbl w0 X Y check(ctx, ccall((:si_train_setup_ex, LIB), Int32,
        (Ptr{Cvoid}, Ptr{SiLayer}, Int32, Int64, Ptr{Float32}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32, Int64, Int64,
         Int32, Float64, Float64, Float64, Int32),
        ctx.h, tbl, length(tbl), N, w0, X, Y, si_dtype(eltype(X)), size(X, 1), size(Y, 1), size(X, 2),
        min(data.batchsize, data.nobs), kind, η, p1, p2, Int32(compute_dtype)))
    loss = Ref{Float64}(0.0)
    world, rank = comm_info(ctx)
    for i in 1:T
        for ids in index_batches(data)
            if data_parallel && world > 0
                # data-parallel step (SURVEY 8e): this rank's share of the batch; the N-double gradient and the SSE are
                # all-reduced inside the library (RCCL, in place, one grouped launch) -- every rank must iterate the
                # same batches (same DataLoader seed)
                lo, hi = col_shard(length(ids), rank, world)
                mine = ids[lo+1:hi]
                GC.@preserve mine check(ctx, ccall((:si_train_step_dp, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Int64}, Int64, Int64, Ref{Float64}), ctx.h, mine, length(mine), length(ids), loss))
            else
                GC.@preserve ids check(ctx, ccall((:si_train_step, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Float64}),
                                                  ctx.h, ids, length(ids), loss))
            end
            # :45-52 with W read in place from the device-resident Float32 weights (no extract_params, no PCIe)
            mod(i, c) == 0 && check(ctx, ccall((:si_train_push, LIB), Int32, (Ptr{Cvoid}, Float64), ctx.h, i / c))
        end
        if (mod(i, print_freq) == 0) || (i == T)
            println("Traing loss: ", loss[], " Epoch: ", i)
        end
    end
    # Flux.update! works in place: hand the trained weights back to the model's arrays
    w = Vector{Float32}(undef, N)
    GC.@preserve w check(ctx, ccall((:si_train_get_weights, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}), ctx.h, w))
    # ... and leave `opt` as Flux.update! would have (its IdDict state persists across calls in the reference)
    m = Vector{Float32}(undef, N); v = Vector{Float32}(undef, N); βp = Vector{Float64}(undef, 2)
    GC.@preserve m v βp check(ctx, ccall((:si_train_get_opt_state, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float64}), ctx.h, m, v, βp))
    off = 0
    for p in ps
        sl = off+1:off+length(p)
        copyto!(p, reshape(view(w, sl), size(p)))
        if opt isa Momentum            # [upstream Flux 0.11.2: Momentum.velocity::IdDict, x => v]
            opt.velocity[p] = reshape(m[sl], size(p))
        elseif opt isa ADAM            # [upstream Flux 0.11.2: ADAM.state::IdDict, x => (mt, vt, βp)]
            opt.state[p] = (reshape(m[sl], size(p)), reshape(v[sl], size(p)), [βp[1], βp[2]])
        end
        off += length(p)
    end
end

# init = :zeros is what the reference's code does (W_swa = zeros, :31); :pretrained is what its docs describe (nn_example.md:44)
function subspace_construction(model, cost, data, opt; T = 10, c = 1, M = 3, print_freq = 1, device = 0,
                               ctx = Ctx(device), max_cols = 0, keep_on_device = false, device_training = false,
                               init = :zeros, data_parallel = false, a_storage = :f64, compute_dtype = :auto)
    training_loss = 0.0
    ps = Flux.params(model)
    N = sum(length, ps)
    npush = count(i -> mod(i, c) == 0, 1:T) * length(data)
    check(ctx, ccall((:si_construct_begin, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int32), ctx.h, N, npush, max_cols))
    # a_storage = :f32 (opt-in): the deviation columns rounded once to Float32 on the device (half the memory / bytes of A)
    a_storage == :f32 && check(ctx, ccall((:si_construct_set_storage, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.h, SI_F32))
    if init == :pretrained
        W0 = extract_params(ps)
        GC.@preserve W0 check(ctx, ccall((:si_construct_set_mean, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32),
                                         ctx.h, pointer(W0), eltype(W0) == Float32 ? SI_F32 : SI_F64))
    end
    if device_training
        # compute_dtype = :auto: the device step computes in the DATA's element type, as the reference's Zygote pass does
        # (Float32 model + Float32 data: Float32 throughout; Float64 data: Float64); :f32 / :f64 override
        cd = compute_dtype == :auto ? SI_DTYPE_OF_DATA : compute_dtype == :f32 ? SI_F32 : SI_F64
        train_on_device!(ctx, model, data, opt, T, c, print_freq; data_parallel = data_parallel, compute_dtype = cd)
    else
        for i in 1:T
            for d in data
                gs = gradient(ps) do
                    training_loss = cost(model, d...)
                    return training_loss
                end
                Flux.update!(opt, ps, gs)
                if mod(i, c) == 0
                    W = extract_params(ps)                      # Float32 for Flux's default init
                    dt = eltype(W) == Float32 ? SI_F32 : SI_F64
                    GC.@preserve W check(ctx, ccall((:si_construct_push, LIB), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64), ctx.h, pointer(W), dt, i / c))
                end
            end
            if (mod(i, print_freq) == 0) || (i == T)
                println("Traing loss: ", training_loss, " Epoch: ", i)
            end
        end
    end
    W_swa = Vector{Float64}(undef, N)
    P = keep_on_device ? nothing : Matrix{Float64}(undef, N, M)
    s = Vector{Float64}(undef, M); K = Ref{Int64}(0)
    GC.@preserve W_swa P s check(ctx, ccall((:si_construct_finish, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}),
        ctx.h, M, W_swa, P === nothing ? C_NULL : pointer(P), s, K))
    return W_swa, P
end

# ---- the two device callbacks every sampler of sub_inference is driven by ------------------------------------------
# src/space_inference.jl:90-95  density(z)
function logdensity(ctx::Ctx, z::AbstractVector{<:Real})
    zz = Vector{Float64}(z); lp = Ref{Float64}(0.0)
    GC.@preserve zz check(ctx, ccall((:si_logdensity, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ref{Float64}),
                                     ctx.h, zz, 1, lp))
    return lp[]
end

# src/space_inference.jl:107  ℓπ_grad(θ) = (density(θ), gradient(density, θ))
function logdensity_grad(ctx::Ctx, z::AbstractVector{<:Real})
    zz = Vector{Float64}(z); lp = Ref{Float64}(0.0); g = Vector{Float64}(undef, length(zz))
    GC.@preserve zz g check(ctx, ccall((:si_logdensity_grad, LIB), Int32,
                                       (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}), ctx.h, zz, lp, g))
    return lp[], g
end

# callable handed to AdvancedMH's DensityModel; MALA asks the model for value + gradient through
# AdvancedMH.logdensity_and_gradient, which defaults to ForwardDiff on the closure [upstream AdvancedMH 0.6.2 src/MALA.jl]
# -- dual numbers cannot enter a ccall, so the device gradient is plugged in at that hook.
struct DeviceDensity <: Function
    ctx::Ctx
end
(d::DeviceDensity)(z) = logdensity(d.ctx, z)
AdvancedMH.logdensity_and_gradient(m::DensityModel{DeviceDensity}, θ) = logdensity_grad(m.logdensity.ctx, θ)

function reconstruct(ctx::Ctx, Z::Matrix{Float64}, N)
    Wm = Matrix{Float64}(undef, N, size(Z, 2))                      # :125 map(z -> W_swa + P*z.params, chm)
    GC.@preserve Z Wm check(ctx, ccall((:si_reconstruct, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}), ctx.h, Z, size(Z, 2), Wm))
    return [Wm[:, t] for t in 1:size(Z, 2)]
end

# include_prior = true adds the term the reference writes after its `return` (dead code, :95); default: as the reference
# compute_dtype = :f32 (non-default; SURVEY section 0 Q6): the density of a Dense or Conv chain on the fp32 matrix instructions -- X rounded
# once, W_swa + P*z formed in Float64 and rounded once per transition, Float32 activations, head + sum of squared errors in Float64
function sub_inference(in_model, data, W_swa, P; σ_z = 1.0, σ_m = 1.0, σ_p = 1.0, itr = 100, M = 3, alg = :rwmh,
                       backend = :forwarddiff, device = 0, ctx = Ctx(device), seed = 0, chain_id = 0, include_prior = false,
                       compute_dtype = :f64)
    alg == :mh && (alg = :rwmh)                                     # README.md:153-154
    alg in (:rwmh, :mala, :hmc, :nuts) || throw("$alg is not available")       # :162 (:advi is outside this build)
    in_model isa Chain || throw("Error: density function is not avaliable for this model")
    compute_dtype in (:f64, :f32) || throw("compute_dtype must be :f64 (the reference's arithmetic) or :f32")
    X, Y, insize = data_matrices(data)                              # split_data (src/libs.jl:75-77)
    tbl, N = layer_table(in_model, insize)
    Wp = W_swa === nothing ? C_NULL : pointer(W_swa)
    Pp = P === nothing ? C_NULL : pointer(P)
    GC.@preserve tbl W_swa P X Y check(ctx, ccall((:si_infer_setup, LIB), Int32,
        (Ptr{Cvoid}, Ptr{SiLayer}, Int32, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Int32, Int32, Int64, Float64, Int32),
        ctx.h, tbl, length(tbl), N, M, Wp, Pp, X, Y, size(X, 1), size(Y, 1), size(X, 2), σ_m, compute_dtype == :f32 ? SI_F32 : SI_F64))
    check(ctx, ccall((:si_infer_set_prior, LIB), Int32, (Ptr{Cvoid}, Float64), ctx.h, include_prior ? Float64(σ_p) : 0.0))
    if alg == :rwmh
        # :111-116 on the device: chain state, proposals (Philox) and accept decisions never leave the GPU
        # ... and the output map (:125) streams out while the chain runs (K4's own output, selected on accept; the DMA and
        # the host copy of sample t hide under transitions t+1 ..): no second pass over P, no PCIe wait at the end
        Z = Matrix{Float64}(undef, M, itr); lp = Vector{Float64}(undef, itr); acc = Ref{Float64}(0.0)
        Wm = Matrix{Float64}(undef, N, itr)
        GC.@preserve Z lp Wm check(ctx, ccall((:si_sample_rwmh_weights, LIB), Int32,
            (Ptr{Cvoid}, Int64, Float64, UInt64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}),
            ctx.h, itr, σ_z, seed, chain_id, 1, Z, lp, acc, Wm))
        return [Wm[:, t] for t in 1:itr], lp
    end
    density = DeviceDensity(ctx)
    ℓπ_grad(θ) = logdensity_grad(ctx, θ)
    if alg == :mala
        # :117-120, the reference's own AdvancedMH calls
        spl = MALA(x -> MvNormal((σ_z^2 / 2) .* x, σ_z))
        chm = sample(DensityModel(density), spl, itr; init_params = rand(MvNormal(zeros(M), σ_z)))
        Z = reduce(hcat, [Vector{Float64}(t.params) for t in chm])
        return reconstruct(ctx, Z, N), map(t -> t.lp, chm)
    end
    # :139-160, the reference's own AdvancedHMC calls
    initial_θ = rand(MvNormal(zeros(M), σ_z))
    n_samples, n_adapts = itr, Int(round(itr / 2))
    metric = DiagEuclideanMetric(M)
    hamiltonian = Hamiltonian(metric, density, ℓπ_grad)
    integrator = Leapfrog(find_good_stepsize(hamiltonian, initial_θ))
    proposal = alg == :hmc ? AdvancedHMC.StaticTrajectory(integrator, 1) :
                             AdvancedHMC.NUTS{MultinomialTS,GeneralisedNoUTurn}(integrator)
    adaptor = StanHMCAdaptor(MassMatrixAdaptor(metric), StepSizeAdaptor(0.8, integrator))
    samples, stats = sample(hamiltonian, proposal, initial_θ, n_samples, adaptor, n_adapts; progress = true)
    return reconstruct(ctx, reduce(hcat, samples), N), map(s -> s.log_density, stats)
end

inference(args...; kwargs...) = sub_inference(args...; kwargs...)

# posterior predictive on new inputs without materialising weight samples (SURVEY 8 f3): what the reference's users
# compute as `re(chn[i])(x)` per sample (docs/src/nn_example.md:207-217).  Z is M x C, Xnew is in x Bn.
function predict(ctx::Ctx, Z::Matrix{Float64}, Xnew::Matrix{Float64}, out_dim::Integer)
    Yh = Array{Float64}(undef, out_dim, size(Xnew, 2), size(Z, 2))
    GC.@preserve Z Xnew Yh check(ctx, ccall((:si_predict, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Int64, Ptr{Float64}), ctx.h, Z, size(Z, 2), Xnew, size(Xnew, 2), Yh))
    return Yh
end

# =====================================================================================================================
# More than one GPU (SURVEY 2.2 R1, 8e).  One Julia PROCESS per GPU (Distributed workers, `addprocs(8)` +
# `@everywhere using SubspaceInferenceHIP`), one Ctx per process, joined by an RCCL communicator that lives INSIDE the
# library (si_comm_*): every collective runs in place on the library's device buffers, on its stream.  Julia ships only
# the 128-byte id (remotecall) and, at the end, the small (Z, lp) results.  The reference is single-process; the loops
# these calls parallelise are src/subspace_construction.jl:37-59 and src/space_inference.jl:94,116.
# =====================================================================================================================
const RANK_CTX = Ref{Union{Nothing,Ctx}}(nothing)        # the ctx of THIS process once init_gpus has run
rank_ctx() = RANK_CTX[] === nothing ? throw("Error: this process has no GPU context (call init_gpus first)") : RANK_CTX[]

function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = GC.@preserve id ccall((:si_comm_unique_id, LIB), Int32, (Ptr{UInt8},), id)
    rc == 0 || throw(unsafe_string(ccall((:si_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    return id
end

comm_init!(ctx::Ctx, world, rank, id::Vector{UInt8}) = GC.@preserve id check(ctx, ccall((:si_comm_init_rank, LIB), Int32,
    (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), ctx.h, world, rank, id))
comm_destroy!(ctx::Ctx) = check(ctx, ccall((:si_comm_destroy, LIB), Int32, (Ptr{Cvoid},), ctx.h))

# (world, rank); world == 0 when the ctx has no communicator
function comm_info(ctx::Ctx)
    w = Ref{Int32}(0); r = Ref{Int32}(0); v = Ref{Int32}(0)
    check(ctx, ccall((:si_comm_info, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}), ctx.h, w, r, v))
    return Int(w[]), Int(r[])
end

comm_barrier(ctx::Ctx) = check(ctx, ccall((:si_comm_barrier, LIB), Int32, (Ptr{Cvoid},), ctx.h))

# small host values over the communicator (losses, timings): op 0 = sum, 1 = max
function comm_allreduce!(ctx::Ctx, v::Vector{Float64}; op = 0)
    GC.@preserve v check(ctx, ccall((:si_comm_allreduce_host, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int32),
                                    ctx.h, v, length(v), op))
    return v
end

function comm_allgather(ctx::Ctx, v::Vector{Float64})
    world, _ = comm_info(ctx)
    out = Matrix{Float64}(undef, length(v), world)
    GC.@preserve v out check(ctx, ccall((:si_comm_allgather_host, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                                        ctx.h, v, length(v), out))
    return out
end

# rows [r0, r1) (0-based, r1 exclusive) of rank `rank`: the partition every sharded entry point of the library assumes
function row_shard(n_total, rank, world)
    r0 = Ref{Int64}(0); r1 = Ref{Int64}(0)
    ccall((:si_row_shard, LIB), Int32, (Int64, Int32, Int32, Ref{Int64}, Ref{Int64}), n_total, rank, world, r0, r1) == 0 ||
        throw("si_row_shard: bad argument")
    return Int(r0[]), Int(r1[])
end

# observation block (0-based, exclusive end) of rank `rank`: contiguous, sizes differ by at most one
function col_shard(b, rank, world)
    base, rem = divrem(b, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (rank < rem ? 1 : 0)
end

"""
    init_gpus(ws = workers())

One GPU per Distributed worker: worker `ws[r]` creates its `Ctx(r - 1)` and joins an RCCL communicator of
`length(ws)` ranks (rank = r - 1).  Run once after `addprocs(n); @everywhere using SubspaceInferenceHIP`.
"""
function init_gpus(ws::Vector{Int} = workers())
    id = comm_unique_id()                                   # needs no GPU; travels to the workers by remotecall
    world = length(ws)
    # `world` processes share this host's CPU quota: each worker's host copy pool gets its share before its first copy
    # (the library also shrinks the pool by itself in si_comm_init_rank; SI_HOST_COPY_THREADS set by the user wins)
    share = max(1, ccall((:si_host_cpu_budget, LIB), Cint, ()) ÷ (2 * world))
    @sync for (r, w) in enumerate(ws)
        @async remotecall_wait(w, id, world, r - 1) do id, world, rank
            haskey(ENV, "SI_HOST_COPY_THREADS") || (ENV["SI_HOST_COPY_THREADS"] = string(share))
            ctx = Ctx(rank)
            comm_init!(ctx, world, rank, id)
            RANK_CTX[] = ctx
            nothing
        end
    end
    return ws
end

# (W_swa, P, s) of the construction finished on rank `root` -> every rank's ctx, device to device (168 MB at cfg2, once);
# collective: every rank calls it.  Receivers then hold a finished construction (sub_inference(..., nothing, nothing)).
bcast_subspace!(ctx::Ctx, root, N, M) = check(ctx, ccall((:si_bcast_subspace, LIB), Int32,
    (Ptr{Cvoid}, Int32, Int64, Int32), ctx.h, root, N, M))

"""
    sharded_construct_finish!(ctx, M, n_total; gather = true)

Row-sharded `psvd` + `P` (src/subspace_construction.jl:63,65): this rank pushed ITS rows `row_shard(n_total, rank, world)`
of every snapshot (si_construct_begin with N = r1 - r0).  ONE all-reduce of the K x K Gram matrix inside the library
(a second one on the ill-conditioned route), replicated eigensolve, row-local projection; `gather` assembles the full
(W_swa, P) on every rank, device to device.  Returns the M singular values.
"""
function sharded_construct_finish!(ctx::Ctx, M, n_total; gather = true)
    check(ctx, ccall((:si_construct_gram, LIB), Int32, (Ptr{Cvoid},), ctx.h))
    check(ctx, ccall((:si_construct_allreduce_gram, LIB), Int32, (Ptr{Cvoid},), ctx.h))
    flag = Ref{Int32}(0)
    check(ctx, ccall((:si_construct_needs_refine, LIB), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}), ctx.h, M, flag))
    if flag[] != 0
        check(ctx, ccall((:si_construct_refine, LIB), Int32, (Ptr{Cvoid},), ctx.h))
        check(ctx, ccall((:si_construct_allreduce_gram, LIB), Int32, (Ptr{Cvoid},), ctx.h))
    end
    s = Vector{Float64}(undef, M); K = Ref{Int64}(0)
    GC.@preserve s check(ctx, ccall((:si_construct_finish, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}), ctx.h, M, C_NULL, C_NULL, s, K))
    gather && check(ctx, ccall((:si_construct_allgather, LIB), Int32, (Ptr{Cvoid}, Int64), ctx.h, n_total))
    return s
end

# host copies of the finished construction a ctx holds (its own, or one received by bcast_subspace! / the all-gather)
function construct_result(ctx::Ctx)
    N = Ref{Int64}(0); M = Ref{Int32}(0)
    check(ctx, ccall((:si_construct_get_result, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Ref{Int32}), ctx.h, C_NULL, C_NULL, C_NULL, N, M))
    W = Vector{Float64}(undef, N[]); P = Matrix{Float64}(undef, N[], M[]); s = Vector{Float64}(undef, M[])
    GC.@preserve W P s check(ctx, ccall((:si_construct_get_result, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Ref{Int32}), ctx.h, W, P, s, N, M))
    return W, P, s
end

"""
    sample_data_sharded(ctx, itr, σ_z, d_total, M; seed, chain_id, nchains)

RWMH with the observations split over the ranks (BASELINE cfg5): `ctx` was set up (si_infer_setup) with THIS rank's
column block of (X, Y) and the full W_swa / P.  One library call: per transition eval -> RCCL all-reduce of the
partial sums of squared errors -> accept, on one stream.  Every rank returns the same chain.
"""
function sample_data_sharded(ctx::Ctx, itr, σ_z, d_total, M; seed = 0, chain_id = 0, nchains = 1)
    Z = Array{Float64}(undef, M, itr, nchains); lp = Matrix{Float64}(undef, itr, nchains); acc = Vector{Float64}(undef, nchains)
    GC.@preserve Z lp acc check(ctx, ccall((:si_sample_rwmh_sharded, LIB), Int32,
        (Ptr{Cvoid}, Int64, Float64, UInt64, Int32, Int32, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, itr, σ_z, seed, chain_id, nchains, d_total, Z, lp, acc))
    return Z, lp, acc
end

"""
    sub_inference_chains(in_model, data, W_swa, P; nchains, ws = workers(), kwargs...)

`nchains` independent chains of `sub_inference` spread over the GPUs of `ws` (BASELINE cfg3: one chain per GPU, no
exchange per transition).  Chain c uses the library's Philox stream c, so the union over ranks is what one GPU running
all chains would produce.  `W_swa === nothing`: every rank's ctx already holds the subspace (bcast_subspace!).
Returns `(chains::Vector, lps::Vector)` in chain order.
"""
function sub_inference_chains(in_model, data, W_swa, P; nchains, ws::Vector{Int} = workers(), seed = 0, kwargs...)
    world = length(ws)
    futs = map(enumerate(ws)) do (r, w)
        base, rem = divrem(nchains, world)
        lo = (r - 1) * base + min(r - 1, rem)
        ids = lo:(lo + base + ((r - 1) < rem ? 1 : 0) - 1)
        remotecall(w, in_model, data, W_swa, P, collect(ids), seed, kwargs) do m, d, Ws, Pm, ids, seed, kw
            [sub_inference(m, d, Ws, Pm; ctx = rank_ctx(), seed = seed, chain_id = c, kw...) for c in ids]
        end
    end
    res = reduce(vcat, map(fetch, futs))
    return map(first, res), map(last, res)
end

# subspace_inference on `ngpu` GPUs: construction on every rank (data-parallel device training: identical (W_swa, P)
# everywhere, one gradient all-reduce per step) or on rank 0 followed by ONE device-to-device broadcast; then `nchains`
# independent chains.  Called by subspace_inference(...; ngpu > 1).
function subspace_inference_multi(model, cost, data, opt; ngpu, nchains, ws::Vector{Int} = workers()[1:ngpu], σ_z, σ_m, σ_p, itr,
                                  T, c, M, print_freq, alg, backend, device_training)
    length(ws) == ngpu || throw("Error: ngpu = $ngpu needs that many initialised workers (init_gpus)")
    N = sum(length, Flux.params(model))
    outs = map(enumerate(ws)) do (r, w)
        remotecall(w, model, cost, data, opt, r - 1) do m, cst, d, o, rank
            ctx = rank_ctx()
            Wl = nothing
            if device_training || rank == 0
                # (device_training: every rank runs the SAME loop on its share of each batch -> same result everywhere)
                Wl, _ = subspace_construction(m, cst, d, o; T = T, c = c, M = M, print_freq = print_freq, ctx = ctx,
                                              keep_on_device = true, device_training = device_training)
            end
            device_training || bcast_subspace!(ctx, 0, N, M)     # collective: all ranks
            rank == 0 ? Wl : nothing
        end
    end
    W_swa = first(filter(x -> x !== nothing, map(fetch, outs)))
    chains, lps = sub_inference_chains(model, data, nothing, nothing; nchains = nchains, ws = ws, σ_z = σ_z, σ_m = σ_m,
                                       σ_p = σ_p, itr = itr, M = M, alg = alg, backend = backend)
    return chains, lps, W_swa
end

# ngpu / nchains are additions: ngpu = 1 (default) is the reference's single chain on one GPU; ngpu > 1 runs `nchains`
# independent chains on the GPUs prepared by init_gpus and returns a Vector of chains / lps (chain order)
function subspace_inference(model, cost, data, opt; σ_z = 1.0, σ_m = 1.0, σ_p = 1.0, itr = 1000, T = 25, c = 1, M = 20,
                            print_freq = 1, alg = :rwmh, backend = :forwarddiff, method = :subspace, device = 0,
                            device_training = false, ngpu = 1, nchains = ngpu)
    method == :subspace || throw("Error: No method found")
    ngpu > 1 && return subspace_inference_multi(model, cost, data, opt; ngpu = ngpu, nchains = nchains, σ_z = σ_z, σ_m = σ_m,
                                                σ_p = σ_p, itr = itr, T = T, c = c, M = M, print_freq = print_freq, alg = alg,
                                                backend = backend, device_training = device_training)
    ctx = Ctx(device)
    W_swa, _ = subspace_construction(model, cost, data, opt; T = T, c = c, M = M, print_freq = print_freq, ctx = ctx,
                                     keep_on_device = true, device_training = device_training)
    chn, lp = sub_inference(model, data, nothing, nothing; σ_z = σ_z, σ_m = σ_m, σ_p = σ_p, itr = itr, M = M,
                            alg = alg, backend = backend, ctx = ctx)   # W_swa / P taken in place on the device
    return chn, lp, W_swa
end

end # module
