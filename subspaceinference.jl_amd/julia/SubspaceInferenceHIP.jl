# SubspaceInferenceHIP.jl -- the `ccall` side of the drop-in boundary (include/subspace_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: Julia is absent from the build image and from the GPU boxes, so this file is
# the binding a maintainer of efmanu/SubspaceInference.jl would add (INTEGRATION.md); every executable test goes
# through the Python ctypes binding (_capi.py) over the IDENTICAL C ABI.
#
# It keeps the reference's exported names and keyword arguments (src/SubspaceInference.jl:27-34):
#   subspace_construction(model, cost, data, opt; T, c, M, print_freq)              -> W_swa, P
#   subspace_inference(model, cost, data, opt; σ_z, σ_m, σ_p, itr, T, c, M, ...)      -> chn, lp, W_swa
#   sub_inference(in_model, data, W_swa, P; σ_z, σ_m, σ_p, itr, M, alg, backend)      -> chn, lp
#   inference(...)  (README.md:153-154 name), alg = :mh ≡ :rwmh
# The gradient / optimiser step (src/subspace_construction.jl:39-43) stays in Julia (Zygote + Flux): an arbitrary
# `cost` closure and optimiser cannot cross a C ABI.  Everything after `extract_params` runs on the GPU.
module SubspaceInferenceHIP

using Flux, Zygote
export subspace_construction, subspace_inference, sub_inference, inference

const LIB = get(ENV, "SUBSPACE_HIP_LIB", joinpath(@__DIR__, "..", "libsubspace_hip.so"))
const SI_F32, SI_F64 = Int32(0), Int32(1)

struct SiLayer
    kind::Int32; in::Int32; out::Int32; act::Int32; w_off::Int64; b_off::Int64
end

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
            (f === σ || f === sigmoid) ? Int32(3) : throw("Error: activation $f is not available on the device")

# static layer-offset table replacing the per-call Flux.destructure/re of model_re (src/libs.jl:55-57)
function layer_table(model)
    model isa Chain || throw("Error: model_re function is not available for this model")
    tbl, off = SiLayer[], 0
    for l in model.layers
        l isa Dense || throw("Error: model_re function is not available for this model")
        o, i = size(l.W)
        push!(tbl, SiLayer(0, i, o, act_id(l.σ), off, off + i * o))
        off += i * o + o
    end
    return tbl, off
end

function subspace_construction(model, cost, data, opt; T = 10, c = 1, M = 3, print_freq = 1, device = 0,
                               ctx = Ctx(device), max_cols = 0, keep_on_device = false)
    training_loss = 0.0
    ps = Flux.params(model)
    N = sum(length, ps)
    npush = count(i -> mod(i, c) == 0, 1:T) * length(data)
    check(ctx, ccall((:si_construct_begin, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int32), ctx.h, N, npush, max_cols))
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
    W_swa = Vector{Float64}(undef, N)
    P = keep_on_device ? nothing : Matrix{Float64}(undef, N, M)
    s = Vector{Float64}(undef, M); K = Ref{Int64}(0)
    GC.@preserve W_swa P s check(ctx, ccall((:si_construct_finish, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}),
        ctx.h, M, W_swa, P === nothing ? C_NULL : pointer(P), s, K))
    return W_swa, P
end

function sub_inference(in_model, data, W_swa, P; σ_z = 1.0, σ_m = 1.0, σ_p = 1.0, itr = 100, M = 3, alg = :rwmh,
                       backend = :forwarddiff, device = 0, ctx = Ctx(device), seed = 0, chain_id = 0)
    (alg == :rwmh || alg == :mh) || throw("$alg is not available")
    in_model isa Chain || throw("Error: density function is not avaliable for this model")
    X, Y = Float64.(data.data[1]), Float64.(data.data[2])          # split_data (src/libs.jl:75-77)
    tbl, N = layer_table(in_model)
    Wp = W_swa === nothing ? C_NULL : pointer(W_swa)
    Pp = P === nothing ? C_NULL : pointer(P)
    GC.@preserve tbl W_swa P X Y check(ctx, ccall((:si_infer_setup, LIB), Int32,
        (Ptr{Cvoid}, Ptr{SiLayer}, Int32, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Int32, Int32, Int64, Float64, Int32),
        ctx.h, tbl, length(tbl), N, M, Wp, Pp, X, Y, size(X, 1), size(Y, 1), size(X, 2), σ_m, SI_F64))
    Z = Matrix{Float64}(undef, M, itr); lp = Vector{Float64}(undef, itr); acc = Ref{Float64}(0.0)
    GC.@preserve Z lp check(ctx, ccall((:si_sample_rwmh, LIB), Int32,
        (Ptr{Cvoid}, Int64, Float64, UInt64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
        ctx.h, itr, σ_z, seed, chain_id, 1, Z, lp, acc))
    Wm = Matrix{Float64}(undef, N, itr)                             # :125 map(z -> W_swa + P*z.params, chm)
    GC.@preserve Z Wm check(ctx, ccall((:si_reconstruct, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}), ctx.h, Z, itr, Wm))
    return [Wm[:, t] for t in 1:itr], lp
end

inference(args...; kwargs...) = sub_inference(args...; kwargs...)

function subspace_inference(model, cost, data, opt; σ_z = 1.0, σ_m = 1.0, σ_p = 1.0, itr = 1000, T = 25, c = 1, M = 20,
                            print_freq = 1, alg = :rwmh, backend = :forwarddiff, method = :subspace, device = 0)
    method == :subspace || throw("Error: No method found")
    ctx = Ctx(device)
    W_swa, _ = subspace_construction(model, cost, data, opt; T = T, c = c, M = M, print_freq = print_freq, ctx = ctx,
                                     keep_on_device = true)
    chn, lp = sub_inference(model, data, nothing, nothing; σ_z = σ_z, σ_m = σ_m, σ_p = σ_p, itr = itr, M = M,
                            alg = alg, backend = backend, ctx = ctx)   # W_swa / P taken in place on the device
    return chn, lp, W_swa
end

end # module
