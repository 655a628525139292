# make_golden_reference.jl -- UNEXECUTED in this repository (Julia is absent from its images).
#
# The only way "parity unpinned" ever closes: a maintainer with Julia 1.5 and the reference's Manifest.toml pins
# (Flux 0.11.2, Zygote 0.5.17, LowRankApprox 0.5.0, Distributions 0.24.18, AdvancedMH 0.6.2, NNlib 0.7.23) runs this
# script against the REAL package and commits the .npz files it writes next to the oracle-generated ones.  It produces the
# same keys as tests/golden/make_golden.py, from the same inputs (read from the committed oracle fixtures, so both sides
# see identical bits), with every quantity computed by the reference's own code:
#
#     julia --project=/path/to/SubspaceInference.jl tests/golden/make_golden_reference.jl
#     python -m pytest tests/test_oracle.py -k reference_fixtures   # holds the oracle against *_reference.npz when present
#
# What can be pinned this way (deterministic parts): W_swa, A (bit for bit), P up to column sign and s (psvd, rtol 1e-8),
# log-density values, forward outputs, the gradient of the log-density, the optimiser steps.  What cannot: the RWMH chain
# itself (Julia's MersenneTwister against the library's Philox stream) -- only its accept rule, checked on a scripted
# sequence of (lp, lp', randexp) triples.
using SubspaceInference, Flux, Zygote, LowRankApprox, Distributions, LinearAlgebra
using NPZ   # ] add NPZ

here = @__DIR__
toy = npzread(joinpath(here, "toy_construct_k12.npz"))
den = npzread(joinpath(here, "toy_density_rwmh.npz"))
trn = npzread(joinpath(here, "toy_train_steps.npz"))

# ---- construction: src/subspace_construction.jl:31-33,45-52,61-65 on the committed snapshot stream ------------------
snaps, ns = toy["snapshots"], toy["ns"]                 # K x N Float32, K
N = size(snaps, 2)
W_swa = zeros(N)                                        # :31 (Q1)
A = Float64[]
for k in 1:size(snaps, 1)
    W = snaps[k, :]                                     # what extract_params(ps) returned at that push
    n = ns[k]
    global W_swa = (n .* W_swa + W) ./ (n + 1)          # :47
    W_dev = W - W_swa                                   # :51
    append!(A, W_dev)                                   # :52
end
A = reshape(A, N, :)                                    # :61
U, s, V = psvd(A)                                       # :63
M = 3
P = U[:, 1:M] * Diagonal(s[1:M])                        # :65
npzwrite(joinpath(here, "toy_construct_k12_reference.npz"), Dict("W_swa" => W_swa, "A" => A, "P" => P, "s" => s[1:M]))

# ---- the same at the README toy's REAL shape (README.md:52-79: K = 1000 deviation columns of N = 682 weights, K > N) -------
big = npzread(joinpath(here, "toy_construct_k1000.npz"))
let snaps = big["snapshots"], ns = big["ns"], N = size(big["snapshots"], 2)
    W = zeros(N); Ab = Float64[]
    for k in 1:size(snaps, 1)
        w = snaps[k, :]; n = ns[k]
        W = (n .* W + w) ./ (n + 1)                     # :47
        append!(Ab, w - W)                              # :51-52
    end
    Ab = reshape(Ab, N, :)                              # :61
    Ub, sb, Vb = psvd(Ab)                               # :63
    npzwrite(joinpath(here, "toy_construct_k1000_reference.npz"), Dict("W_swa" => W, "P" => Ub[:, 1:3] * Diagonal(sb[1:3]), "s" => sb[1:3]))
end

# ---- density: src/space_inference.jl:90-95 through the package's own model_re ---------------------------------------
model = Chain(Dense(10, 20), Dense(20, 20), Dense(20, 2))           # README.md:62
X, Y = den["X"], den["Y"]
Wd, Pd, Z = den["W_swa"], den["P"], den["Z"]
function density(z)
    new_W = Wd + Pd * z                                              # :91
    new_model = SubspaceInference.model_re(model, new_W)             # :92
    return logpdf(MvNormal(vec(new_model(X)), 1.0), vec(Y))          # :94 (the line after it is dead code)
end
lp = [density(Z[:, j]) for j in 1:size(Z, 2)]
Yhat0 = SubspaceInference.model_re(model, Wd + Pd * Z[:, 1])(X)
grad1 = Zygote.gradient(density, Z[:, 2])[1]                          # what ℓπ_grad (:107) differentiates
npzwrite(joinpath(here, "toy_density_rwmh_reference.npz"), Dict("lp" => lp, "Yhat0" => Yhat0, "grad1" => grad1))

# ---- training step: src/subspace_construction.jl:39-43 with Flux's own optimisers -----------------------------------
tx, ty, w0, batches = trn["X"], trn["Y"], trn["w0"], trn["batches"]
out = Dict{String,Any}()
for (name, mk) in (("descent", () -> Descent(0.1)), ("momentum", () -> Momentum(0.01, 0.9)), ("adam", () -> ADAM(0.001, (0.9, 0.999))))
    m = Chain(Dense(10, 20, tanh), Dense(20, 20, relu), Dense(20, 2))
    θ, re = Flux.destructure(m)
    m = re(Float32.(w0))                                 # the fixture's initial weights, Float32
    ps, opt = Flux.params(m), mk()
    losses = Float64[]
    for b in 1:size(batches, 1)
        ids = filter(i -> i >= 0, batches[b, :]) .+ 1
        local training_loss
        gs = gradient(ps) do
            training_loss = Flux.Losses.mse(m(tx[:, ids]), ty[:, ids])
            return training_loss
        end
        Flux.update!(opt, ps, gs)
        push!(losses, training_loss)
    end
    out[name * "_w"] = SubspaceInference.extract_params(ps)
    out[name * "_loss"] = losses
end
npzwrite(joinpath(here, "toy_train_steps_reference.npz"), out)

# ---- the same with Float32 DATA (toy_train_steps_f32.npz): nothing promotes, the Zygote pass and `update!` are Float32 ------
trn32 = npzread(joinpath(here, "toy_train_steps_f32.npz"))
tx32, ty32 = Float32.(trn32["X"]), Float32.(trn32["Y"])
out32 = Dict{String,Any}()
for (name, mk) in (("descent", () -> Descent(0.1)), ("momentum", () -> Momentum(0.01, 0.9)), ("adam", () -> ADAM(0.001, (0.9, 0.999))))
    m = Chain(Dense(10, 20, tanh), Dense(20, 20, relu), Dense(20, 2))
    θ, re = Flux.destructure(m)
    m = re(Float32.(trn32["w0"]))
    ps, opt = Flux.params(m), mk()
    losses = Float64[]
    for b in 1:size(batches, 1)
        ids = filter(i -> i >= 0, batches[b, :]) .+ 1
        local training_loss
        gs = gradient(ps) do
            training_loss = Flux.Losses.mse(m(tx32[:, ids]), ty32[:, ids])
            return training_loss
        end
        Flux.update!(opt, ps, gs)
        push!(losses, Float64(training_loss))
    end
    out32[name * "_w"] = SubspaceInference.extract_params(ps)
    out32[name * "_loss"] = losses
end
npzwrite(joinpath(here, "toy_train_steps_f32_reference.npz"), out32)
println("wrote *_reference.npz next to the oracle fixtures")
