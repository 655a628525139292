"""CPU restatement (NumPy, fp64) of SubspaceInference.jl's hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (subspaceinference.jl_amd) never does; it fails loudly without its HIP library.

PARITY UNPINNED.  The reference (/root/reference, Julia) ships an empty test-suite
(test/runtests.jl is 0 bytes), no fixtures and no golden vectors, and Julia is not
installed in this image, so the reference itself cannot be run.  This restatement follows
the reference SOURCE line by line (citations below) and is cross-checked only by identities
that do not depend on it being right (tests/test_oracle.py: closed-form EMA, LAPACK SVD
identities, scipy.stats logpdf, a hand-computed forward pass, RWMH on a Gaussian target).

Every function cites the reference file:line it restates (paths relative to /root/reference).
Quirks of the reference CODE are reproduced on purpose:
  Q1 W_swa starts at zero                      src/subspace_construction.jl:31
  Q2 n = i/c is the EPOCH counter, per batch   src/subspace_construction.jl:44-47
  Q3 no column shift, A keeps every column     src/subspace_construction.jl:48-52
  Q4 prior term is dead code                   src/space_inference.jl:94-95
  Q5 density uses the full (X, Y)              src/libs.jl:75-77
  Q6 W_swa / A / P / density are Float64       src/subspace_construction.jl:31,33
"""
import math

import numpy as np

from . import philox

ACT_IDENTITY, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
# [upstream Flux 0.11.2 / NNlib 0.7.23 src/activation.jl]: leakyrelu(x, a = 0.01) = max(a x, x); elu(x, alpha = 1) =
# ifelse(x >= 0, x, alpha (exp(x) - 1)); softplus(x) = ifelse(x > 0, x + log1p(exp(-x)), log1p(exp(x)));
# selu(x) = lambda ifelse(x > 0, x, alpha (exp(x) - 1))   -- `exp(x) - 1` as written there, not expm1
ACT_LEAKYRELU, ACT_ELU, ACT_SOFTPLUS, ACT_SELU = 4, 5, 6, 7
SELU_LAMBDA, SELU_ALPHA = 1.0507009873554805, 1.6732632423543772


# --------------------------------------------------------------------------- layout
def layer_table(dims, acts):
    """Flat-vector layout of a Dense chain: [vec(W1) (out x in, column-major); b1; vec(W2); b2; ...].

    Restates the ordering produced by `extract_params` (src/libs.jl:19-22, Flux.params order: W then b
    per Dense layer) and consumed by `Flux.destructure`/`re` in `model_re` (src/libs.jl:55-57).
    dims = [in, h1, ..., out]; returns list of (in, out, act, w_off, b_off) and total N.
    """
    table, off = [], 0
    for l in range(len(dims) - 1):
        fin, fout = dims[l], dims[l + 1]
        w_off = off
        off += fin * fout
        b_off = off
        off += fout
        table.append((fin, fout, acts[l], w_off, b_off))
    return table, off


def extract_params(weights):
    """src/libs.jl:19-22 -- `mapreduce(i -> vec(ps.order.data[i]), vcat, ...)`: column-major vec of each
    array, concatenated; dtype follows the arrays (Float32 for Flux 0.11 default init)."""
    return np.concatenate([np.asarray(a).reshape(-1, order="F") for a in weights])


# --------------------------------------------------------------------------- construction
def swa_dev_push(w_swa, w, n):
    """src/subspace_construction.jl:46-47,51.

        W_swa = (n.*W_swa + W)./(n+1)     # three separate rounded ops per element: mul, add, div
        W_dev = W - W_swa                 # uses the UPDATED mean

    `w` may be float32 (promoted exactly to float64, as Julia's `+` does).  No FMA: Julia materialises
    `n.*W_swa` before the `+`.  Returns (new W_swa, W_dev)."""
    w64 = np.asarray(w).astype(np.float64)
    t = np.float64(n) * w_swa
    u = t + w64
    new = u / (np.float64(n) + 1.0)
    return new, w64 - new


def construct_stream(snapshots, ns, w_init=None):
    """src/subspace_construction.jl:31-33,44-52,61: W_swa0 = zeros (Q1), push every snapshot with its
    caller-supplied n (Q2), keep every deviation column (Q3).  Returns (W_swa, A) with A N x K col-major."""
    n_par = len(snapshots[0])
    # Q1: zeros(...) in the code; `w_init` restates the docs' "W_swa = W_0" (docs/src/nn_example.md:44), the non-default option
    w_swa = np.zeros(n_par, dtype=np.float64) if w_init is None else np.asarray(w_init).astype(np.float64)
    cols = []
    for w, n in zip(snapshots, ns):
        w_swa, dev = swa_dev_push(w_swa, w, n)
        cols.append(dev)
    a = np.stack(cols, axis=1) if cols else np.zeros((n_par, 0))
    return w_swa, np.asfortranarray(a)


def projection_from_A(a, m):
    """src/subspace_construction.jl:63,65:  U,s,V = psvd(A);  P = U[:,1:M]*Diagonal(s[1:M]).

    `psvd` is LowRankApprox 0.5.0 (Manifest.toml:786; NOT vendored).  Its published contract: a partial
    SVD accurate to rtol = 5*eps, singular values descending -- restated here as LAPACK's exact thin SVD.
    Column signs are arbitrary in both.  Raises like Julia's BoundsError when rank < M."""
    u, s, _ = np.linalg.svd(a, full_matrices=False)
    if m > len(s):
        raise IndexError("BoundsError: M=%d exceeds min(N,K)=%d" % (m, len(s)))
    return u[:, :m] * s[:m][None, :], s


# --------------------------------------------------------------------------- density
def _act(a, kind):
    if kind == ACT_IDENTITY:
        return a
    if kind == ACT_RELU:
        return np.maximum(a, 0.0)
    if kind == ACT_TANH:
        return np.tanh(a)
    if kind == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-a))
    if kind == ACT_LEAKYRELU:
        return np.maximum(0.01 * a, a)
    if kind == ACT_ELU:
        return np.where(a >= 0.0, a, (np.exp(np.minimum(a, 0.0)) - 1.0))
    if kind == ACT_SOFTPLUS:
        return np.maximum(a, 0.0) + np.log1p(np.exp(-np.abs(a)))
    if kind == ACT_SELU:
        return SELU_LAMBDA * np.where(a > 0.0, a, SELU_ALPHA * (np.exp(np.minimum(a, 0.0)) - 1.0))
    raise ValueError(kind)


# ---- Conv / MaxPool / flatten rows of a Chain (`model_re` restructures ANY Chain, src/libs.jl:55-57).
# A table row is a Dense 5-tuple (in, out, act, w_off, b_off) or one of
#   ("conv", (KW, KH, CIN, COUT), (Wi, Hi), (sw, sh), (pw, ph), (dw, dh), act, w_off, b_off)
#   ("maxpool", (PW, PH), C, (Wi, Hi), (sw, sh))
#   ("flatten", C, (Wi, Hi))
# Activations between such rows are matrices [features x B] whose feature index runs in Julia's WHC column-major order
# (w fastest), i.e. the (W, H, C, N) arrays of Flux reshaped to (W*H*C, N) -- which is exactly what `flatten` does.
def conv_out_size(wi, k, s, p, d):
    """NNlib 0.7.23 output_size: floor((in + 2 pad - dil*(k-1) - 1) / stride) + 1  [upstream, Manifest.toml:918]"""
    return (wi + 2 * p - d * (k - 1) - 1) // s + 1


def conv_forward(x4, w4, b, stride, pad, dil):
    """Flux 0.11.2 `Conv` forward before the activation: conv(x, weight) .+ reshape(b, 1, 1, :, 1) with NNlib's DEFAULT
    `flipkernel = false`, i.e. a TRUE convolution (kernel index reversed) [upstream NNlib 0.7.23 conv_direct!: x index
    (o-1)*stride - pad + 1 + (k-1)*dil, weight index K - k + 1].  x4 is (W, H, CIN, N), w4 is (KW, KH, CIN, COUT)."""
    kw, kh, cin, cout = w4.shape
    wi, hi, _, n = x4.shape
    wo, ho = conv_out_size(wi, kw, stride[0], pad[0], dil[0]), conv_out_size(hi, kh, stride[1], pad[1], dil[1])
    xp = np.zeros((wi + 2 * pad[0], hi + 2 * pad[1], cin, n), dtype=np.float64)
    xp[pad[0]:pad[0] + wi, pad[1]:pad[1] + hi] = x4
    y = np.zeros((wo, ho, cout, n), dtype=np.float64)
    for a in range(kw):
        for c in range(kh):
            xs = xp[a * dil[0]: a * dil[0] + (wo - 1) * stride[0] + 1: stride[0],
                    c * dil[1]: c * dil[1] + (ho - 1) * stride[1] + 1: stride[1]]
            y += np.einsum("whin,io->whon", xs, w4[kw - 1 - a, kh - 1 - c], optimize=True)
    return y + b[None, None, :, None]


def maxpool_forward(x4, win, stride):
    """Flux 0.11.2 MaxPool(k; pad = 0, stride = k) [upstream NNlib maxpool]."""
    wi, hi, c, n = x4.shape
    wo, ho = (wi - win[0]) // stride[0] + 1, (hi - win[1]) // stride[1] + 1
    y = np.full((wo, ho, c, n), -np.inf)
    for a in range(win[0]):
        for d in range(win[1]):
            y = np.maximum(y, x4[a: a + (wo - 1) * stride[0] + 1: stride[0], d: d + (ho - 1) * stride[1] + 1: stride[1]])
    return y


def _layer_forward(row, wflat, h):
    """one table row: [features x B] -> [features x B]; also returns what the reverse sweep needs"""
    bsz = h.shape[1]
    if row[0] == "conv":
        _, (kw, kh, cin, cout), (wi, hi), stride, pad, dil, act, w_off, b_off = row
        w4 = wflat[w_off:w_off + kw * kh * cin * cout].reshape((kw, kh, cin, cout), order="F")
        y = _act(conv_forward(h.reshape((wi, hi, cin, bsz), order="F"), w4, wflat[b_off:b_off + cout], stride, pad, dil), act)
        return y.reshape((-1, bsz), order="F")
    if row[0] == "maxpool":
        _, win, c, (wi, hi), stride = row
        return maxpool_forward(h.reshape((wi, hi, c, bsz), order="F"), win, stride).reshape((-1, bsz), order="F")
    if row[0] == "flatten":
        return h
    fin, fout, act, w_off, b_off = row
    w = wflat[w_off:w_off + fin * fout].reshape((fout, fin), order="F")
    return _act(w @ h + wflat[b_off:b_off + fout][:, None], act)


def forward(table, wflat, x):
    """src/libs.jl:55-57 (`re(W)`: reshape consecutive slices, column-major) followed by the Flux 0.11.2 forward of every
    layer (src/space_inference.jl:94 `new_model(in_data)`): Dense `sigma.(W*x .+ b)`, Conv, MaxPool, flatten."""
    h = x
    for row in table:
        h = _layer_forward(row, wflat, h)
    return h


def conv_table(spec, in_whc):
    """Layout of a Chain with Conv / MaxPool / flatten / Dense layers inside the flat weight vector, in `Flux.params`
    order (Conv: weight (KW, KH, CIN, COUT) column-major then bias; Dense: W then b) -- src/libs.jl:19-22, 55-57.
    spec entries: ("conv", (kw, kh), cout, act, stride, pad, dil) | ("maxpool", (pw, ph)[, stride]) | ("flatten",) |
    ("dense", out, act).  Returns (table, N)."""
    wi, hi, c = in_whc
    feat, off, table = None, 0, []
    for e in spec:
        if e[0] == "conv":
            (kw, kh), cout, act = e[1], e[2], e[3]
            stride = e[4] if len(e) > 4 else (1, 1)
            pad = e[5] if len(e) > 5 else (0, 0)
            dil = e[6] if len(e) > 6 else (1, 1)
            w_off = off
            off += kw * kh * c * cout
            b_off = off
            off += cout
            table.append(("conv", (kw, kh, c, cout), (wi, hi), tuple(stride), tuple(pad), tuple(dil), act, w_off, b_off))
            wi, hi, c = conv_out_size(wi, kw, stride[0], pad[0], dil[0]), conv_out_size(hi, kh, stride[1], pad[1], dil[1]), cout
        elif e[0] == "maxpool":
            win = e[1]
            stride = e[2] if len(e) > 2 else win
            table.append(("maxpool", tuple(win), c, (wi, hi), tuple(stride)))
            wi, hi = (wi - win[0]) // stride[0] + 1, (hi - win[1]) // stride[1] + 1
        elif e[0] == "flatten":
            table.append(("flatten", c, (wi, hi)))
            feat = wi * hi * c
        else:
            fin = feat if feat is not None else wi * hi * c
            fout, act = e[1], e[2]
            table.append((fin, fout, act, off, off + fin * fout))
            off += fin * fout + fout
            feat = fout
    return table, off


def mvnormal_logpdf_iso(y, mu, sigma):
    """Distributions 0.24.18 `logpdf(MvNormal(mu, sigma::Real), y)` (Manifest.toml:382; not vendored):
    isotropic, sigma is a STANDARD DEVIATION: -(d*log(2pi) + d*log(sigma^2))/2 - ||y-mu||^2/(2 sigma^2)."""
    d = y.size
    r = (y - mu).reshape(-1)
    sse = float(np.dot(r, r))
    return lp_from_sse(sse, d, sigma), sse


def lp_from_sse(sse, d, sigma):
    c0 = -(d * math.log(2.0 * math.pi) + d * math.log(sigma * sigma)) / 2.0
    return c0 - (sse / (sigma * sigma)) / 2.0


def log_prior(new_w, sigma_p):
    """the expression after the reference's `return` (src/space_inference.jl:95, dead code -- Q4):
    logpdf(MvNormal(zeros(length(new_W)), sigma_p), new_W); only used by the NON-default include_prior option"""
    lp, _ = mvnormal_logpdf_iso(new_w, np.zeros_like(new_w), sigma_p)
    return lp


def logdensity(table, w_swa, p, x, y, sigma_m, z):
    """src/space_inference.jl:90-95 `density(z)` for a Chain: new_W = W_swa + P*z; forward over the FULL
    data (Q5); Gaussian log-likelihood only -- the prior line after `return` is dead code (Q4)."""
    new_w = w_swa + p @ z
    yhat = forward(table, new_w, x)
    lp, _ = mvnormal_logpdf_iso(y.reshape(-1, order="F"), yhat.reshape(-1, order="F"), sigma_m)
    return lp


def _dact(h, kind):
    """derivative of the activation expressed through its OUTPUT h (relu' = [h > 0], tanh' = 1 - h^2, ...)"""
    if kind == ACT_IDENTITY:
        return np.ones_like(h)
    if kind == ACT_RELU:
        return (h > 0.0).astype(h.dtype)
    if kind == ACT_TANH:
        return 1.0 - h * h
    if kind == ACT_SIGMOID:
        return h * (1.0 - h)
    if kind == ACT_LEAKYRELU:
        return np.where(h > 0.0, 1.0, 0.01)
    if kind == ACT_ELU:
        return np.where(h >= 0.0, 1.0, h + 1.0)
    if kind == ACT_SOFTPLUS:
        return -np.expm1(-h)
    if kind == ACT_SELU:
        return np.where(h > 0.0, SELU_LAMBDA, h + SELU_LAMBDA * SELU_ALPHA)
    raise ValueError(kind)


def logdensity_grad(table, w_swa, p, x, y, sigma_m, z):
    """src/space_inference.jl:107  `l_pi_grad(theta) = (density(theta), gradient(density, theta))`.

    The reference differentiates `density` with ForwardDiff (M duals through the chain); the value is the gradient of
    the same scalar, restated here as one reverse sweep in NumPy.  Returns (lp, d lp / d z, d lp / d w)."""
    new_w = w_swa + p @ z
    hs = [x]
    for row in table:
        hs.append(_layer_forward(row, new_w, hs[-1]))
    lp, _ = mvnormal_logpdf_iso(y.reshape(-1, order="F"), hs[-1].reshape(-1, order="F"), sigma_m)
    gw = np.zeros_like(new_w)
    g = (y - hs[-1]) / (sigma_m * sigma_m)          # d lp / d (output of the last layer)
    for l in range(len(table) - 1, -1, -1):
        g = _layer_backward(table[l], new_w, hs[l], hs[l + 1], g, gw)
    return lp, p.T @ gw, gw


def _layer_backward(row, wflat, h_in, h_out, g_out, gw):
    """reverse sweep through one row: g_out = d lp / d h_out -> returns d lp / d h_in, adds the parameter gradient to gw"""
    bsz = h_in.shape[1]
    if row[0] == "flatten":
        return g_out
    if row[0] == "maxpool":
        _, win, c, (wi, hi), stride = row
        x4 = h_in.reshape((wi, hi, c, bsz), order="F")
        y4 = h_out.reshape(((wi - win[0]) // stride[0] + 1, (hi - win[1]) // stride[1] + 1, c, bsz), order="F")
        g4 = g_out.reshape(y4.shape, order="F")
        gx = np.zeros_like(x4)
        wo, ho = y4.shape[:2]
        # [upstream NNlib 0.7.23 src/impl/pooling_direct.jl, `∇maxpool_direct!`, from memory]: for each output element the
        # window is walked `for kd in 1:kernel_d, kh in 1:kernel_h, kw in 1:kernel_w` (kw FASTEST) and the gradient goes to
        # the FIRST input with `y_idx ≈ x[...] && !maxpool_already_chosen` -- one element per window, chosen by
        # `isapprox` (rtol = sqrt(eps), atol = 0), not every element equal to the maximum.
        chosen = np.zeros(y4.shape, dtype=bool)
        rtol = math.sqrt(np.finfo(np.float64).eps)
        for d in range(win[1]):           # kh (outer)
            for a in range(win[0]):       # kw (inner, fastest)
                sl = (slice(a, a + (wo - 1) * stride[0] + 1, stride[0]), slice(d, d + (ho - 1) * stride[1] + 1, stride[1]))
                xv = x4[sl]
                approx = (xv == y4) | (np.isfinite(xv) & np.isfinite(y4) & (np.abs(xv - y4) <= rtol * np.maximum(np.abs(xv), np.abs(y4))))
                hit = approx & ~chosen
                gx[sl] += g4 * hit
                chosen |= hit
        return gx.reshape((-1, bsz), order="F")
    if row[0] == "conv":
        _, (kw, kh, cin, cout), (wi, hi), stride, pad, dil, act, w_off, b_off = row
        wo, ho = conv_out_size(wi, kw, stride[0], pad[0], dil[0]), conv_out_size(hi, kh, stride[1], pad[1], dil[1])
        w4 = wflat[w_off:w_off + kw * kh * cin * cout].reshape((kw, kh, cin, cout), order="F")
        d4 = (g_out * _dact(h_out, act)).reshape((wo, ho, cout, bsz), order="F")
        x4 = h_in.reshape((wi, hi, cin, bsz), order="F")
        xp = np.zeros((wi + 2 * pad[0], hi + 2 * pad[1], cin, bsz))
        xp[pad[0]:pad[0] + wi, pad[1]:pad[1] + hi] = x4
        gxp = np.zeros_like(xp)
        gw4 = np.zeros_like(w4)
        for a in range(kw):
            for c in range(kh):
                sl = (slice(a * dil[0], a * dil[0] + (wo - 1) * stride[0] + 1, stride[0]),
                      slice(c * dil[1], c * dil[1] + (ho - 1) * stride[1] + 1, stride[1]))
                gw4[kw - 1 - a, kh - 1 - c] = np.einsum("whin,whon->io", xp[sl], d4, optimize=True)
                gxp[sl] += np.einsum("whon,io->whin", d4, w4[kw - 1 - a, kh - 1 - c], optimize=True)
        gw[w_off:w_off + w4.size] = gw4.reshape(-1, order="F")
        gw[b_off:b_off + cout] = d4.sum(axis=(0, 1, 3))
        return gxp[pad[0]:pad[0] + wi, pad[1]:pad[1] + hi].reshape((-1, bsz), order="F")
    fin, fout, act, w_off, b_off = row
    delta = g_out * _dact(h_out, act)
    gw[w_off:w_off + fin * fout] = (delta @ h_in.T).reshape(-1, order="F")
    gw[b_off:b_off + fout] = delta.sum(axis=1)
    w = wflat[w_off:w_off + fin * fout].reshape((fout, fin), order="F")
    return w.T @ delta


def reconstruct(w_swa, p, z):
    """src/space_inference.jl:91 and :125  `W_swa + P*z`."""
    return w_swa + p @ z


# --------------------------------------------------------------------------- training step
FLUX_EPS = 1e-8   # [upstream Flux 0.11.2 src/optimise/optimisers.jl: `const ϵ = 1e-8`]


def mse_value_and_grad(table, w64, x, y):
    """src/subspace_construction.jl:39-42 for `cost(m, x, y) = Flux.Losses.mse(m(x), y)`:
        gs = gradient(ps) do training_loss = cost(model, d...) end
    [upstream Flux 0.11.2 `mse(ŷ, y; agg = mean) = agg((ŷ .- y).^2)`; Zygote 0.5.17 reverse mode].  The Float32
    parameters meet Float64 data, so every product promotes to Float64 and the gradient comes back as Float64 arrays
    (Zygote 0.5 does not project it onto the parameter's eltype).  Returns (loss, flat gradient in extract_params order).

    With Float32 weights AND Float32 data (pass float32 arrays for all three) nothing promotes: the whole pass is Float32 --
    `W * x` is sgemm, the loss a Float32 mean, the gradient Float32 arrays (NumPy keeps float32 through every line below
    exactly as Julia does; the summation ORDER inside sgemm / mean is the BLAS's and the runtime's, not pinned)."""
    hs = [x]
    for row in table:
        hs.append(_layer_forward(row, w64, hs[-1]))
    diff = hs[-1] - y
    loss = float(np.mean(diff * diff))
    gw = np.zeros_like(w64)
    g = (2.0 / diff.size) * diff
    for l in range(len(table) - 1, -1, -1):
        g = _layer_backward(table[l], w64, hs[l], hs[l + 1], g, gw)
    return loss, gw


def optimiser_state(n, opt):
    """`zero(x)` state of a fresh optimiser, flat over the parameter vector: Float32 like the parameters
    [upstream Flux 0.11.2: Momentum `get!(o.velocity, x, zero(x))`; ADAM `get!(o.state, x, (zero(x), zero(x), β))`]."""
    kind = opt[0]
    st = {"m": np.zeros(n, dtype=np.float32), "v": np.zeros(n, dtype=np.float32), "bp": None}
    if kind == "adam":
        st["bp"] = [float(opt[2]), float(opt[3])]
    return st


def apply_update(w32, state, g64, opt):
    """src/subspace_construction.jl:43 `Flux.update!(opt, ps, gs)` [upstream Flux 0.11.2 src/optimise/{train,optimisers}.jl]:

        update!(opt, x, x̄) = (x .-= apply!(opt, x, x̄))                 # per parameter array; flat here (elementwise rules)
        apply!(o::Descent, x, Δ)   = (Δ .*= o.eta)
        apply!(o::Momentum, x, Δ)  = (v = get!(o.velocity, x, zero(x)); @. v = ρ * v - η * Δ; @. Δ = -v)
        apply!(o::ADAM, x, Δ)      = (mt, vt, βp = get!(o.state, x, (zero(x), zero(x), β));
                                      @. mt = β[1] * mt + (1 - β[1]) * Δ;  @. vt = β[2] * vt + (1 - β[2]) * Δ^2;
                                      @. Δ = mt / (1 - βp[1]) / (√(vt / (1 - βp[2])) + ϵ) * η;  o.state[x] = (mt, vt, βp .* β))

    η, ρ, β are Float64 fields; x and the state are Float32 arrays, Δ is Float64: every broadcast computes in Float64 and
    ROUNDS ONCE on the store into the Float32 array (v, mt, vt, and x itself); ADAM's step reads the stored (rounded) mt,
    vt.  No fused multiply-add (Julia does not contract `a * b + c`).  opt = ("descent", η) | ("momentum", η, ρ) |
    ("adam", η, β1, β2).  Updates w32 / state in place."""
    kind = opt[0]
    eta = np.float64(opt[1])
    # A Float32 gradient (the all-Float32 pass) changes two roundings: `apply!` writes its step back INTO the Float32 array Δ
    # (`Δ .*= η`, `@. Δ = -v`, `@. Δ = mt / ...`: computed in Float64, rounded on the store) and `x .-= Δ` subtracts two
    # Float32 arrays in Float32.  With a Float64 gradient Δ stays Float64 and x - Δ is rounded once on the store into x.
    g32 = g64.dtype == np.float32
    g64 = g64.astype(np.float64)
    if kind == "descent":
        step = g64 * eta
    elif kind == "momentum":
        rho = np.float64(opt[2])
        state["m"][...] = (rho * state["m"].astype(np.float64) - eta * g64).astype(np.float32)
        step = -state["m"].astype(np.float64)
    elif kind == "adam":
        b1, b2 = np.float64(opt[2]), np.float64(opt[3])
        bp = state["bp"]
        state["m"][...] = (b1 * state["m"].astype(np.float64) + (1.0 - b1) * g64).astype(np.float32)
        state["v"][...] = (b2 * state["v"].astype(np.float64) + (1.0 - b2) * (g64 * g64)).astype(np.float32)
        step = state["m"].astype(np.float64) / (1.0 - bp[0]) / (np.sqrt(state["v"].astype(np.float64) / (1.0 - bp[1])) + FLUX_EPS) * eta
        state["bp"] = [bp[0] * float(b1), bp[1] * float(b2)]
    else:
        raise ValueError(kind)
    if g32:
        w32[...] = w32 - step.astype(np.float32)
    else:
        w32[...] = (w32.astype(np.float64) - step).astype(np.float32)


def train_step(table, w32, state, x, y, opt):
    """One pass of the reference's loop body, src/subspace_construction.jl:39-43: Zygote gradient of the mse cost on the
    batch (x, y), then Flux.update!.  Returns the loss BEFORE the update (Zygote's forward value, `training_loss`).
    The arithmetic follows the DATA: Float64 (x, y) promote the Float32 weights (a Float64 pass, a Float64 gradient);
    Float32 (x, y) give the all-Float32 pass."""
    if x.dtype == np.float32 and y.dtype == np.float32:
        loss, g = mse_value_and_grad(table, w32, x, y)
    else:
        loss, g = mse_value_and_grad(table, w32.astype(np.float64), x.astype(np.float64), y.astype(np.float64))
    apply_update(w32, state, g, opt)
    return float(loss)


# --------------------------------------------------------------------------- sampler
def rwmh(density, m, itr, sigma_z, seed, chain=0):
    """src/space_inference.jl:111-116 -- DensityModel + RWMH(MvNormal(zeros(M), sigma_z)) + sample(.., itr).

    AdvancedMH 0.6.2 / AbstractMCMC 3.2.1 semantics (Manifest.toml:38,9; not vendored):
      step 1 : z0 ~ proposal (no init_params), lp0 = density(z0)                -> sample 1
      step t : z' = z + rand(proposal); accept iff -randexp() < lp' - lp        -> sample t
    `itr` samples INCLUDING the initial draw; one density call per sample.  The random stream is the
    build's Philox stream (oracle/philox.py), not Julia's MersenneTwister.  Returns (Z (M x itr), lp, n_accept)."""
    zs = np.empty((m, itr), dtype=np.float64, order="F")
    lps = np.empty(itr, dtype=np.float64)
    z = sigma_z * philox.normals(seed, chain, 0, m)
    lp = density(z)
    zs[:, 0], lps[0] = z, lp
    nacc = 0
    for t in range(1, itr):
        zp = z + sigma_z * philox.normals(seed, chain, t, m)
        lpp = density(zp)
        if -philox.randexp(seed, chain, t) < lpp - lp:
            z, lp = zp, lpp
            nacc += 1
        zs[:, t], lps[t] = z, lp
    return zs, lps, nacc


def sub_inference(table, x, y, w_swa, p, sigma_z, sigma_m, itr, seed, chain=0):
    """src/space_inference.jl:82-125 for `alg = :rwmh`, Chain model.  Returns (Z, lp, weights) where
    weights[:, t] = W_swa + P*Z[:, t] (the reference's `map(z -> W_swa + P*z.params, chm)`, :125)."""
    m = p.shape[1]
    dens = lambda z: logdensity(table, w_swa, p, x, y, sigma_m, z)
    zs, lps, nacc = rwmh(dens, m, itr, sigma_z, seed, chain)
    return zs, lps, w_swa[:, None] + p @ zs, nacc
