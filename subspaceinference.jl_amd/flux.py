"""Caller-side stand-ins for the Flux 0.11.2 objects the reference's API takes as ARGUMENTS:
`Chain(Dense...)`, `DataLoader`, the optimisers and `Flux.Losses.mse` + its gradient.

This is NOT part of the accelerated hot path.  In the reference these objects live in the user's
Julia session (README.md:52-79) and the training step `gradient(ps) do cost(model, d...) end;
Flux.update!(opt, ps, gs)` (src/subspace_construction.jl:39-43) runs on the host in Zygote; SURVEY.md
section 8(f1) lists an on-device training step as the NEXT row.  Julia is not available in this image, so
the Python host mirror needs something that produces the same stream of weight snapshots; everything
downstream of `extract_params` (src/libs.jl:19-22) goes to the GPU through the C ABI.

Semantics restated from Flux 0.11.2 (Manifest.toml:491; not vendored):
  Dense(in, out, sigma=identity): W (out x in) glorot_uniform Float32, b zeros Float32, sigma.(W*x .+ b)
  DataLoader(X, Y; batchsize=1, shuffle=false, partial=true): batches along the LAST dimension; `.data = (X, Y)`
  Descent(eta=0.1); Momentum(eta=0.01, rho=0.9); ADAM(eta=0.001, beta=(0.9, 0.999)), eps = 1e-8
  update!(opt, x, g): x .-= apply!(opt, x, g)
  Flux.Losses.mse(yhat, y) = mean((yhat .- y).^2)
"""
import numpy as np

from ._capi import (ACT_ELU, ACT_IDENTITY, ACT_LEAKYRELU, ACT_RELU, ACT_SELU, ACT_SIGMOID, ACT_SOFTPLUS, ACT_TANH,
                    SubspaceError)

identity, relu, tanh, sigmoid = ACT_IDENTITY, ACT_RELU, ACT_TANH, ACT_SIGMOID
leakyrelu, elu, softplus, selu = ACT_LEAKYRELU, ACT_ELU, ACT_SOFTPLUS, ACT_SELU   # Flux 0.11.2 defaults: a = 0.01, alpha = 1
_SELU_L, _SELU_A = 1.0507009873554805, 1.6732632423543772
_ACT_NAMES = {"identity": identity, "relu": relu, "tanh": tanh, "sigmoid": sigmoid, "σ": sigmoid, "leakyrelu": leakyrelu,
              "elu": elu, "softplus": softplus, "selu": selu}


def _act_id(a):
    if isinstance(a, str):
        return _ACT_NAMES[a]
    return int(a)


def _act_fwd(a, kind):
    if kind == identity:
        return a
    if kind == relu:
        return np.maximum(a, 0)
    if kind == tanh:
        return np.tanh(a)
    if kind == sigmoid:
        return 1.0 / (1.0 + np.exp(-a))
    if kind == leakyrelu:
        return np.maximum(0.01 * a, a)
    if kind == elu:
        return np.where(a >= 0, a, (np.exp(np.minimum(a, 0)) - 1.0))
    if kind == softplus:
        return np.maximum(a, 0) + np.log1p(np.exp(-np.abs(a)))
    if kind == selu:
        return _SELU_L * np.where(a > 0, a, _SELU_A * (np.exp(np.minimum(a, 0)) - 1.0))
    raise SubspaceError("Error: activation %r is not available" % (kind,))


def _act_bwd(pre, post, kind):
    if kind == identity:
        return np.ones_like(post)
    if kind == relu:
        return (pre > 0).astype(post.dtype)
    if kind == tanh:
        return 1.0 - post * post
    if kind == sigmoid:
        return post * (1.0 - post)
    if kind == leakyrelu:
        return np.where(pre > 0, 1.0, 0.01).astype(post.dtype)
    if kind == elu:
        return np.where(pre >= 0, 1.0, post + 1.0).astype(post.dtype)
    if kind == softplus:
        return (1.0 / (1.0 + np.exp(-pre))).astype(post.dtype)
    if kind == selu:
        return np.where(pre > 0, _SELU_L, post + _SELU_L * _SELU_A).astype(post.dtype)
    raise SubspaceError("Error: activation %r is not available" % (kind,))


class Dense:
    def __init__(self, fin, fout, act=identity, rng=None, dtype=np.float32):
        rng = rng if rng is not None else np.random.default_rng()
        scale = np.sqrt(24.0 / (fin + fout))
        self.W = ((rng.random((fout, fin)) - 0.5) * scale).astype(dtype)
        self.b = np.zeros(fout, dtype=dtype)
        self.act = _act_id(act)

    def __call__(self, x):
        return _act_fwd(self.W @ x + self.b[:, None], self.act)


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


class Conv:
    """Flux 0.11.2 `Conv((kw, kh), cin => cout, σ = identity; stride = 1, pad = 0, dilation = 1)`: weight
    (kw, kh, cin, cout) glorot_uniform Float32, bias zeros(cout); forward `σ.(conv(x, weight) .+ b)` on (W, H, C, N)
    arrays with NNlib's default TRUE convolution (flipped kernel) [upstream NNlib 0.7.23]."""

    def __init__(self, k, cin, cout, act=identity, stride=1, pad=0, dilation=1, rng=None, dtype=np.float32):
        rng = rng if rng is not None else np.random.default_rng()
        self.k, self.stride, self.pad, self.dilation = _pair(k), _pair(stride), _pair(pad), _pair(dilation)
        kw, kh = self.k
        scale = np.sqrt(24.0 / (kw * kh * (cin + cout)))        # glorot_uniform: fan_in + fan_out over the receptive field
        self.weight = ((rng.random((kw, kh, cin, cout)) - 0.5) * scale).astype(dtype)
        self.bias = np.zeros(cout, dtype=dtype)
        self.act = _act_id(act)

    def out_size(self, wi, hi):
        (kw, kh), (sw, sh), (pw, ph), (dw, dh) = self.k, self.stride, self.pad, self.dilation
        return (wi + 2 * pw - dw * (kw - 1) - 1) // sw + 1, (hi + 2 * ph - dh * (kh - 1) - 1) // sh + 1

    def __call__(self, x):
        kw, kh, cin, cout = self.weight.shape
        wi, hi, _, n = x.shape
        (sw, sh), (pw, ph), (dw, dh) = self.stride, self.pad, self.dilation
        wo, ho = self.out_size(wi, hi)
        xp = np.zeros((wi + 2 * pw, hi + 2 * ph, cin, n), dtype=np.result_type(x, self.weight))
        xp[pw:pw + wi, ph:ph + hi] = x
        y = np.zeros((wo, ho, cout, n), dtype=xp.dtype)
        for a in range(kw):
            for c in range(kh):
                xs = xp[a * dw: a * dw + (wo - 1) * sw + 1: sw, c * dh: c * dh + (ho - 1) * sh + 1: sh]
                y += np.einsum("whin,io->whon", xs, self.weight[kw - 1 - a, kh - 1 - c])
        return _act_fwd(y + self.bias[None, None, :, None], self.act)


class MaxPool:
    """Flux 0.11.2 `MaxPool(k; pad = 0, stride = k)` on (W, H, C, N) arrays."""

    def __init__(self, k, stride=None):
        self.k = _pair(k)
        self.stride = self.k if stride is None else _pair(stride)

    def out_size(self, wi, hi):
        return (wi - self.k[0]) // self.stride[0] + 1, (hi - self.k[1]) // self.stride[1] + 1

    def __call__(self, x):
        wo, ho = self.out_size(x.shape[0], x.shape[1])
        y = np.full((wo, ho) + x.shape[2:], -np.inf, dtype=x.dtype)
        for a in range(self.k[0]):
            for d in range(self.k[1]):
                y = np.maximum(y, x[a: a + (wo - 1) * self.stride[0] + 1: self.stride[0],
                                    d: d + (ho - 1) * self.stride[1] + 1: self.stride[1]])
        return y


class Flatten:
    """Flux.flatten: (W, H, C, N) -> (W*H*C, N), column-major."""

    def __call__(self, x):
        return x.reshape((-1, x.shape[-1]), order="F")


flatten = Flatten()


class Chain:
    def __init__(self, *layers):
        self.layers = list(layers)

    def __call__(self, x):
        for l in self.layers:
            x = l(x)
        return x


def params(model):
    """Flux.params(model): ordered list of the trainable arrays (W then b per Dense layer)."""
    if not isinstance(model, Chain):
        raise SubspaceError("Error: model_re function is not available for this model")
    out = []
    for l in model.layers:
        if isinstance(l, Dense):
            out += [l.W, l.b]
        elif isinstance(l, Conv):
            out += [l.weight, l.bias]
        elif not isinstance(l, (MaxPool, Flatten)):
            raise SubspaceError("Error: model_re function is not available for this model")
    return out


def extract_params(ps):
    """src/libs.jl:19-22: mapreduce(vec, vcat, ...) -- column-major vec of every array, concatenated."""
    return np.concatenate([p.reshape(-1, order="F") for p in ps])


def has_conv(model):
    return any(isinstance(l, (Conv, MaxPool, Flatten)) for l in model.layers)


def layer_table(model, input_size=None):
    """Static layer-offset table that replaces the per-call Flux.destructure/re of src/libs.jl:55-57.  Dense rows are
    (in, out, act, w_off, b_off); a Chain with Conv / MaxPool / flatten layers needs `input_size` = (W, H, C) of one
    observation and yields the tagged rows _capi._layer_array documents."""
    table, off = [], 0
    whc = tuple(int(v) for v in input_size) if input_size is not None else None
    for l in model.layers:
        if isinstance(l, Dense):
            fout, fin = l.W.shape
            table.append((fin, fout, l.act, off, off + fin * fout))
            off += fin * fout + fout
            whc = None
            continue
        if whc is None:
            raise SubspaceError("DimensionMismatch: Conv / MaxPool / flatten need (W, H, C, N) data: pass a 4-D X "
                                "(or input_size) and keep them in front of the Dense layers")
        wi, hi, c = whc
        if isinstance(l, Conv):
            kw, kh, cin, cout = l.weight.shape
            if cin != c:
                raise SubspaceError("DimensionMismatch: Conv expects %d input channels, got %d" % (cin, c))
            table.append(("conv", (kw, kh, cin, cout), (wi, hi), l.stride, l.pad, l.dilation, l.act, off, off + l.weight.size))
            off += l.weight.size + cout
            whc = l.out_size(wi, hi) + (cout,)
        elif isinstance(l, MaxPool):
            table.append(("maxpool", l.k, c, (wi, hi), l.stride))
            whc = l.out_size(wi, hi) + (c,)
        elif isinstance(l, Flatten):
            table.append(("flatten", c, (wi, hi)))
            whc = None
        else:
            raise SubspaceError("Error: model_re function is not available for this model")
    return table, off


def load_flat(model, w):
    """`re(W)`: write a flat vector back into the model's arrays (column-major slices)."""
    off = 0
    for p in params(model):
        p[...] = w[off:off + p.size].reshape(p.shape, order="F")
        off += p.size


def data_matrices(data):
    """split_data (src/libs.jl:75-77) for the device: X as the (features x B) matrix the C ABI takes -- a (W, H, C, N)
    array of images is the same memory in Julia's column-major order -- plus the (W, H, C) of one observation."""
    x, y = np.asarray(data.data[0]), np.asarray(data.data[1])
    size = tuple(x.shape[:3]) if x.ndim == 4 else None
    if x.ndim > 2:
        x = x.reshape((-1, x.shape[-1]), order="F")
    if y.ndim > 2:
        y = y.reshape((-1, y.shape[-1]), order="F")
    return x, y, size


class DataLoader:
    def __init__(self, *data, batchsize=1, shuffle=False, partial=True, rng=None):
        if len(data) == 1 and isinstance(data[0], (tuple, list)):
            data = tuple(data[0])
        self.data = tuple(np.asarray(d) for d in data)
        n = self.data[0].shape[-1]
        if any(d.shape[-1] != n for d in self.data):
            raise SubspaceError("DimensionMismatch: all data should contain same number of observations")
        self.nobs, self.batchsize, self.shuffle, self.partial = n, int(batchsize), shuffle, partial
        self.rng = rng if rng is not None else np.random.default_rng()

    def __len__(self):
        n = self.nobs / self.batchsize
        return int(np.ceil(n)) if self.partial else int(np.floor(n))

    def index_batches(self):
        """The observation indices of every batch of one epoch (one permutation per epoch when shuffle=true)."""
        idx = self.rng.permutation(self.nobs) if self.shuffle else np.arange(self.nobs)
        imax = self.nobs if self.partial else self.nobs - self.batchsize + 1
        for i in range(0, imax, self.batchsize):
            yield idx[i:i + self.batchsize]

    def __iter__(self):
        for ids in self.index_batches():
            yield tuple(d[..., ids] for d in self.data)


# ---------------------------------------------------------------------------- losses + gradients
class MSE:
    """`L(m, x, y) = Flux.Losses.mse(m(x), y)` together with its reverse-mode gradient (what Zygote supplies)."""

    def __call__(self, model, x, y):
        return float(np.mean((model(x) - y) ** 2))

    def value_and_grad(self, model, x, y):
        if has_conv(model):
            raise SubspaceError("there is no AD on the Python host for Conv / MaxPool layers (Zygote's job in the "
                                "reference): the training step of such a Chain runs on the device (device_training)")
        acts, pres = [x], []
        for l in model.layers:
            pre = l.W @ acts[-1] + l.b[:, None]
            pres.append(pre)
            acts.append(_act_fwd(pre, l.act))
        diff = acts[-1] - y
        loss = float(np.mean(diff ** 2))
        delta = (2.0 / diff.size) * diff
        grads = []
        for i in range(len(model.layers) - 1, -1, -1):
            l = model.layers[i]
            delta = delta * _act_bwd(pres[i], acts[i + 1], l.act)
            grads = [delta @ acts[i].T, delta.sum(axis=1)] + grads
            delta = l.W.T @ delta
        return loss, grads


mse = MSE()


def gradient(cost, model, *batch):
    if hasattr(cost, "value_and_grad"):
        return cost.value_and_grad(model, *batch)
    raise SubspaceError("cost must provide value_and_grad(model, x, y) (there is no AD on the Python host); "
                        "use subspaceinference_jl_amd.flux.mse")


# ---------------------------------------------------------------------------- optimisers
class Descent:
    def __init__(self, eta=0.1):
        self.eta = eta

    def apply(self, x, g):
        return self.eta * g


class Momentum:
    def __init__(self, eta=0.01, rho=0.9):
        self.eta, self.rho, self.v = eta, rho, {}

    def apply(self, x, g):
        v = self.v.setdefault(id(x), np.zeros_like(x))  # zero(x): Float32 state, Float64 arithmetic (Flux 0.11.2)
        v[...] = self.rho * v.astype(np.float64) - self.eta * g  # Float64 arithmetic (ρ::Float64), stored as Float32
        return -v.astype(np.float64)


class ADAM:
    def __init__(self, eta=0.001, beta=(0.9, 0.999)):
        self.eta, self.beta, self.state = eta, beta, {}

    def apply(self, x, g):
        st = self.state.setdefault(id(x), [np.zeros_like(x), np.zeros_like(x), list(self.beta)])  # (zero(x), zero(x), β)
        mt, vt, bp = st
        b1, b2 = self.beta
        mt[...] = b1 * mt.astype(np.float64) + (1 - b1) * g
        vt[...] = b2 * vt.astype(np.float64) + (1 - b2) * g * g
        d = mt.astype(np.float64) / (1 - bp[0]) / (np.sqrt(vt.astype(np.float64) / (1 - bp[1])) + 1e-8) * self.eta
        bp[0] *= b1
        bp[1] *= b2
        return d


def device_optimiser(opt):
    """(kind, eta, p1, p2) for si_train_setup, or None when the optimiser cannot run on the device (unknown type, or
    it already carries state from earlier host steps)."""
    if isinstance(opt, Descent):
        return 0, opt.eta, 0.0, 0.0
    if isinstance(opt, Momentum) and not opt.v:
        return 1, opt.eta, opt.rho, 0.0
    if isinstance(opt, ADAM) and not opt.state:
        return 2, opt.eta, opt.beta[0], opt.beta[1]
    return None


def store_device_state(opt, model, m_flat, v_flat, beta_pows):
    """Leave `opt` as Flux.update! would have left it after the steps the device took for it (src/subspace_construction.jl:43
    keeps the optimiser's IdDict state across calls): Momentum velocity / ADAM moments per parameter array, ADAM's running
    beta powers.  A later call with this optimiser continues on the host (device_optimiser() declines a used one)."""
    if isinstance(opt, Descent):
        return
    off = 0
    for arr in params(model):
        mv = m_flat[off:off + arr.size].reshape(arr.shape, order="F").astype(arr.dtype)
        if isinstance(opt, Momentum):
            opt.v[id(arr)] = mv
        else:
            vv = v_flat[off:off + arr.size].reshape(arr.shape, order="F").astype(arr.dtype)
            opt.state[id(arr)] = [mv, vv, [beta_pows[0], beta_pows[1]]]
        off += arr.size


def update(opt, ps, gs):
    """Flux.update!(opt, ps, gs): `x .-= apply!(opt, x, g)`.  With a Float64 gradient (Float64 data) the step stays Float64 and
    x - step is rounded once on the store into the Float32 x; with a Float32 gradient (the all-Float32 pass of Float32 data)
    `apply!` writes the step back into the Float32 gradient array (rounded) and the subtraction is Float32."""
    for p, g in zip(ps, gs):
        g = np.asarray(g)
        step = opt.apply(p, g.astype(np.float64))
        if g.dtype == np.float32 and p.dtype == np.float32:
            p[...] = p - step.astype(np.float32)
        else:
            p[...] = (p.astype(np.float64) - step).astype(p.dtype)
