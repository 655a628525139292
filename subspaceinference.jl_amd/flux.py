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

from ._capi import ACT_IDENTITY, ACT_RELU, ACT_SIGMOID, ACT_TANH, SubspaceError

identity, relu, tanh, sigmoid = ACT_IDENTITY, ACT_RELU, ACT_TANH, ACT_SIGMOID
_ACT_NAMES = {"identity": identity, "relu": relu, "tanh": tanh, "sigmoid": sigmoid, "σ": sigmoid}


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
    return 1.0 / (1.0 + np.exp(-a))


def _act_bwd(pre, post, kind):
    if kind == identity:
        return np.ones_like(post)
    if kind == relu:
        return (pre > 0).astype(post.dtype)
    if kind == tanh:
        return 1.0 - post * post
    return post * (1.0 - post)


class Dense:
    def __init__(self, fin, fout, act=identity, rng=None, dtype=np.float32):
        rng = rng if rng is not None else np.random.default_rng()
        scale = np.sqrt(24.0 / (fin + fout))
        self.W = ((rng.random((fout, fin)) - 0.5) * scale).astype(dtype)
        self.b = np.zeros(fout, dtype=dtype)
        self.act = _act_id(act)

    def __call__(self, x):
        return _act_fwd(self.W @ x + self.b[:, None], self.act)


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
        if not isinstance(l, Dense):
            raise SubspaceError("Error: model_re function is not available for this model (only Dense layers)")
        out += [l.W, l.b]
    return out


def extract_params(ps):
    """src/libs.jl:19-22: mapreduce(vec, vcat, ...) -- column-major vec of every array, concatenated."""
    return np.concatenate([p.reshape(-1, order="F") for p in ps])


def layer_table(model):
    """Static layer-offset table that replaces the per-call Flux.destructure/re of src/libs.jl:55-57."""
    table, off = [], 0
    for l in model.layers:
        fout, fin = l.W.shape
        table.append((fin, fout, l.act, off, off + fin * fout))
        off += fin * fout + fout
    return table, off


def load_flat(model, w):
    """`re(W)`: write a flat vector back into the model's arrays (column-major slices)."""
    for (fin, fout, _, w_off, b_off), l in zip(layer_table(model)[0], model.layers):
        l.W[...] = w[w_off:w_off + fin * fout].reshape((fout, fin), order="F")
        l.b[...] = w[b_off:b_off + fout]


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
    for (fin, fout, _, w_off, b_off), l in zip(layer_table(model)[0], model.layers):
        for arr, off, shape in ((l.W, w_off, (fout, fin)), (l.b, b_off, (fout,))):
            size = int(np.prod(shape))
            mv = m_flat[off:off + size].reshape(shape, order="F").astype(arr.dtype)
            if isinstance(opt, Momentum):
                opt.v[id(arr)] = mv
            else:
                vv = v_flat[off:off + size].reshape(shape, order="F").astype(arr.dtype)
                opt.state[id(arr)] = [mv, vv, [beta_pows[0], beta_pows[1]]]


def update(opt, ps, gs):
    """Flux.update!(opt, ps, gs): `x .-= apply!(opt, x, g)` -- Float32 x minus Float64 step, rounded once to Float32."""
    for p, g in zip(ps, gs):
        p[...] = (p.astype(np.float64) - opt.apply(p, np.asarray(g, dtype=np.float64))).astype(p.dtype)
