"""The drop-in call itself on random problems: subspace_construction(model, mse, DataLoader(batchsize), opt; T, c, M) with the
training step on the device against the same call with the host stand-in step (ragged last batches, shuffling, the three
optimisers), then sub_inference on the result.  For the guard-page development library.
usage: SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin python3 tools/guard_fuzz_e2e.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from subspaceinference_jl_amd import flux  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):
    si._capi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libsubspace_hip_dev.so")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 25
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ACTS = [flux.identity, flux.relu, flux.tanh, flux.sigmoid]
ctx = si.Context(0)
for case in range(cases):
    nl = int(rng.integers(1, 4))
    dims = [int(rng.choice([1, 3, 10, 17]))] + [int(rng.choice([4, 20, 33, 96])) for _ in range(nl - 1)] + [int(rng.choice([1, 2, 5]))]
    acts = [ACTS[int(rng.integers(0, 4))] for _ in range(nl)]
    btot = int(rng.choice([7, 50, 100, 333]))
    bs = int(rng.choice([1, 7, 32, 100, btot])) if btot > 7 else int(rng.choice([1, 3, 7]))
    bs = min(bs, btot)
    t, c = int(rng.integers(2, 5)), int(rng.integers(1, 3))
    optk = int(rng.integers(0, 3))
    shuffle = bool(rng.random() < 0.5)
    npush = sum(1 for e in range(1, t + 1) if e % c == 0) * ((btot + bs - 1) // bs)
    m = int(rng.integers(1, min(4, max(1, npush)) + 1))
    print("case %d dims %s B %d batchsize %d T %d c %d opt %d shuffle %s M %d (pushes %d)" % (case, dims, btot, bs, t, c, optk, shuffle, m, npush), flush=True)
    x, y = rng.random((dims[0], btot)), rng.random((dims[-1], btot))

    def run(on_device):
        wr = np.random.default_rng(1000 + case)
        model = flux.Chain(*[flux.Dense(dims[i], dims[i + 1], acts[i], rng=wr) for i in range(nl)])
        opt = [flux.Descent(0.05), flux.Momentum(0.05, 0.9), flux.ADAM(0.01)][optk]
        data = flux.DataLoader(x, y, batchsize=bs, shuffle=shuffle, rng=np.random.default_rng(7 + case))
        w, p = si.subspace_construction(model, flux.mse, data, opt, T=t, c=c, M=m, ctx=ctx, verbose=False, device_training=on_device)
        return model, data, w, p
    try:
        md, dd, wd, pd = run(True)
    except si.BoundsError:
        try:
            run(False)
            raise AssertionError("the host-stepped run must hit the same BoundsError")
        except si.BoundsError:
            continue
    mh, dh, wh, ph = run(False)
    assert np.allclose(wd, wh, rtol=2e-3, atol=2e-5), np.abs(wd - wh).max()
    # posterior sampling on the device-trained subspace, through the drop-in call (output map included)
    chn, lp = si.sub_inference(md, dd, wd, pd, sigma_z=0.1, sigma_m=1.0, itr=5, M=m, ctx=ctx, seed=case)
    assert len(chn) == 5 and np.all(np.isfinite(lp)) and np.allclose(chn[-1], chn[-1])
print("guard_fuzz_e2e: %d cases done" % cases, flush=True)
ctx.close()
