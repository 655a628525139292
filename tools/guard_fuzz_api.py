"""Random Dense models through the rest of the C ABI: posterior predictive on new inputs, the optional prior term, the
step-wise RWMH with the observations split over two contexts (the data-sharded flow with the host as the collective),
the output-map pipeline of si_reconstruct with many samples, the device hand-over set-up (copied and borrowed), and the
gradient samplers; values against the oracle / the single-context entry points.  For the guard-page development library.
usage: SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin python3 tools/guard_fuzz_api.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from oracle import subspace_oracle as so  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):
    si._capi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libsubspace_hip_dev.so")
from subspaceinference_jl_amd import samplers  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
WIDTHS = [1, 3, 16, 17, 33, 64, 96, 97, 130, 192, 200]
a, b2, c3 = si.Context(0), si.Context(0), si.Context(0)
for case in range(cases):
    nl = int(rng.integers(1, 4))
    dims = [int(rng.choice([1, 2, 7, 12, 16, 33]))] + [int(rng.choice(WIDTHS)) for _ in range(nl - 1)] + [int(rng.choice([1, 2, 3, 5, 8]))]
    acts = [int(rng.integers(0, 8)) for _ in range(nl)]
    b = int(rng.choice([2, 7, 64, 129, 300, 1000, 2049]))
    m = int(rng.integers(1, 9))
    print("case %d dims %s acts %s B %d M %d" % (case, dims, acts, b, m), flush=True)
    table, n = so.layer_table(dims, acts)
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.1 * rng.standard_normal((n, m)))
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)))
    a.infer_setup(table, n, m, w_swa, p, x, y, 0.8)
    # 1. posterior predictive on new inputs, several chains
    c = int(rng.integers(1, 6))
    bn = int(rng.choice([1, 5, 130, 700]))
    zs = np.asfortranarray(0.4 * rng.standard_normal((m, c)))
    xn = np.asfortranarray(rng.standard_normal((dims[0], bn)))
    yh = a.predict(zs, xn)
    for j in range(c):
        yr = so.forward(table, so.reconstruct(w_swa, p, zs[:, j]), xn)
        assert np.allclose(yh[:, :, j], yr, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(yr).max()))
    # 2. the optional prior term
    sp = float(rng.choice([0.5, 2.0]))
    a.set_prior(sp)
    lp = a.logdensity(zs)
    ref = [so.logdensity(table, w_swa, p, x, y, 0.8, zs[:, j]) + so.log_prior(so.reconstruct(w_swa, p, zs[:, j]), sp) for j in range(c)]
    assert np.allclose(lp, ref, rtol=1e-9)
    lpg, g = a.logdensity_grad(zs[:, 0])
    assert np.isclose(lpg, ref[0], rtol=1e-9)
    a.set_prior(0.0)
    # 3. step-wise RWMH, observations split over two contexts, the host adds the partial sums
    if b >= 2:
        cut = int(rng.integers(1, b))
        b2.infer_setup(table, n, m, w_swa, p, np.asfortranarray(x[:, :cut]), np.asfortranarray(y[:, :cut]), 0.8)
        c3.infer_setup(table, n, m, w_swa, p, np.asfortranarray(x[:, cut:]), np.asfortranarray(y[:, cut:]), 0.8)
        itr, nch = 6, int(rng.integers(1, 3))
        for cc in (b2, c3):
            cc.rwmh_begin(itr, 0.1, case, 0, nch, dims[-1] * b)
        for _ in range(itr):
            tot = b2.rwmh_step_eval() + c3.rwmh_step_eval()
            b2.rwmh_step_accept(tot)
            c3.rwmh_step_accept(tot)
        z_s, lp_s, _ = b2.rwmh_end()
        z_t, lp_t, _ = c3.rwmh_end()
        z_1, lp_1, _ = a.sample_rwmh(itr, 0.1, seed=case, nchains=nch)
        assert np.array_equal(z_s, z_t) and np.array_equal(lp_s, lp_t)
        assert np.allclose(lp_s, lp_1, rtol=1e-10) and np.allclose(z_s, z_1, rtol=1e-12, atol=1e-14)
    # 4. the output-map pipeline with many samples
    cbig = int(rng.choice([1, 3, 33, 150]))
    zb = np.asfortranarray(rng.standard_normal((m, cbig)))
    assert np.allclose(a.reconstruct(zb), w_swa[:, None] + p @ zb, rtol=1e-13, atol=1e-14)
    # 5. device hand-over: the same set-up from device tensors (copied, then borrowed with a padded pitch)
    ld = n + (n & 1) + 2 * int(rng.integers(0, 3))
    pd = torch.zeros((m, ld), dtype=torch.float64, device="cuda")
    pd[:, :n] = torch.from_numpy(np.ascontiguousarray(p.T)).cuda()
    wd = torch.zeros(ld, dtype=torch.float64, device="cuda")
    wd[:n] = torch.from_numpy(w_swa).cuda()
    xd, yd = torch.from_numpy(np.ascontiguousarray(x.T)).cuda(), torch.from_numpy(np.ascontiguousarray(y.T)).cuda()
    torch.cuda.synchronize()
    for borrow in (False, True):
        b2.infer_setup_dev(table, n, m, wd.data_ptr(), pd.data_ptr(), ld, xd.data_ptr(), yd.data_ptr(), dims[0], dims[-1], b, 0.8, borrow=borrow)
        assert np.array_equal(b2.logdensity(zs), a.logdensity(zs))
    b2.infer_setup(table, n, m, w_swa, p, x, y, 0.8)   # drop the borrowed pointers before the tensors go
    del pd, wd, xd, yd
    # 6. a few transitions of each gradient sampler (device gradients)
    for fn in (samplers.mala, samplers.hmc, samplers.nuts):
        out = fn(a.logdensity_grad, m, 4, 0.05, np.random.default_rng(case))
        assert np.all(np.isfinite(np.asarray(out[1], dtype=np.float64)))
print("guard_fuzz_api: %d cases done" % cases, flush=True)
for cc in (a, b2, c3):
    cc.close()
