"""Random NARROW Dense chains (the class of csrc/kernels_chain_grid.hip / csrc/chain_spec.inc) through every sampler schedule:
si_set_chain_loop 0 (one launch per layer and step) against 1 / 2 (kernels specialised at run time where the chain is of that
class) and 3 / 4 (generic kernels), 1 ... 70 chains -- the persistent loop at tiles of 16 / 32 / 64 observations, the B-tiled
density, the register-resident stacked density -- bit for bit; the density against the oracle.
Usage: [SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin] guard_fuzz_chain.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402
from oracle import subspace_oracle as so  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):   # the development build (guard-page runs: SI_GUARD_ALLOC=end|begin)
    si._capi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libsubspace_hip_dev.so")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = si.Context(0)
nspec = 0
for case in range(cases):
    nl = int(rng.integers(2, 6))
    dims = [int(rng.choice([1, 2, 3, 7, 12, 16, 20]))] + [int(rng.choice([3, 15, 16, 17, 31, 40, 50, 64, 65, 100, 128, 200, 256])) for _ in range(nl - 1)] + \
           [int(rng.choice([1, 1, 1, 2, 3, 4]))]
    acts = [int(rng.integers(0, 4)) for _ in range(nl)]
    b = int(rng.choice([1, 15, 16, 17, 100, 255, 256, 257, 777, 1000, 1024]))
    m = int(rng.choice([1, 2, 3, 5, 20, 31, 33, 64]))
    table, n = so.layer_table(dims, acts)
    w_swa = 0.3 * rng.standard_normal(n)
    p = np.asfortranarray(0.05 * rng.standard_normal((n, m)))
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)))
    ctx.infer_setup(table, n, m, w_swa, p, x, y, 0.9)
    zs = np.asfortranarray(0.4 * rng.standard_normal((m, 19)))
    lp_ref = np.array([so.logdensity(table, w_swa, p, x, y, 0.9, zs[:, j]) for j in range(3)])
    info = {}
    for nch in (1, int(rng.integers(2, 9)), int(rng.integers(16, 71))):
        out = {}
        for mode in (0, 1, 2, 3, 4):
            ctx.set_chain_loop(mode)
            out[mode] = ctx.sample_rwmh(12, 0.08, seed=case + 7, chain_id0=1, nchains=nch)
            info[(nch, mode)] = ctx.chain_kernel_info()[:2]
            assert all(np.array_equal(a, c) for a, c in zip(out[0], out[mode])), (case, dims, nch, mode, info[(nch, mode)])
            lpd = ctx.logdensity(zs)
            if mode == 0:
                lp0 = lpd
                assert np.allclose(lpd[:3], lp_ref, rtol=1e-9)
            assert np.array_equal(lpd, lp0), (case, dims, mode)
    ctx.set_chain_loop(1)
    spec = any(info[k][0] or info[k][1] for k in info if k[1] in (1, 2))
    nspec += spec
    # the class of chain_spec.inc (spec_applies): a narrow head, every layer <= 256 wide, <= 96 fragment doubles per lane
    in_class = dims[-1] <= 4 and max(dims) <= 256 and sum(-(-((o + 15) // 16) // 4) * ((i + 3) // 4 + 1) for i, o in zip(dims[:-2], dims[1:-1])) <= 96
    assert spec == in_class, (dims, spec, in_class, ctx.chain_kernel_info()[2][:300])
    assert not any(info[k][0] or info[k][1] for k in info if k[1] in (0, 3, 4))
    print("case %d dims %s acts %s B %d M %d: specialised %s %s" % (case, dims, acts, b, m, spec, "" if spec else "(" + ctx.chain_kernel_info()[2][:60] + ")"), flush=True)
ctx.close()
print("%d cases done (%d ran kernels specialised at run time)" % (cases, nspec))
