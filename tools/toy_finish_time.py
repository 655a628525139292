"""README toy at its real shape (README.md:52-79: N = 682, batchsize 1 x 100 observations x T = 10 epochs => K = 1000 > N,
M = 3): time of si_construct_finish (psvd + P, src/subspace_construction.jl:63,65) and s / P against a LAPACK SVD of the
deviation matrix read back from the device."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

n, k, m = 682, 1000, 3
rng = np.random.default_rng(0)
w0 = rng.standard_normal(n)
snaps = (w0[None, :] + np.cumsum(0.01 * rng.standard_normal((k, n)), axis=0)).astype(np.float32)
ns = np.repeat(np.arange(1, 11, dtype=np.float64), 100)
with si.Context(0) as ctx:
    for rep in range(4):
        ctx.construct_begin(n, k)
        for j in range(k):
            ctx.construct_push(snaps[j], ns[j])
        ctx.synchronize()
        t0 = time.perf_counter()
        w_swa, p, s, kk = ctx.construct_finish(m)
        dt = (time.perf_counter() - t0) * 1e3
        a = ctx.construct_get_A(0, k)
        print("run %d: si_construct_finish %.3f ms (K = %d, N = %d, M = %d)" % (rep, dt, kk, n, m), flush=True)
    u, sv, vt = np.linalg.svd(a, full_matrices=False)
    pr = u[:, :m] * sv[:m]
    sign = np.sign(np.sum(p * pr, axis=0))
    print("s rel err %.2e   P rel err (up to sign) %.2e" % (np.max(np.abs(s - sv[:m]) / sv[:m]), np.max(np.abs(p * sign - pr)) / np.max(np.abs(pr))))
