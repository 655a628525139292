"""BASELINE config 4: construct-only on a flattened N = 5,200,266 conv-net weight vector, K = 200 snapshots, M = 20
(random-walk snapshot stream, device-resident).  Prints wall time and the per-kernel device times."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

N, K, M = 5200266, 200, 20
ldw = N + (N & 1)
gen = torch.Generator(device="cuda").manual_seed(0)
snaps = torch.empty((K, ldw), device="cuda", dtype=torch.float32)
cur = 0.02 * torch.randn(ldw, generator=gen, device="cuda", dtype=torch.float32)
for k in range(K):
    cur = cur + 0.002 * torch.randn(ldw, generator=gen, device="cuda", dtype=torch.float32)
    snaps[k] = cur
torch.cuda.synchronize()
ctx = si.Context(0)
for rep in range(3):
    ctx.set_profiling(rep == 1)
    ctx.reset_stats()
    t0 = time.perf_counter()
    ctx.construct_begin(N, K)
    ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, np.arange(1, K + 1, dtype=np.float64))
    _, _, s, _ = ctx.construct_finish(M, want_swa=False, want_p=False)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    if rep == 1:
        st = ctx.stats()
        for k in ("push", "gram", "gram_reduce", "project", "eig_host"):
            v = st[k]
            print("  %-12s %8.3f ms  %6.2f TB/s algorithmic  %6.1f TFLOP/s" % (k, v["ms"], v["bytes"] / max(v["ms"], 1e-9) / 1e9, v["flops"] / max(v["ms"], 1e-9) / 1e9))
    else:
        print("construct N=%d K=%d M=%d: wall %.2f ms (s1 = %.4g)" % (N, K, M, wall, s[0]))
