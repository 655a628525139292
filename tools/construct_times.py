"""Per-kernel device times of one construction (push batch + Gram + eig + projection) at a given (N, K, M), through the C ABI."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

N, K, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ldw = N + (N & 1)
gen = torch.Generator(device="cuda").manual_seed(0)
snaps = torch.empty((K, ldw), device="cuda", dtype=torch.float32)
cur = 0.02 * torch.randn(ldw, generator=gen, device="cuda", dtype=torch.float32)
for k in range(K):
    cur = cur + 0.002 * torch.randn(ldw, generator=gen, device="cuda", dtype=torch.float32)
    snaps[k] = cur
torch.cuda.synchronize()
ctx = si.Context(0)
ctx.set_profiling(True)
for rep in range(3):
    ctx.reset_stats()
    ctx.construct_begin(N, K)
    ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, np.arange(1, K + 1, dtype=np.float64))
    _, _, s, _ = ctx.construct_finish(M, want_swa=False, want_p=False)
    ctx.synchronize()
st = ctx.stats()
print("N=%d K=%d M=%d %s: " % (N, K, M, " ".join("%s=%s" % (k, os.environ[k]) for k in os.environ if k.startswith("SI_"))) +
      "  ".join("%s %.3f ms (%.1f TF/s, %.2f TB/s)" % (k, st[k]["ms"], st[k]["flops"] / max(st[k]["ms"], 1e-9) / 1e9,
                                                    st[k]["bytes"] / max(st[k]["ms"], 1e-9) / 1e9)
               for k in ("push", "gram", "gram_reduce", "project")) + "  s1=%.6g" % s[0])
