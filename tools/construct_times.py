"""Per-kernel device times of one construction (push batch + Gram + eig + projection) at a given (N, K, M), through the C ABI."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

if os.environ.get("SI_PROBE_LIB"):   # another build of the library, for A/B runs on one device
    si._capi.LIB_PATH = os.path.abspath(os.environ["SI_PROBE_LIB"])
N, K, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A32 = len(sys.argv) > 4 and sys.argv[4] == "a32"   # si_construct_set_storage(SI_F32): the opt-in fp32 storage of the deviation matrix
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
    if A32:
        ctx.construct_set_storage(0)
    ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, np.arange(1, K + 1, dtype=np.float64))
    _, _, s, _ = ctx.construct_finish(M, want_swa=False, want_p=False)
    ctx.synchronize()
import time
walls = []
ctx.set_profiling(False)
for rep in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.construct_begin(N, K)
    if A32:
        ctx.construct_set_storage(0)
    ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, np.arange(1, K + 1, dtype=np.float64))
    ctx.construct_finish(M, want_swa=False, want_p=False)
    ctx.synchronize()
    walls.append((time.perf_counter() - t0) * 1e3)
st = ctx.stats()
print("N=%d K=%d M=%d A %s: wall (batched push + finish, median of 3) %.3f ms | device ms: " % (N, K, M, "fp32" if A32 else "fp64", sorted(walls)[1]) +
      "  ".join("%s %.3f" % (k, st[k]["ms"]) for k in ("push", "gram", "gram_reduce", "project", "eig_host")) + "  s1=%.6g" % s[0])
