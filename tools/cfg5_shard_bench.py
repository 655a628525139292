"""One GPU's share of BASELINE.json config 5 (wide MLP, N = 51,138,049, M = 64, K = 128, data sharded 8 ways):
row-complete construction on ONE device (the 8-way row shard would hold N/8 rows each) and the data-sharded density
step on this rank's B/8 = 16,384 observations.  Prints per-kernel device times: the bandwidth-bound roofline points
(K1 batched push, K3 projection, K4 reconstruct over the 26 GB P) next to the MFMA-bound ones."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

DIMS, ACTS = [1024, 6656, 6656, 1], [1, 1, 0]
K, M, B_SHARD = 128, 64, 16384
table, off = [], 0
for fin, fout, act in zip(DIMS[:-1], DIMS[1:], ACTS):
    table.append((fin, fout, act, off, off + fin * fout))
    off += fin * fout + fout
N = off
assert N == 51138049
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
for rep in range(2):
    ctx.reset_stats()
    t0 = time.perf_counter()
    ctx.construct_begin(N, K)
    ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, np.arange(1, K + 1, dtype=np.float64))
    _, _, s, _ = ctx.construct_finish(M, want_swa=False, want_p=False)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
st = ctx.stats()
print("construct N=%d K=%d M=%d: wall %.1f ms" % (N, K, M, wall))
for k in ("push", "gram", "gram_reduce", "project", "eig_host"):
    v = st[k]
    print("  %-12s %8.3f ms  %7.2f TB/s algorithmic  %6.1f TFLOP/s" % (k, v["ms"], v["bytes"] / max(v["ms"], 1e-9) / 1e9,
                                                                        v["flops"] / max(v["ms"], 1e-9) / 1e9))
del snaps
torch.cuda.empty_cache()
rng = np.random.default_rng(1)
x = np.asfortranarray(rng.standard_normal((DIMS[0], B_SHARD)))
y = np.asfortranarray(rng.standard_normal((1, B_SHARD)))
ctx.infer_setup(table, N, M, None, None, x, y, 1.0)
ctx.rwmh_begin(12, 0.01, 3, 0, 1, d_total=8 * B_SHARD)
ctx.rwmh_step_accept(ctx.rwmh_step_eval() * 8.0)     # warm-up; "all-reduce" of 8 equal shards
ctx.reset_stats()
t0 = time.perf_counter()
for _ in range(11):
    ctx.rwmh_step_accept(ctx.rwmh_step_eval() * 8.0)
dt = (time.perf_counter() - t0) / 11 * 1e3
z, lp, acc = ctx.rwmh_end()
st = ctx.stats()
print("data-sharded density step (B/8 = %d observations on this rank): %.2f ms per step, lp[-1] = %.3f" % (B_SHARD, dt, lp[-1, 0]))
for k in ("reconstruct", "dense", "dense_main", "sse", "rwmh"):
    v = st[k]
    n = max(1, v["launches"])
    print("  %-12s %8.3f ms per launch  %7.2f TB/s algorithmic  %6.1f TFLOP/s" % (
        k, v["ms"] / n, v["bytes"] / max(v["ms"], 1e-9) / 1e9, v["flops"] / max(v["ms"], 1e-9) / 1e9))
