"""The sampling part of bench.py's sequence on one ctx, in stages that can be left out (argv: any of f64 map long32 prof): construct on
the device -> [f64 chain] -> [output map] -> SI_F32 set-up -> short chain [with event pairs] -> [1000-step chain].  Target for
rocprofv3 --pmc passes and guard-page runs (SI_PROBE_LIB = another build of the library)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from subspaceinference_jl_amd import _capi  # noqa: E402

if os.environ.get("SI_PROBE_LIB"):
    _capi.LIB_PATH = os.path.abspath(os.environ["SI_PROBE_LIB"])
flags = set(sys.argv[1:])
DIMS, ACTS, B, M, K = [128, 960, 960, 1], [1, 1, 0], 100000, 20, 100
table, off = [], 0
for fin, fout, act in zip(DIMS[:-1], DIMS[1:], ACTS):
    table.append((fin, fout, act, off, off + fin * fout))
    off += fin * fout + fout
n = off
rng = np.random.default_rng(0)
x = np.asfortranarray(rng.standard_normal((DIMS[0], B)))
y = np.asfortranarray(rng.standard_normal((1, B)))
ldw = n + (n & 1)
gen = torch.Generator(device="cuda").manual_seed(2)
snaps = torch.zeros(K, ldw, device="cuda", dtype=torch.float32)
snaps[:, :n] = 0.03 * torch.randn(n, generator=gen, device="cuda")[None, :] + 0.01 * torch.cumsum(torch.randn(K, n, generator=gen, device="cuda"), 0)
torch.cuda.synchronize()


def say(s):
    print("f32_after_f64:", s, file=sys.stderr, flush=True)


ctx = si.Context(0)
ctx.construct_begin(n, K)
ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, np.arange(1, K + 1, dtype=np.float64))
ctx.construct_finish(M, want_swa=False, want_p=False)
say("constructed")
if "f64" in flags:
    ctx.infer_setup(table, n, M, None, None, x, y, 1.0)
    z, lp, acc = ctx.sample_rwmh(50, 0.1, seed=100)
    say("f64 chain lp %.10e" % lp[-1, 0])
for f in flags:
    if f.startswith("long64="):   # a long fp64 chain: 7 dispatches per transition (does the profiler survive > 16384 dispatches?)
        z, lp, acc = ctx.sample_rwmh(int(f[7:]), 0.1, seed=100, want_z=False)
        say("f64 %s-step chain done" % f[7:])
if "map" in flags:
    zw, lpw, _, wmap = ctx.sample_rwmh_weights(10, 0.1, seed=100)
    say("output map done")
ctx.infer_setup(table, n, M, None, None, x, y, 1.0, compute_dtype=_capi.SI_F32)
say("f32 set up")
if "prof" in flags:
    ctx.set_profiling(True, classes=["dense_main"])
z32, lp32, _ = ctx.sample_rwmh(12, 0.1, seed=100)
ctx.synchronize()
say("f32 chain lp %.10e" % lp32[-1, 0])
if "prof" in flags:
    say("stats %r" % (ctx.stats()["dense_main"],))
    ctx.set_profiling(False)
if "long32" in flags:
    z32, lp32, _ = ctx.sample_rwmh(1000, 0.1, seed=100)
    say("f32 1000-step chain lp %.10e" % lp32[-1, 0])
ctx.close()
say("done")
