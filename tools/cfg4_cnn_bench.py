"""BASELINE config 4 as a SAMPLED model: conv3x3 3->64->128->256->256 (+ MaxPool 2 each) + fc 1024->4096->10 on 32x32x3
images (N = 5,200,266), M = 20.  Times RWMH transitions, the gradient of the log-density and a training step, and prints the
per-class device times (conv = implicit-GEMM convolution kernels; conv_aux = weight re-pack, MaxPool, layout changes)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):   # the development build (guard-page runs: SI_GUARD_ALLOC=end|begin)
    si._capi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin",
                                     "libsubspace_hip_dev.so")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = 20
SPEC = [("conv", 64), ("pool",), ("conv", 128), ("pool",), ("conv", 256), ("pool",), ("conv", 256), ("pool",), ("flatten",),
        ("dense", 4096, 1), ("dense", 10, 0)]
table, off, (w, h, c) = [], 0, (32, 32, 3)
feat = None
flops = 0.0
for e in SPEC:
    if e[0] == "conv":
        table.append(("conv", (3, 3, c, e[1]), (w, h), (1, 1), (1, 1), (1, 1), 1, off, off + 9 * c * e[1]))
        flops += 2.0 * 9 * c * e[1] * w * h
        off += 9 * c * e[1] + e[1]
        c = e[1]
    elif e[0] == "pool":
        table.append(("maxpool", (2, 2), c, (w, h), (2, 2)))
        w, h = w // 2, h // 2
    elif e[0] == "flatten":
        table.append(("flatten", c, (w, h)))
        feat = w * h * c
    else:
        table.append((feat, e[1], e[2], off, off + feat * e[1]))
        flops += 2.0 * feat * e[1]
        off += feat * e[1] + e[1]
        feat = e[1]
N = off
assert N == 5200266
rng = np.random.default_rng(0)
w_swa = np.zeros(N)
for r in table:
    if r[0] == "conv":
        cnt = int(np.prod(r[1]))
        w_swa[r[7]:r[7] + cnt] = rng.standard_normal(cnt) * np.sqrt(2.0 / (9 * r[1][2]))
    elif not isinstance(r[0], str):
        w_swa[r[3]:r[3] + r[0] * r[1]] = rng.standard_normal(r[0] * r[1]) * np.sqrt(2.0 / r[0])
p = np.asfortranarray(1e-3 * rng.standard_normal((N, M)))
x = np.asfortranarray(rng.standard_normal((32 * 32 * 3, B)))
y = np.asfortranarray(rng.standard_normal((10, B)))
ctx = si.Context(0)
ctx.infer_setup(table, N, M, w_swa, p, x, y, 1.0)
if not os.environ.get("CFG4_SETUP_ONLY"):   # tools/cfg4_cnn_grad_profile.py imports the set-up only
    ctx.sample_rwmh(3, 0.01, seed=1)
    ctx.set_profiling(True)
    ctx.reset_stats()
    steps = 10
    t0 = time.perf_counter()
    z, lp, acc = ctx.sample_rwmh(steps, 0.01, seed=1)
    dt = (time.perf_counter() - t0) / steps
    st = ctx.stats()
    print("cfg4 CNN, B = %d images: %.2f ms per RWMH transition, %.1f GFLOP per forward -> %.1f TFLOP/s overall" %
          (B, dt * 1e3, flops * B / 1e9, flops * B / dt / 1e12))
    for k in ("reconstruct", "conv", "conv_aux", "dense", "sse", "rwmh"):
        v = st[k]
        print("  %-12s %8.3f ms per step  %7.2f TFLOP/s  %7.2f TB/s algorithmic" %
              (k, v["ms"] / steps, v["flops"] / max(v["ms"], 1e-9) / 1e9, v["bytes"] / max(v["ms"], 1e-9) / 1e9))
    ctx.set_profiling(False)
    zz = np.ascontiguousarray(z[:, -1, 0])
    ctx.logdensity_grad(zz)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.logdensity_grad(zz)
    print("  value + gradient of the log-density: %.2f ms (%.1f TFLOP/s on 3x the forward flops)" %
          ((time.perf_counter() - t0) / 3 * 1e3, 3 * flops * B / ((time.perf_counter() - t0) / 3) / 1e12))
    bt = min(B, 1024)
    ctx.train_setup(table, N, w_swa.astype(np.float32), x, y, bt, 2, 1e-3, 0.9, 0.999)
    ids = np.arange(bt)
    ctx.train_step(ids)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.train_step(ids, want_loss=False)
    ctx.synchronize()
    print("  ADAM training step on %d images: %.2f ms" % (bt, (time.perf_counter() - t0) / 5 * 1e3))
