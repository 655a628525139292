"""BASELINE config 4's CNN sampled with compute_dtype = SI_F32 beside SI_F64: ms per RWMH transition, the
per-class device times, lp against the fp64 density, and how many accept decisions of a chain differ between the precisions."""
import os
import sys
import time

import numpy as np

os.environ["CFG4_SETUP_ONLY"] = "1"
sys.argv = [sys.argv[0]] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cfg4_cnn_bench as c4  # noqa: E402  (builds the layer table and the fp64 set-up)

si, ctx = c4.si, c4.ctx
itr = 200
out = {}
for name, dt in (("f64", si._capi.SI_F64), ("f32", si._capi.SI_F32)):
    ctx.infer_setup(c4.table, c4.N, c4.M, c4.w_swa, c4.p, c4.x, c4.y, 1.0, compute_dtype=dt)
    ctx.sample_rwmh(3, 0.01, seed=1)
    ctx.set_profiling(True)
    ctx.reset_stats()
    t0 = time.perf_counter()
    z, lp, acc = ctx.sample_rwmh(10, 0.01, seed=1)
    per = (time.perf_counter() - t0) / 10
    st = ctx.stats()
    ctx.set_profiling(False)
    print("cfg4 CNN, B = %d images, %s: %.2f ms per RWMH transition (%.1f TFLOP/s overall)" % (c4.B, name, per * 1e3, c4.flops * c4.B / per / 1e12))
    for k in ("reconstruct", "conv", "conv_aux", "dense", "sse", "rwmh"):
        v = st[k]
        print("  %-12s %8.3f ms per step  %7.2f TFLOP/s" % (k, v["ms"] / 10, v["flops"] / max(v["ms"], 1e-9) / 1e9))
    out[name] = ctx.sample_rwmh(itr, 2e-4, seed=7)
    zz = np.asfortranarray(out["f64"][0][:, ::20, 0]) if name == "f32" else None
    if name == "f32":   # the fp32 density on states of the fp64 chain against the fp64 chain's own lp
        lp32 = ctx.logdensity(zz)
        lp64 = out["f64"][1][::20, 0]
        print("  lp(f32) against lp(f64) on %d states of the fp64 chain: max rel diff %.2e (lp ~ %.4e)" % (len(lp64), np.abs(lp32 / lp64 - 1).max(), lp64[0]))
z64, z32 = out["f64"][0][:, :, 0], out["f32"][0][:, :, 0]
mv64 = np.any(np.diff(z64, axis=1) != 0, axis=0)
mv32 = np.any(np.diff(z32, axis=1) != 0, axis=0)
same = np.cumprod(mv64 == mv32)
first = int(same.sum()) + 1 if not same.all() else -1
print("  %d transitions, sigma_z 2e-4: accepted %d (f64) / %d (f32); first transition where the decisions differ: %s" %
      (itr, int(mv64.sum()), int(mv32.sum()), first if first >= 0 else "none"))
