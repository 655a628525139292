"""Development probe of compute_dtype = SI_F32 (not product code): cfg2-shaped density in both precisions on one GPU --
lp difference, per-class device time of a few transitions, divergence of the accept decisions of a chain.
    python tools/f32_probe.py [B] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from subspaceinference_jl_amd import _capi  # noqa: E402

if os.environ.get("SI_PROBE_LIB"):   # A/B of two builds of the library in one gpurun call (same device)
    _capi.LIB_PATH = os.path.abspath(os.environ["SI_PROBE_LIB"])

DIMS, ACTS = [128, 960, 960, 1], [1, 1, 0]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M = 20


def layer_table(dims, acts):
    table, off = [], 0
    for fin, fout, act in zip(dims[:-1], dims[1:], acts):
        table.append((fin, fout, act, off, off + fin * fout))
        off += fin * fout + fout
    return table, off


def main():
    table, n = layer_table(DIMS, ACTS)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((DIMS[0], B))
    y = rng.standard_normal((1, B))
    w = np.concatenate([np.concatenate([(rng.uniform(-1, 1, (fo, fi)) * np.sqrt(6.0 / (fi + fo))).reshape(-1, order="F"), np.zeros(fo)])
                        for fi, fo in zip(DIMS[:-1], DIMS[1:])])
    p = 0.01 * rng.standard_normal((n, M))
    res = {}
    for mode, dt in (("f64", _capi.SI_F64), ("f32", _capi.SI_F32)):
        with si.Context(0) as ctx:
            ctx.infer_setup(table, n, M, w, p, x, y, 1.0, compute_dtype=dt)
            z = 0.1 * rng.standard_normal((M, 1)) if mode == "f64" else res["z"]
            res["z"] = z
            lp = ctx.logdensity(z)
            ctx.sample_rwmh(5, 0.1, seed=1)   # warm
            ctx.set_profiling(True)
            ctx.reset_stats()
            ctx.sample_rwmh(STEPS, 0.1, seed=1)
            st = ctx.stats()
            ctx.set_profiling(False)
            t0 = time.perf_counter()
            zs, lps, acc = ctx.sample_rwmh(STEPS, 0.1, seed=1)
            dt_s = time.perf_counter() - t0
            res[mode] = dict(lp=float(lp[0]), lps=lps[:, 0].copy(), zs=zs[:, :, 0].copy(), acc=float(acc[0]))
            per = {nm: (v["ms"] / max(1, v["launches"]), v["launches"]) for nm, v in st.items()}
            fl = st["dense_main"]["flops"] / max(1, st["dense_main"]["launches"])
            print("%s: lp %.10e  %.3f ms/step wall  dense_main %.4f ms = %.1f TFLOP/s   per-class avg ms: %s" % (
                mode, lp[0], 1e3 * dt_s / STEPS, per["dense_main"][0], fl / (per["dense_main"][0] * 1e-3) / 1e12,
                {k: round(v[0], 4) for k, v in per.items() if v[1]}), flush=True)
    a, b = res["f64"], res["f32"]
    print("lp rel diff f32 vs f64: %.3e" % (abs(a["lp"] - b["lp"]) / abs(a["lp"])))
    same = np.all(a["zs"] == b["zs"], axis=0)
    first = int(np.argmin(same)) if not same.all() else -1
    print("chain of %d steps: accept rate f64 %.3f f32 %.3f; first step whose z differs: %d" % (STEPS, a["acc"], b["acc"], first))


if __name__ == "__main__":
    main()
