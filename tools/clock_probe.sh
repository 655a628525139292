#!/bin/bash
# sclk / power while the persistent loop runs: a plain Python process against the C harness (same kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}
sample() { for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g'; echo; sleep 0.25; done; }
echo "== python plain (5 x 100000 transitions)"
python3 - <<PY &
import os, sys, time
sys.argv = ["x"]
sys.path.insert(0, "$R")
import numpy as np
import subspaceinference_jl_amd as si
dims, acts, b, m = [2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 20
table, off = [], 0
for fin, fout, act in zip(dims[:-1], dims[1:], acts):
    table.append((fin, fout, act, off, off + fin * fout)); off += fin * fout + fout
rng = np.random.default_rng(0)
ctx = si.Context(0)
ctx.infer_setup(table, off, m, 0.3 * rng.standard_normal(off), 0.05 * rng.standard_normal((off, m)), rng.standard_normal((2, b)), rng.standard_normal((1, b)), 1.0)
ctx.sample_rwmh(20, 0.1, seed=1)
time.sleep(1.0)
for _ in range(5):
    t0 = time.perf_counter(); ctx.sample_rwmh(100000, 0.1, seed=1, want_z=False); print("python: %.2f us" % ((time.perf_counter() - t0) / 100000 * 1e6), flush=True)
PY
sleep 4.5; sample; wait
echo "== C harness (3 x generic + 3 x specialised, 100000 transitions)"
$R/tools/bin/chain_spec_bench_nb1 2 100000 1 | grep "specialised.*transition" &
sleep 8; sample; wait
