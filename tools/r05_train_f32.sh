#!/bin/bash
# fp32 training step at cfg2: per-kernel times of one step (rocprofv3 kernel trace) + the bench's two training numbers
set -o pipefail
mkdir -p gpurun_out
cat > /tmp/train_f32_probe.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import subspaceinference_jl_amd as si
from oracle import subspace_oracle as so
dims, acts, B = [128, 960, 960, 1], [1, 1, 0], 100000
table, n = so.layer_table(dims, acts)
rng = np.random.default_rng(0)
x, y = rng.standard_normal((128, B)), rng.standard_normal((1, B))
w0 = (0.05 * rng.standard_normal(n)).astype(np.float32)
ctx = si.Context(0)
ids = np.arange(B)
for name, xx, yy in (("f64", x, y), ("f32", x.astype(np.float32), y.astype(np.float32))):
    ctx.train_setup(table, n, w0, xx, yy, B, 2, 1e-3, 0.9, 0.999)
    l0 = ctx.train_step(ids)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.train_step(ids, want_loss=False)
    ctx.synchronize()
    print("%s: %.3f ms per full-batch step, first loss %.8f, loss after 11 steps %.8f" % (name, (time.perf_counter() - t0) / 10 * 1e3, l0, ctx.train_step(ids)), flush=True)
ctx.close()
PY
timeout -k 10 300 python /tmp/train_f32_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_train_f32.log || exit 1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tr
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -- python3 /tmp/train_f32_probe.py > /dev/null 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/prof_tr 24 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r05_train_f32.log
