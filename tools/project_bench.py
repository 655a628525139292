"""K3 for wide subspaces (M > 32) at one GPU's share of cfg5 and at the full cfg5 length: the slab-streaming kernel
(kernels_project.hip) against the generic GEMM it replaces (SI_PROJECT_GEMM=1), through the C ABI (si_construct_finish)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6392257
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
M = int(sys.argv[3]) if len(sys.argv) > 3 else 64
gen = torch.Generator(device="cuda").manual_seed(0)
ctx = si.Context(0)
ctx.construct_begin(N, K)
cur = 0.02 * torch.randn(N, generator=gen, device="cuda", dtype=torch.float32)
for k in range(K):
    cur = cur + 0.002 * torch.randn(N, generator=gen, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    ctx.construct_push_dev(cur.data_ptr(), 0, float(k + 1))
    ctx.synchronize()
ctx.construct_gram()
ctx.set_profiling(True)
for rep in range(3):
    ctx.reset_stats()
    _, _, s, _ = ctx.construct_finish(M, want_swa=False, want_p=False)
    st = ctx.stats()["project"]
print("N=%d K=%d M=%d (%s): projection %.3f ms = %.2f TB/s algorithmic, %.1f TFLOP/s; s1=%.6g sM=%.6g" % (
    N, K, M, "GEMM" if os.environ.get("SI_PROJECT_GEMM") == "1" else "slab stream", st["ms"],
    st["bytes"] / st["ms"] / 1e9, st["flops"] / st["ms"] / 1e9, s[0], s[-1]))
wptr, pptr, ld, _ = ctx.construct_result_ptr()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from gpu_helpers import dev_view as _dev_view  # noqa: E402
p_t = _dev_view(pptr, (M, ld))
ptp = (p_t @ p_t.T).cpu().numpy()
print("  max |P'P - diag(s^2)| / s1^2 = %.2e" % (np.abs(ptp - np.diag(s ** 2)).max() / s[0] ** 2))
