#!/bin/bash
# the ceiling of the fp32 density's main kernel (960 x 960, B = 1e5) with the head -- and the whole epilogue -- removed.
# tools/bin/gemm_f32_bench_k0 = kernels_gemm_f32.hip as shipped, _k1 = -DSI_F32_KNOB=1 (returns after the k loop; "maxerr" is then meaningless).
# Variants: 192x128 two stages, 2 workgroups per CU (shipped) | 192x256 two stages at 2 / 1 workgroups per CU | 192x256 with 16 waves.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for k in 0 1; do
  for fused in 0 1; do
    echo "== knob $k ($([ $k = 0 ] && echo shipped epilogue || echo NO epilogue)), fused head $fused"
    timeout -k 10 200 $R/tools/bin/gemm_f32_bench_k$k 960 960 100000 0x30480 $fused || exit 1
  done
done
