#!/bin/bash
# The panel kernel for a wide first layer (csrc/kernels_gemm_panel.hip): bits against dense_f64_kernel on edge shapes, then the
# bench's transition with and without it (SI_PANEL=0 routes the layer back to dense_f64_kernel).
set -o pipefail
cd "$(dirname "$0")/.."
P=tools/bin/panel_bench
for shape in "960 128 100000 1" "50 48 1000 1" "962 128 100001 2" "64 16 20000 0" "1024 112 12345 3" "130 32 129 1" "16 16 1 1" "960 128 127 1" "512 13 20000 1" "300 100 30000 2" "960 127 10000 1" "64 1 5000 0" "130 5 129 3" "1024 2 65536 1"; do
  echo "== out in B act: $shape"
  timeout -k 10 120 $P $shape 512,7 2>&1 | grep -v amdgpu.ids || exit 1
done
