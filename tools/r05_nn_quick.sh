#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_chain_grid.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -5 || exit 1
timeout -k 10 300 python tools/nn_example_modes.py ${1:-021} ${2:-1,8,64,512} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_nn_modes.log
for l in 1 8; do timeout -k 5 60 tools/bin/cgb_stamps 64 2 1000 $l 0 | grep -E "grid loop|transitions|workgroup 0" | tail -3 | cut -c1-560; done
CG_NOSTAGE=1 timeout -k 5 60 tools/bin/cgb_clean 512 2 400 1 0 | grep -E "us per" | tail -2
timeout -k 5 60 tools/bin/cgb_clean 512 2 400 1 0 | grep -E "us per" | tail -2
