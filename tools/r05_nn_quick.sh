#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_chain_grid.py -x -q -m gpu 2>&1 | tail -5 || exit 1
timeout -k 10 300 python tools/nn_example_modes.py ${1:-021} ${2:-1,8,64,512} 2>&1 | tee gpurun_out/r05_nn_modes.log
