#!/bin/bash
# the fp32 dense kernel with 32-deep k tiles in two stages against the shipped 16-deep / three-stage ring (tools/gemm_f32_bench.hip)
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/f32_bk32.log
: > $L
for rep in 1 2; do
  echo "== 960x960 fused head, pass $rep" >> $L
  timeout -k 5 120 $R/tools/bin/gemm_f32_bench 960 960 100000 0x8501 1 >> $L 2>&1 || { echo FAILED >> $L; cat $L; exit 1; }
done
echo "== 960x960 plain store" >> $L
timeout -k 5 120 $R/tools/bin/gemm_f32_bench 960 960 100000 0xF501 0 >> $L 2>&1 || { echo FAILED >> $L; cat $L; exit 1; }
echo "== 960x128 plain store (layer 1)" >> $L
timeout -k 5 120 $R/tools/bin/gemm_f32_bench 960 128 100000 0xF501 0 >> $L 2>&1 || { echo FAILED >> $L; cat $L; exit 1; }
cat $L
