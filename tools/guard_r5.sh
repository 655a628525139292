#!/bin/bash
# Round-5 guard-page campaign over the development library (csrc/guard_alloc.hip: every buffer of the library flush against an
# unmapped granule at its END, then at its BEGINNING): the narrow-chain sweep (generic and run-time specialised kernels, the
# fragment-ordered weight buffers, the permutation table, the sharded barrier counters), the Dense sweep, the CNN sweep, and the
# round-5 GPU test files (fp32 training step, fp32 conv chains, chain grid, fp32-stored A).
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/guard_r5.log
: > $L
for mode in end begin; do
  for spec in "guard_fuzz_chain.py 16 81 300" "guard_fuzz.py 40 82 400" "guard_fuzz_cnn.py 25 83 400"; do
    set -- $spec
    echo "== SI_GUARD_ALLOC=$mode: tools/$1 $2 cases" >> $L
    SI_PROBE_DEV=1 SI_GUARD_ALLOC=$mode timeout -k 10 $4 python3 $R/tools/$1 $2 $3 > $R/gpurun_out/guard_r5_${1%.py}_$mode.log 2>&1
    rc=$?; tail -1 $R/gpurun_out/guard_r5_${1%.py}_$mode.log >> $L; echo "   rc=$rc" >> $L
    if [ $rc -ne 0 ]; then grep -a "fault\|Fault\|Error" $R/gpurun_out/guard_r5_${1%.py}_$mode.log | tail -3 >> $L; cat $L; exit 1; fi
  done
  echo "== SI_GUARD_ALLOC=$mode: round-5 GPU tests" >> $L
  SI_TEST_LIB=tools/bin/libsubspace_hip_dev.so SI_GUARD_ALLOC=$mode timeout -k 10 900 python3 -m pytest $R/tests/test_gpu_chain_grid.py $R/tests/test_gpu_train_f32.py $R/tests/test_gpu_f32.py $R/tests/test_gpu_a32.py $R/tests/test_gpu_conv.py $R/tests/test_gpu_panel.py -q -m gpu -k "not cfg4 and not long_chain" -x > $R/gpurun_out/guard_r5_tests_$mode.log 2>&1
  rc=$?; tail -2 $R/gpurun_out/guard_r5_tests_$mode.log >> $L; echo "   rc=$rc" >> $L
  if [ $rc -ne 0 ]; then cat $L; exit 1; fi
done
cat $L
