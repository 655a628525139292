#!/bin/bash
# Round-4 guard-page campaign over the development library (csrc/guard_alloc.hip): the Dense sweep (now with the fp32 density,
# the device-resident chain loop and the LDS-DMA weight gradient), the Gram / construction sweep (K > N route, fp32-stored A) and
# the round-4 GPU test files, with every buffer flush against an unmapped granule at its END, then at its BEGINNING.
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/guard_r4.log
: > $L
for mode in end begin; do
  echo "== SI_GUARD_ALLOC=$mode: tools/guard_fuzz.py 70 cases (SI_FUZZ_BIG)" >> $L
  SI_PROBE_DEV=1 SI_GUARD_ALLOC=$mode SI_FUZZ_BIG=1 timeout -k 10 500 python3 $R/tools/guard_fuzz.py 70 77 > $R/gpurun_out/guard_r4_fuzz_$mode.log 2>&1
  rc=$?; tail -2 $R/gpurun_out/guard_r4_fuzz_$mode.log >> $L; echo "   rc=$rc" >> $L
  if [ $rc -ne 0 ]; then grep -a "fault\|Fault" $R/gpurun_out/guard_r4_fuzz_$mode.log | tail -3 >> $L; cat $L; exit 1; fi
  echo "== SI_GUARD_ALLOC=$mode: tools/guard_fuzz_gram.py 40 cases" >> $L
  SI_PROBE_DEV=1 SI_GUARD_ALLOC=$mode timeout -k 10 300 python3 $R/tools/guard_fuzz_gram.py 40 78 > $R/gpurun_out/guard_r4_gram_$mode.log 2>&1
  rc=$?; tail -1 $R/gpurun_out/guard_r4_gram_$mode.log >> $L; echo "   rc=$rc" >> $L
  if [ $rc -ne 0 ]; then cat $L; exit 1; fi
  echo "== SI_GUARD_ALLOC=$mode: round-4 GPU tests" >> $L
  SI_TEST_LIB=tools/bin/libsubspace_hip_dev.so SI_GUARD_ALLOC=$mode timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_f32.py $R/tests/test_gpu_chain.py $R/tests/test_gpu_a32.py $R/tests/test_gpu_parity.py -q -m gpu -k "not fuzz" -x > $R/gpurun_out/guard_r4_tests_$mode.log 2>&1
  rc=$?; tail -2 $R/gpurun_out/guard_r4_tests_$mode.log >> $L; echo "   rc=$rc" >> $L
  if [ $rc -ne 0 ]; then cat $L; exit 1; fi
done
cat $L
