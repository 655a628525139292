#!/bin/bash
# A longer run of the random-shape sweeps on the SHIPPED library with every result computed twice and compared bit for bit
# (SI_FUZZ_REPEAT: a race inside a kernel would show as a rare difference).  Log: gpurun_out/fuzz_campaign.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/fuzz_campaign.log
: > $L
run() {
  tool=$1; cases=$2; seed=$3
  echo "== $tool $cases cases, seed $seed" >> $L
  SI_FUZZ_REPEAT=1 timeout -k 10 900 python3 $R/tools/$tool $cases $seed > $R/gpurun_out/fuzz_$tool.log 2>&1
  rc=$?; tail -1 $R/gpurun_out/fuzz_$tool.log >> $L; echo "   rc=$rc" >> $L
  echo "$tool rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/fuzz_$tool.log >> $L; cat $L; exit 1; fi
}
run guard_fuzz.py 300 4101
run guard_fuzz_gram.py 200 4102
run guard_fuzz_cnn.py 120 4103
run guard_fuzz_api.py 80 4104
run guard_fuzz_e2e.py 40 4105
run guard_fuzz_comm.py 40 4106
cat $L
