#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_bisect3
mkdir -p $O
L=$R/gpurun_out/pmc_bisect3.log
: > $L
cd /tmp && export TMPDIR=/tmp
for n in 2000 2600; do
  echo "== pmc f64 long64=$n" >> $L
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/t$n -o p -- python3 $R/tools/f32_after_f64.py f64 long64=$n > $O/t$n.out 2> $O/t$n.err
  rc=$?; echo "   rc=$rc" >> $L
  grep -a "f32_after_f64\|SIGSEGV\|malformed" $O/t$n.err | tail -8 >> $L
  f=$(find $O/t$n -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && echo "   dispatches in the trace: $(($(wc -l < $f) - 1))" >> $L
  rm -rf $O/t$n
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   killed: stopping" >> $L; break; fi
done
cat $L
