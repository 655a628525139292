#!/bin/bash
# One short --pmc pass per candidate program, to find which one rocprofv3's counter collection dies on.  Stops at the first
# step that had to be killed (a hang), carries on after an ordinary crash.  Log: gpurun_out/pmc_bisect.log
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_bisect
mkdir -p $O
L=$R/gpurun_out/pmc_bisect.log
: > $L
cd /tmp && export TMPDIR=/tmp
step() {
  tag=$1; shift
  echo "== $tag: $*" >> $L
  timeout -k 10 ${T:-150} rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/$tag -o p -- "$@" > $O/$tag.out 2> $O/$tag.err
  rc=$?
  echo "   rc=$rc" >> $L
  grep -a "bench.py:\|SIGSEGV\|malformed\|Error\|error" $O/$tag.err | tail -6 >> $L
  rm -rf $O/$tag
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   killed: stopping" >> $L; exit 1; fi
}
step construct64 python3 $R/tools/construct_times.py 1047361 100 20
step construct32 python3 $R/tools/construct_times.py 1047361 100 20 a32
step chain32 python3 $R/tools/f32_chain.py f32 12
T=400 step bench python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline
cat $L
