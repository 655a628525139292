#!/bin/bash
# Which stage of bench.py's sampling sequence does rocprofv3's counter collection die on?  (tools/f32_after_f64.py stages.)
# First the guard-page development library over the whole sequence (an out-of-bounds write would fault there), then one --pmc
# pass per stage subset.  Stops at the first step that had to be killed.  Log: gpurun_out/pmc_bisect2.log
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_bisect2
mkdir -p $O
L=$R/gpurun_out/pmc_bisect2.log
: > $L
check() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "   killed: stopping" >> $L; cat $L; exit 1; fi; }
for mode in end begin; do
  echo "== guard $mode" >> $L
  SI_PROBE_LIB=$R/tools/bin/libsubspace_hip_dev.so SI_GUARD_ALLOC=$mode timeout -k 10 200 python3 $R/tools/f32_after_f64.py f64 map prof long32 > $O/guard_$mode.out 2> $O/guard_$mode.err
  rc=$?; echo "   rc=$rc" >> $L; grep -a "f32_after_f64\|fault\|Fault" $O/guard_$mode.err | tail -8 >> $L; check $rc
done
cd /tmp && export TMPDIR=/tmp
step() {
  tag=$1; shift
  echo "== pmc $tag: $*" >> $L
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/$tag -o p -- python3 $R/tools/f32_after_f64.py "$@" > $O/$tag.out 2> $O/$tag.err
  rc=$?; echo "   rc=$rc" >> $L
  grep -a "f32_after_f64\|SIGSEGV\|malformed" $O/$tag.err | tail -8 >> $L
  rm -rf $O/$tag
  check $rc
}
step plain
step prof prof
step long32 long32
step f64 f64
step map f64 map
step all f64 map prof long32
cat $L
