#!/bin/bash
# dW / dX of cfg2's 960 x 960 layer under the harness knobs (1 = cache-hot operands, 2 = no global loads in the k loop) and
# with other row tiles / split counts for dW.  Log: gpurun_out/bwd_knobs.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/bwd_knobs.log
: > $L
for k in 0 1 2 0; do $R/tools/bin/bwd_bench 960 960 100000 $k >> $L 2>&1 || exit 1; done
for cfg in "96 6" "128 8" "96 19" "96 32" "64 4" "128 16"; do
  set -- $cfg
  echo "== SI_BWD_BM=$1 SI_BWD_NSPLIT=$2" >> $L
  SI_BWD_BM=$1 SI_BWD_NSPLIT=$2 $R/tools/bin/bwd_bench 960 960 100000 0 >> $L 2>&1 || exit 1
done
cat $L
