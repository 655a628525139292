#!/bin/bash
# dW / dX of cfg2's layers: the LDS-DMA weight-gradient kernel (96 x 192 and 96 x 128 tiles) against the register-staged one.
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/bwd_knobs.log
: > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 5 60 $R/tools/bin/bwd_bench ${SHAPE:-960 960 100000} >> $L 2>&1 || { echo "FAILED" >> $L; cat $L; exit 1; }; }
run SI_BWD_NODMA=1
run X=0
run SI_BWD_BN=128 SI_BWD_NSPLIT=6
run SI_BWD_NODMA=1
run X=0
SHAPE="960 128 100000" run SI_BWD_NODMA=1
SHAPE="960 128 100000" run X=0
SHAPE="950 962 65536" run X=0
cat $L
