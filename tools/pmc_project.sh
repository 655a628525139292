#!/bin/bash
# PMC passes on the wide-subspace projection kernels (one rank's cfg5 share: N = 6.39 M, K = 128, M = 64), fp64 and fp32-stored A
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_project
mkdir -p $O
rm -f $O/summary.txt
cd /tmp && export TMPDIR=/tmp
for mode in "" a32; do
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" FETCH_SIZE WRITE_SIZE; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-30)$mode
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o p -- python3 $R/tools/construct_times.py 6392257 128 64 $mode > /dev/null 2> $O/p_$tag.err || echo "pass failed: $c $mode"
    echo "== ${mode:-f64} --pmc $c" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/p_$tag project_glds >> $O/summary.txt
    rm -rf $O/p_$tag
  done
done
cat $O/summary.txt
