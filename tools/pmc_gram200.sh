#!/bin/bash
# PMC passes on the Gram kernel at cfg4's shape (N = 5.2 M, K = 200) and at one rank's cfg5 share (N = 6.39 M, K = 128)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_gram200
mkdir -p $O
rm -f $O/summary.txt
cd /tmp && export TMPDIR=/tmp
for shape in "5200266 200 20" "6392257 128 64"; do
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" FETCH_SIZE; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-24)_$(echo $shape | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o p -- python3 $R/tools/construct_times.py $shape > /dev/null 2> $O/p_$tag.err || echo "pass failed: $c $shape"
    echo "== N K M = $shape --pmc $c" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/p_$tag gram_glds >> $O/summary.txt
    rm -rf $O/p_$tag
  done
done
cat $O/summary.txt
