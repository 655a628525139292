#!/bin/bash
# PMC passes (separate, --kernel-trace only) on the fp32 dense kernels of a short cfg2 chain; summary to gpurun_out/pmc_f32_summary.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_f32
mkdir -p $O
rm -f $O/summary.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o p -- python3 $R/tools/f32_chain.py f32 12 > /dev/null 2> $O/p_$tag.err
  echo "== --pmc $c" >> $O/summary.txt
  python3 $R/tools/pmc_summary.py $O/p_$tag dense_f32 >> $O/summary.txt
  rm -rf $O/p_$tag
done
cat $O/summary.txt
