#!/bin/bash
# PMC passes (separate, --kernel-trace only) on the reverse-sweep GEMMs of tools/bwd_bench (cfg2 layer 2); gpurun_out/pmc_bwd/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_bwd
mkdir -p $O
rm -f $O/summary.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" FETCH_SIZE; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o p -- $R/tools/bin/bwd_bench 960 960 100000 > /dev/null 2> $O/p_$tag.err || { echo "pass $c failed"; tail -3 $O/p_$tag.err; }
  echo "== --pmc $c" >> $O/summary.txt
  python3 $R/tools/pmc_summary.py $O/p_$tag _f64_ >> $O/summary.txt
  rm -rf $O/p_$tag
done
cat $O/summary.txt
