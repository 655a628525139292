#!/bin/bash
# PMC passes on the specialised fused forward (tools/bin/chain_spec_bench_nb2 512): MFMA busy, waits, LDS, instruction mix
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-chain_spec_bench_nb2}
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/pmc_s
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_s -o p -- $R/tools/bin/$B 512 > /dev/null 2> /tmp/pmc_s.err || { tail -5 /tmp/pmc_s.err; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/pmc_s si_spec_fused
done
