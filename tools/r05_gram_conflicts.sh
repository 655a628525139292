#!/bin/bash
# where do gram_glds_kernel's SQ_LDS_BANK_CONFLICT cycles come from, and do they cost time?
# Compile-time variants of kernels_gram_wave.hip (SI_GW_KNOB; built by tools/r05_gram_conflicts_build.sh HERE, run THERE):
#   k0 shipped kernel (round 5: hand-written ds_read_b64; k16 = round 4, compiler-paired reads) | k1 no LDS-DMA staging at all | k2 operand reads of 64 consecutive doubles (conflict-free by construction)
#   k4 operand reads of the unswizzled image (16 columns on the same banks: the positive control) | k3 = k1 + k2
# For each: event-timed runs, then ONE rocprofv3 --pmc pass (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_gram_conflicts
mkdir -p $O
LOG=$O/summary.txt
: > $LOG
cd /tmp && export TMPDIR=/tmp
for shape in "1047361 100" "6400000 128" "5000000 200"; do
  set -- $shape
  for k in 0 1 2 3 4; do
    echo "== K=$2 N=$1 variant k$k" | tee -a $LOG
    SI_BENCH_REPS=3 timeout -k 10 120 $R/tools/bin/gram_bench_k$k $1 $2 20 2>&1 | grep "gram+reduce" | tail -1 | tee -a $LOG || exit 1
    rm -rf /tmp/pmc_g
    SI_BENCH_REPS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_g -o p -- $R/tools/bin/gram_bench_k$k $1 $2 20 > /dev/null 2> $O/pmc_$2_k$k.err || { tail -5 $O/pmc_$2_k$k.err; exit 1; }
    python3 $R/tools/pmc_summary.py /tmp/pmc_g gram_glds | tee -a $LOG
  done
done
