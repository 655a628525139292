#!/bin/bash
# second half of tools/r05_gram_conflicts.sh: k16 (round 4: the compiler pairs the operand reads into ds_read2st64_b64) against k0 (shipped: every operand read a hand-written ds_read_b64, the reduction tiles at pitch 17), bit pattern of G
# (hash), event time over 3 x 5 launches, one PMC pass each
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_gram_conflicts
mkdir -p $O
LOG=$O/summary_ab.txt
: > $LOG
cd /tmp && export TMPDIR=/tmp
for shape in "1047361 100" "6400000 128" "5000000 200" "1047361 104" "1047361 144" "2000000 176" "500000 37"; do
  set -- $shape
  for k in 16 0; do
    echo "== K=$2 N=$1 variant k$k" | tee -a $LOG
    SI_BENCH_REPS=3 timeout -k 10 120 $R/tools/bin/gram_bench_k$k $1 $2 20 2>&1 | grep "gram+reduce\|CHECK\|hash" | tail -3 | tee -a $LOG || exit 1
    rm -rf /tmp/pmc_g
    SI_BENCH_REPS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_g -o p -- $R/tools/bin/gram_bench_k$k $1 $2 20 > /dev/null 2> $O/pmc_$2_k$k.err || { tail -5 $O/pmc_$2_k$k.err; exit 1; }
    python3 $R/tools/pmc_summary.py /tmp/pmc_g gram_glds | tee -a $LOG
  done
done
