#!/bin/bash
# Development: sample clocks / power with rocm-smi while the Gram harness runs in one of its knob modes
# (SI_GRAM_SPEC_DBG: 0 = MFMA + HBM, 2 = MFMA only, 4 = HBM only).  usage: smi_sample.sh <dbg> <N> <K>
d=$1; N=$2; K=$3
SI_BENCH_REPS=1500 SI_GRAM_SPEC=1 SI_GRAM_SPEC_DBG=$d timeout -k 10 120 tools/bin/gram_bench_ts $N $K 20 > gpurun_out/smi_run_$d.log 2>&1 &
pid=$!
sleep 2
for i in 1 2 3; do
  rocm-smi --showclocks --showpower --showuse -d 0 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|GPU use|busy" | sed "s/^/dbg=$d: /"
  sleep 0.7
done
wait $pid
grep "gram+reduce" gpurun_out/smi_run_$d.log | tail -n 1
