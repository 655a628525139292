#!/bin/bash
# Development: sample rocm-smi (clocks, package power) every <period_s> while a command runs.  usage: smi_during.sh <period_s> <cmd...>
period=$1; shift
"$@" > gpurun_out/smi_during_cmd.log 2>&1 &
pid=$!
t=0
while kill -0 $pid 2>/dev/null; do
  sleep $period
  t=$((t + 1))
  echo "sample ${t}: $(rocm-smi --showclocks --showpower -d 0 2>&1 | grep -E 'sclk|Power \(W\)' | tr -s '\t ' ' ' | sed 's/GPU\[0\] : //' | paste -sd' ' -)"
done
wait $pid
tail -n 1 gpurun_out/smi_during_cmd.log | cut -c1-260
