#!/bin/bash
# Knock-outs of the panel kernel (csrc/kernels_gemm_panel.hip, SI_PANEL_KNOB: 1 = no stores, 2 = no W DMA in the loop, 4 = no
# barrier; timing only) by the kernel trace: durations of the kernel itself, comparable with the bench's trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for k in 0 1 3 7; do
  rm -rf /tmp/pk$k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk$k -o p -- $R/tools/bin/panel_bench_k$k 960 128 100000 1 512 > /dev/null 2>&1 || exit 1
  echo "== SI_PANEL_KNOB=$k"
  python3 $R/tools/kstats.py /tmp/pk$k 4 | grep "panel\|dense_f64_kernel" | cut -c1-60,112-200
done
