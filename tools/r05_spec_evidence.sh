#!/bin/bash
# round 5, second half: evidence for the kernels specialised at run time (csrc/chain_spec.inc through hiprtc)
#   gpurun_out/r05_small_model_steps.log       tools/small_model_steps.py (README toy + nn_example MLP, 1 .. 512 chains), default mode
#   gpurun_out/r05_nn_modes.log                modes 0 (per layer), 3 (generic fused + generic loop), 2, 1 (specialised) side by side
#   gpurun_out/r05_spec_harness.log            tools/chain_spec_bench.hip: bits against the generic kernels, time per launch / per
#                                              transition, cycles per phase of the loop (workgroup 0)
#   gpurun_out/r05_spec_kernel_stats.txt       rocprofv3 --kernel-trace --stats: 1 chain x 20000 transitions, 512 chains x 100
#   gpurun_out/r05_pmc_spec.txt                PMC passes on the harness (si_spec_stack_kernel, 512 chains): MFMA busy, waits, LDS
set -o pipefail
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python tools/small_model_steps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_small_model_steps.log || exit 1
timeout -k 10 400 python tools/nn_example_modes.py 0321 1,4,8,16,64,512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_nn_modes.log || exit 1
{
  for nch in 512 64 8; do timeout -k 5 60 tools/bin/chain_spec_bench_nb2 $nch | grep -v "^  generic  *[0-9.]* us per launch = .*" || exit 1; done
  timeout -k 5 60 tools/bin/chain_spec_bench_nb1 2 2000 1 | grep "loop\|per transition" || exit 1
  timeout -k 5 60 tools/bin/chain_spec_bench_nb1_st 2 2000 1 | grep "cycles per" || exit 1
  timeout -k 5 60 tools/bin/chain_spec_bench_nb1 2 500 4 | grep "loop\|per transition" || exit 1
} 2>&1 | tee gpurun_out/r05_spec_harness.log
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r05_spec_kernel_stats.txt
for spec in "1 1" "1 512"; do
  set -- $spec
  rm -rf /tmp/prof_sp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sp -- python3 $R/tools/nn_example_modes.py $1 $2 > /dev/null 2>&1 || exit 1
  echo "== rocprofv3 --kernel-trace --stats: tools/nn_example_modes.py $1 $2 (mode, chains)" >> $R/gpurun_out/r05_spec_kernel_stats.txt
  python3 $R/tools/kstats.py /tmp/prof_sp 8 >> $R/gpurun_out/r05_spec_kernel_stats.txt
done
cat $R/gpurun_out/r05_spec_kernel_stats.txt
: > $R/gpurun_out/r05_pmc_spec.txt
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_s
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_s -o p -- $R/tools/bin/chain_spec_bench_nb2 512 > /dev/null 2> /tmp/pmc_s.err || { echo "rocprofv3 --pmc $c failed" >> $R/gpurun_out/r05_pmc_spec.txt; continue; }
  python3 $R/tools/pmc_summary.py /tmp/pmc_s si_spec_stack >> $R/gpurun_out/r05_pmc_spec.txt
done
cat $R/gpurun_out/r05_pmc_spec.txt
