#!/bin/bash
# round 5 evidence for the fused Dense-chain forward / the persistent grid loop (csrc/kernels_chain_grid.hip):
#   gpurun_out/r05_small_model_steps.log        tools/small_model_steps.py (README toy + nn_example MLP, 1 .. 512 chains)
#   gpurun_out/r05_nn_modes.log                 the three sampler paths side by side
#   gpurun_out/r05_chain_grid_knockouts.log     harness: compile-time knock-outs + phase stamps of workgroup 0
#   gpurun_out/r05_chain_grid_kernel_stats.txt  rocprofv3 --kernel-trace --stats of the 512-chain stacked run and the 1-chain loop
#   gpurun_out/r05_pmc_chain_grid.txt           PMC passes (MFMA busy, instruction mix, FETCH / WRITE) of the same kernels
set -o pipefail
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python tools/small_model_steps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_small_model_steps.log || exit 1
timeout -k 10 300 python tools/nn_example_modes.py 021 1,4,8,16,64,512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_nn_modes.log || exit 1
{
  echo "# tools/chain_grid_bench.hip on docs/src/nn_example.md's model (N = 15801, B = 1000, M = 20); 512 stacked chains / the 1-chain loop"
  for bin in cgb_clean cgb_k1 cgb_k2 cgb_k4 cgb_k8 cgb_k15; do
    for spec in "0 1" "0 2" "1 1"; do
      set -- $spec
      echo "== $bin: $( [ $1 = 1 ] && echo 'a wave per tile' || echo 'a workgroup per tile'), NB $2"
      timeout -k 5 60 tools/bin/$bin 512 $2 400 1 $1 | grep -E "KNOB|us per" | sed -n '1p;4p;7p' | cut -c1-200 || exit 1
    done
  done
  echo "== phase stamps (shader cycles per launch / per transition, workgroup 0)"
  timeout -k 5 60 tools/bin/cgb_stamps 512 2 1000 1 0 | grep -E "fused forward|grid loop|us per|workgroup" | sed -n '1p;6,7p;8p;13,14p' | cut -c1-520 || exit 1
} 2>&1 | tee gpurun_out/r05_chain_grid_knockouts.log
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r05_chain_grid_kernel_stats.txt
for spec in "1 1" "2 512"; do
  set -- $spec
  rm -rf /tmp/prof_cg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cg -- python3 $R/tools/nn_example_modes.py $1 $2 > /dev/null 2>&1 || exit 1
  echo "== rocprofv3 --kernel-trace --stats: tools/nn_example_modes.py $1 $2 (mode, chains)" >> $R/gpurun_out/r05_chain_grid_kernel_stats.txt
  python3 $R/tools/kstats.py /tmp/prof_cg 8 >> $R/gpurun_out/r05_chain_grid_kernel_stats.txt
done
cat $R/gpurun_out/r05_chain_grid_kernel_stats.txt
: > $R/gpurun_out/r05_pmc_chain_grid.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_cg
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_cg -- python3 $R/tools/nn_example_modes.py 2 512 > /dev/null 2>&1 || { echo "rocprofv3 --pmc $set failed" >> $R/gpurun_out/r05_pmc_chain_grid.txt; continue; }
  python3 - >> $R/gpurun_out/r05_pmc_chain_grid.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_cg/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'chain_fused' not in k and 'reconstruct' not in k: continue
    print(k, 'launches', len(next(iter(d.values()))), ' '.join('%s=%.5g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
done
cat $R/gpurun_out/r05_pmc_chain_grid.txt
