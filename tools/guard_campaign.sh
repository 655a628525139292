#!/bin/bash
# All random-shape sweeps (Dense, CNN, construction, API, end to end, communicator flows at world 1) over the development library with the guard-page allocator, both modes, one seed base.
# usage (on the GPU box, after `python subspaceinference.jl_amd/build.py --dev`): tools/guard_campaign.sh <seed base> [scale]
# Stops at the first failure and shows the tail of its log (gpurun_out/campaign_<tool>_<mode>_<seed>.log).
base=${1:-100}; scale=${2:-1}
mkdir -p gpurun_out
export SI_PROBE_DEV=1 PYTHONUNBUFFERED=1
i=0
for spec in "guard_fuzz.py $((200*scale))" "guard_fuzz_cnn.py $((120*scale))" "guard_fuzz_gram.py $((120*scale))" "guard_fuzz_api.py $((80*scale))" "guard_fuzz_e2e.py $((25*scale))" "guard_fuzz_comm.py $((40*scale))"; do
  set -- $spec
  for mode in end begin; do
    seed=$((base + i)); i=$((i + 1))
    log=gpurun_out/campaign_${1%.py}_${mode}_$seed.log
    SI_GUARD_ALLOC=$mode timeout -k 10 900 python tools/$1 $2 $seed > $log 2>&1 || { echo "FAILED: $1 $2 $seed ($mode)"; tail -12 $log | cut -c1-400; exit 1; }
    echo "$(tail -1 $log)   [$mode, seed $seed]"
  done
done
echo "campaign $base: all sweeps done"
