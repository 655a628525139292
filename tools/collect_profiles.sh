#!/bin/bash
# Copies what tools/profile_round.sh and tools/final_validation.sh left under gpurun_out/ into profiles/ (tracked), named per round,
# and regenerates the two PMC records bench.py reads its `traffic` figures from.  usage: ROUND=r4 TAG=r04 bash tools/collect_profiles.sh
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/${ROUND:-r4}prof
T=${TAG:-r04}
python3 tools/make_pmc_json.py $P/pmc_dense_summary.txt profiles/${T}_pmc_dense_main.json > /dev/null
python3 tools/make_pmc_json.py $P/pmc_dense_f32_summary.txt profiles/${T}_pmc_dense_f32_main.json f32 > /dev/null
cp $P/pmc_dense_summary.txt profiles/${T}_pmc_dense_summary.txt
cp $P/pmc_dense_f32_summary.txt profiles/${T}_pmc_dense_f32_summary.txt
cp $P/pmc_gram_summary.txt profiles/${T}_pmc_gram_summary.txt
cp $P/bench_kernel_stats.csv profiles/${T}_rocprofv3_kernel_stats.csv
cp $P/bench_kernel_stats.txt profiles/${T}_rocprofv3_kernel_stats.txt
cp $P/bench_under_rocprof.json profiles/${T}_bench_under_rocprof.json
cp $P/bench_itr1000.json profiles/${T}_bench_itr1000.json
cp $P/bench_modes.json profiles/${T}_bench_modes_n1.jsonl
cp $P/cfg4_cnn.log profiles/${T}_cfg4_cnn.log
for f in cfg4_cnn_kernel_stats cfg4_cnn_sample_kernel_stats cfg4_cnn_grad_kernel_stats; do cp $P/$f.txt profiles/${T}_$f.txt; done
if [ -f gpurun_out/final/bench_n1.json ]; then
  cp gpurun_out/final/bench_n1.json profiles/${T}_bench_n1.json
  cp gpurun_out/final/gpu_tests.log profiles/${T}_gpu_tests_final.log
fi
python3 - <<PY
import json
d = json.load(open("profiles/${T}_bench_n1.json"))
print("value %.1f (frac %.3f, traffic %s)  value_f32 %.1f (frac %.3f, traffic %s)  construct %.3f / %.3f ms  e2e %.0f ms  train step %.2f ms" % (
    d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["value_f32"], d["roofline_f32"]["frac"], d["roofline_f32"]["traffic"],
    d["construct_wall_ms"], d["construct_wall_ms_batched"], d["construct_end_to_end_ms"], d["next_rows"]["train_step_full_batch_ms"]))
PY
