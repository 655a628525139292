#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repository root):
#   kernel-trace statistics of bench.py and of the cfg4 CNN bench, PMC passes for the dominant kernels (separate passes,
#   --kernel-trace only, as gpurun requires), the bench lines of every mode.  Everything lands in gpurun_out/${ROUND:-r4}prof/;
#   the summaries worth judging are copied into profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${ROUND:-r4}prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# ONLY_BENCH=1: step [4] alone (the bench lines read the PMC json that steps [1]-[3] of an earlier call produced); SKIP_BENCH=1: the rest
if [ -z "${ONLY_BENCH:-}" ]; then
echo "[1] bench.py under rocprofv3 --kernel-trace --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -o b -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 $R/tools/kstats.py $O/bench_kt 25 > $O/bench_kernel_stats.txt
cp $(find $O/bench_kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo "[2] PMC passes (dense kernels, Gram) on the short target tools/f32_after_f64.py: the cfg2 construct, an fp64 and an fp32 chain"
# (NOT on bench.py: rocprofv3's counter collection dies -- SIGSEGV in one of its threads, or "AQL packet is malformed" -- once a
#  process has made ~16 K dispatches: 14 564 pass, 18 764 do not, on the plain fp64 chain alone (tools/pmc_bisect*.sh, round 4);
#  the default bench.py makes ~25 K.  Same kernels, same shapes, same launch parameters.)
rm -f $O/pmc_dense_summary.txt $O/pmc_dense_f32_summary.txt $O/pmc_gram_summary.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_')
  echo "    --pmc $c"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -o p -- python3 $R/tools/f32_after_f64.py f64 > /dev/null 2> $O/pmc_$tag.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "    pass failed (rc $rc): stopping"; exit 1; fi
  echo "== --pmc $c" >> $O/pmc_dense_summary.txt
  python3 $R/tools/pmc_summary.py $O/pmc_$tag dense_f64 >> $O/pmc_dense_summary.txt
  echo "== --pmc $c" >> $O/pmc_dense_f32_summary.txt
  python3 $R/tools/pmc_summary.py $O/pmc_$tag dense_f32 >> $O/pmc_dense_f32_summary.txt
  echo "== --pmc $c" >> $O/pmc_gram_summary.txt
  python3 $R/tools/pmc_summary.py $O/pmc_$tag gram_ >> $O/pmc_gram_summary.txt
  rm -rf $O/pmc_$tag
done
echo "[3] cfg4 CNN"
python3 $R/tools/cfg4_cnn_bench.py 4096 > $O/cfg4_cnn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cnn_kt -o c -- python3 $R/tools/cfg4_cnn_bench.py 4096 > /dev/null 2>&1
python3 $R/tools/kstats.py $O/cnn_kt 20 > $O/cfg4_cnn_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cnng_kt -o g -- python3 $R/tools/cfg4_cnn_grad_profile.py > /dev/null 2>&1
python3 $R/tools/kstats.py $O/cnng_kt 20 > $O/cfg4_cnn_grad_kernel_stats.txt
# transitions alone at B = 4096 (the mixed run above also holds the B = 1024 training launches): per-layer times
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cnns_kt -o s -- python3 $R/tools/cfg4_cnn_sample_profile.py > /dev/null 2>&1
python3 $R/tools/kstats.py $O/cnns_kt 12 > $O/cfg4_cnn_sample_kernel_stats.txt
python3 $R/tools/per_layer.py $O/cnns_kt 14 15 >> $O/cfg4_cnn_sample_kernel_stats.txt
rm -rf $O/cnn_kt $O/cnng_kt $O/cnns_kt $O/bench_kt
fi
if [ -n "${SKIP_BENCH:-}" ]; then echo done; exit 0; fi
echo "[4] bench lines"
cd $R
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 bench.py --steps 1000 --warmup 10 --no-cpu-baseline > $O/bench_itr1000.json 2>> $O/bench_n1.err
for m in "construct-sharded --config cfg4" "construct-sharded --config cfg5" "data-sharded"; do
  python3 tools/bench_rehearsal.py --world1-rccl --mode $m >> $O/bench_modes.json 2>> $O/bench_modes.err
done
python3 tools/small_model_steps.py > $O/small_model_steps.log 2>&1
echo done
