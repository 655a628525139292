#!/bin/bash
# Final validation of a round on the GPU box: the whole GPU suite, smoke(), the default bench line, the small-model and toy-finish
# timings.  Everything into gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python tools/small_model_steps.py > $O/small_model_steps.log 2>&1
python tools/toy_finish_time.py > $O/toy_finish.log 2>&1; tail -6 $O/toy_finish.log
python tools/host_eig_time.py > $O/host_eig.log 2>&1
