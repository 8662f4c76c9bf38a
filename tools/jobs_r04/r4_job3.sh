#!/bin/bash
# round-4 GPU job 3: the deflated polar factor -- parity on chosen spectra, the learner loop, configs[2] at full size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j3
O=gpurun_out/j3
echo "== polar tests" | tee $O/progress.log
MOIHGP_POLAR_TRACE=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "polar" > $O/polar_tests.log 2>&1; echo "polar tests rc=$?" | tee -a $O/progress.log
tail -15 $O/polar_tests.log
echo "== learner loop" | tee -a $O/progress.log
g++ -std=c++14 -O2 -I include tools/cxx/learner_bench.cpp -o build/learner_bench -L multioutputihgp_amd/lib -lmoihgp -Wl,-rpath,$PWD/multioutputihgp_amd/lib -Wl,-rpath,/opt/rocm/lib
MOIHGP_POLAR_TRACE=1 LEARNER_BENCH_PHASES=1 timeout -k 10 300 build/learner_bench 4096 4096 16 10 1 1 > $O/loop.json 2> $O/loop_trace.log
echo "loop rc=$?" | tee -a $O/progress.log
cat $O/loop.json; grep -v "^polar:" $O/loop_trace.log | head -20; grep "^polar:" $O/loop_trace.log | tail -12
echo "== learner loop, deflation off" | tee -a $O/progress.log
MOIHGP_POLAR_DEFLATE=0 timeout -k 10 300 build/learner_bench 4096 4096 16 10 1 1 > $O/loop_nodeflate.json 2> /dev/null
cat $O/loop_nodeflate.json
echo "== configs[2] full size + cxx learners + dev entries" | tee -a $O/progress.log
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_cxx_learner.py -x -q -k "c3_learning or dev_entries or cxx or learner" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
echo "== done" | tee -a $O/progress.log
