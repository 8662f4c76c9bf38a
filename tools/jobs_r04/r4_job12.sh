#!/bin/bash
# round-4 GPU job 12: the whole GPU suite + smoke, then the learner loop with the warm-started deflation
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j12
O=gpurun_out/j12
echo "== gpu suite" | tee $O/progress.log
( while true; do sleep 60; echo "[suite running] $(date +%T) $(tail -c 200 $O/suite.log 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
HB=$!
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $O/suite.log 2>&1; rc=$?
kill $HB 2>/dev/null
echo "suite rc=$rc" | tee -a $O/progress.log; tail -5 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/progress.log; tail -1 $O/smoke.log
echo "== learner loop (warm start on: the C++ device learner asks for it)" | tee -a $O/progress.log
g++ -std=c++14 -O2 -I include tools/cxx/learner_bench.cpp -o build/learner_bench -L multioutputihgp_amd/lib -lmoihgp -Wl,-rpath,$PWD/multioutputihgp_amd/lib -Wl,-rpath,/opt/rocm/lib
MOIHGP_POLAR_TRACE=1 LEARNER_BENCH_PHASES=1 timeout -k 10 300 build/learner_bench 4096 4096 16 10 1 1 > $O/loop.json 2> $O/loop_trace.log
cat $O/loop.json; grep -v "^polar:" $O/loop_trace.log | head -6; grep "^polar:" $O/loop_trace.log | tail -6
echo "== done" | tee -a $O/progress.log
