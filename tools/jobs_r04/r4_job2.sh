#!/bin/bash
# round-4 GPU job 2: access-pattern copy microbenchmark, new tests (stream-ordered entries, C++ learners), no-SLP build of the sweep
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j2
O=gpurun_out/j2
echo "== rows_copy" | tee $O/progress.log
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/rows_copy.hip -o /tmp/rows_copy > /dev/null 2>&1 && timeout -k 10 120 /tmp/rows_copy > $O/rows_copy.log 2>&1; echo "rc=$?" >> $O/progress.log
cat $O/rows_copy.log
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/copy_bench.hip -o /tmp/copy_bench > /dev/null 2>&1 && timeout -k 10 120 /tmp/copy_bench > $O/copy_bench.log 2>&1; echo "rc=$?" >> $O/progress.log
cat $O/copy_bench.log
echo "== tests" | tee -a $O/progress.log
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_cxx_learner.py -x -q -k "dev_entries or cxx or learner" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
for lib in tuning tuning_noslp; do
  export MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_$lib.so
  for dt in f32 f64; do
    if [ $dt = f32 ]; then V=10,20; else V=10,24; fi
    echo "== kbench $lib $dt" | tee -a $O/progress.log
    timeout -k 10 200 python tools/kbench.py --dtype $dt --variants $V --rounds 4 --per 10 > $O/kb_${lib}_${dt}_res.log 2>&1; echo "rc=$?" >> $O/progress.log
    tail -2 $O/kb_${lib}_${dt}_res.log
    timeout -k 10 200 python tools/kbench.py --dtype $dt --variants $V --rounds 4 --per 10 --rotate 5 > $O/kb_${lib}_${dt}_rot.log 2>&1; echo "rc=$?" >> $O/progress.log
    tail -2 $O/kb_${lib}_${dt}_rot.log
  done
done
unset MOIHGP_LIB
echo "== done" | tee -a $O/progress.log
