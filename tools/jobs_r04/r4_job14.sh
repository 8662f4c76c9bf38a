#!/bin/bash
# round-4 GPU job 14: missing ticks of the stacked many-latent sweep by imputation: parity, then timing against the second pass
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j14
O=gpurun_out/j14
echo "== tests" | tee $O/progress.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "imputation or stacked_filter_gaps or stacked_missing or stacked_ragged or stacked_segment" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -6 $O/tests.log
echo "== timing" | tee -a $O/progress.log
FILTERNAN_IMPUTE=0 timeout -k 10 300 python tools/filternan.py Matern52x4 Matern52x3 > $O/filternan_second_pass.log 2>&1; cat $O/filternan_second_pass.log | tail -16
FILTERNAN_IMPUTE=1 timeout -k 10 300 python tools/filternan.py Matern52x4 Matern52x3 Matern52x2 > $O/filternan_imputation.log 2>&1; cat $O/filternan_imputation.log | tail -24
echo "== done" | tee -a $O/progress.log
