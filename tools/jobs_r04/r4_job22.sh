#!/bin/bash
# round-4 GPU job 21: fused imputation kernel: parity, timing against the second pass, kernel trace
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/j22
O=$PWD/gpurun_out/j22
echo "== tests" | tee $O/progress.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "imputation or stacked_filter_gaps or stacked_missing or stacked_ragged or stacked_segment" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -6 $O/tests.log
echo "== timing" | tee -a $O/progress.log
export FILTERNAN_FRACS=0.0,0.0001,0.0003,0.001,0.01,0.05
for m in -1; do
  echo "-- filter_impute=$m" | tee -a $O/progress.log
  FILTERNAN_IMPUTE=$m timeout -k 10 300 python tools/filternan.py Matern52x4 Matern52x3 Matern52x2 > $O/filternan_$m.log 2>&1; cat $O/filternan_$m.log
done
echo "-- filter_impute=1 d=6" | tee -a $O/progress.log
FILTERNAN_IMPUTE=1 timeout -k 10 300 python tools/filternan.py Matern52x2 > $O/filternan_d6_1.log 2>&1; cat $O/filternan_d6_1.log
echo "== trace" | tee -a $O/progress.log
export FILTERNAN_IMPUTE=1 FILTERNAN_DTYPE=float64
for f in 0.01 0.0001; do
  FILTERNAN_FRACS=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$f -o t -- python3 tools/filternan.py Matern52x4 > $O/run_$f.log 2>&1
  python3 - <<PY
import csv,glob
for fn in glob.glob("$O/prof_$f/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:6]:
        print("$f", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
echo "== done" | tee -a $O/progress.log
