#!/bin/bash
# round-4 GPU job 15: kernel trace of the imputation path (d = 12 fp64, 1 % and 0.01 % of the ticks missing)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/j15
O=$PWD/gpurun_out/j15
export FILTERNAN_IMPUTE=1 FILTERNAN_DTYPE=float64
for f in 0.01 0.0001; do
  export FILTERNAN_FRACS=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$f -o t -- python3 tools/filternan.py Matern52x4 > $O/run_$f.log 2>&1
  python3 - <<PY
import csv,glob
for fn in glob.glob("$O/prof_$f/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(fn)))
    for r in rows[:14]:
        print("$f", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
echo "== done"
