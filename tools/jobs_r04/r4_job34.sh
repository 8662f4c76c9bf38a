#!/bin/bash
# round-4 GPU job 34: timeline of an fp32 sweep with gaps, imputation forced (Matern32x2)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/j34
O=$PWD/gpurun_out/j34
export FILTERNAN_FRACS=0.01 FILTERNAN_DTYPE=float32 FILTERNAN_IMPUTE=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 tools/filternan.py Matern32x2 > $O/run.log 2>&1
python3 - <<PY
import csv,glob
fn=glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[-24]["Start_Timestamp"])
for r in rows[-24:]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-t0)/1e3:9.1f} q{r.get("Queue_Id","?")} {r["Kernel_Name"][:86]}')
PY
