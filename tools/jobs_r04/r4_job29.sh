#!/bin/bash
# round-4 GPU job 29: timeline of the fp32 sweep with its fp64 side sweep
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/j29
O=$PWD/gpurun_out/j29
export FILTERNAN_FRACS=0.0 FILTERNAN_DTYPE=float32
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 tools/filternan.py Matern32x2 > $O/run.log 2>&1
python3 - <<PY
import csv,glob
fn=glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 40 kernels
t0=int(rows[-40]["Start_Timestamp"])
for r in rows[-40:]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-t0)/1e3:9.1f} q{r.get("Queue_Id","?")} {r["Kernel_Name"][:70]}')
PY
