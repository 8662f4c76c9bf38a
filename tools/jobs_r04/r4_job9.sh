#!/bin/bash
# round-4 GPU job 9: ring probes on segment-major streams; GEMM counters; quick tests of the touched entries
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j9
O=gpurun_out/j9
export MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_tuning.so
for dt in f32 f64; do
  echo "== kbench tiled ring probes $dt" | tee -a $O/progress.log
  timeout -k 10 300 python tools/kbench.py --dtype $dt --variants 0,21,22,23,24 --tiled --rounds 4 --per 10 > $O/kb_${dt}_res.log 2>&1; tail -10 $O/kb_${dt}_res.log
  timeout -k 10 300 python tools/kbench.py --dtype $dt --variants 0,21,22,23,24 --tiled --rounds 4 --per 10 --rotate 5 > $O/kb_${dt}_rot.log 2>&1; tail -10 $O/kb_${dt}_rot.log
done
unset MOIHGP_LIB
echo "== gemm probe" | tee -a $O/progress.log
python tools/gemm_probe.py --dtype f32 > $O/gemm_f32.log 2>&1; cat $O/gemm_f32.log | tail -1
python tools/gemm_probe.py --dtype f32 --T 4096 >> $O/gemm_f32.log 2>&1; tail -1 $O/gemm_f32.log
python tools/gemm_probe.py --dtype f64 > $O/gemm_f64.log 2>&1; tail -1 $O/gemm_f64.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o out -- python3 tools/gemm_probe.py --dtype f32 --reps 3 > /dev/null 2> $O/g$i.err || echo "group $i failed"
done
python3 - $O <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_mfma_kernel<float" in r["Kernel_Name"]:
            acc[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
per = collections.defaultdict(list)
for (name, disp), v in acc.items():
    per[name].append(sum(v))
res = {k: sum(v) / len(v) for k, v in per.items()}
json.dump(res, open(f"{out}/gemm_f32_sq.json", "w"), indent=1); print(json.dumps(res))
PY
rm -rf $O/g?
echo "== tests" | tee -a $O/progress.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "segment_major or device_vector or sharded or stream_vs_golden" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -3 $O/tests.log
echo "== done" | tee -a $O/progress.log
