#!/bin/bash
# round-4 GPU job 4: polar tests again, kernel trace of the learner loop, the new bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j4
O=gpurun_out/j4
echo "== polar tests" | tee $O/progress.log
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "polar" > $O/polar_tests.log 2>&1; echo "polar tests rc=$?" | tee -a $O/progress.log
tail -4 $O/polar_tests.log
echo "== learner loop kernel trace" | tee -a $O/progress.log
g++ -std=c++14 -O2 -I include tools/cxx/learner_bench.cpp -o build/learner_bench -L multioutputihgp_amd/lib -lmoihgp -Wl,-rpath,$PWD/multioutputihgp_amd/lib -Wl,-rpath,/opt/rocm/lib
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_loop -o out -- build/learner_bench 4096 4096 16 4 1 1 > $O/loop_prof.json 2> $O/loop_prof.err
echo "loop prof rc=$?" | tee -a $O/progress.log
f=$(find $O/prof_loop -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-200 > $O/loop_kernel_stats_top.csv; cat $O/loop_kernel_stats_top.csv | cut -c1-160
rm -rf $O/prof_loop
echo "== bench default" | tee -a $O/progress.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/progress.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/j4/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_with_event_pairs")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","frac_resident","kernel_ms","kernel_ms_resident","frac_is","traffic")})
print(d.get("cpu_baseline"))
for k,v in d.get("other_configs",{}).items():
    print(k, {q:v.get(q) for q in ("ms_per_step","kernel_ms","frac","frac_resident","project_ms","unproject_ms","filter_ms","project_frac_of_mfma_peak","update_ms","window_eval_ms","error")})
PY
echo "== done" | tee -a $O/progress.log
