#!/bin/bash
# round-4 GPU job 5: segment-major streams (parity + kernel rate), the learner loop after the Jacobi fix, host CPU share
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j5
O=gpurun_out/j5
echo "== host" | tee $O/progress.log
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)), os.cpu_count())"; lscpu | head -20) > $O/host.log 2>&1; cat $O/host.log | head -8
echo "== tiled parity" | tee -a $O/progress.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "segment_major or polar or unstable or stream_vs_golden" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -4 $O/tests.log
export MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_tuning.so
for dt in f32 f64; do
  echo "== kbench tiled $dt" | tee -a $O/progress.log
  timeout -k 10 200 python tools/kbench.py --dtype $dt --variants 10,0 --tiled --rounds 4 --per 10 > $O/kb_${dt}_res.log 2>&1; echo "rc=$?" >> $O/progress.log
  tail -3 $O/kb_${dt}_res.log
  timeout -k 10 200 python tools/kbench.py --dtype $dt --variants 10,0 --tiled --rounds 4 --per 10 --rotate 5 > $O/kb_${dt}_rot.log 2>&1; echo "rc=$?" >> $O/progress.log
  tail -3 $O/kb_${dt}_rot.log
done
timeout -k 10 200 python tools/kbench.py --dtype f32 --L 16384 --variants 0 --tiled --rounds 3 --per 6 > $O/kb_f32_L16k.log 2>&1; tail -2 $O/kb_f32_L16k.log
unset MOIHGP_LIB
echo "== learner loop" | tee -a $O/progress.log
g++ -std=c++14 -O2 -I include tools/cxx/learner_bench.cpp -o build/learner_bench -L multioutputihgp_amd/lib -lmoihgp -Wl,-rpath,$PWD/multioutputihgp_amd/lib -Wl,-rpath,/opt/rocm/lib
LEARNER_BENCH_PHASES=1 timeout -k 10 300 build/learner_bench 4096 4096 16 10 1 1 > $O/loop.json 2> $O/loop_phases.log
cat $O/loop.json; head -8 $O/loop_phases.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_loop -o out -- build/learner_bench 4096 4096 16 4 1 1 > $O/loop_prof.json 2> $O/loop_prof.err
f=$(find $O/prof_loop -name "*kernel_stats.csv" | head -1); python - "$f" > $O/loop_kernel_stats.txt <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:28]:
    n = re.sub(r'moihgp::\(anonymous namespace\)::', '', r["Name"])
    print(f'{n[:90]:90s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
cat $O/loop_kernel_stats.txt | head -16
rm -rf $O/prof_loop
echo "== done" | tee -a $O/progress.log
