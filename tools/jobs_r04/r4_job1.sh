#!/bin/bash
# round-4 GPU job 1: parity of the LDS-DMA filter kernel, ring / occupancy probes (resident + rotating streams), learner-loop polar trace
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j1
O=gpurun_out/j1
echo "== parity (shipped lib: DMA kernel default)" | tee $O/progress.log
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "stream or full_size or filter or gradient_tables or unstable or nll_total or separate_start" > $O/parity.log 2>&1
echo "parity rc=$?" | tee -a $O/progress.log
tail -3 $O/parity.log
export MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_tuning.so
for dt in f32 f64; do
  if [ $dt = f32 ]; then V=10,20,21,22,23,24; else V=10,20,22,23,24,25,26,27; fi
  echo "== kbench $dt resident" | tee -a $O/progress.log
  timeout -k 10 200 python tools/kbench.py --dtype $dt --variants $V --rounds 4 --per 10 > $O/kb_${dt}_res.log 2>&1; echo "rc=$?" >> $O/progress.log
  cat $O/kb_${dt}_res.log | tail -9
  echo "== kbench $dt rotating" | tee -a $O/progress.log
  timeout -k 10 200 python tools/kbench.py --dtype $dt --variants $V --rounds 4 --per 10 --rotate 5 > $O/kb_${dt}_rot.log 2>&1; echo "rc=$?" >> $O/progress.log
  cat $O/kb_${dt}_rot.log | tail -9
done
echo "== kbench f32 L=32768 T=10000 (c4-like working set) " | tee -a $O/progress.log
timeout -k 10 200 python tools/kbench.py --dtype f32 --L 16384 --variants 10,20,22,23 --rounds 3 --per 6 > $O/kb_f32_L16k.log 2>&1; echo "rc=$?" >> $O/progress.log
tail -5 $O/kb_f32_L16k.log
unset MOIHGP_LIB
echo "== learner loop with polar trace" | tee -a $O/progress.log
g++ -std=c++14 -O2 -I include tools/cxx/learner_bench.cpp -o build/learner_bench -L multioutputihgp_amd/lib -lmoihgp -Wl,-rpath,$PWD/multioutputihgp_amd/lib -Wl,-rpath,/opt/rocm/lib
MOIHGP_POLAR_TRACE=1 MOIHGP_POLAR_DUMP=/tmp/gram.bin MOIHGP_POLAR_DUMP_CALL=45 LEARNER_BENCH_PHASES=1 timeout -k 10 300 build/learner_bench 4096 4096 16 10 1 1 > $O/loop.json 2> $O/loop_trace.log
echo "loop rc=$?" | tee -a $O/progress.log
cat $O/loop.json
timeout -k 10 200 python tools/polar_spectrum.py /tmp/gram.bin $O/sigma_call45.npy > $O/spectrum.log 2>&1
cat $O/spectrum.log
echo "== done" | tee -a $O/progress.log
