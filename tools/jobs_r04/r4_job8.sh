#!/bin/bash
# round-4 GPU job 8: does dropping SLP vectorisation from the sweep (fewer v_mov / v_readlane around v_pk_fma) pay in either layout?
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j8
O=gpurun_out/j8
for lib in tuning tuning_noslp; do
  export MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_$lib.so
  for dt in f32 f64; do
    echo "== kbench $lib $dt" | tee -a $O/progress.log
    timeout -k 10 200 python tools/kbench.py --dtype $dt --variants 0 --tiled --rounds 6 --per 10 > $O/kb_${lib}_${dt}_res.log 2>&1; tail -2 $O/kb_${lib}_${dt}_res.log
    timeout -k 10 200 python tools/kbench.py --dtype $dt --variants 0 --tiled --rounds 6 --per 10 --rotate 5 > $O/kb_${lib}_${dt}_rot.log 2>&1; tail -2 $O/kb_${lib}_${dt}_rot.log
  done
done
echo "== done" | tee -a $O/progress.log
