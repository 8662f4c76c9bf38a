#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/m1
( while true; do sleep 60; echo "[measure running] $(date +%T) $(ls gpurun_out/m1 | wc -l) files"; done ) &
HB=$!
bash tools/measure_r4.sh gpurun_out/m1 v1 2>&1 | tail -40
kill $HB 2>/dev/null
ls gpurun_out/m1 | head -80
