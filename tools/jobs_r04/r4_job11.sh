#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j11
timeout -k 10 300 python tools/dbg_tiled.py > gpurun_out/j11/dbg.log 2>&1; cat gpurun_out/j11/dbg.log | tail -30
