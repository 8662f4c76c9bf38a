#!/bin/bash
# round-4 GPU job 42: the fair CPU baseline with and without gather / scatter transposes, on the GPU box's host
cd "$GRAFT_REPO_ROOT"
for g in 0 1; do
  touch oracle/moihgp_oracle.c
  echo "== ORC_FAST_GATHER=$g"
  FASTFLAGS=-DORC_FAST_GATHER=$g python bench.py --no-others --no-cold 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['cpu_baseline']; print(c['value'], c['ns_per_step_per_thread'], c['cycles_per_step_per_thread'], r['value']/c['value'])"
done
grep -m1 "model name" /proc/cpuinfo
