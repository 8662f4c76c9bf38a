#!/bin/bash
# One measurement pass of the round on the GPU box: every bench configuration, the rocprofv3 kernel trace of the default
# configuration and of c5, the PMC traffic of c3 / c5 / c2d6, and the per-row table.  usage: tools/measure_all.sh <outdir> <tag>
set -e
out=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for c in c3 c2 c3f64 c2d6 c3d6 c3d6f64 c5 c4 c1 c3learn c3grad c5grad c3loop; do
  python3 bench.py --config $c > "$out/${tag}_${c}_bench.json" 2> "$out/${tag}_${c}_bench.err"
  echo "bench $c done"
done
for c in c3 c5 c2 c2d6 c3d6 c3d6f64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$c/kt" -o out -- python3 bench.py --config $c --steps 200 --warmup 150 --no-cpu --no-cold > "$out/${tag}_${c}_bench_under_rocprof.json" 2> "$out/prof_$c.err"
  python3 profiles/summarize.py "$out/prof_$c" "$out" "${tag}_$c" $([ $c = c3 ] && echo filter_scan || echo filter_x) > /dev/null
  echo "rocprof $c done"
done
for c in c3 c5 c2 c2d6 c3d6 c3d6f64; do
  bash tools/pmc_traffic.sh $c "$out/pmc_$c" > "$out/${tag}_${c}_pmc.json"
  echo "pmc $c done"
done
python3 tools/bench_rows.py > "$out/rows_${tag}.jsonl" 2> "$out/rows_${tag}.err"
echo "rows done"
