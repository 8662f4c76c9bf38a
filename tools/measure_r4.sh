#!/bin/bash
# One measurement pass of round 4 on the GPU box.  usage: tools/measure_r4.sh <outdir> <tag>
#   bench lines of every configuration; rocprofv3 kernel stats of the headline resident (--no-cold) AND cold (--rotate) and of the other
#   filter rows; PMC traffic (FETCH_SIZE / WRITE_SIZE, own passes) resident and cold; SQ counters of the headline.
set -e
out=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
python3 bench.py > "$out/${tag}_default_bench_all_rows.json" 2> "$out/${tag}_default_bench.err"; echo "bench default done"
for c in c4 c1 c3grad c5grad c3learn c3loop; do
  python3 bench.py --config $c > "$out/${tag}_${c}_bench.json" 2> "$out/${tag}_${c}_bench.err" || echo "bench $c FAILED"
  echo "bench $c done"
done
python3 bench.py --layout series --no-others > "$out/${tag}_c3_series_major_bench.json" 2> "$out/${tag}_c3_series.err"; echo "bench c3 series done"
prof() {   # name, config, extra flags
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$1/kt" -o out -- python3 bench.py --config $2 --steps 200 --warmup 150 --no-cpu --no-cold --no-others $3 > "$out/${tag}_$1_bench_under_rocprof.json" 2> "$out/prof_$1.err"
  python3 profiles/summarize.py "$out/prof_$1" "$out" "${tag}_$1" filter_ > /dev/null
  rm -rf "$out/prof_$1"
  echo "rocprof $1 done"
}
prof c3 c3 ""
prof c3_cold c3 "--rotate"
prof c3f64 c3f64 ""
prof c3f64_cold c3f64 "--rotate"
prof c4 c4 ""
for c in c2 c2d6 c3d6 c3d6f64 c5; do prof $c $c ""; done
for c in c3 c3f64 c5 c2 c2d6 c3d6 c3d6f64; do
  bash tools/pmc_traffic.sh $c "$out/pmc_$c" > "$out/${tag}_${c}_pmc.json"; rm -rf "$out/pmc_$c"/pmc_*/
  echo "pmc $c done"
done
bash tools/pmc_traffic.sh c3 "$out/pmc_c3_cold" --rotate > "$out/${tag}_c3_cold_pmc.json"; rm -rf "$out/pmc_c3_cold"/pmc_*/
bash tools/pmc_sq.sh c3 "$out/sq_c3" > "$out/${tag}_c3_sq_counters.json"; rm -rf "$out/sq_c3"/g*/
python3 tools/bench_rows.py > "$out/rows_${tag}.jsonl" 2> "$out/rows_${tag}.err" || echo "rows FAILED"
echo "all done"
