#!/bin/bash
# round-4 GPU job 6: SQ counters of the LDS-DMA sweep (both layouts), tiled parity again, the bench line with the tiled headline
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j6
O=gpurun_out/j6
echo "== tiled parity" | tee $O/progress.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "segment_major" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -3 $O/tests.log
echo "== bench c3 (no others)" | tee -a $O/progress.log
timeout -k 10 600 python bench.py --no-others > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?" | tee -a $O/progress.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/j6/bench_c3.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_with_event_pairs","nll_rel_err","filtered_mean_rel_err")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","frac_resident","kernel_ms","kernel_ms_resident","other_layout")})
print(d["config"]["layout"]); c=d.get("cpu_baseline",{}); print({k:c.get(k) for k in ("value","cores","ns_per_step_per_thread","cycles_per_step_per_thread","host")}, c.get("generic_loop",{}).get("value"))
PY
timeout -k 10 600 python bench.py --config c4 --no-others --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; python -c "
import json; d=json.load(open('gpurun_out/j6/bench_c4.json')); print('c4', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['layout'])"
echo "== SQ counters" | tee -a $O/progress.log
export MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_tuning.so
bash tools/pmc_sq_cmd.sh $O/sq_series filter_dma_kernel tools/kbench.py --dtype f32 --variants 0 --rounds 1 --per 6 > $O/sq_series.json 2> $O/sq_series.err; cat $O/sq_series.json
bash tools/pmc_sq_cmd.sh $O/sq_f64 filter_dma_kernel tools/kbench.py --dtype f64 --variants 0 --rounds 1 --per 6 > $O/sq_f64.json 2> $O/sq_f64.err; cat $O/sq_f64.json
rm -rf $O/sq_series/g* $O/sq_f64/g*
unset MOIHGP_LIB
echo "== done" | tee -a $O/progress.log
