#!/bin/bash
# round-4 GPU job 10: the new many-latent oracle test, the bench in its profiled forms (resident / --rotate), tests that call bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j10
O=gpurun_out/j10
echo "== tests" | tee $O/progress.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "many_latent or segment_major or bench or full_size_properties" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -4 $O/tests.log
echo "== bench c3 resident / rotate (no others)" | tee -a $O/progress.log
timeout -k 10 300 python bench.py --no-others --no-cpu --no-cold --steps 100 > $O/b_res.json 2> $O/b_res.err; echo "rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-others --no-cpu --no-cold --steps 100 --rotate > $O/b_rot.json 2> $O/b_rot.err; echo "rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-others --no-cpu > $O/b_def.json 2> $O/b_def.err; echo "rc=$?" >> $O/progress.log
python - <<'PY'
import json
for n in ("b_res","b_rot","b_def"):
    d=json.load(open(f"gpurun_out/j10/{n}.json")); r=d["roofline"]
    print(n, "value %.3e ms %.4f"%(d["value"],d["ms_per_step"]), {k:(round(r[k],4) if isinstance(r.get(k),float) else r.get(k)) for k in ("frac","frac_resident","kernel_ms","kernel_ms_resident","frac_is")})
PY
echo "== done" | tee -a $O/progress.log
