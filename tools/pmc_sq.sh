#!/bin/bash
# SQ activity counters of a bench config (4 per pass, own runs).  usage: tools/pmc_sq.sh <config> <outdir>
set -e
cfg=$1; out=$2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/g$i" -o out -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-cold --no-others > /dev/null 2> "$out/g$i.err"
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_x_" in r["Kernel_Name"] or "filter_scan_kernel" in r["Kernel_Name"] or "filter_dma_kernel" in r["Kernel_Name"]:
            acc[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
per = collections.defaultdict(list)
for (name, disp), v in acc.items():
    per[name].append(sum(v))
res = {k: sum(v) / len(v) for k, v in per.items()}
if "SQ_WAVES" in res and "SQ_WAVE_CYCLES" in res:
    res["valu_active_share_of_wave_life"] = res.get("SQ_ACTIVE_INST_VALU", 0) / res["SQ_WAVE_CYCLES"]
    res["wait_share_of_wave_life"] = res.get("SQ_WAIT_ANY", 0) / res["SQ_WAVE_CYCLES"]
json.dump(res, open(f"{out}/sq.json", "w"), indent=1)
print(json.dumps(res))
PY
