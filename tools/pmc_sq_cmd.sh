#!/bin/bash
# SQ activity counters of one kernel of any python command (4 counters per pass, own runs).
# usage: tools/pmc_sq_cmd.sh <outdir> <kernel-name-substring> <script.py> [args ...]
set -e
out=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/g$i" -o out -- python3 "$@" > /dev/null 2> "$out/g$i.err" || echo "group $i failed (see $out/g$i.err)"
done
python3 - "$out" "$kern" <<'PY'
import csv, glob, json, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
per = collections.defaultdict(list)
for (name, disp), v in acc.items():
    per[name].append(sum(v))
res = {k: sum(v) / len(v) for k, v in per.items()}
if "SQ_WAVES" in res and "SQ_WAVE_CYCLES" in res:
    res["valu_active_share_of_wave_life"] = res.get("SQ_ACTIVE_INST_VALU", 0) / res["SQ_WAVE_CYCLES"]
    res["wait_share_of_wave_life"] = res.get("SQ_WAIT_ANY", 0) / res["SQ_WAVE_CYCLES"]
    res["valu_insts_per_wave"] = res.get("SQ_INSTS_VALU", 0) / res["SQ_WAVES"]
res["kernel"] = kern
json.dump(res, open(f"{out}/sq.json", "w"), indent=1)
print(json.dumps(res))
PY
