#!/bin/bash
# HBM traffic of the dominant kernel of a bench config from the PMC counters, one counter per pass as MI355X_MICROARCH.md
# prescribes (FETCH_SIZE, WRITE_SIZE in KiB; FETCH doubled on gfx950).  usage: tools/pmc_traffic.sh <config> <outdir>
set -e
cfg=$1; out=$2; EXTRA=$3      # $3: extra bench flags, e.g. --rotate
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$c" -o out -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu --no-cold --no-others $EXTRA > /dev/null 2> "$out/pmc_$c.err"
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True))[-1]
    groups = collections.defaultdict(list)          # one group per kernel instantiation: a sweep may be several launches (second passes)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and ("filter_x_" in r["Kernel_Name"] or "filter_scan_kernel" in r["Kernel_Name"] or "filter_dma_kernel" in r["Kernel_Name"]):
            groups[r["Kernel_Name"]].append(r)
    name, rows = max(groups.items(), key=lambda kv: sum(float(r["Counter_Value"]) for r in kv[1]))     # the dominant one
    v = [float(r["Counter_Value"]) for r in rows]
    res[c] = dict(launches=len(v), mean_KiB=sum(v) / len(v), kernel=name[:80], vgpr=rows[0]["VGPR_Count"], lds=rows[0]["LDS_Block_Size"], scratch=rows[0]["Scratch_Size"],
                  other_instantiations_mean_KiB={k[:80]: sum(float(r["Counter_Value"]) for r in g) / len(g) for k, g in groups.items() if k != name})
res["hbm_bytes_per_launch"] = (2.0 * res["FETCH_SIZE"]["mean_KiB"] + res["WRITE_SIZE"]["mean_KiB"]) * 1024.0
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
