#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/prof/...) into the small summaries kept under profiles/.

usage: python profiles/summarize.py <prof_dir> <out_dir> <tag> [kernel_substring]
  <prof_dir>/kt/**/*_kernel_stats.csv            from `rocprofv3 --kernel-trace --stats`
  <prof_dir>/pmc_fetch/**/*_counter_collection.csv  from `rocprofv3 --pmc FETCH_SIZE` (own pass)
  <prof_dir>/pmc_write/**/*_counter_collection.csv  from `rocprofv3 --pmc WRITE_SIZE` (own pass)
HBM bytes per launch follow MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of a wide coalesced streaming read, so the read side is doubled.
"""
import csv, glob, json, os, sys

prof, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
kname = sys.argv[4] if len(sys.argv) > 4 else "filter_scan_kernel"
os.makedirs(out, exist_ok=True)
ks = sorted(glob.glob(os.path.join(prof, "kt", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
summary = {}
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    for r in rows:
        if kname in r["Name"]:
            summary.setdefault("kernels", []).append(dict(name=r["Name"][:120], calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"])))
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cs = sorted(glob.glob(os.path.join(prof, d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not cs:
        continue
    sel = [r for r in csv.DictReader(open(cs[0])) if r["Counter_Name"] == c and kname in r["Kernel_Name"]]
    if sel:
        v = [float(r["Counter_Value"]) for r in sel]
        summary[c] = dict(launches=len(v), mean_KiB=sum(v) / len(v), vgpr=int(sel[0]["VGPR_Count"]), agpr=int(sel[0]["Accum_VGPR_Count"]),
                          sgpr=int(sel[0]["SGPR_Count"]), lds_bytes=int(sel[0]["LDS_Block_Size"]), scratch=int(sel[0]["Scratch_Size"]))
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    summary["hbm_bytes_per_launch"] = (2.0 * summary["FETCH_SIZE"]["mean_KiB"] + summary["WRITE_SIZE"]["mean_KiB"]) * 1024.0
    summary["note"] = "read side = 2 x FETCH_SIZE (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section)"
json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
