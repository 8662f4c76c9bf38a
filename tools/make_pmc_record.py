#!/usr/bin/env python3
"""Assemble profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from the per-config PMC summaries of a measurement pass,
stamped with the digest of the kernel sources they were collected against (bench.csrc_digest()).
usage: python tools/make_pmc_record.py <dir with <prefix><config>_pmc.json> <prefix> <commit> "<how collected>" """
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest
d, prefix, commit, how = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
rec = {"_commit": commit, "_collected": how, "_csrc_sha256": csrc_digest()}
for cfg in ("c3", "c3_cold", "c3f64", "c5", "c2", "c2d6", "c3d6", "c3d6f64"):
    p = os.path.join(d, f"{prefix}{cfg}_pmc.json")
    if not os.path.exists(p):
        continue
    s = json.load(open(p))
    rec[cfg] = {"hbm_bytes_per_launch": s["hbm_bytes_per_launch"], "FETCH_SIZE_KiB": s["FETCH_SIZE"]["mean_KiB"], "WRITE_SIZE_KiB": s["WRITE_SIZE"]["mean_KiB"],
                "kernel": s["FETCH_SIZE"]["kernel"], "source": os.path.relpath(p, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}
json.dump(rec, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v if k.startswith("_") else v["hbm_bytes_per_launch"]) for k, v in rec.items()}, indent=1))
