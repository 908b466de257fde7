#!/usr/bin/env python3
"""Fold the rocprofv3 output of tools/collect_profiles.sh into the three files profiles/ keeps per
round: <tag>_kernel_stats.csv, <tag>_pmc_per_dispatch.json, <tag>_traffic.json."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.dirname(out)
ks = glob.glob(os.path.join(out, "trace", "**", "*_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))

OURS = ("fps_", "k1_", "pcp_", "crop_", "gather", "pack", "sor_", "vox_", "rocprim")
acc = defaultdict(lambda: defaultdict(list))
for sub in ("fetch", "write", "sq", "sq2", "l2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if not any(t in name for t in OURS):
                continue
            short = name.split("(")[0]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
pmc = {k: {c: {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)} for c, v in cs.items()}
       for k, cs in acc.items()}
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_per_dispatch.json"), "w"), indent=1, sort_keys=True)

fps = [k for k in pmc if "fps_bucket" in k and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]]
if fps:
    # the launch may consist of more than one fps kernel (wide + deep shapes): sum them
    fetch_kb = sum(pmc[k]["FETCH_SIZE"]["mean_per_dispatch"] for k in fps)
    write_kb = sum(pmc[k]["WRITE_SIZE"]["mean_per_dispatch"] for k in fps)
    json.dump({"kernel": " + ".join(sorted(fps)), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
               "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 "
                       "--warmup 1`; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half the bytes "
                       "of 16 B/lane coalesced reads, so reads are counted as 2 x FETCH_SIZE; KB = 1024 B; "
                       "Infinity-Cache hits are included in these fabric-side counters",
               "frames_per_launch": 300}, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
l2 = {k: {c: v[c]["mean_per_dispatch"] for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum") if c in v} for k, v in pmc.items()}
l2 = {k: v for k, v in l2.items() if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v}
for v in l2.values():
    tot = v["TCC_HIT_sum"] + v["TCC_MISS_sum"]
    v["l2_hit_rate"] = v["TCC_HIT_sum"] / tot if tot else None
    v["miss_bytes_at_128B"] = v["TCC_MISS_sum"] * 128
if l2:
    json.dump({"command": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- python3 bench.py --steps 2 --warmup 1 (own pass)",
               "mean_per_dispatch": l2}, open(os.path.join(dst, f"{tag}_l2.json"), "w"), indent=1, sort_keys=True)
print("summarised", sorted(pmc))
