"""Summarise the PMC passes of profiles/tools/pmcset.sh: per kernel, the mean per launch of every counter, with the gfx950
corrections of MI355X_MICROARCH.md (FETCH_SIZE counts half of the bytes of wide streaming reads; it and WRITE_SIZE are in KiB).
usage: pmcsum.py gpurun_out TAG cfg kernel-substring [kernel-substring ...]  -> JSON on stdout"""
import collections
import csv
import glob
import json
import sys

root, tag, cfg = sys.argv[1:4]
subs = sys.argv[4:]
out = {}
for sub in subs:
    acc = collections.defaultdict(list)
    name = None
    for p in ("fetch", "write", "tcc", "sq"):
        files = glob.glob("%s/%s_%s_%s/*/*counter_collection.csv" % (root, tag, cfg, p))
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if sub in r["Kernel_Name"]:
                name = r["Kernel_Name"]
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    d = {"kernel": name, "launches_seen": len(next(iter(acc.values()))), "counters_mean_per_launch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        d["hbm_read_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2      # KiB; x2: gfx950 tallies 128-B requests as 64 B
        d["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    if "TCC_HIT_sum" in m:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    out[sub] = d
print(json.dumps(out, indent=1))
