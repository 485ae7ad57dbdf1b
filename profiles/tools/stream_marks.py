#!/usr/bin/env python3
"""Reads the csv QK_TIMELINE=<file> writes (eight stream marks per block, ms from the first) and prints, per stage, its
mean duration, its mean start-to-start period and the mean waits between the stages."""
import sys, csv
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
a = {k: np.array([float(r[k]) for r in rows]) for k in rows[0]}
def per(k): return np.diff(a[k]).mean()
print("blocks", len(rows), " period (fe_start)", round(per("fe_start"), 4), " (agc_end)", round(per("agc_end"), 4))
for name, s, t in [("front end", "fe_start", "fe_end"), ("chain", "chain_start", "chain_end"), ("gain pass", "agc_start", "gain_end"),
                   ("apply+out", "gain_end", "agc_end"), ("tails etc.", "chain_end", "block_end")]:
    d = a[t] - a[s]
    print(f"{name:11s} mean {d.mean():.4f}  min {d.min():.4f}  max {d.max():.4f}")
print("fe_end -> chain_start   ", round((a["chain_start"] - a["fe_end"]).mean(), 4))
print("chain_end -> agc_start  ", round((a["agc_start"] - a["chain_end"]).mean(), 4))
print("fe_end[k] -> fe_start[k+1]", round((a["fe_start"][1:] - a["fe_end"][:-1]).mean(), 4))
print("block_end[k] -> chain_start[k+1]", round((a["chain_start"][1:] - a["block_end"][:-1]).mean(), 4))
print("agc_end[k] -> agc_start[k+1]", round((a["agc_start"][1:] - a["agc_end"][:-1]).mean(), 4))
print("gain_end[k] -> agc_start[k+1]", round((a["agc_start"][1:] - a["gain_end"][:-1]).mean(), 4))
print("latency fe_start -> agc_end", round((a["agc_end"] - a["fe_start"]).mean(), 4))
if len(sys.argv) > 2:
    for i in range(min(int(sys.argv[2]), len(rows))):
        print(" ".join(f"{a[k][i] - a['fe_start'][0]:8.4f}" for k in list(rows[0])[1:]))
