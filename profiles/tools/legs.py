"""Print the step time of each bench leg from bench.py's JSON line on stdin."""
import json, sys
for l in sys.stdin:
    l = l.strip()
    if not l.startswith("{"): continue
    d = json.loads(l)
    print("c2 %.4f ms (fe %.3f)" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]), end="")
    for k in ("config4_slice", "config5_slice"):
        if k in d: print("  %s %.4f ms (fe %.3f)" % (k, d[k]["ms_per_step"], d[k]["roofline"]["avg_launch_ms"]), end="")
    if "spectrum" in d: print("  spectrum %.0f frames/s" % d["spectrum"]["frames_per_s"], end="")
    print()
