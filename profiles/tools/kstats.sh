#!/bin/bash
# kernel-time summary of one hostpace run: profiles/tools/kstats.sh <tag> <hostpace args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python $GRAFT_REPO_ROOT/profiles/tools/hostpace.py "$@" > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:${TOP:-8}]: print("%-62s %6s %10.1f us %6s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
