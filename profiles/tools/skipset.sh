#!/bin/bash
# LDS counters and duration of k_frontend_wave for the FE_SKIP variants (profiles/tools/libqk_skipN.bin)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1}; do
  export QK_ENGINE_LIB=$R/profiles/tools/libqk_skip$v.bin
  timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/skip${v}_pmc -- python $R/profiles/tools/hostpace.py 128 61440000 mixed 3 > /dev/null 2>&1 || { echo "pmc pass $v failed"; exit 1; }
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/skip${v}_k -- python $R/profiles/tools/hostpace.py 128 61440000 mixed 8 > /dev/null 2>&1 || { echo "trace pass $v failed"; exit 1; }
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$R/gpurun_out/skip${v}_pmc/*/*counter_collection.csv")[0])):
    if "k_frontend_wave" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in acc.items()}
dur = [float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(glob.glob("$R/gpurun_out/skip${v}_k/*/*kernel_stats.csv")[0])) if "k_frontend_wave" in r["Name"]][0]
print("skip %3d: %7.1f us  LDS active %6.1f M  conflict %6.1f M  LDS instr %5.1f M  VALU %6.1f M" % ($v, dur, m.get("SQ_LDS_IDX_ACTIVE", 0) / 1e6, m.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6, m.get("SQ_INSTS_LDS", 0) / 1e6, m.get("SQ_INSTS_VALU", 0) / 1e6))
PY
done
