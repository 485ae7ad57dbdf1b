#!/bin/bash
# kernel-time summary of the config-1 loop (profiles/tools/c1trace.py): profiles/tools/c1k.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$1 -- python $R/profiles/tools/c1trace.py 600 > $R/gpurun_out/$1.log 2>&1 || exit 1
python3 $R/profiles/tools/kstats.py $R/gpurun_out/$1 16
