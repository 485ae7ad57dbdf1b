#!/bin/bash
# kernel-time summary of the bench's headline leg: profiles/tools/kbench.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-spectrum --no-config4 --no-config5 --no-config1 --steps 40 --warmup 5 "$@" > /dev/null 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/profiles/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/$tag ${TOP:-6}
