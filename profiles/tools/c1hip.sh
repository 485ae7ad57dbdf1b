cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/c1hip -- python $R/profiles/tools/c1trace.py 300 > $R/gpurun_out/c1hip.log 2>&1
