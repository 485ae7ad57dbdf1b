#!/bin/bash
# usage (GPU box, repo root): profiles/tools/fftprof.sh TAG  -- spectrum parity tests, then kernel stats of the spectrum-only loop
tag=$1
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 300 python -u -m pytest tests/test_gpu_spectrum.py tests/test_gpu_graphs2.py -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/${tag}_t.log 2>&1 || { tail -30 $R/gpurun_out/${tag}_t.log; exit 1; }
tail -2 $R/gpurun_out/${tag}_t.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python $R/profiles/tools/fftbench.py 8 64 > $R/gpurun_out/${tag}_wall.log 2>&1 && cat $R/gpurun_out/${tag}_wall.log &&
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_k -- python $R/profiles/tools/fftbench.py 8 16 > $R/gpurun_out/${tag}_k.log 2>&1 &&
python $R/profiles/tools/kstats.py $R/gpurun_out/${tag}_k | head -8
