#!/bin/bash
# usage (GPU box, repo root): profiles/tools/checkpoint.sh TAG -- full GPU suite, the default bench line, kernel-time summaries
# (rocprofv3 --kernel-trace --stats) of the default bench command and of the C4 slice, then the PMC passes.
tag=$1
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/${tag}_tests.log 2>&1 || { tail -30 $R/gpurun_out/${tag}_tests.log; exit 1; }
tail -2 $R/gpurun_out/${tag}_tests.log
timeout -k 10 600 python bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err || { tail -5 $R/gpurun_out/${tag}_bench.err; exit 1; }
python profiles/tools/legs.py < $R/gpurun_out/${tag}_bench.json
bash profiles/tools/roundset.sh $tag || exit 1
bash profiles/tools/pmcset.sh $tag || exit 1
echo checkpoint done
