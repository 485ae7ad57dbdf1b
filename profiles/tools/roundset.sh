#!/bin/bash
# usage: scratch/roundset.sh TAG   (GPU box, repo root): the profile set of a round checkpoint
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_c2 -- python $R/bench.py --no-cpu-baseline --no-one-gpu > $R/gpurun_out/${tag}_c2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_c4 -- python $R/bench.py --no-cpu-baseline --no-spectrum --no-config5 --no-config1 --no-config4 --steps 10 --warmup 2 --receivers 128 --rate 61440000 --mix mixed > $R/gpurun_out/${tag}_c4.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -- python $R/bench.py --no-cpu-baseline --no-spectrum --no-one-gpu --steps 3 --warmup 1 > $R/gpurun_out/${tag}_fetch.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -- python $R/bench.py --no-cpu-baseline --no-spectrum --no-one-gpu --steps 3 --warmup 1 > $R/gpurun_out/${tag}_write.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/${tag}_sq -- python $R/bench.py --no-cpu-baseline --no-spectrum --no-one-gpu --steps 3 --warmup 1 > $R/gpurun_out/${tag}_sq.log 2>&1 &&
echo roundset done
